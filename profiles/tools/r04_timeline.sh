#!/bin/bash
OUT=gpurun_out/${1:-r04t}; mkdir -p $OUT
LF_TIMING=1 timeout 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive > /dev/null 2> $OUT/timing.err
grep -n "lf_map_batch total\|setup (strlen" $OUT/timing.err | tail -8
grep "timeline" $OUT/timing.err | tail -16 | cut -c 1-400
grep "chunk of\|map_chunk\|chunk_free" $OUT/timing.err | tail -40 | cut -c 1-200 > $OUT/chunks.txt
LF_PHASES=1 timeout 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive > /dev/null 2> $OUT/phases.err
grep "phase \|batch of" $OUT/phases.err | tail -60
