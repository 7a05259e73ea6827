#!/bin/bash
# r05_timeline.sh <tag> -- per-chunk stage timelines (LF_TIMING) of the last host-boundary step, chunk ramp 0 and 25 %
OUT=gpurun_out/${1:-r05tl}; mkdir -p $OUT
for R in 0 25; do
  LF_CHUNK_RAMP=$R LF_TIMING=1 timeout 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive > /dev/null 2> $OUT/timing_ramp$R.err
  echo "== ramp $R"; grep -n "lf_map_batch total" $OUT/timing_ramp$R.err | tail -3
  grep "timeline\|map_chunk \|egress\|scatter kernels\|lf_map_batch total" $OUT/timing_ramp$R.err | tail -34 | cut -c 1-330 > $OUT/timeline_ramp$R.txt; cat $OUT/timeline_ramp$R.txt
done
