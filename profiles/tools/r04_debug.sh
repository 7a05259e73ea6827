#!/bin/bash
# r04_debug.sh <tag> -- where does the full-size bench hang?  watchdog + python stacks, bounded by timeouts
TAG=${1:-r04c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export LF_BENCH_STACKS=150
LF_WATCHDOG=100 timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_dbg.json 2> $OUT/bench_dbg.err
echo "rc $?"; tail -c 600 $OUT/bench_dbg.json; grep -v "^\[lf\]" $OUT/bench_dbg.err | tail -60
