#!/bin/bash
TAG=${1:-r04f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive"
export LF_WATCHDOG=50 LF_TIMING=1
for mode in full holes; do
  case $mode in
    full) export LF_SAM_FULL=1;;
    holes) unset LF_SAM_FULL;;
  esac
  timeout 300 $B > $OUT/dbg_$mode.json 2> $OUT/dbg_$mode.err
  echo "== $mode rc $?"; tail -c 400 $OUT/dbg_$mode.json; echo; grep -E "watchdog|Error|error|fault" $OUT/dbg_$mode.err | head -24
  grep -n "lane . chunk\|timeline" $OUT/dbg_$mode.err | tail -40 > $OUT/dbg_${mode}_tail.txt
done
