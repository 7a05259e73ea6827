#!/bin/bash
TAG=${1:-r04d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
B="python3 bench.py --genome-mbp 100 --reads 20000 --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive"
export LF_WATCHDOG=40
for mode in full holes1lane holes; do
  case $mode in
    full) export LF_SAM_FULL=1; unset LF_LANES;;
    holes1lane) unset LF_SAM_FULL; export LF_LANES=1;;
    holes) unset LF_SAM_FULL; unset LF_LANES;;
  esac
  timeout 240 $B > $OUT/dbg_$mode.json 2> $OUT/dbg_$mode.err
  echo "== $mode rc $?"; tail -c 300 $OUT/dbg_$mode.json; echo; grep -E "watchdog|Error|error|fault" $OUT/dbg_$mode.err | head -20
done
