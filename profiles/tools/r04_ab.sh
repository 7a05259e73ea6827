#!/bin/bash
# r04_ab.sh <tag> <VAR=value> -- A/B of one environment setting on ONE box, alternating, 3 rounds (C2 default bench, no CPU baseline / exclusive pass)
TAG=${1:-r04s}; SET=${2:-LF_UPLOAD_TURNS=0}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in 1 2 3; do for mode in default set; do
  if [ $mode = set ]; then export $SET; else unset ${SET%%=*}; fi
  timeout 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-exclusive 2>$OUT/ab_${mode}_$i.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', '$SET' if '$mode'=='set' else '', 'value', round(d['value']), round(d['ms_per_step'],2), 'hbm', round(d['value_hbm_resident']), round(d['ms_per_step_hbm_resident'],2), 'cpu', round(d['host_cpu_seconds_per_step'],3))"
done; done
