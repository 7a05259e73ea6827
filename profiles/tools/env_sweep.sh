#!/bin/bash
# env_sweep.sh VAR v1 v2 ... -- the HBM-resident bench rate (6 steps) for every value of one environment variable
VAR=$1; shift
mkdir -p gpurun_out; rm -f gpurun_out/env_sweep.log
for v in "$@"; do
  echo "== $VAR=$v" >> gpurun_out/env_sweep.log
  env $VAR=$v timeout 400 python3 bench.py --steps 6 --warmup 1 --no-exclusive --no-cpu-baseline --no-host-region 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],1))
" >> gpurun_out/env_sweep.log
done
cat gpurun_out/env_sweep.log
