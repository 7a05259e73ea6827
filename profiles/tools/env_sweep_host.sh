#!/bin/bash
# env_sweep_host.sh VAR v1 v2 ... -- HBM-resident and host-buffer bench rates (5 steps) for every value of one environment variable
VAR=$1; shift
mkdir -p gpurun_out; rm -f gpurun_out/env_sweep.log
for v in "$@"; do
  echo "== $VAR=$v" >> gpurun_out/env_sweep.log
  env $VAR=$v timeout 400 python3 bench.py --steps 5 --warmup 1 --no-exclusive --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],1), 'host', round(d['value_pcie_inclusive']), round(d['ms_per_step_pcie_inclusive'],1))
" >> gpurun_out/env_sweep.log
done
cat gpurun_out/env_sweep.log
