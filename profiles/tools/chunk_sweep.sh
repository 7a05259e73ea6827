#!/bin/bash
# chunk-size / lane-count sweep of the bench line (HBM-resident and host-buffer rates), one index build per setting
mkdir -p gpurun_out
for cfg in ${LF_SWEEP:-6250:8 12500:8 12500:4 8334:6 25000:4 4167:8}; do
  set -- ${cfg/:/ }
  echo "== LF_CHUNK_READS=$1 LF_LANES=$2" >> gpurun_out/chunk_sweep.log
  LF_CHUNK_READS=$1 LF_LANES=$2 timeout 400 python3 bench.py --steps 6 --warmup 1 --no-exclusive --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), round(d['ms_per_step'],1), round(d.get('value_pcie_inclusive') or 0))
" >> gpurun_out/chunk_sweep.log
done
cat gpurun_out/chunk_sweep.log
