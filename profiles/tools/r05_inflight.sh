#!/bin/bash
# r05_inflight.sh <tag> -- host-boundary and HBM-resident rate of 100 k-read steps with 1 / 2 / 3 steps in flight (complete calls from D threads)
OUT=gpurun_out/${1:-r05inflight}; mkdir -p $OUT
for D in ${DS:-1 2 3 1 2}; do
  timeout 600 python3 bench.py --steps 6 --warmup 2 --inflight $D --no-cpu-baseline --no-exclusive > $OUT/bench_inflight$D.json 2> $OUT/bench_inflight$D.err
  python3 - $OUT/bench_inflight$D.json $D <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('in flight',sys.argv[2],'host boundary',round(d['value']),round(d['ms_per_step'],1),'ms; hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms; cpu/step',round(d['host_cpu_seconds_per_step'],3),round(d['host_cpu_seconds_per_step_hbm_resident'],3),'digest',(d.get('sam_digests') or {}).get('host_boundary_timed_steps',{}).get('xxh3_128'))
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
