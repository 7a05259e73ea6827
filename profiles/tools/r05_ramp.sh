#!/bin/bash
# r05_ramp.sh <tag> -- host-boundary step per chunk ramp (LF_CHUNK_RAMP, percent) and forward / traceback times of the coalesced 32-step rows
OUT=gpurun_out/${1:-r05ramp}; mkdir -p $OUT
for R in ${RAMPS:-0 25 40 60}; do
  LF_CHUNK_RAMP=$R timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline ${EXTRA:---no-exclusive} > $OUT/bench_ramp$R.json 2> $OUT/bench_ramp$R.err
  python3 - $OUT/bench_ramp$R.json $R <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('ramp',sys.argv[2],'host boundary',round(d['value']),round(d['ms_per_step'],1),'ms; hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms; cpu/step',round(d['host_cpu_seconds_per_step'],3),round(d['host_cpu_seconds_per_step_hbm_resident'],3),'waits/chunk',d.get('host_waits_per_chunk'),'digest',(d.get('sam_digests') or {}).get('host_boundary_timed_steps',{}).get('xxh3_128'))
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
