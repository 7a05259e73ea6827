#!/bin/bash
# r05_lanes.sh <tag> -- step time and host CPU per number of lanes (chunks in flight), poll budget 3000 / 300 us
OUT=gpurun_out/${1:-r05lanes}; mkdir -p $OUT
for CFG in ${CFGS:-8_3000 4_3000 6_3000 8_300 4_300}; do
  set -- ${CFG/_/ }
  LF_LANES=$1 LF_SPIN_US=$2 timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exclusive > $OUT/bench_l$1_s$2.json 2> $OUT/bench_l$1_s$2.err
  python3 - $OUT/bench_l$1_s$2.json "$CFG" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('lanes, spin us',sys.argv[2],'host boundary',round(d['value']),round(d['ms_per_step'],1),'ms; hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms; cpu/step',round(d['host_cpu_seconds_per_step'],3),round(d['host_cpu_seconds_per_step_hbm_resident'],3),'waits/chunk',round(d.get('host_waits_per_chunk',0),1),'chunks/step',d.get('chunks_per_step'))
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
