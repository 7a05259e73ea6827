#!/bin/bash
# r05_ramp2.sh <tag> -- host-boundary step: chunks per lane x ramp
OUT=gpurun_out/${1:-r05ramp2}; mkdir -p $OUT
for CFG in "1 25" "2 25" "2 40" "2 0" "1 25"; do
  set -- $CFG
  LF_CHUNKS_PER_LANE=$1 LF_CHUNK_RAMP=$2 timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exclusive > $OUT/bench_pl$1_ramp$2.json 2> $OUT/bench_pl$1_ramp$2.err
  python3 - $OUT/bench_pl$1_ramp$2.json "$CFG" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('chunks/lane, ramp',sys.argv[2],'host boundary',round(d['value']),round(d['ms_per_step'],1),'ms; hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms; cpu/step',round(d['host_cpu_seconds_per_step'],3),round(d['host_cpu_seconds_per_step_hbm_resident'],3),'digest',(d.get('sam_digests') or {}).get('host_boundary_timed_steps',{}).get('xxh3_128'))
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
