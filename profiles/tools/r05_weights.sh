#!/bin/bash
# r05_weights.sh <tag> -- host-boundary step per chunk-size profile (LF_CHUNK_WEIGHTS; "ramp" = the default linear ramp of 25 %)
OUT=gpurun_out/${1:-r05weights}; mkdir -p $OUT
i=0
for W in ${WEIGHTS:-ramp 0.75,0.85,1.0,1.15,1.25,1.2,0.95,0.85 ramp 0.8,0.9,1.05,1.2,1.25,1.1,0.9,0.8 0.75,0.9,1.05,1.2,1.3,1.2,0.9,0.7}; do
  i=$((i+1))
  if [ $W = ramp ]; then unset LF_CHUNK_WEIGHTS; else export LF_CHUNK_WEIGHTS=$W; fi
  timeout 600 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-exclusive > $OUT/b_$i.json 2> $OUT/b_$i.err
  python3 - $OUT/b_$i.json $W <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('weights',sys.argv[2],'| host boundary',round(d['value']),round(d['ms_per_step'],1),'ms; cpu/step',round(d['host_cpu_seconds_per_step'],3),'digest',(d.get('sam_digests') or {}).get('host_boundary_timed_steps',{}).get('xxh3_128'))
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
