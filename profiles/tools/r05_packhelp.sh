#!/bin/bash
# r05_packhelp.sh <tag> -- host-boundary step with the waiting lane drivers helping the pack job of the lane whose turn it is (default) against sleeping (LF_PACK_HELP=0), A B A B
OUT=gpurun_out/${1:-r05packhelp}; mkdir -p $OUT
for k in 1 2 3; do for M in 1 0; do
  LF_PACK_HELP=$M timeout 600 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-exclusive > $OUT/b_$M.$k.json 2> $OUT/b_$M.$k.err
  python3 - $OUT/b_$M.$k.json $M <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('drivers help the pack' if sys.argv[2]=='1' else 'drivers sleep         ','| host boundary',round(d['value']),round(d['ms_per_step'],1),'ms; cpu/step',round(d['host_cpu_seconds_per_step'],3),'digest',(d.get('sam_digests') or {}).get('host_boundary_timed_steps',{}).get('xxh3_128'))
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done; done
LF_TIMING=1 timeout 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive 2>&1 >/dev/null | grep "timeline" | tail -8 | cut -c 1-200
