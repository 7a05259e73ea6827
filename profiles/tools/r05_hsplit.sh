#!/bin/bash
# r05_hsplit.sh <tag> -- nodes above 16 384 rows as two workgroups (one per half; default) against one workgroup of sixteen wavefronts (LF_HIRSCH_SPLIT=0): configs C4 and C5 (T2T-like)
OUT=gpurun_out/${1:-r05hsplit}; mkdir -p $OUT
for k in 1 2; do for M in 1 0; do
  LF_HIRSCH_SPLIT=$M timeout 900 python3 bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline --no-exclusive > $OUT/c4_$M.$k.json 2> $OUT/c4_$M.$k.err
  python3 - $OUT/c4_$M.$k.json $M <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('c4', 'two workgroups per node' if sys.argv[2]=='1' else 'one workgroup per node ','| host boundary',round(d['value']),round(d['ms_per_step'],1),'ms; hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms; digest',(d.get('sam_digests') or {}).get('host_boundary_timed_steps',{}).get('xxh3_128'))
PY
done; done
for M in 1 0; do
  LF_HIRSCH_SPLIT=$M timeout 900 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive > $OUT/c5_$M.json 2> $OUT/c5_$M.err
  python3 - $OUT/c5_$M.json $M <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('c5 t2tlike', 'two workgroups per node' if sys.argv[2]=='1' else 'one workgroup per node ','| host boundary',round(d['value']),round(d['ms_per_step'],1),'ms; hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms; digest',(d.get('sam_digests') or {}).get('host_boundary_timed_steps',{}).get('xxh3_128'))
PY
done
