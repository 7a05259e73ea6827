#!/bin/bash
# r05_hclass.sh <tag> -- the size classes of a Hirschberg level on three streams (default) against one after the other on one (LF_HIRSCH_CLASS_STREAMS=0): configs C4 and C5 (T2T-like), A B A B
OUT=gpurun_out/${1:-r05hclass}; mkdir -p $OUT
for k in 1 2; do for M in 1 0; do
  LF_HIRSCH_CLASS_STREAMS=$M timeout 900 python3 bench.py --config c4 --steps 6 --warmup 1 --no-cpu-baseline --no-exclusive > $OUT/c4_$M.$k.json 2> $OUT/c4_$M.$k.err
  python3 - $OUT/c4_$M.$k.json $M <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('c4', 'three streams' if sys.argv[2]=='1' else 'one stream   ','| host boundary',round(d['value']),round(d['ms_per_step'],1),'ms; hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms; digest',(d.get('sam_digests') or {}).get('host_boundary_timed_steps',{}).get('xxh3_128'))
PY
done; done
for M in 1 0; do
  LF_HIRSCH_CLASS_STREAMS=$M timeout 900 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive > $OUT/c5_$M.json 2> $OUT/c5_$M.err
  python3 - $OUT/c5_$M.json $M <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('c5 t2tlike', 'three streams' if sys.argv[2]=='1' else 'one stream   ','| host boundary',round(d['value']),round(d['ms_per_step'],1),'ms; hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms')
PY
done
