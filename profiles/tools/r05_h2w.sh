#!/bin/bash
# r05_h2w.sh <tag> -- config C4 and C5 (T2T-like) after the two-wavefront Hirschberg class (queries of 4 097 .. 8 192 rows), HBM-resident and host to host; run on the tree before
# and after (the class is not switchable at run time)
OUT=gpurun_out/${1:-r05h2w}; mkdir -p $OUT
for k in 1 2; do
  timeout 900 python3 bench.py --config c4 --steps 6 --warmup 1 --no-cpu-baseline > $OUT/c4.$k.json 2> $OUT/c4.$k.err
  python3 - $OUT/c4.$k.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('c4 | host boundary',round(d['value']),round(d['ms_per_step'],1),'ms; hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms; hirsch exclusive',round([v['ms_per_step'] for k,v in r['by_kernel'].items() if k.startswith('lf_hirsch')][0],1),'digest',(d.get('sam_digests') or {}).get('host_boundary_timed_steps',{}).get('xxh3_128'),'match',d.get('all_records_match_rate'))
PY
done
timeout 900 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/c5.json 2> $OUT/c5.err
python3 - $OUT/c5.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('c5 t2tlike | host boundary',round(d['value']),round(d['ms_per_step'],1),'ms; hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms; hirsch exclusive',round([v['ms_per_step'] for k,v in r['by_kernel'].items() if k.startswith('lf_hirsch')][0],1),'match',d.get('all_records_match_rate'))
PY
