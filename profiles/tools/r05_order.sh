#!/bin/bash
# r05_order.sh <tag> -- forward sweep / traceback workgroups handed out long problems first (default) against ascending order (LF_ALIGN_LONG_FIRST=0): HBM-resident steps
# of 12.5 k and 100 k reads, A B A B on one box
OUT=gpurun_out/${1:-r05order}; mkdir -p $OUT
for N in 12500 100000; do for k in 1 2; do for M in 1 0; do
  LF_ALIGN_LONG_FIRST=$M timeout 600 python3 bench.py --reads $N --steps 8 --warmup 2 --no-cpu-baseline --no-exclusive --no-host-region > $OUT/b_${N}_$M.$k.json 2> $OUT/b_${N}_$M.$k.err
  python3 - $OUT/b_${N}_$M.$k.json $N $M <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('reads',sys.argv[2],'long first' if sys.argv[3]=='1' else 'ascending ','| hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],2),'ms; digest',(d.get('sam_digests') or {}).get('hbm_resident_timed_steps',{}).get('xxh3_128'))
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done; done; done
