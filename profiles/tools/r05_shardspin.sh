#!/bin/bash
# r05_shardspin.sh <tag> -- a rank's shard (12.5 k / 25 k reads, HBM-resident) per poll budget: one step at a time and two in flight
OUT=gpurun_out/${1:-r05shardspin}; mkdir -p $OUT
for R in 12500 25000; do for U in 200 1000 3000; do for D in 1 2; do
  LF_SPIN_US=$U timeout 300 python3 bench.py --reads $R --steps 16 --warmup 2 --inflight $D --no-cpu-baseline --no-exclusive --no-host-region > $OUT/s_${R}_${U}_$D.json 2> $OUT/e.err
  python3 - $OUT/s_${R}_${U}_$D.json $R $U $D <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('reads',sys.argv[2],'spin us',sys.argv[3],'in flight',sys.argv[4],'ms per step',round(d['ms_per_step_hbm_resident'],2),'reads/s',round(d['value_hbm_resident']),'cpu/step',round(d['host_cpu_seconds_per_step_hbm_resident'],3),'waits/chunk',round(d.get('host_waits_per_chunk',0),1),'chunks/step',d.get('chunks_per_step'))
except Exception as e:
    print('FAILED',e)
PY
done; done; done
