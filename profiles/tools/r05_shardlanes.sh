#!/bin/bash
# r05_shardlanes.sh <tag> -- small shards with several steps in flight: 4 lanes (default for HBM-resident batches) against 8 (LF_LANES)
OUT=gpurun_out/${1:-r05shardlanes}; mkdir -p $OUT
for N in 12500 25000; do for D in 2 4; do for L in 4 8 4 8; do
  LF_LANES=$L timeout 600 python3 bench.py --reads $N --steps 16 --warmup 2 --inflight $D --no-cpu-baseline --no-exclusive --no-host-region > $OUT/b_${N}_${D}_$L.json 2> $OUT/b.err
  python3 - $OUT/b_${N}_${D}_$L.json $N $D $L <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('reads',sys.argv[2],'in flight',sys.argv[3],'lanes',sys.argv[4],'| ms per step',round(d['ms_per_step_hbm_resident'],2),'reads/s',round(d['value_hbm_resident']),'cpu/step',round(d['host_cpu_seconds_per_step_hbm_resident'],3),'chunks/step',d.get('chunks_per_step'))
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done; done; done
