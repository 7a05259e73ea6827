#!/bin/bash
# r05_smallchunks.sh <tag> -- a 12.5 k / 25 k-read shard cut into chunks of LF_CHUNK_READS reads (default: at least 6250 per chunk), alone and two shards in flight
OUT=gpurun_out/${1:-r05smallchunks}; mkdir -p $OUT
for N in 12500 25000; do for C in 0 1600 2100 3200 4200; do for D in 1 2; do
  if [ $C = 0 ]; then unset LF_CHUNK_READS; else export LF_CHUNK_READS=$C; fi
  timeout 600 python3 bench.py --reads $N --steps 16 --warmup 2 --inflight $D --no-cpu-baseline --no-exclusive --no-host-region > $OUT/b_${N}_${C}_$D.json 2> $OUT/b.err
  python3 - $OUT/b_${N}_${C}_$D.json $N $C $D <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('reads',sys.argv[2],'chunk reads',sys.argv[3],'in flight',sys.argv[4],'| ms per step',round(d['ms_per_step_hbm_resident'],2),'reads/s',round(d['value_hbm_resident']),'cpu/step',round(d['host_cpu_seconds_per_step_hbm_resident'],3),'chunks/step',d.get('chunks_per_step'))
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done; done; done
