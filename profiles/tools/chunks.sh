#!/bin/bash
for i in 1 2; do
 for C in ${CHUNKS:-6250 8334 10000 12500}; do
  LF_LANES=${LANES:-8} LF_CHUNK_READS=$C timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exclusive 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('lanes ${LANES:-8} chunk $C', round(b['value']), round(b['ms_per_step'],1), b['host_cpu_seconds_per_step'])"
 done
done
