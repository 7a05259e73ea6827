#!/bin/bash
for i in 1 2; do
 for Q in ${QS:-16 8 24 32}; do
  GPU_MAX_HW_QUEUES=$Q timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exclusive 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('hw queues $Q', round(b['value']), round(b['ms_per_step'],1))"
 done
done
