#!/bin/bash
# lanes with a fixed chunk size, alternating, one box
for i in 1 2; do
 for L in 8 12 6; do
  LF_LANES=$L LF_CHUNK_READS=4167 timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exclusive 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('lanes $L chunk 4167', round(b['value']), round(b['ms_per_step'],1), b['host_cpu_seconds_per_step'])"
 done
done
LF_LANES=8 LF_CHUNK_READS=6250 timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exclusive 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('lanes 8 chunk 6250', round(b['value']), round(b['ms_per_step'],1))"
LF_LANES=8 LF_CHUNK_READS=3125 timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exclusive 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('lanes 8 chunk 3125', round(b['value']), round(b['ms_per_step'],1))"
