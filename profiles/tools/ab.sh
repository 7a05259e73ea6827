#!/bin/bash
# ab.sh VAR -- A/B of an environment switch on ONE box: bench with VAR unset / VAR=1, alternating, 3 rounds
V=${1:?env var}
for i in 1 2 3; do
  for mode in off on; do
    if [ $mode = on ]; then export $V=1; else unset $V; fi
    timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exclusive 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$V', '$mode', round(b['value']), round(b['ms_per_step'],1))"
  done
done
