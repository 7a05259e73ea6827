#!/bin/bash
# r04_sweep_final.sh <tag> -- shard sweep on the round's last tree: HBM-resident steps of 12.5 k / 25 k / 50 k / 100 k reads, 1 or 2 steps in flight
OUT=gpurun_out/${1:-r04sw}; mkdir -p $OUT
B="timeout 300 python3 bench.py --no-cpu-baseline --no-exclusive --no-host-region"
for R in 100000 50000 25000 12500; do for D in 1 2; do
  ST=$((800000 / R)); [ $ST -gt 32 ] && ST=32; [ $ST -lt 6 ] && ST=6
  $B --reads $R --steps $ST --warmup 2 --inflight $D > $OUT/sweep_final_${R}_d$D.json 2> $OUT/sweep_final_${R}_d$D.err
  python3 -c "import json; d=json.loads(open('$OUT/sweep_final_${R}_d$D.json').read().strip().splitlines()[-1]); print('reads $R inflight $D', round(d['value_hbm_resident']), 'reads/s', round(d['ms_per_step_hbm_resident'],2), 'ms', 'cpu', round(d['host_cpu_seconds_per_step_hbm_resident'],3))" 2>/dev/null || echo "reads $R inflight $D FAILED"
done; done
