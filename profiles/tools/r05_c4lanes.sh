OUT=gpurun_out/r05c4lanes; mkdir -p $OUT
for L in 4 8 6 4 8; do
  LF_LANES=$L timeout 900 python3 bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline --no-exclusive --no-host-region > $OUT/c4_l$L.json 2> $OUT/e.err
  python3 -c "
import json; d=json.loads(open('$OUT/c4_l$L.json').read().strip().splitlines()[-1]); print('c4 HBM-resident, lanes $L:', round(d['value_hbm_resident']), round(d['ms_per_step_hbm_resident'],1), 'ms; cpu/step', round(d['host_cpu_seconds_per_step_hbm_resident'],3), 'chunks/step', d.get('chunks_per_step'))"
done
