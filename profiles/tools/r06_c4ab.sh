# (no LF_WATCHDOG in timing runs)
mkdir -p gpurun_out/r6_c4
for band in 1 0; do
  LF_HIRSCH_BAND=$band timeout 900 python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r6_c4/bench_band$band.json 2> gpurun_out/r6_c4/bench_band$band.err
  tail -c 600 gpurun_out/r6_c4/bench_band$band.err
done
LF_HIRSCH_DEBUG=1 timeout 600 python3 bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive --no-host-region > gpurun_out/r6_c4/dbg.json 2> gpurun_out/r6_c4/levels_band1.txt
python3 - <<'PY'
import json
for b in (1,0):
    try:
        j=json.loads(open('gpurun_out/r6_c4/bench_band%d.json'%b).read().strip().splitlines()[-1])
        print(b, j['value'], j['ms_per_step'], j.get('value_hbm_resident'), {k:v for k,v in j.get('exclusive_ms',{}).items()} if 'exclusive_ms' in j else '', j.get('sam_digests'))
    except Exception as e: print(b, 'ERR', e)
PY
