# (no LF_WATCHDOG in timing runs)
mkdir -p gpurun_out/r6_s2
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "edlib" 2>&1 | tail -5 > gpurun_out/r6_s2/tests.log; cat gpurun_out/r6_s2/tests.log
LF_HIRSCH_DEBUG=1 timeout 600 python3 profiles/tools/r06_hlat.py > gpurun_out/r6_s2/lat.txt 2> gpurun_out/r6_s2/levels.txt; cat gpurun_out/r6_s2/lat.txt
for band in 1 0; do
  LF_HIRSCH_BAND=$band timeout 900 python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r6_s2/c4_band$band.json 2> gpurun_out/r6_s2/c4_band$band.err
done
timeout 1200 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r6_s2/c5_band1.json 2> gpurun_out/r6_s2/c5_band1.err
python3 - <<'PY'
import json
for f in ('c4_band1','c4_band0','c5_band1'):
    try:
        j=json.loads(open('gpurun_out/r6_s2/%s.json'%f).read().strip().splitlines()[-1])
        print(f, round(j['value']), round(j['ms_per_step'],1), round(j.get('value_hbm_resident',0)), j['sam_digests']['exclusive_pass']['xxh3_128'], j.get('match_rate_all_records'), [ (k, round(v,1)) for k,v in sorted(j['roofline'].get('exclusive_ms_by_group',{}).items(), key=lambda x:-x[1])[:6]])
    except Exception as e: print(f, 'ERR', e)
PY
