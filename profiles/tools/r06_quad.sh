# sixteen-lane Hirschberg queues: stage tests, then C5 / C4 with and without them (LF_HIRSCH_BAND=64: whole wavefronts only)
mkdir -p gpurun_out/r6_quad
timeout 1500 python3 -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "edlib" 2>&1 | tail -5
for band in 1 64 1 64; do
timeout 900 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-host-region > gpurun_out/r6_quad/c5_band${band}_$RANDOM.json 2> gpurun_out/r6_quad/c5_err.txt
done
for f in gpurun_out/r6_quad/c5_band*.json; do python3 - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
bk=d["roofline"]["by_kernel"]
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), "hbm-resident", round(d["value_hbm_resident"]), {k:round(v["ms_per_step"],1) for k,v in bk.items() if "hirsch" in k or "hband" in k}, d.get("timed_output_equals_exclusive_pass_output"))
PY
done
LF_HIRSCH_DEBUG=1 timeout 600 python3 bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive --no-host-region > gpurun_out/r6_quad/c5dbg.json 2> gpurun_out/r6_quad/c5_levels.txt
grep "level 1:" gpurun_out/r6_quad/c5_levels.txt | tail -3 | cut -c1-400
