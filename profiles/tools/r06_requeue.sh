# failed trials go back with the bound their sweep found: stage tests, then C5 / C4
mkdir -p gpurun_out/r6_requeue
timeout 1500 python3 -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "edlib" 2>&1 | tail -5
for cfg in c5 c4; do
for band in 1 1; do
timeout 900 python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-host-region > gpurun_out/r6_requeue/${cfg}_band${band}_$RANDOM.json 2> gpurun_out/r6_requeue/err.txt
done
for f in gpurun_out/r6_requeue/${cfg}_band*.json; do python3 - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
bk=d["roofline"]["by_kernel"]
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), "hbm-resident", round(d["value_hbm_resident"]), {k.split(' ')[0]:round(v["ms_per_step"],1) for k,v in bk.items() if "hirsch" in k or "hband" in k or "ksw" in k or "rsweep" in k}, d.get("timed_output_equals_exclusive_pass_output"))
PY
done
done
