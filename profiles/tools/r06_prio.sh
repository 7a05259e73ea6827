# wave priority of the long-chain Hirschberg kernels (s_setprio): C5 / C4 with (LF_HIRSCH_BAND=1) and without (=3)
mkdir -p gpurun_out/r6_prio
for cfg in c5 c4; do
for band in 1 3 1 3; do
timeout 900 python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-host-region > gpurun_out/r6_prio/${cfg}_band${band}_$RANDOM.json 2> gpurun_out/r6_prio/err.txt
done
for f in gpurun_out/r6_prio/${cfg}_band*.json; do python3 - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
bk=d["roofline"]["by_kernel"]
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), "hbm-resident", round(d["value_hbm_resident"]), {k.split(' ')[0]:round(v["ms_per_step"],1) for k,v in bk.items() if "hirsch" in k or "hband" in k or "ksw" in k or "rsweep" in k}, d.get("timed_output_equals_exclusive_pass_output"))
PY
done
done
