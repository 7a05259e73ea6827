# C5: fixed trial bounds (sixteenths of the rows, NW,SHW) against the adaptive choice -- a failed trial now costs a second sweep inside the bound it found
mkdir -p gpurun_out/r6_trial
for tr in adaptive 2,2 2,1 1,1 adaptive 2,2; do
if [ $tr = adaptive ]; then unset LF_HIRSCH_TRIAL; else export LF_HIRSCH_TRIAL=$tr; fi
f=gpurun_out/r6_trial/c5_trial_${tr/,/_}_$RANDOM.json
timeout 900 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-host-region > $f 2> gpurun_out/r6_trial/err.txt
python3 - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
bk=d["roofline"]["by_kernel"]
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), {k.split(' ')[0]:round(v["ms_per_step"],1) for k,v in bk.items() if "hirsch" in k}, d.get("timed_output_equals_exclusive_pass_output"))
PY
done
