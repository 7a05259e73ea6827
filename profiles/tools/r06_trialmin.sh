# C5: trial bounds for the roots above 512 rows against only those above 4096 (fixed 3 / 2 sixteenths = what the adaptive choice arrives at)
mkdir -p gpurun_out/r6_trialmin
for i in 1 2 3 4; do for mn in 512 4096; do
f=gpurun_out/r6_trialmin/c5_min${mn}_$i.json
LF_HIRSCH_TRIAL=3,2,$mn timeout 600 python3 bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-host-region --no-exclusive > $f 2> gpurun_out/r6_trialmin/err.txt
python3 - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1))
PY
done; done
