# C4: lane-group queues on (LF_HIRSCH_BAND=1) / off (=64) / no wave priority (=3), HBM-resident steps, three rounds
mkdir -p gpurun_out/r6_c4ab2
for i in 1 2 3; do for band in 1 64 3; do
f=gpurun_out/r6_c4ab2/c4_band${band}_$i.json
LF_HIRSCH_BAND=$band timeout 600 python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-host-region --no-exclusive > $f 2> gpurun_out/r6_c4ab2/err.txt
python3 - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1))
PY
done; done
