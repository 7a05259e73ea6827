# C2: what moved the step -- wave priority of the multi-wavefront Hirschberg kernels (LF_HIRSCH_BAND=3: off), the lane-group queues (=64: off)
mkdir -p gpurun_out/r6_c2ab
for band in 1 3 64 1 3 64; do
f=gpurun_out/r6_c2ab/c2_band${band}_$RANDOM.json
LF_HIRSCH_BAND=$band timeout 600 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-region --no-exclusive > $f 2> gpurun_out/r6_c2ab/err.txt
python3 - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1))
PY
done
