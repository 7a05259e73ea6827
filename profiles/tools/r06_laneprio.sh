# experiment: stream priority by lane id + chunk k on lane k for host batches (LF_LANE_PRIO=1) against the default
mkdir -p gpurun_out/r6_laneprio
for i in 1 2; do for pr in 0 1; do
if [ $pr = 1 ]; then export LF_LANE_PRIO=1; else unset LF_LANE_PRIO; fi
f=gpurun_out/r6_laneprio/c2_prio${pr}_$i.json
timeout 400 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exclusive > $f 2> gpurun_out/r6_laneprio/err.txt
python3 - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "host boundary", round(d["value"]), round(d["ms_per_step"],1), "hbm", round(d["value_hbm_resident"]), round(d["ms_per_step_hbm_resident"],1), "prepacked", round(d.get("value_prepacked_batch") or 0), d["sam_digests"]["host_boundary_timed_steps"]["xxh3_128"][:8])
PY
done; done
LF_LANE_PRIO=1 LF_TIMING=1 timeout 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive > /dev/null 2> gpurun_out/r6_laneprio/timing.err
grep "map_chunk \|egress\|lf_map_batch total" gpurun_out/r6_laneprio/timing.err | tail -26 | cut -c1-200
