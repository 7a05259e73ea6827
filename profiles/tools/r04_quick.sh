#!/bin/bash
# r04_quick.sh <tag> [nopytest] -- GPU suite + the default bench line (with the CPU baseline leg) + an A/B of the SAM egress + shard sweep, into gpurun_out/<tag>/
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "$2" != "nopytest" ]; then timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log; fi
export LF_BENCH_STACKS=200
timeout 500 python3 bench.py --steps 6 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
LF_SAM_FULL=1 timeout 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exclusive > $OUT/bench_samfull.json 2> $OUT/bench_samfull.err
for R in 12500 25000 50000; do timeout 200 python3 bench.py --reads $R --steps 8 --warmup 2 --no-cpu-baseline --no-exclusive > $OUT/sweep_$R.json 2> $OUT/sweep_$R.err; done
LF_TIMING=1 timeout 200 python3 bench.py --reads 12500 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive --no-host-region > /dev/null 2> $OUT/timeline_12500.err
grep "timeline" $OUT/timeline_12500.err | tail -12 > $OUT/timeline_12500.txt
python3 - <<PY
import json
for f in ("bench", "bench_samfull", "sweep_12500", "sweep_25000", "sweep_50000"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); print(open("$OUT/%s.err" % f).read()[-2000:]); continue
    print(f, "value", round(d["value"]), "ms", round(d["ms_per_step"], 1), "hbm", round(d["value_hbm_resident"]), round(d["ms_per_step_hbm_resident"], 1),
          "cpu/step", round(d["host_cpu_seconds_per_step"], 3), round(d["host_cpu_seconds_per_step_hbm_resident"], 3), "match", d.get("all_records_match_rate"), d.get("timed_output_equals_exclusive_pass_output"), d.get("reads_compared"))
    if f == "bench":
        r = d["roofline"]
        print("  ", r["kernel"], round(r["frac"], 4), {k.split(" ")[0]: round(v["ms_per_step"], 2) for k, v in r["by_kernel"].items()}, round(r["exclusive_ms_sum_all_kernels"], 1))
        print("  ", d["config"]["index"])
PY
tail -4 $OUT/timeline_12500.txt | cut -c 1-900
