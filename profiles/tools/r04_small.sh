#!/bin/bash
# quick correctness gate with tight timeouts: selected tests, a small C4 (dup) bench with the CPU baseline leg, steps in flight
TAG=${1:-r04k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_stress.py tests/test_gpu_dist.py -x -q -m gpu > $OUT/pytest_sel.log 2>&1; tail -4 $OUT/pytest_sel.log
export LF_BENCH_STACKS=100 LF_WATCHDOG=80
timeout 280 python3 bench.py --config c4 --genome-mbp 120 --reads 8000 --steps 2 --warmup 1 --cpu-seconds 6 > $OUT/bench_c4_small.json 2> $OUT/bench_c4_small.err
echo "rc $?"
python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_c4_small.json").read().strip().splitlines()[-1])
    print("c4 small value", round(d["value"]), "ms", round(d["ms_per_step"], 1), "hbm", round(d["value_hbm_resident"]), "match", d.get("all_records_match_rate"), d.get("primary_record_match_rate"), d.get("reads_compared"), d.get("timed_output_equals_exclusive_pass_output"))
    print(d["per_read"])
    r = d["roofline"]
    print({k.split(" ")[0]: round(v["ms_per_step"], 2) for k, v in r["by_kernel"].items()})
except Exception as e:
    print("FAILED", e); print(open("$OUT/bench_c4_small.err").read()[-2500:])
PY
unset LF_WATCHDOG
for D in 1 2 4; do
timeout 120 python3 bench.py --genome-mbp 120 --reads 4000 --steps 12 --warmup 2 --no-cpu-baseline --no-exclusive --no-host-region --inflight $D 2>$OUT/inflight$D.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight $D: hbm', round(d['value_hbm_resident']), round(d['ms_per_step_hbm_resident'],2))"
done
tail -3 $OUT/inflight4.err
