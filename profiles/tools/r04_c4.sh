#!/bin/bash
TAG=${1:-r04j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_configs.py -x -q -m gpu -k "clasp or c4" > $OUT/pytest_clasp.log 2>&1; tail -3 $OUT/pytest_clasp.log
export LF_BENCH_STACKS=300
timeout 900 python3 bench.py --config c4 --steps 4 --warmup 1 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_c4.json").read().strip().splitlines()[-1])
    print("c4 value", round(d["value"]), "ms", round(d["ms_per_step"], 1), "hbm", round(d["value_hbm_resident"]), round(d["ms_per_step_hbm_resident"], 1), "match", d.get("all_records_match_rate"), d.get("primary_record_match_rate"), d.get("reads_compared"), d.get("timed_output_equals_exclusive_pass_output"))
    print(d["per_read"])
    r = d["roofline"]
    print({k.split(" ")[0]: round(v["ms_per_step"], 2) for k, v in r["by_kernel"].items()}, round(r["exclusive_ms_sum_all_kernels"], 1))
    print(d.get("cpu_baseline"))
except Exception as e:
    print("FAILED", e); print(open("$OUT/bench_c4.err").read()[-3000:])
PY
