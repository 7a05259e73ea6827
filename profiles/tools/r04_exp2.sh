#!/bin/bash
TAG=${1:-r04q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
export LF_BENCH_STACKS=200
B="timeout 300 python3 bench.py --no-cpu-baseline --no-exclusive"
line() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', 'value', round(d['value']), round(d['ms_per_step'],2), 'hbm', round(d['value_hbm_resident']), round(d['ms_per_step_hbm_resident'],2), 'cpu', round(d['host_cpu_seconds_per_step'],3))" 2>/dev/null || { echo "$2 FAILED"; tail -5 ${1%.json}.err; }; }
for R in 12500 25000 50000; do for D in 2 4; do
  $B --reads $R --steps 24 --warmup 2 --inflight $D --no-host-region > $OUT/sweep_${R}_d$D.json 2> $OUT/sweep_${R}_d$D.err; line $OUT/sweep_${R}_d$D.json "reads $R inflight $D"
done; done
$B --reads 12500 --steps 24 --warmup 2 --inflight 4 > $OUT/sweep_host_12500_d4.json 2> $OUT/sweep_host_12500_d4.err; line $OUT/sweep_host_12500_d4.json "host+hbm reads 12500 inflight 4"
timeout 1000 python3 bench.py --config c4 --steps 10 --warmup 1 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
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
