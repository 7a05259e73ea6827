#!/bin/bash
TAG=${1:-r04r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export LF_BENCH_STACKS=200
B="timeout 300 python3 bench.py --no-cpu-baseline --no-exclusive"
line() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', 'value', round(d['value']), round(d['ms_per_step'],2), 'hbm', round(d['value_hbm_resident']), round(d['ms_per_step_hbm_resident'],2), 'cpu', round(d['host_cpu_seconds_per_step'],3), d.get('all_records_match_rate'), d.get('timed_output_equals_exclusive_pass_output'))" 2>/dev/null || { echo "$2 FAILED"; tail -5 ${1%.json}.err; }; }
timeout 500 python3 bench.py --steps 8 --warmup 2 > $OUT/c2.json 2> $OUT/c2.err; line $OUT/c2.json "c2 default"
python3 -c "import json; d=json.loads(open('$OUT/c2.json').read().strip().splitlines()[-1]); r=d['roofline']; print(r['kernel'], round(r['frac'],4), {k.split(' ')[0]: round(v['ms_per_step'],2) for k,v in r['by_kernel'].items()}, round(r['exclusive_ms_sum_all_kernels'],1)); print(d['config']['index'])"
LF_FIRST_CHUNK_READS=0 $B --steps 8 --warmup 2 > $OUT/c2_noramp.json 2> $OUT/c2_noramp.err; line $OUT/c2_noramp.json "c2 no ramp"
LF_FIRST_CHUNK_READS=1024 $B --steps 8 --warmup 2 > $OUT/c2_ramp1k.json 2> $OUT/c2_ramp1k.err; line $OUT/c2_ramp1k.json "c2 ramp 1024"
$B --steps 8 --warmup 2 --inflight 2 > $OUT/c2_d2.json 2> $OUT/c2_d2.err; line $OUT/c2_d2.json "c2 inflight 2"
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
