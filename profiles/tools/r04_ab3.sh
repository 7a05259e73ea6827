#!/bin/bash
# r04_ab3.sh <tag> -- parity tests, then exclusive kernel times: default tree, LF_TB_HK=16, and the ksw microbenchmark
OUT=gpurun_out/${1:-r04x}; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_stages.py tests/test_gpu_map.py tests/test_gpu_configs.py tests/test_dropin.py -x -q -m gpu > $OUT/pytest_sel.log 2>&1; tail -3 $OUT/pytest_sel.log
run() {   # name, config, env...
  local name=$1; shift; local cfg=$1; shift
  env "$@" timeout 600 python3 bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-host-region > $OUT/b_$name.json 2> $OUT/b_$name.err
  python3 - $OUT/b_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2], 'hbm', round(d['value_hbm_resident']), round(d['ms_per_step_hbm_resident'],2), 'equal', d.get('timed_output_equals_exclusive_pass_output'), d['sam_digests']['exclusive_pass']['xxh3_128'], {k.split(' ')[0]: round(v['ms_per_step'],2) for k,v in r['by_kernel'].items()}, round(r['exclusive_ms_sum_all_kernels'],1))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run c2 c2 LF_X=1
run c2b c2 LF_X=2
run c4 c4 LF_X=1
timeout 600 python3 profiles/tools/ksw_bench.py > $OUT/ksw_bench.jsonl 2> $OUT/ksw_bench.err; cat $OUT/ksw_bench.jsonl; tail -3 $OUT/ksw_bench.err
