#!/bin/bash
# r04_vote_ab.sh <tag> -- parity tests of the stages the new vote kernel feeds, then A / B of the exclusive kernel times:
#   lf_vote_cell_kernel (default) against LF_VOTE_SCAN=1 (lf_vote_hash_kernel), LF_TB_HK=4 against 8
OUT=gpurun_out/${1:-r04v}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_map.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/pytest_sel.log 2>&1; tail -3 $OUT/pytest_sel.log
run() {   # name, env...
  local name=$1; shift
  env "$@" timeout 400 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-region > $OUT/b_$name.json 2> $OUT/b_$name.err
  python3 - $OUT/b_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2], 'hbm', round(d['value_hbm_resident']), round(d['ms_per_step_hbm_resident'],2), 'match', d.get('all_records_match_rate'), d.get('timed_output_equals_exclusive_pass_output'), {k.split(' ')[0]: round(v['ms_per_step'],2) for k,v in r['by_kernel'].items()}, round(r['exclusive_ms_sum_all_kernels'],1))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run cell LF_X=1
run scan LF_VOTE_SCAN=1
run tbhk4 LF_TB_HK=4
run cell2 LF_X=1
