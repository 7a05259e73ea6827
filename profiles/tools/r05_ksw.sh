#!/bin/bash
# r05_ksw.sh <tag> -- the register ksw kernel: parity (stage / drop-in / map tests), rows per microsecond against the four-wavefront kernel, config C2 and C4 steps
OUT=gpurun_out/${1:-r05ksw}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_map.py tests/test_dropin.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/pytest_sel.log 2>&1; tail -3 $OUT/pytest_sel.log
timeout 600 python3 profiles/tools/ksw_bench.py > $OUT/ksw_bench.jsonl 2> $OUT/ksw_bench.err; cat $OUT/ksw_bench.jsonl | cut -c 1-400
run() {   # name, config, env...
  local name=$1 cfg=$2; shift 2
  env "$@" timeout 900 python3 bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-host-region > $OUT/b_$name.json 2> $OUT/b_$name.err
  python3 - $OUT/b_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2], 'hbm', round(d['value_hbm_resident']), round(d['ms_per_step_hbm_resident'],2), 'equal', d.get('timed_output_equals_exclusive_pass_output'), d['sam_digests']['exclusive_pass']['xxh3_128'], {k.split(' ')[0]: round(v['ms_per_step'],2) for k,v in r['by_kernel'].items()}, round(r['exclusive_ms_sum_all_kernels'],1))
except Exception as e:
    print(sys.argv[2], 'FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1200:])
PY
}
run c2_r4 c2 LF_X=1
run c2_mw c2 LF_KSW_R4=0
run c4_r4 c4 LF_X=1
run c4_mw c4 LF_KSW_R4=0
