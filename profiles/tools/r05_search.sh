#!/bin/bash
# r05_search.sh <tag> -- the search kernel: seed tests, then the exclusive time of lf_seed_search_kernel per number of slots (LF_SEARCH_SLOTS)
OUT=gpurun_out/${1:-r05search}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_seed.py tests/test_gpu_map.py -x -q -m gpu > $OUT/pytest_seed_map.log 2>&1; tail -5 $OUT/pytest_seed_map.log
for S in ${SLOTS:-2 1 3}; do
  LF_SEARCH_SLOTS=$S timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-region > $OUT/bench_s$S.json 2> $OUT/bench_s$S.err
  python3 - $OUT/bench_s$S.json $S <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print('slots',sys.argv[2],'hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms; search',round(r['by_kernel']['lf_seed_search_kernel']['ms_per_step'],2),'ms', 'locate', round(r['by_kernel']['lf_seed_locate_kernel']['ms_per_step'],2), 'digest', d.get('sam_digests'), 'excl sum', round(r['exclusive_ms_sum_all_kernels'],1))
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
