#!/bin/bash
# r05_align.sh <tag> -- alignment group: stage + map tests, then exclusive times of the forward / traceback kernels per traceback part width
OUT=gpurun_out/${1:-r05align}; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_stages.py tests/test_gpu_map.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/pytest_align.log 2>&1; tail -5 $OUT/pytest_align.log
for HK in ${HKS:-8 16}; do
  LF_TB_HK=$HK timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-region > $OUT/bench_hk$HK.json 2> $OUT/bench_hk$HK.err
  python3 - $OUT/bench_hk$HK.json $HK <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; k=r['by_kernel']
    print('HK',sys.argv[2],'hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms; rsweep',round(k['lf_edlib_rsweep_kernel']['ms_per_step'],2),'tb',round(k['lf_edlib_tb_kernel']['ms_per_step'],2),'group',round(r['alignment_group_exclusive_ms'],2),'search',round(k['lf_seed_search_kernel']['ms_per_step'],2),'digest', d['sam_digests']['exclusive_pass']['xxh3_128'], d.get('timed_output_equals_exclusive_pass_output'), 'excl sum', round(r['exclusive_ms_sum_all_kernels'],1))
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
