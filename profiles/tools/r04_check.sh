#!/bin/bash
OUT=gpurun_out/${1:-r04w}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_map.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/pytest_sel.log 2>&1; tail -3 $OUT/pytest_sel.log
timeout 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>$OUT/b1.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value', round(d['value']), round(d['ms_per_step'],2), 'hbm', round(d['value_hbm_resident']), round(d['ms_per_step_hbm_resident'],2), 'cpu', round(d['host_cpu_seconds_per_step'],3), round(d['host_cpu_seconds_per_step_hbm_resident'],3), d.get('timed_output_equals_exclusive_pass_output'))
print({k.split(' ')[0]: round(v['ms_per_step'],2) for k,v in r['by_kernel'].items()}, round(r['exclusive_ms_sum_all_kernels'],1))"
