#!/bin/bash
# r05_mix.sh <tag> -- full GPU suite, CPU phases of an HBM-resident step, fine chunk-ramp sweep, exclusive times of the current tree
OUT=gpurun_out/${1:-r05mix}; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_full.log 2>&1; tail -4 $OUT/pytest_gpu_full.log
LF_PHASES=1 LF_SPIN_US=0 timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-region --no-exclusive > $OUT/bench_phases.json 2> $OUT/bench_phases.err
grep -A40 "batch of 100000 reads" $OUT/bench_phases.err | tail -45 > $OUT/phases_last_step.txt; cat $OUT/phases_last_step.txt | cut -c 1-200
RAMPS="15 20 30" ./profiles/tools/r05_ramp.sh ${1:-r05mix}/ramp
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-region > $OUT/bench_excl.json 2> $OUT/bench_excl.err
python3 - $OUT/bench_excl.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms', {k.split(' ')[0]: round(v['ms_per_step'],2) for k,v in r['by_kernel'].items()}, 'sum', round(r['exclusive_ms_sum_all_kernels'],1), 'digest', d['sam_digests'])
PY
