#!/bin/bash
# r05_kswab.sh <tag> -- config C4 with all extension requests of a chain in one round (default) against one request per round (LF_KSW_ONE_PER_ROUND=1, round 4), same box, A B A B;
# then the GPU-busy share of 12.5 k-read steps (one step at a time)
OUT=gpurun_out/${1:-r05kswab}; mkdir -p $OUT
for k in 1 2; do for M in 0 1; do
  LF_KSW_ONE_PER_ROUND=$M timeout 900 python3 bench.py --config c4 --steps 6 --warmup 1 --no-cpu-baseline --no-exclusive > $OUT/c4_one$M.$k.json 2> $OUT/c4_one$M.$k.err
  python3 - $OUT/c4_one$M.$k.json $M <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('one request per round' if sys.argv[2]=='1' else 'all requests in one round','| host boundary',round(d['value']),round(d['ms_per_step'],1),'ms; hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'ms; cpu/step',round(d['host_cpu_seconds_per_step'],3),round(d['host_cpu_seconds_per_step_hbm_resident'],3),'waits/chunk',d.get('host_waits_per_chunk'),'digest',(d.get('sam_digests') or {}).get('host_boundary_timed_steps',{}).get('xxh3_128'))
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done; done
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf /tmp/lfp_busy
rocprofv3 --kernel-trace --output-format csv -d /tmp/lfp_busy -- python3 bench.py --reads 12500 --steps 4 --warmup 2 --no-cpu-baseline --no-exclusive --no-host-region > $OUT/b12500.json 2> $OUT/b12500.err
python3 profiles/tools/busy.py $(ls /tmp/lfp_busy/*/*kernel_trace.csv | head -1) --last-step 4 > $OUT/gpu_busy_12500.txt; tail -5 $OUT/gpu_busy_12500.txt
python3 profiles/tools/gaps.py $(ls /tmp/lfp_busy/*/*kernel_trace.csv | head -1) --last-step 4 30 > $OUT/gpu_gaps_12500.txt; tail -40 $OUT/gpu_gaps_12500.txt
