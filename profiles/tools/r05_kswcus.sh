#!/bin/bash
# r05_kswcus.sh <tag> -- LF_KSW_CUS=0 / 8 / 16: duration of the replay's ksw rounds inside full host-boundary steps (LF_TIMING timelines) and the step times
OUT=gpurun_out/${1:-r05kswcus}; mkdir -p $OUT
for C in 0 8 0 8 16; do
  LF_KSW_CUS=$C LF_TIMING=1 timeout 400 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exclusive > $OUT/bench_cus$C.json 2> $OUT/timing_cus$C.err
  python3 - $OUT/bench_cus$C.json $OUT/timing_cus$C.err $C <<'PY'
import json,sys,re
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks=[]
for l in open(sys.argv[2]):
    if 'timeline' in l:
        ks += [float(x) for x in re.findall(r' KSW ([0-9.]+)', l)]
ks.sort()
print('LF_KSW_CUS',sys.argv[3],'host boundary',round(d['ms_per_step'],1),'ms; hbm-resident',round(d['ms_per_step_hbm_resident'],1),'ms; ksw rounds',len(ks),'mean %.1f ms, the 10 longest:'%(sum(ks)/max(1,len(ks))),[round(x,1) for x in ks[-10:]],'digest',(d.get('sam_digests') or {}).get('host_boundary_timed_steps',{}).get('xxh3_128'))
PY
done
