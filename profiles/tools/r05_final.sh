#!/bin/bash
# r05_final.sh -- the round's last runs: GPU-busy trace of HBM-resident steps, the other configurations, and the default bench line twice (with the
# committed counter profile of the same tree attached)
OUT=$PWD/gpurun_out/r05_final; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf /tmp/lfp_busy
rocprofv3 --kernel-trace --output-format csv -d /tmp/lfp_busy -- python3 bench.py --reads 100000 --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive --no-host-region > /dev/null 2> /tmp/lfp_busy.err
python3 profiles/tools/busy.py $(ls /tmp/lfp_busy/*/*kernel_trace.csv | head -1) --last-step 8 > $OUT/gpu_busy_last_step.txt; cat $OUT/gpu_busy_last_step.txt | tail -3
python3 profiles/tools/gaps.py $(ls /tmp/lfp_busy/*/*kernel_trace.csv | head -1) --last-step 8 10 > $OUT/gpu_idle_gaps_last_step.txt
./profiles/tools/collect_r05.sh configs
for k in 1 2; do timeout 900 python3 bench.py > $OUT/bench_default_run$k.json 2> $OUT/bench_default_run$k.err; python3 - $OUT/bench_default_run$k.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('default bench: value',round(d['value']),round(d['ms_per_step'],1),'ms; hbm',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'; match',d.get('all_records_match_rate'),d.get('reads_compared'),'; roofline',r['kernel'],round(r['frac'],4),'traffic',r['traffic'],'alu',(r.get('alu') or {}).get('frac'))
PY
done
python3 bench.py --tree-hash > $OUT/SOURCE_TREE.txt
