#!/bin/bash
# collect_r06.sh <part> -- round 6's profile sets, run on the GPU box from the repo root; results under gpurun_out/<dir>/ (copied into profiles/<dir>/)
#   part c2      : profiles/tools/collect.sh r06_c2 (bench with the CPU baseline, kernel trace + stats, serialized stats, FETCH / WRITE / SQ counter passes)
#   part configs : C4 (reads out of segmental duplications, clasp, -n 30, 10 x 100 k), C5 (ONT-like reads, T2T-like repeats), C2 on GRCh38-like repeats -- each with the CPU baseline
#   part hirsch  : every Hirschberg launch of one C5 step in the exclusive pass's mode, in time order (profiles/r06_hirsch/)
#   part shards  : the one-GPU proxy of strong scaling: 12.5 k / 25 k / 50 k-read steps alone and with 2 / 4 steps in flight
set -u
PART=${1:-c2}
summ() { python3 - "$1" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], 'value', round(d['value']), round(d['ms_per_step'],1), 'ms; hbm', round(d['value_hbm_resident']), round(d['ms_per_step_hbm_resident'],1), 'ms; prepacked', round(d.get('value_prepacked_batch',0)), '; cpu/step', round(d['host_cpu_seconds_per_step'],3), round(d['host_cpu_seconds_per_step_hbm_resident'],3), '; match', d.get('primary_record_match_rate'), d.get('all_records_match_rate'), d.get('reads_compared'), d.get('timed_output_equals_exclusive_pass_output'))
    print('   exclusive ms', {k.split(' ')[0]: round(v['ms_per_step'],1) for k,v in r['by_kernel'].items()}, 'cpu baseline', round(d.get('cpu_baseline',{}).get('value',0)), 'in flight', d.get('steps_in_flight'), 'roofline kernel', r['kernel'], round(r['frac'],4))
except Exception as e:
    print(f, 'FAILED', e)
PY
}
case $PART in
c2)
  ./profiles/tools/collect.sh r06_c2
  summ gpurun_out/r06_c2/bench_unprofiled.json
  ;;
configs)
  OUT=$PWD/gpurun_out/r06_configs; mkdir -p $OUT
  timeout 1500 python3 bench.py --config c4 --steps 10 --warmup 1 > $OUT/bench_c4_segdup_clasp_n30.json 2> $OUT/bench_c4.err; summ $OUT/bench_c4_segdup_clasp_n30.json
  timeout 1200 python3 bench.py --config c5 --steps 3 --warmup 1 > $OUT/bench_c5_ont50k_k17c2000_t2tlike.json 2> $OUT/bench_c5.err; summ $OUT/bench_c5_ont50k_k17c2000_t2tlike.json
  timeout 900 python3 bench.py --repeat-profile grch38like --steps 4 --warmup 1 > $OUT/bench_c2_grch38like.json 2> $OUT/bench_grch38like.err; summ $OUT/bench_c2_grch38like.json
  ;;
hirsch)
  for c in c5 c4; do bash profiles/tools/r06_c5trace.sh $c; tail -1 gpurun_out/r06_hirsch/${c}_last_step_launches_alone_on_the_gpu.txt; done
  ;;
shards)
  OUT=$PWD/gpurun_out/r06_shard_sweep; mkdir -p $OUT
  timeout 400 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exclusive --no-host-region > $OUT/step_100000_inflight1.json 2> $OUT/e.err; summ $OUT/step_100000_inflight1.json
  for R in 12500 25000 50000; do for D in 1 2 4; do
    timeout 300 python3 bench.py --reads $R --steps 16 --warmup 2 --inflight $D --no-cpu-baseline --no-exclusive --no-host-region > $OUT/step_${R}_inflight$D.json 2> $OUT/e.err; summ $OUT/step_${R}_inflight$D.json
  done; done
  ;;
esac
