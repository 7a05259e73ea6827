#!/bin/bash
# r04_exp.sh <tag> -- experiments on the full-size C2 workload: shard sweep with steps in flight, CU split, C4
TAG=${1:-r04o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export LF_BENCH_STACKS=200
B="timeout 300 python3 bench.py --no-cpu-baseline --no-exclusive"
line() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', 'value', round(d['value']), round(d['ms_per_step'],2), 'hbm', round(d['value_hbm_resident']), round(d['ms_per_step_hbm_resident'],2), 'cpu', round(d['host_cpu_seconds_per_step'],3))" 2>/dev/null || { echo "$2 FAILED"; tail -5 ${1%.json}.err; }; }
$B --steps 6 --warmup 2 > $OUT/c2.json 2> $OUT/c2.err; line $OUT/c2.json "c2 100k"
for R in 12500 25000 50000; do for D in 1 2 4; do
  $B --reads $R --steps 16 --warmup 2 --inflight $D --no-host-region > $OUT/sweep_${R}_d$D.json 2> $OUT/sweep_${R}_d$D.err; line $OUT/sweep_${R}_d$D.json "reads $R inflight $D"
done; done
for S in 2 4; do LF_CU_SPLIT=$S $B --steps 6 --warmup 2 --no-host-region > $OUT/cusplit_$S.json 2> $OUT/cusplit_$S.err; line $OUT/cusplit_$S.json "cu split $S"; done
LF_CU_SPLIT=4 timeout 300 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-host-region > $OUT/cusplit_4_excl.json 2> $OUT/cusplit_4_excl.err
python3 - <<PY
import json
for f in ("c2", "cusplit_4_excl"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f, {k.split(" ")[0]: round(v["ms_per_step"], 2) for k, v in r["by_kernel"].items()})
    except Exception as e:
        print(f, "FAILED", e)
PY
