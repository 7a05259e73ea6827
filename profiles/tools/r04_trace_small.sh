#!/bin/bash
# r04_trace_small.sh <tag> <reads> <chunks per step> -- kernel trace of an HBM-resident step of a SMALL batch: GPU busy fraction, idle gaps, per-kernel sums
TAG=${1:-r04i}; R=${2:-12500}; K=${3:-2}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf /tmp/lfp_small
rocprofv3 --kernel-trace --output-format csv -d /tmp/lfp_small -- python3 bench.py --reads $R --steps 3 --warmup 2 --no-cpu-baseline --no-exclusive --no-host-region > $OUT/bench_small_traced.json 2> /tmp/lfp_small.err
T=$(ls /tmp/lfp_small/*/*kernel_trace.csv | head -1)
python3 profiles/tools/busy.py $T --last-step $K > $OUT/gpu_busy_small.txt
python3 profiles/tools/gaps.py $T --last-step $K 5 > $OUT/gpu_gaps_small.txt
python3 - "$T" $K > $OUT/launches_small.txt <<'PY'
import csv, sys
from collections import Counter
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in rows)
k = int(sys.argv[2])
starts = [s for s, _, n in iv if n.startswith("lf_seed_search_kernel")]
lo = starts[-k]
sel = [x for x in iv if x[0] >= lo]
c = Counter(n for _, _, n in sel)
print("launches in the last step:", len(sel), "distinct kernels:", len(c))
for n, v in c.most_common(60):
    print(f"  {v:5d}  {n}")
PY
head -30 $OUT/gpu_busy_small.txt; head -24 $OUT/gpu_gaps_small.txt; head -8 $OUT/launches_small.txt
