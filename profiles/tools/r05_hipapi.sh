#!/bin/bash
# r05_hipapi.sh <tag> -- which HIP runtime calls the lane drivers spend their CPU time in: rocprofv3 --hip-trace --stats of HBM-resident steps
OUT=$PWD/gpurun_out/${1:-r05hipapi}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf /tmp/lfp_api
LF_SPIN_US=0 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/lfp_api -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-region --no-exclusive > $OUT/bench_under_hip_trace.json 2> /tmp/lfp_api.err
ls /tmp/lfp_api/*/ | head
python3 - $(ls /tmp/lfp_api/*/*hip_api_stats.csv | head -1) > $OUT/hip_api_stats.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("HIP API calls of the whole run (index load + 4 HBM-resident steps of 100 k reads); total %.2f s in the runtime" % (tot/1e9))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:25]:
    print("%-40s calls %8s  total %9.1f ms  avg %8.1f us" % (r['Name'][:40], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
cat $OUT/hip_api_stats.txt
tail -2 /tmp/lfp_api.err
