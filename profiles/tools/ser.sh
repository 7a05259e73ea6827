#!/bin/bash
# ser.sh <tag> -- serialized (one chunk at a time, size classes on one stream) kernel stats for config C2 + LF_HIST_STATS
set -u
TAG=${1:-ser}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python3 bench.py --reads 100000"
rm -rf /tmp/lfp_ser
LF_SERIAL_CLASSES=1 LF_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lfp_ser -- $B --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_serialized.json 2> $OUT/ser.err
python3 profiles/tools/trim_stats.py $(ls /tmp/lfp_ser/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_serialized.csv
