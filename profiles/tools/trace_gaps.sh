#!/bin/bash
# kernel trace of the HBM-resident steps only: busy fraction and idle gaps of the last step (chunks per step = $1, default 8)
K=${1:-8}
R=$PWD
mkdir -p $R/gpurun_out/tg
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg && mkdir -p /tmp/tg
rocprofv3 --kernel-trace --stats -d /tmp/tg -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-exclusive --no-host-region --no-cpu-baseline > $R/gpurun_out/tg/bench.json 2> $R/gpurun_out/tg/bench.err
T=$(find /tmp/tg -name '*kernel_trace.csv' | head -1)
python3 $R/profiles/tools/busy.py $T --last-step $K > $R/gpurun_out/tg/busy.txt
python3 $R/profiles/tools/gaps.py $T --last-step $K 10 > $R/gpurun_out/tg/gaps.txt
cp $(find /tmp/tg -name '*kernel_stats.csv' | head -1) $R/gpurun_out/tg/kernel_stats.csv
python3 $R/profiles/tools/timeline.py $T --last-step $K > $R/gpurun_out/tg/timeline.txt
cat $R/gpurun_out/tg/busy.txt $R/gpurun_out/tg/gaps.txt; head -150 $R/gpurun_out/tg/timeline.txt
