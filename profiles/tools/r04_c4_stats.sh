#!/bin/bash
# r04_c4_stats.sh -- rocprofv3 --kernel-trace --stats of the C4 workload (reads out of segmental duplications, clasp, -n 30): two timed HBM-resident steps + the exclusive pass
OUT=$PWD/gpurun_out/r04_configs; mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lfp_c4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lfp_c4 -- python3 $R/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-host-region > $OUT/bench_c4_under_rocprof_kernel_trace.json 2> /tmp/lfp_c4.err
cp $(ls /tmp/lfp_c4/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_c4.csv
head -32 $OUT/kernel_stats_c4.csv | cut -c1-150
