#!/bin/bash
# r05_cfgstats.sh -- rocprofv3 --kernel-trace --stats of configs C4 and C5 (T2T-like genome): per-kernel totals with the chunks of a step in flight together
OUT=$PWD/gpurun_out/r05_configs; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf /tmp/lfp_c4 /tmp/lfp_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lfp_c4 -- python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-exclusive > $OUT/bench_c4_under_rocprof.json 2> /tmp/lfp_c4.err
python3 profiles/tools/trim_stats.py $(ls /tmp/lfp_c4/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_c4.csv; head -12 $OUT/kernel_stats_c4.csv | cut -c 1-120
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lfp_c5 -- python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive > $OUT/bench_c5_t2tlike_under_rocprof.json 2> /tmp/lfp_c5.err
python3 profiles/tools/trim_stats.py $(ls /tmp/lfp_c5/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_c5_t2tlike.csv; head -12 $OUT/kernel_stats_c5_t2tlike.csv | cut -c 1-120
