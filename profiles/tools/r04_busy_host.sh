#!/bin/bash
# r04_busy_host.sh <tag> -- GPU busy fraction and idle gaps of the last HOST-BOUNDARY step (kernel trace; --no-exclusive, so the last 8 chunks are a host step)
OUT=$PWD/gpurun_out/${1:-r04bh}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf /tmp/lfp_bh
rocprofv3 --kernel-trace --output-format csv -d /tmp/lfp_bh -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive > $OUT/bench_under_trace.json 2> $OUT/bench_under_trace.err
python3 $R/profiles/tools/busy.py $(ls /tmp/lfp_bh/*/*kernel_trace.csv | head -1) --last-step 8 > $OUT/gpu_busy_last_host_step.txt
python3 $R/profiles/tools/gaps.py $(ls /tmp/lfp_bh/*/*kernel_trace.csv | head -1) --last-step 8 12 > $OUT/gpu_idle_gaps_last_host_step.txt
python3 $R/profiles/tools/timeline.py $(ls /tmp/lfp_bh/*/*kernel_trace.csv | head -1) --last-step 8 15 0.7 > $OUT/timeline_last_host_step.txt; head -150 $OUT/timeline_last_host_step.txt
