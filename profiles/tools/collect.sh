#!/bin/bash
# collect.sh <tag> -- run on the GPU box from the repo root: bench + rocprofv3 passes for config C2, summaries into
# gpurun_out/<tag>/ (copy what you want judged into profiles/<tag>/).  rocprofv3 gets the program itself after `--`.
set -u
TAG=${1:-r04_c2}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python3 bench.py --reads 100000"
$B --steps 3 --warmup 1 > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
rm -rf /tmp/lfp_kt /tmp/lfp_ser /tmp/lfp_f /tmp/lfp_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lfp_kt -- $B --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof_kernel_trace.json 2> /tmp/lfp_kt.err
python3 profiles/tools/trim_stats.py $(ls /tmp/lfp_kt/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf /tmp/lfp_busy
rocprofv3 --kernel-trace --output-format csv -d /tmp/lfp_busy -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive --no-host-region > /dev/null 2> /tmp/lfp_busy.err
python3 profiles/tools/busy.py $(ls /tmp/lfp_busy/*/*kernel_trace.csv | head -1) --last-step 8 > $OUT/gpu_busy_last_step.txt
python3 profiles/tools/gaps.py $(ls /tmp/lfp_busy/*/*kernel_trace.csv | head -1) --last-step 8 10 > $OUT/gpu_idle_gaps_last_step.txt
cp $(ls /tmp/lfp_kt/*/*agent_info.csv | head -1) $OUT/agent_info.csv 2>/dev/null
LF_CHUNK_READS=1073741824 LF_CHUNK_BASES=1099511627776 LF_SERIAL_CLASSES=1 LF_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lfp_ser -- $B --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_serialized.json 2> /tmp/lfp_ser.err
python3 profiles/tools/trim_stats.py $(ls /tmp/lfp_ser/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_serialized.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/lfp_f -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive --no-host-region > $OUT/bench_under_pmc_fetch.json 2> /tmp/lfp_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/lfp_w -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive --no-host-region > $OUT/bench_under_pmc_write.json 2> /tmp/lfp_w.err
python3 profiles/tools/summarize_pmc.py $OUT/pmc_fetch_write_summary.json $(ls /tmp/lfp_f/*/*counter_collection.csv | head -1) $(ls /tmp/lfp_w/*/*counter_collection.csv | head -1)
# SQ instruction / stall counters, one chunk at a time, 50 k reads (three 8-slot passes)
export LF_LANES=1 LF_SERIAL_CLASSES=1 LF_CHUNK_READS=1073741824 LF_CHUNK_BASES=1099511627776
BS="python3 bench.py --reads 50000 --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive --no-host-region"
rm -rf /tmp/lfp_s1 /tmp/lfp_s2 /tmp/lfp_s3
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/lfp_s1 -- $BS > /dev/null 2> /tmp/lfp_s1.err
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/lfp_s2 -- $BS > /dev/null 2> /tmp/lfp_s2.err
rocprofv3 --pmc SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d /tmp/lfp_s3 -- $BS > /dev/null 2> /tmp/lfp_s3.err
LF_PMC_READS=50000 python3 profiles/tools/summarize_pmc.py $OUT/sq_counters_50k_reads.json $(ls /tmp/lfp_s1/*/*counter_collection.csv /tmp/lfp_s2/*/*counter_collection.csv /tmp/lfp_s3/*/*counter_collection.csv 2>/dev/null)
unset LF_LANES LF_SERIAL_CLASSES LF_CHUNK_READS LF_CHUNK_BASES
python3 bench.py --tree-hash > $OUT/SOURCE_TREE.txt; (git rev-parse HEAD 2>/dev/null || echo "no git on this box") >> $OUT/SOURCE_TREE.txt
ls -la $OUT
