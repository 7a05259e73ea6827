#!/bin/bash
# sq.sh <tag> -- SQ instruction / stall counters per kernel, one chunk at a time (LF_LANES=1 LF_SERIAL_CLASSES=1), config C2.
# Two --pmc passes (8 SQ slots each), nothing else traced.  Summary: gpurun_out/<tag>/sq_summary.json
set -u
TAG=${1:-r02_sq}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export LF_LANES=1 LF_SERIAL_CLASSES=1
B="python3 bench.py --reads ${READS:-50000} --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive"
rm -rf /tmp/lfp_s1 /tmp/lfp_s2 /tmp/lfp_s3
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/lfp_s1 -- $B > $OUT/bench_sq1.json 2> /tmp/lfp_s1.err
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/lfp_s2 -- $B > $OUT/bench_sq2.json 2> /tmp/lfp_s2.err
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_GDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --output-format csv -d /tmp/lfp_s3 -- $B > $OUT/bench_sq3.json 2> /tmp/lfp_s3.err
tail -3 /tmp/lfp_s1.err /tmp/lfp_s2.err /tmp/lfp_s3.err > $OUT/err_tails.txt
python3 profiles/tools/summarize_pmc.py $OUT/sq_summary.json $(ls /tmp/lfp_s1/*/*counter_collection.csv /tmp/lfp_s2/*/*counter_collection.csv /tmp/lfp_s3/*/*counter_collection.csv 2>/dev/null)
ls -la $OUT
