#!/bin/bash
# one SQ instruction-count pass (single chunk, 50 k reads): per-kernel VALU / SALU / VMEM instruction counts and wave cycles
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export LF_LANES=1 LF_SERIAL_CLASSES=1 LF_CHUNK_READS=1073741824 LF_CHUNK_BASES=1099511627776
rm -rf /tmp/lfp_s1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/lfp_s1 -- python3 $R/bench.py --reads 50000 --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive --no-host-region > /dev/null 2> /tmp/lfp_s1.err
cd $R
python3 profiles/tools/summarize_pmc.py gpurun_out/sq1.json $(ls /tmp/lfp_s1/*/*counter_collection.csv)
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/sq1.json'))
for k,v in d.items():
    if k.startswith('_') or 'cache_level' in k or 'full_sa' in k: continue
    if v.get('sq_insts_valu',0) < 5e7: continue
    print(f"{k[:50]:50s} n={v['launches']:3d} valu {v['sq_insts_valu']/1e9:7.3f} G salu {v['sq_insts_salu']/1e9:7.3f} G vmem {(v['sq_insts_vmem_rd']+v['sq_insts_vmem_wr'])/1e9:6.3f} G wavecyc {v['sq_wave_cycles']/1e9:7.2f} G")
PY
