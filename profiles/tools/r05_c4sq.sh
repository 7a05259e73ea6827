#!/bin/bash
# r05_c4sq.sh -- where config C4's instruction issue goes: one SQ counter pass (wave-level VALU / SALU / LDS / VMEM instruction counts, busy cycles) over ONE HBM-resident step of
# 25 k C4 reads (default lanes), summed per kernel
OUT=$PWD/gpurun_out/r05_configs; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf /tmp/lfp_c4sq
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/lfp_c4sq -- python3 bench.py --config c4 --reads 25000 --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive --no-host-region > $OUT/bench_c4_under_sq.json 2> /tmp/lfp_c4sq.err
python3 profiles/tools/summarize_pmc.py $OUT/sq_counters_c4_25k_reads.json $(ls /tmp/lfp_c4sq/*/*counter_collection.csv | head -1) > /dev/null
python3 - $OUT/sq_counters_c4_25k_reads.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
rows=[(v.get("sq_insts_valu",0),v.get("sq_insts_salu",0),v.get("sq_wave_cycles",0),v.get("launches",0),k) for k,v in d.items() if isinstance(v,dict) and k.startswith("lf_") and not any(x in k for x in ("lf_cache","lf_full_sa","lf_bucket","lf_occ2","lf_sa_","lf_bwt","lf_idx"))]
tot=sum(r[0] for r in rows)
for va,sa,wc,l,k in sorted(rows,reverse=True)[:14]:
    print(f"{k[:44]:44s} launches {l:5d}  VALU {va/1e9:7.3f} G ({100*va/tot:4.1f} %)  SALU {sa/1e9:6.3f} G  wave-cycles {wc/1e9:7.2f} G")
print("all mapping kernels: VALU %.2f G per 25 k reads" % (tot/1e9))
PY
