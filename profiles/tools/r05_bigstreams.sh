#!/bin/bash
# r05_bigstreams.sh <tag> -- the problems of >= 8 blocks of each mode on a stream of their own (default) against with their mode's bulk (LF_ALIGN_BIG_NB=0): HBM-resident steps of
# 12.5 k, 25 k and 100 k reads (A B A B on one box), then the launch chain of one 6 250-read chunk
OUT=gpurun_out/${1:-r05big}; mkdir -p $OUT
for N in 12500 25000 100000; do for k in 1 2; do for M in 8 0; do
  LF_ALIGN_BIG_NB=$M timeout 600 python3 bench.py --reads $N --steps 10 --warmup 2 --no-cpu-baseline --no-exclusive --no-host-region > $OUT/b_${N}_$M.$k.json 2> $OUT/b_${N}_$M.$k.err
  python3 - $OUT/b_${N}_$M.$k.json $N $M <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('reads',sys.argv[2],'big problems on their own streams' if sys.argv[3]!='0' else 'with their mode                  ','| hbm-resident',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],2),'ms')
except Exception as e:
    print('FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done; done; done
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf /tmp/lfp_chain
rocprofv3 --kernel-trace --output-format csv -d /tmp/lfp_chain -- python3 bench.py --reads 6250 --steps 4 --warmup 2 --no-cpu-baseline --no-exclusive --no-host-region > $OUT/b_6250.json 2> $OUT/b_6250.err
python3 profiles/tools/chain.py $(ls /tmp/lfp_chain/*/*kernel_trace.csv | head -1) 20 > $OUT/chain_6250.txt; grep "rsweep\|tb_kernel\|^chain" $OUT/chain_6250.txt
