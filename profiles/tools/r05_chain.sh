#!/bin/bash
# r05_chain.sh <tag> -- the launch chain of a single chunk alone on the GPU (1600 and 6250 reads, HBM-resident, one step at a time): where its fixed latency is
OUT=gpurun_out/${1:-r05chain}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for N in 1600 6250; do
  rm -rf /tmp/lfp_chain
  rocprofv3 --kernel-trace --output-format csv -d /tmp/lfp_chain -- python3 bench.py --reads $N --steps 4 --warmup 2 --no-cpu-baseline --no-exclusive --no-host-region > $OUT/b_$N.json 2> $OUT/b_$N.err
  python3 -c "
import json; d=json.loads(open('$OUT/b_$N.json').read().strip().splitlines()[-1]); print('reads $N ms per step', round(d['ms_per_step_hbm_resident'],2), 'chunks/step', d.get('chunks_per_step'), 'waits/chunk', d.get('host_waits_per_chunk'))"
  python3 profiles/tools/chain.py $(ls /tmp/lfp_chain/*/*kernel_trace.csv | head -1) 20 > $OUT/chain_$N.txt; tail -1 $OUT/chain_$N.txt
done
