#!/bin/bash
# r05_waits.sh <tag> -- which host waits a chunk's launch chain consists of: LF_WAIT_TRACE=1 counts the waits per call site (lf_mem.hip) over
# HBM-resident steps of 12.5 k and 100 k reads; divided here by the chunks the steps mapped
OUT=gpurun_out/${1:-r05waits}; mkdir -p $OUT
for N in 12500 100000; do
  LF_WAIT_TRACE=1 timeout 600 python3 bench.py --reads $N --steps 4 --warmup 1 --no-cpu-baseline --no-exclusive --no-host-region > $OUT/b_$N.json 2> $OUT/b_$N.err
  python3 - $OUT/b_$N.json $OUT/b_$N.err $N <<'PY'
import json,sys,re
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("reads",sys.argv[3],"ms per step (HBM-resident)",round(d['ms_per_step_hbm_resident'],2),"waits/chunk",d.get('host_waits_per_chunk'),"chunks/step",d.get('chunks_per_step'))
rows=[(int(m.group(1)),m.group(2)) for m in (re.match(r"\[lf\] waits\s+(\d+) at (\S+)",l) for l in open(sys.argv[2])) if m]
tot=sum(n for n,_ in rows)
for n,s in sorted(rows,key=lambda r:r[1]): print("   %6d  %5.1f %%  %s"%(n,100*n/tot,s))
PY
done
