#!/bin/bash
# r04_c4_hist.sh <tag> -- where the DP cells of the C4 workload's replay rounds are: LF_HIST_STATS=1 prints, per alignment call with host
# descriptors (the replay rounds), problems / cells / block steps by blocks per problem; summed here over one 20 k-read step
OUT=gpurun_out/${1:-r04hist}; mkdir -p $OUT
LF_HIST_STATS=1 timeout 600 python3 bench.py --config c4 --reads 20000 --steps 1 --warmup 0 --no-cpu-baseline --no-host-region --no-exclusive > $OUT/b.json 2> $OUT/b.err
python3 - $OUT/b.err <<'PY'
import re,sys,collections
cnt=collections.Counter(); cells=collections.Counter(); steps=collections.Counter(); calls=0
for l in open(sys.argv[1]):
    m=re.match(r"\[lf\] dp nb<=(\d+): (\d+) problems, ([\d.]+) Mcells, ([\d.]+) M block steps", l)
    if m:
        k=int(m.group(1)); cnt[k]+=int(m.group(2)); cells[k]+=float(m.group(3)); steps[k]+=float(m.group(4))
for k in sorted(cnt): print(f"nb<={k:4d}: {cnt[k]:9d} problems {cells[k]/1e3:10.1f} Gcells {steps[k]/1e3:9.2f} G block steps")
print("total", sum(cnt.values()), "problems", round(sum(steps.values())/1e3,2), "G block steps in the replay rounds of one 20 k-read step")
PY
