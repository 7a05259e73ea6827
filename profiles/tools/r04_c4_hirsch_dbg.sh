#!/bin/bash
# r04_c4_hirsch_dbg.sh <tag> -- the Hirschberg levels of the C4 workload, call by call (LF_HIRSCH_DEBUG=1), one 20 k-read step as ONE chunk on one lane
OUT=gpurun_out/${1:-r04hd}; mkdir -p $OUT
LF_HIRSCH_DEBUG=1 LF_LANES=1 LF_CHUNK_READS=1000000 LF_CHUNK_BASES=100000000000 timeout 600 python3 bench.py --config c4 --reads 20000 --steps 1 --warmup 0 --no-cpu-baseline --no-host-region --no-exclusive > $OUT/b.json 2> $OUT/b.err
grep "hirschberg\|level" $OUT/b.err | head -60
python3 - $OUT/b.err <<'PY'
import re,sys
tot=0.0; calls=0; roots=0
for l in open(sys.argv[1]):
    m=re.search(r"level \d+: .* ([\d.]+) ms", l)
    if m: tot+=float(m.group(1))
    m=re.search(r"hirschberg: (\d+) roots", l)
    if m: calls+=1; roots+=int(m.group(1))
print("calls", calls, "roots", roots, "ms in levels", round(tot,1))
PY
