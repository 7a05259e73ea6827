# C5 (T2T-like): the Hirschberg levels of one step, per call and level
mkdir -p gpurun_out/r6_c5dbg
LF_HIRSCH_DEBUG=1 timeout 1200 python3 bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive --no-host-region > gpurun_out/r6_c5dbg/c5.json 2> gpurun_out/r6_c5dbg/c5_levels.txt
grep -c "hirschberg:" gpurun_out/r6_c5dbg/c5_levels.txt
python3 - <<'PY'
import re
t=0.0; per=[]; cur=None
by_level={}
for line in open('gpurun_out/r6_c5dbg/c5_levels.txt'):
    m=re.search(r'level (\d+): .*?, ([0-9.]+) ms$', line.strip())
    if m:
        l=int(m.group(1)); ms=float(m.group(2)); by_level[l]=by_level.get(l,0)+ms
print({k:round(v,1) for k,v in sorted(by_level.items())})
PY
grep "trial bounds" gpurun_out/r6_c5dbg/c5_levels.txt | tail -3 | cut -c1-300
grep "level 1:" gpurun_out/r6_c5dbg/c5_levels.txt | sort -t, -k5 -n | tail -5 | cut -c1-330
