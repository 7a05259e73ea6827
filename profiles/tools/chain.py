#!/usr/bin/env python3
"""chain.py <kernel_trace.csv> [min_us] -- the launch chain of ONE chunk: the kernels the busiest host thread of the trace launched between its last two
lf_seed_pos_kernel launches' first ... i.e. its last chunk, in start order: start (ms after the chunk's first kernel), duration, idle time since the previous kernel of the
chain ended, name.  Kernels shorter than min_us (default 20) are folded into the next printed line (count, summed duration, summed idle)."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
by = defaultdict(list)
for r in rows: by[r.get("Thread_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:56]))
# the thread whose LAST chunk we print: the one with the latest lf_seed_pos_kernel
best = None
for t, iv in by.items():
    iv.sort()
    pos = [i for i, (_, _, n) in enumerate(iv) if n.startswith("lf_seed_pos_kernel")]
    if pos and (best is None or iv[pos[-1]][0] > best[0]): best = (iv[pos[-1]][0], t, pos[-1])
_, t, i0 = best
iv = by[t][i0:]
# kernels of the chunk may precede lf_seed_pos_kernel (gather / pack): take up to 6 kernels of the same thread that ended less than 1 ms before
j = i0
while j > 0 and i0 - j < 8 and by[t][j][0] - by[t][j - 1][1] < 1_000_000: j -= 1
iv = by[t][j:]
t0 = iv[0][0]; prev_end = None
fold_n = 0; fold_d = 0.0; fold_g = 0.0
tot_d = tot_g = 0.0
for s, e, n in iv:
    d = (e - s) / 1e3; g = 0.0 if prev_end is None else max(0.0, (s - prev_end) / 1e3)
    prev_end = e if prev_end is None else max(prev_end, e)
    tot_d += d; tot_g += g
    if d < min_us and g < min_us:
        fold_n += 1; fold_d += d; fold_g += g; continue
    if fold_n: print(f"            ... {fold_n} short kernels, {fold_d:.0f} us running, {fold_g:.0f} us idle between them")
    fold_n = 0; fold_d = fold_g = 0.0
    print(f"{(s - t0) / 1e6:8.3f} ms  +{d:8.0f} us  idle before {g:7.0f} us  {n}")
if fold_n: print(f"            ... {fold_n} short kernels, {fold_d:.0f} us running, {fold_g:.0f} us idle between them")
print(f"chain: {len(iv)} kernels, {(prev_end - t0) / 1e6:.2f} ms from first start to last end, {tot_d / 1e3:.2f} ms running (summed), {tot_g / 1e3:.2f} ms idle between consecutive kernels")
