#!/usr/bin/env python3
"""gaps.py <kernel_trace.csv> --last-step <chunks per step> [min_us] -- where the GPU idles: every interval of the window with
NO kernel running, grouped by (last kernel that ended before it -> first kernel that starts after it)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]) for r in rows)
k = int(sys.argv[3])
min_ns = float(sys.argv[4]) * 1e3 if len(sys.argv) > 4 else 20e3
starts = [s for s, _, n in iv if "lf_seed_search_kernel" in n.split("(")[0]]
lo = starts[-k]
hi = max(e for _, e, _ in iv)
iv = [x for x in iv if x[1] > lo]
gaps = defaultdict(lambda: [0, 0])
cur_e, cur_n = None, None
idle = 0
for s, e, n in iv:
    if cur_e is not None and s > cur_e:
        g = s - cur_e
        idle += g
        if g >= min_ns:
            key = f"{cur_n} -> {n}"
            gaps[key][0] += 1
            gaps[key][1] += g
    if cur_e is None or e > cur_e:
        cur_e, cur_n = e, n
print(f"window {(hi - lo) / 1e6:.1f} ms, idle {idle / 1e6:.1f} ms ({100.0 * idle / (hi - lo):.1f} %)")
for key, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {t / 1e6:8.2f} ms in {c:4d} gaps   {key}")
