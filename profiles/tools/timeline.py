#!/usr/bin/env python3
"""timeline.py <kernel_trace.csv> --last-step <chunks per step> [lead_ms] [min_ms] -- the kernels of the last step of a rocprofv3 --kernel-trace CSV in start
order: start (ms after the window's begin = lead_ms before the step's first seed search), duration, name; kernels shorter than min_ms are summed per 5 ms."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
k = int(sys.argv[3]); lead = float(sys.argv[4]) if len(sys.argv) > 4 else 15.0; min_ms = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5
starts = [s for s, _, n in iv if "lf_seed_search_kernel" in n.split("(")[0]]
lo = starts[-k] - int(lead * 1e6); hi = max(e for _, e, _ in iv)
small = defaultdict(float)
for s, e, n in iv:
    if e <= lo: continue
    d = (e - s) / 1e6
    if d >= min_ms: print(f"{(s - lo) / 1e6:8.2f} ms  +{d:7.2f}  {n.split('(')[0][:60]}")
    else: small[int((s - lo) / 5e6)] += d
print("kernels below", min_ms, "ms, summed per 5 ms bucket:", {5 * b: round(v, 2) for b, v in sorted(small.items())})
print("window", round((hi - lo) / 1e6, 1), "ms")
