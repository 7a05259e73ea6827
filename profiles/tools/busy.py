#!/usr/bin/env python3
"""busy.py <kernel_trace.csv> [t0_frac t1_frac | --last-step <chunks per step>] -- union of kernel execution intervals from a rocprofv3 --kernel-trace CSV:
how much of the wall time had at least one kernel running, and each kernel family's share of the summed durations."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
lo, hi = iv[0][0], max(e for _, e, _ in iv)
if len(sys.argv) > 3 and sys.argv[2] == "--last-step":
    # window = the last K launches of the seed search kernel (K = chunks per step) up to the last kernel of the trace
    k = int(sys.argv[3])
    starts = [s for s, _, n in iv if "lf_seed_search_kernel" in n.split("(")[0]]
    lo = starts[-k]
elif len(sys.argv) > 3:
    a, b = float(sys.argv[2]), float(sys.argv[3])
    lo, hi = lo + int((hi - lo) * a), lo + int((hi - lo) * b)
busy = 0
cur_s = cur_e = None
fam = defaultdict(int)
for s, e, n in iv:
    s, e = max(s, lo), min(e, hi)
    if e <= s:
        continue
    fam[n.split("(")[0][:48]] += e - s
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    busy += cur_e - cur_s
wall = hi - lo
print(f"window {wall / 1e6:.1f} ms, >=1 kernel running {busy / 1e6:.1f} ms ({100.0 * busy / wall:.1f} %), sum of durations {sum(fam.values()) / 1e6:.1f} ms")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:24]:
    print(f"  {k:48s} {v / 1e6:9.1f} ms")
