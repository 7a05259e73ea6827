#!/usr/bin/env python3
"""timeline.py <kernel_trace.csv> --last-step <chunks per step> -- per queue (= lane stream) the sequence of kernels of the last
step with start offsets and durations (ms), to see what a chunk waits for between its stages."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[3])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", "?")) for r in rows)
starts = [s for s, _, n, _ in iv if n.startswith("lf_seed_search_kernel")]
lo = starts[-k]
by_q = defaultdict(list)
for s, e, n, q in iv:
    if s >= lo:
        by_q[q].append((s, e, n))
for q, lst in sorted(by_q.items(), key=lambda kv: kv[1][0][0]):
    big = [x for x in lst if x[1] - x[0] > 150e3 or x[2].startswith(("lf_ksw", "lf_walk", "lf_render", "lf_sam", "lf_seed_search", "lf_vote_hash"))]
    if len(big) < 3:
        continue
    print(f"queue {q}: {len(lst)} launches")
    prev_e = None
    for s, e, n in big:
        gap = "" if prev_e is None else f"  (+{(s - prev_e) / 1e6:6.2f} after previous listed)"
        print(f"   {(s - lo) / 1e6:8.2f} ms  {(e - s) / 1e6:7.2f} ms  {n}{gap}")
        prev_e = e
