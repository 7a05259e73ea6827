#!/usr/bin/env python3
"""clasp_classes.py <kernel_trace.csv>: launch statistics of lf_clasp_kernel per size class."""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
CLASSES = ["<=1024", "<=768", "<=512", "<=384", "<=256", "<=192", "<=128", "<=64"]     # launch order in lf_vote.hip
NCLASS = len(CLASSES)
seq = {}
with open(sys.argv[1]) as fh:
    for r in sorted(csv.DictReader(fh), key=lambda r: int(r["Dispatch_Id"])):
        name = r.get("Kernel_Name", "")
        if "lf_clasp_kernel" not in name:
            continue
        # dynamic LDS is not reported by the trace; every size class runs on its own stream of its lane, largest class
        # first, so the position of a launch inside its group of consecutive clasp launches of one thread gives the class
        lds = seq.setdefault(r.get("Thread_Id", "?"), [0])
        cls = lds[0] % NCLASS; lds[0] += 1
        rows[(name.split("(")[0][-24:], CLASSES[cls])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("kernel, size class (fragments), launches, total_ms, avg_ms, max_ms")
for (k, lds), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k}, {lds}, {len(v)}, {sum(v):.1f}, {sum(v) / len(v):.2f}, {max(v):.2f}")
