#!/usr/bin/env python3
"""clasp_classes.py <kernel_trace.csv>: per (lf_clasp_kernel instantiation, LDS bytes) launch statistics -- the LDS size
identifies the size class of a launch."""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        name = r.get("Kernel_Name", "")
        if "lf_clasp_kernel" not in name:
            continue
        lds = r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?"))
        rows[(name.split("(")[0][-24:], lds)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("kernel, lds_bytes, launches, total_ms, avg_ms, max_ms")
for (k, lds), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k}, {lds}, {len(v)}, {sum(v):.1f}, {sum(v) / len(v):.2f}, {max(v):.2f}")
