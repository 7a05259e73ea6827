#!/usr/bin/env python3
"""trim_stats.py <kernel_stats.csv> <out.csv> -- rocprofv3 --stats table with kernel names cut to the function name
(+ template arguments); rows of the same trimmed name are merged."""
import csv
import sys
from collections import OrderedDict

rows = OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].split("(")[0]
    if name.startswith("void "):
        name = name[5:]
    name = name[:70]
    a = rows.setdefault(name, {"Calls": 0, "TotalDurationNs": 0, "MinNs": None, "MaxNs": 0})
    a["Calls"] += int(r["Calls"]); a["TotalDurationNs"] += int(float(r["TotalDurationNs"]))
    a["MinNs"] = int(float(r["MinNs"])) if a["MinNs"] is None else min(a["MinNs"], int(float(r["MinNs"])))
    a["MaxNs"] = max(a["MaxNs"], int(float(r["MaxNs"])))
tot = sum(a["TotalDurationNs"] for a in rows.values()) or 1
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for name, a in sorted(rows.items(), key=lambda kv: -kv[1]["TotalDurationNs"]):
    w.writerow([name, a["Calls"], a["TotalDurationNs"], a["TotalDurationNs"] // max(1, a["Calls"]), f"{100.0 * a['TotalDurationNs'] / tot:.3f}", a["MinNs"], a["MaxNs"]])
