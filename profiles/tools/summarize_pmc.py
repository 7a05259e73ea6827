#!/usr/bin/env python3
"""summarize_pmc.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...]
Per kernel (template arguments kept, parameter lists dropped): dispatches and the SUM over dispatches of every counter
found in the rocprofv3 --pmc CSVs (FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports them)."""
import csv
import json
import sys
from collections import defaultdict

out = defaultdict(lambda: {"launches": 0})
for path in sys.argv[2:]:
    seen = defaultdict(set)
    dur = defaultdict(float)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if name.startswith("void "):
            name = name[5:]
        if len(name) > 80:
            name = name[:80]
        ctr = r["Counter_Name"]
        key = ctr.lower().replace("_size", "_kb")
        out[name][key] = out[name].get(key, 0.0) + float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[name] and r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[name] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])        # ns, under the counter pass (serialised dispatches)
        seen[name].add(r["Dispatch_Id"])
    for name, ids in seen.items():
        out[name]["launches"] = max(out[name]["launches"], len(ids))
        if dur[name] > 0 and "grbm_gui_active" in out[name]:
            out[name]["dur_ns_in_the_grbm_pass"] = dur[name]
            out[name]["clock_ghz"] = out[name]["grbm_gui_active"] / 8.0 / dur[name]      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
res = dict(sorted(out.items(), key=lambda kv: -(kv[1].get("fetch_kb", 0) + kv[1].get("write_kb", 0))))
# which kernels these counters describe: digest of lordfast_amd/csrc (bench.py --tree-hash) + the git commit if there is one.
# bench.py only quotes `roofline.traffic` from a summary whose source_tree is the tree it runs on.
import os
import subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
meta = {}
try:
    meta["source_tree"] = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--tree-hash"], capture_output=True, text=True, timeout=60).stdout.strip()
except Exception:                                                        # noqa: BLE001
    pass
try:
    meta["git_commit"] = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
except Exception:                                                        # noqa: BLE001
    pass
if os.environ.get("LF_PMC_READS"):
    meta["reads_per_step"] = int(os.environ["LF_PMC_READS"])
res = {"_meta": meta, **res}
json.dump(res, open(sys.argv[1], "w"), indent=1)
print("wrote", sys.argv[1], len(out), "kernels")
