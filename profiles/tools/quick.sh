#!/bin/bash
# quick.sh <tag> [pytest files...] -- parity tests of the files given, then the C2 bench line (no CPU baseline), summary on stdout
TAG=$1; shift
mkdir -p gpurun_out
if [ $# -gt 0 ]; then timeout 1200 python -m pytest "$@" -x -q 2>&1 | tail -3; fi
python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>gpurun_out/$TAG.err > gpurun_out/$TAG.json
python3 - <<PY
import json
d=json.loads(open("gpurun_out/$TAG.json").read().strip().splitlines()[-1])
print("$TAG", round(d["value"]), round(d["ms_per_step"],1), round(d.get("value_pcie_inclusive") or 0), d.get("all_records_match_rate"))
r=d["roofline"]
print({k:round(v["ms_per_step"],2) for k,v in r["by_kernel"].items()}, round(r["exclusive_ms_sum_all_kernels"],1))
PY
