import json,sys
b=json.load(open(sys.argv[1])); print(b["value"], b["ms_per_step"], b.get("all_records_match_rate"), b.get("host_cpu_seconds_per_step")); r=b["roofline"]; print(r["exclusive_ms_per_step"], r["exclusive_ms_sum_all_kernels"]); print({k:round(v["ms_per_step"],1) for k,v in r["by_kernel"].items()})
