#!/usr/bin/env python3
"""Per-kernel statistics of the TIMED REGION of a bench.py run from a rocprofv3 kernel trace: the last `n` launches
of each sweep kernel (the 20 timed cycles; the earlier ones are warm-up and BlockGrid.tune_placement's trials).

    tools/trace_timed_region.py <..._kernel_trace.csv> <out.json> [n=20] [kernel substring ...]
"""
import csv
import json
import statistics
import sys

trace, out = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
subs = sys.argv[4:] or ["k_sweep", "k_euler_projection"]
by = {}
for row in csv.DictReader(open(trace)):
    k = row["Kernel_Name"]
    if any(s in k for s in subs):
        by.setdefault(k, []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
res = {"source": trace.split("/")[-1], "launches_kept": n, "kernels": {}}
for k, v in by.items():
    v.sort()
    d = [x[1] for x in v[-n:]]
    short = k.replace("(anonymous namespace)::", "").replace("armon::fused::", "").replace("void ", "").split("(")[0]
    res["kernels"][short] = {"calls_in_trace": len(v), "avg_ns": statistics.mean(d), "median_ns": statistics.median(d),
                             "min_ns": min(d), "max_ns": max(d)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
