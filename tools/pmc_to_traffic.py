#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-launch HBM traffic per kernel.

Correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a coalesced streaming read, WRITE_SIZE reads exactly:
    traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  bytes per launch.
usage: pmc_to_traffic.py <pmc dir> <out.json> <workload note> [kernel substring ...]"""
import collections
import csv
import glob
import json
import os
import sys

root, out, note = sys.argv[1], sys.argv[2], sys.argv[3]
filt = sys.argv[4:] or ["k_sweep"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if any(s in k for s in filt) and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            short = k.replace("(anonymous namespace)::", "").replace("armon::fused::", "").replace("void ", "").split("(")[0]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {"workload": note, "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes (gfx950: FETCH_SIZE counts half)", "kernels": {}}
for k, cs in acc.items():
    f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
    w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
    res["kernels"][k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "launches": len(cs["FETCH_SIZE"]),
                         "traffic_bytes_per_launch": (2 * f + w) * 1024}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
