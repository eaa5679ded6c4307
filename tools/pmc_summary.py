#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: mean counter value per kernel name (substring filter)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
filt = sys.argv[2:] or [""]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not any(s in k for s in filt):
            continue
        short = k.replace("(anonymous namespace)::", "").replace("armon::fused::", "").replace("void ", "")
        short = short.split("(")[0][:70]
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} mean {sum(v) / len(v):16.1f}  (n={len(v)})")
