#!/usr/bin/env python3
"""Summary table of tools/probes/probe_layout runs (tools/r04/layout_runs.sh): per layout, the spread of the X+Y pair's
TB/s over all instances and processes, and in how many processes the layout's BEST / EVERY instance reaches a level.
usage: layout_summary.py gpurun_out/r04_layout_probe.txt"""
import collections
import re
import sys

rows = collections.defaultdict(list)
proc = 0
for l in open(sys.argv[1]):
    if l.startswith('## process'):
        proc = int(l.split()[2])
        continue
    m = re.match(r'(.+?)\s+X ([\d.]+)\s+Y ([\d.]+)\s+X\+Y ([\d.]+) ms\s+([\d.]+) TB/s', l)
    if m:
        tag = re.sub(r' #\d+', '', m.group(1)).strip()
        rows[tag].append((proc, float(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5))))
print(f"{'layout':30s}  n  X+Y ms min/med/max     TB/s min/med/max   processes with EVERY instance >= 6.0 | best instance >= 6.0 | best >= 6.2")
for tag, v in rows.items():
    ms = sorted(x[3] for x in v)
    tb = sorted(x[4] for x in v)
    best, worst = collections.defaultdict(float), collections.defaultdict(lambda: 99.)
    for p, _, _, _, t in v:
        best[p] = max(best[p], t)
        worst[p] = min(worst[p], t)
    n = len(best)
    print(f"{tag:30s} {len(v):3d}  {ms[0]:.3f} {ms[len(ms)//2]:.3f} {ms[-1]:.3f}   {tb[0]:.2f} {tb[len(tb)//2]:.2f} {tb[-1]:.2f}     "
          f"{sum(w >= 6.0 for w in worst.values()):2d}/{n}   {sum(b >= 6.0 for b in best.values()):2d}/{n}   {sum(b >= 6.2 for b in best.values()):2d}/{n}")
