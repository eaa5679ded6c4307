#!/usr/bin/env python3
"""How much does the physical placement of the 8 state vectors matter? Allocate several sets (each set held
while the next is made, so every set lands elsewhere), run the same X / Y sweeps and the 4-in/4-out copy on each.

    python tools/placement_lottery.py [--n 16384] [--sets 8] [--reps 5]
"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd
from armon_amd.blocking import Axis
from armon_amd.solver import BlockGrid, STATE_VARS, fused_sweep, init_test

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--sets", type=int, default=8)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()

params = armon_amd.ArmonParameters(test="Sod", N=(args.n, args.n), silent=5, maxcycle=10, placement_tries=0)
dev = params.device
dx = params.domain_size[0] / args.n
dt = 0.3 * dx
grid = BlockGrid(params)
init_test(params, grid)
held = []
for s in range(args.sets):
    if s:
        new_data = {f: dev.empty(len(grid.data[f]), params.data_type) for f in STATE_VARS}
        new_alt = {f: dev.empty(len(grid.data[f]), params.data_type) for f in STATE_VARS}
        for f in STATE_VARS:
            new_data[f].copy_from_device(grid.data[f])
            held += [grid.data[f], grid.alt[f]]
            grid.data[f], grid.alt[f] = new_data[f], new_alt[f]
    t = {"X": [], "Y": [], "copy": []}
    for r in range(args.reps + 1):
        for axis in (Axis.X, Axis.Y):
            dev.event_record(0)
            fused_sweep(params, grid, axis, dt, dx, emit_dt=axis == Axis.Y, swap=False)
            dev.event_record(1)
            if r:
                t[axis.name].append(dev.event_elapsed_ms(0, 1))
        dev.event_record(0)
        dev.stream_copy4([grid.data[f] for f in STATE_VARS], [grid.alt[f] for f in STATE_VARS], grid.data["rho"].nbytes & ~15)
        dev.event_record(1)
        if r:
            t["copy"].append(dev.event_elapsed_ms(0, 1))
    vas = [grid.data[f].ptr for f in STATE_VARS] + [grid.alt[f].ptr for f in STATE_VARS]
    print(f"set {s}: X {statistics.median(t['X']):6.3f}  Y {statistics.median(t['Y']):6.3f}  copy {statistics.median(t['copy']):6.3f} ms   "
          f"VA/2MiB: " + " ".join(f"{v >> 21:x}" for v in vas), flush=True)
