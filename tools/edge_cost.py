#!/usr/bin/env python3
"""Cost of splitting a sweep into interior + two LAG-wide boundary strips (what a tile with two remote sides along
the sweep axis launches to overlap its halo exchange), against the single full sweep. One GPU, no communication."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd
from armon_amd.blocking import Axis
from armon_amd.solver import BlockGrid, fused_sweep, init_test, sweep_lag

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
params = armon_amd.ArmonParameters(test="Sod", N=(n, n), silent=5, maxcycle=10)
grid = BlockGrid(params)
init_test(params, grid)
dev = params.device
dx = params.domain_size[0] / n
dt = 0.3 * dx
lag = sweep_lag(params)
for axis in (Axis.X, Axis.Y):
    res = {"full": [], "interior": [], "edges": []}
    for r in range(12):
        for what, ranges in (("full", [None]), ("interior", [(lag, n - lag)]), ("edges", [(0, lag), (n - lag, n)])):
            dev.event_record(0)
            for k, rg in enumerate(ranges):
                fused_sweep(params, grid, axis, dt, dx, emit_dt=axis == Axis.Y, out_range=rg, swap=False,
                            dt_accumulate=what == "edges")
            dev.event_record(1)
            if r >= 2:
                res[what].append(dev.event_elapsed_ms(0, 1))
    print(f"sweep_{axis.name}: " + "  ".join(f"{k} {statistics.median(v):.3f} ms" for k, v in res.items()))
