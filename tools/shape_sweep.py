#!/usr/bin/env python3
"""Time the fused X / Y sweeps and the 4-in/4-out copy on grids of the same cell count but different aspect:
does the Y march (one new row, i.e. one jump of a whole pitch, per step) depend on the row pitch?

    python tools/shape_sweep.py [--cells-log2 28] [--rounds 10]
"""
import argparse
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd
from armon_amd.blocking import Axis
from armon_amd.solver import STATE_VARS, BlockGrid, init_test, sweep_desc

ap = argparse.ArgumentParser()
ap.add_argument("--cells-log2", type=int, default=28)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--shapes", default="2048,4096,8192,16384,32768,65536")
args = ap.parse_args()
total = 1 << args.cells_log2
for nx in (int(v) for v in args.shapes.split(",")):
    ny = total // nx
    params = armon_amd.ArmonParameters(test="Sod", N=(nx, ny), silent=5, maxcycle=10)
    grid = BlockGrid(params)
    init_test(params, grid)
    dev = params.device
    dt = 0.3 * params.domain_size[0] / max(nx, ny)
    res = {"X": [], "Y": [], "copy": []}
    src, dst = [grid.data[f] for f in STATE_VARS], [grid.alt[f] for f in STATE_VARS]
    nb = src[0].nbytes & ~15
    for r in range(args.rounds + 2):
        for axis in (Axis.X, Axis.Y):
            d = sweep_desc(params, grid, axis, dt, params.cell_size(int(axis) - 1), emit_dt=axis == Axis.Y)
            dev.event_record(0)
            armon_amd._lib.check(params.fn("sweep")(dev.ctx, C.byref(d)))
            dev.event_record(1)
            if r >= 2:
                res[axis.name].append(dev.event_elapsed_ms(0, 1))
        dev.event_record(0)
        dev.stream_copy4(src, dst, nb)
        dev.event_record(1)
        if r >= 2:
            res["copy"].append(dev.event_elapsed_ms(0, 1))
    med = {k: statistics.median(v) for k, v in res.items()}
    print(f"{nx:6d} x {ny:6d}: X {med['X']:.3f} ms  Y {med['Y']:.3f} ms  copy {med['copy']:.3f} ms   "
          f"Y/copy {med['Y'] / med['copy']:.3f}  X/copy {med['X'] / med['copy']:.3f}  placement {grid.placement and grid.placement['x_plus_y_ms']}", flush=True)
    del grid, params
