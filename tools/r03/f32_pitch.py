#!/usr/bin/env python3
"""fp32 fused sweeps against the row pitch: with nghost = 4 a row of 16384 cells is 16392 floats = 32 B past a multiple of
64 B, so NO column shift puts every row's 512-B wave segments on sector boundaries (they are for fp64: the pitch is 64 B past
a multiple of 128). Same cell count, pitches ≡ 8 / 0 (mod 16 floats).
    python tools/r03/f32_pitch.py [--rounds 10]"""
import argparse
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import armon_amd
from armon_amd.blocking import Axis
from armon_amd.solver import STATE_VARS, BlockGrid, init_test, sweep_desc

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--dtype", default="float32")
ap.add_argument("--cases", default="16384:4,16384:8,16392:4,16376:4,16384:6")
args = ap.parse_args()
for case in args.cases.split(","):
    nx, g = (int(v) for v in case.split(":"))
    ny = 16384
    params = armon_amd.ArmonParameters(test="Sod", N=(nx, ny), nghost=g, silent=5, maxcycle=10, data_type=args.dtype)
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
    cells = nx * ny
    print(f"{args.dtype} nx {nx:6d} nghost {g}: pitch {nx + 2 * g} (mod 16: {(nx + 2 * g) % 16:2d})  X {med['X']:.3f} ms  Y {med['Y']:.3f} ms  copy {med['copy']:.3f} ms   "
          f"Y/copy {med['Y'] / med['copy']:.3f}  X/copy {med['X'] / med['copy']:.3f}   {2 * cells / (med['X'] + med['Y']) / 1e6:.1f} Gcells/s", flush=True)
    del grid, params
