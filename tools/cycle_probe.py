#!/usr/bin/env python3
"""Whole-cycle kernel (armon_hip_cycle_xy: X sweep + Y sweep in one pass) against the two fused sweeps: same bits,
and the time of both forms, from the A/B build libarmon_hip_alt.so.   python tools/cycle_probe.py"""
import argparse
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import armon_amd
from armon_amd import _lib
from armon_amd.blocking import Axis
from armon_amd.solver import STATE_VARS, BlockGrid, init_test, sweep_desc

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--seg", type=int, default=0)
args = ap.parse_args()
# the whole-cycle kernels are compiled into the A/B build only (libarmon_hip_alt.so, -DARMON_ALT_KERNELS)
_alt = _lib.alt_kernels()
L = _alt.__enter__()


def check_small(test, N):
    params = armon_amd.ArmonParameters(test=test, N=N, silent=5, maxcycle=10)
    grid = BlockGrid(params)
    init_test(params, grid)
    dev = params.device
    dx, dy = params.cell_size(0), params.cell_size(1)
    from armon_amd.solver import local_time_step, update_EOS
    update_EOS(params, grid)
    dt = params.cfl * local_time_step(params, grid)          # the reference's first time step
    for emit_p in (False, True):
        d_x = sweep_desc(params, grid, Axis.X, dt, dx)
        d_y = sweep_desc(params, grid, Axis.Y, dt, dy, emit_dt=True, emit_p=emit_p)
        # whole cycle: data -> alt
        _lib.check(L.armon_hip_cycle_xy(dev.ctx, C.byref(d_x), C.byref(d_y)))
        got = {f: grid.alt[f].to_host() for f in STATE_VARS}
        got_dt = grid.dt_scalar.to_host()[0]
        got_p = grid.data["p"].to_host() if emit_p else None
        # two sweeps: data -> alt (X), alt -> tmp (Y)
        tmp = {f: dev.empty(grid.data[f].n, grid.data[f].dtype) for f in STATE_VARS}
        _lib.check(params.fn("sweep")(dev.ctx, C.byref(d_x)))
        d_y.rho_in, d_y.u_in, d_y.v_in, d_y.E_in = (grid.alt[f].ptr for f in STATE_VARS)
        d_y.rho_out, d_y.u_out, d_y.v_out, d_y.E_out = (tmp[f].ptr for f in STATE_VARS)
        _lib.check(params.fn("sweep")(dev.ctx, C.byref(d_y)))
        ref_dt = grid.dt_scalar.to_host()[0]
        for f in STATE_VARS:
            a, b = grid.real_view(got[f]), grid.real_view(tmp[f].to_host())
            assert np.array_equal(a, b), (test, N, f, np.abs(a - b).max(), np.argwhere(a != b)[:5])
        assert got_dt == ref_dt, (got_dt, ref_dt)
        if emit_p:
            assert np.array_equal(grid.real_view(got_p), grid.real_view(grid.data["p"].to_host()))
    print("bits equal:", test, N, flush=True)


for test, N in (("Sod_circ", (300, 200)), ("Sod_circ", (57, 61)), ("Sedov", (123, 77)), ("Sod", (8, 500)), ("Sod_y", (500, 9))):
    check_small(test, N)

params = armon_amd.ArmonParameters(test="Sod", N=(args.n, args.n), silent=5, maxcycle=10)
grid = BlockGrid(params)
init_test(params, grid)
dev = params.device
if args.seg:
    _lib.check(L.armon_hip_set_tuning(dev.ctx, b"ARMON_Y_SEG", args.seg))
dx = params.cell_size(0)
dt = 0.3 * dx
res = {"X+Y": [], "cycle_xy": []}
d_x = sweep_desc(params, grid, Axis.X, dt, dx)
d_y = sweep_desc(params, grid, Axis.Y, dt, dx, emit_dt=True)
d_y2 = sweep_desc(params, grid, Axis.Y, dt, dx, emit_dt=True)
d_y2.rho_in, d_y2.u_in, d_y2.v_in, d_y2.E_in = (grid.alt[f].ptr for f in STATE_VARS)
d_y2.rho_out, d_y2.u_out, d_y2.v_out, d_y2.E_out = (grid.data[f].ptr for f in STATE_VARS)
for r in range(args.rounds + 2):
    dev.event_record(0)
    _lib.check(params.fn("sweep")(dev.ctx, C.byref(d_x)))
    _lib.check(params.fn("sweep")(dev.ctx, C.byref(d_y2)))
    dev.event_record(1)
    a = dev.event_elapsed_ms(0, 1)
    dev.event_record(0)
    _lib.check(L.armon_hip_cycle_xy(dev.ctx, C.byref(d_x), C.byref(d_y)))
    dev.event_record(1)
    b = dev.event_elapsed_ms(0, 1)
    init_test(params, grid, tune=False)
    if r >= 2:
        res["X+Y"].append(a)
        res["cycle_xy"].append(b)
cells = args.n * args.n
for k, v in res.items():
    med = statistics.median(v)
    print(f"{k:9s}: median {med:7.3f} ms  min {min(v):7.3f}   {2 * cells / med / 1e6:8.1f} Mcells/s per sweep", flush=True)
