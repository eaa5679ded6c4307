#!/usr/bin/env python3
"""Time the staged acoustic_GAD and advection_second_order kernels along x and along y across several builds of the library
(round-robin per launch, one process, one placement).   python tools/r05/staged_y_probe.py [--n 16384] name=path ..."""
import argparse
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import armon_amd
from armon_amd import _lib
from armon_amd.blocking import Axis
from armon_amd.solver import BlockGrid, init_test, update_EOS

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("libs", nargs="+")
args = ap.parse_args()
n = args.n
params = armon_amd.ArmonParameters(test="Sod", N=(n, n), silent=5, use_fused_sweep=False, placement_tries=0)
grid = BlockGrid(params)
init_test(params, grid)
update_EOS(params, grid)
params.wait()
dx = 1.0 / n
dt = 0.3 * dx
builds = []
for spec in args.libs:
    name, path = spec.split("=", 1)
    L = _lib.load_at(os.path.abspath(path))
    ctx = C.c_void_p()
    _lib.check(L.armon_hip_init(0, None, C.byref(ctx)))
    builds.append((name, L, ctx))
p = grid.ptr
res = {}
for r in range(args.rounds + 2):
    for axis in (Axis.X, Axis.Y):
        s = params.block_size.stride_along(axis)
        ua = p("u") if axis == Axis.X else p("v")
        rg = params.block_size.domain_range(*params.steps_ranges[axis].fluxes).to_c()
        ra = params.block_size.domain_range(*params.steps_ranges[axis].advection).to_c()
        for name, L, ctx in builds:
            for kern, call in (("acoustic_GAD", lambda: L.armon_hip_acoustic_GAD(ctx, rg, s, dt, dx, p("us"), p("ps"), p("rho"), ua, p("p"), p("c"), 1)),
                               ("advection_2nd", lambda: L.armon_hip_advection_second_order(ctx, ra, s, dx, dt, p("us"), p("rho"), p("u"), p("v"), p("E"),
                                                                                            p("work_1"), p("work_2"), p("work_3"), p("work_4")))):
                _lib.check(L.armon_hip_event_record(ctx, 0))
                _lib.check(call())
                _lib.check(L.armon_hip_event_record(ctx, 1))
                ms = C.c_double()
                _lib.check(L.armon_hip_event_elapsed_ms(ctx, 0, 1, C.byref(ms)))
                if r >= 2:
                    res.setdefault((kern, axis.name, name), []).append(ms.value)
B = {"acoustic_GAD": 48, "advection_2nd": 72}
for (kern, axis, name), v in sorted(res.items()):
    med = statistics.median(v)
    print(f"{kern}_{axis.lower()} {name:10s}: median {med:7.3f} ms  min {min(v):7.3f}   {B[kern] * n * n / med / 1e6:7.1f} GB/s  ({B[kern] * n * n / med / 1e6 / 8000:.3f} of 8 TB/s)")
