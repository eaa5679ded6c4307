#!/usr/bin/env python3
"""Is the HBM placement effect DETERMINISTIC inside one allocation? 16 vectors are carved out of ONE slab at
chosen offsets; armon_hip_choose_placement (fixed random sequence, no early exit) times the same 24 role assignments
in every process. If draw i costs the same in every process and on every box, a fixed (offsets, roles) table can
replace the measured search.

    python tools/slab_probe.py [--n 16384] [--tries 24] [--patterns zero,mib,rand,malloc]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd
from armon_amd import _lib
from armon_amd.blocking import Axis
from armon_amd.device import DeviceArray
from armon_amd.solver import BlockGrid, sweep_desc

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--tries", type=int, default=24)
ap.add_argument("--patterns", default="zero,mib,rand,malloc")
ap.add_argument("--pool", type=int, default=16)
args = ap.parse_args()

MiB = 1 << 20
params = armon_amd.ArmonParameters(test="Sod", N=(args.n, args.n), silent=5, maxcycle=10, placement_tries=0)
grid = BlockGrid(params)          # shapes for the descriptors only; its own vectors are not used
dev = params.device
n, dt_ = grid.data["rho"].n, grid.data["rho"].dtype
nbytes = grid.data["rho"].nbytes
dx = params.cell_size(0)
d_x = sweep_desc(params, grid, Axis.X, 1e-3 * dx, dx)
d_y = sweep_desc(params, grid, Axis.Y, 1e-3 * dx, params.cell_size(1), emit_dt=True)
S = -(-nbytes // (16 * MiB)) * 16 * MiB + 16 * MiB          # stride ≡ 0 mod 16 MiB, room for a 16-MiB offset


def offsets(pattern, k):
    if pattern == "zero":
        return 0
    if pattern == "mib":
        return (k * MiB) % (16 * MiB)
    x = (k * 2654435761 + 12345) & 0xFFFFFFFF           # "rand": fixed pseudo-random multiples of 64 KiB
    x ^= x >> 13
    return (x % 256) * 64 * 1024


def run(ptrs):
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    picks, times, done = (C.c_int * 8)(), (C.c_double * args.tries)(), C.c_int(0)
    _lib.check(params.fn("choose_placement")(dev.ctx, C.byref(d_x), C.byref(d_y), arr, len(ptrs), nbytes, args.tries, 0.0,
                                             C.byref(picks), times, C.byref(done)))
    return list(times)[:done.value], list(picks)


for pattern in args.patterns.split(","):
    if pattern == "malloc":
        vecs = [dev.empty(n, dt_) for _ in range(args.pool)]
        ptrs = [v.ptr for v in vecs]
    else:
        slab = DeviceArray(dev, args.pool * S, "uint8")
        ptrs = [slab.ptr + k * S + offsets(pattern, k) for k in range(args.pool)]
    t, picks = run(ptrs)
    print(f"{pattern:7s} base%16MiB={ptrs[0] % (16 * MiB) // 1024:6d}KiB  " + " ".join(f"{v:.2f}" for v in t) + f"   best picks {picks}", flush=True)
    if pattern == "malloc":
        for v in vecs:
            v.free()
    else:
        slab.free()
