#!/usr/bin/env python3
"""Which assignments of the 8 sweep roles to separately allocated vectors are fast? Times explicit arrangements
(indices in allocation order) with armon_hip_choose_placement(tries=1): roles 0-3 = read by X / written by Y,
roles 4-7 = written by X / read by Y."""
import ctypes as C
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd
from armon_amd import _lib
from armon_amd.blocking import Axis
from armon_amd.solver import BlockGrid, sweep_desc

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 16384
params = armon_amd.ArmonParameters(test="Sod", N=(n, n), silent=5, placement_tries=0)
grid = BlockGrid(params)
dev = params.device
nel, dt_ = grid.data["rho"].n, grid.data["rho"].dtype
vec = [dev.empty(nel, dt_) for _ in range(M)]
base = min(v.ptr for v in vec)
print("offsets from the lowest address (MiB):", [round((v.ptr - base) / 2 ** 20, 3) for v in vec])
dx = params.cell_size(0)
d_x = sweep_desc(params, grid, Axis.X, 1e-3 * dx, dx)
d_y = sweep_desc(params, grid, Axis.Y, 1e-3 * dx, dx, emit_dt=True)


def time_arrangement(idx):
    ptrs = (C.c_void_p * 8)(*[vec[i].ptr for i in idx])
    picks, times, done = (C.c_int * 8)(), (C.c_double * 1)(), C.c_int()
    _lib.check(params.fn("choose_placement")(dev.ctx, C.byref(d_x), C.byref(d_y), ptrs, 8, vec[0].nbytes, 1, 0.0,
                                            C.byref(picks), times, C.byref(done)))
    return times[0]


named = {
    "identity 0-7": list(range(8)),
    "interleaved R even / W odd": [0, 2, 4, 6, 1, 3, 5, 7],
    "R 0-3, W 8-11": [0, 1, 2, 3, 8, 9, 10, 11],
    "R even 0-6, W even 8-14": [0, 2, 4, 6, 8, 10, 12, 14],
    "R 0,4,8,12 W 2,6,10,14": [0, 4, 8, 12, 2, 6, 10, 14],
    "reversed 7-0": list(range(7, -1, -1)),
    "R 4-7, W 0-3": [4, 5, 6, 7, 0, 1, 2, 3],
    "R 0,1,4,5 W 2,3,6,7": [0, 1, 4, 5, 2, 3, 6, 7],
}
for name, idx in named.items():
    if max(idx) < M:
        print(f"{time_arrangement(idx):7.3f} ms  {name}  {idx}", flush=True)
rng = random.Random(1)
res = []
for _ in range(60):
    idx = rng.sample(range(M), 8)
    res.append((time_arrangement(idx), idx))
for t, idx in sorted(res):
    print(f"{t:7.3f} ms  random {idx}  sorted {sorted(idx)}", flush=True)
