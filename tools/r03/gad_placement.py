#!/usr/bin/env python3
"""What the staged acoustic_GAD kernel does when ITS six arrays are well placed: the kernel is timed over random
assignments of (us, ps, rho, u, p, c) to a pool of equal allocations (the staged path's own placement search optimises the
whole cycle, not this kernel). Prints the distribution per axis.
    python tools/r03/gad_placement.py [--n 16384] [--tries 24] [--pool 24]"""
import argparse
import ctypes as C
import os
import random
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import armon_amd
from armon_amd import _lib
from armon_amd.blocking import Axis
from armon_amd.solver import BlockGrid, init_test, update_EOS

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--tries", type=int, default=24)
ap.add_argument("--pool", type=int, default=24)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
n = args.n
params = armon_amd.ArmonParameters(test="Sod", N=(n, n), silent=5, use_fused_sweep=False, placement_tries=0)
grid = BlockGrid(params)
init_test(params, grid)
update_EOS(params, grid)
params.wait()
dev = params.device
L = armon_amd.lib()
dx = 1.0 / n
dt = 0.3 * dx
names = ("us", "ps", "rho", "u", "p", "c")
# pool: the 16 vectors of the block + spares, every one a copy of a valid field (positive, smooth)
pool = [grid.data[f] for f in ("rho", "c", "p", "E")]
src = {"rho": grid.data["rho"], "c": grid.data["c"], "p": grid.data["p"], "u": grid.data["c"]}
others = [grid.data[f] for f in armon_amd.solver.FIELDS if f not in ("rho", "c", "p", "E")]
spares = [dev.empty(pool[0].n, pool[0].dtype) for _ in range(max(0, args.pool - 16))]
free_pool = others + spares
for v in free_pool:
    v.copy_from_device(grid.data["rho"])
vectors = pool + free_pool
rng = random.Random(1234)
res = {"X": [], "Y": []}
for t in range(args.tries):
    pick = rng.sample(range(len(vectors)), 6)
    ptr = {k: C.c_void_p(vectors[i].ptr) for k, i in zip(names, pick)}
    for axis in (Axis.X, Axis.Y):
        rg = params.block_size.domain_range(*params.steps_ranges[axis].fluxes).to_c()
        s = params.block_size.stride_along(axis)
        ms_all = []
        for r in range(args.reps + 1):
            _lib.check(L.armon_hip_event_record(dev.ctx, 0))
            _lib.check(L.armon_hip_acoustic_GAD(dev.ctx, rg, s, dt, dx, ptr["us"], ptr["ps"], ptr["rho"], ptr["u"], ptr["p"], ptr["c"], 1))
            _lib.check(L.armon_hip_event_record(dev.ctx, 1))
            ms = C.c_double()
            _lib.check(L.armon_hip_event_elapsed_ms(dev.ctx, 0, 1, C.byref(ms)))
            if r:
                ms_all.append(ms.value)
        res[axis.name].append(statistics.median(ms_all))
for axis, v in res.items():
    v = sorted(v)
    f = lambda ms: 48 * n * n / ms / 1e6 / 8000
    print(f"acoustic_GAD_{axis.lower()}: best {v[0]:.3f} ms ({f(v[0]):.3f} of 8 TB/s)  quartile {v[len(v)//4]:.3f}  median {statistics.median(v):.3f} ({f(statistics.median(v)):.3f})"
          f"  worst {v[-1]:.3f} ({f(v[-1]):.3f})   over {len(v)} placements of the six arrays")
