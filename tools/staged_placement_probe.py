#!/usr/bin/env python3
"""Does the assignment of the 16 equally sized BlockData allocations to the 16 fields matter for the STAGED kernels?
Times one staged cycle (EOS, fluxes, cell update, advection, projection; X then Y) per kernel for random permutations."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd
from armon_amd._lib import FIELDS
from armon_amd.blocking import Axis
from armon_amd import solver as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
tries = int(sys.argv[2]) if len(sys.argv) > 2 else 10
params = armon_amd.ArmonParameters(test="Sod", N=(n, n), silent=5, use_fused_sweep=False, maxcycle=5)
grid = S.BlockGrid(params)
dev = params.device
vectors = [grid.data[f] for f in FIELDS]


class Timer:
    def __init__(self):
        self.t, self.k = {}, 0

    def start(self, name):
        dev.event_record(2 * self.k)

    def end(self, name):
        dev.event_record(2 * self.k + 1)
        self.t.setdefault(name, []).append((2 * self.k, 2 * self.k + 1))
        self.k += 1


rng = random.Random(7)
for t in range(tries):
    perm = list(range(16)) if t == 0 else rng.sample(range(16), 16)
    for f, k in zip(FIELDS, perm):
        grid.data[f] = vectors[k]
    S.init_test(params, grid, tune=False)
    S.update_EOS(params, grid)
    tm = Timer()
    params.kernel_callbacks[:] = [tm]
    dx = params.cell_size(0)
    dt = 0.2 * dx
    for axis in (Axis.X, Axis.Y):
        S.update_EOS(params, grid, axis)
        S.block_ghost_exchange(params, grid, axis)
        S.numerical_fluxes(params, grid, axis, dt, dx)
        S.cell_update(params, grid, axis, dt, dx)
        S.projection_remap(params, grid, axis, dt, dx)
    params.wait()
    res = {k: [dev.event_elapsed_ms(a, b) for a, b in v] for k, v in tm.t.items()}
    tot = sum(sum(v) for v in res.values())
    print(f"{tot:7.3f} ms total  " + "  ".join(f"{k[:14]} {'/'.join(f'{x:.2f}' for x in v)}" for k, v in res.items()) + f"  perm {perm}", flush=True)
