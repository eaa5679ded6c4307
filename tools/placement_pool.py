#!/usr/bin/env python3
"""Diagnostic for BlockGrid.tune_placement: time many random assignments of the 8 sweep vectors within a pool.

    python tools/placement_pool.py [--n 16384] [--pool 32] [--tries 24]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd
from armon_amd.solver import BlockGrid, init_test

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--pool", type=int, default=32)
ap.add_argument("--tries", type=int, default=24)
args = ap.parse_args()
params = armon_amd.ArmonParameters(test="Sod", N=(args.n, args.n), silent=5, maxcycle=10, placement_tries=0)
grid = BlockGrid(params)
init_test(params, grid)
params.placement_tries = args.tries
rep = grid.tune_placement(spare=args.pool - 8)
ts = sorted(rep["x_plus_y_ms"])
print(f"pool {rep['pool']} tries {rep['tries']}: first (back-to-back) {rep['x_plus_y_ms'][0]:.3f}  best {ts[0]:.3f}  median {ts[len(ts) // 2]:.3f}  worst {ts[-1]:.3f} ms")
print("sorted:", " ".join(f"{t:.2f}" for t in ts))
