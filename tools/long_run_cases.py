#!/usr/bin/env python3
"""Whole physical runs (to the test's own maxtime) of the three BASELINE test cases in tuned arithmetic against the
same run in exact arithmetic: cycle counts, final time, maximum deviation of the fields relative to the field maximum,
conservation (Sod, Sedov), axis invariance (Sod, Bizarrium), mirror symmetry (Sedov).

    python tools/long_run_cases.py [--n 2048]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd
from armon_amd.solver import BlockGrid, conservation_vars, init_test, time_loop

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2048)
args = ap.parse_args()
n = args.n
for test in ("Sod", "Bizarrium", "Sedov"):
    fields = {}
    for exact in (True, False):
        params = armon_amd.ArmonParameters(test=test, N=(n, n), silent=5, exact_arithmetic=exact)
        grid = BlockGrid(params)
        init_test(params, grid)
        m0, e0 = conservation_vars(params, grid)
        t0 = time.time()
        t, dt, cycles, _, _ = time_loop(params, grid)
        wall = time.time() - t0
        m1, e1 = conservation_vars(params, grid)
        f = {k: grid.real_view(grid.data[k].to_host()).copy() for k in ("rho", "u", "v", "E")}
        fields[exact] = (f, cycles, t, dt)
        extra = ""
        if test in ("Sod", "Bizarrium"):
            extra = f" rows identical {all(np.array_equal(a, np.broadcast_to(a[0:1], a.shape)) for a in f.values())}"
        if test == "Sedov":
            r = f["rho"]
            extra = f" mirror asymmetry {max(np.abs(r - r[::-1]).max(), np.abs(r - r[:, ::-1]).max()) / np.abs(r).max():.2e}"
        print(f"{test:9s} {n}² {'exact' if exact else 'tuned'}: {cycles} cycles to t = {t:.6g}, last dt {dt:.6g}, {wall:.1f} s, "
              f"{n * n * 2 * cycles / wall / 1e9:.1f} Gcells/s per sweep, dM {abs(m1 - m0) / abs(m0):.2e} dE {abs(e1 - e0) / abs(e0):.2e},"
              f" finite {all(np.isfinite(a).all() for a in f.values())}{extra}", flush=True)
        del grid
    (fe, ce, te, de), (ft, ct, tt, dtt) = fields[True], fields[False]
    dev = max(np.abs(fe[k] - ft[k]).max() / max(np.abs(fe[k]).max(), 1e-300) for k in fe)
    print(f"{test:9s} tuned vs exact after the whole run: cycles {ct} vs {ce}, final time {tt - te:+.2e}, "
          f"max |Δfield| / max|field| = {dev:.2e}", flush=True)
