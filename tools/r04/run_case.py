#!/usr/bin/env python3
"""Run one case with the library named by ARMON_HIP_LIB and save the final fields: the bit-level A/B of two builds.
    ARMON_HIP_LIB=variants/a/libarmon_hip.so python tools/r04/run_case.py out_a.npz [--test Bizarrium] [--n 512] [--maxcycle 30] [--exact] [--f32]
    python tools/r04/run_case.py --compare out_a.npz out_b.npz"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="+")
ap.add_argument("--compare", action="store_true")
ap.add_argument("--test", default="Sod_circ")
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--ny", type=int, default=0)
ap.add_argument("--maxcycle", type=int, default=30)
ap.add_argument("--exact", action="store_true")
ap.add_argument("--f32", action="store_true")
ap.add_argument("--splitting", default="Sequential")
args = ap.parse_args()
if args.compare:
    a, b = np.load(args.out[0]), np.load(args.out[1])
    ok = True
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True)
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max() / max(np.abs(a[k]).max(), 1e-300)
        print(f"{k:6s} {'identical' if same else f'DIFFERENT max rel-to-max {d:.3e}'}")
        ok &= same
    sys.exit(0 if ok else 1)
import armon_amd
params = armon_amd.ArmonParameters(test=args.test, N=(args.n, args.ny or args.n), maxcycle=args.maxcycle, silent=5, return_data=True,
                                   exact_arithmetic=args.exact, axis_splitting=args.splitting,
                                   data_type="float32" if args.f32 else "float64")
stats = armon_amd.armon(params)
host = stats.data.device_to_host()
np.savez(args.out[0], dt=np.array([stats.last_dt]), cycles=np.array([stats.cycles]),
         **{k: stats.data.real_view(host[k]) for k in ("rho", "u", "v", "E", "p")})
print(args.out[0], "cycles", stats.cycles, "dt", stats.last_dt)
