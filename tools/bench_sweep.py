#!/usr/bin/env python3
"""Time armon_hip_sweep per axis / arithmetic on one GPU with HIP events (interleaved rounds, one process).

    python tools/bench_sweep.py [--n 16384] [--rounds 5] [--test Sod] [--scheme GAD] [--emit]
"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd
from armon_amd.blocking import Axis
from armon_amd._lib import check
from armon_amd.solver import BlockGrid, fused_sweep, init_test

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--ny", type=int, default=0)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--test", default="Sod")
ap.add_argument("--scheme", default="GAD")
ap.add_argument("--projection", default="euler_2nd")
ap.add_argument("--emit", action="store_true")
ap.add_argument("--track", action="store_true", help="fused dt/CFL reduction on the Y sweep")
ap.add_argument("--modes", default="exact,fast")
ap.add_argument("--niter", default="", help="comma list of ARMON_XS_NITER values to interleave (X sweep)")
ap.add_argument("--seg", default="", help="comma list of ARMON_Y_SEG values to interleave (Y sweep)")
ap.add_argument("--align", default="", help="comma list of ARMON_SWEEP_ALIGN values to interleave (0 = unaligned origins)")
ap.add_argument("--xk", default="0", help="comma list of X kernel forms: 0 spatial K=2, 3 spatial K=1, 1 LDS vec, 2 LDS generic")
args = ap.parse_args()
ny = args.ny or args.n
if args.xk != "0":       # forms 2 and 3 are compiled into the A/B build only (libarmon_hip_alt.so, -DARMON_ALT_KERNELS)
    from armon_amd import _lib
    _lib.alt_kernels().__enter__()

params = armon_amd.ArmonParameters(test=args.test, N=(args.n, ny), scheme=args.scheme, projection=args.projection,
                                   silent=5, maxcycle=10)
grid = BlockGrid(params)
init_test(params, grid)
dev = params.device
dx = params.domain_size[0] / args.n
dt = 0.3 * dx          # a plausible CFL-limited step for Sod (c ~ 1.2)
cells = args.n * ny
res = {}
for r in range(args.rounds + 1):
    for mode in args.modes.split(","):
        params.exact_arithmetic = mode == "exact"
        als = args.align.split(",") if args.align else [""]
        cfgs = [(Axis.X, int(k), nit, "", al) for k in args.xk.split(",") for nit in (args.niter.split(",") if args.niter else [""]) for al in als]
        cfgs += [(Axis.Y, 0, "", sg, al) for sg in (args.seg.split(",") if args.seg else [""]) for al in als]
        for axis, xk, nit, sg, al in cfgs:
            params.x_kernel = xk
            for key, val in (("ARMON_XS_NITER", nit), ("ARMON_Y_SEG", sg), ("ARMON_SWEEP_ALIGN", al)):
                check(armon_amd.lib().armon_hip_set_tuning(dev.ctx, key.encode(), int(val) if val else (-1 if key == "ARMON_SWEEP_ALIGN" else 0)))
            dev.event_record(0)
            fused_sweep(params, grid, axis, dt, dx, emit_p=args.emit, emit_c=args.emit, emit_dt=args.track and axis == Axis.Y)
            dev.event_record(1)
            ms = dev.event_elapsed_ms(0, 1)
            if r > 0:
                tag = axis.name + (str(xk) if axis == Axis.X else "") + (f" niter={nit}" if nit else "") + (f" seg={sg}" if sg else "") + (f" align={al}" if al else "")
                res.setdefault((mode, tag), []).append(ms)
    # keep the state sane (a few sweeps of Sod are harmless, but do not let it drift for long)
    if r % 3 == 2:
        init_test(params, grid)
for (mode, axis), v in res.items():
    med = statistics.median(v)
    print(f"{mode:5s} sweep_{axis:22s}: median {med:7.3f} ms  min {min(v):7.3f} ms   "
          f"{64 * cells / med / 1e6:7.1f} GB/s algorithmic   {cells / med / 1e3:8.1f} Mcells/s")
