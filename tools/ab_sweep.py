#!/usr/bin/env python3
"""A/B timing of armon_hip_sweep across several BUILDS of the library in one process, finely interleaved
(round-robin per launch), so that device-level drift (clocks, temperature, HBM refresh) hits every build
alike. The state is initialised once and never swapped: every launch does the same work.

    python tools/ab_sweep.py [--n 16384] [--rounds 20] [--exact] base=armon.jl_amd/libarmon_hip.so nt2=variants/nt2/libarmon_hip.so ...
"""
import argparse
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd
from armon_amd import _lib
from armon_amd.blocking import Axis
from armon_amd.solver import BlockGrid, init_test, sweep_desc

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--shape", default=None, metavar="NXxNY", help="non-square block (overrides --n)")
ap.add_argument("--rounds", type=int, default=20)
ap.add_argument("--test", default="Sod")
ap.add_argument("--exact", action="store_true")
ap.add_argument("--scheme", default="GAD")
ap.add_argument("--projection", default="euler_2nd")
ap.add_argument("--f32", action="store_true")
ap.add_argument("--track-x", action="store_true", help="fused dt/CFL reduction on the X sweep too (X-last splittings)")
ap.add_argument("--no-track-y", action="store_true", help="Y sweep without the fused dt/CFL reduction (a cycle's first sweep in Y-first splittings)")
ap.add_argument("--env", default="", help="per-build tuning knobs: name:KEY=VAL,KEY=VAL;name2:... (armon_hip_set_tuning on that build's context)")
ap.add_argument("--gap-ms", type=float, default=0., help="idle time before every launch (clock/power recovery experiments)")
ap.add_argument("--copy", action="store_true", help="also time armon_hip_stream_copy4 on the same arrays (same bytes, no arithmetic)")
ap.add_argument("libs", nargs="+", help="name=path")
args = ap.parse_args()

shape = tuple(int(v) for v in args.shape.lower().split("x")) if args.shape else (args.n, args.n)
params = armon_amd.ArmonParameters(test=args.test, N=shape, silent=5, maxcycle=10, exact_arithmetic=args.exact,
                                   scheme=args.scheme, projection=args.projection, data_type="float32" if args.f32 else "float64")
grid = BlockGrid(params)
init_test(params, grid)
params.wait()
dx = params.domain_size[0] / shape[0]
dt = 0.3 * dx
envs = {}
for part in filter(None, args.env.split(";")):
    name, kv = part.split(":", 1)
    envs[name] = dict(x.split("=", 1) for x in kv.split(","))
builds = []
for spec in args.libs:
    name, path = spec.split("=", 1)
    L = _lib.load_at(os.path.abspath(path))
    ctx = C.c_void_p()
    _lib.check(L.armon_hip_init(0, None, C.byref(ctx)))
    builds.append((name, L, ctx))
import time
from armon_amd.solver import STATE_VARS
src = (C.c_void_p * 4)(*[grid.data[f].ptr for f in STATE_VARS])
dst = (C.c_void_p * 4)(*[grid.alt[f].ptr for f in STATE_VARS])
nbytes = grid.data["rho"].nbytes & ~15
res = {}
knobs = sorted({k for e in envs.values() for k in e})
for r in range(args.rounds + 2):
    for axis in (Axis.X, Axis.Y):
        d = sweep_desc(params, grid, axis, dt, dx, emit_dt=(axis == Axis.Y and not args.no_track_y) or (axis == Axis.X and args.track_x))
        for name, L, ctx in builds:
            for k in knobs:                       # knobs live in the context: reset, then apply this build's values
                _lib.check(L.armon_hip_set_tuning(ctx, k.encode(), int(envs.get(name, {}).get(k, -1 if k == "ARMON_SWEEP_ALIGN" else 0))))
            if args.gap_ms:
                time.sleep(args.gap_ms * 1e-3)
            _lib.check(L.armon_hip_event_record(ctx, 0))
            _lib.check(getattr(L, "armon_hip_sweep" + params.suffix)(ctx, C.byref(d)))
            _lib.check(L.armon_hip_event_record(ctx, 1))
            ms = C.c_double()
            _lib.check(L.armon_hip_event_elapsed_ms(ctx, 0, 1, C.byref(ms)))
            if r >= 2:
                res.setdefault((axis.name, name), []).append(ms.value)
        if args.copy:
            name, L, ctx = builds[0]
            if args.gap_ms:
                time.sleep(args.gap_ms * 1e-3)
            _lib.check(L.armon_hip_event_record(ctx, 0))
            _lib.check(L.armon_hip_stream_copy4(ctx, C.byref(src), C.byref(dst), nbytes))
            _lib.check(L.armon_hip_event_record(ctx, 1))
            ms = C.c_double()
            _lib.check(L.armon_hip_event_elapsed_ms(ctx, 0, 1, C.byref(ms)))
            if r >= 2:
                res.setdefault((axis.name, "copy4"), []).append(ms.value)
base = {}
for (axis, name), v in res.items():
    med = statistics.median(v)
    base.setdefault(axis, med)
    print(f"sweep_{axis} {name:12s}: median {med:7.3f} ms  min {min(v):7.3f}  max {max(v):7.3f}   "
          f"{(32 if args.f32 else 64) * shape[0] * shape[1] / med / 1e6:7.1f} GB/s   x{med / base[axis]:.3f} vs first")
