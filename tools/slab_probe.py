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

# ---- windows along ONE large slab: does the level depend on WHERE in the allocation the 8 vectors lie? -----------------
if os.environ.get("SLAB_WINDOWS"):
    K = int(os.environ["SLAB_WINDOWS"])              # vectors' worth of slab, e.g. 96 → 200 GB
    slab = DeviceArray(dev, K * S, "uint8")
    args.tries = 6
    for w in range(0, K - 8 + 1, 4):
        ptrs = [slab.ptr + (w + k) * S + offsets("rand", k) for k in range(8)]
        assert ptrs[-1] + nbytes <= slab.ptr + K * S, "vectors must lie inside the slab"
        t, picks = run(ptrs)
        print(f"window at {w * S / 2**30:6.1f} GiB: " + " ".join(f"{v:.2f}" for v in t), flush=True)
    slab.free()

# ---- uniform stride inside one slab: X + Y sweep time against the stride's residue modulo 16 MiB, at several bases -----
if os.environ.get("SLAB_STRIDES"):
    GiB = 1 << 30
    bases = (0, 1 * GiB, 5 * GiB, 22 * GiB)
    slab_bytes = 8 * (2064 + 48) * MiB + bases[-1] + GiB + 64 * MiB
    slab = DeviceArray(dev, slab_bytes, "uint8")
    origin = -slab.ptr % GiB                                  # 1-GiB aligned virtual address
    args.tries = 1                                            # the identity assignment only: roles in slab order
    print(f"slab VA {slab.ptr:#x}; columns: base +0, +1 GiB, +5 GiB, +22 GiB; X+Y ms of the identity assignment")
    for r in list(range(0, 32)) + [46, 47, 48]:
        stride = (2064 + r) * MiB
        row = []
        for b in bases:
            ptrs = [slab.ptr + origin + b + k * stride for k in range(8)]
            assert ptrs[-1] + nbytes <= slab.ptr + slab_bytes and stride >= nbytes, "vectors must lie inside the slab"
            t, _ = run(ptrs)
            row.append(t[0])
        print(f"stride 2064 + {r:2d} MiB (≡ {r % 16:2d} mod 16): " + " ".join(f"{v:.2f}" for v in row), flush=True)
    slab.free()

# ---- wide stride scan with the real sweeps: 2 GiB ... 4 GiB in 48-MiB steps, roles in slab order and interleaved --------
if os.environ.get("SLAB_WIDE"):
    GiB = 1 << 30
    bases = (0, 5 * GiB)
    t_max = 4224
    slab_bytes = 8 * t_max * MiB + bases[-1] + GiB + 64 * MiB
    slab = DeviceArray(dev, slab_bytes, "uint8")
    origin = -slab.ptr % GiB
    args.tries = 1
    print(f"slab VA {slab.ptr:#x}; X+Y ms: [roles in slab order @base, @base+5GiB] [reads on even, writes on odd @base, @base+5GiB]")
    for t_mib in range(2064, t_max + 1, 48):
        stride = t_mib * MiB
        row = []
        for order in ((0, 1, 2, 3, 4, 5, 6, 7), (0, 2, 4, 6, 1, 3, 5, 7)):
            for b in bases:
                ptrs = [slab.ptr + origin + b + k * stride for k in order]
                assert max(ptrs) + nbytes <= slab.ptr + slab_bytes and stride >= nbytes, "vectors must lie inside the slab"
                t, _ = run(ptrs)
                row.append(t[0])
        print(f"stride {t_mib:5d} MiB (mod 2 GiB = {t_mib % 2048:4d}): " + " ".join(f"{v:.2f}" for v in row), flush=True)
    slab.free()

# ---- 16 vectors at a uniform stride in one slab, the same 24 role assignments for every residue of the stride -----------
if os.environ.get("SLAB_RESIDUES"):
    args.tries = 24
    for t_mib in (2054, 2066, 2062, 2058, 2052, 2060, 2064, 2056):
        stride = t_mib * MiB
        slab_bytes = 16 * stride + 64 * MiB
        slab = DeviceArray(dev, slab_bytes, "uint8")
        ptrs = [slab.ptr + k * stride for k in range(16)]
        assert max(ptrs) + nbytes <= slab.ptr + slab_bytes and stride >= nbytes, "vectors must lie inside the slab"
        t, picks = run(ptrs)
        fast = sum(v <= 5.85 for v in t)
        print(f"stride {t_mib} MiB (≡ {t_mib % 16:2d} mod 16) fast {fast:2d}/24: " + " ".join(f"{v:.2f}" for v in t) + f"  best {picks}", flush=True)
        slab.free()
