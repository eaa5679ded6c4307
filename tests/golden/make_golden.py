"""Re-encode the reference's own golden result files as small binary fixtures.

Source (data files held by the reference's tests, MIT-licensed):
    /root/reference/test/reference_data/ref_{Sod,Sod_y,Sod_circ,Bizarrium,Sedov}_{64,32}bits.csv
Format (ref test/reference_data/reference_functions.jl:37-43, src/io.jl:4-27): line 1 = "dt, cycles";
then one line "x, y, rho, u, v, p" per real cell, rows separated by a blank line, ascending (x, y).
Parameters that produced them: ref test/reference_data/reference_functions.jl:6-18 (100x100, GAD,
minmod, euler_2nd, nghost 4, default cfl/maxtime, maxcycle 1000).

Output: tests/golden/ref_<test>_<bits>bits.npz with dt, cycles, x, y, rho, u, v, p ((ny, nx) arrays).
Run once in the build container (the reference tree does not travel to the GPU box):
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

REF = "/root/reference/test/reference_data"
HERE = os.path.dirname(os.path.abspath(__file__))


def convert(test, bits, n=(100, 100)):
    path = os.path.join(REF, f"ref_{test}_{bits}bits.csv")
    with open(path) as f:
        head = f.readline().split(",")
        dt, cycles = float(head[0]), int(head[1])
        rows = [list(map(float, line.split(","))) for line in f if line.strip()]
    a = np.asarray(rows, dtype=np.float64)
    assert a.shape == (n[0] * n[1], 6), a.shape
    a = a.reshape(n[1], n[0], 6)
    dtype = np.float64 if bits == 64 else np.float32
    out = {k: a[:, :, i].astype(dtype) for i, k in enumerate(("x", "y", "rho", "u", "v", "p"))}
    np.savez_compressed(os.path.join(HERE, f"ref_{test}_{bits}bits.npz"),
                        dt=np.float64(dt), cycles=np.int64(cycles), **out)
    return dt, cycles


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not present: fixtures are already committed, nothing to do")
    for bits in (64, 32):
        for test in ("Sod", "Sod_y", "Sod_circ", "Bizarrium", "Sedov"):
            print(test, bits, convert(test, bits))
