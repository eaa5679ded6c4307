#!/usr/bin/env python3
"""Cost of the tile structure itself on ONE GPU: the same problem (default 16384² Sod) as one block and as P tiles in one process
(armon_hip_mgpu_init, all tiles on device 0: pack / peer copy / unpack / partial sweeps, no second GPU to overlap with)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd
from armon_amd.multi_tile import TileGroup

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--global", dest="shape", default="16384x16384", help="global grid NXxNY")
ap.add_argument("--test", default="Sod")
ap.add_argument("--grids", default="1x1,2x1,1x2,2x2,4x2", help="tile grids to time, comma separated")
ap.add_argument("--cycles", type=int, default=20)
args = ap.parse_args()
nx, ny = (int(v) for v in args.shape.lower().split("x"))
cycles = args.cycles
for P in [tuple(int(v) for v in g.split("x")) for g in args.grids.split(",")]:
    g = TileGroup(P, test=args.test, N=(nx, ny), maxcycle=cycles + 2, silent=5, maxtime=1e9)
    g.init_test()
    g.global_dt.reset()
    for _ in range(2):
        g.solver_cycle(last_cycle=False)
        g.global_dt.next_cycle()
    g.wait()
    t0 = time.perf_counter()
    for _ in range(cycles):
        g.solver_cycle(last_cycle=False)
        g.global_dt.next_cycle()
    g.wait()
    ms = (time.perf_counter() - t0) / cycles * 1e3
    tile = tuple(g.params[0].N)
    print(f"{args.test} {nx}x{ny}, P = {P} (tiles of {tile[0]}x{tile[1]}): {ms:7.3f} ms per cycle  "
          f"{2 * nx * ny / ms / 1e6:7.1f} Gcells/s per sweep", flush=True)
    g.close()
    del g
    import gc
    gc.collect()
