#!/usr/bin/env python3
"""Cost of the tile structure itself on ONE GPU: the same 16384² problem as one block and as P tiles in one process
(armon_hip_mgpu_init, all tiles on device 0: pack / peer copy / unpack / partial sweeps, no second GPU to overlap with)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd
from armon_amd.multi_tile import TileGroup

n, cycles = 16384, 20
for P in ((1, 1), (2, 1), (1, 2), (2, 2), (4, 2)):
    g = TileGroup(P, test="Sod", N=(n, n), maxcycle=cycles + 2, silent=5, maxtime=1e9)
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
    print(f"P = {P}: {ms:7.3f} ms per cycle  {2 * n * n / ms / 1e6:7.1f} Gcells/s per sweep", flush=True)
    g.close()
