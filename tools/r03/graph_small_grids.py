#!/usr/bin/env python3
"""Whole runs on small and medium grids, host-driven time loop against graph replay (device-resident time step):
wall time per cycle. Same bits either way (tests/test_gpu_graph.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import armon_amd

for n, cycles in ((128, 4000), (256, 4000), (512, 3000), (1024, 2000), (2048, 1000), (4096, 400), (8192, 100)):
    row = []
    for graph in (False, True):
        params = armon_amd.ArmonParameters(test="Sod_circ", N=(n, n), maxcycle=cycles, maxtime=1e9, silent=5, graph_cycles=graph,
                                           placement_tries=0)
        stats = armon_amd.armon(params)
        row.append(stats.solve_time / stats.cycles * 1e6)
    print(f"{n:5d}²: host-driven {row[0]:8.1f} us per cycle   graph replay {row[1]:8.1f} us per cycle   x{row[0] / row[1]:.2f}"
          f"   ({2 * n * n / row[1] / 1e3:7.1f} Gcells/s per sweep with graphs)", flush=True)
