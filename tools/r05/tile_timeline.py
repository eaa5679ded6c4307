#!/usr/bin/env python3
"""Kernel timeline of one cycle of a tile with remote sides, from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/r05/tile_timeline.py run [--tile 4096x8192] [--grid 1x1]
    python3 tools/r05/tile_timeline.py show OUT/**/..._kernel_trace.csv [cycles-from-the-end=2]

run : a periodic in-process group (every tile its own or its neighbours' neighbour on all four sides), 30 cycles, native cycle driver.
show: the last cycles' kernels in start order: offset from the first one, duration, queue, short name.
"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("mode")
    ap.add_argument("--tile", default="4096x8192")
    ap.add_argument("--grid", default="1x1")
    ap.add_argument("--cycles", type=int, default=30)
    a = ap.parse_args()
    tx, ty = (int(v) for v in a.tile.split("x"))
    P = tuple(int(v) for v in a.grid.split("x"))
    from armon_amd.multi_tile import TileGroup
    g = TileGroup(P, test="Sod", N=(tx * P[0], ty * P[1]), silent=5, maxcycle=10 ** 9, maxtime=1e9, periodic=(True, True), placement_tries=0)
    g.init_test()
    g.global_dt.reset()
    for _ in range(a.cycles):
        g.solver_cycle(last_cycle=False)
        g.global_dt.next_cycle()
    g.drain()
    g.wait()
    g.close()
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    n_cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a cycle starts at each k_sweep_x launch that follows a k_sweep_y / fold (the interior of the X sweep)
    starts = [i for i, r in enumerate(rows) if "k_sweep_x_dpp" in r["Kernel_Name"] and "false, 2, false, true, 0" in r["Kernel_Name"]]
    first = starts[-n_cycles] if len(starts) >= n_cycles else 0
    t0 = int(rows[first]["Start_Timestamp"])
    print(f"# {'start us':>9s} {'dur us':>8s} {'end us':>9s}  queue  grid        kernel")
    for r in rows[first:]:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("armon::fused::", "").replace("void ", "").split("(")[0][:60]
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"  {s / 1e3:9.1f} {(e - s) / 1e3:8.1f} {e / 1e3:9.1f}  {r.get('Queue_Id', '?'):>5s}  {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>10s}  {name}")
