import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import armon_amd
from armon_amd.multi_tile import TileGroup
n = 16384
g = TileGroup((2, 2), test="Sod", N=(n, n), maxcycle=100, silent=5, maxtime=1e9, placement_tries=0)
g.init_test(); g.global_dt.reset()
for _ in range(6):
    g.solver_cycle(last_cycle=False); g.global_dt.next_cycle()
g.wait(); g.close()
