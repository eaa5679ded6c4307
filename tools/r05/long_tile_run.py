#!/usr/bin/env python3
"""One-off soak of the one-call cycle with a host thread per tile: thousands of cycles (barriers, prefetched exchanges, the dt
chain on the transfer streams) on a 4 x 2 and a 3 x 3 group, with and without injected delays, against the single block: bits.
    timeout 600 python tools/r05/long_tile_run.py [cycles=3000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import armon_amd
from armon_amd.multi_tile import TileGroup

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for P, N, test, extra in (((4, 2), (512, 256), "Sod_circ", {}), ((3, 3), (300, 330), "Sedov", dict(axis_splitting="Strang")),
                          ((4, 2), (512, 256), "Bizarrium", dict(axis_splitting="Godunov", maxtime=0.))):     # its own end time: 80 us
    kw = dict(dict(test=test, N=N, maxcycle=cycles, maxtime=1e9, silent=5), **extra)
    ref = armon_amd.armon(armon_amd.ArmonParameters(return_data=True, **kw))
    host = ref.data.device_to_host(("rho", "u", "v", "E", "p"))
    for chaos in (0, 40):
        g = TileGroup(P, **kw)
        try:
            g.set_threads(True)
            if chaos:
                g.set_chaos(chaos, 7)
            t0 = time.time()
            stats = g.run()
            got = g.gather()
            same = (stats.cycles == ref.cycles and stats.last_dt == ref.last_dt and stats.final_time == ref.final_time
                    and all(np.array_equal(got[k], ref.data.real_view(host[k])) for k in got))
            print(f"{test:10s} {N[0]}x{N[1]} on {P[0]}x{P[1]} tiles, {stats.cycles} cycles, chaos {chaos:3d} us: "
                  f"{'identical to the single block' if same else 'DIFFERENT'}   ({time.time() - t0:.1f} s)", flush=True)
            assert same
        finally:
            g.close()
