import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import armon_amd
from armon_amd.solver import BlockGrid, conservation_vars, init_test, time_loop
for N in (4096, 8192):
    params = armon_amd.ArmonParameters(test="Sod", N=(N, N), silent=5)
    grid = BlockGrid(params); init_test(params, grid)
    m0, e0 = conservation_vars(params, grid)
    t0 = time.time(); t, dt, cycles, cps, ns = time_loop(params, grid); el = time.time() - t0
    m1, e1 = conservation_vars(params, grid)
    rho = grid.real_view(grid.data["rho"].to_host())
    inv = np.array_equal(rho, np.broadcast_to(rho[0:1], rho.shape))
    print(N, "cycles", cycles, "time", t, "wall", round(el, 2), "s", "Gcells/s/sweep", round(N * N * 2 * cycles / el / 1e9, 1),
          "dM", abs(m1 - m0) / m0, "dE", abs(e1 - e0) / e0, "rows identical", inv, "finite", np.isfinite(rho).all(), flush=True)
