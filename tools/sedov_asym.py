import sys; sys.path.insert(0,'.')
import numpy as np, armon_amd
from armon_amd.solver import BlockGrid, init_test, time_loop
N=16384
for exact in (True, False):
    p = armon_amd.ArmonParameters(test="Sedov", N=(N,N), maxcycle=12, silent=5, exact_arithmetic=exact)
    g = BlockGrid(p); init_test(p,g); time_loop(p,g)
    for k,sx,sy in (("rho",1,1),("E",1,1),("p",1,1),("u",-1,1),("v",1,-1)):
        a = g.real_view(g.data[k].to_host()); s=np.abs(a).max()
        print(exact, k, "x-asym", np.abs(a - sx*a[:, ::-1]).max()/s, "y-asym", np.abs(a - sy*a[::-1,:]).max()/s, flush=True)
    del g, p
