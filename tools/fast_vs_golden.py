import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import armon_amd
EPS = np.finfo(float).eps
for test in ("Sod", "Sod_y", "Sod_circ", "Bizarrium", "Sedov"):
    g = np.load(f"tests/golden/ref_{test}_64bits.npz")
    for exact in (True, False):
        params = armon_amd.ArmonParameters(test=test, N=(100, 100), maxcycle=1000, silent=5, return_data=True, exact_arithmetic=exact)
        st = armon_amd.armon(params)
        h = st.data.device_to_host()
        out = []
        for k in ("rho", "u", "v", "p"):
            a = st.data.real_view(h[k]); b = g[k]
            bad = int((np.abs(a - b) > np.maximum(1e-13, 4 * EPS * np.maximum(np.abs(a), np.abs(b)))).sum())
            out.append(f"{k}: bad={bad} max={np.abs(a-b).max()/max(np.abs(b).max(),1e-300):.2e}")
        print(test, "exact" if exact else "fast ", st.cycles, int(g["cycles"]), f"dt rel {abs(st.last_dt-float(g['dt']))/float(g['dt']):.2e}", " | ".join(out))
