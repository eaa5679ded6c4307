"""fp32 (ref data_type=Float32, src/parameters.jl:185) on the GPU: the `_f32` entry points against the fp32 build
of the oracle (bit-exact in exact arithmetic) and against the reference's 32-bit golden files
(ref test/reference_data/ref_*_32bits.csv; tolerance of reference_functions.jl:55-57: atol 1e-5, rtol 20 eps32)."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

EPS32 = np.finfo(np.float32).eps
G = 4


def isapprox_count32(a, b, atol=1e-5, rtol=20 * EPS32):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return int((np.abs(a - b) > np.maximum(atol, rtol * np.maximum(np.abs(a), np.abs(b)))).sum())


def run32(test, N=(100, 100), **kw):
    import armon_amd
    opts = dict(test=test, N=N, maxcycle=1000, silent=5, return_data=True, data_type=np.float32, exact_arithmetic=True)
    opts.update(kw)
    params = armon_amd.ArmonParameters(**opts)
    stats = armon_amd.armon(params)
    return params, stats, stats.data.device_to_host()


MODES = [pytest.param(dict(use_fused_sweep=False), id="staged"),
         pytest.param(dict(use_fused_sweep=True, exact_arithmetic=True), id="fused-exact"),
         pytest.param(dict(use_fused_sweep=True, exact_arithmetic=False), id="fused-tuned")]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("test", ["Sod", "Sod_y", "Sod_circ"])
def test_f32_reference_golden_sod_family(test, mode):
    g = load_golden(test, bits=32)
    params, stats, host = run32(test, **mode)
    assert stats.cycles == int(g["cycles"])
    assert abs(stats.last_dt - float(g["dt"])) <= max(1e-5, 20 * EPS32 * float(g["dt"]))
    for k in ("x", "y", "rho", "u", "v", "p"):
        assert host[k].dtype == np.float32
        assert isapprox_count32(stats.data.real_view(host[k]), g[k]) == 0, k


@pytest.mark.parametrize("mode", MODES[:2])
@pytest.mark.parametrize("test,N,opts", [
    ("Sod", (100, 100), {}),
    ("Sod_circ", (67, 41), {}),
    ("Sedov", (50, 50), dict(maxcycle=40)),
    ("Bizarrium", (64, 32), dict(maxcycle=30)),
    ("Sod_circ", (48, 48), dict(scheme="Godunov", projection="euler", nghost=2, maxcycle=25)),
    ("Sod_circ", (48, 40), dict(axis_splitting="Strang", riemann_limiter="superbee", maxcycle=15)),
    ("Sod_circ", (517, 263), dict(axis_splitting="Godunov", nghost=5, maxcycle=8)),     # many workgroups, X-last cycles
])
def test_f32_bit_exact_against_f32_oracle(oracle, test, N, opts, mode):
    params, stats, host = run32(test, N=N, **mode, **opts)
    orun, f = oracle.solve(test=test, N=N, data_type=np.float32, **{"maxcycle": 1000, **opts})
    assert stats.cycles == orun.cycles
    assert np.float32(stats.last_dt) == np.float32(orun.last_dt) and np.float32(stats.final_time) == np.float32(orun.final_time)
    g = opts.get("nghost", 4)
    for k in ("rho", "u", "v", "E", "p"):
        a = stats.data.real_view(host[k])
        b = oracle.real_view(f[k], N[0], N[1], g)
        assert np.array_equal(a, b), f"{k}: max abs diff {np.abs(a - b).max()}"


def test_f32_tuned_close_to_oracle(oracle):
    N = (67, 41)
    params, stats, host = run32("Sod_circ", N=N, use_fused_sweep=True, exact_arithmetic=False, maxcycle=40)
    orun, f = oracle.solve(test="Sod_circ", N=N, data_type=np.float32, maxcycle=40)
    assert stats.cycles == orun.cycles
    for k in ("rho", "u", "v", "E", "p"):
        a = stats.data.real_view(host[k]).astype(np.float64)
        b = oracle.real_view(f[k], N[0], N[1], G).astype(np.float64)
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max(), k


def test_f32_staged_kernels_against_oracle(oracle):
    """A few `_f32` kernels directly through the C ABI (the full list is exercised by the solver tests)."""
    import armon_amd
    from armon_amd._lib import Range
    from armon_amd.device import HIPDevice
    L = armon_amd.lib()
    dev = HIPDevice(0)
    nx, ny = 37, 29
    n = (nx + 2 * G) * (ny + 2 * G)
    rng = np.random.default_rng(5)
    f = {k: rng.uniform(lo, hi, n).astype(np.float32) for k, (lo, hi) in dict(
        rho=(0.1, 2), u=(-1, 1), v=(-1, 1), E=(2, 4), p=(0.1, 2), c=(0.5, 2), g=(1, 2), us=(-1, 1), ps=(0.1, 2)).items()}
    d = {k: dev.from_host(a) for k, a in f.items()}
    P = lambda k: C.c_void_p(d[k].ptr)
    OL = oracle.lib(f32=True)
    r = oracle.domain_range(nx, ny, G)
    cr = Range(r.col_start, r.col_step, r.col_len, r.row_start, r.row_len)
    OL.armon_oracle_perfect_gas_EOS(r, 1.4, *(oracle.ptr(f[k]) for k in ("rho", "E", "u", "v", "p", "c", "g")))
    assert L.armon_hip_perfect_gas_EOS_f32(dev.ctx, cr, 1.4, *(P(k) for k in ("rho", "E", "u", "v", "p", "c", "g"))) == 0
    for k in ("p", "c", "g"):
        assert np.array_equal(d[k].to_host(), f[k]), k
    fl = oracle.domain_range(nx, ny, G, (0, -2), (0, 3))
    cfl = Range(fl.col_start, fl.col_step, fl.col_len, fl.row_start, fl.row_len)
    s = nx + 2 * G
    OL.armon_oracle_acoustic_GAD(fl, s, 1e-3, 1.0 / nx, *(oracle.ptr(f[k]) for k in ("us", "ps", "rho", "v", "p", "c")), 1)
    assert L.armon_hip_acoustic_GAD_f32(dev.ctx, cfl, s, 1e-3, 1.0 / nx, *(P(k) for k in ("us", "ps", "rho", "v", "p", "c")), 1) == 0
    for k in ("us", "ps"):
        assert np.array_equal(d[k].to_host(), f[k]), k
    ref = OL.armon_oracle_dtCFL(r, 0.01, 0.02, *(oracle.ptr(f[k]) for k in ("u", "v", "c")))
    out = C.c_float()
    assert L.armon_hip_dtCFL_f32(dev.ctx, cr, 0.01, 0.02, *(P(k) for k in ("u", "v", "c")), C.byref(out)) == 0
    assert out.value == ref
    dev.close()


@pytest.mark.parametrize("test,N", [("Sod_circ", (334, 77)), ("Sedov", (120, 200)), ("Bizarrium", (256, 64))])
def test_f32_two_column_y_march_equals_the_one_column_form(monkeypatch, test, N):
    """The tuned fp32 Y sweep runs two columns per lane (8-B accesses) when rows are 8-B aligned; it must give the
    bits of the one-column kernel (forced with ARMON_Y_COLS1, and taken anyway for odd sizes)."""
    opts = dict(N=N, maxcycle=9, use_fused_sweep=True, exact_arithmetic=False)
    _p, s2, h2 = run32(test, **opts)
    monkeypatch.setenv("ARMON_Y_COLS1", "1")
    _p, s1, h1 = run32(test, **opts)
    assert s1.cycles == s2.cycles and s1.last_dt == s2.last_dt
    for k in ("rho", "u", "v", "E", "p"):
        assert np.array_equal(h1[k], h2[k]), k


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "tuned"])
@pytest.mark.parametrize("test,N,opts", [("Sod_circ", (517, 77), {}), ("Sod_circ", (130, 40), dict(axis_splitting="Godunov")),
                                         ("Bizarrium", (1000, 9), dict(axis_splitting="X_only"))])
def test_f32_x_sweep_workgroup_shapes_give_the_same_bits(monkeypatch, test, N, opts, exact):
    """fp32 X sweeps put 4 consecutive strips of ONE row in a workgroup (fp64: one strip of 4 rows): who computes a cell
    must not change it — also on the cycles that end with a dt-tracking X sweep."""
    o = dict(N=N, maxcycle=9, use_fused_sweep=True, exact_arithmetic=exact, **opts)
    _p, s0, h0 = run32(test, **o)
    for shape in ("1", "2"):
        monkeypatch.setenv("ARMON_X_ROWS", shape)
        _p, s1, h1 = run32(test, **o)
        assert s1.cycles == s0.cycles and s1.last_dt == s0.last_dt
        for k in ("rho", "u", "v", "E", "p"):       # real cells: the ghost rows of a ping-pong partner are never written
            assert np.array_equal(s1.data.real_view(h1[k]), s0.data.real_view(h0[k])), (shape, k)


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "tuned"])
@pytest.mark.parametrize("test,N,opts", [("Sod_circ", (520, 300), {}),                  # pitch 528 floats: rows on sectors
                                         ("Sod_circ", (512, 300), {}),                  # 520: every other row mid-sector (two columns per lane)
                                         ("Sod_circ", (1030, 130), dict(nghost=6)),     # 1042: 8-B steps, several workgroups per row
                                         ("Sedov", (515, 263), dict(nghost=5)),         # odd pitch: one column per lane
                                         ("Bizarrium", (40, 600), dict(axis_splitting="Y_only"))])
def test_f32_y_march_store_exchange_gives_the_same_bits(monkeypatch, test, N, opts, exact):
    """fp32 Y marches (one and two columns per lane) with the LDS store exchange forced on / off / automatic."""
    o = dict(N=N, maxcycle=7, use_fused_sweep=True, exact_arithmetic=exact, **opts)
    monkeypatch.setenv("ARMON_Y_SX", "2")
    _p, s0, h0 = run32(test, **o)
    for mode in ("1", "0"):
        monkeypatch.setenv("ARMON_Y_SX", mode)
        _p, s1, h1 = run32(test, **o)
        assert s1.cycles == s0.cycles and s1.last_dt == s0.last_dt
        for k in ("rho", "u", "v", "E", "p"):
            assert np.array_equal(s1.data.real_view(h1[k]), s0.data.real_view(h0[k])), (mode, k)
