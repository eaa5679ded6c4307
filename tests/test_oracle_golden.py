"""Pin the CPU oracle to the reference's own golden results.

Golden files: ref test/reference_data/ref_*_64bits.csv re-encoded by tests/golden/make_golden.py.
Comparison rule = the reference's own (ref test/reference_data/reference_functions.jl:54-57,
Base.isapprox): |a-b| <= max(atol, rtol*max(|a|,|b|)) with atol=1e-13, rtol=4eps — asserted by the
reference for the three Sod cases only (ref test/convergence.jl:24-27). Bizarrium and Sedov are run but
not asserted there; here they are held to cycles-exact + 1e-12 relative to the field's max magnitude.
"""
import numpy as np
import pytest

from conftest import load_golden

EPS = np.finfo(np.float64).eps
N = (100, 100)
G = 4


def isapprox_count(a, b, atol=1e-13, rtol=4 * EPS):
    return int((np.abs(a - b) > np.maximum(atol, rtol * np.maximum(np.abs(a), np.abs(b)))).sum())


@pytest.mark.parametrize("test", ["Sod", "Sod_y", "Sod_circ"])
def test_oracle_matches_reference_golden_sod_family(oracle, test):
    g = load_golden(test)
    run, f = oracle.solve(test=test, N=N, maxcycle=1000)
    assert run.cycles == int(g["cycles"])
    assert abs(run.last_dt - float(g["dt"])) <= max(1e-13, 4 * EPS * float(g["dt"]))
    for k in ("x", "y", "rho", "u", "v", "p"):
        a = oracle.real_view(f[k], *N, G)
        assert isapprox_count(a, g[k]) == 0, k


@pytest.mark.parametrize("test", ["Bizarrium", "Sedov"])
def test_oracle_matches_reference_golden_unasserted_cases(oracle, test):
    g = load_golden(test)
    run, f = oracle.solve(test=test, N=N, maxcycle=1000)
    assert run.cycles == int(g["cycles"])
    assert abs(run.last_dt - float(g["dt"])) <= 1e-12 * float(g["dt"])
    for k in ("rho", "u", "v", "p"):
        a = oracle.real_view(f[k], *N, G)
        scale = np.abs(g[k]).max()
        assert np.abs(a - g[k]).max() <= 1e-12 * max(scale, 1e-300), k
    for k in ("x", "y"):
        a = oracle.real_view(f[k], *N, G)
        assert np.abs(a - g[k]).max() <= 2 * EPS * 2.0


@pytest.mark.parametrize("test", ["Sod", "Sod_y", "Sod_circ"])
def test_oracle_conservation(oracle, test):
    """ref test/conservation.jl:1-16: mass and energy before/after, atol 1e-12."""
    run, _ = oracle.solve(test=test, N=N, maxcycle=10000)
    assert abs(run.final_mass - run.initial_mass) <= 1e-12
    assert abs(run.final_energy - run.initial_energy) <= 1e-12


@pytest.mark.parametrize("test,axis", [("Sod", 0), ("Sod_y", 1), ("Bizarrium", 0)])
def test_oracle_axis_invariance(oracle, test, axis):
    """ref test/convergence.jl:31-64: Sod invariant along y, Sod_y along x, Bizarrium along y."""
    run, f = oracle.solve(test=test, N=(40, 40), maxcycle=30)
    for k in ("rho", "u", "v", "p", "E"):
        a = oracle.real_view(f[k], 40, 40, G)
        ref = a[0:1, :] if axis == 0 else a[:, 0:1]
        assert np.array_equal(a, np.broadcast_to(ref, a.shape)), k


def test_oracle_ghost_garbage(oracle):
    """ref test/convergence.jl:67-102: garbage in every ghost cell must not reach the result."""
    nx, ny = 32, 24
    fields = oracle.alloc_fields(nx, ny, G)
    run0, f0 = oracle.solve(test="Sod_circ", N=(nx, ny), maxcycle=12)
    # initialise, then poison the ghosts of every array and run from that state
    L = oracle.lib()
    run, fields = oracle.solve(test="Sod_circ", N=(nx, ny), maxcycle=0, fields=fields)
    for k in oracle.FIELDS:
        if k in ("x", "y", "mask"):
            continue
        a = fields[k].reshape(ny + 2 * G, nx + 2 * G)
        m = np.ones_like(a, dtype=bool)
        m[G:G + ny, G:G + nx] = False
        a[m] = 1e100
    run, fields = oracle.solve(test="Sod_circ", N=(nx, ny), maxcycle=12, fields=fields, skip_init=True)
    assert run.cycles == run0.cycles and run.last_dt == run0.last_dt
    for k in ("rho", "u", "v", "E", "p"):
        assert np.array_equal(oracle.real_view(fields[k], nx, ny, G), oracle.real_view(f0[k], nx, ny, G)), k


def test_oracle_threads_do_not_change_results(oracle):
    r1, f1 = oracle.solve(test="Sod_circ", N=(64, 48), maxcycle=10, threads=1)
    r4, f4 = oracle.solve(test="Sod_circ", N=(64, 48), maxcycle=10, threads=4)
    assert r1.last_dt == r4.last_dt
    for k in ("rho", "u", "v", "E"):
        assert np.array_equal(f1[k], f4[k])


# ---- fp32 build of the oracle (ref data_type=Float32), pinned to the reference's 32-bit golden files ----------
EPS32 = np.finfo(np.float32).eps


def isapprox_count32(a, b, atol=1e-5, rtol=20 * EPS32):
    """ref test/reference_data/reference_functions.jl:55-57: atol 1e-5, rtol 20 eps(Float32)"""
    a, b = a.astype(np.float64), b.astype(np.float64)
    return int((np.abs(a - b) > np.maximum(atol, rtol * np.maximum(np.abs(a), np.abs(b)))).sum())


@pytest.mark.parametrize("test", ["Sod", "Sod_y", "Sod_circ"])
def test_oracle_f32_matches_reference_golden_sod_family(oracle, test):
    g = load_golden(test, bits=32)
    run, f = oracle.solve(test=test, N=N, maxcycle=1000, data_type=np.float32)
    assert run.cycles == int(g["cycles"])
    assert abs(run.last_dt - float(g["dt"])) <= max(1e-5, 20 * EPS32 * float(g["dt"]))
    for k in ("x", "y", "rho", "u", "v", "p"):
        assert f[k].dtype == np.float32
        assert isapprox_count32(oracle.real_view(f[k], *N, G), g[k]) == 0, k


@pytest.mark.parametrize("test", ["Bizarrium", "Sedov"])
def test_oracle_f32_unasserted_cases(oracle, test):
    """Run but not asserted by the reference (ref test/convergence.jl:24-27): cycles exact, 1e-4 of the maximum."""
    g = load_golden(test, bits=32)
    run, f = oracle.solve(test=test, N=N, maxcycle=1000, data_type=np.float32)
    assert run.cycles == int(g["cycles"])
    for k in ("rho", "u", "v", "p"):
        a = oracle.real_view(f[k], *N, G).astype(np.float64)
        assert np.abs(a - g[k]).max() <= 1e-4 * max(np.abs(g[k]).max(), 1e-30), k
