"""Parity at BASELINE.json's full size (Sod 16384², the bench workload) through size-independent properties:

* axis invariance (ref test/convergence.jl:31-64): Sod depends on x only, so every one of the 16384 rows must hold
  the same bits — both sweeps, every workgroup, every strip/run boundary and the ghost handling are exercised;
* that common row must equal the CPU oracle run on a 16384×8 strip of the same cell size (bit for bit in exact
  arithmetic, staged and fused; within the tuned tolerance otherwise), together with dt, time and cycle count;
* conservation of mass and energy (ref test/conservation.jl) to 1e-11 relative over the run.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, G, CYCLES = 16384, 4, 5
NAMES = ("rho", "u", "v", "E", "p")


@pytest.fixture(scope="module")
def oracle_strip(oracle):
    """The oracle on 16384×8 cells of the same size (domain 1 × 8/16384): its rows are the full problem's rows."""
    run, f = oracle.solve(test="Sod", N=(N, 8), domain_size=(1., 8. / N), maxcycle=CYCLES, threads=8)
    rows = {k: oracle.real_view(f[k], N, 8, G) for k in NAMES}
    for k in NAMES:
        assert np.array_equal(rows[k], np.broadcast_to(rows[k][0:1], rows[k].shape)), k
    return run, {k: rows[k][0].copy() for k in NAMES}


@pytest.mark.parametrize("mode", ["fused-exact", "staged", "fused-tuned"])
def test_sod_16384_rows_identical_and_equal_to_the_oracle(oracle_strip, mode):
    import armon_amd
    from armon_amd.solver import BlockGrid, conservation_vars, init_test, time_loop
    orun, orow = oracle_strip
    params = armon_amd.ArmonParameters(test="Sod", N=(N, N), maxcycle=CYCLES, silent=5,
                                       use_fused_sweep=mode != "staged", exact_arithmetic=mode != "fused-tuned")
    grid = BlockGrid(params)
    init_test(params, grid)
    if mode != "staged":
        assert grid.placement and grid.placement["tries"] >= 2          # the placement tuning ran at this size
    m0, e0 = conservation_vars(params, grid)
    time_, dt, cycles, _, _ = time_loop(params, grid)
    m1, e1 = conservation_vars(params, grid)
    assert abs(m1 - m0) <= 1e-11 * abs(m0) and abs(e1 - e0) <= 1e-11 * abs(e0)
    assert cycles == orun.cycles == CYCLES
    exact = mode != "fused-tuned"
    if exact:
        assert dt == orun.last_dt and time_ == orun.final_time
    else:
        assert abs(dt - orun.last_dt) <= 1e-12 * orun.last_dt
    for k in NAMES:
        a = grid.real_view(grid.data[k].to_host())
        assert np.array_equal(a, np.broadcast_to(a[0:1], a.shape)), f"{k}: rows differ"
        if exact:
            assert np.array_equal(a[0], orow[k]), k
        else:
            assert np.abs(a[0] - orow[k]).max() <= 1e-11 * np.abs(orow[k]).max(), k
        del a


@pytest.mark.parametrize("test", ["Sod_y", "Bizarrium"])
def test_other_cases_16384_fused_exact(oracle, test):
    """Sod_y varies along y only (columns identical, oracle strip 8×16384); Bizarrium along x with its own EOS."""
    import armon_amd
    along_x = test != "Sod_y"
    ds = {"Sod_y": (8. / N, 1.), "Bizarrium": (1., 8. / N)}[test]
    n_strip = (N, 8) if along_x else (8, N)
    orun, f = oracle.solve(test=test, N=n_strip, domain_size=ds, maxcycle=CYCLES, threads=8)
    params = armon_amd.ArmonParameters(test=test, N=(N, N), maxcycle=CYCLES, silent=5, exact_arithmetic=True,
                                       return_data=True)
    stats = armon_amd.armon(params)
    assert stats.cycles == orun.cycles and stats.last_dt == orun.last_dt and stats.final_time == orun.final_time
    for k in NAMES:
        a = stats.data.real_view(stats.data.data[k].to_host())
        o = oracle.real_view(f[k], n_strip[0], n_strip[1], G)
        line = a[0:1] if along_x else a[:, 0:1]
        assert np.array_equal(a, np.broadcast_to(line, a.shape)), f"{k}: lines differ"
        assert np.array_equal(line.ravel(), (o[0] if along_x else o[:, 0])), k
        del a


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("test,shape", [("Sod", (16388, 1030)), ("Sod_y", (4099, 8200)), ("Sod", (8187, 2050)), ("Sod_y", (16384, 2048))])
def test_irregular_row_pitch_keeps_the_axis_invariance(test, shape, dtype):
    """Row pitches that are not a multiple of a 64-B sector (16396, 4107, 8195 cells; 16392 floats for the last shape in fp32)
    take the X strip origins row by row and pass the Y march's rows through LDS (column (t - r) mod 512 per thread): Sod must
    still give the same bits in every ROW although each row has its own origin, Sod_y in every COLUMN although the march
    rotates them — at sizes with several strips, workgroups and runs, tuned arithmetic (the form that carries the exchange)."""
    import armon_amd
    params = armon_amd.ArmonParameters(test=test, N=shape, maxcycle=CYCLES, silent=5, exact_arithmetic=False, return_data=True,
                                       data_type=dtype)
    stats = armon_amd.armon(params)
    assert stats.cycles == CYCLES and np.isfinite(stats.last_dt)
    for k in NAMES:
        a = stats.data.real_view(stats.data.data[k].to_host())
        line = a[0:1] if test == "Sod" else a[:, 0:1]
        assert np.array_equal(a, np.broadcast_to(line, a.shape)), f"{k}: lines differ"
        assert np.isfinite(a).all()
        del a


def test_sod_4096_full_run_to_maxtime():
    """A whole physical run (t = 0.2, ≈1900 cycles, tuned arithmetic, deferred dt read-back all the way): mass and
    energy conserved to 1e-12 relative (ref test/conservation.jl's bound is 1e-12 absolute at 100²), rows identical."""
    import armon_amd
    from armon_amd.solver import BlockGrid, conservation_vars, init_test, time_loop
    n = 4096
    params = armon_amd.ArmonParameters(test="Sod", N=(n, n), silent=5)
    grid = BlockGrid(params)
    init_test(params, grid)
    m0, e0 = conservation_vars(params, grid)
    time_, dt, cycles, _, _ = time_loop(params, grid)
    m1, e1 = conservation_vars(params, grid)
    assert time_ >= 0.2 and 1800 < cycles < 2000 and 0 < dt < 1e-3
    assert abs(m1 - m0) <= 1e-12 * m0 and abs(e1 - e0) <= 1e-12 * e0
    for k in ("rho", "u", "E"):
        a = grid.real_view(grid.data[k].to_host())
        assert np.isfinite(a).all() and np.array_equal(a, np.broadcast_to(a[0:1], a.shape)), k
    assert not grid.real_view(grid.data["v"].to_host()).any()


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "tuned"])
def test_sod_16384_f32(oracle, exact):
    """Float32 at full size (the tuned flavour runs the two-column Y march): rows identical; equal to the fp32 oracle
    strip bit for bit in exact arithmetic, within 2e-5 of the field maximum (≈170 eps32) otherwise."""
    import armon_amd
    orun, f = oracle.solve(test="Sod", N=(N, 8), domain_size=(1., 8. / N), maxcycle=CYCLES, threads=8, data_type=np.float32)
    params = armon_amd.ArmonParameters(test="Sod", N=(N, N), maxcycle=CYCLES, silent=5, data_type="float32",
                                       exact_arithmetic=exact, return_data=True)
    stats = armon_amd.armon(params)
    assert stats.cycles == orun.cycles
    if exact:
        assert np.float32(stats.last_dt) == np.float32(orun.last_dt)
    for k in NAMES:
        a = stats.data.real_view(stats.data.data[k].to_host())
        assert a.dtype == np.float32
        assert np.array_equal(a, np.broadcast_to(a[0:1], a.shape)), f"{k}: rows differ"
        o = oracle.real_view(f[k], N, 8, G)[0]
        if exact:
            assert np.array_equal(a[0], o), k
        else:
            assert np.abs(a[0].astype(np.float64) - o).max() <= 2e-5 * np.abs(o).max(), k
        del a


# ---- BASELINE.json configs 2, 3 and 5's test case at their full single-GPU sizes -----------------------------------
@pytest.mark.parametrize("exact", [True, False], ids=["exact", "tuned"])
def test_config2_sod_8192_godunov_fused(oracle, exact):
    """BASELINE configs[1]: Sod 8192² with the first-order acoustic solver (scheme=Godunov, projection euler_2nd),
    fused sweep: rows identical, equal to the oracle's 8192×8 strip (bit for bit / tuned tolerance), conservation."""
    import armon_amd
    from armon_amd.solver import BlockGrid, conservation_vars, init_test, time_loop
    n = 8192
    orun, f = oracle.solve(test="Sod", N=(n, 8), domain_size=(1., 8. / n), maxcycle=CYCLES, threads=8, scheme="Godunov")
    params = armon_amd.ArmonParameters(test="Sod", N=(n, n), maxcycle=CYCLES, silent=5, scheme="Godunov",
                                       exact_arithmetic=exact)
    grid = BlockGrid(params)
    init_test(params, grid)
    m0, e0 = conservation_vars(params, grid)
    time_, dt, cycles, _, _ = time_loop(params, grid)
    m1, e1 = conservation_vars(params, grid)
    assert abs(m1 - m0) <= 1e-11 * abs(m0) and abs(e1 - e0) <= 1e-11 * abs(e0)
    assert cycles == orun.cycles == CYCLES
    if exact:
        assert dt == orun.last_dt and time_ == orun.final_time
    else:
        assert abs(dt - orun.last_dt) <= 1e-12 * orun.last_dt
    for k in NAMES:
        a = grid.real_view(grid.data[k].to_host())
        o = oracle.real_view(f[k], n, 8, G)[0]
        assert np.array_equal(a, np.broadcast_to(a[0:1], a.shape)), f"{k}: rows differ"
        if exact:
            assert np.array_equal(a[0], o), k
        else:
            assert np.abs(a[0] - o).max() <= 1e-11 * np.abs(o).max(), k
        del a


def test_config3_sedov_16384_tuned_symmetry_and_conservation():
    """BASELINE configs[2]: Sedov 16384² GAD+minmod+euler_2nd, tuned arithmetic (no strip reduction exists: the blast
    radius scales with the cell size). Checked at full size:
    * against the EXACT arithmetic of the same kernels at the same size (itself bit-identical to the oracle wherever
      the oracle can follow): same cycle count, dt within 1e-12, fields within 1e-11 of the field maximum;
    * mirror symmetry x→−x and y→−y — ρ, E, p even; u odd in x, v odd in y. The reference's own formula breaks exact
      symmetry in the GAD velocity ratios (the +1e-6 in their denominators, ref src/riemann_schemes.jl:84-87, does not
      change sign with the velocities), so the bound is SYM_TOL of the field maximum, not zero;
    * mass and energy conserved to 1e-11 (ref test/conservation.jl); the blast has formed and has not reached far cells."""
    import armon_amd
    from armon_amd.solver import BlockGrid, conservation_vars, init_test, time_loop
    SYM_TOL, cyc = 1e-10, 12          # measured: 1.1e-12 after 12 cycles (tools/sedov_asym.py)
    fields = {}
    for exact in (True, False):
        params = armon_amd.ArmonParameters(test="Sedov", N=(N, N), maxcycle=cyc, silent=5, exact_arithmetic=exact)
        grid = BlockGrid(params)
        init_test(params, grid)
        m0, e0 = conservation_vars(params, grid)
        time_, dt, cycles, _, _ = time_loop(params, grid)
        m1, e1 = conservation_vars(params, grid)
        assert cycles == cyc and dt > 0 and time_ > 0
        assert abs(m1 - m0) <= 1e-11 * abs(m0) and abs(e1 - e0) <= 1e-11 * abs(e0)
        fields[exact] = (dt, {k: grid.real_view(grid.data[k].to_host()).copy() for k in NAMES})
        del grid, params
    (dt_e, fe), (dt_t, ft) = fields[True], fields[False]
    assert abs(dt_t - dt_e) <= 1e-12 * dt_e
    for k, sx, sy in (("rho", 1, 1), ("E", 1, 1), ("p", 1, 1), ("u", -1, 1), ("v", 1, -1)):
        a, scale = ft[k], np.abs(fe[k]).max()
        assert np.isfinite(a).all(), k
        assert np.abs(a - fe[k]).max() <= 1e-11 * scale, f"{k}: tuned vs exact arithmetic"
        assert np.abs(a - sx * a[:, ::-1]).max() <= SYM_TOL * scale, f"{k}: x -> -x symmetry"
        assert np.abs(a - sy * a[::-1, :]).max() <= SYM_TOL * scale, f"{k}: y -> -y symmetry"
    rho, c = ft["rho"], N // 2
    assert (rho > 0).all() and (ft["E"] > 0).all()
    assert np.unique(rho[c - 40:c + 40, c - 40:c + 40]).size > 50                  # the blast wave has formed
    assert (rho[:c - 200] == 1.0).all() and (rho[:, :c - 200] == 1.0).all()        # and has not reached far cells


def test_config5_bizarrium_16384_tuned(oracle):
    """BASELINE configs[4]'s test case on one GPU, tuned arithmetic: rows identical, within the tuned tolerance of
    the oracle's 16384×8 strip, same cycle count."""
    import armon_amd
    orun, f = oracle.solve(test="Bizarrium", N=(N, 8), domain_size=(1., 8. / N), maxcycle=CYCLES, threads=8)
    params = armon_amd.ArmonParameters(test="Bizarrium", N=(N, N), maxcycle=CYCLES, silent=5, return_data=True)
    stats = armon_amd.armon(params)
    assert stats.cycles == orun.cycles and abs(stats.last_dt - orun.last_dt) <= 1e-12 * orun.last_dt
    for k in NAMES:
        a = stats.data.real_view(stats.data.data[k].to_host())
        o = oracle.real_view(f[k], N, 8, G)[0]
        assert np.array_equal(a, np.broadcast_to(a[0:1], a.shape)), f"{k}: rows differ"
        assert np.abs(a[0] - o).max() <= 1e-11 * np.abs(o).max(), k
        del a


@pytest.mark.parametrize("test,shape,mode", [
    ("Sod", (32768, 16384), "fused-tuned"),      # BASELINE config 4's global grid on ONE GPU: 4.3 GB per vector
    ("Sod", (32768, 16384), "staged"),
    ("Sod_y", (16384, 32768), "fused-exact"),
    ("Bizarrium", (32768, 32768), "fused-tuned"),   # config 5's global grid on ONE GPU: 8.6 GB per vector, 172 GB in all
])
def test_vectors_beyond_4GiB(oracle, test, shape, mode):
    """Maximum sizes: 537 M cells per vector (> 2³² bytes — every 32-bit byte offset in the kernels has to be relative
    to something close), non-square cells (the domain stays 1 × 1). Same properties as at 16384²: identical lines,
    the common line equal to the oracle's strip (bit for bit in exact arithmetic), dt, conservation."""
    import armon_amd
    from armon_amd.solver import BlockGrid, conservation_vars, init_test, time_loop
    nx, ny = shape
    along_x = test != "Sod_y"
    n_strip = (nx, 8) if along_x else (8, ny)
    ds = (1., 8. / ny) if along_x else (8. / nx, 1.)
    cycles = 3
    orun, f = oracle.solve(test=test, N=n_strip, domain_size=ds, maxcycle=cycles, threads=8)
    params = armon_amd.ArmonParameters(test=test, N=shape, maxcycle=cycles, silent=5,
                                       use_fused_sweep=mode != "staged", exact_arithmetic=mode != "fused-tuned")
    grid = BlockGrid(params)
    assert grid.data["rho"].nbytes > 1 << 32
    init_test(params, grid)
    m0, e0 = conservation_vars(params, grid)
    time_, dt, ncyc, _, _ = time_loop(params, grid)
    m1, e1 = conservation_vars(params, grid)
    if test != "Bizarrium":                       # Bizarrium has an inflow boundary (ref src/tests.jl:48-49)
        assert abs(m1 - m0) <= 1e-11 * abs(m0) and abs(e1 - e0) <= 1e-11 * abs(e0)
    assert ncyc == orun.cycles == cycles
    exact = mode != "fused-tuned"
    if exact:
        assert dt == orun.last_dt and time_ == orun.final_time
    else:
        assert abs(dt - orun.last_dt) <= 1e-12 * orun.last_dt
    for k in ("rho", "u" if along_x else "v", "E"):
        a = grid.real_view(grid.data[k].to_host())
        o = oracle.real_view(f[k], n_strip[0], n_strip[1], G)
        line = a[0:1] if along_x else a[:, 0:1]
        assert np.array_equal(a, np.broadcast_to(line, a.shape)), f"{k}: lines differ"
        oline = o[0] if along_x else o[:, 0]
        if exact:
            assert np.array_equal(line.ravel(), oline), k
        else:
            assert np.abs(line.ravel() - oline).max() <= 1e-11 * np.abs(oline).max(), k
        del a


@pytest.mark.parametrize("P,shape,test", [((2, 2), (32768, 16384), "Sod"), ((4, 2), (32768, 32768), "Bizarrium")],
                         ids=["config4", "config5"])
def test_baseline_tile_configurations_at_full_size(P, shape, test):
    """BASELINE configs 4 and 5 as they are written — Sod 32768×16384 on 2×2 tiles of 16384×8192, Bizarrium 32768² on
    4×2 tiles of 8192×16384 — through the library's multi-GPU entry points (armon_hip_mgpu_init, halo_exchange_start /
    finish around the interior sweeps, dt_allreduce), every tile on this one GPU: the tiles together must hold the same
    bits as the single block of the same grid (itself held against the oracle above), with the same dt and time."""
    import gc
    import armon_amd
    from armon_amd.multi_tile import TileGroup
    kw = dict(test=test, N=shape, maxcycle=3, silent=5)          # tuned arithmetic: what bench.py runs
    ref = armon_amd.armon(armon_amd.ArmonParameters(return_data=True, **kw))
    want = (ref.cycles, ref.last_dt, ref.final_time)
    full = {k: ref.data.real_view(ref.data.data[k].to_host()) for k in ("rho", "E")}
    del ref
    gc.collect()                                                  # 86 / 172 GB back before the tiles take as much
    group = TileGroup(P, **kw)
    try:
        assert [tuple(p.N) for p in group.params][0] == (shape[0] // P[0], shape[1] // P[1])
        stats = group.run()
        assert (stats.cycles, stats.last_dt, stats.final_time) == want
        got = group.gather(("rho", "E"))
        for k in ("rho", "E"):
            assert np.array_equal(got[k], full[k]), k
    finally:
        group.close()


# ---- problems that vary along BOTH axes, at sizes with many runs per column, many strips per row and several XCD groups ------
# (parity hole 1.) Every full-size check above is a 1-D problem reduced to a strip; the 2-D problems were held
# against the oracle up to 1031 x 130 / 770 x 515 only — one run of the Y march per column, fewer than two whole groups of the
# X sweep's XCD-aware workgroup order. Here the oracle runs the SAME size (all host cores; ≈ 2 GB of host arrays at 4096²):
#   Sod_circ, Sedov 4096²     : 16 column blocks x 15 runs of 274 rows, 1024 rows of X workgroups = 128 XCD groups
#   Bizarrium 2048 x 6000     : Y_only, 9 column blocks x >= 10 runs per column, its own EOS
#   Sod_circ 3000 x 133       : 33 rows of X workgroups + 1 ragged: 4 whole XCD groups and a partial one (ny = 4·8·4 + 5)
#   Sedov 1500 x 261          : 8 whole groups + 5 rows (ny = 4·8·8 + 5), FreeFlow sides
# exact arithmetic (fused and staged): every bit of rho, u, v, E, p, dt, time; tuned: the stated 1e-11 x max|field| (SURVEY §8c;
# the reference's own bar for the 100² goldens is atol 1e-13 / rtol 4 eps, test/reference_data/reference_functions.jl:54-57).
BIG_2D = [
    ("Sod_circ", (4096, 4096), {}),
    ("Sedov", (4096, 4096), {}),
    ("Bizarrium", (2048, 6000), dict(axis_splitting="Y_only")),
    ("Sod_circ", (3000, 133), {}),
    ("Sedov", (1500, 261), dict(axis_splitting="Godunov")),
]
_oracle_cache = {}


def _oracle_2d(oracle, test, shape, opts):
    key = (test, shape, tuple(sorted(opts.items())))
    if key not in _oracle_cache:
        import os
        _oracle_cache.clear()                                  # one problem at a time: 2 GB each
        run, f = oracle.solve(test=test, N=shape, maxcycle=CYCLES, threads=min(32, os.cpu_count() or 8), **opts)
        _oracle_cache[key] = (run.cycles, run.last_dt, run.final_time,
                              {k: oracle.real_view(f[k], shape[0], shape[1], G).copy() for k in NAMES})
    return _oracle_cache[key]


@pytest.mark.parametrize("mode", ["fused-exact", "staged", "fused-tuned"])
@pytest.mark.parametrize("test,shape,opts", BIG_2D, ids=[f"{t}-{s[0]}x{s[1]}" for t, s, _ in BIG_2D])
def test_two_dimensional_problems_at_scale_against_the_oracle(oracle, test, shape, opts, mode):
    import armon_amd
    cycles, last_dt, final_time, ref = _oracle_2d(oracle, test, shape, opts)
    params = armon_amd.ArmonParameters(test=test, N=shape, maxcycle=CYCLES, silent=5, return_data=True,
                                       use_fused_sweep=mode != "staged", exact_arithmetic=mode != "fused-tuned", **opts)
    stats = armon_amd.armon(params)
    assert stats.cycles == cycles == CYCLES
    if mode != "fused-tuned":
        assert stats.last_dt == last_dt and stats.final_time == final_time
    else:
        assert abs(stats.last_dt - last_dt) <= 1e-12 * last_dt
    for k in NAMES:
        a = stats.data.real_view(stats.data.data[k].to_host())
        if mode != "fused-tuned":
            assert np.array_equal(a, ref[k]), f"{k}: {np.count_nonzero(a != ref[k])} cells differ, max {np.abs(a - ref[k]).max()}"
        else:
            assert np.abs(a - ref[k]).max() <= 1e-11 * np.abs(ref[k]).max(), k
        del a


def test_y_runs_and_xcd_groups_of_the_2d_shapes():
    """What the shapes above are chosen for, read back from the library's own launch model: ARMON_Y_SEG / ARMON_X_XCD knobs
    give the same bits (covered elsewhere); here only the counts, so that a change of the model cannot silently turn these
    tests into one-run / one-group cases."""
    import armon_amd
    from armon_amd.solver import BlockGrid, init_test, fused_sweep
    from armon_amd.blocking import Axis
    for shape, min_runs in (((4096, 4096), 10), ((2048, 6000), 10)):
        params = armon_amd.ArmonParameters(test="Sod_circ", N=shape, silent=5, placement_tries=0)
        grid = BlockGrid(params)
        init_test(params, grid)
        fused_sweep(params, grid, Axis.Y, 1e-6, params.cell_size(1))
        params.wait()
        seg = params.device.y_run_rows()
        assert seg > 0 and -(-shape[1] // seg) >= min_runs, (shape, seg)
    for ny in (133, 261):
        assert (ny - 5) % 32 == 0 and (ny - 5) // 32 >= 3          # whole groups of 8 rows of 4-row workgroups + a ragged rest
