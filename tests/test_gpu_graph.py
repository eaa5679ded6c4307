"""Graph replay of the cycle with the dt state machine on the device (armon_dt_state, armon_hip_dt_state_step,
armon_hip_graph_*; host: solver.time_loop_graph, option graph_cycles=True): the same bits, cycle count, final time and last
dt as the host-driven time loop (ref src/solver.jl:323-403, src/solver_state.jl:102-166), whichever of maxtime / maxcycle
ends the run, for every splitting, both EOS, both arithmetics, fp32, constant dt — and the invalid-step error."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def both(test, N, **o):
    import armon_amd
    out = []
    for graph in (False, True):
        params = armon_amd.ArmonParameters(test=test, N=N, silent=5, return_data=True, graph_cycles=graph, **o)
        stats = armon_amd.armon(params)
        host = stats.data.device_to_host(("rho", "u", "v", "E", "p"))
        out.append((stats, {k: stats.data.real_view(v).copy() for k, v in host.items()}))
    return out


@pytest.mark.parametrize("test,N,opts", [
    ("Sod", (100, 100), {}),                                                   # ends on maxtime (45 cycles), the golden run
    ("Sod_circ", (67, 41), dict(maxcycle=37)),                                 # ends on maxcycle, not a multiple of the batch
    ("Sod_circ", (67, 41), dict(maxcycle=3)),                                  # the graph part runs a single cycle
    ("Sod_circ", (64, 48), dict(maxcycle=2)),                                  # never reaches the graph part
    ("Bizarrium", (96, 32), dict(maxcycle=30)),
    ("Sedov", (50, 50), dict(maxcycle=41)),
    ("Sod_circ", (80, 56), dict(maxcycle=26, axis_splitting="Godunov")),       # two graphs (cycle parity)
    ("Sod_circ", (80, 56), dict(maxcycle=27, axis_splitting="Strang")),        # three sweeps: the ping-pong parity alternates
    ("Sod", (96, 8), dict(maxcycle=25, axis_splitting="X_only")),
    ("Sod_y", (8, 96), dict(maxcycle=24, axis_splitting="Y_only")),
    ("Sod_circ", (48, 48), dict(maxcycle=30, scheme="Godunov", projection="euler", nghost=2)),
    ("Sod_circ", (64, 64), dict(maxcycle=33, cst_dt=True, Dt=1e-4)),
    ("Sod_circ", (64, 64), dict(maxtime=0.004, cst_dt=True, Dt=3e-4)),
    ("Sod", (200, 20), dict(maxtime=0.05)),
])
@pytest.mark.parametrize("exact", [True, False], ids=["exact", "tuned"])
def test_graph_replay_equals_the_host_driven_loop(test, N, opts, exact):
    (s0, f0), (s1, f1) = both(test, N, exact_arithmetic=exact, **opts)
    assert s1.cycles == s0.cycles and s1.final_time == s0.final_time and s1.last_dt == s0.last_dt
    for k in f0:
        assert np.array_equal(f1[k], f0[k]), k


@pytest.mark.parametrize("test,N,opts", [("Sod_circ", (67, 41), dict(maxcycle=37)), ("Sod", (100, 100), {}),
                                         ("Bizarrium", (64, 32), dict(maxcycle=20, axis_splitting="Strang"))])
def test_graph_replay_f32(test, N, opts):
    (s0, f0), (s1, f1) = both(test, N, data_type="float32", **opts)
    assert s1.cycles == s0.cycles and s1.final_time == s0.final_time and s1.last_dt == s0.last_dt
    for k in f0:
        assert np.array_equal(f1[k], f0[k]), k


def test_graph_replay_matches_the_golden_and_the_oracle(oracle):
    from conftest import load_golden
    import armon_amd
    g = load_golden("Sod_circ")
    params = armon_amd.ArmonParameters(test="Sod_circ", N=(100, 100), silent=5, return_data=True, graph_cycles=True,
                                       exact_arithmetic=True)
    stats = armon_amd.armon(params)
    orun, f = oracle.solve(test="Sod_circ", N=(100, 100))
    assert stats.cycles == int(g["cycles"]) == orun.cycles and stats.last_dt == orun.last_dt and stats.final_time == orun.final_time
    host = stats.data.device_to_host(("rho", "u", "v", "E", "p"))
    for k in ("rho", "u", "v", "E", "p"):
        assert np.array_equal(stats.data.real_view(host[k]), oracle.real_view(f[k], 100, 100, 4)), k


def test_graph_replay_reports_an_invalid_time_step():
    """One cell goes bad after the hand-over to the device: the state machine stops the loop at the first non-finite CFL step
    and the host raises SolverException(:time) (ref src/solver_state.jl:123-124) instead of replaying NaNs to maxcycle."""
    import armon_amd
    from armon_amd import solver as S
    params = armon_amd.ArmonParameters(test="Sod_circ", N=(96, 64), maxcycle=100000, silent=5, graph_cycles=True)
    grid = S.BlockGrid(params)
    S.init_test(params, grid)

    def poison(params, grid):
        g = params.nghost
        a = grid.data["E"].to_host()
        a.reshape(params.block_size.size[1], params.block_size.size[0])[g + 30, g + 40] = -1.0
        grid.data["E"].copy_from_host(a)

    with pytest.raises(armon_amd.SolverException) as e:
        S.time_loop_graph(params, grid, _after_handover=poison)
    assert e.value.category == "time" and "cycle 3" in e.value.msg or "cycle 4" in e.value.msg, e.value.msg


def test_dt_state_step_follows_update_dt_and_next_cycle():
    """The one-thread kernel against the host's GlobalTimeStep (ref src/solver_state.jl:102-166) through the C ABI."""
    import ctypes as C
    import armon_amd
    from armon_amd import _lib
    from armon_amd.device import HIPDevice
    from armon_amd.solver import GlobalTimeStep
    dev = HIPDevice(0)
    L = _lib.lib()
    rng = np.random.default_rng(3)
    for dtype, suffix in ((np.float64, ""), (np.float32, "_f32")):
        params = armon_amd.ArmonParameters(test="Sod", N=(16, 16), silent=5, data_type=dtype, maxtime=0.01, maxcycle=60)
        T = params.T
        gdt = GlobalTimeStep(params)
        gdt.cycle, gdt.time, gdt.current_dt = 2, T(3e-4), T(1.5e-4)
        L_prev = T(2.1e-4)
        st = _lib.DtState(current_dt=float(gdt.current_dt), time=float(gdt.time), L_prev=float(L_prev), cycle=2)
        state = dev.empty(C.sizeof(st) // 8, np.float64)
        _lib.check(L.armon_hip_memcpy(dev.ctx, C.c_void_p(state.ptr), C.byref(st), C.sizeof(st), 1))
        scalar = dev.empty(2, dtype)
        for _ in range(70):
            L_new = T(rng.uniform(0.5e-4, 4e-4))
            scalar.copy_from_host(np.array([L_new, 0], dtype=dtype))
            _lib.check(getattr(L, "armon_hip_dt_state_step" + suffix)(dev.ctx, C.c_void_p(state.ptr), C.c_void_p(scalar.ptr),
                                                                        float(params.cfl), float(params.maxtime), params.maxcycle, 0, 0.0))
            got = _lib.DtState.from_buffer_copy(state.to_host().tobytes())
            over = not (gdt.time < T(params.maxtime) and gdt.cycle < params.maxcycle)
            if not over:
                gdt.update_dt(L_prev)
                gdt.next_cycle()
                L_prev = L_new
            assert got.cycle == gdt.cycle and got.time == float(gdt.time) and got.current_dt == float(gdt.current_dt), (dtype, got.cycle)
            now_over = not (gdt.time < T(params.maxtime) and gdt.cycle < params.maxcycle)
            assert bool(got.done) == now_over
            if not now_over:
                ends = gdt.cycle + 1 >= params.maxcycle or T(gdt.time + gdt.current_dt) >= T(params.maxtime)
                assert bool(got.emit_p) == bool(ends)
        assert got.done and not got.invalid
        # a NaN, an infinite and a negative CFL step stop the machine and are reported
        for bad in (np.nan, np.inf, -1e-4):
            st = _lib.DtState(current_dt=1e-4, time=0.0, L_prev=float(T(bad)), cycle=7)
            _lib.check(L.armon_hip_memcpy(dev.ctx, C.c_void_p(state.ptr), C.byref(st), C.sizeof(st), 1))
            _lib.check(getattr(L, "armon_hip_dt_state_step" + suffix)(dev.ctx, C.c_void_p(state.ptr), C.c_void_p(scalar.ptr),
                                                                        0.5, 1.0, 1000, 0, 0.0))
            got = _lib.DtState.from_buffer_copy(state.to_host().tobytes())
            assert got.done == 1 and got.invalid == 1 and got.invalid_cycle == 7 and got.cycle == 7
    dev.close()


def test_graph_replay_really_replays():
    import armon_amd
    params = armon_amd.ArmonParameters(test="Sod_circ", N=(80, 56), maxcycle=27, axis_splitting="Strang", silent=5,
                                       return_data=True, graph_cycles=True)
    stats = armon_amd.armon(params)
    rep = stats.data.graph_report
    assert rep["graphs"] == 2 and rep["cycles_replayed"] == 25 and stats.cycles == 27 and rep["cycles_enqueued"] >= 25


def test_graph_mode_is_refused_where_the_host_must_act_inside_a_cycle(tmp_path):
    """Per-cycle output needs the host inside the cycle: the option then falls back to the host-driven loop (same result)."""
    import armon_amd
    from armon_amd.solver import graph_cycles_usable
    assert graph_cycles_usable(armon_amd.ArmonParameters(test="Sod", N=(32, 32), silent=5, graph_cycles=True))
    for o in (dict(silent=1), dict(animation_step=2, output_dir=str(tmp_path)), dict(use_fused_sweep=False), dict(graph_cycles=False)):
        kw = dict(dict(silent=5, graph_cycles=True), **o)
        assert not graph_cycles_usable(armon_amd.ArmonParameters(test="Sod", N=(32, 32), **kw))


def test_graph_capture_guards():
    """The rules next to armon_hip_graph_* in include/armon_hip.h: capture only on a context that owns its stream; while a
    graph of a context is alive its reduction scratch may not move (a launch that would grow it fails instead of leaving the
    graph with a dangling pointer); after the graph is destroyed the same launch goes through."""
    import ctypes as C
    import armon_amd
    from armon_amd import _lib
    from armon_amd.blocking import Axis
    from armon_amd.device import HIPDevice
    from armon_amd.solver import BlockGrid, init_test, sweep_desc
    L = armon_amd.lib()
    # a context on a borrowed stream: capture is refused, and so is graph mode on the host side
    owner = HIPDevice(0)
    borrowed = HIPDevice(0, stream=owner.stream)
    assert L.armon_hip_graph_begin(borrowed.ctx) != 0
    assert b"owns its stream" in L.armon_hip_last_error()
    borrowed.close()
    owner.close()
    # a small block captured, then a much larger one on the same context: its dt tracking needs more partials
    small = armon_amd.ArmonParameters(test="Sod_circ", N=(64, 64), silent=5)
    gs = BlockGrid(small)
    init_test(small, gs)
    dev = small.device
    d = sweep_desc(small, gs, Axis.Y, 1e-4, 1.0 / 64, emit_dt=True)
    _lib.check(small.fn("sweep")(dev.ctx, C.byref(d)))                  # runs once outside the capture, as the rule says
    _lib.check(L.armon_hip_graph_begin(dev.ctx))
    _lib.check(small.fn("sweep")(dev.ctx, C.byref(d)))
    g = C.c_void_p()
    _lib.check(L.armon_hip_graph_end(dev.ctx, C.byref(g)))
    big = armon_amd.ArmonParameters(test="Sod_circ", N=(4096, 2048), silent=5, ctx=dev.ctx)
    gb = BlockGrid(big)
    init_test(big, gb, tune=False)
    db = sweep_desc(big, gb, Axis.X, 1e-5, 1.0 / 4096, emit_dt=True)    # one pair of maxima per WAVE: far more than 64² needed
    assert big.fn("sweep")(dev.ctx, C.byref(db)) != 0
    assert b"captured graph" in L.armon_hip_last_error()
    _lib.check(L.armon_hip_graph_launch(dev.ctx, g))                     # the graph is still whole
    small.wait()
    _lib.check(L.armon_hip_graph_destroy(g))
    _lib.check(big.fn("sweep")(dev.ctx, C.byref(db)))                    # … and now the scratch may grow
    small.wait()
    # `big` only borrowed the context: its vectors go before the context's owner does
    import gc
    del db, gb, big
    gc.collect()


def test_a_context_destroyed_before_its_graph_disowns_it():
    """The wrong order (a garbage-collected host's finalizers): destroying the context first must not leave the graph with a
    dangling back pointer — its handle stays valid for armon_hip_graph_destroy and is refused everywhere else."""
    import ctypes as C
    import armon_amd
    from armon_amd import _lib
    L = armon_amd.lib()
    ctx, other = C.c_void_p(), C.c_void_p()
    _lib.check(L.armon_hip_init(0, None, C.byref(ctx)))
    _lib.check(L.armon_hip_init(0, None, C.byref(other)))
    graphs = []
    for _ in range(3):
        _lib.check(L.armon_hip_graph_begin(ctx))
        # (an empty capture is a valid graph)
        g = C.c_void_p()
        _lib.check(L.armon_hip_graph_end(ctx, C.byref(g)))
        graphs.append(g)
    assert L.armon_hip_graph_launch(other, graphs[0]) != 0 and b"another context" in L.armon_hip_last_error()
    _lib.check(L.armon_hip_graph_destroy(graphs.pop(1)))          # the middle one, the right way round
    _lib.check(L.armon_hip_destroy(ctx))                          # … then the context, with two graphs alive
    assert L.armon_hip_graph_launch(other, graphs[0]) != 0 and b"has been destroyed" in L.armon_hip_last_error()
    for g in graphs:
        _lib.check(L.armon_hip_graph_destroy(g))                  # no write through the dead context
    _lib.check(L.armon_hip_destroy(other))


def test_dt_state_is_refused_by_the_kernels_that_ignore_it():
    """The LDS X march and the whole-cycle kernels of the A/B build never read armon_dt_state: a descriptor that carries
    one is an error there, not a silently wrong time step."""
    import ctypes as C
    import armon_amd
    from armon_amd import _lib
    from armon_amd.blocking import Axis
    from armon_amd.solver import BlockGrid, init_test, sweep_desc
    with _lib.alt_kernels():
        L = armon_amd.lib()
        params = armon_amd.ArmonParameters(test="Sod_circ", N=(64, 48), silent=5)
        grid = BlockGrid(params)
        init_test(params, grid)
        state = params.device.zeros(C.sizeof(_lib.DtState) // 8)
        dx = sweep_desc(params, grid, Axis.X, 1.0, 1.0 / 64)
        dy = sweep_desc(params, grid, Axis.Y, 1.0, 1.0 / 48)
        dx.dt_state = state.ptr
        dx.x_kernel = 2
        assert params.fn("sweep")(params.device.ctx, C.byref(dx)) != 0 and b"dt_state" in L.armon_hip_last_error()
        dx.x_kernel = 0
        if params.suffix == "":
            assert L.armon_hip_cycle_xy(params.device.ctx, C.byref(dx), C.byref(dy)) != 0 and b"dt_state" in L.armon_hip_last_error()
        params.wait()


def test_a_failed_capture_leaves_the_context_and_the_ping_pong_usable(monkeypatch):
    """An exception while a cycle is being captured must end the capture, leave no graph behind and put the ping-pong of
    the two state sets back: the same grid then runs host-driven to the result of an undisturbed run."""
    import armon_amd
    from armon_amd import solver
    from armon_amd.solver import BlockGrid, init_test, time_loop
    kw = dict(test="Sod_circ", N=(80, 56), maxcycle=21, axis_splitting="Strang", silent=5)
    ref_params = armon_amd.ArmonParameters(**kw)
    ref_grid = BlockGrid(ref_params)
    init_test(ref_params, ref_grid)
    ref = time_loop(ref_params, ref_grid)
    ref_rho = ref_grid.real_view(ref_grid.device_to_host(("rho",))["rho"]).copy()

    params = armon_amd.ArmonParameters(graph_cycles=True, **kw)
    grid = BlockGrid(params)
    init_test(params, grid)
    real = solver._graph_sweep_descs
    calls = {"n": 0}

    def failing(p, g, parity, state_ptr):
        calls["n"] += 1
        descs = real(p, g, parity, state_ptr)          # swaps the state sets as it goes (3 sweeps: an odd number)
        raise RuntimeError("injected while capturing")

    monkeypatch.setattr(solver, "_graph_sweep_descs", failing)
    origin = grid.data["rho"].ptr
    with pytest.raises(RuntimeError, match="injected"):
        time_loop(params, grid)
    assert calls["n"] == 1
    # cycles 0 and 1 ran host-driven (2 x 3 sweeps: back on the first set); the aborted capture's 3 swaps were undone
    assert grid.data["rho"].ptr == origin
    monkeypatch.setattr(solver, "_graph_sweep_descs", real)
    # the stream is out of capture mode: a capture can start again (and is closed again, graph or not)
    import ctypes as C
    L = armon_amd.lib()
    assert L.armon_hip_graph_begin(params.device.ctx) == 0
    g = C.c_void_p()
    if L.armon_hip_graph_end(params.device.ctx, C.byref(g)) == 0:
        L.armon_hip_graph_destroy(g)
    # from the initial state again, host-driven: the undisturbed run's bits
    params.graph_cycles = False
    init_test(params, grid, tune=False)
    out = time_loop(params, grid)
    assert out[:3] == ref[:3]
    assert np.array_equal(grid.real_view(grid.device_to_host(("rho",))["rho"]), ref_rho)
