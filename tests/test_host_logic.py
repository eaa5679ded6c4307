"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, and the host mirror of
the reference interface (index math, ranges, parameters, dt state machine, splitting, process grid) behaves
like the reference. Index tests follow ref test/blocking.jl:108-183 and test/domains.jl."""
import math
import os
import re

import numpy as np
import pytest

import armon_amd
from armon_amd.blocking import Axis, BlockSize, Side, StepRange, axis_of, compute_steps_ranges
from armon_amd.parameters import ArmonParameters, cart_coords, cart_neighbours, proc_grid_for
from armon_amd.solver import GlobalTimeStep, split_axes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- the boundary ---------------------------------------------------------------------------------------
def declared_symbols():
    text = open(os.path.join(ROOT, "include", "armon_hip.h")).read()
    return sorted(set(re.findall(r"ARMON_API[^;]*?\b(armon_hip_\w+)\s*\(", text)))


def test_library_exports_every_symbol_of_the_header():
    import ctypes
    syms = declared_symbols()
    assert len(syms) >= 34
    from armon_amd._lib import ALT_LIB_PATH
    for path in (armon_amd.LIB_PATH, ALT_LIB_PATH):      # the product library and the A/B build with the alternative kernels
        L = ctypes.CDLL(path)
        for s in syms:
            assert hasattr(L, s), f"{s} declared in include/armon_hip.h but not exported by {path}"
    # and the ctypes binding table covers exactly the same set
    from armon_amd._lib import SIGNATURES
    assert sorted(SIGNATURES) == syms


def test_library_sanity_calls_without_gpu():
    L = armon_amd.lib()
    assert L.armon_hip_flt_size() == 8 and L.armon_hip_idx_size() == 8       # ref ext/ArmonKokkos.jl:122-139
    assert b"gfx950" in L.armon_hip_version()


def test_no_device_is_reported_not_hidden():
    import ctypes as C
    L = armon_amd.lib()
    n = C.c_int(-1)
    rc = L.armon_hip_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    assert L.armon_hip_init(0, None, C.byref(ctx)) == 3          # ARMON_ERR_NO_DEVICE
    assert b"device" in L.armon_hip_last_error().lower()
    with pytest.raises(armon_amd.SolverException):
        ArmonParameters(test="Sod", N=(8, 8)).device              # no CPU fallback: creating the device raises


# ---- BlockSize index math (ref test/blocking.jl:108-183) ------------------------------------------------------
@pytest.mark.parametrize("size,g", [((64, 64), 5), ((37, 39), 5), ((64, 37), 5), ((37, 39), 1)])
def test_block_size_index_math(size, g):
    bs = BlockSize(size, g)
    rs = bs.real_size
    assert rs == (size[0] - 2 * g, size[1] - 2 * g)
    full = bs.domain_range((-g, -g), (g, g))
    assert full.size == size
    for ij, j in enumerate(full.col, start=1):
        for ii, i in enumerate(range(full.row.first + j - 1, full.row.stop + j), start=1):
            I = (ii - g, ij - g)
            assert bs.position(i) == I
            assert bs.lin_position(I) == i
            assert bs.is_ghost(i) == (any(c <= 0 for c in I) or any(I[d] > rs[d] for d in range(2)))
    real = bs.domain_range()
    assert real.size == rs
    assert all(not bs.is_ghost(i) for i in real)
    for side in Side:
        border = bs.border_domain(side)
        other = rs[1] if axis_of(side) == Axis.X else rs[0]
        assert len(border) == other
        assert border.size == ((1, other) if axis_of(side) == Axis.X else (other, 1))
        gb = bs.ghost_domain(side, single_strip=False)
        assert len(gb) == other * g
        assert gb.size == ((g, other) if axis_of(side) == Axis.X else (other, g))
        assert all(bs.is_ghost(i) for i in gb)
        assert all(not bs.is_ghost(i) for i in bs.border_domain(side, single_strip=False))
    assert bs.stride_along(Axis.X) == abs(bs.lin_position((1, 1)) - bs.lin_position((2, 1)))
    assert bs.stride_along(Axis.Y) == abs(bs.lin_position((1, 1)) - bs.lin_position((1, 2)))
    assert bs.size_along(Axis.X) == bs.size_along(Side.Left) == size[0]
    assert bs.size_along(Axis.Y) == bs.size_along(Side.Top) == size[1]


# ---- range utilities (ref test/domains.jl) ------------------------------------------------------------------
@pytest.mark.parametrize("r,n", [(StepRange(1, 1, 10), 5), (StepRange(1, 50, 1051), 17)])
def test_step_range_utilities(r, n):
    assert len(r.shift(n)) == len(r) and r.shift(n).first == r.first + n * r.step
    assert len(r.expand(n)) == len(r) + n and r.expand(n).first == r.first
    assert len(r.prepend(n)) == len(r) + n and r.prepend(n).stop == r.stop
    assert len(r.inflate(n)) == len(r) + 2 * n
    assert len(r.expand(-len(r))) == 0 and len(r.prepend(1 - len(r))) == 1


def test_domain_range_to_c_and_membership():
    bs = BlockSize((108, 108), 4)
    dr = bs.domain_range()
    c = dr.to_c()
    assert (c.col_start, c.col_step, c.col_len, c.row_start, c.row_len) == (436, 108, 100, 0, 100)
    assert dr.first() in dr and dr.last() in dr and (dr.first() - 1) not in dr
    g = bs.ghost_domain(Side.Left, single_strip=False).to_c()
    assert g.row_start == -4 and g.row_len == 4          # rows shifted below 1: legal, first index still >= 0
    assert g.col_start + g.row_start >= 0


def test_steps_ranges_match_reference_table():
    """ref src/parameters.jl:992-1025 with euler_2nd (w = 2)."""
    sx = compute_steps_ranges(Axis.X, 4, 2)
    assert sx.fluxes == ((-2, 0), (3, 0)) and sx.cell_update == ((-2, 0), (2, 0)) and sx.advection == ((0, 0), (1, 0))
    sy = compute_steps_ranges(Axis.Y, 4, 1)
    assert sy.fluxes == ((0, -1), (0, 2)) and sy.cell_update == ((0, -1), (0, 1)) and sy.advection == ((0, 0), (0, 1))
    assert sx.EOS == sx.projection == sx.real_domain == ((0, 0), (0, 0)) and sx.full_domain == ((-4, -4), (4, 4))


# ---- ArmonParameters (ref src/parameters.jl) ---------------------------------------------------------------
def test_parameters_defaults_follow_the_test_case():
    for test, cfl, maxtime, dom in [("Sod", 0.95, 0.20, (1., 1.)), ("Bizarrium", 0.6, 80e-6, (1., 1.)),
                                    ("Sedov", 0.7, 1.0, (2., 2.))]:
        p = ArmonParameters(test=test, N=(50, 40))
        assert (p.cfl, p.maxtime, p.domain_size) == (cfl, maxtime, dom)
        assert p.nghost == 4 and p.riemann_scheme == "GAD" and p.projection_scheme == "euler_2nd"
        assert p.block_size.size == (58, 48) and p.N_origin == (1, 1) and p.global_grid == (50, 40)
    p = ArmonParameters(test="Sedov", N=(100, 100))
    assert p.test.r == float(np.hypot(0.02, 0.02) / math.sqrt(2))          # ref src/tests.jl:15-19
    assert p.test.boundary_condition(Side.Left) == (1., 1.)
    assert ArmonParameters(test="Sod").test.boundary_condition(Side.Left) == (-1., 1.)
    assert ArmonParameters(test="Sod_y").test.boundary_condition(Side.Top) == (1., -1.)
    assert ArmonParameters(test="Bizarrium").test.boundary_condition(Side.Right) == (1., 1.)


@pytest.mark.parametrize("kw,category", [
    (dict(test="Nope"), "config"),
    (dict(scheme="WENO"), "config"),
    (dict(riemann_limiter="vanleer"), "config"),
    (dict(projection="euler_3rd"), "config"),
    (dict(axis_splitting="Diagonal"), "config"),
    (dict(nghost=3), "config"),                         # GAD + euler_2nd needs 4 (ref src/parameters.jl:609-613)
    (dict(scheme="Godunov", projection="euler", nghost=1), "config"),   # stricter than the reference: see DESIGN §2
    (dict(cst_dt=True, Dt=0.), "config"),
    (dict(P=(1, 1, 1)), "config"),
    (dict(use_gpu=False), "config"),
    (dict(data_type=np.float16), "config"),
])
def test_parameters_reject_invalid_configurations(kw, category):
    with pytest.raises(armon_amd.SolverException) as e:
        ArmonParameters(**kw)
    assert e.value.category == category


def test_parameters_reject_unknown_options():
    """ref src/parameters.jl:369-372"""
    with pytest.raises(ValueError, match="unconsumed options"):
        ArmonParameters(test="Sod", not_an_option=1)


def test_reference_cpu_options_are_accepted():
    """A reference script's kwargs run unchanged (the CPU-machinery knobs are inert on the device path)."""
    p = ArmonParameters(test="Sod", N=(32, 32), use_threading=True, use_simd=True, numa_aware=False,
                        lock_memory=False, busy_wait_limit=100, measure_time=True, silent=3,
                        write_output=False, output_precision=None, animation_step=0, maxcycle=10)
    assert p.maxcycle == 10 and p.use_fused_sweep and not p.exact_arithmetic


# ---- dt state machine (ref src/solver_state.jl:102-166) ------------------------------------------
@pytest.mark.parametrize("N", [(7, 38), (514, 238), (50, 50), (120, 200), (333, 77)])
def test_sedov_radius_in_float32_is_evaluated_as_the_reference_does(N):
    """ref src/tests.jl:13-19: ``r::T = hypot(Δx...) / sqrt(2)`` — hypot in T, the division in Float64, then the conversion to
    T. numpy 2 would keep ``float32 / python_float`` in float32 (one ulp off on non-square cells); the oracle's C evaluation
    (float hypotf, double division) is the reference's."""
    from armon_amd.test_cases import create_test
    from oracle import oracle
    T = np.float32
    dX = (T(2) / T(N[0]), T(2) / T(N[1]))
    r = create_test("Sedov", dX, T).r
    assert r == float(T(float(np.hypot(dX[0], dX[1])) / math.sqrt(2.0)))
    # the oracle's initial state holds the same radius: the energy of its high region is T((1/1.033)^5 / (π r²))
    run, f = oracle.solve(test="Sedov", N=N, maxcycle=0, data_type=np.float32)
    e_hi = oracle.real_view(f["E"], N[0], N[1], 4).max()
    assert e_hi == T(math.pow(1. / 1.033, 5) / float(T(T(math.pi) * T(T(r) * T(r)))))


def test_global_time_step_lag_and_growth_cap():
    p = ArmonParameters(test="Sod", N=(8, 8), cfl=0.5)
    g = GlobalTimeStep(p)
    g.update_dt(1.0)                      # cycle 0: previous dt == 0 → dt0 = cfl·local, used at once
    assert g.current_dt == 0.5 and g.next_cycle_dt == 0.5
    g.next_cycle()
    assert (g.cycle, g.time, g.current_dt) == (1, 0.5, 0.5)
    g.update_dt(10.0)                     # cycle 1 still runs with dt0; growth capped at +5 %
    assert g.current_dt == 0.5 and g.next_cycle_dt == 1.05 * 0.5
    g.next_cycle()
    assert g.current_dt == 1.05 * 0.5 and g.time == 1.0
    g.update_dt(0.2)
    assert g.next_cycle_dt == 0.1
    for bad in (float("nan"), float("inf"), 0., -1.):
        with pytest.raises(armon_amd.SolverException) as e:
            g.update_dt(bad)
        assert e.value.category == "time"


def test_constant_dt():
    p = ArmonParameters(test="Sod", N=(8, 8), cst_dt=True, Dt=1e-3)
    g = GlobalTimeStep(p)
    assert g.current_dt == 1e-3
    g.next_cycle()
    assert g.current_dt == 1e-3 and g.time == 1e-3


def test_split_axes():
    """ref src/axis_splitting.jl:24-46"""
    X, Y = Axis.X, Axis.Y
    assert split_axes("Sequential", 3) == ((X, 1.0), (Y, 1.0))
    assert split_axes("Godunov", 0) == ((X, 1.0), (Y, 1.0)) and split_axes("SequentialSym", 1) == ((Y, 1.0), (X, 1.0))
    assert split_axes("Strang", 0) == ((X, 0.5), (Y, 1.0), (X, 0.5)) and split_axes("Strang", 1) == ((Y, 0.5), (X, 1.0), (Y, 0.5))
    assert split_axes("X_only", 5) == ((X, 1.0),) and split_axes("Y_only", 5) == ((Y, 1.0),)


# ---- process grid (ref src/parameters.jl:408-467,673-697; test/mpi.jl:200-223) --------------------------------
def test_cartesian_grid_and_neighbours():
    dims = (4, 2)
    assert [cart_coords(r, dims) for r in range(8)] == [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1), (3, 0), (3, 1)]
    nb = cart_neighbours((0, 0), dims)
    assert nb[Side.Left] == -1 and nb[Side.Bottom] == -1 and nb[Side.Right] == 2 and nb[Side.Top] == 1
    nb = cart_neighbours((3, 1), dims)
    assert nb[Side.Right] == -1 and nb[Side.Top] == -1 and nb[Side.Left] == 5 and nb[Side.Bottom] == 6
    assert [proc_grid_for(w) for w in (1, 2, 4, 8)] == [(1, 1), (2, 1), (2, 2), (4, 2)]


def test_memory_required():
    """Fused path: x, y, rho, u, v, E, p + the 4 ping-pong partners of the state; the nine staged-only vectors are allocated
    on first access (a 16384² block: 23.6 GB instead of 43; config 5's 32768² global grid on one GPU: 95 GB instead of 172)."""
    n = 16392 * 16392 * 8
    assert armon_amd.memory_required((16384, 16384), 4) == 11 * n
    assert armon_amd.memory_required((16384, 16384), 4, transient=True) == 19 * n        # + the 8 spares of the placement search
    assert armon_amd.memory_required((16384, 16384), 4, fused=False) == 16 * n
    assert armon_amd.memory_required((100, 100), 4, data_type="float32", transient=True) == 13 * 108 * 108 * 4   # + c, g in cycle 0


# ---- text I/O in the reference's format (ref src/io.jl) -----------------------------------------------------
def test_output_format_roundtrip_and_reference_line(tmp_path):
    import io as _io
    from armon_amd import io as aio
    from conftest import load_golden
    g = load_golden("Sod")
    p = ArmonParameters(test="Sod", N=(100, 100), output_dir=str(tmp_path))
    n = p.block_size.n_cells
    sx = p.block_size.size[0]
    host = {}
    for k in aio.SAVED_VARS:
        a = np.full((sx, sx), -7.0)
        a[4:104, 4:104] = g[k]
        host[k] = a.ravel()
    buf = _io.StringIO()
    aio.write_blocks_to_file(p, host, buf)
    lines = buf.getvalue().split("\n")
    # first cell of the reference's golden file, character for character (ref test/reference_data/ref_Sod_64bits.csv:2)
    assert lines[0] == (" 0.00000000000000000e+00,  0.00000000000000000e+00,  1.00000000000000000e+00,"
                        "  0.00000000000000000e+00,  0.00000000000000000e+00,  9.99999999999999778e-01")
    assert sum(1 for l in lines if l.strip()) == 100 * 100 and lines[100] == ""      # blank line between rows
    back = {k: np.full(n, np.nan) for k in aio.SAVED_VARS}
    aio.read_data_from_file(p, back, _io.StringIO(buf.getvalue()))
    for k in aio.SAVED_VARS:      # 17 significant digits round-trip fp64 exactly
        assert np.array_equal(np.asarray(back[k]).reshape(sx, sx)[4:104, 4:104], g[k])
    # golden-file reader (header + cells)
    path = tmp_path / "ref.csv"
    with open(path, "w") as f:
        f.write("%#.15g, %d\n" % (float(g["dt"]), int(g["cycles"])))
        f.write(buf.getvalue())
    dt, cycles, vals = aio.read_reference_file(str(path), (100, 100))
    assert cycles == 45 and abs(dt - float(g["dt"])) < 1e-17 and np.array_equal(vals["rho"], g["rho"])


def test_file_paths_and_time_step_file(tmp_path):
    from armon_amd import io as aio
    p = ArmonParameters(test="Sod", N=(8, 8), output_dir=str(tmp_path / "out"), output_file="run")
    assert aio.build_file_path(p, "run_003_EOS_X") == str(tmp_path / "out" / "run_003_EOS_X")
    aio.write_time_step_file(p, 0.00431962688696710, "dt")
    assert open(tmp_path / "out" / "dt").read() == " 4.31962688696709961e-03\n"      # %#24.17e
    assert aio.read_time_step_file(p, "dt") == 0.00431962688696710


def test_bench_refuses_an_impossible_process_grid_before_touching_a_gpu():
    """bench.py --grid must multiply to the number of ranks (checked before any device is created or any child started),
    and --transport peer is one process: refused under a launcher."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "1", "--grid", "2x2"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "cannot form that process grid" in r.stderr and r.stdout == ""
    r = subprocess.run([sys.executable, bench, "--gpus", "4", "--grid", "3x1"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "cannot form that process grid" in r.stderr and r.stdout == ""
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--transport", "peer"], capture_output=True, text=True,
                       env={**env, "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "without a launcher" in r.stderr and r.stdout == ""


def test_bench_launches_its_own_ranks_and_reports_failure_with_a_status():
    """`python bench.py --gpus 2` from a bare shell starts the ranks itself (child torch.distributed.run), falls back to the
    in-process transport when they give no line, and — here, where there is no GPU at all — ends with a non-zero status,
    an empty stdout and both attempts named on stderr. (The successful forms run on the GPU box: tests/test_gpu_bench.py.)"""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--cells", "64", "--steps", "1", "--warmup", "0", "--launch-timeout", "240"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and r.stdout == ""
    assert "rank launch under torch.distributed.run gave no line" in r.stderr and "falling back to --transport peer" in r.stderr
    assert "the in-process (peer) run gave no line either" in r.stderr


def test_design_md_quotes_what_the_committed_profiles_say():
    """DESIGN.md's measured tables sit between `evidence` markers and are injected from profiles/r05_* by
    tools/evidence_table.py: a number typed by hand, or a profile re-collected without re-injecting, fails here."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "evidence_table.py"), "5", "--check", os.path.join(ROOT, "DESIGN.md")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "evidence_table.py"), "5"], capture_output=True, text=True)
    assert gen.returncode == 0 and gen.stdout == open(os.path.join(ROOT, "profiles", "r05_evidence.md")).read()


def test_bench_launcher_helpers_kill_by_process_group_and_find_the_line(tmp_path):
    """bench.py's parent side without any GPU: a child that hangs is killed through ITS process group after the limit (and
    the grandchild it started with it), a child's stdout is searched for the one JSON line whatever the libraries printed
    around it, and the exit status travels."""
    import importlib.util
    import subprocess
    import sys
    import time
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    marker = tmp_path / "grandchild.pid"
    hang = tmp_path / "hang.py"
    hang.write_text("import subprocess, sys, time\n"
                    f"p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(120)'])\n"
                    f"open({str(marker)!r}, 'w').write(str(p.pid))\n"
                    "print('banner on stdout', flush=True)\ntime.sleep(120)\n")
    t0 = time.time()
    rc, out = bench.run_child([sys.executable, str(hang)], dict(os.environ), timeout=3)
    assert rc == -9 and time.time() - t0 < 30
    pid = int(marker.read_text())
    for _ in range(50):                                   # the grandchild went with the group
        if subprocess.run(["ps", "-p", str(pid)], capture_output=True).returncode != 0:
            break
        time.sleep(0.1)
    assert subprocess.run(["ps", "-p", str(pid), "-o", "stat="], capture_output=True, text=True).stdout.strip() in ("", "Z")
    ok = tmp_path / "ok.py"
    ok.write_text("import sys\nprint('RCCL version : x')\nprint('{\"not\": \"the line\"}')\n"
                  "print('{\"metric\": \"m\", \"value\": 1.5}')\nprint('trailing noise')\nsys.exit(3)\n")
    rc, out = bench.run_child([sys.executable, str(ok)], dict(os.environ), timeout=30)
    assert rc == 3 and bench.last_json_line(out) == {"metric": "m", "value": 1.5}
    assert bench.last_json_line("no json here\n") is None and bench.last_json_line("") is None
    # placement bookkeeping of the N > 1 line
    pl = bench.annotate_placements([{"rank": 0, "chosen_ms": 0.75, "tries": 12}, {"rank": 1, "tries": 0}, {"rank": 2, "chosen_ms": 0.80, "tries": 16}],
                                   [33554432, 33554432, 33554432])
    assert pl[0]["chosen_vs_group_best"] == 1.0 and round(pl[2]["chosen_vs_group_best"], 3) == 1.067 and "chosen_vs_group_best" not in pl[1]
    assert bench.slowest_rank(pl) == 2 and bench.slowest_rank([{"rank": 0, "tries": 0}]) is None and bench.slowest_rank(None) is None


def test_cycle_plan_follows_split_axes(monkeypatch):
    """multi_tile.cycle_plan = the armon_cycle_plan of one solver cycle: the sweeps of split_axes (ref src/axis_splitting.jl:
    24-46) with their steps current_dt x factor, the last cycle's emit_p, cst_dt's missing reduction, the NEXT cycle's first
    axis for the exchange posted ahead (it alternates under Godunov / Strang splitting), the landing slot of the next CFL step."""
    from armon_amd.multi_tile import cycle_plan
    from armon_amd.solver import DT_EVENT_SLOT, GlobalTimeStep

    class Pinned:
        ptr = 0x1000

    X, Y = 0, 1
    for splitting, expect in (("Sequential", {0: ([X, Y], [1, 1], X), 1: ([X, Y], [1, 1], X)}),
                              ("Godunov", {0: ([X, Y], [1, 1], Y), 1: ([Y, X], [1, 1], X)}),
                              ("Strang", {0: ([X, Y, X], [.5, 1, .5], Y), 1: ([Y, X, Y], [.5, 1, .5], X)}),
                              ("Y_only", {0: ([Y], [1], Y), 3: ([Y], [1], Y)})):
        p = ArmonParameters(test="Sod", N=(64, 64), axis_splitting=splitting)
        gdt = GlobalTimeStep(p)
        for cycle, (axes, factors, nxt) in expect.items():
            gdt.cycle, gdt.current_dt = cycle, 0.25
            plan, n = cycle_plan(p, gdt, last_cycle=False, dt_host=Pinned())
            assert n == plan.n_sweeps == len(axes) and list(plan.axis)[:n] == axes
            assert list(plan.dt)[:n] == [0.25 * f for f in factors]
            assert plan.next_axis == nxt and plan.emit_p == 0 and plan.emit_dt == 1 and plan.overlap == 1
            assert plan.dt_host == 0x1000 + 8 * (cycle & 1) and plan.dt_event_slot == DT_EVENT_SLOT + (cycle & 1) and plan.event_slot == -1
            last, _ = cycle_plan(p, gdt, last_cycle=True, dt_host=Pinned())
            assert last.emit_p == 1 and last.next_axis == -1
    p = ArmonParameters(test="Sod", N=(64, 64), cst_dt=True, Dt=1e-3, data_type="float32", overlap_halo=False)
    gdt = GlobalTimeStep(p)
    gdt.current_dt = p.T(1e-3)
    plan, _ = cycle_plan(p, gdt, last_cycle=False, dt_host=Pinned())
    assert plan.emit_dt == 0 and plan.dt_host is None and plan.dt_event_slot == -1 and plan.overlap == 0 and plan.next_axis == -1
