"""Tile-decomposed runs on the GPU must equal the single-block run bit for bit: same kernels,
deterministic ghost recompute. Two ranks share cuda:0 and talk over gloo with host staging (RCCL cannot put
two ranks on one device); the transport is the only difference from the multi-GPU configuration."""
import numpy as np
import pytest

import dist_workers
from test_distributed import spawn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [False, True], ids=["staged", "fused"])
@pytest.mark.parametrize("P,N,test,opts", [
    ((2, 1), (64, 40), "Sod_circ", dict(maxcycle=12)),
    ((1, 2), (64, 40), "Sod_circ", dict(maxcycle=12)),
    ((2, 1), (37, 41), "Sod", dict(maxcycle=8)),                    # uneven split (ref test/mpi.jl:551-561)
    ((1, 2), (48, 33), "Sod_y", dict(maxcycle=8, axis_splitting="Strang")),
])
def test_two_tiles_equal_one_block(tmp_path, P, N, test, opts, fused):
    check_tiles(tmp_path, P, N, test, opts, fused)


@pytest.mark.parametrize("fused", [False, True], ids=["staged", "fused"])
@pytest.mark.parametrize("P,N,test,opts", [
    ((3, 1), (70, 24), "Sod_circ", dict(maxcycle=10)),                  # middle tile: remote on BOTH sides of x (as px = 4)
    ((1, 3), (24, 70), "Sod_circ", dict(maxcycle=10)),
    ((2, 2), (48, 40), "Sod_circ", dict(maxcycle=10)),                  # remote sides on both axes
    ((2, 2), (45, 37), "Sedov", dict(maxcycle=10, axis_splitting="Godunov")),
])
def test_more_tiles_equal_one_block(tmp_path, P, N, test, opts, fused):
    check_tiles(tmp_path, P, N, test, opts, fused)


def check_tiles(tmp_path, P, N, test, opts, fused):
    import armon_amd
    world = P[0] * P[1]
    o = dict(opts, use_fused_sweep=fused, exact_arithmetic=True)
    spawn(dist_workers.gpu_solver_worker, world, P, N, test, o, str(tmp_path))
    params = armon_amd.ArmonParameters(test=test, N=N, silent=5, return_data=True, **o)
    ref = armon_amd.armon(params)
    host = ref.data.device_to_host(("rho", "u", "v", "E", "p"))
    full = {k: ref.data.real_view(v) for k, v in host.items()}
    for r in range(world):
        t = np.load(tmp_path / f"tile{r}.npz")
        assert int(t["cycles"]) == ref.cycles and float(t["dt"]) == ref.last_dt and float(t["time"]) == ref.final_time
        ox, oy = (int(v) - 1 for v in t["origin"])
        nx, ny = (int(v) for v in t["n"])
        for k in ("rho", "u", "v", "E", "p"):
            assert np.array_equal(t[k], full[k][oy:oy + ny, ox:ox + nx]), (r, k)


@pytest.mark.parametrize("fused", [False, True], ids=["staged", "fused"])
def test_single_rank_over_rccl_equals_plain_run(tmp_path, fused):
    """The RCCL configuration (context on an adopted torch stream, stream-ordered exchange, in-place device
    all-reduce of dt, all-reduce of the conservation sums) with the one rank a single GPU allows."""
    import armon_amd
    N, test = (96, 72), "Sod_circ"
    o = dict(maxcycle=15, use_fused_sweep=fused, exact_arithmetic=True, native_halo=False)   # torch.distributed's RCCL
    spawn(dist_workers.gpu_solver_worker, 1, (1, 1), N, test, o, str(tmp_path), "nccl")
    ref = armon_amd.armon(armon_amd.ArmonParameters(test=test, N=N, silent=5, return_data=True, **o))
    host = ref.data.device_to_host(("rho", "u", "v", "E", "p"))
    t = np.load(tmp_path / "tile0.npz")
    assert int(t["cycles"]) == ref.cycles and float(t["dt"]) == ref.last_dt and float(t["time"]) == ref.final_time
    for k in ("rho", "u", "v", "E", "p"):
        assert np.array_equal(t[k], ref.data.real_view(host[k])), k


# ---- Bizarrium tiles and the reference's uneven domains over gloo (ref test/mpi.jl:465-475,551-561) ----------------
@pytest.mark.parametrize("fused", [False, True], ids=["staged", "fused"])
@pytest.mark.parametrize("P,N,test,opts", [
    ((2, 1), (64, 24), "Bizarrium", dict(maxcycle=10)),                 # the interface sits on the tile boundary
    ((2, 2), (48, 40), "Bizarrium", dict(maxcycle=10)),
    ((2, 2), (107, 113), "Sod_circ", dict(maxcycle=6)),                 # uneven: 53+54 × 56+57
    ((1, 3), (37, 241), "Sedov", dict(maxcycle=6)),                     # uneven: 80+80+81
])
def test_bizarrium_and_uneven_tiles_equal_one_block(tmp_path, P, N, test, opts, fused):
    check_tiles(tmp_path, P, N, test, opts, fused)


# ---- the library's own multi-GPU entry points: every tile in ONE process on device 0 ---------------------------------
# armon_hip_mgpu_init + armon_hip_halo_exchange_start/finish + armon_hip_dt_allreduce: pack → peer copy on the
# tile's transfer stream → unpack, ordered by events only (no host synchronisation between pack and unpack), the
# interior of each fused sweep enqueued while the faces travel. Must equal the single block bit for bit.
@pytest.mark.parametrize("fused", [False, True], ids=["staged", "fused"])
@pytest.mark.parametrize("P,N,test,opts", [
    ((2, 2), (48, 40), "Sod_circ", dict(maxcycle=12)),
    ((2, 1), (64, 24), "Bizarrium", dict(maxcycle=10)),
    ((2, 2), (48, 40), "Bizarrium", dict(maxcycle=10)),
    ((4, 2), (96, 48), "Bizarrium", dict(maxcycle=10)),                 # BASELINE config 5's layout
    ((4, 2), (128, 64), "Sedov", dict(maxcycle=10, axis_splitting="Godunov")),
    ((2, 2), (107, 113), "Sod_circ", dict(maxcycle=8)),                 # ref test/mpi.jl:551-561 uneven domains
    ((1, 3), (37, 241), "Sod_circ", dict(maxcycle=8)),
    ((3, 2), (20, 20), "Sod_circ", dict(maxcycle=8)),                   # 6+6+8 × 10+10: tiles too small to overlap
    ((3, 3), (61, 59), "Sod_circ", dict(maxcycle=8, axis_splitting="Strang")),   # a tile with 4 neighbours
    ((5, 1), (83, 16), "Sod", dict(maxcycle=8, scheme="Godunov", projection="euler")),
])
def test_native_tile_group_equals_one_block(P, N, test, opts, fused):
    import armon_amd
    from armon_amd.multi_tile import TileGroup
    o = dict(opts, use_fused_sweep=fused, exact_arithmetic=True)
    ref = armon_amd.armon(armon_amd.ArmonParameters(test=test, N=N, silent=5, return_data=True, **o))
    host = ref.data.device_to_host(("rho", "u", "v", "E", "p"))
    full = {k: ref.data.real_view(v) for k, v in host.items()}
    group = TileGroup(P, test=test, N=N, silent=5, **o)
    try:
        stats = group.run()
        assert stats.cycles == ref.cycles and stats.last_dt == ref.last_dt and stats.final_time == ref.final_time
        got = group.gather()
        for k in ("rho", "u", "v", "E", "p"):
            assert np.array_equal(got[k], full[k]), k
        m, e = group.conservation_vars()
        from armon_amd.solver import conservation_vars
        m1, e1 = conservation_vars(ref.data.params, ref.data)
        assert abs(m - m1) <= 1e-13 * abs(m1) and abs(e - e1) <= 1e-13 * abs(e1)
    finally:
        group.close()


def test_native_tile_group_tuned_f32_and_no_overlap():
    """The other knobs of the same path: tuned arithmetic, Float32 (the _f32 entry points), overlap switched off, the
    boundary strips on the compute stream instead of the edge stream (the default everywhere else in this file)."""
    import armon_amd
    from armon_amd.multi_tile import TileGroup
    for o in (dict(), dict(data_type="float32", exact_arithmetic=True), dict(overlap_halo=False, exact_arithmetic=True),
              dict(edge_stream=False), dict(edge_stream=False, data_type="float32")):     # strips after the interior, in order
        kw = dict(test="Sod_circ", N=(96, 80), maxcycle=10, silent=5, **o)
        ref = armon_amd.armon(armon_amd.ArmonParameters(return_data=True, **kw))
        host = ref.data.device_to_host(("rho", "u", "v", "E", "p"))
        group = TileGroup((2, 2), **kw)
        try:
            stats = group.run()
            got = group.gather()
            assert stats.cycles == ref.cycles and stats.last_dt == ref.last_dt
            for k in ("rho", "u", "v", "E", "p"):
                assert np.array_equal(got[k], ref.data.real_view(host[k])), (o, k)
        finally:
            group.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("P,N,test,opts", [
    ((2, 2), (96, 80), "Sod_circ", dict(maxcycle=12)),
    ((4, 2), (160, 96), "Bizarrium", dict(maxcycle=10)),                       # BASELINE config 5's layout
    ((3, 3), (93, 99), "Sedov", dict(maxcycle=10, axis_splitting="Strang")),    # a tile with 4 neighbours, X-last cycles
    ((3, 1), (70, 48), "Sod", dict(maxcycle=10, edge_stream=False)),            # the in-order form of the strips
])
def test_native_tile_group_survives_injected_delays(P, N, test, opts, seed):
    """The event ordering of the library's exchange (pack → transfer → unpack → strips on the edge stream → join → next
    pack; the dt reduction) under timings one GPU never produces by itself: armon_hip_mgpu_set_chaos puts busy-wait
    kernels of up to 300 µs — longer than any kernel of these small tiles — in front of the group's stream operations
    at random. A missing dependency would show as stale ghost cells; the result must stay bit-identical."""
    import armon_amd
    from armon_amd.multi_tile import TileGroup
    o = dict(opts)
    ref = armon_amd.armon(armon_amd.ArmonParameters(test=test, N=N, silent=5, return_data=True,
                                                    **{k: v for k, v in o.items() if k != "edge_stream"}))
    host = ref.data.device_to_host(("rho", "u", "v", "E", "p"))
    group = TileGroup(P, test=test, N=N, silent=5, **o)
    try:
        group.set_chaos(300, seed)
        stats = group.run()
        assert stats.cycles == ref.cycles and stats.last_dt == ref.last_dt and stats.final_time == ref.final_time
        got = group.gather()
        for k in ("rho", "u", "v", "E", "p"):
            assert np.array_equal(got[k], ref.data.real_view(host[k])), k
    finally:
        group.close()


def test_native_halo_exchange_moves_index_encoded_faces():
    """The exchange alone, with the index-encoding design of ref test/mpi.jl:272-360: every cell of every variable
    holds var·1e7 + its GLOBAL linear index; after both axes each remote ghost cell holds the index of the cell it
    mirrors, physical sides and corners keep their marker."""
    import ctypes as C
    import armon_amd
    from armon_amd.blocking import Axis, Side
    from armon_amd.multi_tile import TileGroup
    from armon_amd.parameters import PROC_NULL
    P, NG, names = (3, 2), (41, 27), ("rho", "u", "v", "E", "p", "c", "g")
    group = TileGroup(P, test="Sod", N=NG, silent=5)
    try:
        g = group.root.nghost
        for p, grid in zip(group.params, group.grids):
            nx, ny = p.N
            gx0, gy0 = p.N_origin[0] - 1, p.N_origin[1] - 1
            for vi, k in enumerate(names):
                a = np.full((ny + 2 * g, nx + 2 * g), -1.0)
                iy, ix = np.mgrid[0:ny, 0:nx]
                a[g:g + ny, g:g + nx] = vi * 1e7 + (gy0 + iy) * NG[0] + (gx0 + ix)
                grid.data[k].copy_from_host(a.ravel())
        for subset in (names[:4], names):            # 4 variables first (the fused path's set), then 7: the face buffers grow
            for axis in (Axis.X, Axis.Y):
                group.exchange_start(axis, subset)
                group.exchange_finish(axis, subset)
        group.wait()
        for p, grid in zip(group.params, group.grids):
            nx, ny = p.N
            gx0, gy0 = p.N_origin[0] - 1, p.N_origin[1] - 1
            jj, ii = np.mgrid[-g:ny + g, -g:nx + g]
            for vi, k in enumerate(names):
                a = grid.data[k].to_host().reshape(ny + 2 * g, nx + 2 * g)
                expected = np.full_like(a, -1.0)
                enc = vi * 1e7 + (gy0 + jj) * NG[0] + (gx0 + ii)
                inside_x, inside_y = (ii >= 0) & (ii < nx), (jj >= 0) & (jj < ny)
                expected[inside_x & inside_y] = enc[inside_x & inside_y]
                for side, mask in ((Side.Left, inside_y & (ii < 0)), (Side.Right, inside_y & (ii >= nx)),
                                   (Side.Bottom, inside_x & (jj < 0)), (Side.Top, inside_x & (jj >= ny))):
                    if p.neighbours[side] != PROC_NULL:
                        expected[mask] = enc[mask]
                assert np.array_equal(a, expected), (p.rank, k)
    finally:
        group.close()


def test_single_rank_over_native_rccl_equals_plain_run(tmp_path):
    """armon_hip_mgpu_init_rank with the one rank a single GPU allows: RCCL loaded and initialised by the library
    (two communicators), the dt minimum reduced by ncclAllReduce on the compute stream, ids through the torch store."""
    import armon_amd
    N, test = (96, 72), "Sod_circ"
    o = dict(maxcycle=15, use_fused_sweep=True, exact_arithmetic=True)
    spawn(dist_workers.gpu_solver_worker, 1, (1, 1), N, test, o, str(tmp_path), "nccl")
    t = np.load(tmp_path / "tile0.npz")
    assert bool(t["native"])
    ref = armon_amd.armon(armon_amd.ArmonParameters(test=test, N=N, silent=5, return_data=True, **o))
    host = ref.data.device_to_host(("rho", "u", "v", "E", "p"))
    assert int(t["cycles"]) == ref.cycles and float(t["dt"]) == ref.last_dt
    for k in ("rho", "u", "v", "E", "p"):
        assert np.array_equal(t[k], ref.data.real_view(host[k])), k


# ---- the whole cycle of every tile in one library call (armon_hip_mgpu_cycle) -------------------------------------------
# The default of every fused tile-group test above. Here the three drivers of the same protocol are held against each other
# and against the single block: the host mirror calling the library step by step (native_cycle=False: exchange_start,
# interior, finish_edge, strips, edge_join, dt_allreduce), the native cycle from the calling thread, and the native cycle
# with one host thread per tile (barriers between the steps).
@pytest.mark.parametrize("driver", ["host-calls", "native-serial", "native-threads"])
@pytest.mark.parametrize("P,N,test,opts", [
    ((2, 2), (48, 40), "Sod_circ", dict(maxcycle=12)),
    ((4, 2), (128, 64), "Sedov", dict(maxcycle=10, axis_splitting="Godunov")),     # the first sweep's axis alternates: prefetch X, then Y
    ((3, 3), (61, 59), "Sod_circ", dict(maxcycle=9, axis_splitting="Strang")),     # 3 sweeps: the state ends in the other set
    ((3, 2), (20, 20), "Sod_circ", dict(maxcycle=8)),                              # tiles too small to overlap: in-order form per tile
    ((2, 2), (96, 80), "Bizarrium", dict(maxcycle=8, data_type="float32")),
    ((2, 1), (64, 24), "Sod", dict(maxcycle=8, cst_dt=True, Dt=1e-4)),             # no dt reduction at all
    ((1, 3), (37, 241), "Sod_circ", dict(maxcycle=8, overlap_halo=False)),
    ((4, 1), (96, 32), "Sod_circ", dict(maxcycle=7, axis_splitting="X_only")),     # one sweep per cycle, always prefetched
])
def test_tile_cycle_drivers_agree(P, N, test, opts, driver):
    import armon_amd
    from armon_amd.multi_tile import TileGroup
    o = dict(opts, exact_arithmetic=True)
    ref = armon_amd.armon(armon_amd.ArmonParameters(test=test, N=N, silent=5, return_data=True, **o))
    host = ref.data.device_to_host(("rho", "u", "v", "E", "p"))
    group = TileGroup(P, test=test, N=N, silent=5, native_cycle=driver != "host-calls", **o)
    try:
        group.set_threads(driver == "native-threads")
        stats = group.run()
        assert stats.cycles == ref.cycles and stats.last_dt == ref.last_dt and stats.final_time == ref.final_time
        got = group.gather()
        for k in ("rho", "u", "v", "E", "p"):
            assert np.array_equal(got[k], ref.data.real_view(host[k])), k
    finally:
        group.close()


@pytest.mark.parametrize("seed", [4, 5])
def test_threaded_tile_cycles_survive_injected_delays(seed):
    """One host thread per tile + busy-wait kernels of up to 300 µs in front of the group's stream operations."""
    import armon_amd
    from armon_amd.multi_tile import TileGroup
    kw = dict(test="Bizarrium", N=(160, 96), maxcycle=10, silent=5)
    ref = armon_amd.armon(armon_amd.ArmonParameters(return_data=True, **kw))
    host = ref.data.device_to_host(("rho", "u", "v", "E", "p"))
    group = TileGroup((4, 2), force_peer_copy=seed == 5, **kw)
    try:
        group.set_threads(True)
        group.set_chaos(300, seed)
        stats = group.run()
        assert stats.cycles == ref.cycles and stats.last_dt == ref.last_dt
        got = group.gather()
        for k in ("rho", "u", "v", "E", "p"):
            assert np.array_equal(got[k], ref.data.real_view(host[k])), k
    finally:
        group.close()


def test_tile_cycle_refuses_what_it_cannot_run():
    """Errors of the one-call cycle reach the caller as SolverException(:cpp) with the tile named, and leave the group usable."""
    import ctypes as C
    import armon_amd
    from armon_amd import _lib
    from armon_amd.multi_tile import TileGroup, cycle_plan, tile_cycle_descs
    group = TileGroup((2, 1), test="Sod", N=(64, 24), silent=5, maxcycle=4)
    try:
        group.init_test()
        group.global_dt.reset()
        plan, _ = cycle_plan(group.root, group.global_dt, last_cycle=False)
        plan.n_sweeps = 5
        with pytest.raises(_lib.SolverException, match="1 to 3 sweeps"):
            _lib.check(group._fn("mgpu_cycle")(group.handle, C.byref(plan), group._tile_cycles()))
        tcs = tile_cycle_descs(group.params, group.grids)
        tcs[1].x.rho_out = tcs[1].x.rho_in                      # in == out: the sweep of tile 1 refuses it, from its own thread
        plan, _ = cycle_plan(group.root, group.global_dt, last_cycle=False)
        plan.dt[0] = plan.dt[1] = 1e-4
        with pytest.raises(_lib.SolverException, match="tile 1"):
            _lib.check(group._fn("mgpu_cycle")(group.handle, C.byref(plan), tcs))
        stats = group.run()                                     # and the group still runs
        assert stats.cycles == 4
    finally:
        group.close()
