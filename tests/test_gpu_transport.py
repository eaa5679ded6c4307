"""The two transport calls of the multi-GPU path, executed on ONE GPU in degenerate form:

 * ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd of armon_hip_halo_exchange_start in rank mode — a periodic 1 x 1
   process grid (test aid armon_hip_mgpu_set_periodic) makes the rank its own neighbour, RCCL moves every face, the second
   communicator's all-reduce runs on the compute stream beside it;
 * hipMemcpyPeerAsync of the in-process group — forced (armon_hip_mgpu_force_peer_copy) although the tiles share a device,
   for the faces and for the several-device form of the dt reduction.

Replaces ref src/halo_exchange.jl:229-283 (MPI.Start / MPI.Wait of the persistent requests) and src/solver_state.jl:89-111.
What a multi-GPU node adds to these is a peer that is another device; the calls, their stream ordering and the buffer
protocol are the ones run here."""
import os

import numpy as np
import pytest

import dist_workers
from test_distributed import spawn

pytestmark = pytest.mark.gpu


def _run_single(test, N, **o):
    import armon_amd
    ref = armon_amd.armon(armon_amd.ArmonParameters(test=test, N=N, silent=5, return_data=True, **o))
    host = ref.data.device_to_host(("rho", "u", "v", "E", "p"))
    return ref, {k: ref.data.real_view(v) for k, v in host.items()}


@pytest.mark.parametrize("periodic", [(True, True), (True, False), (False, True)])
@pytest.mark.parametrize("N,opts", [((41, 27), {}), ((64, 24), dict(data_type="float32"))])
def test_rccl_send_recv_to_self_moves_index_encoded_faces(tmp_path, periodic, N, opts):
    """ref test/mpi.jl:272-360 with wrap-around neighbours: after the exchange every ghost cell of a periodic side holds the
    global index of the OPPOSITE border's cell, physical sides and corners keep their marker; three chaos seeds; the dt
    all-reduce (second communicator, compute stream) and the host-value all-reduce give the right values meanwhile."""
    spawn(dist_workers.rccl_periodic_worker, 1, "index", N, "Sod", dict(opts, periodic=periodic), str(tmp_path))
    lines = open(tmp_path / "rank0.txt").read().splitlines()
    assert lines[0] == "OK", lines
    nb = dict(eval(lines[1]))
    assert [nb[s] for s in (1, 2, 3, 4)] == [0 if periodic[0] else -1] * 2 + [0 if periodic[1] else -1] * 2


@pytest.mark.parametrize("fused", [False, True], ids=["staged", "fused"])
@pytest.mark.parametrize("test,periodic,opts", [
    ("Sod_y", (True, False), dict(maxcycle=12)),                         # invariant along x: periodic in x == mirror
    ("Sod", (False, True), dict(maxcycle=12)),                           # invariant along y
    ("Sod_y", (True, False), dict(maxcycle=10, axis_splitting="Strang")),
])
def test_run_over_rccl_self_exchange_equals_plain_run(tmp_path, test, periodic, opts, fused):
    """A problem that does not vary along an axis has the same ghost values whether that axis is mirrored (FreeFlow) or
    periodic — bit for bit. So a whole run whose X (or Y) halos travel through ncclSend / ncclRecv to the rank itself, with
    the interior / strip split, the edge stream and the RCCL dt all-reduce, must equal the plain single-block run."""
    N = (96, 72)
    o = dict(opts, use_fused_sweep=fused, exact_arithmetic=True)
    spawn(dist_workers.rccl_periodic_worker, 1, "run", N, test, dict(o, periodic=periodic), str(tmp_path))
    t = np.load(tmp_path / "tile0.npz")
    ref, full = _run_single(test, N, **o)
    assert int(t["cycles"]) == ref.cycles and float(t["dt"]) == ref.last_dt and float(t["time"]) == ref.final_time
    for k in ("rho", "u", "v", "E", "p"):
        assert np.array_equal(t[k], full[k]), k


@pytest.mark.parametrize("test,N,opts", [
    ("Sod_circ", (96, 80), dict(maxcycle=12)),
    ("Bizarrium", (64, 48), dict(maxcycle=10)),
    ("Sedov", (60, 60), dict(maxcycle=10, axis_splitting="Godunov")),
])
def test_doubly_periodic_run_is_the_same_through_every_transport(tmp_path, oracle, test, N, opts):
    """A doubly periodic problem has no plain counterpart in the library, so (1) the transports are held against each other:
    the 1 x 1 rank over RCCL (send/recv to itself), the in-process 1 x 1 and 2 x 2 groups with plain device copies, and the
    2 x 1 / 2 x 2 groups with forced hipMemcpyPeerAsync (faces AND the several-device dt reduction) — same bits, dt and cycle
    count; and (2) all of them against the CPU oracle with periodic ghosts (a test aid of the oracle too): bit for bit."""
    from armon_amd.multi_tile import TileGroup
    o = dict(opts, use_fused_sweep=True, exact_arithmetic=True)
    spawn(dist_workers.rccl_periodic_worker, 1, "run", N, test, dict(o, periodic=(True, True)), str(tmp_path))
    t = np.load(tmp_path / "tile0.npz")
    orun, f = oracle.solve(test=test, N=N, periodic=(True, True), **opts)
    assert int(t["cycles"]) == orun.cycles and float(t["dt"]) == orun.last_dt and float(t["time"]) == orun.final_time
    for k in ("rho", "u", "v", "E", "p"):
        assert np.array_equal(t[k], oracle.real_view(f[k], N[0], N[1], 4)), k
    for P, force in (((1, 1), False), ((2, 2), False), ((2, 1), True), ((2, 2), True)):
        group = TileGroup(P, test=test, N=N, silent=5, periodic=(True, True), force_peer_copy=force, **o)
        try:
            stats = group.run()
            got = group.gather()
            assert stats.cycles == int(t["cycles"]) and stats.last_dt == float(t["dt"]) and stats.final_time == float(t["time"]), (P, force)
            for k in ("rho", "u", "v", "E", "p"):
                assert np.array_equal(got[k], t[k]), (P, force, k)
        finally:
            group.close()


def _random_periodic_cases(seed, count):
    import random
    rng = random.Random(seed)
    cases = []
    for _ in range(count):
        scheme = rng.choice(["GAD", "GAD", "Godunov"])
        projection = rng.choice(["euler_2nd", "euler_2nd", "euler"])
        lag = 2 + (scheme == "GAD") + (projection == "euler_2nd")
        P = rng.choice([(1, 1), (2, 1), (1, 2), (2, 2), (3, 2)])
        N = (rng.randint(P[0] * 8, P[0] * 60), rng.randint(P[1] * 8, P[1] * 50))
        cases.append(dict(P=P, N=N, test=rng.choice(["Sod_circ", "Sedov", "Bizarrium"]), periodic=rng.choice([(True, True), (True, False), (False, True)]),
                          force=rng.random() < 0.6, fused=rng.random() < 0.75,
                          opts=dict(scheme=scheme, projection=projection, nghost=max(lag, rng.choice([lag, 4, 6])), maxcycle=rng.randint(3, 7),
                                    axis_splitting=rng.choice(["Sequential", "Godunov", "Strang"]))))
    return cases


@pytest.mark.parametrize("case", _random_periodic_cases(int(os.environ.get("ARMON_RANDOM_SEED", "2718")), int(os.environ.get("ARMON_RANDOM_CASES", "8"))),
                         ids=lambda c: f"{c['P'][0]}x{c['P'][1]}-{c['test']}-{c['N'][0]}x{c['N'][1]}-per{int(c['periodic'][0])}{int(c['periodic'][1])}-{'peer' if c['force'] else 'direct'}-{'fused' if c['fused'] else 'staged'}")
def test_random_periodic_runs_through_rccl_and_peer_copies(tmp_path, oracle, case):
    """Drawn shapes, ghost widths, options and periodic axes: the 1 x 1 rank over RCCL (ncclSend / ncclRecv to itself on the
    periodic axes) and a drawn in-process tile group (plain or forced peer copies) must both give the periodic oracle's bits."""
    from armon_amd.multi_tile import TileGroup
    N, test, periodic = case["N"], case["test"], case["periodic"]
    o = dict(case["opts"], use_fused_sweep=case["fused"], exact_arithmetic=True)
    g = o["nghost"]
    orun, f = oracle.solve(test=test, N=N, periodic=periodic, **case["opts"])
    ref = {k: oracle.real_view(f[k], N[0], N[1], g) for k in ("rho", "u", "v", "E", "p")}
    spawn(dist_workers.rccl_periodic_worker, 1, "run", N, test, dict(o, periodic=periodic), str(tmp_path))
    t = np.load(tmp_path / "tile0.npz")
    assert int(t["cycles"]) == orun.cycles and float(t["dt"]) == orun.last_dt and float(t["time"]) == orun.final_time
    for k in ref:
        assert np.array_equal(t[k], ref[k]), ("rccl", k)
    group = TileGroup(case["P"], test=test, N=N, silent=5, periodic=periodic, force_peer_copy=case["force"], **o)
    try:
        stats = group.run()
        got = group.gather()
        assert stats.cycles == orun.cycles and stats.last_dt == orun.last_dt and stats.final_time == orun.final_time
        for k in ref:
            assert np.array_equal(got[k], ref[k]), ("group", k)
    finally:
        group.close()


@pytest.mark.parametrize("fused", [False, True], ids=["staged", "fused"])
@pytest.mark.parametrize("P,N,test,opts", [
    ((2, 2), (48, 40), "Sod_circ", dict(maxcycle=12)),
    ((4, 2), (96, 48), "Bizarrium", dict(maxcycle=10)),                 # BASELINE config 5's layout
    ((3, 3), (61, 59), "Sod_circ", dict(maxcycle=8, axis_splitting="Strang")),
])
def test_forced_peer_copies_equal_one_block(P, N, test, opts, fused):
    """hipMemcpyPeerAsync(dst, dev0, src, dev0) for every face, gather / fold / scatter with peer copies for dt."""
    from armon_amd.multi_tile import TileGroup
    o = dict(opts, use_fused_sweep=fused, exact_arithmetic=True)
    ref, full = _run_single(test, N, **o)
    group = TileGroup(P, test=test, N=N, silent=5, force_peer_copy=True, **o)
    try:
        stats = group.run()
        assert stats.cycles == ref.cycles and stats.last_dt == ref.last_dt and stats.final_time == ref.final_time
        got = group.gather()
        for k in ("rho", "u", "v", "E", "p"):
            assert np.array_equal(got[k], full[k]), k
    finally:
        group.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_forced_peer_copies_survive_injected_delays(seed):
    import armon_amd  # noqa: F401
    from armon_amd.multi_tile import TileGroup
    test, N, o = "Sod_circ", (96, 80), dict(maxcycle=12)
    ref, full = _run_single(test, N, **o)
    group = TileGroup((2, 2), test=test, N=N, silent=5, force_peer_copy=True, **o)
    try:
        group.set_chaos(300, seed)
        stats = group.run()
        assert stats.cycles == ref.cycles and stats.last_dt == ref.last_dt
        got = group.gather()
        for k in ("rho", "u", "v", "E", "p"):
            assert np.array_equal(got[k], full[k]), k
    finally:
        group.close()


@pytest.mark.parametrize("P,periodic,force", [((1, 1), (True, True), True), ((2, 1), (True, False), True), ((3, 2), (True, True), False)])
def test_in_process_periodic_exchange_moves_index_encoded_faces(P, periodic, force):
    """The same index-encoded exchange through the in-process group, periodic, with (forced) peer copies."""
    from armon_amd.blocking import Axis, Side
    from armon_amd.multi_tile import TileGroup
    from armon_amd.parameters import PROC_NULL
    NG, names = (41, 27), ("rho", "u", "v", "E", "p", "c", "g")
    group = TileGroup(P, test="Sod", N=NG, silent=5, periodic=periodic, force_peer_copy=force)
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
        for subset in (names[:4], names):
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
                enc = vi * 1e7 + ((gy0 + jj) % NG[1]) * NG[0] + ((gx0 + ii) % NG[0])
                inside_x, inside_y = (ii >= 0) & (ii < nx), (jj >= 0) & (jj < ny)
                expected[inside_x & inside_y] = enc[inside_x & inside_y]
                for side, mask in ((Side.Left, inside_y & (ii < 0)), (Side.Right, inside_y & (ii >= nx)),
                                   (Side.Bottom, inside_x & (jj < 0)), (Side.Top, inside_x & (jj >= ny))):
                    if p.neighbours[side] != PROC_NULL:
                        expected[mask] = enc[mask]
                assert np.array_equal(a, expected), (p.rank, k)
    finally:
        group.close()


def test_failed_exchange_start_leaves_the_group_usable():
    """a start that is refused (tiles disagreeing on their common face) must not leave sides marked in flight."""
    import ctypes as C
    import armon_amd
    from armon_amd import _lib
    from armon_amd.blocking import Axis
    from armon_amd.multi_tile import TileGroup, _halo_descs
    from armon_amd.solver import STATE_VARS
    group = TileGroup((2, 1), test="Sod", N=(40, 24), silent=5)
    try:
        group.init_test()
        descs = _halo_descs(group.params, group.grids, STATE_VARS)
        descs[1].ny = 20                                   # tile 1 now claims a shorter common face
        rc = _lib.lib().armon_hip_halo_exchange_start(group.handle, 0, descs)
        assert rc != 0 and b"disagree" in _lib.lib().armon_hip_last_error()
        group.exchange_start(Axis.X, STATE_VARS)           # not "already started"
        group.exchange_finish(Axis.X, STATE_VARS)
        group.wait()
    finally:
        group.close()
