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
    o = dict(maxcycle=15, use_fused_sweep=fused, exact_arithmetic=True)
    spawn(dist_workers.gpu_solver_worker, 1, (1, 1), N, test, o, str(tmp_path), "nccl")
    ref = armon_amd.armon(armon_amd.ArmonParameters(test=test, N=N, silent=5, return_data=True, **o))
    host = ref.data.device_to_host(("rho", "u", "v", "E", "p"))
    t = np.load(tmp_path / "tile0.npz")
    assert int(t["cycles"]) == ref.cycles and float(t["dt"]) == ref.last_dt and float(t["time"]) == ref.final_time
    for k in ("rho", "u", "v", "E", "p"):
        assert np.array_equal(t[k], ref.data.real_view(host[k])), k
