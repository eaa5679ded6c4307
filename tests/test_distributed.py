"""Multi-process tests of the N>1 path on CPU: world_size 2 over gloo (ref test/mpi.jl design)."""
import os
import socket

import pytest

import dist_workers


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn(fn, world, *args):
    import torch.multiprocessing as mp
    mp.spawn(fn, args=(world, free_port()) + args, nprocs=world, join=True)


@pytest.mark.parametrize("P,N", [((2, 1), (24, 16)), ((1, 2), (24, 16)), ((2, 1), (37, 41)), ((1, 2), (20, 23))])
def test_halo_exchange_with_index_encoded_data(tmp_path, P, N):
    spawn(dist_workers.halo_index_worker, 2, P, N, str(tmp_path))
    for r in range(2):
        lines = open(tmp_path / f"rank{r}.txt").read().splitlines()
        assert lines[0] == "OK", lines
    # partition rule of ref src/parameters.jl:673-697: even split, remainder on the last rank of the axis
    d = 0 if P[0] == 2 else 1
    n0 = N[d] // 2
    sizes = [eval(open(tmp_path / f"rank{r}.txt").read().splitlines()[1].split(") (")[0] + ")") for r in range(2)]
    assert sizes[0][d] == n0 and sizes[1][d] == N[d] - n0 and sizes[0][1 - d] == N[1 - d]


@pytest.mark.parametrize("P,N", [((3, 1), (40, 12)), ((1, 3), (12, 40)), ((2, 2), (24, 20)), ((4, 1), (50, 9))])
def test_halo_exchange_more_ranks(tmp_path, P, N):
    """Middle tiles talk to two peers along an axis (as with px = 4 on 8 GPUs); 2×2 tiles have remote sides on both axes."""
    world = P[0] * P[1]
    spawn(dist_workers.halo_index_worker, world, P, N, str(tmp_path))
    for r in range(world):
        lines = open(tmp_path / f"rank{r}.txt").read().splitlines()
        assert lines[0] == "OK", lines


def test_split_too_small_for_ghosts_is_rejected():
    """ref src/parameters.jl:684-690"""
    import armon_amd
    from armon_amd.parameters import ArmonParameters
    p = ArmonParameters.__new__(ArmonParameters)
    p.N, p.nghost, p.proc_dims, p.cart_coords, p.projection_scheme = (6, 40), 4, (2, 1), (0, 0), "euler_2nd"
    with pytest.raises(armon_amd.SolverException):
        p._init_indexing()
