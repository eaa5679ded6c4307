"""Multi-process tests of the N>1 path on CPU: world_size 2 over gloo (ref test/mpi.jl design)."""
import socket

import pytest

import dist_workers


def free_port():
    """A rendezvous token for one spawn: the workers meet through a FILE store (tests/dist_workers._init), not a TCP port —
    a port found free here can be taken (or still be in TIME_WAIT) by the time the workers bind it, which made one run in
    six of the GPU suite fail with EADDRINUSE."""
    import os
    import tempfile
    fd, path = tempfile.mkstemp(prefix="armon_rdv_", dir="/tmp")
    os.close(fd)
    os.unlink(path)                      # the store creates it
    return path


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


@pytest.mark.parametrize("P,N", [((2, 2), (107, 113)), ((1, 3), (37, 241)), ((4, 2), (96, 48)), ((5, 1), (20, 20))])
def test_halo_exchange_reference_domains_and_the_8_gpu_layout(tmp_path, P, N):
    """The reference's uneven MPI test domains (ref test/mpi.jl:551-561: 107×113, 37×241, 20×20) and the 4×2 layout of
    BASELINE config 5, with the partition rule checked on every rank (N ÷ P, remainder on the last rank of the axis)."""
    world = P[0] * P[1]
    spawn(dist_workers.halo_index_worker, world, P, N, str(tmp_path))
    for r in range(world):
        lines = open(tmp_path / f"rank{r}.txt").read().splitlines()
        assert lines[0] == "OK", lines
        size = eval(lines[1].split(") (")[0] + ")")
        cx, cy = r // P[1], r % P[1]
        assert size[0] == N[0] // P[0] + (N[0] % P[0] if cx == P[0] - 1 else 0)
        assert size[1] == N[1] // P[1] + (N[1] % P[1] if cy == P[1] - 1 else 0)


def test_native_halo_ranges_match_the_reference_domains():
    """armon_hip_halo_ranges (what the library's own halo exchange packs and unpacks) against border_domain /
    ghost_domain of ref src/blocking/blocking.jl:141-187 as mirrored in blocking.py — no GPU needed."""
    import ctypes as C
    import armon_amd
    from armon_amd import _lib
    from armon_amd.blocking import BlockSize, Side
    L = armon_amd.lib()
    for nx, ny, g in [(24, 16, 4), (37, 41, 4), (5, 7, 2), (64, 64, 5), (107, 113, 4), (1, 9, 1)]:
        bs = BlockSize((nx + 2 * g, ny + 2 * g), g)
        for tag, side in enumerate((Side.Left, Side.Right, Side.Bottom, Side.Top)):
            b, gh, face = _lib.Range(), _lib.Range(), C.c_int64()
            assert L.armon_hip_halo_ranges(nx, ny, g, tag, C.byref(b), C.byref(gh), C.byref(face)) == 0
            for got, ref in ((b, bs.border_domain(side, single_strip=False).to_c()),
                             (gh, bs.ghost_domain(side, single_strip=False).to_c())):
                assert ((got.col_start, got.col_step, got.col_len, got.row_start, got.row_len)
                        == (ref.col_start, ref.col_step, ref.col_len, ref.row_start, ref.row_len)), (nx, ny, g, side)
            assert face.value == bs.real_face_size(side)
    assert L.armon_hip_halo_ranges(0, 4, 4, 0, None, None, None) != 0            # empty tile is an error, not a crash


def test_split_too_small_for_ghosts_is_rejected():
    """ref src/parameters.jl:684-690"""
    import armon_amd
    from armon_amd.parameters import ArmonParameters
    p = ArmonParameters.__new__(ArmonParameters)
    p.N, p.nghost, p.proc_dims, p.cart_coords, p.projection_scheme = (6, 40), 4, (2, 1), (0, 0), "euler_2nd"
    with pytest.raises(armon_amd.SolverException):
        p._init_indexing()
