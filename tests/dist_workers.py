"""Worker functions for the multi-process tests (run under torch.multiprocessing.spawn)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _init(rank, world, port, backend="gloo"):
    """`port` = the rendezvous token of test_distributed.free_port(): the path of a file store (no TCP port to collide on)."""
    import torch.distributed as dist
    os.environ["RANK"] = str(rank)
    os.environ["WORLD_SIZE"] = str(world)
    os.environ["LOCAL_RANK"] = "0"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")      # RCCL's own bootstrap still wants an address
    method = "file://" + str(port)
    if backend == "nccl":
        import torch
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method=method, rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", init_method=method, rank=rank, world_size=world)
    return dist


def halo_index_worker(rank, world, port, P, N_global, out_dir):
    """Index-encoded halo exchange on the host (the design of ref test/mpi.jl:272-360): every cell of every
    variable holds var·1e7 + its GLOBAL linear index; after the exchange each remote ghost cell must hold the
    index of the neighbour's real cell it mirrors, physical-side ghosts stay untouched."""
    import numpy as np
    dist = _init(rank, world, port)
    import armon_amd
    from armon_amd.blocking import Axis, Side, sides_along
    from armon_amd.halo_exchange import HaloExchanger, allreduce_min, allreduce_sum
    from armon_amd.parameters import PROC_NULL

    params = armon_amd.ArmonParameters(test="Sod", N=N_global, use_MPI=True, P=P)
    bs, g = params.block_size, params.nghost
    nx, ny = params.N
    names = ("rho", "u", "v", "E", "p", "c", "g")
    gx0, gy0 = params.N_origin[0] - 1, params.N_origin[1] - 1          # 0-based global position of the tile

    def encode(vi):
        a = np.full((ny + 2 * g, nx + 2 * g), -1.0)
        iy, ix = np.mgrid[0:ny, 0:nx]
        a[g:g + ny, g:g + nx] = vi * 1e7 + (gy0 + iy) * N_global[0] + (gx0 + ix)
        return a.ravel()

    arrays = {k: encode(vi) for vi, k in enumerate(names)}
    ex = HaloExchanger(params, host_arrays=arrays)
    errors = []
    for axis in (Axis.X, Axis.Y):
        ex.exchange(sides_along(axis), names)
    for vi, k in enumerate(names):
        a = arrays[k].reshape(ny + 2 * g, nx + 2 * g)
        for j in range(-g, ny + g):
            for i in range(-g, nx + g):
                inside_x, inside_y = 0 <= i < nx, 0 <= j < ny
                if inside_x and inside_y:
                    continue
                val = a[j + g, i + g]
                side = None
                if inside_y and i < 0: side = Side.Left
                elif inside_y and i >= nx: side = Side.Right
                elif inside_x and j < 0: side = Side.Bottom
                elif inside_x and j >= ny: side = Side.Top
                if side is not None and params.neighbours[side] != PROC_NULL:
                    expected = vi * 1e7 + (gy0 + j) * N_global[0] + (gx0 + i)
                else:
                    expected = -1.0          # corners and physical sides are never written by the exchange
                if val != expected:
                    errors.append((k, i, j, val, expected))
    mn = allreduce_min(params, float(rank + 1))
    sm = allreduce_sum(params, (1.0, float(rank)))
    ok = not errors and mn == 1.0 and sm == (float(world), float(sum(range(world))))
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write("OK\n" if ok else f"FAIL {errors[:5]} {mn} {sm}\n")
        f.write(f"{params.N} {params.N_origin} {params.cart_coords} {sorted((int(s), n) for s, n in params.neighbours.items())}\n")
    dist.destroy_process_group()


def gpu_solver_worker(rank, world, port, P, N_global, test, opts, out_dir, backend="gloo"):
    """Tile-decomposed run on the GPU (all ranks share cuda:0, gloo transport with host staging): every rank
    saves its tile so that the parent can compare with the single-process result."""
    import numpy as np
    dist = _init(rank, world, port, backend)
    import torch  # noqa: F401  (torch's HIP runtime must be the one the process uses)
    import armon_amd
    params = armon_amd.ArmonParameters(test=test, N=N_global, use_MPI=True, P=P, device_id=0, silent=5,
                                       return_data=True, **opts)
    stats = armon_amd.armon(params)
    if backend == "nccl":      # the RCCL configuration: kernels on the adopted torch stream, stream-ordered exchange
        assert params.shared_stream and stats.data.comm.stream_ordered
    host = stats.data.device_to_host(("rho", "u", "v", "E", "p"))
    np.savez(os.path.join(out_dir, f"tile{rank}.npz"), cycles=stats.cycles, dt=stats.last_dt, time=stats.final_time,
             native=bool(getattr(stats.data.comm, "native", False)),
             origin=np.array(params.N_origin), n=np.array(params.N),
             **{k: stats.data.real_view(v) for k, v in host.items()})
    dist.destroy_process_group()


def rccl_periodic_worker(rank, world, port, mode, N, test, opts, out_dir):
    """ONE rank over RCCL on a PERIODIC 1 x 1 process grid (test aid, armon_hip_mgpu_set_periodic): the rank is its own
    neighbour on the periodic sides, so exchange_start really issues ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on
    the transfer stream (multi_gpu.hip) — the call a multi-GPU run makes, with the one peer a single GPU has.
    mode "index": the index-encoded exchange of ref test/mpi.jl:272-360, with wrapped expectations, also under injected
    delays and with the dt all-reduce of the second communicator running beside the faces; mode "run": a whole solve."""
    import ctypes as C
    import numpy as np
    dist = _init(rank, world, port, "nccl")
    import torch  # noqa: F401
    import armon_amd
    from armon_amd.blocking import Axis, Side, sides_along
    from armon_amd.halo_exchange import setup
    from armon_amd.parameters import PROC_NULL
    from armon_amd.solver import BlockGrid
    periodic = tuple(opts.pop("periodic"))
    if mode == "run":
        params = armon_amd.ArmonParameters(test=test, N=N, use_MPI=True, P=(1, 1), periodic=periodic, device_id=0, silent=5,
                                           return_data=True, **opts)
        stats = armon_amd.armon(params)
        assert getattr(stats.data.comm, "native", False), "the library's RCCL exchange was not selected"
        host = stats.data.device_to_host(("rho", "u", "v", "E", "p"))
        np.savez(os.path.join(out_dir, "tile0.npz"), cycles=stats.cycles, dt=stats.last_dt, time=stats.final_time,
                 **{k: stats.data.real_view(v) for k, v in host.items()})
        stats.data.comm.close()
        dist.destroy_process_group()
        return
    params = armon_amd.ArmonParameters(test="Sod", N=N, use_MPI=True, P=(1, 1), periodic=periodic, device_id=0, silent=5, **opts)
    grid = BlockGrid(params)
    comm = setup(params, grid)
    assert getattr(comm, "native", False), "the library's RCCL exchange was not selected"
    g = params.nghost
    nx, ny = params.N
    names = ("rho", "u", "v", "E", "p", "c", "g")
    jj, ii = np.mgrid[-g:ny + g, -g:nx + g]
    inside_x, inside_y = (ii >= 0) & (ii < nx), (jj >= 0) & (jj < ny)
    dtype = params.data_type
    errors = []
    for seed in (0, 1, 2, 3):                       # 0: no injected delay
        comm.set_chaos(300 if seed else 0, seed)
        for vi, k in enumerate(names):
            a = np.full((ny + 2 * g, nx + 2 * g), -1.0, dtype=dtype)
            a[inside_x & inside_y] = (vi * 1e5 + jj * nx + ii)[inside_x & inside_y]
            grid.data[k].copy_from_host(a.ravel())
        scalar = params.device.zeros(2, dtype)
        scalar.copy_from_host(np.array([3.5 + seed, 0.], dtype=dtype))
        for subset in (names[:4], names):
            for axis in (Axis.X, Axis.Y):
                h = comm.start(sides_along(axis), subset)
                comm.allreduce_min_device_async(scalar)      # comm_red on the compute stream while comm_halo moves the faces
                comm.finish(h)
        params.wait()
        if float(scalar.to_host()[0]) != 3.5 + seed:
            errors.append(("allreduce", seed, float(scalar.to_host()[0])))
        for vi, k in enumerate(names):
            a = grid.data[k].to_host().reshape(ny + 2 * g, nx + 2 * g)
            expected = np.full_like(a, -1.0)
            enc = (vi * 1e5 + (jj % ny) * nx + (ii % nx)).astype(dtype)          # the opposite border's global indices
            expected[inside_x & inside_y] = enc[inside_x & inside_y]
            for side, mask in ((Side.Left, inside_y & (ii < 0)), (Side.Right, inside_y & (ii >= nx)),
                               (Side.Bottom, inside_x & (jj < 0)), (Side.Top, inside_x & (jj >= ny))):
                if params.neighbours[side] != PROC_NULL:
                    expected[mask] = enc[mask]
            if not np.array_equal(a, expected):
                errors.append((k, seed, int((a != expected).sum())))
        sums = comm.allreduce_host([1.0, 2.0], "sum")
        if sums != [1.0, 2.0]:
            errors.append(("allreduce_host", sums))
        # a NaN time step must survive the RCCL minimum (ncclMin may drop a NaN operand): it travels as -inf, which fails
        # the host's validity check just as well (ref src/solver_state.jl:123-124)
        scalar.copy_from_host(np.array([np.nan, 0.], dtype=dtype))
        comm.allreduce_min_device_async(scalar)
        params.wait()
        got = float(scalar.to_host()[0])
        if not (got == -np.inf or np.isnan(got)):
            errors.append(("nan through the dt all-reduce", got))
    with open(os.path.join(out_dir, "rank0.txt"), "w") as f:
        f.write("OK\n" if not errors else f"FAIL {errors[:8]}\n")
        f.write(f"{sorted((int(s), n) for s, n in params.neighbours.items())}\n")
    comm.close()
    dist.destroy_process_group()
