"""Process-boundary halo exchange and global reductions for the tile-decomposed grid.

Replaces the reference's MPI path (ref src/halo_exchange.jl:187-368: ``pack_to_array!`` → persistent
``MPI.Send_init/Recv_init`` started per sweep → ``unpack_from_array!``; ref src/solver_state.jl:107-111,
src/utils.jl:126-143: ``MPI_Iallreduce(MIN)`` on dt; ref src/reductions.jl:317-320: ``Allreduce(SUM)``) with
``torch.distributed``: one process per GPU, RCCL point-to-point over xGMI (backend "nccl") on device-resident
buffers, or "gloo" with host staging (the reference's ``gpu_aware=false`` branch, :234-237,277-280) — the
latter is what the CPU tests and the single-GPU rehearsal use.

Semantics kept from the reference: per sweep only the two sides ALONG the sweep axis are exchanged
(:323-354), 4 ghost layers, no corners; the buffer layout is ``pack_to_array!``'s
((i_g·face + i)·nvars + v); a side whose neighbour is PROC_NULL gets the physical boundary condition
instead. Each rank talks to at most two peers per sweep; all sends and receives of a sweep are posted as one
batch (``batch_isend_irecv`` = one ncclGroupStart/End).
"""
import ctypes as C

import numpy as np

from ._lib import check
from .blocking import Side
from .parameters import PROC_NULL


class HaloExchanger:
    """Owns the send/recv buffers of one block and runs pack → exchange → unpack.

    ``pack``/``unpack`` are the device kernels by default; tests may substitute host implementations
    (``host_arrays`` = dict name → numpy array) to exercise the protocol without a GPU."""

    def __init__(self, params, grid=None, host_arrays=None, max_vars=7):
        import torch
        import torch.distributed as dist
        if any(getattr(params, "periodic", (False, False))):
            # a periodic grid can make both sides of an axis talk to the SAME peer: batch_isend_irecv then depends on
            # message order, which only the library's exchange arranges (multi_gpu.hip, exchange_start)
            from ._lib import solver_error
            solver_error("config", "periodic process grids (a test aid) need the native halo exchange")
        self.params = params
        self.grid = grid
        self.host_arrays = host_arrays
        self.dist = dist
        self.torch = torch
        self.group = params.global_comm
        backend = dist.get_backend(self.group)
        self.device_buffers = backend == "nccl" and host_arrays is None
        if self.device_buffers:
            params.device                      # make sure the context exists (it may adopt a torch stream)
        # RCCL + every kernel on torch's current stream: pack → send and recv → unpack are ordered on the device
        self.stream_ordered = self.device_buffers and getattr(params, "shared_stream", False)
        bs = params.block_size
        self.buf = {}
        for side in Side:
            if params.neighbours[side] == PROC_NULL:
                continue
            n = bs.real_face_size(side) * bs.ghosts * max_vars       # ref src/blocking/block_grid.jl:134-138
            tdt = torch.float32 if params.data_type == np.float32 else torch.float64
            self.tdt = tdt
            if host_arrays is not None:
                mk = lambda: torch.zeros(n, dtype=tdt)
            else:
                dev = torch.device("cuda", params.device_id)
                mk = lambda: torch.zeros(n, dtype=tdt, device=dev)
            self.buf[side] = (mk(), mk())                            # (send, recv)

    # -- pack / unpack ---------------------------------------------------------------------------------
    def _vars_ptrs(self, names):
        arr = (C.c_void_p * len(names))(*(self.grid.data[k].ptr for k in names))
        return arr

    def pack(self, side, names):
        p, bs = self.params, self.params.block_size
        send = self.buf[side][0]
        dom = bs.border_domain(side, single_strip=False).to_c()
        face = bs.real_face_size(side)
        if self.host_arrays is not None:
            _host_pack(dom, bs.ghosts, face, send.numpy(), [self.host_arrays[k] for k in names], True)
            return
        check(p.fn("pack_to_array")(p.device.ctx, dom, bs.ghosts, face, C.c_void_p(send.data_ptr()),
                                                 len(names), self._vars_ptrs(names)))

    def unpack(self, side, names):
        p, bs = self.params, self.params.block_size
        recv = self.buf[side][1]
        dom = bs.ghost_domain(side, single_strip=False).to_c()
        face = bs.real_face_size(side)
        if self.host_arrays is not None:
            _host_pack(dom, bs.ghosts, face, recv.numpy(), [self.host_arrays[k] for k in names], False)
            return
        check(p.fn("unpack_from_array")(p.device.ctx, dom, bs.ghosts, face, C.c_void_p(recv.data_ptr()),
                                                     len(names), self._vars_ptrs(names)))

    # -- exchange ---------------------------------------------------------------------------------------
    def start(self, sides, names):
        """start_exchange of the reference (ref src/halo_exchange.jl:229-250) for every remote side in
        ``sides``: pack, wait for the pack, post all sends and receives as one batch. Returns a handle for
        ``finish``; device work enqueued in between overlaps with the transfers."""
        dist, torch, p = self.dist, self.torch, self.params
        sides = [s for s in sides if p.neighbours[s] != PROC_NULL]
        if not sides:
            return None
        bs = p.block_size
        for s in sides:
            self.pack(s, names)
        if self.host_arrays is None and not self.stream_ordered:
            p.wait()                                  # ref src/halo_exchange.jl:244: wait before the sends
        ops, staged = [], {}
        for s in sides:
            n = bs.real_face_size(s) * bs.ghosts * len(names)
            send, recv = self.buf[s][0][:n], self.buf[s][1][:n]
            if not self.device_buffers and self.host_arrays is None:
                staged[s] = (send.cpu(), torch.empty(n, dtype=self.tdt))   # host staging (gpu_aware=false)
                send, recv = staged[s]
            peer = p.neighbours[s]
            ops.append(dist.P2POp(dist.isend, send, peer, self.group, tag=int(s)))
            ops.append(dist.P2POp(dist.irecv, recv, peer, self.group, tag=int(_opposite(s))))
        return sides, names, staged, dist.batch_isend_irecv(ops)

    def finish(self, handle):
        """finish_exchange (ref src/halo_exchange.jl:264-283): wait for the transfers, unpack into the ghosts."""
        if handle is None:
            return
        torch, p = self.torch, self.params
        sides, names, staged, reqs = handle
        for req in reqs:
            req.wait()
        for s in sides:
            if s in staged:
                n = staged[s][1].numel()
                self.buf[s][1][:n].copy_(staged[s][1])
        if self.host_arrays is None and not self.stream_ordered:
            torch.cuda.synchronize(p.device_id)       # receives landed before the unpack kernels read them
        # (stream-ordered: req.wait() made the current stream — the kernels' stream — wait for the transfers)
        for s in sides:
            self.unpack(s, names)

    def allreduce_min_device_async(self, scalar):
        """MIN over the ranks of element 0 of a device vector, in place (RCCL), ordered on the kernels' stream:
        nothing is waited for on the host."""
        torch, dist = self.torch, self.dist
        if getattr(self, "_scalar_view", None) is None or self._scalar_view[0] != scalar.ptr:
            self._scalar_view = (scalar.ptr, torch.as_tensor(scalar, device=torch.device("cuda", self.params.device_id)))
        work = dist.all_reduce(self._scalar_view[1][:1], op=dist.ReduceOp.MIN, group=self.group, async_op=True)
        work.wait()                # RCCL: makes the current stream (= the kernels' stream) wait, not the host

    def exchange(self, sides, names):
        self.finish(self.start(sides, names))


def _opposite(side):
    return {Side.Left: Side.Right, Side.Right: Side.Left, Side.Bottom: Side.Top, Side.Top: Side.Bottom}[side]


def _host_pack(dom, nghost, face, array, arrays, pack):
    """pack_to_array!/unpack_from_array! on the host (ref src/halo_exchange.jl:187-216) — tests only."""
    nv = len(arrays)
    for j in range(dom.col_len):
        for k in range(dom.row_len):
            idx = dom.col_start + j * dom.col_step + dom.row_start + k
            itr = j * dom.row_len + k
            i, i_g = divmod(itr, nghost)
            base = (i_g * face + i) * nv
            for v in range(nv):
                if pack:
                    array[base + v] = arrays[v][idx]
                else:
                    arrays[v][idx] = array[base + v]


# ---- module-level API used by solver.py ---------------------------------------------------------------------
def setup(params, grid, native=None):
    """Create the exchanger of this rank's tile. Over RCCL (backend "nccl") the library's own multi-GPU entry points
    are used when ``params.native_halo`` (``armon_hip_mgpu_init_rank``: RCCL send/recv on a transfer stream, ordered
    by events only); ``native=False`` or a failed native initialisation on any rank selects the torch.distributed
    exchanger on every rank."""
    import torch
    import torch.distributed as dist
    from ._lib import SolverException
    want = params.native_halo if native is None else native
    if want and dist.get_backend(params.global_comm) == "nccl":
        from .multi_tile import NativeRcclExchanger
        comm, ok = None, 1.0
        try:
            comm = NativeRcclExchanger(params, grid)      # agrees on local readiness itself before its collective part
        except Exception as e:           # whatever went wrong locally, every rank must reach the all_reduce below
            ok = 0.0
            params.native_halo_error = f"{type(e).__name__}: {e}"       # bench.py reports it
        flag = torch.tensor([ok], device=torch.device("cuda", params.device_id))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=params.global_comm)
        if float(flag.item()) == 1.0:
            grid.comm = comm
            return grid.comm
        if comm is not None:
            comm.close()
    grid.comm = HaloExchanger(params, grid)
    return grid.comm


def exchange_sides(params, grid, axis, sides, names):
    """Staged path: the reference's comm_vars (ρ,u,v,E,p,c,g) into the ghosts of the remote sides."""
    grid.comm.exchange(sides, names)


def _allreduce(params, values, op):
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", params.device_id) if dist.get_backend(params.global_comm) == "nccl" else "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=op, group=params.global_comm)
    return [float(x) for x in t.cpu()]


def allreduce_min(params, value):
    import torch.distributed as dist
    return _allreduce(params, (value,), dist.ReduceOp.MIN)[0]


def allreduce_sum(params, values):
    import torch.distributed as dist
    return tuple(_allreduce(params, values, dist.ReduceOp.SUM))
