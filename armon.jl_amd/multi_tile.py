"""Tile-decomposed runs driven through the library's own multi-GPU entry points (include/armon_hip.h:
``armon_hip_mgpu_init`` / ``armon_hip_halo_exchange_start|finish`` / ``armon_hip_dt_allreduce``).

``TileGroup`` keeps every tile of a px × py decomposition in ONE process — the shape a Julia host without MPI, or
a one-GPU rehearsal of the 8-GPU layout, uses: tile r on device ``device_ids[r]`` (all on device 0 by default), each
with its own compute and transfer stream, faces moved by peer copies ordered by events only. The cycle below is the
reference's ``solver_cycle`` (ref src/solver.jl:288-320) with ``block_ghost_exchange`` (ref src/halo_exchange.jl:
286-368) replaced by the native exchange, the interior of each fused sweep enqueued between its start and its finish
(the overlap the reference gets from its asynchronous block state machine, ref src/solver.jl:58-285), and the dt
minimum reduced on the device (ref src/solver_state.jl:89-111). The host never synchronises inside a cycle.

``NativeRcclExchanger`` is the one-process-per-GPU counterpart (``armon_hip_mgpu_init_rank``): the same native
choreography over RCCL send/recv, plugged into ``solver.fused_sweep_overlapped`` in place of the torch.distributed
``HaloExchanger``.
"""
import ctypes as C
import time as _time

import numpy as np

from . import _lib
from ._lib import CyclePlan, HaloDesc, TileCycle, check
from .blocking import Axis, Side, first_side, last_side, sides_along
from .parameters import PROC_NULL, ArmonParameters
from . import solver as S

_SIDE_TAG = {Side.Left: 0, Side.Right: 1, Side.Bottom: 2, Side.Top: 3}     # ARMON_SIDE_*


def _halo_descs(params_list, grids, names):
    descs = (HaloDesc * len(grids))()
    for d, p, g in zip(descs, params_list, grids):
        d.nx, d.ny = p.N
        d.nghost, d.nvars = p.nghost, len(names)
        for k, f in enumerate(names):
            d.vars[k] = g.data[f].ptr
    return descs


def tile_cycle_descs(params_list, grids):
    """``armon_tile_cycle[n]``: per tile the descriptors of a full X and a full Y sweep from the set of vectors that holds the
    state NOW into its partner (armon_hip_mgpu_cycle fills in the steps, the sides and the cycle's outputs)."""
    tcs = (TileCycle * len(grids))()
    for tc, p, g in zip(tcs, params_list, grids):
        tc.x = S.sweep_desc(p, g, Axis.X, 0., p.cell_size(0), emit_p=True, emit_dt=True)
        tc.y = S.sweep_desc(p, g, Axis.Y, 0., p.cell_size(1), emit_p=True, emit_dt=True)
    return tcs


def cycle_plan(params, gdt, last_cycle, prefetch=True, dt_host=None):
    """``armon_cycle_plan`` of the cycle ``gdt`` is about to run (ref split_axes + update_solver_state!,
    src/axis_splitting.jl:24-46, src/solver_state.jl:339-345). ``dt_host``: pinned array of two elements, the landing zone
    of the next CFL step (slot = parity of the cycle; the event of that slot is recorded on tile 0's edge context)."""
    sweeps = S.split_axes(params.axis_splitting, gdt.cycle)
    plan = CyclePlan(n_sweeps=len(sweeps), emit_p=int(last_cycle), emit_dt=int(not params.cst_dt),
                     overlap=int(params.overlap_halo), next_axis=-1, event_slot=-1, dt_event_slot=-1)
    if dt_host is not None and not params.cst_dt:
        plan.dt_host = dt_host.ptr + (gdt.cycle & 1) * np.dtype(params.data_type).itemsize
        plan.dt_event_slot = S.DT_EVENT_SLOT + (gdt.cycle & 1)
    for k, (axis, dt_factor) in enumerate(sweeps):
        plan.axis[k] = int(axis) - 1
        plan.dt[k] = gdt.current_dt * params.T(dt_factor)
    if prefetch and not last_cycle and params.overlap_halo:
        plan.next_axis = int(S.split_axes(params.axis_splitting, gdt.cycle + 1)[0][0]) - 1
    names = ["sweep_x" if a == Axis.X else "sweep_y" for a, _ in sweeps]
    for cb in params.kernel_callbacks:        # timers that can hand out event slots (bench.py's EventTimer)
        if hasattr(cb, "reserve"):
            plan.event_slot = cb.reserve(names)
            plan.event_ctx = params.device.ctx      # the pool the timer reads (a rank's own context shares the tile's stream)
    return plan, len(sweeps)


def native_cycle_usable(params):
    """The native cycle has two forms of a sweep with remote sides: interior + edge stream (overlap), or in order."""
    return (params.use_fused_sweep and getattr(params, "native_cycle", True)
            and (not params.overlap_halo or params.edge_stream))


class TileGroup:
    """All tiles of a ``P = (px, py)`` decomposition of one problem, in this process."""

    def __init__(self, P, device_ids=None, force_peer_copy=False, **options):
        """``periodic=(x, y)`` and ``force_peer_copy`` are test aids for the transport (``armon_hip_mgpu_set_periodic``,
        ``armon_hip_mgpu_force_peer_copy``): wrap-around neighbours, and ``hipMemcpyPeerAsync`` for every face and dt
        scalar even though the tiles share a device."""
        L = _lib.lib()
        self.P = (int(P[0]), int(P[1]))
        nt = self.P[0] * self.P[1]
        ids = list(device_ids) if device_ids is not None else [0] * nt
        assert len(ids) == nt
        self.handle = C.c_void_p()
        check(L.armon_hip_mgpu_init(self.P[0], self.P[1], (C.c_int * nt)(*ids), C.byref(self.handle)))
        self._L = L
        periodic = tuple(bool(v) for v in options.get("periodic", (False, False)))
        if any(periodic):
            check(L.armon_hip_mgpu_set_periodic(self.handle, int(periodic[0]), int(periodic[1])))
        if force_peer_copy:
            check(L.armon_hip_mgpu_force_peer_copy(self.handle, 1))
        for k in ("use_MPI", "device_id", "P"):
            options.pop(k, None)
        self.params, self.grids, self._edge_ctx, self._edge_dt = [], [], [], []
        for r in range(nt):
            ctx = C.c_void_p(L.armon_hip_mgpu_ctx(self.handle, r))
            p = ArmonParameters(**options, tile_of=(r, self.P), ctx=ctx, device_id=ids[r])
            rank, coords, nb = C.c_int(), (C.c_int * 2)(), (C.c_int * 4)()
            check(L.armon_hip_mgpu_tile_info(self.handle, r, C.byref(rank), C.byref(coords), C.byref(nb)))
            # the library's topology is the reference's (and this package's) cartesian grid
            assert rank.value == r and tuple(coords) == p.cart_coords
            assert [p.neighbours[s] for s in (Side.Left, Side.Right, Side.Bottom, Side.Top)] == list(nb)
            self.params.append(p)
            self.grids.append(S.BlockGrid(p))
            self._edge_ctx.append(C.c_void_p(L.armon_hip_mgpu_edge_ctx(self.handle, r)))
            self._edge_dt.append(int(L.armon_hip_mgpu_edge_dt(self.handle, r)))
        self.root = self.params[0]
        self.global_dt = self.grids[0].global_dt
        for g in self.grids:
            g.global_dt = self.global_dt               # one clock for every tile (ref GlobalTimeStep is global)
        self.dt_host = None
        self.dt_inflight = {}
        self._tcs = {}              # tile-cycle descriptors per ping-pong parity (key: where tile 0's rho lives)

    def set_threads(self, on):
        """One host thread per tile inside ``armon_hip_mgpu_cycle`` (default) or everything from the calling thread."""
        check(self._L.armon_hip_mgpu_set_threads(self.handle, int(on)))

    def _tile_cycles(self):
        key = tuple(g.data["rho"].ptr for g in self.grids)
        if key not in self._tcs:
            self._tcs[key] = tile_cycle_descs(self.params, self.grids)
        return self._tcs[key]

    def drain(self):
        """Complete an exchange the last native cycle posted ahead and no cycle consumed."""
        if self.handle and native_cycle_usable(self.root):
            check(self._fn("mgpu_drain")(self.handle, self._tile_cycles()))

    def close(self):
        if self.handle:
            # the tile contexts belong to the group: release everything allocated through them first
            for g in self.grids:
                for a in list(g.data.values()) + (list(g.alt.values()) if g.alt else []) + [g.dt_scalar]:
                    a.free()
            if self.dt_host is not None:
                self.dt_host.free()
            self._L.armon_hip_mgpu_destroy(self.handle)
            self.handle = None
            for p in self.params:
                if p._device is not None:
                    p._device.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_chaos(self, max_delay_us, seed=0):
        """Test aid: random busy-wait kernels in front of the group's stream operations (armon_hip_mgpu_set_chaos)."""
        check(self._L.armon_hip_mgpu_set_chaos(self.handle, int(max_delay_us), int(seed)))

    def _fn(self, name):
        return getattr(self._L, "armon_hip_" + name + self.root.suffix)

    # ---- native exchange / reduction ----------------------------------------------------------------------------
    def exchange_start(self, axis, names):
        check(self._fn("halo_exchange_start")(self.handle, int(axis) - 1, _halo_descs(self.params, self.grids, names)))

    def exchange_finish(self, axis, names):
        check(self._fn("halo_exchange_finish")(self.handle, int(axis) - 1, _halo_descs(self.params, self.grids, names)))

    def exchange_finish_edge(self, axis, names):
        check(self._fn("halo_exchange_finish_edge")(self.handle, int(axis) - 1, _halo_descs(self.params, self.grids, names)))

    def edge_join(self, with_dt):
        ptrs = (C.c_void_p * len(self.grids))(*[g.dt_scalar.ptr for g in self.grids]) if with_dt else None
        check(self._fn("mgpu_edge_join")(self.handle, ptrs))

    def dt_allreduce(self):
        ptrs = (C.c_void_p * len(self.grids))(*[g.dt_scalar.ptr for g in self.grids])
        check(self._fn("dt_allreduce")(self.handle, ptrs))

    def wait(self):
        check(self._L.armon_hip_mgpu_sync(self.handle))       # compute and transfer streams of every tile

    # ---- solver ---------------------------------------------------------------------------------------------------
    def init_test(self):
        for p, g in zip(self.params, self.grids):
            S.init_test(p, g)

    def conservation_vars(self):
        mass = energy = 0.
        for p, g in zip(self.params, self.grids):
            m, e = S.conservation_vars(p, g)
            mass, energy = mass + m, energy + e
        return mass, energy

    def _remote(self, p, axis):
        return (p.neighbours[first_side(axis)] != PROC_NULL, p.neighbours[last_side(axis)] != PROC_NULL)

    def _fused_sweep(self, axis, dt, dx, **emit):
        """One fused sweep of every tile: post the halos, sweep what reads no ghost cell while they travel, finish the
        exchange, sweep the LAG-wide strips next to the remote sides."""
        lag = S.sweep_lag(self.root)
        i_ax = int(axis) - 1
        any_remote = any(any(self._remote(p, axis)) for p in self.params)
        if any_remote:
            self.exchange_start(axis, S.STATE_VARS)
        late = []
        for p, g in zip(self.params, self.grids):
            lo_r, hi_r = self._remote(p, axis)
            n = p.N[i_ax]
            if not (lo_r or hi_r):
                S.fused_sweep(p, g, axis, dt, dx, swap=False, **emit)
            elif n < 2 * lag + 1 or not p.overlap_halo:
                late.append((p, g, None))
            else:
                lo, hi = (lag if lo_r else 0), (n - lag if hi_r else n)
                S.fused_sweep(p, g, axis, dt, dx, out_range=(lo, hi), swap=False, **emit)
                late.append((p, g, (lo, hi, n)))
        # Edge stream (default): unpack and strips run on each tile's transfer stream, concurrent with the interiors,
        # when every tile with a remote side overlaps (a tile too small to have an interior sweeps after the unpack, on
        # its compute stream, so then the whole group takes the in-order form).
        on_edge = any_remote and self.root.edge_stream and all(part is not None for _, _, part in late)
        if any_remote:
            (self.exchange_finish_edge if on_edge else self.exchange_finish)(axis, S.STATE_VARS)
        sz = np.dtype(self.root.data_type).itemsize
        for p, g, part in late:
            if part is None:
                S.fused_sweep(p, g, axis, dt, dx, swap=False, **emit)
                continue
            lo, hi, n = part
            k = self.params.index(p)
            for side, rng in enumerate(((0, lo), (hi, n))):
                if rng[0] == rng[1]:
                    continue
                if on_edge:
                    S.fused_sweep(p, g, axis, dt, dx, out_range=rng, swap=False, ctx=self._edge_ctx[k],
                                  dt_out=self._edge_dt[k] + side * sz, **emit)
                else:
                    S.fused_sweep(p, g, axis, dt, dx, out_range=rng, swap=False, dt_accumulate=True, **emit)
        if on_edge:
            self.edge_join(bool(emit.get("emit_dt")))
        for g in self.grids:
            g.swap_state()

    def _staged_sweep(self, axis, dt, dx):
        for p, g in zip(self.params, self.grids):
            S.update_EOS(p, g, axis)
        if any(any(self._remote(p, axis)) for p in self.params):
            self.exchange_start(axis, S.COMM_VARS)
            self.exchange_finish(axis, S.COMM_VARS)
        for p, g in zip(self.params, self.grids):
            for side in sides_along(axis):
                if p.neighbours[side] == PROC_NULL:
                    S.boundary_conditions(p, g, axis, side)
            S.numerical_fluxes(p, g, axis, dt, dx)
            S.cell_update(p, g, axis, dt, dx)
            S.projection_remap(p, g, axis, dt, dx)

    def _local_dt_to_device(self):
        """dtCFL of every tile into its device scalar (no host round trip), then the global minimum."""
        for p, g in zip(self.params, self.grids):
            r = p.block_size.domain_range(*p.steps_ranges[Axis.X].real_domain).to_c()
            check(p.fn("dtCFL_async")(p.device.ctx, r, p.cell_size(0), p.cell_size(1), g.ptr("u"), g.ptr("v"),
                                        g.ptr("c"), C.c_void_p(g.dt_scalar.ptr)))
        self.dt_allreduce()

    def _post_dt_readback(self):
        """Tile 0's scalar (already the global minimum, ordered on its stream) into a pinned slot + event."""
        p, g, cycle = self.root, self.grids[0], self.global_dt.cycle
        if self.dt_host is None:
            self.dt_host = p.device.pinned(2, p.data_type)
        self.dt_host.copy_from_device_async(g.dt_scalar, n=1, dst_offset=cycle & 1)
        p.device.event_record(S.DT_EVENT_SLOT + (cycle & 1))
        self.dt_inflight[cycle] = S.DT_EVENT_SLOT + (cycle & 1)

    def _take_dt_readback(self, posted_in_cycle):
        slot = self.dt_inflight.pop(posted_in_cycle)
        if isinstance(slot, tuple):                       # posted by armon_hip_mgpu_cycle: the event lives in tile 0's edge context
            check(self._L.armon_hip_event_sync(self._edge_ctx[0], slot[1]))
        else:
            self.root.device.event_sync(slot)
        return float(self.dt_host.array[posted_in_cycle & 1])

    def solver_cycle(self, last_cycle=True):
        """ref src/solver.jl:288-320 over every tile."""
        p0, gdt = self.root, self.global_dt
        fused = p0.use_fused_sweep
        deferred = (gdt.cycle - 1) in self.dt_inflight
        if gdt.cycle == 0:
            for p, g in zip(self.params, self.grids):
                S.update_EOS(p, g)
        if not deferred and not p0.cst_dt:
            self._local_dt_to_device()
            self._post_dt_readback()
            gdt.update_dt(self._take_dt_readback(gdt.cycle))
            if fused and gdt.cycle == 0:
                for g in self.grids:
                    g.release_scratch()          # c, g: only the EOS + dtCFL of cycle 0 needed them
        if native_cycle_usable(p0):
            # the whole cycle of every tile in one library call (one host thread per tile inside)
            if self.dt_host is None:
                self.dt_host = p0.device.pinned(2, p0.data_type)
            plan, n_sweeps = cycle_plan(p0, gdt, last_cycle, dt_host=self.dt_host)
            check(self._fn("mgpu_cycle")(self.handle, C.byref(plan), self._tile_cycles()))
            if n_sweeps & 1:
                for g in self.grids:
                    g.swap_state()
            if not p0.cst_dt:
                # the library reduced the next CFL step and posted its read-back on tile 0's transfer stream
                self.dt_inflight[gdt.cycle] = ("edge", plan.dt_event_slot)
                if deferred:
                    gdt.update_dt(self._take_dt_readback(gdt.cycle - 1))
            return
        sweeps = S.split_axes(p0.axis_splitting, gdt.cycle)
        for k, (axis, dt_factor) in enumerate(sweeps):
            dx = p0.cell_size(int(axis) - 1)
            dt = gdt.current_dt * p0.T(dt_factor)
            if not fused:
                self._staged_sweep(axis, dt, dx)
                continue
            last = k == len(sweeps) - 1
            self._fused_sweep(axis, dt, dx, emit_p=last and last_cycle, emit_dt=last and not p0.cst_dt)
            if last and not p0.cst_dt:
                self.dt_allreduce()
                self._post_dt_readback()
                if deferred:
                    gdt.update_dt(self._take_dt_readback(gdt.cycle - 1))

    def time_loop(self):
        """ref src/solver.jl:323-403"""
        p0, gdt = self.root, self.global_dt
        gdt.reset()
        self.dt_inflight.clear()
        self.wait()
        t1 = _time.perf_counter_ns()
        maxtime = p0.T(p0.maxtime)
        while gdt.time < maxtime and gdt.cycle < p0.maxcycle:
            if p0.cst_dt:
                ends = p0.T(gdt.time + gdt.current_dt) >= maxtime or gdt.cycle + 1 >= p0.maxcycle
            else:
                ends = (gdt.cycle + 1 >= p0.maxcycle or gdt.current_dt == 0
                        or p0.T(gdt.time + gdt.current_dt) >= maxtime)
            self.solver_cycle(last_cycle=ends)
            gdt.next_cycle()
        self.drain()
        self.wait()
        return _time.perf_counter_ns() - t1

    def run(self):
        """``armon(params)`` for the whole group → SolverStats (``data`` = this group)."""
        self.init_test()
        solve_ns = self.time_loop()
        gdt, g = self.global_dt, self.root.global_grid
        cells = g[0] * g[1]
        return S.SolverStats(float(gdt.time), float(gdt.current_dt), gdt.cycle, solve_ns / 1e9, cells,
                             gdt.cycle * cells / max(solve_ns, 1), data=self)

    def gather(self, names=("rho", "u", "v", "E", "p")):
        """The real cells of every tile assembled into global (NY, NX) arrays on the host."""
        self.wait()
        gx, gy = self.root.global_grid
        out = {k: np.empty((gy, gx), dtype=self.root.data_type) for k in names}
        for p, g in zip(self.params, self.grids):
            ox, oy = p.N_origin[0] - 1, p.N_origin[1] - 1
            nx, ny = p.N
            for k in names:
                out[k][oy:oy + ny, ox:ox + nx] = g.real_view(g.data[k].to_host())
        return out


class NativeRcclExchanger:
    """One process per GPU: this rank's tile exchanges its halos and reduces dt through the library's RCCL group
    (``armon_hip_mgpu_init_rank``). Same interface as ``halo_exchange.HaloExchanger`` (start / finish / exchange /
    allreduce_min_device_async), so ``solver.fused_sweep_overlapped`` and the staged ``block_ghost_exchange`` drive
    it unchanged. The rendezvous (128-byte RCCL ids from rank 0) travels through torch.distributed's store — the
    launcher's job, as MPI_Bcast would be for a Julia host."""

    stream_ordered = True           # everything is ordered on the device: the host never waits inside a cycle
    native = True

    def __init__(self, params, grid):
        import torch.distributed as dist
        L = _lib.lib()
        self.params, self.grid, self._L = params, grid, L
        group = params.global_comm
        buf = C.create_string_buffer(_lib.MGPU_ID_BYTES)
        obj = [None]
        if params.rank == 0:
            # a failure here (no RCCL in the process) still has to reach the broadcast every other rank is waiting in
            try:
                check(L.armon_hip_mgpu_unique_id(buf))
                obj = [bytes(buf.raw)]
            except _lib.SolverException as e:
                obj = [str(e)]
        dist.broadcast_object_list(obj, src=0, group=group)
        if not isinstance(obj[0], bytes):
            raise _lib.SolverException("cpp", f"rank 0 could not create the RCCL ids: {obj[0]}")
        ident = C.create_string_buffer(obj[0], _lib.MGPU_ID_BYTES)
        # Local part first (RCCL symbols, context, streams, scratch: it can fail on one rank alone), then the ranks agree
        # over the launcher's group, and only then the collective ncclCommInitRank — entered by every rank or by none: a
        # rank that skipped it after a local failure would leave the others blocked in it for ever.
        self.handle, err = C.c_void_p(), None
        try:
            dev = params.device                 # the tile's kernels keep running on the context the run already uses
            check(L.armon_hip_mgpu_prepare_rank(params.proc_dims[0], params.proc_dims[1], params.rank, params.device_id,
                                                C.c_void_p(dev.stream), C.byref(self.handle)))
            if any(params.periodic):
                check(L.armon_hip_mgpu_set_periodic(self.handle, int(params.periodic[0]), int(params.periodic[1])))
        except Exception as e:           # ANY local failure must still reach the all_gather every other rank is waiting in
            err = e
        ready = [None] * dist.get_world_size(group)
        dist.all_gather_object(ready, err is None, group=group)
        if not all(ready):
            self.close()
            raise _lib.SolverException("cpp", f"native RCCL group: local initialisation failed on rank(s) "
                                              f"{[r for r, ok in enumerate(ready) if not ok]}" + (f": {err}" if err else ""))
        try:
            check(L.armon_hip_mgpu_connect(self.handle, ident))
        except BaseException:
            self.close()                 # the prepared handle (context, streams, scratch) does not outlive a failed connect
            raise
        # the group made its own context on that same stream; sweeps stay on params.device (same stream → same order)
        rank, coords, nb = C.c_int(), (C.c_int * 2)(), (C.c_int * 4)()
        check(L.armon_hip_mgpu_tile_info(self.handle, 0, C.byref(rank), C.byref(coords), C.byref(nb)))
        assert rank.value == params.rank and tuple(coords) == params.cart_coords
        assert [params.neighbours[s] for s in (Side.Left, Side.Right, Side.Bottom, Side.Top)] == list(nb)
        self.edge_ctx = C.c_void_p(L.armon_hip_mgpu_edge_ctx(self.handle, 0))      # this tile's transfer stream as a context
        self.edge_dt = int(L.armon_hip_mgpu_edge_dt(self.handle, 0))
        self._tcs = {}

    def _fn(self, name):
        return getattr(self._L, "armon_hip_" + name + self.params.suffix)

    def _desc(self, names):
        return _halo_descs([self.params], [self.grid], names)

    def start(self, sides, names):
        sides = [s for s in sides if self.params.neighbours[s] != PROC_NULL]
        if not sides:
            return None
        axis = Axis.X if sides[0] in (Side.Left, Side.Right) else Axis.Y
        check(self._fn("halo_exchange_start")(self.handle, int(axis) - 1, self._desc(names)))
        return axis, tuple(names)

    def finish(self, handle):
        if handle is None:
            return
        axis, names = handle
        check(self._fn("halo_exchange_finish")(self.handle, int(axis) - 1, self._desc(names)))

    def finish_edge(self, handle):
        """Unpack on the transfer stream (no wait on the compute stream): the strips follow there, then ``edge_join``."""
        if handle is None:
            return
        axis, names = handle
        check(self._fn("halo_exchange_finish_edge")(self.handle, int(axis) - 1, self._desc(names)))

    def edge_join(self, dt_scalar=None):
        ptrs = (C.c_void_p * 1)(dt_scalar.ptr) if dt_scalar is not None else None
        check(self._fn("mgpu_edge_join")(self.handle, ptrs))

    def exchange(self, sides, names):
        self.finish(self.start(sides, names))

    def allreduce_min_device_async(self, scalar):
        ptrs = (C.c_void_p * 1)(scalar.ptr)
        check(self._fn("dt_allreduce")(self.handle, ptrs))

    def set_chaos(self, max_delay_us, seed=0):
        """Test aid: random busy-wait kernels in front of the group's stream operations (armon_hip_mgpu_set_chaos)."""
        check(self._L.armon_hip_mgpu_set_chaos(self.handle, int(max_delay_us), int(seed)))

    # the whole cycle of this rank's tile in one library call (solver.solver_cycle takes this branch when it can)
    native_cycle = True

    def _tile_cycles(self):
        key = self.grid.data["rho"].ptr
        if key not in self._tcs:
            self._tcs[key] = tile_cycle_descs([self.params], [self.grid])
        return self._tcs[key]

    def cycle(self, gdt, last_cycle):
        """One solver cycle (exchanges, sweeps, the global minimum of the next CFL step and its read-back, posted on the
        transfer stream) → the event slot of that read-back in the edge context, or None (cst_dt)."""
        grid = self.grid
        if grid.dt_host is None:
            grid.dt_host = self.params.device.pinned(2, self.params.data_type)
        plan, n_sweeps = cycle_plan(self.params, gdt, last_cycle, dt_host=grid.dt_host)
        check(self._fn("mgpu_cycle")(self.handle, C.byref(plan), self._tile_cycles()))
        if n_sweeps & 1:
            grid.swap_state()
        return None if self.params.cst_dt else plan.dt_event_slot

    def event_sync(self, slot):
        check(self._L.armon_hip_event_sync(self.edge_ctx, slot))

    def drain(self):
        """End of a run: complete the exchange posted ahead, and let the transfer stream finish (the last reduction and its
        read-back are still on it; ``params.wait()`` only covers the compute stream)."""
        if self.handle and native_cycle_usable(self.params):
            check(self._fn("mgpu_drain")(self.handle, self._tile_cycles()))
            check(self._L.armon_hip_mgpu_sync(self.handle))

    def allreduce_host(self, values, op):
        v = (C.c_double * len(values))(*values)
        check(self._L.armon_hip_mgpu_allreduce_host(self.handle, 0 if op == "sum" else 1, len(values), v))
        return list(v)

    def close(self):
        if self.handle:
            self._L.armon_hip_mgpu_destroy(self.handle)
            self.handle = None
