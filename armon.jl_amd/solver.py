"""Solver driver of the HIP backend: ``armon(params)``, ``time_loop``, ``solver_cycle`` and the dt state.

Mirrors ref src/solver.jl:288-516 (synchronous cycle — the path a device backend uses),
src/solver_state.jl:58-166 (GlobalTimeStep), src/axis_splitting.jl:24-46, and the per-block kernel
wrappers of src/kernels.jl:151-230, src/riemann_schemes.jl:46-120, src/projection_schemes.jl:44-157,
src/halo_exchange.jl:32-36,286-368, src/reductions.jl:89-110,164-199,301-323. Every kernel call goes
through the C ABI of libarmon_hip.so; nothing here computes on the host.
"""
import ctypes as C
import math
import time as _time
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import BlockDataPtrs, FIELDS, SweepDesc, check, solver_error
from .blocking import Axis, Side, first_side, last_side, sides_along
from .parameters import LIMITERS, PROC_NULL, PROJECTIONS, SCHEMES

MAIN_VARS = ("x", "y", "rho", "u", "v", "E", "p", "c", "g", "us", "ps")   # ref src/blocking/blocks.jl:47
SAVED_VARS = ("x", "y", "rho", "u", "v", "p")                              # ref :49
COMM_VARS = ("rho", "u", "v", "E", "p", "c", "g")                          # ref :50
STATE_VARS = ("rho", "u", "v", "E")


@dataclass
class SolverStats:   # ref src/solver.jl:13-23
    final_time: float
    last_dt: float
    cycles: int
    solve_time: float          # seconds
    cell_count: int
    giga_cells_per_sec: float
    data: object = None
    timer: object = None
    grid_log: object = None

    def __str__(self):   # ref src/solver.jl:26-35
        return (f"Solver stats:\n - final time:  {self.final_time}\n - last Δt:     {self.last_dt}\n"
                f" - cycles:      {self.cycles}\n - performance: {self.giga_cells_per_sec} ×10⁹ cell-cycles/sec "
                f"({self.solve_time} sec, {self.cell_count} cells)")


class GlobalTimeStep:
    """ref src/solver_state.jl:30-166 — cycle/time/dt with the reference's one-cycle lag. All arithmetic is
    done in the run's data type T (numpy scalars), like the reference's ``GlobalTimeStep{T}``."""

    def __init__(self, params):
        self.params = params
        self.reset()

    def reset(self):   # ref :58-68
        p = self.params
        T = p.T
        self.cycle = 0
        self.time = T(0.)
        self.current_dt = T(p.Dt) if p.cst_dt else T(0.)
        self.next_cycle_dt = T(math.inf)

    def update_dt(self, new_dt):
        """ref update_dt!, :102-142 (``new_dt`` = global minimum of the local CFL time steps)."""
        p = self.params
        T = p.T
        new_dt = T(new_dt)
        previous_dt = self.current_dt
        if not math.isfinite(new_dt) or new_dt <= 0:
            solver_error("time", f"Invalid time step for cycle {self.cycle}: {new_dt}")
        elif previous_dt == 0:
            new_dt = T(p.cfl) * new_dt
        else:
            # convert(T, min(cfl·new_dt, 1.05·previous_dt)) with the 1.05 product in Float64 (ref :129)
            new_dt = T(min(float(T(p.cfl) * new_dt), 1.05 * float(previous_dt)))
        self.next_cycle_dt = new_dt
        if self.current_dt == 0:
            self.current_dt = self.next_cycle_dt

    def next_cycle(self):
        """ref next_cycle!, :145-166"""
        p = self.params
        T = p.T
        self.cycle += 1
        self.time = T(self.time + self.current_dt)
        if p.cst_dt:
            self.current_dt = self.next_cycle_dt = T(p.Dt)
            return
        self.current_dt = self.next_cycle_dt
        self.next_cycle_dt = T(math.inf)


def split_axes(splitting, cycle):
    """ref src/axis_splitting.jl:24-46 → sequence of (axis, dt_factor)."""
    even = cycle % 2 == 0
    if splitting == "Sequential":
        return ((Axis.X, 1.0), (Axis.Y, 1.0))
    if splitting in ("Godunov", "SequentialSym"):
        return ((Axis.X, 1.0), (Axis.Y, 1.0)) if even else ((Axis.Y, 1.0), (Axis.X, 1.0))
    if splitting == "Strang":
        return (((Axis.X, 0.5), (Axis.Y, 1.0), (Axis.X, 0.5)) if even
                else ((Axis.Y, 0.5), (Axis.X, 1.0), (Axis.Y, 0.5)))
    if splitting == "X_only":
        return ((Axis.X, 1.0),)
    if splitting == "Y_only":
        return ((Axis.Y, 1.0),)
    solver_error("config", f"Unknown splitting method: '{splitting}'")


# what the fused path touches in every run: the saved variables (ref saved_vars, src/blocking/blocks.jl:49) and the state
FUSED_FIELDS = ("x", "y", "rho", "u", "v", "E", "p")


class _Fields(dict):
    """The 16 ``BlockData`` vectors by name. On the staged path all of them exist from the start, as in the reference (ref
    src/blocking/blocks.jl:36-44). On the fused path only the seven of ``FUSED_FIELDS`` do — ``us, ps, work_1..4, mask`` are
    never touched by a fused run, ``c, g`` only by the EOS + dtCFL of cycle 0 — and any other is allocated on first access,
    holding what ``init_test`` would have left in it (zeros; 0 / 1 for ``mask``): a 16384² block keeps 11 + the 8 transient
    spares of the placement search instead of 20 + 8 vectors (config 5's global grid on one GPU: 172 -> 95 GB), while the
    ``BlockData`` the C ABI sees stays 16 pointers (NULL = not in ``vars_to_zero``)."""

    def __init__(self, grid, names):
        super().__init__()
        import weakref
        self._grid_ref = weakref.ref(grid)     # no reference cycle: a dropped grid must release its vectors at once (86 GB blocks follow each other in the tests)
        for f in names:
            self[f] = grid.params.device.empty(grid.size.n_cells, grid.params.data_type)

    def __missing__(self, key):
        if key not in FIELDS:
            raise KeyError(key)
        grid = self._grid_ref()
        a = grid.params.device.zeros(grid.size.n_cells, grid.params.data_type)
        if key == "mask" and grid.initialised:
            g = grid.size.ghosts
            nx, ny = grid.size.real_size
            m = np.zeros((ny + 2 * g, nx + 2 * g), dtype=grid.params.data_type)
            m[g:g + ny, g:g + nx] = 1
            a.copy_from_host(m.ravel())
        self[key] = a
        grid.lazy.add(key)
        return a


class BlockGrid:
    """One block per GPU (ref BlockGrid with use_cache_blocking=false, src/blocking/block_grid.jl:352-355):
    the 16 ``BlockData`` vectors (ref src/blocking/blocks.jl:18-44) in HBM, plus the 4 ping-pong state
    vectors the fused sweep writes to."""

    def __init__(self, params):
        self.params = params
        self.size = params.block_size
        dev = params.device
        n = self.size.n_cells
        dt_ = params.data_type
        self.initialised = False               # init_test has run (a vector created later is given its initial content)
        self.lazy = set()                      # fused path: the vectors created on first access
        self.data = _Fields(self, FUSED_FIELDS if params.use_fused_sweep else FIELDS)
        self.alt = {f: dev.empty(n, dt_) for f in STATE_VARS} if params.use_fused_sweep else None
        self.placement = None                  # report of tune_placement (bench / tests)
        self.global_dt = GlobalTimeStep(params)
        self.dt_scalar = dev.zeros(2, dt_)     # device scalar written by the fused dt reduction
        self.dt_host = None                    # pinned landing zone of dt_scalar, one slot per cycle parity
        self.dt_inflight = {}                  # cycle that posted it -> event slot (see post_dt_readback)
        self.comm = None                       # set by halo_exchange.setup when use_MPI
        self.halo_prefetch = None              # (axis, handle) of an exchange posted ahead of its sweep

    def ptr(self, name):
        return C.c_void_p(self.data[name].ptr)

    def block_data_ptrs(self):
        """``armon_block_data``: the vectors that exist (NULL for those the fused path has not needed yet)."""
        bd = BlockDataPtrs()
        for f in FIELDS:
            if f in self.data:
                setattr(bd, f, self.data[f].ptr)
        return bd

    def release_scratch(self, names=("c", "g")):
        """Fused path: free vectors that were only created on the way (``c, g`` by the EOS + dtCFL of cycle 0)."""
        for f in names:
            if f in self.lazy and f in self.data:
                self.params.wait()
                self.data.pop(f).free()
                self.lazy.discard(f)

    def swap_state(self):
        """After a fused sweep the fresh state lives in ``alt``: exchange the roles."""
        for f in STATE_VARS:
            self.data[f], self.alt[f] = self.alt[f], self.data[f]

    def tune_placement(self, min_bytes=None, spare=None, keep_state=True):
        """Pick a good PHYSICAL placement for the 8 vectors a fused sweep streams (4 read + 4 written).

        On MI355X the same kernels on the same virtual layout run 10-20 % apart depending on where the 8
        vectors sit in HBM relative to each other: vectors allocated back to back end up at a REGULAR
        physical spacing, and for many spacings the 8 streams then keep meeting on the same channels/banks
        (tools/probes/probe_skew.hip, probe_pairs.hip: 2.93-3.77 ms for the same copy as a function of the
        spacing; random picks of 8 among 24 such vectors: 75 % at 2.71-2.79 ms, the back-to-back groups 2.94-3.08).
        Nothing visible from user space predicts it, so it is measured, natively: ``spare`` extra vectors are
        allocated, up to ``placement_tries`` assignments of the 8 roles to vectors of the pool are timed with one X
        and one Y sweep (the first one being the back-to-back assignment), the fastest is kept and the unused
        vectors are freed.

        ``keep_state=False`` (what ``init_test`` does, BEFORE it writes the initial condition):
        ``armon_hip_choose_placement`` — the vectors hold nothing yet, candidates are timed on a uniform state, the
        search may stop after 12 draws once two of them lie within 1 % of the best and 7 % under the worst; transient
        memory = the 8 ``spare`` vectors only (17 GB at 16384²). The pool is deliberately small: among 12-16 vectors
        allocated one after the other about half of the random draws land on the fast level, among 32 one in twenty
        (profiles/r02_placement_pool_sizes.txt). A pool is either slow for EVERY assignment or fast for most of them
        (profiles/r02_placement_what_it_is_not.txt), so a round is short — ``placement_tries`` = 24 draws at most, 12 when two of them already lie within 0.4 % of the best — and a round
        that found nothing is followed by a fresh batch of spares, up to ``placement_rounds`` = 8 times.
        ``keep_state=True``: ``armon_hip_tune_placement`` moves a LIVE state around (4 more vectors park it meanwhile). ~20 ms per try, outside any timed region.
        Returns the report also stored in ``self.placement``."""
        params, dev = self.params, self.params.device
        tries = getattr(params, "placement_tries", 0)
        nbytes = self.data["rho"].nbytes
        if min_bytes is None:
            min_bytes = getattr(params, "placement_min_bytes", 256 << 20)
        if spare is None:
            spare = 8
        if self.alt is None or tries <= 1 or nbytes < min_bytes:
            return None
        free, _total = dev.memory_info()
        spare = int(min(spare, (free * 0.8) // max(nbytes, 1) - (4 if keep_state else 0)))
        if spare < 1:
            return None
        dx = params.cell_size(0)
        dt = 1e-3 * dx                    # any small step: the arithmetic does not depend on the data
        n, dt_ = self.data["rho"].n, self.data["rho"].dtype
        pool = [self.data[f] for f in STATE_VARS] + [self.alt[f] for f in STATE_VARS]
        try:
            for _ in range(spare):
                pool.append(dev.empty(n, dt_))
        except _lib.SolverException:          # not enough memory after all: keep the placement we have
            for v in pool[8:]:
                v.free()
            return None
        d_x = sweep_desc(params, self, Axis.X, dt, dx)
        d_y = sweep_desc(params, self, Axis.Y, dt, params.cell_size(1), emit_dt=True)     # as in a cycle
        # A search can come back empty-handed: in about one process in four none of the draws of a 16-vector pool
        # reaches the fast level (profiles/r02_placement_pool_sizes.txt). The fast level is recognisable on its own —
        # an X + Y pair then streams its 16 vectors' worth of bytes at >= 0.93 of the 6.29 TB/s this part sustains —
        # so when the best draw stays below that, a fresh batch of spares is allocated NEXT TO the losers (freeing them
        # first would hand back the same memory) and the search is repeated with the best assignment so far as its
        # first draw, at most `placement_rounds` times. (Arithmetic-bound sweeps — exact flavour, Bizarrium — may never
        # reach the mark: they just use their rounds.)
        rounds = max(1, int(getattr(params, "placement_rounds", 3))) if not keep_state else 1
        # (fp32 sweeps top out at 0.89-0.91 of that rate, so their mark sits at 0.88: the search must be able to stop)
        fast_frac = 0.88 if np.dtype(params.data_type).itemsize == 4 else 0.93
        fast_ms = 2 * 8 * nbytes / (fast_frac * 6.29e12) * 1e3
        losers, all_times, report_rounds = [], [], 0
        best = pool[:8]                           # what the grid uses if nothing better is found
        while True:
            report_rounds += 1
            ptrs = (C.c_void_p * len(pool))(*[v.ptr for v in pool])
            picks, times, done = (C.c_int * 8)(), (C.c_double * tries)(), C.c_int(tries)
            try:
                if keep_state:
                    check(params.fn("tune_placement")(dev.ctx, C.byref(d_x), C.byref(d_y), ptrs, len(pool), nbytes, tries,
                                                      C.byref(picks), times))
                else:
                    check(params.fn("choose_placement")(dev.ctx, C.byref(d_x), C.byref(d_y), ptrs, len(pool), nbytes,
                                                        tries, 0.004, C.byref(picks), times, C.byref(done)))
            except _lib.SolverException:          # e.g. no room for the 4 transient vectors: nothing was moved in THIS round
                # pool[:8] is the assignment this round started from — the grid's own vectors in round 1, the best of the
                # previous rounds afterwards (the grid's original vectors may then be among the losers): install it before
                # anything is freed, so that the grid never points at a released vector
                for k, f in enumerate(STATE_VARS):
                    self.data[f], self.alt[f] = best[k], best[4 + k]
                for v in pool[8:] + losers:
                    v.free()
                return None
            picks, times = list(picks), list(times)[:done.value]
            all_times.append([round(t, 3) for t in times])
            best = [pool[k] for k in picks]
            losers += [v for i, v in enumerate(pool) if i not in set(picks)]
            if min(times) <= fast_ms or report_rounds >= rounds:
                break
            free, _total = dev.memory_info()
            if free < (spare + 2) * nbytes:
                break
            try:
                pool = best + [dev.empty(n, dt_) for _ in range(spare)]
            except _lib.SolverException:
                break
        for k, f in enumerate(STATE_VARS):
            self.data[f], self.alt[f] = best[k], best[4 + k]
        for v in losers:
            v.free()
        flat = [t for r in all_times for t in r]
        self.placement = {"tries": len(flat), "max_tries": tries, "pool": 8 + spare, "rounds": report_rounds,
                          "x_plus_y_ms": all_times[0] if report_rounds == 1 else all_times,
                          "chosen_ms": round(min(all_times[-1]), 3), "fast_level_ms": round(fast_ms, 3),
                          "transient_GB": round((report_rounds * spare + (4 if keep_state else 0)) * nbytes / 1e9, 2)}
        return self.placement

    def device_to_host(self, names=MAIN_VARS):
        """ref device_to_host!, src/blocking/blocks.jl:121-131"""
        self.params.wait()
        return {f: self.data[f].to_host() for f in names}

    def host_to_device(self, host):
        for f, a in host.items():
            self.data[f].copy_from_host(a)

    def real_view(self, a):
        g = self.size.ghosts
        nx, ny = self.size.real_size
        return a.reshape(self.size.size[1], self.size.size[0])[g:g + ny, g:g + nx]

    def memory_required(self):
        """Bytes of device memory the block's vectors take (ref memory_required, src/blocking/block_grid.jl): the 16 of
        ``BlockData`` on the staged path; 7 + 4 ping-pong partners on the fused one (+ any created on first access)."""
        n = self.size.n_cells * np.dtype(self.params.data_type).itemsize
        return n * (len(self.data) + (4 if self.alt else 0))


# ---- per-block kernel wrappers -----------------------------------------------------------------

class _KernelSection:
    """Kernel profiling callbacks — ref ``kernel_start``/``kernel_end`` wrapped around every kernel launch
    when profiling is enabled (ref src/generic_kernel.jl:869-876,892-908, src/profiling.jl:6-68).
    ``params.kernel_callbacks`` = list of objects with ``start(name)`` / ``end(name)``."""
    __slots__ = ("cbs", "name")

    def __init__(self, params, name):
        self.cbs = params.kernel_callbacks
        self.name = name

    def __enter__(self):
        for cb in self.cbs:
            cb.start(self.name)

    def __exit__(self, *exc):
        for cb in self.cbs:
            cb.end(self.name)
        return False


def _k(params, name):
    return _KernelSection(params, name)


def _range(params, corners):
    return params.block_size.domain_range(*corners).to_c()


def init_test(params, grid, tune=True):
    """ref src/kernels.jl:176-207 (+ the measured choice of the HBM placement of the vectors, made BEFORE they are
    written; ``tune=False`` to skip it, e.g. when re-initialising a grid that has been placed already)"""
    if tune and grid.placement is None:
        if params.use_fused_sweep:
            grid.tune_placement(keep_state=False)
        else:
            tune_staged_placement(params, grid)
    _init_test_kernel(params, grid)
    grid.initialised = True


def tune_staged_placement(params, grid, min_bytes=None):
    """The staged path's counterpart of ``BlockGrid.tune_placement``: the 16 ``BlockData`` vectors have the same size,
    so ANY assignment of the 16 allocations to the 16 fields is a valid layout, and which one is taken moves the staged
    kernels by 5-20 % (advection_second_order 3.25 ... 4.0 ms at 16384², tools/staged_placement_probe.py). Up to 12 random
    assignments — of 16 among the 16 vectors and 8 spares that are freed afterwards — are timed with one staged cycle (X then Y: EOS, boundary conditions, fluxes, cell update, advection,
    projection) on the initial condition, the fastest is kept. ≈40 ms per try, before any timed region."""
    import random
    tries = getattr(params, "placement_tries", 0)
    nbytes = grid.data["rho"].nbytes
    if min_bytes is None:
        min_bytes = getattr(params, "placement_min_bytes", 256 << 20)
    # a tile with a remote side (an MPI rank, or a tile of an in-process group: use_MPI is False there and grid.comm is
    # None) cannot run the timing cycle below on its own — its ghost exchange needs the neighbours
    if tries <= 1 or nbytes < min_bytes or params.use_MPI or any(n != PROC_NULL for n in params.neighbours.values()):
        return None
    dev = params.device
    vectors = [grid.data[f] for f in FIELDS]
    free, _total = dev.memory_info()
    try:        # 8 spare vectors widen the choice (freed afterwards): 16 of 24 allocations
        n_spare = int(min(8, (free * 0.5) // max(nbytes, 1)))
        vectors += [dev.empty(vectors[0].n, vectors[0].dtype) for _ in range(max(n_spare, 0))]
    except _lib.SolverException:
        pass
    tries = min(12, getattr(params, "placement_tries", 0))
    rng = random.Random(0x5EED)
    dx = params.cell_size(0)
    dt = params.T(0.2) * dx
    times, perms = [], []
    callbacks, params.kernel_callbacks = params.kernel_callbacks, []
    for t in range(tries):
        perm = list(range(len(FIELDS))) if t == 0 else rng.sample(range(len(vectors)), len(FIELDS))
        for f, k in zip(FIELDS, perm):
            grid.data[f] = vectors[k]
        _init_test_kernel(params, grid)
        update_EOS(params, grid)
        dev.event_record(1002)
        for axis in (Axis.X, Axis.Y):
            update_EOS(params, grid, axis)
            block_ghost_exchange(params, grid, axis)
            numerical_fluxes(params, grid, axis, dt, params.cell_size(int(axis) - 1))
            cell_update(params, grid, axis, dt, params.cell_size(int(axis) - 1))
            projection_remap(params, grid, axis, dt, params.cell_size(int(axis) - 1))
        dev.event_record(1003)
        times.append(dev.event_elapsed_ms(1002, 1003))
        perms.append(perm)
    params.kernel_callbacks = callbacks
    k = times.index(min(times))
    for f, i in zip(FIELDS, perms[k]):
        grid.data[f] = vectors[i]
    for i, v in enumerate(vectors):
        if i not in perms[k]:
            v.free()
    grid.placement = {"staged": True, "tries": tries, "pool": len(vectors), "cycle_ms": [round(t, 3) for t in times], "chosen": k,
                      "chosen_ms": round(times[k], 3)}
    return grid.placement


def _init_test_kernel(params, grid):
    bs = params.block_size
    full = params.steps_ranges[Axis.X].full_domain
    gpos = (C.c_int64 * 2)(params.N_origin[0] - 1, params.N_origin[1] - 1)
    gN = (C.c_int64 * 2)(*params.global_grid)
    real = C.c_float if params.data_type == np.float32 else C.c_double
    origin = (real * 2)(*params.origin)
    dX = (real * 2)(params.cell_size(0), params.cell_size(1))
    bd = grid.block_data_ptrs()
    check(params.fn("init_test")(params.device.ctx, _range(params, full), params.test.tag,
                                   bs.size[0], bs.size[1], bs.ghosts, C.byref(gpos), C.byref(gN),
                                   C.byref(origin), C.byref(dX), params.test.r, C.byref(bd)))


def update_EOS(params, grid, axis=Axis.X):
    """ref src/kernels.jl:151-173"""
    r = _range(params, params.steps_ranges[axis].EOS)
    p = grid.ptr
    if params.test.eos == "bizarrium":
        with _k(params, "bizarrium_EOS"):
            check(params.fn("bizarrium_EOS")(params.device.ctx, r, p("rho"), p("u"), p("v"), p("E"),
                                               p("p"), p("c"), p("g")))
    else:
        with _k(params, "perfect_gas_EOS"):
            check(params.fn("perfect_gas_EOS")(params.device.ctx, r, params.test.gamma, p("rho"), p("E"),
                                                 p("u"), p("v"), p("p"), p("c"), p("g")))


def boundary_conditions(params, grid, axis, side):
    """ref src/halo_exchange.jl:32-36"""
    bs = params.block_size
    uf, vf = params.test.boundary_condition(side)
    domain = bs.border_domain(side).to_c()
    incr = bs.stride_along(axis)
    if side in (Side.Left, Side.Bottom):
        incr = -incr
    p = grid.ptr
    check(params.fn("boundary_conditions")(params.device.ctx, domain, incr, bs.ghosts, uf, vf,
                                             p("rho"), p("u"), p("v"), p("p"), p("c"), p("g"), p("E")))


def block_ghost_exchange(params, grid, axis):
    """ref src/halo_exchange.jl:286-368: physical boundary → mirror BC; process boundary → exchange."""
    remote = [s for s in sides_along(axis) if params.neighbours[s] != PROC_NULL]
    if remote:
        from .halo_exchange import exchange_sides
        exchange_sides(params, grid, axis, remote, COMM_VARS)
    for side in sides_along(axis):
        if params.neighbours[side] == PROC_NULL:
            boundary_conditions(params, grid, axis, side)


def numerical_fluxes(params, grid, axis, dt, dx):
    """ref src/riemann_schemes.jl:46-52,107-113"""
    r = _range(params, params.steps_ranges[axis].fluxes)
    s = params.block_size.stride_along(axis)
    p = grid.ptr
    ua = p("u") if axis == Axis.X else p("v")
    if params.riemann_scheme == "GAD":
        with _k(params, "acoustic_GAD"):
            check(params.fn("acoustic_GAD")(params.device.ctx, r, s, dt, dx, p("us"), p("ps"), p("rho"), ua,
                                              p("p"), p("c"), LIMITERS[params.riemann_limiter]))
    else:
        with _k(params, "acoustic"):
            check(params.fn("acoustic")(params.device.ctx, r, s, p("us"), p("ps"), p("rho"), ua, p("p"), p("c")))


def cell_update(params, grid, axis, dt, dx):
    """ref src/kernels.jl:217-223"""
    r = _range(params, params.steps_ranges[axis].cell_update)
    s = params.block_size.stride_along(axis)
    p = grid.ptr
    ua = p("u") if axis == Axis.X else p("v")
    with _k(params, "cell_update"):
        check(params.fn("cell_update")(params.device.ctx, r, s, dx, dt, p("us"), p("ps"), p("rho"), ua, p("E")))


def projection_remap(params, grid, axis, dt, dx):
    """ref src/projection_schemes.jl:44-157: advection_fluxes! then euler_projection!"""
    s = params.block_size.stride_along(axis)
    p = grid.ptr
    ra = _range(params, params.steps_ranges[axis].advection)
    w = (p("work_1"), p("work_2"), p("work_3"), p("work_4"))
    if params.projection_scheme == "euler_2nd":
        with _k(params, "advection_second_order"):
            check(params.fn("advection_second_order")(params.device.ctx, ra, s, dx, dt, p("us"), p("rho"),
                                                        p("u"), p("v"), p("E"), *w))
    else:
        with _k(params, "advection_first_order"):
            check(params.fn("advection_first_order")(params.device.ctx, ra, s, dt, p("us"), p("rho"),
                                                       p("u"), p("v"), p("E"), *w))
    rp = _range(params, params.steps_ranges[axis].projection)
    with _k(params, "euler_projection"):
        check(params.fn("euler_projection")(params.device.ctx, rp, s, dx, dt, p("us"), p("rho"), p("u"),
                                              p("v"), p("E"), *w))


def local_time_step(params, grid):
    """ref src/reductions.jl:89-110 (dx, dy are the GLOBAL cell sizes, :92)"""
    r = _range(params, params.steps_ranges[Axis.X].real_domain)
    dx, dy = params.cell_size(0), params.cell_size(1)
    out = (C.c_float if params.data_type == np.float32 else C.c_double)()
    p = grid.ptr
    with _k(params, "dtCFL"):
        check(params.fn("dtCFL")(params.device.ctx, r, dx, dy, p("u"), p("v"), p("c"), C.byref(out)))
    return out.value


def conservation_vars(params, grid):
    """ref src/reductions.jl:262-323 → (total_mass, total_energy), summed over ranks when use_MPI"""
    r = _range(params, params.steps_ranges[Axis.X].real_domain)
    ds = params.cell_size(0) * params.cell_size(1)
    out = ((C.c_float if params.data_type == np.float32 else C.c_double) * 2)()
    check(params.fn("conservation_vars")(params.device.ctx, r, ds, grid.ptr("rho"), grid.ptr("E"), C.byref(out)))
    mass, energy = out[0], out[1]
    if params.use_MPI:
        from .halo_exchange import allreduce_sum
        mass, energy = allreduce_sum(params, (mass, energy))
    return mass, energy


def global_min(params, local_dt):
    """MPI_Iallreduce(MIN) of ref src/solver_state.jl:107-111, src/utils.jl:126-143"""
    if params.use_MPI:
        from .halo_exchange import allreduce_min
        return allreduce_min(params, local_dt)
    return local_dt


# ---- fused sweep -----------------------------------------------------------------------------------

def sweep_lag(params):
    """Cells a fused sweep reads past the cells it writes: 2 + [GAD] + [euler_2nd]."""
    return 2 + (params.riemann_scheme == "GAD") + (params.projection_scheme == "euler_2nd")


def fused_sweep(params, grid, axis, dt, dx, emit_p=False, emit_c=False, emit_dt=False,
                out_range=None, dt_accumulate=False, swap=True, ctx=None, dt_out=None):
    """One directional sweep as a single kernel launch (armon_hip_sweep). The halo cells of process
    boundaries must already hold the neighbour's (ρ,u,v,E). ``emit_dt``: also reduce the CFL time step
    of the resulting state into ``grid.dt_scalar`` (device) — or into ``dt_out``. ``ctx``: another context of the same
    device to launch on (the tile's edge stream, multi_gpu.hip); such launches are not seen by the kernel callbacks,
    whose events live on the compute stream."""
    d = sweep_desc(params, grid, axis, dt, dx, emit_p, emit_c, emit_dt, out_range, dt_accumulate)
    if emit_dt and dt_out is not None:
        d.dt_cfl_out = dt_out
    if ctx is not None:
        check(params.fn("sweep")(ctx, C.byref(d)))
    else:
        with _k(params, "sweep_x" if axis == Axis.X else "sweep_y"):
            check(params.fn("sweep")(params.device.ctx, C.byref(d)))
    if swap:
        grid.swap_state()


def sweep_desc(params, grid, axis, dt, dx, emit_p=False, emit_c=False, emit_dt=False, out_range=None,
               dt_accumulate=False):
    """The ``armon_sweep_desc`` of one sweep of ``grid``'s current state into its alternate arrays."""
    d = SweepDesc()
    d.axis = 0 if axis == Axis.X else 1
    d.scheme = SCHEMES[params.riemann_scheme]
    d.limiter = LIMITERS[params.riemann_limiter]
    d.projection = PROJECTIONS[params.projection_scheme]
    d.eos = 1 if params.test.eos == "bizarrium" else 0
    d.nghost = params.nghost
    lo, hi = first_side(axis), last_side(axis)
    d.bc_low = int(params.neighbours[lo] == PROC_NULL)
    d.bc_high = int(params.neighbours[hi] == PROC_NULL)
    d.exact = int(params.exact_arithmetic)
    d.x_kernel = int(getattr(params, "x_kernel", 0))
    d.nx, d.ny = params.N
    d.dt, d.dx, d.gamma = dt, dx, params.test.gamma
    d.u_factor_low, d.v_factor_low = params.test.boundary_condition(lo)
    d.u_factor_high, d.v_factor_high = params.test.boundary_condition(hi)
    d.rho_in, d.u_in, d.v_in, d.E_in = (grid.data[f].ptr for f in STATE_VARS)
    d.rho_out, d.u_out, d.v_out, d.E_out = (grid.alt[f].ptr for f in STATE_VARS)
    d.p_out = grid.data["p"].ptr if emit_p else None
    d.c_out = grid.data["c"].ptr if emit_c else None
    if emit_dt:
        d.dt_cfl_out = grid.dt_scalar.ptr
        d.cfl_dx = params.cell_size(0)
        d.cfl_dy = params.cell_size(1)
        d.dt_accumulate = int(dt_accumulate)
    if out_range is not None:
        d.out_lo, d.out_hi = out_range
    return d


def fused_sweep_overlapped(params, grid, axis, dt, dx, **emit):
    """Fused sweep of a tile with process boundaries along ``axis``: the halo exchange of (ρ,u,v,E) runs while
    the interior — the cells at least LAG away from the remote sides, which read no ghost cell — is computed;
    the LAG-wide boundary strips follow once the ghosts are unpacked (ref src/solver.jl:58-285 gets the same
    overlap from its async block state machine)."""
    lag = sweep_lag(params)
    n = params.N[int(axis) - 1]
    lo_remote = params.neighbours[first_side(axis)] != PROC_NULL
    hi_remote = params.neighbours[last_side(axis)] != PROC_NULL
    comm = grid.comm
    if not (lo_remote or hi_remote):
        return fused_sweep(params, grid, axis, dt, dx, **emit)
    if n < 2 * lag + 1 or not getattr(params, "overlap_halo", True):
        comm.exchange(sides_along(axis), STATE_VARS)
        return fused_sweep(params, grid, axis, dt, dx, **emit)
    if grid.halo_prefetch is not None and grid.halo_prefetch[0] == axis:
        handle = grid.halo_prefetch[1]            # posted at the end of the previous cycle (prefetch_halo)
        grid.halo_prefetch = None
    else:
        drain_halo(grid)
        handle = comm.start(sides_along(axis), STATE_VARS)
    lo, hi = (lag if lo_remote else 0), (n - lag if hi_remote else n)
    fused_sweep(params, grid, axis, dt, dx, out_range=(lo, hi), swap=False, **emit)
    if getattr(comm, "edge_ctx", None) and getattr(params, "edge_stream", True):
        # the library's group: unpack and strips on the tile's transfer stream, in the shadow of the interior
        comm.finish_edge(handle)
        sz = np.dtype(params.data_type).itemsize
        for k, (remote, rng) in enumerate(((lo_remote, (0, lag)), (hi_remote, (n - lag, n)))):
            if remote:
                fused_sweep(params, grid, axis, dt, dx, out_range=rng, swap=False, ctx=comm.edge_ctx,
                            dt_out=comm.edge_dt + k * sz, **emit)
        comm.edge_join(grid.dt_scalar if emit.get("emit_dt") else None)
    else:
        comm.finish(handle)
        if lo_remote:
            fused_sweep(params, grid, axis, dt, dx, out_range=(0, lag), swap=False, dt_accumulate=True, **emit)
        if hi_remote:
            fused_sweep(params, grid, axis, dt, dx, out_range=(n - lag, n), swap=False, dt_accumulate=True, **emit)
    grid.swap_state()


def prefetch_halo(params, grid):
    """Post the halo exchange of the NEXT cycle's first sweep right after this cycle's last sweep: it depends on
    the state only, not on the next dt, so it travels while the dt reduction is folded, all-reduced and read
    back by the host, and the next cycle's interior sweep can start as soon as its dt is known."""
    if not (params.use_MPI and params.use_fused_sweep) or grid.comm is None or not getattr(params, "overlap_halo", True):
        return
    axis = split_axes(params.axis_splitting, grid.global_dt.cycle + 1)[0][0]
    remote = [s for s in sides_along(axis) if params.neighbours[s] != PROC_NULL]
    if not remote or params.N[int(axis) - 1] < 2 * sweep_lag(params) + 1:
        return
    drain_halo(grid)
    grid.halo_prefetch = (axis, grid.comm.start(sides_along(axis), STATE_VARS))


def drain_halo(grid):
    """Complete a prefetched exchange nobody consumed (end of a run, change of plan)."""
    if getattr(grid.comm, "native_cycle", False):
        grid.comm.drain()                        # what the library's own cycle posted ahead
    if grid.halo_prefetch is not None:
        grid.comm.finish(grid.halo_prefetch[1])
        grid.halo_prefetch = None


# ---- cycle / time loop ---------------------------------------------------------------------------------

def next_time_step(params, grid):
    """ref src/reductions.jl:164-199 (grid form): CFL step of the current state with the dtCFL kernel, global
    minimum, ``update_dt!``. Synchronous; the fused path only needs it on its first cycle (see below)."""
    gdt = grid.global_dt
    if params.cst_dt:
        return
    gdt.update_dt(global_min(params, local_time_step(params, grid)))


# The reference consumes a cycle's CFL reduction in the NEXT cycle (one-cycle lag of GlobalTimeStep,
# ref src/solver_state.jl:89-99,145-166: the MPI_Iallreduce started by cycle n is waited for by cycle n+1).
# The fused path uses that slack: the last sweep of cycle n-1 reduces L(n), the CFL step of the state cycle n
# starts from; its read-back is only POSTED then (all-reduce on the device scalar, asynchronous copy into a pinned
# slot, event), cycle n's sweeps are launched with the dt already known, and L(n) is picked up afterwards for
# ``update_dt!`` -> cycle n+1. The host never drains the stream: it runs at most one cycle ahead of the GPU.
DT_EVENT_SLOT = 1012       # event-pool slots 1012, 1013: one per parity of the posting cycle


def post_dt_readback(params, grid):
    """After the last sweep of the current cycle (which reduced the next cycle's L into ``grid.dt_scalar``)."""
    dev, cycle = params.device, grid.global_dt.cycle
    if grid.dt_host is None:
        grid.dt_host = dev.pinned(2, params.data_type)
    if grid.comm is not None and getattr(grid.comm, "stream_ordered", False):
        grid.comm.allreduce_min_device_async(grid.dt_scalar)      # RCCL, in place, ordered on the stream
    grid.dt_host.copy_from_device_async(grid.dt_scalar, n=1, dst_offset=cycle & 1)
    dev.event_record(DT_EVENT_SLOT + (cycle & 1))
    grid.dt_inflight[cycle] = DT_EVENT_SLOT + (cycle & 1)


def take_dt_readback(params, grid, posted_in_cycle):
    """Global CFL step posted by ``posted_in_cycle`` (blocks only if the GPU has not got there yet)."""
    slot = grid.dt_inflight.pop(posted_in_cycle)
    if isinstance(slot, tuple):                   # posted by armon_hip_mgpu_cycle on the tile's transfer stream (its edge context)
        grid.comm.event_sync(slot[1])
    else:
        params.device.event_sync(slot)
    local_dt = float(grid.dt_host.array[posted_in_cycle & 1])
    if grid.comm is not None and getattr(grid.comm, "stream_ordered", False):
        return local_dt                                           # already the minimum over the ranks
    return global_min(params, local_dt)


def _checkpoint(params, grid, label, axis=Axis.X):
    """ref @checkpoint, src/solver.jl:40-55 → src/io.jl:185-227 (compare / is_ref options)"""
    if not params.compare:
        return False
    from .io import step_checkpoint
    return step_checkpoint(params, grid, label, axis.name)


def solver_cycle(params, grid, last_cycle=True):
    """ref src/solver.jl:288-320 — returns True when a step comparison (``compare``) found a difference.
    ``last_cycle`` (fused path only): materialise p (the reference's saved_vars hold the EOS of the state
    before the last sweep) after this cycle."""
    gdt = grid.global_dt
    deferred = (gdt.cycle - 1) in grid.dt_inflight      # fused path, cycle >= 1: this cycle's dt is already known
    if gdt.cycle == 0:
        if _checkpoint(params, grid, "init_test"):
            return True
        update_EOS(params, grid)
        if _checkpoint(params, grid, "EOS_init"):
            return True
    if not deferred:
        next_time_step(params, grid)
        if params.use_fused_sweep and gdt.cycle == 0:
            grid.release_scratch()               # c, g: only the EOS + dtCFL of cycle 0 needed them
    if _checkpoint(params, grid, "time_step"):
        return True
    if params.use_MPI and getattr(grid.comm, "native_cycle", False) and not params.compare:
        from .multi_tile import native_cycle_usable
        if native_cycle_usable(params):
            # the library's own group: exchanges, sweeps and the all-reduce of the next CFL step in ONE call
            # (armon_hip_mgpu_cycle); what is left for the host is the read-back of that scalar, one cycle late
            slot = grid.comm.cycle(gdt, last_cycle)
            if slot is not None:
                grid.dt_inflight[gdt.cycle] = ("edge", slot)
                if deferred:
                    gdt.update_dt(take_dt_readback(params, grid, gdt.cycle - 1))
            return False
    sweeps = split_axes(params.axis_splitting, gdt.cycle)
    for k, (axis, dt_factor) in enumerate(sweeps):
        # update_solver_state!: ref src/solver_state.jl:339-345
        i_ax = int(axis) - 1
        dx = params.cell_size(i_ax)
        dt = gdt.current_dt * params.T(dt_factor)
        if params.use_fused_sweep:
            # The last sweep of a cycle also reduces the next cycle's CFL step (post-sweep u, v with its
            # own pre-sweep c: what the reference's dtCFL_kernel reads) and, on the final
            # cycle, materialises the pre-sweep p that the reference leaves in memory.
            last = k == len(sweeps) - 1
            sweep = fused_sweep_overlapped if params.use_MPI else fused_sweep
            sweep(params, grid, axis, dt, dx, emit_p=last and last_cycle, emit_dt=last and not params.cst_dt)
            if last and not last_cycle:
                prefetch_halo(params, grid)
            if last and not params.cst_dt:
                post_dt_readback(params, grid)
                if deferred:
                    gdt.update_dt(take_dt_readback(params, grid, gdt.cycle - 1))
        else:
            update_EOS(params, grid, axis)
            if _checkpoint(params, grid, "EOS", axis):
                return True
            block_ghost_exchange(params, grid, axis)
            if _checkpoint(params, grid, "boundary_conditions", axis):
                return True
            numerical_fluxes(params, grid, axis, dt, dx)
            if _checkpoint(params, grid, "numerical_fluxes", axis):
                return True
            cell_update(params, grid, axis, dt, dx)
            if _checkpoint(params, grid, "cell_update", axis):
                return True
            projection_remap(params, grid, axis, dt, dx)
            if _checkpoint(params, grid, "projection_remap", axis):
                return True
    return False


# ---- graph replay of the cycle (device-resident time step) ----------------------------------------------------------------
GRAPH_BATCH = 8            # cycles enqueued between two looks at the device state
GRAPH_EVENT_SLOT = 1014    # event-pool slots 1014, 1015


def graph_cycles_usable(params):
    """The cycle can be captured once and replayed when nothing on the host has to happen inside it: one block, fused
    sweeps, no per-cycle output (conservation print-outs, animation frames, step comparisons)."""
    if not (params.use_fused_sweep and getattr(params, "graph_cycles", False)):
        return False
    if params.use_MPI or any(n != PROC_NULL for n in params.neighbours.values()):
        return False
    if not getattr(params.device, "owns_ctx", True):      # a tile context of a group: its stream is the group's, not ours to capture
        return False
    return params.silent > 1 and params.animation_step == 0 and not params.compare and not params.kernel_callbacks


def _graph_sweep_descs(params, grid, cycle_parity, state_ptr):
    """Descriptors of one cycle's sweeps with the time step read from the device state (dt = the split factor)."""
    descs = []
    sweeps = split_axes(params.axis_splitting, cycle_parity)
    for k, (axis, dt_factor) in enumerate(sweeps):
        last = k == len(sweeps) - 1
        d = sweep_desc(params, grid, axis, float(params.T(dt_factor)), params.cell_size(int(axis) - 1),
                       emit_p=last, emit_dt=last and not params.cst_dt)
        d.dt_state = state_ptr
        descs.append(d)
        grid.swap_state()
    return descs


def time_loop_graph(params, grid, _after_handover=None):
    """The time loop with the dt state machine on the device and each cycle replayed from a hipGraph
    (include/armon_hip.h: armon_dt_state, armon_hip_dt_state_step, armon_hip_graph_*). Cycles 0 and 1 run host-driven as
    in ``time_loop`` — cycle 0 needs the staged dtCFL kernel, and everything a capture records must have run once —, then the
    device owns (cycle, time, current_dt): a cycle is ONE host call, the host reads the state back every GRAPH_BATCH
    cycles without stopping the stream. Same bits, same cycle count, same final time as the host-driven loop (tests)."""
    dev, gdt, T = params.device, grid.global_dt, params.T
    gdt.reset()
    grid.dt_inflight.clear()
    params.wait()
    t1 = _time.perf_counter_ns()
    maxtime = T(params.maxtime)

    def ends():
        if params.cst_dt:
            return T(gdt.time + gdt.current_dt) >= maxtime or gdt.cycle + 1 >= params.maxcycle
        return gdt.cycle + 1 >= params.maxcycle or gdt.current_dt == 0 or T(gdt.time + gdt.current_dt) >= maxtime

    # host-driven start: cycles 0 and 1 (the second one consumes the first deferred CFL step)
    while gdt.time < maxtime and gdt.cycle < params.maxcycle and gdt.cycle < 2:
        if solver_cycle(params, grid, last_cycle=ends()):
            break
        gdt.next_cycle()
    n_parities = 1 if params.axis_splitting in ("Sequential", "X_only", "Y_only") else 2
    swaps_per_cycle = len(split_axes(params.axis_splitting, 0))
    if gdt.time < maxtime and gdt.cycle < params.maxcycle:
        # hand the state machine over: the CFL step of the state cycle `gdt.cycle` starts from was posted by the previous one
        L_prev = 0.0
        if not params.cst_dt:
            L_prev = take_dt_readback(params, grid, gdt.cycle - 1)
        # auto_step: the fold of the last sweep's dt reduction steps the state machine itself (one kernel less per cycle);
        # with a constant time step there is no reduction to ride on and the step stays a kernel of its own
        auto = not params.cst_dt
        st = _lib.DtState(current_dt=float(gdt.current_dt), time=float(gdt.time), L_prev=float(L_prev), cycle=gdt.cycle,
                          done=0, invalid=0, emit_p=int(ends()), auto_step=int(auto), cst_dt=int(params.cst_dt),
                          maxcycle=params.maxcycle, cfl=float(params.cfl), maxtime=float(params.maxtime), Dt=float(params.Dt))
        state = dev.empty(C.sizeof(_lib.DtState) // 8, np.float64)
        host_state = dev.pinned(C.sizeof(_lib.DtState) // 8 * 2, np.float64)
        check(dev._L.armon_hip_memcpy(dev.ctx, C.c_void_p(state.ptr), C.byref(st), C.sizeof(st), 1))
        step = params.fn("dt_state_step")
        L_new = C.c_void_p(grid.dt_scalar.ptr)
        # the step kernel runs once outside any capture, on a state that is already `done` (a no-op)
        dummy = dev.empty(C.sizeof(_lib.DtState) // 8, np.float64)
        over = _lib.DtState(done=1)
        check(dev._L.armon_hip_memcpy(dev.ctx, C.c_void_p(dummy.ptr), C.byref(over), C.sizeof(over), 1))
        check(step(dev.ctx, C.c_void_p(dummy.ptr), L_new, float(params.cfl), float(params.maxtime), params.maxcycle,
                   int(params.cst_dt), float(params.Dt)))
        params.wait()
        dummy.free()

        def enqueue_cycle(parity):
            for d in _graph_sweep_descs(params, grid, parity, state.ptr):
                check(params.fn("sweep")(dev.ctx, C.byref(d)))
            if not auto:
                check(step(dev.ctx, C.c_void_p(state.ptr), L_new, float(params.cfl), float(params.maxtime), params.maxcycle,
                           int(params.cst_dt), float(params.Dt)))

        # One graph per cycle parity, and per parity of the ping-pong (a cycle of an odd number of sweeps leaves the state in
        # the other set of vectors): captured lazily, keyed by (parity of the cycle, which set holds the state).
        graphs, origin = {}, grid.data["rho"].ptr

        def launch_cycle(cycle):
            key = (cycle % n_parities, grid.data["rho"].ptr == origin)
            if key not in graphs:
                check(dev._L.armon_hip_graph_begin(dev.ctx))
                held, g = grid.data["rho"].ptr, C.c_void_p()
                try:
                    enqueue_cycle(cycle)
                except BaseException:
                    # leave no capture open, no graph behind and the ping-pong where it was: a caller that catches the
                    # error finds the state it had before this cycle
                    if dev._L.armon_hip_graph_end(dev.ctx, C.byref(g)) == 0 and g:
                        dev._L.armon_hip_graph_destroy(g)
                    if grid.data["rho"].ptr != held:
                        grid.swap_state()
                    raise
                check(dev._L.armon_hip_graph_end(dev.ctx, C.byref(g)))
                graphs[key] = g
                # the capture only recorded the cycle: undo its ping-pong bookkeeping, then replay it for real below
                for _ in range(swaps_per_cycle):
                    grid.swap_state()
            check(dev._L.armon_hip_graph_launch(dev.ctx, graphs[key]))
            for _ in range(swaps_per_cycle):
                grid.swap_state()

        def read_state(slot):
            check(dev._L.armon_hip_memcpy_async(dev.ctx, C.c_void_p(host_state.ptr + slot * C.sizeof(_lib.DtState)),
                                               C.c_void_p(state.ptr), C.sizeof(_lib.DtState), 2))
            dev.event_record(GRAPH_EVENT_SLOT + slot)

        def state_at(slot):
            dev.event_sync(GRAPH_EVENT_SLOT + slot)
            return _lib.DtState.from_buffer_copy(bytes(host_state.array.view(np.uint8)[slot * C.sizeof(_lib.DtState):
                                                                                       (slot + 1) * C.sizeof(_lib.DtState)]))

        # Replay. The host does not know which cycle is the last one: it enqueues batches and looks at the state of the
        # PREVIOUS batch while the current one runs; cycles enqueued past the end are no-ops on the device (`done`), but
        # the ping-pong bookkeeping above assumed they ran, so the final state's set is recomputed from the cycle count.
        if _after_handover is not None:        # test hook: the device owns the time step from here on
            _after_handover(params, grid)
        cycle, batch, final = gdt.cycle, 0, None
        start_cycle, start_in_origin = gdt.cycle, grid.data["rho"].ptr == origin
        try:
            while final is None:
                for _ in range(GRAPH_BATCH):
                    launch_cycle(cycle)
                    cycle += 1
                read_state(batch & 1)
                if batch > 0:
                    prev = state_at((batch - 1) & 1)
                    if prev.done:
                        final = prev
                batch += 1
                if final is None and cycle - start_cycle > params.maxcycle + 2 * GRAPH_BATCH:
                    # every cycle the run could hold has been enqueued and the device still does not say `done`: the state
                    # machine is not stepping (it cannot happen with the kernels as they are; it must not pass for a result)
                    last = state_at((batch - 1) & 1)
                    if not last.done:
                        solver_error("time", f"the device-resident time loop did not finish within maxcycle = {params.maxcycle} "
                                             f"cycles (device state: cycle {last.cycle}, time {last.time}, dt {last.current_dt})")
                    final = last
            last = state_at((batch - 1) & 1)
            final = last if last.done else final
        except BaseException:
            params.wait()                 # no copy into the pinned slots is in flight any more
            host_state.free()
            raise
        finally:
            params.wait()
            for g in graphs.values():
                dev._L.armon_hip_graph_destroy(g)
            state.free()
        if final.invalid:
            host_state.free()
            solver_error("time", f"Invalid time step for cycle {final.invalid_cycle}: {final.invalid_value}")
        # where the state really is: cycles [start_cycle, final.cycle) ran, the later ones were no-ops
        ran = final.cycle - start_cycle
        in_origin = start_in_origin if (ran * swaps_per_cycle) % 2 == 0 else not start_in_origin
        if (grid.data["rho"].ptr == origin) != in_origin:
            grid.swap_state()
        gdt.cycle, gdt.time, gdt.current_dt = int(final.cycle), T(final.time), T(final.current_dt)
        grid.graph_report = {"graphs": len(graphs), "cycles_replayed": int(ran), "cycles_enqueued": cycle - start_cycle}
        host_state.free()
    params.wait()
    t2 = _time.perf_counter_ns()
    solve_time = t2 - t1
    cells = params.N[0] * params.N[1]
    grind_time = solve_time / max(gdt.cycle * cells, 1)
    if params.is_root and params.silent < 3:
        print(" ")
        print(f"Total time:  {solve_time / 1e9:.5f} sec")
        print(f"Grind time:  {grind_time / 1e3:.5f} µs/cell/cycle")
        print(f"Cells/sec:   {1 / grind_time * 1e3:.5f} Mega cells/sec")
        print(f"Cycles:      {gdt.cycle}")
        print(f"Last cycle:  {gdt.time:.18f} sec, Δt={gdt.current_dt:.18f} sec")
    return float(gdt.time), float(gdt.current_dt), gdt.cycle, 1 / grind_time, solve_time


def time_loop(params, grid):
    """ref src/solver.jl:323-403 → (time, dt, cycles, cells_per_ns, solve_time_ns)"""
    if graph_cycles_usable(params):
        return time_loop_graph(params, grid)
    grid.global_dt.reset()
    grid.dt_inflight.clear()
    gdt = grid.global_dt
    params.wait()
    t1 = _time.perf_counter_ns()
    maxtime = params.T(params.maxtime)
    while gdt.time < maxtime and gdt.cycle < params.maxcycle:
        if params.cst_dt:
            ends = params.T(gdt.time + gdt.current_dt) >= maxtime or gdt.cycle + 1 >= params.maxcycle
        else:
            # the cycle's dt is only known after next_time_step on cycle 0: be conservative there
            ends = (gdt.cycle + 1 >= params.maxcycle or gdt.current_dt == 0
                    or params.T(gdt.time + gdt.current_dt) >= maxtime)
        # animation frames (ref :373-378) are written after next_cycle! when (cycle - 1) % animation_step == 0; the fused
        # path only materialises p (a saved var) on request, so a cycle that ends with a frame asks for it like the last one
        frame_due = params.animation_step != 0 and gdt.cycle % params.animation_step == 0
        if solver_cycle(params, grid, last_cycle=ends or frame_due):
            break
        gdt.next_cycle()
        if frame_due:
            from .io import write_animation_frame
            params.wait()
            write_animation_frame(params, grid)
        if params.silent <= 1:
            params.wait()
            mass, energy = conservation_vars(params, grid)
            if params.is_root:
                dM = abs(params.initial_mass - mass) / params.initial_mass * 100
                dE = abs(params.initial_energy - energy) / params.initial_energy * 100
                print(f"Cycle {gdt.cycle:4d}: dt = {gdt.current_dt:.18f}, t = {gdt.time:.18f}, "
                      f"|ΔM| = {dM:#8.6g}%, |ΔE| = {dE:#8.6g}%")
    drain_halo(grid)
    params.wait()   # "Last fence"
    t2 = _time.perf_counter_ns()
    solve_time = t2 - t1
    cells = params.N[0] * params.N[1]
    grind_time = solve_time / max(gdt.cycle * cells, 1)
    if params.is_root and params.silent < 3:
        print(" ")
        print(f"Total time:  {solve_time / 1e9:.5f} sec")
        print(f"Grind time:  {grind_time / 1e3:.5f} µs/cell/cycle")
        print(f"Cells/sec:   {1 / grind_time * 1e3:.5f} Mega cells/sec")
        print(f"Cycles:      {gdt.cycle}")
        print(f"Last cycle:  {gdt.time:.18f} sec, Δt={gdt.current_dt:.18f} sec")
    return float(gdt.time), float(gdt.current_dt), gdt.cycle, 1 / grind_time, solve_time


def armon(params):
    """ref src/solver.jl:411-516 — main entry point, returns SolverStats."""
    if params.is_root and params.silent < 3:
        print(params)
    if params.animation_step != 0:
        from .io import prepare_animation_dir
        prepare_animation_dir(params)
    grid = BlockGrid(params)
    if params.use_MPI:
        from .halo_exchange import setup
        setup(params, grid)
    init_test(params, grid)
    if params.check_result or params.silent <= 1:
        params.initial_mass, params.initial_energy = conservation_vars(params, grid)
    final_time, dt, cycles, cells_per_ns, solve_time = time_loop(params, grid)
    if params.check_result and params.test.conservative:
        mass, energy = conservation_vars(params, grid)
        if params.is_root:
            dM = abs(params.initial_mass - mass)
            dE = abs(params.initial_energy - energy)
            if not (dM <= 1e-12 and dE <= 1e-12):   # ref src/solver.jl:484-501 (comparison_tolerance)
                solver_error("check", f"mass or energy are not conserved: |ΔM| = {dM}, |ΔE| = {dE}")
    if params.write_output:
        from .io import write_sub_domain_file
        write_sub_domain_file(params, grid, params.output_file)
    if params.write_slices:
        from .io import write_slices_files
        write_slices_files(params, grid, params.output_file)
    stats = SolverStats(final_time, dt, cycles, solve_time / 1e9, params.N[0] * params.N[1], cells_per_ns)
    if params.return_data:
        stats.data = grid
    return stats
