"""``ArmonParameters`` — the reference's configuration surface (ref src/parameters.jl) for the HIP backend.

Same keyword names, defaults and validation as ``ArmonParameters(; kwargs...)``: every ``init_*`` step
consumes its own options and anything left over is an error (ref src/parameters.jl:349-372). Options
that only steer the reference's CPU machinery (threads, NUMA, cache blocking, logging) are accepted and
recorded so that a reference script runs unchanged, but have no effect on the device path.
"""
import numpy as np

from ._lib import solver_error
from .blocking import Axis, BlockSize, Side, compute_steps_ranges
from .test_cases import create_test, default_domain_origin, default_domain_size

SCHEMES = {"Godunov": 0, "GAD": 1}                       # ref src/riemann_schemes.jl:5-6
SCHEME_STENCIL = {"Godunov": 1, "GAD": 2}                # ref :17-18
PROJECTIONS = {"euler": 0, "euler_2nd": 1}               # ref src/projection_schemes.jl:4-5
PROJECTION_STENCIL = {"euler": 1, "euler_2nd": 2}        # ref :11-12
LIMITERS = {"no_limiter": 0, "minmod": 1, "superbee": 2}  # ref src/limiters.jl:10-12
SPLITTINGS = ("Sequential", "Godunov", "SequentialSym", "Strang", "X_only", "Y_only")  # ref src/axis_splitting.jl:7-12

PROC_NULL = -1


class ArmonParameters:
    def __init__(self, *, data_type=np.float64, N=(10, 10), **options):
        self.data_type = np.dtype(data_type)
        if self.data_type not in (np.dtype(np.float64), np.dtype(np.float32)):
            solver_error("config", f"unsupported data_type {data_type}: Float64 or Float32")
        self.T = self.data_type.type          # scalar constructor: host-side arithmetic is done in T, like the reference
        self.suffix = "_f32" if self.data_type == np.float32 else ""
        if len(N) != 2:
            solver_error("config", "only 2-D domains are supported")
        self.N = tuple(int(n) for n in N)
        options = self._get_device(**options)
        options = self._init_scheme(**options)
        options = self._init_test(**options)
        options = self._init_MPI(**options)
        options = self._init_device(**options)
        options = self._init_profiling(**options)
        options = self._init_indexing(**options)
        options = self._init_output(**options)
        options = self._init_backend(**options)
        if options:   # ref src/parameters.jl:369-372
            names = ", ".join(f"'{k}'" for k in options)
            raise ValueError(f"{len(options)} unconsumed options:\n{names}")

    # ref src/parameters.jl:392-405 — this backend IS the device: use_gpu defaults to True here
    def _get_device(self, device="HIP_native", use_gpu=True, use_kokkos=False, **options):
        if use_kokkos:
            solver_error("config", "use_kokkos is not available in the HIP backend")
        if not use_gpu:
            solver_error("config", "the HIP backend has no CPU path (use_gpu=false)")
        if str(device) not in ("HIP_native", "ROCM"):
            solver_error("config", f"Unknown device: '{device}' (this backend provides :HIP_native)")
        self.device_name = str(device)
        self.use_gpu = True
        self.use_kokkos = False
        return options

    # ref src/parameters.jl:577-630
    def _init_scheme(self, scheme="GAD", projection="euler_2nd", riemann_limiter="minmod",
                     axis_splitting="Sequential", nghost=4, cst_dt=False, Dt=0.,
                     dt_on_even_cycles=False, **options):
        scheme, projection = str(scheme), str(projection)
        riemann_limiter, axis_splitting = str(riemann_limiter), str(axis_splitting)
        if projection not in PROJECTIONS:
            solver_error("config", f"Unknown scheme: '{projection}'")
        if scheme not in SCHEMES:
            solver_error("config", f"Unknown scheme: '{scheme}'")
        if axis_splitting not in SPLITTINGS:
            solver_error("config", f"Unknown splitting method: '{axis_splitting}'")
        if riemann_limiter not in LIMITERS:
            solver_error("config", f"Unknown limiter name: '{riemann_limiter}'")
        # The reference asks for stencil(scheme)·stencil(projection) ghosts (ref src/parameters.jl:609-613),
        # but its own step ranges (ref :1005-1014) compute fluxes up to interface N+w+1, whose stencil
        # reads cell N+w+stencil(scheme): with fewer than w+stencil(scheme) ghost layers that read leaves
        # the block (undefined behaviour under @inbounds in the reference, a memory fault on a GPU).
        # Both rules agree for the default GAD + euler_2nd (4); the stricter one is enforced here.
        min_nghost = max(SCHEME_STENCIL[scheme] * PROJECTION_STENCIL[projection],
                         SCHEME_STENCIL[scheme] + PROJECTION_STENCIL[projection])
        if nghost < min_nghost:
            solver_error("config", "Not enough ghost cells for the riemann solver and projection, "
                                   f"at least {min_nghost} are needed, got {nghost}")
        if cst_dt and Dt == 0:
            solver_error("config", "Dt == 0 with constant step enabled")
        if dt_on_even_cycles:
            # ref src/reductions.jl:165-170 skips update_dt! on odd cycles, after which next_cycle!
            # (ref src/solver_state.jl:158-161) rejects the Ready state: unusable in the sync solver.
            solver_error("config", "dt_on_even_cycles is not supported by the synchronous solver cycle")
        self.nghost = int(nghost)
        self.riemann_scheme = scheme
        self.projection_scheme = projection
        self.riemann_limiter = riemann_limiter
        self.axis_splitting = axis_splitting
        self.cst_dt = bool(cst_dt)
        self.Dt = float(Dt)
        self.dt_on_even_cycles = False
        return options

    # ref src/parameters.jl:632-670
    def _init_test(self, test="Sod", domain_size=None, origin=None, cfl=0., maxtime=0.,
                   maxcycle=500_000, **options):
        name = str(test)
        if domain_size is None:
            domain_size = default_domain_size(name)
        self.domain_size = tuple(float(d) for d in domain_size)
        if origin is None:
            origin = default_domain_origin(name)
        self.origin = tuple(float(o) for o in origin)
        T = self.T
        dX = (T(self.domain_size[0]) / T(self.N[0]), T(self.domain_size[1]) / T(self.N[1]))
        self.test = create_test(name, dX, T)
        self.maxcycle = int(maxcycle)
        self.cfl = float(cfl) if cfl != 0 else self.test.cfl
        self.maxtime = float(maxtime) if maxtime != 0 else self.test.maxtime
        return options

    # ref src/parameters.jl:408-467. The MPI transport is replaced by torch.distributed (RCCL on GPUs,
    # gloo in CPU tests): `use_MPI=True` needs an initialised process group, `global_comm` may carry a
    # torch ProcessGroup. Default is False here because a single process owns a single GPU.
    def _init_MPI(self, use_MPI=False, P=(1, 1), reorder_grid=True, global_comm=None, gpu_aware=True,
                  tile_of=None, periodic=(False, False), **options):
        """``tile_of=(rank, (px, py))``: this parameter set describes ONE tile of a px×py grid whose tiles all live in
        this process (``multi_tile.TileGroup`` over ``armon_hip_mgpu_init``): same partition and neighbours as an MPI
        rank of the reference, no process group involved. ``periodic=(x, y)``: TEST AID for the transport (the
        reference's ``MPI.Cart_create`` is never periodic, ref src/parameters.jl:432) — neighbours wrap around along
        that axis, so a 1×1 rank is its own neighbour and one GPU runs the real send/recv calls
        (``armon_hip_mgpu_set_periodic``); only the library's native exchange supports it."""
        if tile_of is not None:
            P = tile_of[1]
        self.periodic = (bool(periodic[0]), bool(periodic[1]))
        if len(P) != len(self.N):
            solver_error("config", f"Mismatched dimensions: expected a grid of {len(self.N)} processes, got: {len(P)}")
        self.use_MPI = bool(use_MPI)
        self.reorder_grid = reorder_grid
        self.gpu_aware = gpu_aware
        self.global_comm = global_comm
        if self.use_MPI:
            import torch.distributed as dist
            if not dist.is_initialized():
                solver_error("config", "'use_MPI=true' but the process group has not yet been initialized")
            self.rank = dist.get_rank(global_comm)
            self.proc_size = dist.get_world_size(global_comm)
            self.proc_dims = tuple(int(p) for p in P)
            if self.proc_dims[0] * self.proc_dims[1] != self.proc_size:
                solver_error("config", f"could not create a {P[0]}×{P[1]} cartesian topology "
                                       f"using {self.proc_size} processes")
            self.cart_coords = cart_coords(self.rank, self.proc_dims)
            self.neighbours = cart_neighbours(self.cart_coords, self.proc_dims, self.periodic)
        elif tile_of is not None:
            self.proc_dims = tuple(int(p) for p in P)
            self.rank = int(tile_of[0])
            self.proc_size = self.proc_dims[0] * self.proc_dims[1]
            if not 0 <= self.rank < self.proc_size:
                solver_error("config", f"tile {self.rank} is not part of a {P[0]}×{P[1]} grid")
            self.cart_coords = cart_coords(self.rank, self.proc_dims)
            self.neighbours = cart_neighbours(self.cart_coords, self.proc_dims, self.periodic)
        else:
            if any(self.periodic):
                solver_error("config", "periodic=... needs use_MPI=true or a tile group: it describes the process grid")
            self.rank = 0
            self.proc_size = 1
            self.proc_dims = (1, 1)
            self.cart_coords = (0, 0)
            self.neighbours = {s: PROC_NULL for s in Side}
        self.root_rank = 0
        self.is_root = self.rank == self.root_rank
        return options

    # ref src/parameters.jl:470-530 — CPU machinery knobs: accepted, inert on the device path.
    def _init_device(self, use_threading=True, use_simd=True, block_size=None, use_cache_blocking=False,
                     async_cycle=False, use_two_step_reduction=False, workload_distribution="simple",
                     distrib_params=None, numa_aware=False, lock_memory=False, busy_wait_limit=100,
                     **options):
        if use_cache_blocking:
            # ref src/blocking/block_grid.jl:352-355: without cache blocking the grid is one block.
            solver_error("config", "use_cache_blocking is a CPU cache optimisation; the HIP backend keeps one block per GPU")
        if async_cycle:
            solver_error("config", "async_cycle (CPU thread state machine) is replaced by HIP streams here")
        self.use_threading, self.use_simd = use_threading, use_simd
        self.use_cache_blocking, self.async_cycle = False, False
        # ref src/reductions.jl:70-78,276-284: "two steps" = per-cell values into a temporary array, then a plain reduce of
        # it, for GPU backends whose one-step mapreduce misbehaves. Every reduction of this backend already IS two-step
        # (one partial per workgroup, then a fold kernel; reductions.hip) and deterministic, so both settings select the
        # same, compliant behaviour: the option is satisfied, not dropped.
        self.use_two_step_reduction = bool(use_two_step_reduction)
        self.kernel_block_size = block_size
        return options

    # ref src/parameters.jl:532-575
    def _init_profiling(self, profiling=(), measure_time=True, time_async=True, log_blocks=False,
                        estimated_blk_log_size=0, **options):
        self.profiling = tuple(profiling)
        self.kernel_callbacks = []      # ref register_kernel_callback, src/profiling.jl:30-68
        self.measure_time = measure_time
        self.time_async = time_async
        return options

    # ref src/parameters.jl:673-697
    def _init_indexing(self, **options):
        self.global_grid = self.N
        P, cc, g = self.proc_dims, self.cart_coords, self.global_grid
        self.N = tuple(g[d] // P[d] + (g[d] % P[d] if cc[d] == P[d] - 1 else 0) for d in range(2))
        if any(P[d] > 1 and self.N[d] < self.nghost for d in range(2)):
            solver_error("config", f"domain {g} is too small to be split by {P} processes while keeping "
                                   f"more than {self.nghost} cells along each axis")
        self.N_origin = tuple(cc[d] * (g[d] // P[d]) + 1 for d in range(2))
        self.block_size = BlockSize((self.N[0] + 2 * self.nghost, self.N[1] + 2 * self.nghost), self.nghost)
        w = PROJECTION_STENCIL[self.projection_scheme]
        self.steps_ranges = {ax: compute_steps_ranges(ax, self.nghost, w) for ax in (Axis.X, Axis.Y)}
        return options

    # ref src/parameters.jl:700-748
    def _init_output(self, silent=0, output_dir=".", output_file="output", write_output=False,
                     write_ghosts=False, write_slices=False, output_precision=None, animation_step=0,
                     compare=False, is_ref=False, comparison_tolerance=1e-10, check_result=False,
                     return_data=False, **options):
        self.compare, self.is_ref = bool(compare), bool(is_ref)
        self.comparison_tolerance = float(comparison_tolerance)
        self.silent = silent
        self.output_dir, self.output_file = output_dir, output_file
        self.write_output, self.write_ghosts = write_output, write_ghosts
        self.write_slices = bool(write_slices)          # 3 more files at the end of armon(): io.write_slices_files
        self.animation_step = int(animation_step)       # a frame every `animation_step` cycles: solver.time_loop
        if self.animation_step < 0:
            solver_error("config", f"animation_step must be >= 0, got {animation_step}")
        self.output_precision = 17 if output_precision is None else int(output_precision)
        self.check_result = check_result
        self.return_data = return_data
        self.initial_mass = 0.
        self.initial_energy = 0.
        return options

    # ref src/parameters.jl:766-778 — backend specific options (like ext/ArmonKokkos.jl:83-87)
    def _init_backend(self, device_id=None, use_fused_sweep=True, exact_arithmetic=False, stream=None,
                      placement_tries=24, stream_ordered_halo=True, ctx=None, native_halo=True, overlap_halo=True, edge_stream=True,
                      placement_min_bytes=256 << 20, placement_rounds=8, graph_cycles=False, native_cycle=True, **options):
        """``device_id``: GPU ordinal (default LOCAL_RANK or 0). ``use_fused_sweep``: run each sweep as
        the fused HIP kernel instead of the 5 staged kernels. ``exact_arithmetic=True``: IEEE division/sqrt
        and no FMA contraction in the fused sweep — bit-identical to the staged path and to the CPU oracle, every bit,
        subnormal values included (quotients that can fall below the normal range take the IEEE expansion; the
        unguarded form of rounds 1-3, one subnormal unit off in 1 run of 800, is the build flag -DARMON_LOOSE_SUBNORMAL);
        the default tuned arithmetic (shared 1-ulp reciprocals, FMAs) stays within the reference's own
        golden-file tolerance (atol 1e-13, rtol 4 eps on the Sod family). ``placement_tries``: how many
        placements of the state vectors in HBM ``init_test`` may try (0/1 = take the first; see
        ``BlockGrid.tune_placement``)."""
        import os
        if device_id is None:
            device_id = int(os.environ.get("LOCAL_RANK", "0")) if self.use_MPI else 0
        self.device_id = int(device_id)
        self._stream = stream
        self._ctx = ctx         # an existing armon_ctx to adopt (a tile context owned by an armon_mgpu group)
        self.native_halo = bool(native_halo)    # N>1 over RCCL: halo exchange / dt all-reduce by the library itself
        self.overlap_halo = bool(overlap_halo)  # sweep the interior while the halos travel
        self.edge_stream = bool(edge_stream)    # native groups: unpack + boundary strips on the transfer stream, concurrent with the interior
        self._device = None     # created on first use, so that configuration errors need no GPU
        # per-step dumps / comparisons need the intermediate states: only the staged path has them
        self.use_fused_sweep = bool(use_fused_sweep) and not self.compare
        self.exact_arithmetic = bool(exact_arithmetic)
        self.placement_tries = int(placement_tries)
        self.placement_min_bytes = int(placement_min_bytes)     # vectors smaller than this are not worth placing
        self.placement_rounds = int(placement_rounds)           # batches of spares a search may go through
        # dt state machine on the device + each cycle replayed from a hipGraph (solver.time_loop_graph): one host call per
        # cycle instead of one per kernel plus a scalar read-back — what small grids are bound by
        self.graph_cycles = bool(graph_cycles)
        # tile groups / RCCL ranks: a whole cycle (exchanges, sweeps, dt minimum) enqueued by ONE library call
        # (armon_hip_mgpu_cycle) instead of call by call from this host mirror
        self.native_cycle = bool(native_cycle)
        self.stream_ordered_halo = bool(stream_ordered_halo)   # RCCL: order the exchange on the stream, no host syncs
        self.shared_stream = False
        self.backend_options = dict(device_id=device_id, use_fused_sweep=self.use_fused_sweep,
                                    exact_arithmetic=self.exact_arithmetic)
        return options

    # ---- hooks ---------------------------------------------------------------------------------
    @property
    def device(self):
        """``create_device(Val(:HIP_native))`` (ref src/parameters.jl:751-755): context on first use."""
        if self._device is None:
            from .device import HIPDevice
            if self._ctx is not None:
                self._device = HIPDevice.adopt(self._ctx, self.device_id)
                return self._device
            stream = self._stream
            if stream is None and self.use_MPI and self.stream_ordered_halo:
                stream = self._torch_stream_for_rccl()
            self._device = HIPDevice(self.device_id, stream)
        return self._device

    def _torch_stream_for_rccl(self):
        """With the RCCL backend every kernel of this rank runs on ONE torch stream (made current): torch's
        point-to-point ops then order themselves after the pack kernels and the unpack kernels after the
        receives on the device, and the halo exchange needs no host synchronisation at all."""
        try:
            import torch
            import torch.distributed as dist
            if not (dist.is_initialized() and dist.get_backend(self.global_comm) == "nccl"):
                return None
            torch.cuda.set_device(self.device_id)
            self._torch_stream = torch.cuda.Stream(device=self.device_id)
            torch.cuda.set_stream(self._torch_stream)
            self.shared_stream = True
            return self._torch_stream.cuda_stream
        except ImportError:
            return None

    def fn(self, name):
        """The C-ABI entry point ``armon_hip_<name>`` for this run's data_type (``_f32`` suffix for Float32)."""
        from . import _lib
        L = self._device._L if self._device is not None else _lib.lib()      # the library this run's context came from
        return getattr(L, "armon_hip_" + name + self.suffix)

    def cell_size(self, i_ax):
        """Global cell size along an axis, computed in T (ref src/solver_state.jl:341, src/reductions.jl:92)."""
        return self.T(self.domain_size[i_ax]) / self.T(self.global_grid[i_ax])

    def wait(self):
        """``Base.wait(params)`` (ref src/parameters.jl:1031-1038)."""
        self.device.wait()

    def device_memory_info(self):
        return self.device.memory_info()

    def data_type_size(self):
        return self.data_type.itemsize

    def __repr__(self):   # ref print_parameters, src/parameters.jl:805-918 (abridged)
        return (f"ArmonParameters(test={self.test.name}, N={self.N} of {self.global_grid}, "
                f"scheme={self.riemann_scheme}, limiter={self.riemann_limiter}, "
                f"projection={self.projection_scheme}, splitting={self.axis_splitting}, "
                f"nghost={self.nghost}, cfl={self.cfl}, maxtime={self.maxtime}, maxcycle={self.maxcycle}, "
                f"P={self.proc_dims}, coords={self.cart_coords}, device_id={self.device_id})")


def cart_coords(rank, dims):
    """MPI_Cart_coords for a non-periodic row-major grid (last dimension varies fastest)."""
    return (rank // dims[1], rank % dims[1])


def cart_rank(coords, dims, periodic=(False, False)):
    coords = tuple(c % dims[d] if periodic[d] else c for d, c in enumerate(coords))
    if not all(0 <= coords[d] < dims[d] for d in range(2)):
        return PROC_NULL
    return coords[0] * dims[1] + coords[1]


def cart_neighbours(coords, dims, periodic=(False, False)):
    """ref src/parameters.jl:441-447 (MPI.Cart_shift along dim 0 = X, dim 1 = Y); ``periodic``: test aid, see ``_init_MPI``."""
    cx, cy = coords
    return {
        Side.Left: cart_rank((cx - 1, cy), dims, periodic),
        Side.Right: cart_rank((cx + 1, cy), dims, periodic),
        Side.Bottom: cart_rank((cx, cy - 1), dims, periodic),
        Side.Top: cart_rank((cx, cy + 1), dims, periodic),
    }


def memory_required(N, nghost=4, data_type=np.float64, fused=True, transient=False):
    """Device bytes for one block (ref ``memory_required``, src/blocking/block_grid.jl): the 16 ``BlockData`` vectors on the
    staged path; on the fused path the 7 it touches (x, y, rho, u, v, E, p) + 4 ping-pong partners of the state — the
    other nine (us, ps, work_1..4, mask; c, g after the first cycle) are only allocated when something reads them.
    ``transient=True``: the peak instead — + c, g during cycle 0, or the 8 spare vectors of the placement search of
    ``init_test`` (blocks whose vectors reach 256 MiB), whichever is larger."""
    n = (N[0] + 2 * nghost) * (N[1] + 2 * nghost) * np.dtype(data_type).itemsize
    if not fused:
        return n * 16
    spares = 8 if n >= (256 << 20) else 2
    return n * (11 + (spares if transient else 0))


def proc_grid_for(world):
    """Process grid used by the benchmarks: 1→(1,1), 2→(2,1), 4→(2,2), 8→(4,2) (BASELINE.json configs);
    otherwise the most square px ≥ py factorisation."""
    py = int(world ** 0.5)
    while world % py:
        py -= 1
    return (world // py, py)
