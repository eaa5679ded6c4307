# ArmonHIPNative.jl — package extension binding libarmon_hip.so into Armon.jl (Keluaa/Armon.jl @ 2024_08_07).
#
# NOT EXECUTED in the build image (no `julia` there). Written against the reference's sources, following its own
# Kokkos extension (ext/ArmonKokkos.jl). What CAN be checked without Julia is checked by tests/test_julia_binding.py:
# every `ccall(fn(:armon_hip_…), ret, (argtypes…), …)` below is parsed and compared — symbol, arity, C type of every
# argument and of the return value — with include/armon_hip.h as bound by armon.jl_amd/_lib.py (SIGNATURES), and the
# struct mirrors (CRange, CBlockData, SweepDesc, HaloDesc) with the ctypes structures field by field.
# The ABI itself is exercised end to end by the Python host (tests/) and by examples/native_cycle.c.
#
# Install: copy to Armon.jl/ext/, declare it in Project.toml ([weakdeps]/[extensions] ArmonHIPNative = "Libdl" (MPI is already a dependency of Armon)), then
#   ArmonParameters(; use_gpu=true, device=:HIP_native, use_cache_blocking=false, async_cycle=false,
#                     armon_hip_lib="/path/libarmon_hip.so", ...)
module ArmonHIPNative

using Armon
using Libdl
import MPI
import Armon: ArmonParameters, BlockGrid, BlockData, DomainRange, SolverState, LocalTaskBlock, Side, Axis
import Armon: create_device, init_backend, device_array_type, host_array_type, device_memory_info,
              print_device_info, solver_error, block_device_data, block_domain_range, stride_along,
              ghosts, real_face_size, real_block_size, block_size, first_sides, first_side, last_side, has_neighbour,
              boundary_condition, specific_heat_ratio, all_blocks, first_state, split_axes, update_solver_state!,
              update_EOS!, next_time_step, contribute_to_dt!, solver_cycle
import Armon: perfect_gas_EOS!, bizarrium_EOS!, acoustic!, acoustic_GAD!, cell_update!,
              advection_first_order!, advection_second_order!, euler_projection!, boundary_conditions!,
              pack_to_array!, unpack_from_array!, dtCFL_kernel, conservation_vars, init_test

# ---- library handle and symbol lookup -------------------------------------------------------------------------
const LIB = Ref{Ptr{Cvoid}}(C_NULL)                 # dlopen handle, set by init_backend(armon_hip_lib=...)
const SYMS = Dict{Symbol, Ptr{Cvoid}}()
function fn(name::Symbol)                           # fp64 entry point
    get!(SYMS, name) do
        LIB[] == C_NULL && (LIB[] = Libdl.dlopen("libarmon_hip.so"))
        Libdl.dlsym(LIB[], name)
    end
end
fn(name::Symbol, ::Type{Float64}) = fn(name)
fn(name::Symbol, ::Type{Float32}) = fn(Symbol(name, :_f32))     # data_type=Float32: same symbols, `_f32` suffix

check(rc) = rc == 0 ? nothing :                      # ref ext/ArmonKokkos.jl:72-76, src/utils.jl:108
    solver_error(rc == 4 ? :time : :cpp, unsafe_string(ccall(fn(:armon_hip_last_error), Cstring, ())))

# ---- device object + device array type -------------------------------------------------------------------------
mutable struct HIPNative                             # becomes the `Device` parameter of ArmonParameters
    ctx::Ptr{Cvoid}
    device_id::Int
end
const CURRENT_DEVICE = Ref{HIPNative}()

"""
Flat device array: `device_array_type(dev){T, 1}(undef, n)` of ref src/blocking/block_grid.jl:52-57 — the reference
applies `{T, 1}` to the type this returns, so it takes the element type AND the rank as parameters.
"""
mutable struct HIPVector{T, N} <: AbstractArray{T, N}
    ptr::Ptr{T}
    dims::NTuple{N, Int}
    dev::HIPNative
    function HIPVector{T, N}(::UndefInitializer, dims::NTuple{N, Integer}; kwargs...) where {T, N}   # kwargs: alloc_array_kwargs
        dev = CURRENT_DEVICE[]
        p = Ref{Ptr{Cvoid}}()
        check(ccall(fn(:armon_hip_malloc), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Ptr{Cvoid}}), dev.ctx, prod(dims) * sizeof(T), p))
        v = new{T, N}(Ptr{T}(p[]), Int.(dims), dev)
        finalizer(x -> ccall(fn(:armon_hip_free), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), x.dev.ctx, x.ptr), v)
    end
end
HIPVector{T, N}(u::UndefInitializer, dims::Integer...; kw...) where {T, N} = HIPVector{T, N}(u, dims; kw...)
HIPVector{T}(u::UndefInitializer, dims::Integer...; kw...) where {T} = HIPVector{T, length(dims)}(u, dims; kw...)
Base.size(v::HIPVector) = v.dims
Base.sizeof(v::HIPVector{T}) where T = length(v) * sizeof(T)
Base.pointer(v::HIPVector) = v.ptr
Base.unsafe_convert(::Type{Ptr{T}}, v::HIPVector{T}) where T = v.ptr
Base.getindex(v::HIPVector, i...) = error("scalar indexing of a device array: copyto! a host Array first")
# copyto! both ways (ref device_to_host!/host_to_device!, src/blocking/blocks.jl:121-143); kind: 1 H2D, 2 D2H, 3 D2D
Base.copyto!(dst::Array{T}, src::HIPVector{T}) where T = (check(ccall(fn(:armon_hip_memcpy), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), src.dev.ctx, dst, src.ptr, sizeof(dst), 2)); dst)
Base.copyto!(dst::HIPVector{T}, src::Array{T}) where T = (check(ccall(fn(:armon_hip_memcpy), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), dst.dev.ctx, dst.ptr, src, sizeof(src), 1)); dst)
Base.copyto!(dst::HIPVector{T}, src::HIPVector{T}) where T = (check(ccall(fn(:armon_hip_memcpy), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), dst.dev.ctx, dst.ptr, src.ptr, sizeof(src), 3)); dst)

# ---- gpu_aware=true: the reference types its MPI buffers as the DEVICE array (ref src/blocking/block_grid.jl:72,
# src/blocking/blocks.jl:188-201: MPI.Buffer(B(undef, size)) → MPI.Send_init / Recv_init), packs into them with
# pack_to_array! on the device, waits, then MPI.Startall (ref src/halo_exchange.jl:242-248). A ROCm-aware MPI takes the
# device pointer as it is — the same three methods MPI.jl's own AMDGPU extension defines for ROCArray:
MPI.Buffer(v::HIPVector{T}) where T = MPI.Buffer(v, Cint(length(v)), MPI.Datatype(T))
Base.cconvert(::Type{MPI.MPIPtr}, v::HIPVector) = v
Base.unsafe_convert(::Type{MPI.MPIPtr}, v::HIPVector) = reinterpret(MPI.MPIPtr, v.ptr)

# ---- backend hooks (ref src/parameters.jl:751-802,921-951,1031-1038) -----------------------------------------------
function create_device(::Val{:HIP_native})
    CURRENT_DEVICE[] = HIPNative(C_NULL, 0)           # the context is created by init_backend, once the library is known
end

function init_backend(params::ArmonParameters, dev::HIPNative;
                      armon_hip_lib = "libarmon_hip.so", device_id = 0, fused_sweep = true, exact_arithmetic = false,
                      native_cycle = true, options...)
    LIB[] = Libdl.dlopen(armon_hip_lib)
    empty!(SYMS)
    @assert ccall(fn(:armon_hip_flt_size), Cint, ()) == 8      # ref ext/ArmonKokkos.jl:122-139
    @assert ccall(fn(:armon_hip_idx_size), Cint, ()) == 8
    ctx = Ref{Ptr{Cvoid}}()
    check(ccall(fn(:armon_hip_init), Cint, (Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), device_id, C_NULL, ctx))
    dev.ctx, dev.device_id = ctx[], device_id
    params.backend_options = (; fused_sweep, exact_arithmetic, native_cycle)
    return options
end

device_array_type(::HIPNative) = HIPVector            # the reference applies {T, 1} (src/blocking/block_grid.jl:52)
host_array_type(::HIPNative) = Array
Base.wait(params::ArmonParameters{<:Any, HIPNative}) =
    check(ccall(fn(:armon_hip_sync), Cint, (Ptr{Cvoid},), params.device.ctx))
function device_memory_info(dev::HIPNative)
    free = Ref{Csize_t}(); total = Ref{Csize_t}()
    check(ccall(fn(:armon_hip_device_memory_info), Cint, (Ptr{Cvoid}, Ptr{Csize_t}, Ptr{Csize_t}), dev.ctx, free, total))
    (total = UInt64(total[]), free = UInt64(free[]))
end
function print_device_info(io::IO, pad::Int, p::ArmonParameters{<:Any, HIPNative})
    buf = zeros(UInt8, 256)
    check(ccall(fn(:armon_hip_device_name), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Csize_t), p.device.ctx, buf, 256))
    Armon.print_parameter(io, pad, "GPU", unsafe_string(pointer(buf)))
end

# ---- ranges: DomainRange (1-based) → armon_range (0-based), cf. ext/ArmonKokkos.jl:10-30 ---------------------------
struct CRange; col_start::Int64; col_step::Int64; col_len::Int64; row_start::Int64; row_len::Int64; end
CRange(r::DomainRange) = CRange(first(r.col) - 1, step(r.col), length(r.col), first(r.row) - 1, length(r.row))

# ---- kernel main functions: one method per @generic_kernel (signatures: SURVEY §8b) --------------------------------
const HP{T} = ArmonParameters{T, HIPNative}

perfect_gas_EOS!(p::HP{T}, d::BlockData, r::DomainRange, γ; kw...) where T = check(ccall(fn(:armon_hip_perfect_gas_EOS, T), Cint,
    (Ptr{Cvoid}, CRange, T, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}),
    p.device.ctx, CRange(r), γ, d.ρ, d.E, d.u, d.v, d.p, d.c, d.g))

bizarrium_EOS!(p::HP{T}, d::BlockData, r::DomainRange; kw...) where T = check(ccall(fn(:armon_hip_bizarrium_EOS, T), Cint,
    (Ptr{Cvoid}, CRange, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}),
    p.device.ctx, CRange(r), d.ρ, d.u, d.v, d.E, d.p, d.c, d.g))

acoustic!(p::HP{T}, d::BlockData, r::DomainRange, s::Int, uˢ, pˢ, uₐ; kw...) where T = check(ccall(fn(:armon_hip_acoustic, T), Cint,
    (Ptr{Cvoid}, CRange, Int64, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}),
    p.device.ctx, CRange(r), s, uˢ, pˢ, d.ρ, uₐ, d.p, d.c))

limiter_tag(::Armon.NoLimiter) = Cint(0); limiter_tag(::Armon.MinmodLimiter) = Cint(1)
limiter_tag(::Armon.SuperbeeLimiter) = Cint(2)                    # ref ext/ArmonKokkos.jl:50-57
acoustic_GAD!(p::HP{T}, d::BlockData, r::DomainRange, s::Int, dt, dx, uₐ, lim; kw...) where T = check(ccall(fn(:armon_hip_acoustic_GAD, T), Cint,
    (Ptr{Cvoid}, CRange, Int64, T, T, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Cint),
    p.device.ctx, CRange(r), s, dt, dx, d.uˢ, d.pˢ, d.ρ, uₐ, d.p, d.c, limiter_tag(lim)))

cell_update!(p::HP{T}, d::BlockData, r::DomainRange, s::Int, dx, dt, uₐ; kw...) where T = check(ccall(fn(:armon_hip_cell_update, T), Cint,
    (Ptr{Cvoid}, CRange, Int64, T, T, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}),
    p.device.ctx, CRange(r), s, dx, dt, d.uˢ, d.pˢ, d.ρ, uₐ, d.E))

advection_first_order!(p::HP{T}, d::BlockData, r::DomainRange, s::Int, dt, aρ, auρ, avρ, aEρ; kw...) where T = check(ccall(
    fn(:armon_hip_advection_first_order, T), Cint,
    (Ptr{Cvoid}, CRange, Int64, T, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}),
    p.device.ctx, CRange(r), s, dt, d.uˢ, d.ρ, d.u, d.v, d.E, aρ, auρ, avρ, aEρ))

advection_second_order!(p::HP{T}, d::BlockData, r::DomainRange, s::Int, dx, dt, aρ, auρ, avρ, aEρ; kw...) where T = check(ccall(
    fn(:armon_hip_advection_second_order, T), Cint,
    (Ptr{Cvoid}, CRange, Int64, T, T, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}),
    p.device.ctx, CRange(r), s, dx, dt, d.uˢ, d.ρ, d.u, d.v, d.E, aρ, auρ, avρ, aEρ))

euler_projection!(p::HP{T}, d::BlockData, r::DomainRange, s::Int, dx, dt, aρ, auρ, avρ, aEρ; kw...) where T = check(ccall(
    fn(:armon_hip_euler_projection, T), Cint,
    (Ptr{Cvoid}, CRange, Int64, T, T, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}),
    p.device.ctx, CRange(r), s, dx, dt, d.uˢ, d.ρ, d.u, d.v, d.E, aρ, auρ, avρ, aEρ))

function boundary_conditions!(p::HP{T}, d::BlockData, r::DomainRange, bsize, axis, side, u_factor, v_factor; kw...) where T
    incr = stride_along(bsize, axis); side in first_sides() && (incr = -incr)      # ref src/halo_exchange.jl:8-10
    check(ccall(fn(:armon_hip_boundary_conditions, T), Cint,
        (Ptr{Cvoid}, CRange, Int64, Cint, T, T, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}),
        p.device.ctx, CRange(r), incr, ghosts(bsize), u_factor, v_factor, d.ρ, d.u, d.v, d.p, d.c, d.g, d.E))
end

function pack_to_array!(p::HP{T}, r::DomainRange, bsize, side, array, vars::NTuple{N}; kw...) where {T, N}
    ptrs = Ptr{T}[Base.unsafe_convert(Ptr{T}, v) for v in vars]            # host array of N device pointers
    GC.@preserve vars check(ccall(fn(:armon_hip_pack_to_array, T), Cint,
        (Ptr{Cvoid}, CRange, Cint, Int64, Ptr{T}, Cint, Ptr{Ptr{T}}),
        p.device.ctx, CRange(r), ghosts(bsize), real_face_size(bsize, side), array, N, ptrs))
end
function unpack_from_array!(p::HP{T}, r::DomainRange, bsize, side, array, vars::NTuple{N}; kw...) where {T, N}
    ptrs = Ptr{T}[Base.unsafe_convert(Ptr{T}, v) for v in vars]
    GC.@preserve vars check(ccall(fn(:armon_hip_unpack_from_array, T), Cint,
        (Ptr{Cvoid}, CRange, Cint, Int64, Ptr{T}, Cint, Ptr{Ptr{T}}),
        p.device.ctx, CRange(r), ghosts(bsize), real_face_size(bsize, side), array, N, ptrs))
end

function dtCFL_kernel(p::HP{T}, state::SolverState, blk::LocalTaskBlock, Δx::NTuple{2}) where T   # ref src/reductions.jl:65
    d = block_device_data(blk); r = block_domain_range(blk.size, state.steps_ranges.real_domain); out = Ref{T}()
    check(ccall(fn(:armon_hip_dtCFL, T), Cint, (Ptr{Cvoid}, CRange, T, T, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}),
        p.device.ctx, CRange(r), Δx[1], Δx[2], d.u, d.v, d.c, out))
    out[]
end

function conservation_vars(p::HP{T}, blk::LocalTaskBlock) where T                                  # ref src/reductions.jl:271
    d = block_device_data(blk); r = block_domain_range(blk.size, blk.state.steps_ranges.real_domain)
    out = Ref{NTuple{2, T}}(); ds = T(prod(p.domain_size ./ p.global_grid))
    check(ccall(fn(:armon_hip_conservation_vars, T), Cint, (Ptr{Cvoid}, CRange, T, Ptr{T}, Ptr{T}, Ptr{NTuple{2, T}}),
        p.device.ctx, CRange(r), ds, d.ρ, d.E, out))
    out[]
end

# init_test (ref src/kernels.jl:106-145,176-207): test tag as in ext/ArmonKokkos.jl:60-69, the 16 BlockData vectors
# in field order (== armon_block_data), the block's global position and the cell sizes. The library zeroes uˢ,pˢ,work_*.
struct CBlockData{T}
    x::Ptr{T}; y::Ptr{T}; ρ::Ptr{T}; u::Ptr{T}; v::Ptr{T}; E::Ptr{T}; p::Ptr{T}; c::Ptr{T}; g::Ptr{T}
    uˢ::Ptr{T}; pˢ::Ptr{T}; work_1::Ptr{T}; work_2::Ptr{T}; work_3::Ptr{T}; work_4::Ptr{T}; mask::Ptr{T}
end
CBlockData{T}(d::BlockData) where T = CBlockData{T}((Base.unsafe_convert(Ptr{T}, getfield(d, f)) for f in Armon.block_vars())...)
test_tag(::Armon.Sod) = Cint(0); test_tag(::Armon.Sod_y) = Cint(1); test_tag(::Armon.Sod_circ) = Cint(2)
test_tag(::Armon.Bizarrium) = Cint(3); test_tag(::Armon.Sedov) = Cint(4); test_tag(::Armon.DebugIndexes) = Cint(5)
function init_test(p::HP{T}, d::BlockData, r::DomainRange, global_pos, bsize, ΔX, vars_to_zero, test; kw...) where T
    bd = Ref(CBlockData{T}(d))
    gpos = Ref(NTuple{2, Int64}(global_pos)); gN = Ref(NTuple{2, Int64}(p.global_grid))
    origin = Ref(NTuple{2, T}(p.origin)); dX = Ref(NTuple{2, T}(ΔX))
    sz = block_size(bsize)
    GC.@preserve d check(ccall(fn(:armon_hip_init_test, T), Cint,
        (Ptr{Cvoid}, CRange, Cint, Int64, Int64, Cint, Ptr{NTuple{2, Int64}}, Ptr{NTuple{2, Int64}}, Ptr{NTuple{2, T}},
         Ptr{NTuple{2, T}}, T, Ptr{CBlockData{T}}),
        p.device.ctx, CRange(r), test_tag(test), sz[1], sz[2], ghosts(bsize), gpos, gN, origin, dX,
        test isa Armon.Sedov ? T(test.r) : zero(T), bd))
end

# ---- fused sweep: `solver_cycle` on this device (INTEGRATION.md §2) ----------------------------------------------------
# The staged overrides above make the backend a drop-in at the reference's five passes per sweep. The override below
# replaces the sweep body of solver_cycle (ref src/solver.jl:288-320) by ONE kernel per sweep, with what surrounds it:
# ping-pong state vectors, the dt/CFL reduction fused into the last sweep of a cycle and consumed one cycle late (the
# reference's own lag, ref src/solver_state.jl:89-99,145-166), `p` materialised on the final cycle.
# Precedent for overriding whole steps on a device type: ext/ArmonKokkos.jl:212-258.
struct SweepDesc{T}      # == armon_sweep_desc / armon_sweep_desc_f32, include/armon_hip.h
    axis::Cint; scheme::Cint; limiter::Cint; projection::Cint; eos::Cint; nghost::Cint
    bc_low::Cint; bc_high::Cint; exact::Cint; x_kernel::Cint
    nx::Int64; ny::Int64; dt::Float64; dx::Float64; gamma::Float64
    u_factor_low::Float64; v_factor_low::Float64; u_factor_high::Float64; v_factor_high::Float64
    rho_in::Ptr{T}; u_in::Ptr{T}; v_in::Ptr{T}; E_in::Ptr{T}
    rho_out::Ptr{T}; u_out::Ptr{T}; v_out::Ptr{T}; E_out::Ptr{T}
    p_out::Ptr{T}; c_out::Ptr{T}
    dt_cfl_out::Ptr{T}; cfl_dx::Float64; cfl_dy::Float64
    out_lo::Int64; out_hi::Int64; dt_accumulate::Cint; reserved::Cint
    dt_state::Ptr{Cvoid}     # armon_dt_state on the device (graph replay of a cycle), or C_NULL: dt above is the time step itself
end

scheme_tag(::Armon.RiemannGodunov) = Cint(0); scheme_tag(::Armon.RiemannGAD) = Cint(1)
projection_tag(::Armon.EulerProjection) = Cint(0); projection_tag(::Armon.Euler2ndProjection) = Cint(1)
eos_tag(::Armon.TestCase) = Cint(0); eos_tag(::Armon.Bizarrium) = Cint(1)

"What the fused path keeps per grid besides the reference's BlockData."
mutable struct FusedState{T}
    alt::IdDict{Any, NTuple{4, HIPVector{T, 1}}}   # block → ping-pong partners of (ρ, u, v, E)
    dt_dev::HIPVector{T, 1}                        # device scalar written by the fused reduction
    dt_host::Ptr{T}                                # pinned landing zone, one slot per parity of the posting cycle
    posted::Set{Int}                               # cycles whose CFL step is on its way to the host
end
const FUSED = IdDict{Any, FusedState}()
const DT_EVENT_SLOT = 1012                         # event-pool slots 1012, 1013

function fused_state(p::HP{T}, grid::BlockGrid) where T
    get!(FUSED, grid) do
        alt = IdDict{Any, NTuple{4, HIPVector{T, 1}}}()
        for blk in all_blocks(grid)
            n = length(block_device_data(blk).ρ)
            alt[blk] = ntuple(_ -> HIPVector{T, 1}(undef, n), 4)
        end
        host = Ref{Ptr{Cvoid}}()
        check(ccall(fn(:armon_hip_malloc_host), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Ptr{Cvoid}}), p.device.ctx, 2 * sizeof(T), host))
        FusedState{T}(alt, HIPVector{T, 1}(undef, 2), Ptr{T}(host[]), Set{Int}())
    end
end

"The armon_sweep_desc of the sweep `state` is set up for (update_solver_state!): current state of `blk` → its ping-pong partners."
function sweep_desc(p::HP{T}, state::SolverState, blk::LocalTaskBlock, fs::FusedState{T}; emit_p::Bool, emit_dt::Bool,
                    out::NTuple{2, Int} = (0, 0), dt_out::Ptr{T} = pointer(fs.dt_dev)) where T
    d = block_device_data(blk); alt = fs.alt[blk]
    nx, ny = real_block_size(blk.size)
    lo, hi = first_side(state.axis), last_side(state.axis)
    (ufl, vfl) = boundary_condition(state.test_case, lo); (ufh, vfh) = boundary_condition(state.test_case, hi)
    Δ = p.domain_size ./ p.global_grid
    SweepDesc{T}(
        Int(state.axis) - 1, scheme_tag(state.riemann_scheme), limiter_tag(state.riemann_limiter),
        projection_tag(state.projection_scheme), eos_tag(state.test_case), ghosts(blk.size),
        !has_neighbour(p, lo), !has_neighbour(p, hi), p.backend_options.exact_arithmetic, 0,
        nx, ny, state.dt, state.dx, specific_heat_ratio(state.test_case), ufl, vfl, ufh, vfh,
        pointer(d.ρ), pointer(d.u), pointer(d.v), pointer(d.E),
        pointer(alt[1]), pointer(alt[2]), pointer(alt[3]), pointer(alt[4]),
        emit_p ? pointer(d.p) : Ptr{T}(C_NULL), Ptr{T}(C_NULL),
        emit_dt ? dt_out : Ptr{T}(C_NULL), Δ[1], Δ[2], out[1], out[2], 0, 0, C_NULL)
end

"One directional sweep of one block as ONE launch (armon_hip_sweep), then exchange the roles of the two state sets."
function fused_sweep!(p::HP{T}, state::SolverState, blk::LocalTaskBlock, fs::FusedState{T}; emit_p::Bool, emit_dt::Bool,
                      out::NTuple{2, Int} = (0, 0),            # cells [out[1], out[2]) of the sweep axis; (0, 0) = all
                      ctx::Ptr{Cvoid} = p.device.ctx,          # or the tile's edge context (transfer stream)
                      dt_out::Ptr{T} = pointer(fs.dt_dev), swap::Bool = true) where T
    d = block_device_data(blk); alt = fs.alt[blk]
    desc = Ref(sweep_desc(p, state, blk, fs; emit_p, emit_dt, out, dt_out))
    GC.@preserve d alt check(ccall(fn(:armon_hip_sweep, T), Cint, (Ptr{Cvoid}, Ptr{SweepDesc{T}}), ctx, desc))
    swap && swap_state!(blk, fs)
end

"ping-pong: the fresh state lives in `alt`; BlockData is immutable but HIPVector is ours — swap the allocations"
function swap_state!(blk::LocalTaskBlock, fs::FusedState)
    d = block_device_data(blk)
    for (v, a) in zip((d.ρ, d.u, d.v, d.E), fs.alt[blk])
        v.ptr, a.ptr = a.ptr, v.ptr
    end
end

# Under MPI the fused sweeps exchange their halos through the library's RCCL group (one process per GPU, INTEGRATION §3)
# instead of the reference's MPI requests: faces of (ρ,u,v,E) only, posted before the interior of the sweep, unpacked on
# the transfer stream where the LAG-wide boundary strips follow, nothing waited for on the host
# (replaces ref src/halo_exchange.jl:229-354 for this path).
const RANK_GROUPS = IdDict{Any, Any}()
function rank_group(p::HP)
    get!(RANK_GROUPS, p) do
        px, py = p.proc_dims
        me = p.cart_coords[1] * py + p.cart_coords[2]      # rank in p.cart_comm: cartesian ranks are row-major (MPI 7.5)
        id = me == 0 ? unique_id() : zeros(UInt8, 256)
        MPI.Bcast!(id, 0, p.cart_comm)
        stream = ccall(fn(:armon_hip_stream), Ptr{Cvoid}, (Ptr{Cvoid},), p.device.ctx)
        TileGroup(px, py, me, p.device.device_id, id; stream)    # adopts the stream the sweeps are already enqueued on
    end
end

sweep_lag(state::SolverState) = 2 + Int(scheme_tag(state.riemann_scheme)) + Int(projection_tag(state.projection_scheme))

function fused_sweep_mpi!(p::HP{T}, state::SolverState, blk::LocalTaskBlock, fs::FusedState{T}; emit_p::Bool, emit_dt::Bool) where T
    lo_r = has_neighbour(p, first_side(state.axis)); hi_r = has_neighbour(p, last_side(state.axis))
    (lo_r || hi_r) || return fused_sweep!(p, state, blk, fs; emit_p, emit_dt)
    g = rank_group(p)
    d = block_device_data(blk)
    nx, ny = real_block_size(blk.size)
    n = state.axis == Axis.X ? nx : ny
    lag = sweep_lag(state)
    halo = [HaloDesc(nx, ny, ghosts(blk.size), (d.ρ, d.u, d.v, d.E))]
    GC.@preserve d begin
        halo_exchange_start!(g, state.axis, halo, T)
        if n < 2lag + 1                                    # no interior to hide the transfer behind
            halo_exchange_finish!(g, state.axis, halo, T)
            return fused_sweep!(p, state, blk, fs; emit_p, emit_dt)
        end
        fused_sweep!(p, state, blk, fs; emit_p, emit_dt, out = (lo_r ? lag : 0, hi_r ? n - lag : n), swap = false)
        halo_exchange_finish_edge!(g, state.axis, halo, T)
        edge = edge_context(g, 0); e_dt = edge_dt(g, 0, T)
        lo_r && fused_sweep!(p, state, blk, fs; emit_p, emit_dt, out = (0, lag), ctx = edge, dt_out = e_dt, swap = false)
        hi_r && fused_sweep!(p, state, blk, fs; emit_p, emit_dt, out = (n - lag, n), ctx = edge, dt_out = e_dt + sizeof(T), swap = false)
        emit_dt ? edge_join!(g, [pointer(fs.dt_dev)]) : edge_join!(g)
    end
    swap_state!(blk, fs)
end

# The whole cycle of this rank's tile in ONE library call (armon_hip_mgpu_cycle, include/armon_hip.h): the exchanges along each
# sweep's axis, the interior while the faces travel, unpack + strips on the transfer stream, the all-reduce of the next CFL
# step and its read-back into the pinned slot — everything fused_sweep_mpi! / dt_allreduce! / the read-back below do call by
# call, enqueued natively (≈ 0.1 ms of host time per cycle instead of ≈ 0.2–0.6), with the exchange of the NEXT cycle's
# first sweep posted ahead. `false` when the library refuses nothing; the caller keeps the call-by-call path as its fallback.
struct CyclePlan         # == armon_cycle_plan
    n_sweeps::Cint; emit_p::Cint; emit_dt::Cint; overlap::Cint
    axis::NTuple{4, Cint}; dt::NTuple{4, Float64}
    next_axis::Cint; event_slot::Cint
    event_ctx::Ptr{Cvoid}; dt_host::Ptr{Cvoid}; dt_event_slot::Cint; reserved::Cint
end
struct TileCycle{T}      # == armon_tile_cycle / armon_tile_cycle_f32
    x::SweepDesc{T}; y::SweepDesc{T}
end

function native_cycle!(p::HP{T}, state::SolverState, blk::LocalTaskBlock, fs::FusedState{T}, will_end::Bool) where T
    g = rank_group(p)
    gdt = state.global_dt
    sweeps = collect(split_axes(state))
    axes = zeros(Cint, 4); dts = zeros(Float64, 4)
    descs = Dict{Axis.T, SweepDesc{T}}()
    for (k, (axis, dt_factor)) in enumerate(sweeps)
        update_solver_state!(p, state, axis, dt_factor)          # dx, dt = current_dt·factor, steps_ranges of that axis
        axes[k] = Int(axis) - 1; dts[k] = state.dt
        descs[axis] = sweep_desc(p, state, blk, fs; emit_p = true, emit_dt = !p.cst_dt)
    end
    for axis in (Axis.X, Axis.Y)                                  # a splitting that skips an axis: its descriptor is never read
        haskey(descs, axis) || (descs[axis] = descs[first(keys(descs))])
    end
    next_first = Int(first(first(split_axes(state.splitting, T, gdt.cycle + 1)))) - 1     # ref src/axis_splitting.jl:22
    slot = gdt.cycle & 1
    plan = Ref(CyclePlan(length(sweeps), will_end, !p.cst_dt, 1, Tuple(axes), Tuple(dts), will_end ? -1 : next_first, -1,
        C_NULL, p.cst_dt ? C_NULL : Ptr{Cvoid}(fs.dt_host + slot * sizeof(T)), p.cst_dt ? -1 : DT_EVENT_SLOT + slot, 0))
    tile = Ref(TileCycle{T}(descs[Axis.X], descs[Axis.Y]))
    d = block_device_data(blk); alt = fs.alt[blk]
    GC.@preserve d alt check(ccall(fn(:armon_hip_mgpu_cycle, T), Cint, (Ptr{Cvoid}, Ptr{CyclePlan}, Ptr{TileCycle{T}}), g.handle, plan, tile))
    isodd(length(sweeps)) && swap_state!(blk, fs)
    p.cst_dt && return false
    push!(fs.posted, gdt.cycle)
    if gdt.cycle > 0                                              # the read-back the previous cycle posted, on the EDGE context
        prev = gdt.cycle - 1
        check(ccall(fn(:armon_hip_event_sync), Cint, (Ptr{Cvoid}, Cint), edge_context(g, 0), DT_EVENT_SLOT + (prev & 1)))
        delete!(fs.posted, prev)
        contribute_to_dt!(p, gdt, unsafe_load(fs.dt_host, (prev & 1) + 1); all_blocks = true)
    end
    return false
end

function solver_cycle(p::HP{T}, grid::BlockGrid) where T
    # per-step dumps/comparisons need the intermediate arrays of the staged kernels (then MPI runs keep the reference's
    # exchange between them, on gpu_aware buffers); otherwise MPI runs take the library's RCCL group, see above
    if !p.backend_options.fused_sweep || p.compare
        return invoke(solver_cycle, Tuple{ArmonParameters, BlockGrid}, p, grid)
    end
    mpi = p.use_MPI && p.proc_size > 1
    state = first_state(grid)
    gdt = state.global_dt
    fs = fused_state(p, grid)
    if gdt.cycle == 0
        update_EOS!(p, state, grid)                       # c of the initial state, for the first time step only
        next_time_step(p, state, grid) && return true     # staged dtCFL kernel, synchronous, once per run
    else
        state.dt = gdt.current_dt                         # known since the previous cycle (one-cycle lag)
    end
    will_end = gdt.cycle + 1 ≥ p.maxcycle || gdt.time + gdt.current_dt ≥ p.maxtime    # time_loop's exit test (src/solver.jl:350)
    if mpi && p.backend_options.native_cycle                  # default: the library enqueues the whole cycle
        return native_cycle!(p, state, first(all_blocks(grid)), fs, will_end)
    end
    sweeps = collect(split_axes(state))
    for (k, (axis, dt_factor)) in enumerate(sweeps)
        update_solver_state!(p, state, axis, dt_factor)
        last = k == length(sweeps)
        for blk in all_blocks(grid)                       # one block per GPU (use_cache_blocking=false)
            (mpi ? fused_sweep_mpi! : fused_sweep!)(p, state, blk, fs; emit_p = last && will_end, emit_dt = last && !p.cst_dt)
        end
    end
    p.cst_dt && return false
    # global minimum over the tiles, on the device (replaces the MPI_Iallreduce of ref src/solver_state.jl:89-111; the
    # reference's own all-reduce in update_dt! then sees the same value on every rank)
    mpi && dt_allreduce!(rank_group(p), [pointer(fs.dt_dev)])
    # post the read-back of the CFL step this cycle's last sweep reduced (the state the NEXT cycle starts from) …
    slot = gdt.cycle & 1
    check(ccall(fn(:armon_hip_memcpy_async), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint),
        p.device.ctx, fs.dt_host + slot * sizeof(T), pointer(fs.dt_dev), sizeof(T), 2))
    check(ccall(fn(:armon_hip_event_record), Cint, (Ptr{Cvoid}, Cint), p.device.ctx, DT_EVENT_SLOT + slot))
    push!(fs.posted, gdt.cycle)
    # … and pick up the one the previous cycle posted: it is this cycle's `local_time_step` (ref src/reductions.jl:
    # 164-199), contributed exactly as next_time_step would, so next_cycle! finds the state machine where it expects it.
    if gdt.cycle > 0
        prev = gdt.cycle - 1
        check(ccall(fn(:armon_hip_event_sync), Cint, (Ptr{Cvoid}, Cint), p.device.ctx, DT_EVENT_SLOT + (prev & 1)))
        delete!(fs.posted, prev)
        contribute_to_dt!(p, gdt, unsafe_load(fs.dt_host, (prev & 1) + 1); all_blocks = true)
    end
    return false
end

# ---- placement of the streamed vectors (DESIGN §3) -----------------------------------------------------------------------
"pool[1:4] = (ρ,u,v,E) holding the state, pool[5:8] their ping-pong partners, the rest spares → 1-based indices to keep"
function tune_placement!(p::HP{T}, x_desc::SweepDesc{T}, y_desc::SweepDesc{T}, pool::Vector{<:HIPVector{T}}; tries = 12) where T
    ptrs = Ptr{Cvoid}[Ptr{Cvoid}(v.ptr) for v in pool]; picks = zeros(Cint, 8); times = zeros(Float64, tries)
    GC.@preserve pool check(ccall(fn(:armon_hip_tune_placement, T), Cint,
        (Ptr{Cvoid}, Ptr{SweepDesc{T}}, Ptr{SweepDesc{T}}, Ptr{Ptr{Cvoid}}, Cint, Csize_t, Cint, Ptr{Cint}, Ptr{Float64}),
        p.device.ctx, Ref(x_desc), Ref(y_desc), ptrs, length(pool), sizeof(pool[1]), tries, picks, times))
    picks .+ 1, times
end

# ---- multi-GPU without MPI: every tile in this process (INTEGRATION.md §3) ------------------------------------------------
struct HaloDesc      # == armon_halo_desc
    nx::Int64; ny::Int64; nghost::Cint; nvars::Cint
    vars::NTuple{8, Ptr{Cvoid}}
end
HaloDesc(nx, ny, g, vars) = HaloDesc(nx, ny, g, length(vars),
    ntuple(i -> i ≤ length(vars) ? Ptr{Cvoid}(pointer(vars[i])) : C_NULL, 8))

mutable struct TileGroup; handle::Ptr{Cvoid}; px::Int; py::Int; end
function TileGroup(px, py, device_ids::Vector{Cint})
    g = Ref{Ptr{Cvoid}}()
    check(ccall(fn(:armon_hip_mgpu_init), Cint, (Cint, Cint, Ptr{Cint}, Ptr{Ptr{Cvoid}}), px, py, device_ids, g))
    finalizer(x -> ccall(fn(:armon_hip_mgpu_destroy), Cint, (Ptr{Cvoid},), x.handle), TileGroup(g[], px, py))
end
"One process per GPU under MPI: `id` = the bytes of armon_hip_mgpu_unique_id from rank 0 after MPI.Bcast!."
function TileGroup(px, py, rank, device_id, id::Vector{UInt8}; stream::Ptr{Cvoid} = C_NULL)
    g = Ref{Ptr{Cvoid}}()
    check(ccall(fn(:armon_hip_mgpu_init_rank), Cint, (Cint, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
        px, py, rank, device_id, stream, id, g))
    finalizer(x -> ccall(fn(:armon_hip_mgpu_destroy), Cint, (Ptr{Cvoid},), x.handle), TileGroup(g[], px, py))
end
unique_id() = (id = zeros(UInt8, 256); check(ccall(fn(:armon_hip_mgpu_unique_id), Cint, (Ptr{Cvoid},), id)); id)
tile_context(g::TileGroup, k) = ccall(fn(:armon_hip_mgpu_ctx), Ptr{Cvoid}, (Ptr{Cvoid}, Cint), g.handle, k)
halo_exchange_start!(g::TileGroup, axis, tiles::Vector{HaloDesc}, ::Type{T} = Float64) where T =
    check(ccall(fn(:armon_hip_halo_exchange_start, T), Cint, (Ptr{Cvoid}, Cint, Ptr{HaloDesc}), g.handle, Int(axis) - 1, tiles))
halo_exchange_finish!(g::TileGroup, axis, tiles::Vector{HaloDesc}, ::Type{T} = Float64) where T =
    check(ccall(fn(:armon_hip_halo_exchange_finish, T), Cint, (Ptr{Cvoid}, Cint, Ptr{HaloDesc}), g.handle, Int(axis) - 1, tiles))
dt_allreduce!(g::TileGroup, dt_dev::Vector{Ptr{T}}) where T =
    check(ccall(fn(:armon_hip_dt_allreduce, T), Cint, (Ptr{Cvoid}, Ptr{Ptr{T}}), g.handle, dt_dev))
# edge stream: unpack + the LAG-wide boundary strips on the tile's transfer stream, concurrent with the interior sweep
# (launch the strips with `fused_sweep!` on `edge_context(g, k)`, `dt_cfl_out = edge_dt(g, k, T) + side index`)
edge_context(g::TileGroup, k) = ccall(fn(:armon_hip_mgpu_edge_ctx), Ptr{Cvoid}, (Ptr{Cvoid}, Cint), g.handle, k)
edge_dt(g::TileGroup, k, ::Type{T} = Float64) where T =
    Ptr{T}(ccall(fn(:armon_hip_mgpu_edge_dt), Ptr{Cvoid}, (Ptr{Cvoid}, Cint), g.handle, k))
halo_exchange_finish_edge!(g::TileGroup, axis, tiles::Vector{HaloDesc}, ::Type{T} = Float64) where T =
    check(ccall(fn(:armon_hip_halo_exchange_finish_edge, T), Cint, (Ptr{Cvoid}, Cint, Ptr{HaloDesc}), g.handle, Int(axis) - 1, tiles))
edge_join!(g::TileGroup, dt_dev::Vector{Ptr{T}}) where T =          # after the last sweep of a cycle: folds the strips' dt in
    check(ccall(fn(:armon_hip_mgpu_edge_join, T), Cint, (Ptr{Cvoid}, Ptr{Ptr{T}}), g.handle, dt_dev))
edge_join!(g::TileGroup) = check(ccall(fn(:armon_hip_mgpu_edge_join), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}), g.handle, C_NULL))

end # module
