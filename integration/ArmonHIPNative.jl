# ArmonHIPNative.jl — package extension binding libarmon_hip.so into Armon.jl (Keluaa/Armon.jl @ 2024_08_07).
#
# NOT EXECUTED in the build image (no `julia` there): written against the reference's sources, following its own
# Kokkos extension (ext/ArmonKokkos.jl). The ABI it binds is exercised end to end by the Python host
# (armon.jl_amd/_lib.py, tests/) and by examples/native_cycle.c. See INTEGRATION.md for the walk-through.
#
# Install: copy to Armon.jl/ext/, declare it in Project.toml ([extensions] ArmonHIPNative = ...), then
#   ArmonParameters(; use_gpu=true, device=:HIP_native, use_cache_blocking=false, armon_hip_lib="/path/libarmon_hip.so", ...)
module ArmonHIPNative

using Armon
import Armon: ArmonParameters, BlockData, DomainRange, SolverState, LocalTaskBlock, Side, Axis
import Armon: create_device, init_backend, device_array_type, host_array_type, device_memory_info,
              print_device_info, solver_error, block_device_data, block_domain_range, stride_along,
              ghosts, real_face_size, limiter_from_name
import Armon: perfect_gas_EOS!, bizarrium_EOS!, acoustic!, acoustic_GAD!, cell_update!,
              advection_first_order!, advection_second_order!, euler_projection!, boundary_conditions!,
              pack_to_array!, unpack_from_array!, dtCFL_kernel, conservation_vars, init_test

const lib = Ref{String}("libarmon_hip.so")          # path set by init_backend(armon_hip_lib=...)

# ---- device object + device array type -------------------------------------------------------------------
mutable struct HIPNative                             # becomes the `Device` parameter of ArmonParameters
    ctx::Ptr{Cvoid}
end

"Flat device vector: what `device_array_type(dev){T,1}(undef, n)` must return (src/blocking/blocks.jl:36-44)."
mutable struct HIPVector{T} <: AbstractVector{T}
    ptr::Ptr{T}; n::Int; dev::HIPNative
end
function HIPVector{T,1}(::UndefInitializer, n::Integer) where T     # ref src/blocking/block_grid.jl:52-57
    dev = CURRENT_DEVICE[]; p = Ref{Ptr{Cvoid}}()
    check(ccall((:armon_hip_malloc, lib[]), Cint, (Ptr{Cvoid}, Csize_t, Ptr{Ptr{Cvoid}}), dev.ctx, n*sizeof(T), p))
    v = HIPVector{T}(Ptr{T}(p[]), n, dev)
    finalizer(x -> ccall((:armon_hip_free, lib[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), x.dev.ctx, x.ptr), v)
end
Base.size(v::HIPVector) = (v.n,)
Base.unsafe_convert(::Type{Ptr{T}}, v::HIPVector{T}) where T = v.ptr
# copyto! both ways (ref device_to_host!/host_to_device!, src/blocking/blocks.jl:121-143)
Base.copyto!(dst::Vector{T}, src::HIPVector{T}) where T = (check(ccall((:armon_hip_memcpy, lib[]), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), src.dev.ctx, dst, src.ptr, sizeof(dst), 2)); dst)
Base.copyto!(dst::HIPVector{T}, src::Vector{T}) where T = (check(ccall((:armon_hip_memcpy, lib[]), Cint,
    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), dst.dev.ctx, dst.ptr, src, sizeof(src), 1)); dst)

check(rc) = rc == 0 ? nothing :                      # ref ext/ArmonKokkos.jl:72-76, src/utils.jl:108
    solver_error(:cpp, unsafe_string(ccall((:armon_hip_last_error, lib[]), Cstring, ())))

# ---- backend hooks (ref src/parameters.jl:751-802,921-951,1031-1038) ---------------------------------------
const CURRENT_DEVICE = Ref{HIPNative}()
function create_device(::Val{:HIP_native})
    ctx = Ref{Ptr{Cvoid}}()
    check(ccall((:armon_hip_init, lib[]), Cint, (Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), 0, C_NULL, ctx))
    @assert ccall((:armon_hip_flt_size, lib[]), Cint, ()) == 8      # ref ext/ArmonKokkos.jl:122-139
    @assert ccall((:armon_hip_idx_size, lib[]), Cint, ()) == 8
    CURRENT_DEVICE[] = HIPNative(ctx[])
end
init_backend(params::ArmonParameters, ::HIPNative; armon_hip_lib = nothing, options...) =
    (isnothing(armon_hip_lib) || (lib[] = armon_hip_lib); params.backend_options = nothing; options)
device_array_type(::HIPNative) = HIPVector
host_array_type(::HIPNative) = Vector
Base.wait(params::ArmonParameters{<:Any, HIPNative}) =
    check(ccall((:armon_hip_sync, lib[]), Cint, (Ptr{Cvoid},), params.device.ctx))
function device_memory_info(dev::HIPNative)
    free = Ref{Csize_t}(); total = Ref{Csize_t}()
    check(ccall((:armon_hip_device_memory_info, lib[]), Cint, (Ptr{Cvoid}, Ptr{Csize_t}, Ptr{Csize_t}), dev.ctx, free, total))
    (total = UInt64(total[]), free = UInt64(free[]))
end

# ---- ranges: DomainRange (1-based) → armon_range (0-based), cf. ext/ArmonKokkos.jl:10-30 -------------------
struct CRange; col_start::Int64; col_step::Int64; col_len::Int64; row_start::Int64; row_len::Int64; end
CRange(r::DomainRange) = CRange(first(r.col) - 1, step(r.col), length(r.col), first(r.row) - 1, length(r.row))
const P = Ptr{Float64}

# ---- kernel main functions: one method per @generic_kernel (signatures: SURVEY §8b) -------------------------
const HP = ArmonParameters{Float64, HIPNative}

perfect_gas_EOS!(p::HP, d::BlockData, r::DomainRange, γ; kw...) = check(ccall((:armon_hip_perfect_gas_EOS, lib[]), Cint,
    (Ptr{Cvoid}, CRange, Float64, P, P, P, P, P, P, P), p.device.ctx, CRange(r), γ, d.ρ, d.E, d.u, d.v, d.p, d.c, d.g))

bizarrium_EOS!(p::HP, d::BlockData, r::DomainRange; kw...) = check(ccall((:armon_hip_bizarrium_EOS, lib[]), Cint,
    (Ptr{Cvoid}, CRange, P, P, P, P, P, P, P), p.device.ctx, CRange(r), d.ρ, d.u, d.v, d.E, d.p, d.c, d.g))

acoustic!(p::HP, d::BlockData, r::DomainRange, s::Int, uˢ, pˢ, uₐ; kw...) = check(ccall((:armon_hip_acoustic, lib[]), Cint,
    (Ptr{Cvoid}, CRange, Int64, P, P, P, P, P, P), p.device.ctx, CRange(r), s, uˢ, pˢ, d.ρ, uₐ, d.p, d.c))

limiter_tag(::Armon.NoLimiter) = Cint(0); limiter_tag(::Armon.MinmodLimiter) = Cint(1)
limiter_tag(::Armon.SuperbeeLimiter) = Cint(2)                    # ref ext/ArmonKokkos.jl:50-57
acoustic_GAD!(p::HP, d::BlockData, r::DomainRange, s::Int, dt, dx, uₐ, lim; kw...) = check(ccall((:armon_hip_acoustic_GAD, lib[]), Cint,
    (Ptr{Cvoid}, CRange, Int64, Float64, Float64, P, P, P, P, P, P, Cint),
    p.device.ctx, CRange(r), s, dt, dx, d.uˢ, d.pˢ, d.ρ, uₐ, d.p, d.c, limiter_tag(lim)))

cell_update!(p::HP, d::BlockData, r::DomainRange, s::Int, dx, dt, uₐ; kw...) = check(ccall((:armon_hip_cell_update, lib[]), Cint,
    (Ptr{Cvoid}, CRange, Int64, Float64, Float64, P, P, P, P, P), p.device.ctx, CRange(r), s, dx, dt, d.uˢ, d.pˢ, d.ρ, uₐ, d.E))

advection_first_order!(p::HP, d::BlockData, r::DomainRange, s::Int, dt, aρ, auρ, avρ, aEρ; kw...) = check(ccall(
    (:armon_hip_advection_first_order, lib[]), Cint, (Ptr{Cvoid}, CRange, Int64, Float64, P, P, P, P, P, P, P, P, P),
    p.device.ctx, CRange(r), s, dt, d.uˢ, d.ρ, d.u, d.v, d.E, aρ, auρ, avρ, aEρ))

advection_second_order!(p::HP, d::BlockData, r::DomainRange, s::Int, dx, dt, aρ, auρ, avρ, aEρ; kw...) = check(ccall(
    (:armon_hip_advection_second_order, lib[]), Cint, (Ptr{Cvoid}, CRange, Int64, Float64, Float64, P, P, P, P, P, P, P, P, P),
    p.device.ctx, CRange(r), s, dx, dt, d.uˢ, d.ρ, d.u, d.v, d.E, aρ, auρ, avρ, aEρ))

euler_projection!(p::HP, d::BlockData, r::DomainRange, s::Int, dx, dt, aρ, auρ, avρ, aEρ; kw...) = check(ccall(
    (:armon_hip_euler_projection, lib[]), Cint, (Ptr{Cvoid}, CRange, Int64, Float64, Float64, P, P, P, P, P, P, P, P, P),
    p.device.ctx, CRange(r), s, dx, dt, d.uˢ, d.ρ, d.u, d.v, d.E, aρ, auρ, avρ, aEρ))

function boundary_conditions!(p::HP, d::BlockData, r::DomainRange, bsize, axis, side, u_factor, v_factor; kw...)
    incr = stride_along(bsize, axis); side in Armon.first_sides() && (incr = -incr)      # ref src/halo_exchange.jl:8-10
    check(ccall((:armon_hip_boundary_conditions, lib[]), Cint, (Ptr{Cvoid}, CRange, Int64, Cint, Float64, Float64,
        P, P, P, P, P, P, P), p.device.ctx, CRange(r), incr, ghosts(bsize), u_factor, v_factor, d.ρ, d.u, d.v, d.p, d.c, d.g, d.E))
end

function pack_to_array!(p::HP, r::DomainRange, bsize, side, array, vars::NTuple{N}; kw...) where N
    ptrs = [Base.unsafe_convert(P, v) for v in vars]            # host array of N device pointers
    check(ccall((:armon_hip_pack_to_array, lib[]), Cint, (Ptr{Cvoid}, CRange, Cint, Int64, P, Cint, Ptr{P}),
        p.device.ctx, CRange(r), ghosts(bsize), real_face_size(bsize, side), array, N, ptrs))
end
function unpack_from_array!(p::HP, r::DomainRange, bsize, side, array, vars::NTuple{N}; kw...) where N
    ptrs = [Base.unsafe_convert(P, v) for v in vars]
    check(ccall((:armon_hip_unpack_from_array, lib[]), Cint, (Ptr{Cvoid}, CRange, Cint, Int64, P, Cint, Ptr{P}),
        p.device.ctx, CRange(r), ghosts(bsize), real_face_size(bsize, side), array, N, ptrs))
end

function dtCFL_kernel(p::HP, state::SolverState, blk::LocalTaskBlock, Δx::NTuple{2})   # ref src/reductions.jl:65
    d = block_device_data(blk); r = block_domain_range(blk.size, state.steps_ranges.real_domain); out = Ref{Float64}()
    check(ccall((:armon_hip_dtCFL, lib[]), Cint, (Ptr{Cvoid}, CRange, Float64, Float64, P, P, P, Ptr{Float64}),
        p.device.ctx, CRange(r), Δx[1], Δx[2], d.u, d.v, d.c, out))
    out[]
end

function conservation_vars(p::HP, blk::LocalTaskBlock)                                   # ref src/reductions.jl:271
    d = block_device_data(blk); r = block_domain_range(blk.size, blk.state.steps_ranges.real_domain)
    out = Ref{NTuple{2, Float64}}(); ds = prod(p.domain_size ./ p.global_grid)
    check(ccall((:armon_hip_conservation_vars, lib[]), Cint, (Ptr{Cvoid}, CRange, Float64, P, P, Ptr{NTuple{2,Float64}}),
        p.device.ctx, CRange(r), ds, d.ρ, d.E, out))
    out[]
end
# init_test (ref src/kernels.jl:106-145,176-207): test tag as in ext/ArmonKokkos.jl:60-69, the 16 BlockData vectors
# in field order (== armon_block_data), the block's global position and the cell sizes. The library zeroes uˢ,pˢ,work_*.
struct CBlockData; x::P; y::P; ρ::P; u::P; v::P; E::P; p::P; c::P; g::P; uˢ::P; pˢ::P; w1::P; w2::P; w3::P; w4::P; mask::P; end
test_tag(::Armon.Sod) = Cint(0); test_tag(::Armon.Sod_y) = Cint(1); test_tag(::Armon.Sod_circ) = Cint(2)
test_tag(::Armon.Bizarrium) = Cint(3); test_tag(::Armon.Sedov) = Cint(4); test_tag(::Armon.DebugIndexes) = Cint(5)
function init_test(p::HP, d::BlockData, r::DomainRange, global_pos, bsize, ΔX, vars_to_zero, test; kw...)
    bd = Ref(CBlockData(d.x, d.y, d.ρ, d.u, d.v, d.E, d.p, d.c, d.g, d.uˢ, d.pˢ, d.work_1, d.work_2, d.work_3, d.work_4, d.mask))
    gpos = Ref(NTuple{2,Int64}(global_pos)); gN = Ref(NTuple{2,Int64}(p.global_grid))
    origin = Ref(NTuple{2,Float64}(p.origin)); dX = Ref(NTuple{2,Float64}(ΔX))
    sz = Armon.block_size(bsize)
    check(ccall((:armon_hip_init_test, lib[]), Cint,
        (Ptr{Cvoid}, CRange, Cint, Int64, Int64, Cint, Ptr{NTuple{2,Int64}}, Ptr{NTuple{2,Int64}}, Ptr{NTuple{2,Float64}},
         Ptr{NTuple{2,Float64}}, Float64, Ptr{CBlockData}),
        p.device.ctx, CRange(r), test_tag(test), sz[1], sz[2], ghosts(bsize), gpos, gN, origin, dX,
        test isa Armon.Sedov ? test.r : 0.0, bd))
end

# ---- fused sweep + placement of the streamed vectors (INTEGRATION.md §2) -------------------------------------
Base.@kwdef struct SweepDesc      # == armon_sweep_desc, include/armon_hip.h
    axis::Cint; scheme::Cint; limiter::Cint; projection::Cint; eos::Cint; nghost::Cint
    bc_low::Cint; bc_high::Cint; exact::Cint = 0; x_kernel::Cint = 0
    nx::Int64; ny::Int64; dt::Float64; dx::Float64; gamma::Float64 = 7/5
    u_factor_low::Float64; v_factor_low::Float64; u_factor_high::Float64; v_factor_high::Float64
    rho_in::P; u_in::P; v_in::P; E_in::P; rho_out::P; u_out::P; v_out::P; E_out::P
    p_out::P = C_NULL; c_out::P = C_NULL; dt_cfl_out::P = C_NULL; cfl_dx::Float64 = 0; cfl_dy::Float64 = 0
    out_lo::Int64 = 0; out_hi::Int64 = 0; dt_accumulate::Cint = 0; reserved::Cint = 0
end
fused_sweep!(p::HP, desc::SweepDesc) =
    check(ccall((:armon_hip_sweep, lib[]), Cint, (Ptr{Cvoid}, Ref{SweepDesc}), p.device.ctx, desc))

"pool[1:4] = (ρ,u,v,E) holding the state, pool[5:8] their ping-pong partners, the rest spares → indices to keep"
function tune_placement!(p::HP, x_desc::SweepDesc, y_desc::SweepDesc, pool::Vector{<:HIPVector}; tries = 12)
    ptrs = [Ptr{Cvoid}(v.ptr) for v in pool]; picks = zeros(Cint, 8); times = zeros(Float64, tries)
    check(ccall((:armon_hip_tune_placement, lib[]), Cint,
        (Ptr{Cvoid}, Ref{SweepDesc}, Ref{SweepDesc}, Ptr{Ptr{Cvoid}}, Cint, Csize_t, Cint, Ptr{Cint}, Ptr{Float64}),
        p.device.ctx, x_desc, y_desc, ptrs, length(pool), sizeof(eltype(pool[1])) * pool[1].n, tries, picks, times))
    picks .+ 1, times
end

end # module
