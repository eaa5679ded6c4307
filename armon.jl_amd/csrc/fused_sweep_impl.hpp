// fused_sweep_impl.hpp — armon_hip_sweep: one directional sweep of solver_cycle (ref src/solver.jl:300-316)
// as ONE kernel launch: EOS → boundary mirror → fluxes → cell update → advection → projection, reading
// ρ,u,v,E once and writing them once (64 B per cell instead of the 352 B of the five staged passes),
// optionally followed by the dt/CFL reduction of the next cycle (ref src/reductions.jl:2-53) on the state
// it has just produced.
//
// Kernels:
//  * Y sweep (k_sweep_y): lane ↔ column, the register pipeline of sweep_pipeline.hpp marches along y.
//    Rows are x-contiguous, so every step is one fully coalesced row segment per wave; no LDS. A
//    workgroup owns 256 columns × one run of rows; 4 rows per lane are kept in flight. fp32 (tuned) uses
//    k_sweep_y2: two adjacent columns per lane, so that a wave still moves 512 B per row and array.
//  * X sweep, spatial form (k_sweep_x_dpp, default): lane ↔ cell(s) of a row, neighbours fetched with
//    DPP wavefront shifts (sweep_spatial.hpp); no LDS, no barrier, coalesced 16-B accesses.
//  * X sweep, LDS-transposed march (k_sweep_x_lds, alternative form kept for A/B measurements): a wave
//    transposes 64-row × 8-column tiles through LDS and each lane marches along its row with the same
//    pipeline as the Y sweep.
// Redundant work is confined to the LAG (≤4) cells at both ends of a run / strip. Block and strip origins are
// aligned to the 64-B sectors of the ghosted rows, stores are non-temporal (profiles/NOTES.md has the A/B numbers).
// Tuning knobs (tools/ab_sweep.py, parity tests), read from the environment once per context or set with
// armon_hip_set_tuning: ARMON_SWEEP_ALIGN, ARMON_XS_NITER, ARMON_Y_SEG, ARMON_Y_COLS1; compile-time: ARMON_NT_X, ARMON_NT_Y,
// ARMON_Y_PF, ARMON_Y_WAVES, ARMON_Y_BLOCK, ARMON_PROBE_NOCOMPUTE, ARMON_ONLY_HEADLINE.
#pragma once
#include "common.hpp"
#include "dt_state.hpp"
#include "reduce.hpp"
#include "sweep_pipeline.hpp"
#include "sweep_spatial.hpp"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

using namespace armon;

// This file is compiled twice: fused_sweep_f64.hip (real = double, armon_hip_sweep) and fused_sweep_f32.hip
// (real = float, armon_hip_sweep_f32); everything else lives in the translation unit's anonymous namespace.
#ifndef ARMON_SWEEP_REAL
#error "include through fused_sweep_f64.hip / fused_sweep_f32.hip"
#endif

namespace {

using real = ARMON_SWEEP_REAL;
using vec2 = std::conditional<std::is_same<real, double>::value, double2, float2>::type;

// Streaming hints (tuning macros, see tools/build_variant.sh): the state is read once and written once per sweep and is far
// larger than L2 + MALL. Bit 0: `nt` loads, bit 1: `nt` stores; one macro per sweep kernel family, because they answer
// differently (profiles/r05_ab_nt.txt, launches interleaved in one process):
//  * ARMON_NT_X (the X sweep's 16-B global loads / stores): 3. Non-temporal LOADS too since round 5 — for the flavour they
//    pay for, the tuned fp64 perfect-gas sweep (x_nt_loads below): 2.755 -> 2.68-2.72 ms at 16384², 0.682 -> 0.666 at 8192²,
//    0.342 -> 0.330 at 4096 x 8192, every size 2.4-3.8 % — a strip is read once by one wave, and lines that are not kept do not
//    push the rest out. The same hint costs the Bizarrium sweep 2.4 % and the fp32 one 3.3 % (their waves hold their loads
//    longer / share lines inside the workgroup) and does nothing for the exact flavour: they keep ordinary loads.
//  * ARMON_NT_Y (the Y march's buffer loads / stores): 2. Its runs re-read 2·LAG halo rows that the run below has just
//    loaded: with `nt` loads they are gone (+4 % at 16384² with nt stores, +21 % without).
#ifdef ARMON_NT
#define ARMON_NT_X ARMON_NT
#define ARMON_NT_Y ARMON_NT
#endif
#ifndef ARMON_NT_X
#define ARMON_NT_X 3
#endif
#ifndef ARMON_NT_Y
#define ARMON_NT_Y 2
#endif
typedef real vreal2 __attribute__((ext_vector_type(2)));
template <bool NT = false>
__device__ __forceinline__ vec2 ld2(const real* p)
{
    const vreal2* q = reinterpret_cast<const vreal2*>(p);
    const vreal2 v = NT ? __builtin_nontemporal_load(q) : *q;
    return vec2{v.x, v.y};
}
// which instantiations of the X sweep load their strips non-temporally (see ARMON_NT_X above)
constexpr bool x_nt_loads(int eos, bool exact)
{
    return (ARMON_NT_X & 1) && sizeof(real) == 8 && eos == ARMON_EOS_PERFECT_GAS && !exact;
}
__device__ __forceinline__ void st2(real* p, real x, real y)
{
    vreal2* q = reinterpret_cast<vreal2*>(p);
    const vreal2 v = {x, y};
    if (ARMON_NT_X & 2) __builtin_nontemporal_store(v, q);
    else *q = v;
}

struct sweep_args {
    int64_t nx, ny, row_len;       // real cells and array pitch (nx + 2g)
    int32_t g;                     // ghost layers
    int32_t bc_low, bc_high;       // mirror BC applied in-kernel on that side of the sweep axis
    int32_t emit;                  // bit 0: write p_out, bit 1: write c_out
    int32_t seg;                   // cells per run along the sweep axis (marching kernels)
    int32_t x_kernel;              // X sweep form: 0 spatial K=2, 3 spatial K=1, 2 LDS-transposed march
    int64_t o_lo, o_hi;            // cells to produce along the sweep axis: [o_lo, o_hi)
    int64_t x_first;               // X sweep: first cell of strip 0 (<= o_lo, sector-aligned in the ghosted row)
    int32_t xshift;                // Y sweep: columns the block origin is moved left (line-aligned row segments)
    int32_t xcd_remap;             // X sweep: XCD-aware workgroup placement (ARMON_X_XCD)
    int32_t gx = 0, gy = 0;        // X sweep: the launch's grid (gridDim comes from the dispatch packet: one more dependent scalar load)
    int32_t x_wg_along_x = 0;      // X sweep: a workgroup = kXSRows consecutive strips of one row (else: one strip of kXSRows rows)
    int32_t x_row_align = 0;       // X sweep: strip origins aligned row by row (row pitch not a multiple of a 64-B sector)
    int32_t y_sx = 0;              // Y sweep: rows are stored in sector-aligned windows handed over through LDS (see k_sweep_y)
    armon_dt_state* st = nullptr;   // device-resident time step (graph replay): dt is then a factor of st->current_dt
    real dt, dx, gamma;
    real inv_dx, dt_dx;            // 1 / dx and dt / dx in the run's precision (host; sweep_begin redoes dt / dx under a device-resident dt)
    real fa_low, ft_low, fa_high, ft_high;   // BC factors: axial / transverse velocity
    const real *rho_in, *ua_in, *ut_in, *E_in;    // ua = velocity along the sweep axis
    real *rho_out, *ua_out, *ut_out, *E_out;
    real *p_out, *c_out;
    real* partials;              // dt/CFL tracking: [2 * n_blocks] (max |u|±c, max |v|±c per workgroup)
    // whole-cycle kernel only: boundary of the SECOND (y) sweep — mirror flags and (u, v) factors per side
    int32_t bc_low_t = 0, bc_high_t = 0;
    real tu_low = 1, tv_low = 1, tu_high = 1, tv_high = 1;
};

// Source index and velocity factors of cell `j` (0-based real coordinate along the sweep axis, may be
// a ghost): physical boundaries mirror the inside (ref src/halo_exchange.jl:2-29), process boundaries
// read the ghost cells filled by the halo exchange.
__device__ __forceinline__ int64_t bc_source(const sweep_args& a, int64_t n, int64_t j, real& fa, real& ft)
{
    fa = 1;
    ft = 1;
    if (j < 0 && a.bc_low) {
        fa = a.fa_low;
        ft = a.ft_low;
        return -1 - j;
    }
    if (j >= n && a.bc_high) {
        fa = a.fa_high;
        ft = a.ft_high;
        return 2 * n - 1 - j;
    }
    return j;
}

// Device-resident time step (armon_dt_state): nothing to do once the time loop is over; otherwise the sweep's step is its
// factor times the cycle's step — the product the host forms in the run's precision (ref src/solver_state.jl:339-345) —
// and p is only materialised on the cycle the state machine marked as the last one.
__device__ __forceinline__ bool sweep_begin(sweep_args& a)
{
    if (a.st) {
        // nothing writes the state machine while a sweep runs (it steps in the fold kernel that follows), so it is read
        // through the constant address space: scalar loads whatever the compiler can prove about the rest of the kernel
        const auto* st = (const __attribute__((address_space(4))) armon_dt_state*)a.st;
        if (st->done) return false;
        a.dt = (real)st->current_dt * a.dt;
        a.dt_dx = a.dt / a.dx;
        if (!st->emit_p) a.emit &= ~1;
    }
    return true;
}

template <int... Is, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F&& f)
{
    (f(std::integral_constant<int, Is>{}), ...);
}

// Buffer addressing (T8): 128-bit resource descriptor in SGPRs + 32-bit lane offset + 32-bit scalar
// row offset. No per-access 64-bit VALU address arithmetic; advancing a row is one s_add_u32.
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xFFFFFFFFu, 0x00020000);
}
// gfx950 cache policy of a buffer access: bit 0 = sc0, bit 1 = nt, bit 4 = sc1 (ARMON_Y_AUX_LD / _ST: raw values, A/B builds)
#ifndef ARMON_Y_AUX_LD
#define ARMON_Y_AUX_LD ((ARMON_NT_Y & 1) ? 2 : 0)
#endif
#ifndef ARMON_Y_AUX_ST
#define ARMON_Y_AUX_ST ((ARMON_NT_Y & 2) ? 2 : 0)
#endif
constexpr int kAuxLoad = ARMON_Y_AUX_LD, kAuxStore = ARMON_Y_AUX_ST;
template <typename T> __device__ __forceinline__ T buf_load(rsrc_t r, unsigned voff, unsigned soff);
template <>
__device__ __forceinline__ double buf_load<double>(rsrc_t r, unsigned voff, unsigned soff)
{
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, kAuxLoad);
    return __builtin_bit_cast(double, v);
}
template <>
__device__ __forceinline__ float buf_load<float>(rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, kAuxLoad));
}
__device__ __forceinline__ float2 buf_load2(rsrc_t r, unsigned voff, unsigned soff)      // fp32 pairs only
{
    // NB: cast the whole vector. Extracting .x/.y from the builtin's <2 x i32> result makes hipcc (ROCm 7.2) narrow
    // the load to one dword and hand element 0 to both users (seen in the ISA: buffer_load_dword + op_sel_hi:[0,1]).
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, kAuxLoad));
    return float2{v.x, v.y};
}
__device__ __forceinline__ void buf_store2(rsrc_t r, unsigned voff, unsigned soff, float x, float y)
{
    const v2u v = {__builtin_bit_cast(unsigned int, x), __builtin_bit_cast(unsigned int, y)};
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, kAuxStore);
}
__device__ __forceinline__ void buf_store(rsrc_t r, unsigned voff, unsigned soff, double x)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x), r, voff, soff, kAuxStore);
}
__device__ __forceinline__ void buf_store(rsrc_t r, unsigned voff, unsigned soff, float x)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, x), r, voff, soff, kAuxStore);
}

// dt/CFL tracking (ref src/reductions.jl:13-20). The reference takes min over cells of
// min(dx/|max(|u+c|,|u-c|)|, dy/|max(|v+c|,|v-c|)|); IEEE division is monotonic, so that minimum equals
// min(dx / max_cells(..u..), dy / max_cells(..v..)) bit for bit: track the two maxima, divide once.
struct cfl_track {
    real au = 0, av = 0;
    __device__ __forceinline__ void add(real u, real v, real c)
    {
#ifdef ARMON_CFL_PLAIN_MAX     // A/B builds (tools/build_variant.sh): the floating-point maxima of rounds 1-2, which drop a NaN
        au = phys::mx(au, phys::abs_(phys::mx(phys::abs_(u + c), phys::abs_(u - c))));
        av = phys::mx(av, phys::abs_(phys::mx(phys::abs_(v + c), phys::abs_(v - c))));
#else
        // max(|u + c|, |u - c|) = |u| + |c| — bit for bit, not only in exact arithmetic: the two candidates are fl(|u| + |c|)
        // and |fl(|u| - |c|)| in some order, and rounding is monotone and symmetric. One addition instead of two, two
        // absolute values and a select. amax: unsigned maximum of the bit patterns — the same maximum for these
        // non-negative values, and a NaN sticks.
#ifdef ARMON_CFL_TWO_SUMS      // A/B builds: the reference's expression as it stands
        au = phys::amax(au, phys::abs_(phys::mx(phys::abs_(u + c), phys::abs_(u - c))));
        av = phys::amax(av, phys::abs_(phys::mx(phys::abs_(v + c), phys::abs_(v - c))));
#else
        au = phys::amax(au, phys::abs_(u) + phys::abs_(c));
        av = phys::amax(av, phys::abs_(v) + phys::abs_(c));
#endif
#endif
    }
};

template <int NWAVES>
__device__ __forceinline__ void cfl_block_store(const cfl_track& t, real* partials, int64_t block, int tid)
{
    __shared__ real lds[NWAVES];
    const real au = red::block_reduce<red::op_max, NWAVES>(t.au, lds, tid);
    const real av = red::block_reduce<red::op_max, NWAVES>(t.av, lds, tid);
    if (tid == 0) {
        partials[2 * block] = au;
        partials[2 * block + 1] = av;
    }
}

constexpr int kFoldBlocks = 512;

// first level of the fold when a launch leaves more partial pairs than one workgroup should walk
__global__ void __launch_bounds__(256)
k_fold_pairs(const real* __restrict__ partials, int64_t n, real* __restrict__ out)
{
    __shared__ real lds[4];
    real au = 0, av = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const vec2 v = ld2(partials + 2 * k);
        au = phys::amax(au, v.x);
        av = phys::amax(av, v.y);
    }
    au = red::block_reduce<red::op_max, 4>(au, lds, threadIdx.x);
    av = red::block_reduce<red::op_max, 4>(av, lds, threadIdx.x);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = au;
        out[2 * blockIdx.x + 1] = av;
    }
}

__global__ void __launch_bounds__(256)
k_fold_dt(const real* __restrict__ partials, int64_t n_blocks, real dx, real dy, real* __restrict__ out,
          int accumulate, armon_dt_state* st = nullptr)
{
    if (st && st->done) return;                     // the sweep did nothing either (graph replay past the last cycle)
    __shared__ real lds[4];
    real au = 0, av = 0;
    for (int64_t k = threadIdx.x; k < n_blocks; k += blockDim.x) {
        au = phys::amax(au, partials[2 * k]);
        av = phys::amax(av, partials[2 * k + 1]);
    }
    au = red::block_reduce<red::op_max, 4>(au, lds, threadIdx.x);
    av = red::block_reduce<red::op_max, 4>(av, lds, threadIdx.x);
    if (threadIdx.x == 0) {
        const real dt = phys::mn_nan(dx / au, dy / av);             // a NaN maximum gives a NaN step, and it wins
        out[0] = accumulate ? phys::mn_nan(out[0], dt) : dt;
        // graph replay: the state machine steps right here instead of in a kernel of its own (armon_dt_state::auto_step)
        if (st && st->auto_step && !accumulate)
            dt_state_step<real>(st, dt, (real)st->cfl, (real)st->maxtime, st->maxcycle, st->cst_dt, (real)st->Dt);
    }
}

// min(dx / max au, dy / max av) over `n` partial pairs into *out, on the context's stream
int fold_dt_launch(armon_ctx* ctx, real* partials, int64_t n, real dx, real dy, real* out, int accumulate,
                   armon_dt_state* st = nullptr)
{
    if (n > 16384) {
        real* level1 = partials + 2 * n;                      // room reserved by max_blocks()
        hipLaunchKernelGGL(k_fold_pairs, dim3(kFoldBlocks), dim3(256), 0, ctx->stream, partials, n, level1);
        int rc = check_launch("fold_pairs");
        if (rc != ARMON_OK) return rc;
        partials = level1;
        n = kFoldBlocks;
    }
    hipLaunchKernelGGL(k_fold_dt, dim3(1), dim3(256), 0, ctx->stream, partials, n, dx, dy, out, accumulate, st);
    return check_launch("fold_dt");
}

// ---- Y sweep ---------------------------------------------------------------------------------------
#ifndef ARMON_Y_BLOCK
#define ARMON_Y_BLOCK 256        // columns (= lanes) per workgroup of the Y march (tuning macro)
#endif
constexpr int kYBlock = ARMON_Y_BLOCK;
constexpr int kYSxBlock = 512;     // lanes per workgroup of the Y march with the LDS store exchange (sweep_args::y_sx)
#ifndef ARMON_Y_PF
#define ARMON_Y_PF 4             // rows prefetched ahead of the march (≤ 5: the cell ring has 8 slots)
#endif
#ifndef ARMON_Y_PRIO
#define ARMON_Y_PRIO 0           // > 0: issue priority of a step's loads (s_setprio around them)
#endif
#ifndef ARMON_Y_WAVES
#define ARMON_Y_WAVES 2          // minimum waves per SIMD the Y march is compiled for (register budget)
#endif

template <class PIPE, bool TRACK, int BLOCK = kYBlock, bool SX = false>
__global__ void __launch_bounds__(BLOCK, ARMON_Y_WAVES)
k_sweep_y(sweep_args a)
{
    if (!sweep_begin(a)) return;
    constexpr int LAG = PIPE::LAG;
    constexpr int PF = ARMON_Y_PF;   // rows in flight per lane, ahead of the march
    const int nx = (int)a.nx, ny = (int)a.ny, g = a.g;
    // The block origin is shifted left by a.xshift columns so that a wave's 512-B row segment starts on a
    // 64-B sector / 128-B line of the ghosted row instead of g cells into one (probe_access: -9 % time).
    const int xr = (int)(blockIdx.x * BLOCK + threadIdx.x) - a.xshift;
    const bool active = xr >= 0 && xr < nx;
    const int x = active ? xr : (xr < 0 ? 0 : nx - 1);   // idle lanes shadow an edge column and never store
    const int o_hi = (int)a.o_hi;
    const int o0 = (int)a.o_lo + (int)blockIdx.y * a.seg;
    const int o1 = (o0 + a.seg < o_hi) ? o0 + a.seg : o_hi;
    const int jb = o0 - LAG, je = o1 + LAG;

    // Descriptors are based at the first row this run touches, so every scalar row offset is a small
    // non-negative 32-bit number whatever the size of the arrays (mirrored rows lie inside the run).
    const unsigned colb = (unsigned)(x + g) * (unsigned)sizeof(real);
    const unsigned pitchb = (unsigned)a.row_len * (unsigned)sizeof(real);
    const int64_t in_base = (int64_t)(jb + g) * a.row_len, out_base = (int64_t)(o0 + g) * a.row_len;
    const rsrc_t r_rho = make_rsrc(a.rho_in + in_base), r_ua = make_rsrc(a.ua_in + in_base);
    const rsrc_t r_ut = make_rsrc(a.ut_in + in_base), r_E = make_rsrc(a.E_in + in_base);
    const rsrc_t w_rho = make_rsrc(a.rho_out + out_base), w_ua = make_rsrc(a.ua_out + out_base);
    const rsrc_t w_ut = make_rsrc(a.ut_out + out_base), w_E = make_rsrc(a.E_out + out_base);

    PIPE pipe(a.dt, a.dx, a.gamma, a.inv_dx, a.dt_dx);
    cfl_track cfl;

    int lj = jb;                     // next row to load and its offset from the run's first row
    unsigned lo_off = 0;
    unsigned so_off = (unsigned)(jb - LAG - o0) * pitchb;     // row j - LAG relative to row o0 (wraps until valid)

    // Store exchange (SX, chosen by sweep_args::y_sx; rows that do not all start on 64-B sectors, i.e. a pitch that is not a multiple of a sector).
    // A lane's column is fixed for the whole march, so on such rows every wave's 512-B store would begin and end inside a
    // sector, and it is the partial-sector STORES that cost (tools/probes/probe_ypitch.hip: misplaced loads +0.6 %, stores
    // +11 % at half a sector, +21 % on odd pitches). The workgroup's row is therefore passed through LDS: thread t stores
    // column (t - r) mod BLOCK of the workgroup, r = the row's phase in cells, so that all but the workgroup's two end
    // pieces are whole sectors; one barrier per row, two LDS buffers. The kernel is instantiated for it with workgroups of
    // kYSxBlock = 512 lanes (half as many end pieces: fp32 bench shape 1.55 -> 1.50 ms; without the exchange 256 lanes are
    // faster, profiles/r03_row_pitch_repairs.txt) and without any of this code for the usual, sector-aligned pitches.
    __shared__ real sx_lds[SX ? 2 : 1][SX ? 4 : 1][SX ? BLOCK : 1];
    static_assert((BLOCK & (BLOCK - 1)) == 0, "the store exchange wraps columns with a mask");
    constexpr int kSec = 64 / (int)sizeof(real);             // cells per sector
    const int c0 = (int)(blockIdx.x * BLOCK) - a.xshift;   // first column of the workgroup
    int sx_r = (int)(((int64_t)(jb - LAG + g) * a.row_len + g + c0) & (kSec - 1));    // phase of row j - LAG, j = jb
    const int sx_dr = (int)(a.row_len & (kSec - 1));
    int sx_buf = 0;

    // The state of row lj is loaded straight into the pipeline's cell ring, slot lj mod 8, PF steps before
    // the march reaches it. CHECKED steps handle everything (mirrored / clamped loads, masked stores,
    // p/c output); the steady state of a run uses the unchecked form: plain loads, unconditional stores.
    auto load = [&](auto slot, auto checked) {
        constexpr int K = decltype(slot)::value & 7;
        constexpr bool CHECKED = decltype(checked)::value;
        auto& dst = pipe.c[K];
        if (CHECKED) {
            const bool m_lo = lj < 0 && a.bc_low, m_hi = lj >= ny && a.bc_high;     // uniform, rare
            // physical boundary: mirror of the inside (ref src/halo_exchange.jl:2-29)
            const int src = m_lo ? -1 - lj : (m_hi ? 2 * ny - 1 - lj : lj);
            const unsigned off = (unsigned)(src - jb) * pitchb;
            const real fa = m_lo ? a.fa_low : (m_hi ? a.fa_high : real(1));
            const real ft = m_lo ? a.ft_low : (m_hi ? a.ft_high : real(1));
            dst.rho = buf_load<real>(r_rho, colb, off);
            dst.ua = buf_load<real>(r_ua, colb, off) * fa;
            dst.ut = buf_load<real>(r_ut, colb, off) * ft;
            dst.E = buf_load<real>(r_E, colb, off);
            if (lj + 1 < je) {       // stay on the last row once the run is exhausted (padding steps)
                lj++;
                lo_off += pitchb;
            }
        } else {
            dst.rho = buf_load<real>(r_rho, colb, lo_off);
            dst.ua = buf_load<real>(r_ua, colb, lo_off);
            dst.ut = buf_load<real>(r_ut, colb, lo_off);
            dst.E = buf_load<real>(r_E, colb, lo_off);
            lj++;
            lo_off += pitchb;
        }
    };
    using std::integral_constant;
    auto step = [&](auto ph, auto checked, int j) {
        constexpr int PH8 = decltype(ph)::value;
        constexpr bool CHECKED = decltype(checked)::value;
#if ARMON_Y_PRIO
        __builtin_amdgcn_s_setprio(ARMON_Y_PRIO);            // the row's loads go out ahead of the other wave's arithmetic
#endif
        load(integral_constant<int, PH8 + PF>{}, checked);   // row j + PF → slot (j + PF) mod 8
#if ARMON_Y_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        real p, c, c_lag;
#ifdef ARMON_PROBE_NOCOMPUTE   // calibration build: same loads/stores, no arithmetic (tools/build_variant.sh)
        p = c = c_lag = 0;
        const auto& cc = pipe.c[PH8 & 7];
        const fused::Out4<real> out{cc.rho, cc.ua, cc.ut, cc.E};
#else
        const fused::Out4<real> out = pipe.template advance<true, PH8>(p, c, c_lag);
#endif
        const int o = j - LAG;
        if (CHECKED) {
            if (a.emit && j >= o0 && j < o1 && active) {
                const unsigned off = so_off + LAG * pitchb;
                if (a.emit & 1) buf_store(make_rsrc(a.p_out + out_base), colb, off, p);
                if (a.emit & 2) buf_store(make_rsrc(a.c_out + out_base), colb, off, c);
            }
        }
        if (!CHECKED || (o >= o0 && o < o1)) {
            if constexpr (SX) {
                auto& L = sx_lds[sx_buf];
                L[0][threadIdx.x] = out.rho;
                L[1][threadIdx.x] = out.ua;
                L[2][threadIdx.x] = out.ut;
                L[3][threadIdx.x] = out.E;
                __syncthreads();
                const int ci = ((int)threadIdx.x - sx_r) & (BLOCK - 1);
                const int cx = c0 + ci;
                if (cx >= 0 && cx < nx) {
                    const unsigned cb = (unsigned)(cx + g) * (unsigned)sizeof(real);
                    buf_store(w_rho, cb, so_off, L[0][ci]);
                    buf_store(w_ua, cb, so_off, L[1][ci]);
                    buf_store(w_ut, cb, so_off, L[2][ci]);
                    buf_store(w_E, cb, so_off, L[3][ci]);
                }
                sx_buf ^= 1;
            } else if (active) {
                buf_store(w_rho, colb, so_off, out.rho);
                buf_store(w_ua, colb, so_off, out.ua);
                buf_store(w_ut, colb, so_off, out.ut);
                buf_store(w_E, colb, so_off, out.E);
            }
            if (TRACK) cfl.add(out.ut, out.ua, c_lag);      // Y sweep: ut = u, ua = v
        }
        so_off += pitchb;
        sx_r = (sx_r + sx_dr) & (kSec - 1);
    };
    auto run = [&](auto checked, int t0, int t1) {           // steps [t0, t1), both multiples of 8
        for (int t = t0; t < t1; t += 8)
            static_for(std::make_integer_sequence<int, 8>{}, [&](auto ph) { step(ph, checked, jb + t + decltype(ph)::value); });
    };

    const int T = je - jb;                                   // steps of the run
    const int T8 = (T + 7) & ~7;                             // … padded to the unroll
    const int P = (2 * LAG + 7) & ~7;                        // after P steps every step emits a valid cell
    // last step (exclusive) whose prefetch needs neither mirroring nor clamping and whose store is valid
    const int plain_end = (a.bc_high && ny < je ? ny : je) - jb - PF;
    int M = (T < plain_end ? T : plain_end) & ~7;
    if (M < P || (a.emit & 3)) M = P;                        // p/c output: everything through the checked form

    // The block origin shift leaves the last workgroup of a row mostly past the last column: a wave with no
    // column at all skips the march (it still joins the block reduction below with neutral values).
    // (not with the store exchange: every wave of the workgroup takes part in its barriers)
    const bool wave_idle = !SX && (int)(blockIdx.x * BLOCK + (threadIdx.x & ~63u)) - a.xshift >= nx;   // wave-uniform
    if (!wave_idle) {
        static_for(std::make_integer_sequence<int, PF>{}, [&](auto k) { load(k, std::true_type{}); });
        run(std::true_type{}, 0, P < T8 ? P : T8);
        run(std::false_type{}, P, M);
        run(std::true_type{}, M, T8);
    }

    if (TRACK) cfl_block_store<BLOCK / 64>(cfl, a.partials, (int64_t)blockIdx.y * gridDim.x + blockIdx.x, threadIdx.x);
}

// Two columns per lane: the fp32 form of the Y march. With 4-B elements one column per lane moves only 256 B per
// wave and instruction; two adjacent columns per lane (8-B accesses, two independent pipelines = twice the ILP at
// the register cost of one fp64 pipeline) restore the 512-B row segments of the fp64 kernel.
template <class PIPE, bool TRACK, int BLOCK = kYBlock, bool SX = false>
__global__ void __launch_bounds__(BLOCK, ARMON_Y_WAVES)
k_sweep_y2(sweep_args a)
{
    if (!sweep_begin(a)) return;
    constexpr int LAG = PIPE::LAG;
    constexpr int PF = ARMON_Y_PF;   // rows in flight per lane, ahead of the march
    const int nx = (int)a.nx, ny = (int)a.ny, g = a.g;
    // Two adjacent columns per lane (nx, g and a.xshift are even here): xr, xr + 1; 8-B accesses.
    const int xr = (int)(blockIdx.x * BLOCK + threadIdx.x) * 2 - a.xshift;
    const bool active = xr >= 0 && xr < nx;
    const int x = active ? xr : (xr < 0 ? 0 : nx - 2);   // idle lanes shadow an edge pair and never store
    const int o_hi = (int)a.o_hi;
    const int o0 = (int)a.o_lo + (int)blockIdx.y * a.seg;
    const int o1 = (o0 + a.seg < o_hi) ? o0 + a.seg : o_hi;
    const int jb = o0 - LAG, je = o1 + LAG;

    // Descriptors are based at the first row this run touches, so every scalar row offset is a small
    // non-negative 32-bit number whatever the size of the arrays (mirrored rows lie inside the run).
    const unsigned colb = (unsigned)(x + g) * (unsigned)sizeof(real);
    const unsigned pitchb = (unsigned)a.row_len * (unsigned)sizeof(real);
    const int64_t in_base = (int64_t)(jb + g) * a.row_len, out_base = (int64_t)(o0 + g) * a.row_len;
    const rsrc_t r_rho = make_rsrc(a.rho_in + in_base), r_ua = make_rsrc(a.ua_in + in_base);
    const rsrc_t r_ut = make_rsrc(a.ut_in + in_base), r_E = make_rsrc(a.E_in + in_base);
    const rsrc_t w_rho = make_rsrc(a.rho_out + out_base), w_ua = make_rsrc(a.ua_out + out_base);
    const rsrc_t w_ut = make_rsrc(a.ut_out + out_base), w_E = make_rsrc(a.E_out + out_base);

    // the two columns of a lane are ONE pipeline on 2-vectors: packed v_pk_* arithmetic (sweep_pipeline.hpp, float2v)
    using V = fused::fast::float2v;
    using PIPEV = fused::PipeFast<PIPE::SCHEME, PIPE::LIM, PIPE::PROJ, PIPE::EOS, V>;
    PIPEV pipe(a.dt, a.dx, a.gamma, a.inv_dx, a.dt_dx);
    cfl_track cfl;

    int lj = jb;                     // next row to load and its offset from the run's first row
    unsigned lo_off = 0;
    unsigned so_off = (unsigned)(jb - LAG - o0) * pitchb;     // row j - LAG relative to row o0 (wraps until valid)

    // store exchange as in k_sweep_y, in units of a lane's column PAIR (8 B; eight pairs per sector)
    __shared__ float2 sx_lds[SX ? 2 : 1][SX ? 4 : 1][SX ? BLOCK : 1];
    static_assert((BLOCK & (BLOCK - 1)) == 0, "the store exchange wraps columns with a mask");
    const int c0 = (int)(blockIdx.x * BLOCK) * 2 - a.xshift;     // first column of the workgroup (even)
    int sx_r = (int)((((int64_t)(jb - LAG + g) * a.row_len + g + c0) >> 1) & 7);
    const int sx_dr = (int)((a.row_len >> 1) & 7);
    int sx_buf = 0;

    // The state of row lj is loaded straight into the pipeline's cell ring, slot lj mod 8, PF steps before
    // the march reaches it. CHECKED steps handle everything (mirrored / clamped loads, masked stores,
    // p/c output); the steady state of a run uses the unchecked form: plain loads, unconditional stores.
    auto load = [&](auto slot, auto checked) {
        constexpr int K = decltype(slot)::value & 7;
        constexpr bool CHECKED = decltype(checked)::value;
        auto& dst = pipe.c[K];
        auto vec = [](float2 q) { return V{q.x, q.y}; };
        auto put = [&](unsigned off, real fa, real ft) {
            dst.rho = vec(buf_load2(r_rho, colb, off));
            dst.ua = vec(buf_load2(r_ua, colb, off)) * fa;
            dst.ut = vec(buf_load2(r_ut, colb, off)) * ft;
            dst.E = vec(buf_load2(r_E, colb, off));
        };
        if (CHECKED) {
            const bool m_lo = lj < 0 && a.bc_low, m_hi = lj >= ny && a.bc_high;     // uniform, rare
            // physical boundary: mirror of the inside (ref src/halo_exchange.jl:2-29)
            const int src = m_lo ? -1 - lj : (m_hi ? 2 * ny - 1 - lj : lj);
            const unsigned off = (unsigned)(src - jb) * pitchb;
            const real fa = m_lo ? a.fa_low : (m_hi ? a.fa_high : real(1));
            const real ft = m_lo ? a.ft_low : (m_hi ? a.ft_high : real(1));
            put(off, fa, ft);
            if (lj + 1 < je) {       // stay on the last row once the run is exhausted (padding steps)
                lj++;
                lo_off += pitchb;
            }
        } else {
            dst.rho = vec(buf_load2(r_rho, colb, lo_off));
            dst.ua = vec(buf_load2(r_ua, colb, lo_off));
            dst.ut = vec(buf_load2(r_ut, colb, lo_off));
            dst.E = vec(buf_load2(r_E, colb, lo_off));
            lj++;
            lo_off += pitchb;
        }
    };
    using std::integral_constant;
    auto step = [&](auto ph, auto checked, int j) {
        constexpr int PH8 = decltype(ph)::value;
        constexpr bool CHECKED = decltype(checked)::value;
        load(integral_constant<int, PH8 + PF>{}, checked);   // row j + PF → slot (j + PF) mod 8
        V p, c, c_lag;
#ifdef ARMON_PROBE_NOCOMPUTE   // calibration build: same loads/stores, no arithmetic (tools/build_variant.sh)
        p = c = c_lag = V(0.f);
        const auto& cc = pipe.c[PH8 & 7];
        const fused::Out4<V> out{cc.rho, cc.ua, cc.ut, cc.E};
#else
        const fused::Out4<V> out = pipe.template advance<true, PH8>(p, c, c_lag);
#endif
        const int o = j - LAG;
        if (CHECKED) {
            if (a.emit && j >= o0 && j < o1 && active) {
                const unsigned off = so_off + LAG * pitchb;
                if (a.emit & 1) buf_store2(make_rsrc(a.p_out + out_base), colb, off, p.x, p.y);
                if (a.emit & 2) buf_store2(make_rsrc(a.c_out + out_base), colb, off, c.x, c.y);
            }
        }
        if (!CHECKED || (o >= o0 && o < o1)) {
            if constexpr (SX) {
                auto& L = sx_lds[sx_buf];
                L[0][threadIdx.x] = float2{out.rho.x, out.rho.y};
                L[1][threadIdx.x] = float2{out.ua.x, out.ua.y};
                L[2][threadIdx.x] = float2{out.ut.x, out.ut.y};
                L[3][threadIdx.x] = float2{out.E.x, out.E.y};
                __syncthreads();
                const int ci = ((int)threadIdx.x - sx_r) & (BLOCK - 1);
                const int cx = c0 + 2 * ci;
                if (cx >= 0 && cx < nx) {
                    const unsigned cb = (unsigned)(cx + g) * (unsigned)sizeof(real);
                    const float2 q0 = L[0][ci], q1 = L[1][ci], q2 = L[2][ci], q3 = L[3][ci];
                    buf_store2(w_rho, cb, so_off, q0.x, q0.y);
                    buf_store2(w_ua, cb, so_off, q1.x, q1.y);
                    buf_store2(w_ut, cb, so_off, q2.x, q2.y);
                    buf_store2(w_E, cb, so_off, q3.x, q3.y);
                }
                sx_buf ^= 1;
            } else if (active) {
                buf_store2(w_rho, colb, so_off, out.rho.x, out.rho.y);
                buf_store2(w_ua, colb, so_off, out.ua.x, out.ua.y);
                buf_store2(w_ut, colb, so_off, out.ut.x, out.ut.y);
                buf_store2(w_E, colb, so_off, out.E.x, out.E.y);
            }
            if (TRACK) {                                     // Y sweep: ut = u, ua = v
                cfl.add(out.ut.x, out.ua.x, c_lag.x);
                cfl.add(out.ut.y, out.ua.y, c_lag.y);
            }
        }
        so_off += pitchb;
        sx_r = (sx_r + sx_dr) & 7;
    };
    auto run = [&](auto checked, int t0, int t1) {           // steps [t0, t1), both multiples of 8
        for (int t = t0; t < t1; t += 8)
            static_for(std::make_integer_sequence<int, 8>{}, [&](auto ph) { step(ph, checked, jb + t + decltype(ph)::value); });
    };

    const int T = je - jb;                                   // steps of the run
    const int T8 = (T + 7) & ~7;                             // … padded to the unroll
    const int P = (2 * LAG + 7) & ~7;                        // after P steps every step emits a valid cell
    // last step (exclusive) whose prefetch needs neither mirroring nor clamping and whose store is valid
    const int plain_end = (a.bc_high && ny < je ? ny : je) - jb - PF;
    int M = (T < plain_end ? T : plain_end) & ~7;
    if (M < P || (a.emit & 3)) M = P;                        // p/c output: everything through the checked form

    static_for(std::make_integer_sequence<int, PF>{}, [&](auto k) { load(k, std::true_type{}); });
    run(std::true_type{}, 0, P < T8 ? P : T8);
    run(std::false_type{}, P, M);
    run(std::true_type{}, M, T8);

    if (TRACK) cfl_block_store<BLOCK / 64>(cfl, a.partials, (int64_t)blockIdx.y * gridDim.x + blockIdx.x, threadIdx.x);
}

// ---- X sweep, spatial form (lanes along x, DPP neighbour exchange) -------------------------------------
// blockDim = (64, kXSRows): one wave per row, kXSRows consecutive rows per workgroup. Each wave walks
// NITER strips of 64*K cells along its row; a strip yields 64*K - 2*HALO new cells.
#ifndef ARMON_XS_ROWS
#define ARMON_XS_ROWS 4          // rows (= waves) per workgroup of the X sweep (tuning macro)
#endif
constexpr int kXSRows = ARMON_XS_ROWS;
// ARMON_X_PRIO > 0: a new wave runs its prologue and issues its strip's loads at that priority (s_setprio), ahead of the
// arithmetic of the older waves of its SIMD, and drops to 0 for its own arithmetic: with four waves per SIMD and ~600 vector
// instructions per strip, a newcomer otherwise waits its turn behind three compute phases before anything of its own is in
// flight. Bizarrium (the longest compute phase) 2.91 -> 2.82 ms at 16384², the copy's rate; perfect gas, already at that
// rate, unchanged (profiles/r04_ab_x_prologue.txt).
#ifndef ARMON_X_PRIO
#define ARMON_X_PRIO 3
#endif
// One strip per wave (SINGLE): load, sweep, store, exit — no second register set for a prefetched strip, 104-119 VGPRs, 4
// waves per SIMD; the other waves of the SIMD hide the load. Rounds 1-2 ran TWO strips per wave with the second one's 32
// input registers prefetched during the first (166-184 VGPRs, 2-3 waves per SIMD): A/B in one process
// (profiles/r03_ab_x_single_strip.txt) tuned 2.875 -> 2.817 ms, tuned with dt tracking 3.240 -> 2.923 (2 -> 4 waves), exact
// 3.444 -> 3.083, exact with tracking 3.818 -> 3.435, a 4096 x 8192 tile 0.415 -> 0.367. The multi-strip form stays in
// the A/B build only (-DARMON_ALT_KERNELS, knob ARMON_XS_NITER > 1).
constexpr int kXSNiter = 1;
// ROW = 1 (with K = 1): the NARROW form for the LAG-wide boundary strips of a tile (partial sweeps of <= 8 cells along x,
// which follow the halo exchange): a wave holds FOUR rows of 16 lanes instead of one row of 64 or 128 cells, neighbours by
// row_shr / row_shl shifts. A 4-cell strip then costs 16 loaded cells per row instead of 128 (a strip of a 4096-cell-wide
// tile: 3 % of the whole sweep -> 0.4 %). Same arithmetic per cell, hence the same bits.
template <int SCHEME, int LIM, int PROJ, int EOS, bool EXACT, int K, bool TRACK, bool SINGLE, int ROW = 0>
__device__ __forceinline__ void sweep_x_dpp_body(const sweep_args& a, int niter)
{
    using SW = fused::SpatialSweep<SCHEME, LIM, PROJ, EOS, EXACT, K, real, ROW>;
    using St = fused::Strip<K, real>;
    constexpr int LAG = SW::LAG;
    static_assert(ROW == 0 || K == 1, "the narrow form holds one cell per lane");
    // K = 2: 4 cells whatever the scheme (LAG <= 4), so that STRIDE = 120 cells = 15 whole 64-B sectors and every
    // strip's stores stay sector-aligned (Godunov + euler, LAG 2: -5.6 % time against HALO = 2, STRIDE = 124)
    constexpr int HALO = (K == 1) ? LAG : 4;
    static_assert(LAG <= 4, "strip halo");
    constexpr int WIDTH = ROW ? 16 : 64 * K;
    constexpr int STRIDE = WIDTH - 2 * HALO;
    constexpr int RW = ROW ? 4 : 1;                           // rows per wave

    const int lane = ROW ? (threadIdx.x & 15) : threadIdx.x;  // position in the strip
    // XCD-aware placement of the workgroups: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own
    // L2), so with the plain mapping two strips that are neighbours along x — they share their 4-cell halos — always
    // sit on different XCDs and both fetch the shared sectors from HBM. Within every group of 8 rows of workgroups,
    // XCD k (ids ≡ k mod 8) gets the whole row k: neighbouring strips then follow each other on the same L2.
    unsigned vbx = blockIdx.x, vby = blockIdx.y;
    if (a.xcd_remap) {
        const unsigned G = (unsigned)a.gx, chunk = vby & ~7u;
        if (chunk + 8 <= (unsigned)a.gy) {                     // whole groups only: the last rows keep the plain mapping
            const unsigned local = (vby - chunk) * G + vbx;    // 0 .. 8G-1 in dispatch order
            vby = chunk + (local & 7u);
            vbx = local >> 3;
        }
    }
    // Which strip and row a wave takes. a.x_wg_along_x: the kXSRows waves of a workgroup take kXSRows CONSECUTIVE STRIPS of
    // ONE row (the 128-B lines two neighbouring strips share — a strip's loads start 32 B before its sector-aligned stores —
    // are then fetched once per workgroup, from its CU's L1, instead of by two workgroups on two XCDs); otherwise one strip
    // of kXSRows consecutive rows (the narrow form, blocks of more than 65535 rows, the A/B forms).
    const bool along_x = SINGLE && ROW == 0 && a.x_wg_along_x;
    const int64_t strip0 = along_x ? (int64_t)vbx * kXSRows + threadIdx.y : (int64_t)vbx * niter;
    const int64_t row_r = along_x ? (int64_t)vby
                                  : ((int64_t)vby * kXSRows + threadIdx.y) * RW + (ROW ? (threadIdx.x >> 4) : 0);
    const bool row_ok = row_r < a.ny;                         // whole wave (ROW: a row of 16 lanes)
    const int64_t row = row_ok ? row_r : a.ny - 1;
    const int64_t row_off = (row + a.g) * a.row_len + a.g;
    const real* in[4] = {a.rho_in + row_off, a.ua_in + row_off, a.ut_in + row_off, a.E_in + row_off};
    real* out[4] = {a.rho_out + row_off, a.ua_out + row_off, a.ut_out + row_off, a.E_out + row_off};
    // 16-B accesses need every lane pair on an even cell of an even-pitched row; strip origins are multiples of 8 cells
    // from the row start (x_first), so an odd o_lo (partial sweeps with LAG = 3) only masks half of one pair
    bool vec_ok = (K == 2) && (a.row_len % 2 == 0) && ((a.x_first + a.g) % 2 == 0);         // uniform
    // When the row pitch is not a multiple of a sector (8 doubles, 16 floats) the rows start at different places of their 64-B sectors and no single
    // origin aligns them all (fp64 at 16388 cells per row: X 1.05x the copy, profiles/r03_row_pitch.txt): the origin is then
    // taken row by row, less than a sector below o_lo, where that row's stores start on a sector (arrays start on one). Every
    // lane pair then sits on 16 B (fp32: 8 B) whatever the parity of the pitch.
    int64_t x_first = a.x_first;
    if (ROW == 0 && a.x_row_align) {
        // (a wave's row is uniform, which the compiler cannot know: without readfirstlane every strip index below becomes
        // 64-bit vector arithmetic — +3 % on the VALU-bound exact flavour)
        x_first = a.o_lo - __builtin_amdgcn_readfirstlane((int)(row_off + a.o_lo) & (64 / (int)sizeof(real) - 1));
        vec_ok = (K == 2);
    }

    SW sw{a.dt, a.dx, a.gamma, a.inv_dx, a.dt_dx};
    cfl_track cfl;
    // Strip origins are aligned so that a strip's stores start on a 64-B sector of the ghosted row (for the
    // usual STRIDE = 120 = 15 sectors); the first strip of a row is then a short one (stores masked below o_lo).
    const int64_t w_first = x_first + strip0 * STRIDE;
    // Strips are real-buffered in registers: the loads of strip it+1 are issued before strip it is
    // computed (the loop is unrolled by the two buffers, so no loaded register is ever copied).
    St buf[2][4];
    auto load_strip = [&](auto slot, int it) {
        constexpr int B = decltype(slot)::value;
        St& rho = buf[B][0];
        St& ua = buf[B][1];
        St& ut = buf[B][2];
        St& E = buf[B][3];
        const int64_t cb = w_first + (int64_t)it * STRIDE - HALO;     // first cell of the strip
        const int64_t j0 = cb + (int64_t)lane * K;                    // this lane's first cell
        const bool interior = cb >= 0 && cb + WIDTH <= a.nx;          // uniform: no ghost, no clamping
        if (interior && (K == 1 || vec_ok)) {
            if (K == 2) {
                constexpr bool NT = x_nt_loads(EOS, EXACT);
                const vec2 r = ld2<NT>(in[0] + j0);
                const vec2 u = ld2<NT>(in[1] + j0);
                const vec2 v = ld2<NT>(in[2] + j0);
                const vec2 e = ld2<NT>(in[3] + j0);
                rho.v[0] = r.x; rho.v[K - 1] = r.y;
                ua.v[0] = u.x; ua.v[K - 1] = u.y;
                ut.v[0] = v.x; ut.v[K - 1] = v.y;
                E.v[0] = e.x; E.v[K - 1] = e.y;
            } else {
                rho.v[0] = in[0][j0]; ua.v[0] = in[1][j0]; ut.v[0] = in[2][j0]; E.v[0] = in[3][j0];
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) {
                // clamp into the block (ghosts included), then mirror physical boundaries
                int64_t j = j0 + k;
                j = j < -(int64_t)a.g ? -(int64_t)a.g : (j > a.nx + a.g - 1 ? a.nx + a.g - 1 : j);
                real fa, ft;
                const int64_t src = bc_source(a, a.nx, j, fa, ft);
                rho.v[k] = in[0][src];
                ua.v[k] = in[1][src] * fa;
                ut.v[k] = in[2][src] * ft;
                E.v[k] = in[3][src];
            }
        }
    };
    auto strip_exists = [&](int it) { return it < niter && w_first + (int64_t)it * STRIDE < a.o_hi; };
    auto do_strip = [&](auto slot, int it) {
        constexpr int B = decltype(slot)::value;
        if (!SINGLE && strip_exists(it + 1)) load_strip(std::integral_constant<int, 1 - B>{}, it + 1);
        const int64_t w0 = w_first + (int64_t)it * STRIDE;    // first cell this strip produces
        const int64_t j0 = w0 - HALO + (int64_t)lane * K;

        St o_rho, o_u, o_v, o_E, p, cs;
#ifdef ARMON_PROBE_NOCOMPUTE
        o_rho = buf[B][0]; o_u = buf[B][1]; o_v = buf[B][2]; o_E = buf[B][3]; p = buf[B][0]; cs = buf[B][0];
#else
        sw.run(buf[B][0], buf[B][1], buf[B][2], buf[B][3], o_rho, o_u, o_v, o_E, p, cs);
#endif

        // cells this lane may store: inside the strip's valid window and inside the block
        const int64_t hi = (w0 + STRIDE < a.o_hi) ? w0 + STRIDE : a.o_hi;
        const int64_t lo = w0 > a.o_lo ? w0 : a.o_lo;
        if (K == 2 && vec_ok && j0 >= lo && j0 + 1 < hi) {
            st2(out[0] + j0, o_rho.v[0], o_rho.v[K - 1]);
            st2(out[1] + j0, o_u.v[0], o_u.v[K - 1]);
            st2(out[2] + j0, o_v.v[0], o_v.v[K - 1]);
            st2(out[3] + j0, o_E.v[0], o_E.v[K - 1]);
            if (a.emit) {
                if (a.emit & 1) st2(a.p_out + row_off + j0, p.v[0], p.v[K - 1]);
                if (a.emit & 2) st2(a.c_out + row_off + j0, cs.v[0], cs.v[K - 1]);
            }
            if (TRACK) {
                cfl.add(o_u.v[0], o_v.v[0], cs.v[0]);
                cfl.add(o_u.v[K - 1], o_v.v[K - 1], cs.v[K - 1]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int64_t j = j0 + k;
                if (j >= lo && j < hi) {
                    out[0][j] = o_rho.v[k];
                    out[1][j] = o_u.v[k];
                    out[2][j] = o_v.v[k];
                    out[3][j] = o_E.v[k];
                    if (a.emit & 1) (a.p_out + row_off)[j] = p.v[k];
                    if (a.emit & 2) (a.c_out + row_off)[j] = cs.v[k];
                    if (TRACK) cfl.add(o_u.v[k], o_v.v[k], cs.v[k]);
                }
            }
        }
    };
    if (SINGLE) {
        if (row_ok && strip_exists(0)) {
            load_strip(std::integral_constant<int, 0>{}, 0);
#if ARMON_X_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            do_strip(std::integral_constant<int, 0>{}, 0);
        }
    } else if (row_ok && strip_exists(0)) {
        load_strip(std::integral_constant<int, 0>{}, 0);
#if ARMON_X_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        for (int it = 0; strip_exists(it); it += 2) {
            do_strip(std::integral_constant<int, 0>{}, it);
            if (!strip_exists(it + 1)) break;
            do_strip(std::integral_constant<int, 1>{}, it + 1);
        }
    }
    // The waves of this kernel are short-lived (2 strips): a block-level LDS reduction + one partial per block cost
    // more than the strips themselves (282k blocks at 16384²: +1.1 ms), and atomic maxima into shared slots go to the
    // memory side on this part (2.3 M of them: +0.5 ms). Each wave simply stores its two maxima in its own slot
    // (18 MB at 16384², 0.1 % of the sweep's traffic); fold_dt_launch reduces them in two levels.
    if (TRACK) {
        const real au = red::wave_reduce<red::op_max>(cfl.au), av = red::wave_reduce<red::op_max>(cfl.av);
        if (threadIdx.x == 0) {
            const int64_t wave_id = ((int64_t)blockIdx.y * a.gx + blockIdx.x) * kXSRows + threadIdx.y;
            st2(a.partials + 2 * wave_id, au, av);
        }
    }
}

#ifndef ARMON_XS_WAVES
#define ARMON_XS_WAVES 1         // minimum waves per SIMD the X sweep is compiled for (tuning macro; 3 = cap at 168 VGPRs)
#endif
// hipcc loads a kernel argument from the kernarg segment in the basic block that first uses it: this kernel's prologue then
// holds FOUR dependent scalar-load round trips (st, the geometry, niter, the pointers) before its first vector load — and a
// wave of the one-strip form lives for one strip, so that latency is paid per strip with nothing of the wave's own in
// flight. Naming every field the hot path reads in one empty asm statement at entry makes the compiler issue all the
// s_loads together, behind ONE wait (ARMON_X_PRELOAD=0: the lazy form, for A/B).
#ifndef ARMON_X_PRELOAD
#define ARMON_X_PRELOAD 1
#endif
__device__ __forceinline__ void preload_x_args(const sweep_args& a, int niter)
{
#if ARMON_X_PRELOAD
    asm volatile("" ::"s"(a.nx), "s"(a.ny), "s"(a.row_len), "s"(a.g), "s"(a.bc_low), "s"(a.bc_high), "s"(a.emit), "s"(a.o_lo),
                 "s"(a.o_hi), "s"(a.x_first), "s"(a.xcd_remap), "s"(a.x_wg_along_x), "s"(a.x_row_align), "s"(niter), "s"(a.gx), "s"(a.gy));
    asm volatile("" ::"s"(a.dt), "s"(a.dx), "s"(a.gamma), "s"(a.inv_dx), "s"(a.dt_dx), "s"(a.rho_in), "s"(a.ua_in), "s"(a.ut_in),
                 "s"(a.E_in), "s"(a.rho_out), "s"(a.ua_out), "s"(a.ut_out), "s"(a.E_out));
#endif
}

template <int SCHEME, int LIM, int PROJ, int EOS, bool EXACT, int K, bool TRACK, bool SINGLE = true, int ROW = 0>
__global__ void __launch_bounds__(64 * kXSRows, ARMON_XS_WAVES)
k_sweep_x_dpp(sweep_args a, int niter)
{
#if ARMON_X_PRIO
    __builtin_amdgcn_s_setprio(ARMON_X_PRIO);     // until the strip's loads are issued (sweep_x_dpp_body lowers it again)
#endif
    preload_x_args(a, niter);
    if (!sweep_begin(a)) return;
    sweep_x_dpp_body<SCHEME, LIM, PROJ, EOS, EXACT, K, TRACK, SINGLE, ROW>(a, niter);
}

// =====================================================================================================================
// Measured-and-rejected alternatives: the whole-cycle kernels k_cycle_xy / k_cycle_pc, the
// LDS-transposed X march k_sweep_x_lds (and, in launch(), the one-cell-per-lane form of the DPP sweep). They are correct and
// tested but 1.3-4x slower than what the solver runs, so they are only compiled with -DARMON_ALT_KERNELS, into
// libarmon_hip_alt.so (build.py), which the tests and tools that exercise them load; the product library carries none.
#ifdef ARMON_ALT_KERNELS
#include "fused_sweep_alt_kernels.hpp"
#endif  // ARMON_ALT_KERNELS

// ---- launch ----------------------------------------------------------------------------------------------
template <class PIPE, bool TRACK>
int launch(armon_ctx* ctx, const sweep_args& a, int axis, int64_t* n_blocks)
{
    const int64_t n_out = a.o_hi - a.o_lo;
    if constexpr (std::is_same<real, float>::value && !PIPE::kExact) {
        // fp32, tuned arithmetic: two columns per lane when every row is 8-B aligned (even pitch and ghost width).
        // (Not instantiated for the exact flavour: build time; the tests require it to equal the one-column kernel.)
        if (axis == ARMON_AXIS_Y && a.nx % 2 == 0 && a.g % 2 == 0 && a.nx >= 2 && !ctx->tune_y_cols1) {
            const int block = a.y_sx ? kYSxBlock : kYBlock;
            dim3 grid((unsigned)((a.nx + a.xshift + 2 * block - 1) / (2 * block)), (unsigned)((n_out + a.seg - 1) / a.seg));
            *n_blocks = (int64_t)grid.x * grid.y;
            if (a.y_sx)
                hipLaunchKernelGGL((k_sweep_y2<PIPE, TRACK, kYSxBlock, true>), grid, dim3(block), 0, ctx->stream, a);
            else
                hipLaunchKernelGGL((k_sweep_y2<PIPE, TRACK>), grid, dim3(block), 0, ctx->stream, a);
            return check_launch("sweep_y2");
        }
    }
    if (axis == ARMON_AXIS_Y) {
        const int block = a.y_sx ? kYSxBlock : kYBlock;
        dim3 grid((unsigned)((a.nx + a.xshift + block - 1) / block), (unsigned)((n_out + a.seg - 1) / a.seg));
        *n_blocks = (int64_t)grid.x * grid.y;
        if constexpr (!PIPE::kExact) {                       // (y_sx is never set for the exact flavour: library size)
            if (a.y_sx) {
                hipLaunchKernelGGL((k_sweep_y<PIPE, TRACK, kYSxBlock, true>), grid, dim3(block), 0, ctx->stream, a);
                return check_launch("sweep_y (store exchange)");
            }
        }
        hipLaunchKernelGGL((k_sweep_y<PIPE, TRACK>), grid, dim3(block), 0, ctx->stream, a);
        return check_launch("sweep_y");
    }
#if defined(ARMON_ALT_KERNELS) && !defined(ARMON_ONLY_HEADLINE)
    if (a.x_kernel == 2) {
        dim3 grid((unsigned)((n_out + a.seg - 1) / a.seg), (unsigned)((a.ny + kXRows - 1) / kXRows));
        *n_blocks = (int64_t)grid.x * grid.y;
        const size_t lds = (size_t)(a.emit ? 6 : 4) * kXRows * (kXChunk + 1) * sizeof(real);
        hipLaunchKernelGGL((k_sweep_x_lds<PIPE, kXChunk, TRACK>), grid, dim3(kXRows), lds, ctx->stream, a);
        return check_launch("sweep_x_lds");
    }
#endif
    // one strip per wave; the A/B build also carries the multi-strip form with its prefetch buffer (ARMON_XS_NITER > 1)
    int niter = 1;
#if defined(ARMON_ALT_KERNELS) && !defined(ARMON_ONLY_HEADLINE)
    if (ctx->tune_xs_niter > 1) niter = ctx->tune_xs_niter;
#elif defined(ARMON_XS_MULTI)      // tools/build_variant.sh: the round-2 form alone, for A/B timing
    niter = ctx->tune_xs_niter > 0 ? ctx->tune_xs_niter : 2;
#endif
    const bool k1 = a.x_kernel == 3;
    // the boundary strips of a tile (partial sweeps of at most 8 cells): four rows of 16 lanes per wave
    if (a.x_kernel == 0 && a.o_hi - a.o_lo <= 8) {
        const int64_t per_block = 16 - 2 * PIPE::LAG;
        dim3 grid((unsigned)((a.o_hi - a.x_first + per_block - 1) / per_block), (unsigned)((a.ny + 4 * kXSRows - 1) / (4 * kXSRows)));
        *n_blocks = (int64_t)grid.x * grid.y * kXSRows;
        sweep_args b = a;
        b.gx = (int32_t)grid.x;
        b.gy = (int32_t)grid.y;
        hipLaunchKernelGGL((k_sweep_x_dpp<PIPE::SCHEME, PIPE::LIM, PIPE::PROJ, PIPE::EOS, PIPE::kExact, 1, TRACK, true, 1>),
                           grid, dim3(64, kXSRows), 0, ctx->stream, b, 1);
        return check_launch("sweep_x_dpp (narrow)");
    }
    const int halo = k1 ? PIPE::LAG : 4;
    const int64_t per_block = (int64_t)niter * (64 * (k1 ? 1 : 2) - 2 * halo);
    const int64_t x_lowest = a.x_row_align ? a.o_lo - (64 / (int64_t)sizeof(real) - 1) : a.x_first;   // strips are counted from the lowest origin of any row
    dim3 grid((unsigned)((a.o_hi - x_lowest + per_block - 1) / per_block), (unsigned)((a.ny + kXSRows - 1) / kXSRows));
    if (a.x_wg_along_x && !k1 && niter == 1)                  // kXSRows strips of one row per workgroup
        grid = dim3((unsigned)((a.o_hi - x_lowest + kXSRows * per_block - 1) / (kXSRows * per_block)), (unsigned)a.ny);
    *n_blocks = (int64_t)grid.x * grid.y * kXSRows;          // one pair of maxima per wave
    sweep_args b = a;
    b.gx = (int32_t)grid.x;
    b.gy = (int32_t)grid.y;
#if defined(ARMON_ALT_KERNELS) && !defined(ARMON_ONLY_HEADLINE)
    if (k1)
        hipLaunchKernelGGL((k_sweep_x_dpp<PIPE::SCHEME, PIPE::LIM, PIPE::PROJ, PIPE::EOS, PIPE::kExact, 1, TRACK, false>),
                           grid, dim3(64, kXSRows), 0, ctx->stream, b, niter);
    else if (niter > 1)
        hipLaunchKernelGGL((k_sweep_x_dpp<PIPE::SCHEME, PIPE::LIM, PIPE::PROJ, PIPE::EOS, PIPE::kExact, 2, TRACK, false>),
                           grid, dim3(64, kXSRows), 0, ctx->stream, b, niter);
    else
#endif
#ifdef ARMON_XS_MULTI
        hipLaunchKernelGGL((k_sweep_x_dpp<PIPE::SCHEME, PIPE::LIM, PIPE::PROJ, PIPE::EOS, PIPE::kExact, 2, TRACK, false>),
                           grid, dim3(64, kXSRows), 0, ctx->stream, b, niter);
#else
        hipLaunchKernelGGL((k_sweep_x_dpp<PIPE::SCHEME, PIPE::LIM, PIPE::PROJ, PIPE::EOS, PIPE::kExact, 2, TRACK, true>),
                           grid, dim3(64, kXSRows), 0, ctx->stream, b, niter);
#endif
    return check_launch("sweep_x_dpp");
}

// Rows per run of the Y march. A run re-reads 2·LAG halo rows (and recomputes them), so long runs are cheaper per
// cell, but the launch must still fill the device evenly: the chip holds n_cu·ARMON_Y_WAVES·4 waves of this kernel
// at once and a launch of 4.06 such rounds takes nearly 5. Choose the number of runs per column that minimises
// rounds × (rows + halo), the rounds counted half-way between the exact ratio and its ceiling (waves drift apart,
// so a partial last round costs less than a whole one), among the launches of at least two rounds (a single round
// of long-lived workgroups exposes the whole ramp-up and tail: 4096², 137 rows in one round is 5 % slower than 32
// rows in 4.25). Measured at 16384² (tools/y_ab_r02.sh, one process): 128 rows 3.17 ms, 256: 3.12, 421: 3.09,
// 529: 3.08, 713: 3.09, 1093: 3.12, 2341 (one round, 11 % of the slots empty): 3.19.
int y_run_length(int n_cu, int64_t nx, int64_t ny, int lag, int cols_per_lane, int block)
{
    const double slots = (double)n_cu * ARMON_Y_WAVES * 4 / (block / 64);           // workgroups resident at once
    const int64_t cols = (nx + 16 + (int64_t)block * cols_per_lane - 1) / ((int64_t)block * cols_per_lane);
    int best = (int)(ny < 32 ? ny : 32);
    double best_cost = 1e300;
    for (int64_t nruns = 1; nruns <= ny; nruns++) {
        const int64_t seg = (ny + nruns - 1) / nruns;
        if (seg < 32) break;
        if ((ny + seg - 1) / seg != nruns) continue;                              // same launch as a smaller nruns
        const double rounds = (double)(cols * nruns) / slots;
        // One round that (nearly) fills every slot is the other good launch: every workgroup is resident from the start,
        // the runs are as long as they can be. Small tiles need it — 4096 x 8192 (the tile of 16384² on 8 GPUs): 273 rows in
        // 0.996 rounds 0.388 ms, 137 rows in 1.99 rounds 0.394, 92 rows in 2.99 rounds (the choice until round 3) 0.401,
        // 512 rows in 0.53 rounds 0.590 (profiles/r03_ab_yseg_4096x8192.txt) — while at 16384² the only single round
        // leaves 11 % of the slots empty and stays excluded (3.19 ms against 3.08).
        const bool full_single_round = rounds <= 1. && rounds >= 0.93;
        if (rounds < 2. && !full_single_round) continue;
        const double cost = (full_single_round ? 1. : 0.5 * (rounds + std::ceil(rounds))) * (double)(seg + 2 * lag);
        if (cost < best_cost) {
            best_cost = cost;
            best = (int)seg;
        }
    }
    return best;
}

// Upper bound of the number of workgroups any form launches for this block (sizes the partials buffer).
int64_t max_blocks(const sweep_args& a)
{
    const int64_t by = (a.nx + 16 + kYBlock - 1) / kYBlock * ((a.ny + a.seg - 1) / a.seg);
#ifdef ARMON_ALT_KERNELS
    const int64_t bx_lds = (a.nx + a.seg - 1) / a.seg * ((a.ny + kXRows - 1) / kXRows);
#else
    const int64_t bx_lds = 0;
#endif
    // per wave; niter >= 1, K = 1, LAG = 4; + kXSRows: a row's strips are rounded up to whole workgroups (x_wg_along_x)
    const int64_t bx_dpp = ((a.nx + 8) / 56 + 1 + kXSRows) * ((a.ny + kXSRows - 1) / kXSRows) * kXSRows;
    int64_t m = by > bx_lds ? by : bx_lds;
    m = m > bx_dpp ? m : bx_dpp;
    return m + kFoldBlocks;                                   // + the first-level results of fold_dt_launch
}

template <class PIPE>
int dispatch_track(armon_ctx* ctx, const sweep_args& a, int axis, bool track, int64_t* n_blocks)
{
    if (track) return launch<PIPE, true>(ctx, a, axis, n_blocks);
    return launch<PIPE, false>(ctx, a, axis, n_blocks);
}

template <int SCHEME, int LIM, int PROJ, int EOS>
int dispatch_exact(armon_ctx* ctx, const sweep_args& a, int axis, bool exact, bool track, int64_t* nb)
{
    if (exact) return dispatch_track<fused::Pipe<SCHEME, LIM, PROJ, EOS, real>>(ctx, a, axis, track, nb);
    return dispatch_track<fused::PipeFast<SCHEME, LIM, PROJ, EOS, real>>(ctx, a, axis, track, nb);
}

template <int SCHEME, int LIM, int PROJ>
int dispatch_eos(armon_ctx* ctx, const sweep_args& a, int axis, int eos, bool exact, bool track, int64_t* nb)
{
    if (eos == ARMON_EOS_BIZARRIUM)
        return dispatch_exact<SCHEME, LIM, PROJ, ARMON_EOS_BIZARRIUM>(ctx, a, axis, exact, track, nb);
    return dispatch_exact<SCHEME, LIM, PROJ, ARMON_EOS_PERFECT_GAS>(ctx, a, axis, exact, track, nb);
}

template <int SCHEME, int LIM>
int dispatch_proj(armon_ctx* ctx, const sweep_args& a, int axis, int eos, int proj, bool exact, bool track, int64_t* nb)
{
    if (proj == ARMON_PROJECTION_EULER_2ND)
        return dispatch_eos<SCHEME, LIM, ARMON_PROJECTION_EULER_2ND>(ctx, a, axis, eos, exact, track, nb);
    return dispatch_eos<SCHEME, LIM, ARMON_PROJECTION_EULER>(ctx, a, axis, eos, exact, track, nb);
}

}  // namespace

extern "C" int ARMON_SWEEP_FN(armon_ctx* ctx, const ARMON_SWEEP_DESC* d)
{
    ARMON_REQUIRE(ctx && d, "NULL argument");
    ARMON_REQUIRE(d->axis == ARMON_AXIS_X || d->axis == ARMON_AXIS_Y, "invalid axis %d", d->axis);
    ARMON_REQUIRE(d->scheme == ARMON_SCHEME_GODUNOV || d->scheme == ARMON_SCHEME_GAD, "unknown scheme %d", d->scheme);
    ARMON_REQUIRE(d->projection == ARMON_PROJECTION_EULER || d->projection == ARMON_PROJECTION_EULER_2ND,
                  "unknown projection %d", d->projection);
    ARMON_REQUIRE(d->eos == ARMON_EOS_PERFECT_GAS || d->eos == ARMON_EOS_BIZARRIUM, "unknown EOS %d", d->eos);
    ARMON_REQUIRE(d->scheme != ARMON_SCHEME_GAD || (d->limiter >= ARMON_LIMITER_NONE && d->limiter <= ARMON_LIMITER_SUPERBEE),
                  "unknown limiter tag %d", d->limiter);
    ARMON_REQUIRE(d->nx > 0 && d->ny > 0, "empty block %lld x %lld", (long long)d->nx, (long long)d->ny);
    ARMON_REQUIRE(d->ny < (1ll << 30) && d->nx < (1ll << 30), "block too large (%lld x %lld)", (long long)d->nx, (long long)d->ny);
    const int lag = (d->scheme == ARMON_SCHEME_GAD ? 1 : 0) + (d->projection == ARMON_PROJECTION_EULER_2ND ? 1 : 0) + 2;
    ARMON_REQUIRE(d->nghost >= lag, "nghost = %d but this scheme/projection reads %d cells past the block", d->nghost, lag);
    const int64_t n_axis = d->axis == ARMON_AXIS_X ? d->nx : d->ny;
    ARMON_REQUIRE(!(d->bc_low || d->bc_high) || n_axis >= lag,
                  "mirror boundary needs at least %d cells along the sweep axis", lag);
    ARMON_REQUIRE(d->rho_in && d->u_in && d->v_in && d->E_in && d->rho_out && d->u_out && d->v_out && d->E_out,
                  "NULL state array");
    ARMON_REQUIRE(d->rho_in != d->rho_out && d->u_in != d->u_out && d->v_in != d->v_out && d->E_in != d->E_out,
                  "in and out arrays must not alias (ping-pong)");
#ifdef ARMON_ALT_KERNELS
    ARMON_REQUIRE(d->x_kernel == 0 || d->x_kernel == 2 || d->x_kernel == 3, "unknown x_kernel form %d", d->x_kernel);
#else
    ARMON_REQUIRE(d->x_kernel == 0, "x_kernel form %d is a measured alternative: only libarmon_hip_alt.so (-DARMON_ALT_KERNELS) carries it",
                  d->x_kernel);
#endif
    // only the kernels the solver runs read the device-resident time step (sweep_begin): the LDS X march of the A/B build
    // would take `dt` (then a factor) for the step itself, silently
    ARMON_REQUIRE(!d->dt_state || d->x_kernel == 0 || d->x_kernel == 3,
                  "dt_state is not honoured by x_kernel form %d (only the DPP X sweeps and the Y march read it)", d->x_kernel);
    const bool exact = d->exact != 0;
    const bool track = d->dt_cfl_out != nullptr;
    ARMON_REQUIRE(!track || (d->cfl_dx > 0 && d->cfl_dy > 0), "dt_cfl_out needs cfl_dx, cfl_dy > 0");

    sweep_args a;
    a.nx = d->nx;
    a.ny = d->ny;
    a.row_len = d->nx + 2 * (int64_t)d->nghost;
    a.g = d->nghost;
    a.bc_low = d->bc_low;
    a.bc_high = d->bc_high;
    a.emit = (d->p_out ? 1 : 0) | (d->c_out ? 2 : 0);
    a.dt = (real)d->dt;
    a.dx = (real)d->dx;
    a.gamma = (real)d->gamma;
    a.inv_dx = real(1) / a.dx;        // IEEE quotients: the bits the kernels' own divisions gave until round 4
    a.dt_dx = a.dt / a.dx;
    const bool X = d->axis == ARMON_AXIS_X;
    a.fa_low = (real)(X ? d->u_factor_low : d->v_factor_low);
    a.ft_low = (real)(X ? d->v_factor_low : d->u_factor_low);
    a.fa_high = (real)(X ? d->u_factor_high : d->v_factor_high);
    a.ft_high = (real)(X ? d->v_factor_high : d->u_factor_high);
    a.rho_in = d->rho_in;
    a.ua_in = X ? d->u_in : d->v_in;
    a.ut_in = X ? d->v_in : d->u_in;
    a.E_in = d->E_in;
    a.rho_out = d->rho_out;
    a.ua_out = X ? d->u_out : d->v_out;
    a.ut_out = X ? d->v_out : d->u_out;
    a.E_out = d->E_out;
    a.p_out = d->p_out;
    a.c_out = d->c_out;
    a.st = d->dt_state;
    // the Y march stores through LDS when its rows do not all start on sectors (ARMON_Y_SX: 1 always, 2 never). Tuned
    // arithmetic only: the exact flavour is bound by its arithmetic (§4.2) and is not instantiated with the exchange.
    const bool align = ctx->tune_align != 0;
    a.y_sx = 0;
    if (!X && align && ctx->tune_y_sx != 2 && !d->exact) {
        const uintptr_t outs = (uintptr_t)d->rho_out | (uintptr_t)d->u_out | (uintptr_t)d->v_out | (uintptr_t)d->E_out;
        a.y_sx = outs % 64 == 0 && ((a.row_len * (int64_t)sizeof(real)) % 64 != 0 || ctx->tune_y_sx == 1);
    }
    if (X) {
        a.seg = 512;
    } else if (ctx->tune_y_seg > 0) {
        a.seg = ctx->tune_y_seg;
    } else {
        // the tuned fp32 march holds two columns per lane (k_sweep_y2, same condition as in launch()): half the workgroups per row
        const int cols_per_lane = (std::is_same<real, float>::value && !d->exact && d->nx % 2 == 0 && d->nghost % 2 == 0 &&
                                   d->nx >= 2 && !ctx->tune_y_cols1) ? 2 : 1;
        const int y_block = a.y_sx ? kYSxBlock : kYBlock;
        if (ctx->seg_nx != d->nx || ctx->seg_ny != n_axis || ctx->seg_lag != lag || ctx->seg_cols != cols_per_lane * y_block) {
            ctx->seg_value = y_run_length(ctx->n_cu, d->nx, n_axis, lag, cols_per_lane, y_block);
            ctx->seg_cols = cols_per_lane * y_block;
            ctx->seg_nx = d->nx;
            ctx->seg_ny = n_axis;
            ctx->seg_lag = lag;
        }
        a.seg = ctx->seg_value;
    }
    // the Y march addresses a run of rows with 32-bit byte offsets from the run's first row
    ARMON_REQUIRE(X || a.row_len * (int64_t)sizeof(real) * (a.seg + 2 * lag + 16) < (1ll << 32),
                  "block too wide for 32-bit row offsets (%lld cells per row, runs of %d rows)", (long long)d->nx, a.seg);
    a.x_kernel = d->x_kernel;
    a.o_lo = 0;
    a.o_hi = n_axis;
    if (d->out_hi != 0) {
        ARMON_REQUIRE(d->out_lo >= 0 && d->out_lo < d->out_hi && d->out_hi <= n_axis,
                      "invalid partial sweep [%lld, %lld) of %lld cells", (long long)d->out_lo, (long long)d->out_hi, (long long)n_axis);
        a.o_lo = d->out_lo;
        a.o_hi = d->out_hi;
    }
    a.xshift = (X || !align) ? 0 : d->nghost % 16;
    a.x_first = X ? (align ? a.o_lo - (a.o_lo + d->nghost) % 8 : a.o_lo) : 0;
    // XCD-aware placement of the X sweep's workgroups (sweep_x_dpp_body): neighbouring strips of a row — they share a 128-B
    // line, a strip's loads start 32 B before its sector-aligned stores — then follow each other on ONE XCD's L2 instead of
    // being fetched from the fabric by two. With one strip per wave that is the whole over-fetch of the sweep: 18.90 ->
    // 17.3 GB per launch by counters at 16384² (1.10x -> 1.01x the algorithmic bytes), time equal to 1.5 % better
    // (profiles/r04_ab_x_xcd.txt). fp64 only: fp32 shares its lines inside a workgroup instead (x_wg_along_x below).
    // The order assumes what the part does in its default mode: 8 XCDs of 32 CUs, workgroups dealt round-robin. A device
    // that shows another CU count (a partitioned MI355X: CPX / DPX modes) has fewer XCDs per agent, the map would only
    // scramble rows there: it is switched off. The workgroup shape is decided first (a block with more than 65535 rows cannot
    // take the along-x shape and falls back to one strip of 4 rows, which the remap then serves); the explicit knob wins.
    const bool want_along_x = ctx->tune_x_rows == 2 || (ctx->tune_x_rows == 0 && sizeof(real) == 4);
    const bool along_x = X && want_along_x && d->ny <= 65535 && ctx->tune_x_xcd <= 0;
    const bool eight_xcds = ctx->n_cu == 256;
    a.xcd_remap = (ctx->tune_x_xcd < 0 ? sizeof(real) == 8 : ctx->tune_x_xcd != 0) && !along_x && eight_xcds;
    // origins row by row when one origin cannot align every row (the one-strip-per-wave form only; the A/B forms keep one)
    a.x_row_align = 0;
#ifndef ARMON_XS_MULTI
    {
        const uintptr_t bases = (uintptr_t)d->rho_in | (uintptr_t)d->u_in | (uintptr_t)d->v_in | (uintptr_t)d->E_in |
                                (uintptr_t)d->rho_out | (uintptr_t)d->u_out | (uintptr_t)d->v_out | (uintptr_t)d->E_out |
                                (uintptr_t)d->p_out | (uintptr_t)d->c_out;
        a.x_row_align = X && align && d->x_kernel == 0 && ctx->tune_xs_niter <= 1 && a.row_len % (64 / (int64_t)sizeof(real)) != 0 &&
                        bases % 64 == 0;
    }
#endif
    // workgroup shape of the X sweep (profiles/r03_ab_x_workgroup_shape.txt): 4 consecutive strips of one row pay for fp32
    // (1.54 -> 1.43 ms at 16384²: a 512-B strip shares a quarter of its 128-B lines with its neighbours) and not for fp64
    // (equal at 16384² and 4096 x 8192, +3 % at 8192²), which keeps one strip of 4 rows. ARMON_X_ROWS: 1 / 2 force a shape.
    a.x_wg_along_x = along_x ? 1 : 0;                                                            // grid.y carries the rows
    a.partials = nullptr;
    if (track) {
        int rc = ensure_partials(ctx, (size_t)(2 * max_blocks(a)));
        if (rc != ARMON_OK) return rc;
        a.partials = reinterpret_cast<real*>(ctx->partials);
    }

    int64_t n_blocks = 0;
    int rc;
#ifdef ARMON_ONLY_HEADLINE   // variant builds for A/B timing (tools/build_variant.sh): one instantiation, seconds to compile
#ifndef ARMON_ONLY_EOS
#define ARMON_ONLY_EOS ARMON_EOS_PERFECT_GAS       // -DARMON_ONLY_EOS=ARMON_EOS_BIZARRIUM: the one instantiation is config 5's
#endif
    ARMON_REQUIRE(d->scheme == ARMON_SCHEME_GAD && d->limiter == ARMON_LIMITER_MINMOD && d->projection == ARMON_PROJECTION_EULER_2ND &&
                  d->eos == ARMON_ONLY_EOS && d->x_kernel == 0, "headline-only variant build");
    rc = dispatch_exact<ARMON_SCHEME_GAD, ARMON_LIMITER_MINMOD, ARMON_PROJECTION_EULER_2ND, ARMON_ONLY_EOS>(ctx, a, d->axis, exact, track, &n_blocks);
    if (rc != ARMON_OK || !track) return rc;
    return fold_dt_launch(ctx, a.partials, n_blocks, (real)d->cfl_dx, (real)d->cfl_dy, d->dt_cfl_out, d->dt_accumulate, d->dt_state);
#else
    if (d->scheme == ARMON_SCHEME_GODUNOV) {
        rc = dispatch_proj<ARMON_SCHEME_GODUNOV, ARMON_LIMITER_NONE>(ctx, a, d->axis, d->eos, d->projection, exact, track, &n_blocks);
    } else {
        switch (d->limiter) {
        case ARMON_LIMITER_MINMOD:
            rc = dispatch_proj<ARMON_SCHEME_GAD, ARMON_LIMITER_MINMOD>(ctx, a, d->axis, d->eos, d->projection, exact, track, &n_blocks);
            break;
        case ARMON_LIMITER_SUPERBEE:
            rc = dispatch_proj<ARMON_SCHEME_GAD, ARMON_LIMITER_SUPERBEE>(ctx, a, d->axis, d->eos, d->projection, exact, track, &n_blocks);
            break;
        default:
            rc = dispatch_proj<ARMON_SCHEME_GAD, ARMON_LIMITER_NONE>(ctx, a, d->axis, d->eos, d->projection, exact, track, &n_blocks);
        }
    }
    if (rc != ARMON_OK || !track) return rc;
    return fold_dt_launch(ctx, a.partials, n_blocks, (real)d->cfl_dx, (real)d->cfl_dy, d->dt_cfl_out, d->dt_accumulate, d->dt_state);
#endif
}

// ---- placement of the 8 streamed vectors -------------------------------------------------------
// The same sweeps run 10-20 % apart depending on where the 4 read and 4 written vectors sit in HBM relative to each
// other, and nothing visible from user space predicts it: time `tries` assignments of the 8 roles to the vectors of
// `pool` (the first one = pool[0..7] as given) with the caller's own X and Y sweeps, as in a cycle — X reads
// roles 0..3 and writes roles 4..7, Y reads 4..7 and writes 0..3 — and report the fastest.
extern "C" int ARMON_TUNE_FN(armon_ctx* ctx, const ARMON_SWEEP_DESC* x_desc, const ARMON_SWEEP_DESC* y_desc,
                             void* const* pool, int n_pool, size_t bytes, int tries, int* picks, double* times_ms)
{
    ARMON_REQUIRE(ctx && x_desc && y_desc && pool && picks, "NULL argument");
    ARMON_REQUIRE(n_pool >= 8 && tries >= 1 && bytes > 0, "need at least 8 vectors and 1 try (n_pool = %d, tries = %d)", n_pool, tries);
    for (int k = 0; k < n_pool; k++) ARMON_REQUIRE(pool[k], "pool[%d] is NULL", k);
    void* master[4] = {nullptr, nullptr, nullptr, nullptr};     // the state is parked here while roles move around
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = ARMON_OK;
    bool parked = false;              // true once the state is safe in `master`: a failure puts it back in pool[0..3]
    auto cleanup = [&]() {
        if (parked) {
            for (int k = 0; k < 4; k++)
                (void)hipMemcpyAsync(pool[k], master[k], bytes, hipMemcpyDeviceToDevice, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
        }
        for (void* m : master)
            if (m) (void)hipFree(m);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define TUNE_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail_hip(e_, #expr); } } while (0)
    for (int k = 0; k < 4; k++) {
        TUNE_TRY(hipMalloc(&master[k], bytes));
        TUNE_TRY(hipMemcpyAsync(master[k], pool[k], bytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
    parked = true;
    TUNE_TRY(hipEventCreate(&e0));
    TUNE_TRY(hipEventCreate(&e1));
    std::vector<int> idx(n_pool), best(8);
    double best_ms = 1e300;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    for (int t = 0; t < tries; t++) {
        for (int k = 0; k < n_pool; k++) idx[k] = k;
        if (t > 0)
            for (int k = 0; k < 8; k++) {                         // partial Fisher-Yates: 8 distinct vectors
                rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
                std::swap(idx[k], idx[k + (int)(rng % (uint64_t)(n_pool - k))]);
            }
        for (int k = 0; k < 4; k++)
            TUNE_TRY(hipMemcpyAsync(pool[idx[k]], master[k], bytes, hipMemcpyDeviceToDevice, ctx->stream));
        ARMON_SWEEP_DESC dx = *x_desc, dy = *y_desc;
        using ptr_t = decltype(dx.rho_out);
        auto P = [&](int role) { return static_cast<ptr_t>(pool[idx[role]]); };
        dx.rho_in = P(0); dx.u_in = P(1); dx.v_in = P(2); dx.E_in = P(3);
        dx.rho_out = P(4); dx.u_out = P(5); dx.v_out = P(6); dx.E_out = P(7);
        dy.rho_in = P(4); dy.u_in = P(5); dy.v_in = P(6); dy.E_in = P(7);
        dy.rho_out = P(0); dy.u_out = P(1); dy.v_out = P(2); dy.E_out = P(3);
        double ms_min = 1e300;
        for (int rep = 0; rep < 3; rep++) {
            TUNE_TRY(hipEventRecord(e0, ctx->stream));
            rc = ARMON_SWEEP_FN(ctx, &dx);
            if (rc == ARMON_OK) rc = ARMON_SWEEP_FN(ctx, &dy);
            if (rc != ARMON_OK) { cleanup(); return rc; }
            TUNE_TRY(hipEventRecord(e1, ctx->stream));
            TUNE_TRY(hipEventSynchronize(e1));
            float ms = 0.f;
            TUNE_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < ms_min) ms_min = ms;
        }
        if (times_ms) times_ms[t] = ms_min;
        if (ms_min < best_ms) {
            best_ms = ms_min;
            for (int k = 0; k < 8; k++) best[k] = idx[k];
        }
    }
    for (int k = 0; k < 4; k++)
        TUNE_TRY(hipMemcpyAsync(pool[best[k]], master[k], bytes, hipMemcpyDeviceToDevice, ctx->stream));
    TUNE_TRY(hipStreamSynchronize(ctx->stream));
#undef TUNE_TRY
    for (int k = 0; k < 8; k++) picks[k] = best[k];
    parked = false;                   // the state now lives in pool[picks[0..3]]
    cleanup();
    return ARMON_OK;
}

// Same choice for a pool that holds NO state worth keeping (a host calls it BEFORE init_test writes the initial
// condition): nothing is parked or restored, so the only transient memory is the caller's spare vectors. Each
// candidate's four input vectors are filled with a uniform state (the sweeps' instruction stream does not depend on the
// data) and timed like above; after 12 draws the search stops as soon as two of them lie within `tolerance` of the best
// one while a draw at least 7 % slower has been seen as well (the good placements form a plateau), after
// at most `tries` draws.
namespace {
__global__ void __launch_bounds__(256) k_fill_uniform(real* __restrict__ p, size_t n, real value)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = value;
}
}  // namespace

extern "C" int ARMON_CHOOSE_FN(armon_ctx* ctx, const ARMON_SWEEP_DESC* x_desc, const ARMON_SWEEP_DESC* y_desc,
                               void* const* pool, int n_pool, size_t bytes, int tries, double tolerance, int* picks,
                               double* times_ms, int* tries_done)
{
    ARMON_REQUIRE(ctx && x_desc && y_desc && pool && picks, "NULL argument");
    ARMON_REQUIRE(n_pool >= 8 && tries >= 1 && bytes >= sizeof(real), "need at least 8 vectors and 1 try (n_pool = %d, tries = %d)", n_pool, tries);
    for (int k = 0; k < n_pool; k++) ARMON_REQUIRE(pool[k], "pool[%d] is NULL", k);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define CHOOSE_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail_hip(e_, #expr); } } while (0)
    CHOOSE_TRY(hipEventCreate(&e0));
    CHOOSE_TRY(hipEventCreate(&e1));
    std::vector<int> idx(n_pool), best(8);
    std::vector<double> seen;
    double best_ms = 1e300;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    const size_t n = bytes / sizeof(real);
    const real uniform[4] = {real(1), real(0), real(0), real(2.5)};          // rho, u, v, E: Sod's left state
    int t = 0;
    for (; t < tries; t++) {
        for (int k = 0; k < n_pool; k++) idx[k] = k;
        if (t > 0)
            for (int k = 0; k < 8; k++) {                         // partial Fisher-Yates: 8 distinct vectors
                rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
                std::swap(idx[k], idx[k + (int)(rng % (uint64_t)(n_pool - k))]);
            }
        ARMON_SWEEP_DESC dx = *x_desc, dy = *y_desc;
        using ptr_t = decltype(dx.rho_out);
        auto P = [&](int role) { return static_cast<ptr_t>(pool[idx[role]]); };
        for (int k = 0; k < 4; k++)
            hipLaunchKernelGGL(k_fill_uniform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, P(k), n, uniform[k]);
        dx.rho_in = P(0); dx.u_in = P(1); dx.v_in = P(2); dx.E_in = P(3);
        dx.rho_out = P(4); dx.u_out = P(5); dx.v_out = P(6); dx.E_out = P(7);
        dy.rho_in = P(4); dy.u_in = P(5); dy.v_in = P(6); dy.E_in = P(7);
        dy.rho_out = P(0); dy.u_out = P(1); dy.v_out = P(2); dy.E_out = P(3);
        double ms_min = 1e300;
        for (int rep = 0; rep < 3; rep++) {
            CHOOSE_TRY(hipEventRecord(e0, ctx->stream));
            int rc = ARMON_SWEEP_FN(ctx, &dx);
            if (rc == ARMON_OK) rc = ARMON_SWEEP_FN(ctx, &dy);
            if (rc != ARMON_OK) { cleanup(); return rc; }
            CHOOSE_TRY(hipEventRecord(e1, ctx->stream));
            CHOOSE_TRY(hipEventSynchronize(e1));
            float ms = 0.f;
            CHOOSE_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < ms_min) ms_min = ms;
        }
        if (times_ms) times_ms[t] = ms_min;
        seen.push_back(ms_min);
        if (ms_min < best_ms) {
            best_ms = ms_min;
            for (int k = 0; k < 8; k++) best[k] = idx[k];
        }
        // stop once the plateau of good placements has been hit twice — two draws within `tolerance` of the best —
        // AND a draw at least 7 % slower has been seen too, after 12 draws at least: the X+Y times come in three
        // levels (16384²: ≈5.65 / 6.17 / 6.45 ms) and two equal draws of the middle level must not end the search
        int near = 0;
        double worst = 0;
        for (double v : seen) {
            near += v <= best_ms * (1. + tolerance);
            worst = v > worst ? v : worst;
        }
        if (tolerance > 0 && t >= 11 && near >= 2 && best_ms <= 0.93 * worst) { t++; break; }
    }
#undef CHOOSE_TRY
    for (int k = 0; k < 8; k++) picks[k] = best[k];
    if (tries_done) *tries_done = t;
    cleanup();
    return ARMON_OK;
}

// ---- whole cycle (X sweep then Y sweep, Sequential splitting) in one launch: k_cycle_xy --------------------------------
#if defined(ARMON_CYCLE_FN) && !defined(ARMON_ALT_KERNELS)
extern "C" int ARMON_CYCLE_FN(armon_ctx* ctx, const ARMON_SWEEP_DESC* x, const ARMON_SWEEP_DESC* y)
{
    ARMON_REQUIRE(ctx && x && y, "NULL argument");
    ARMON_REQUIRE(false, "the whole-cycle kernel is a measured alternative (slower than the two sweeps, DESIGN.md section 4.2): "
                         "only libarmon_hip_alt.so (-DARMON_ALT_KERNELS) carries it");
    return ARMON_ERR_INVALID_ARG;
}
#endif
#if defined(ARMON_CYCLE_FN) && defined(ARMON_ALT_KERNELS)
#include "fused_sweep_alt_cycle.hpp"
#endif
