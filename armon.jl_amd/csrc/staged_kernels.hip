// staged_kernels.hip — one HIP kernel per reference @generic_kernel (SURVEY §8a rows a1–a10, a13, a16).
// These are the drop-in replacements of the generated "main" functions
// (ref src/generic_kernel.jl:825-921): same arguments, same iteration range, same arrays.
//
// Launch shape: a row of the range is x-contiguous, so blockIdx.x/threadIdx.x walk a row (coalesced
// 512 B per wave per array) and blockIdx.y walks rows — no integer division per cell, unlike the
// reference's 1-D ndrange + divrem (ref src/generic_kernel.jl:784-791).
#include "common.hpp"
#include "physics.hpp"
#include "sweep_spatial.hpp"   // wavefront shifts (from_prev_lane / from_next_lane)

using namespace armon;

namespace {

__device__ __forceinline__ int64_t row_base(const armon_range& r, int64_t j)
{
    return r.col_start + j * r.col_step + r.row_start;
}

// Outputs that the next kernel re-reads only after 2+ GB of other traffic (fluxes, advected quantities): streaming
// stores, so that they do not displace the neighbour rows the stencils DO re-read from L2.
template <typename T> __device__ __forceinline__ void st_stream(T* p, T v)
{
#ifdef ARMON_ST_PLAIN            // A/B builds: ordinary stores
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}

// Threads are laid along a row of the range, shifted left so that every wave's row segment starts on a 64-B sector
// of the array (ranges start g or g-w cells into a row: without the shift each 512-B segment straddles one more sector;
// the fused sweeps gained 9-11 % from the same alignment). Grids are sized for row_len + 15 threads.
template <typename T> __device__ __forceinline__ int64_t row_thread(const armon_range& r)
{
    constexpr int64_t per_sector = 64 / sizeof(T);
    return (int64_t)blockIdx.x * blockDim.x + threadIdx.x - ((r.col_start + r.row_start) & (per_sector - 1));
}

#define ARMON_FOR_RANGE(r, i)                                                         \
    const int64_t k_ = row_thread<T>(r);                                               \
    if (k_ >= 0 && k_ < (r).row_len)                                                   \
        for (int64_t j_ = blockIdx.y, i = row_base((r), j_) + k_; j_ < (r).col_len;    \
             j_ += gridDim.y, i = row_base((r), j_) + k_)

// ---- a1: perfect_gas_EOS! (ref src/kernels.jl:4-13) ------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_perfect_gas_EOS(armon_range r, T gamma, const T* __restrict__ rho,
                  const T* __restrict__ E, const T* __restrict__ u,
                  const T* __restrict__ v, T* __restrict__ p, T* __restrict__ c,
                  T* __restrict__ g)
{
    ARMON_FOR_RANGE(r, i) {
        T pi, ci;
        phys::perfect_gas(gamma, rho[i], E[i], u[i], v[i], pi, ci);
        st_stream(p + i, pi);
        st_stream(c + i, ci);
        st_stream(g + i, (T(1.) + gamma) / 2);
    }
}

// ---- a2: bizarrium_EOS! (ref src/kernels.jl:16-55) -------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_bizarrium_EOS(armon_range r, const T* __restrict__ rho, const T* __restrict__ u,
                const T* __restrict__ v, const T* __restrict__ E, T* __restrict__ p,
                T* __restrict__ c, T* __restrict__ g)
{
    ARMON_FOR_RANGE(r, i) {
        T pi, ci, gi;
        phys::bizarrium<true>(rho[i], E[i], u[i], v[i], pi, ci, gi);
        st_stream(p + i, pi);
        st_stream(c + i, ci);
        st_stream(g + i, gi);
    }
}

// ---- a4: acoustic! (ref src/riemann_schemes.jl:33-43) ----------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_acoustic(armon_range r, int64_t s, T* __restrict__ us, T* __restrict__ ps,
           const T* __restrict__ rho, const T* __restrict__ u,
           const T* __restrict__ p, const T* __restrict__ c)
{
    ARMON_FOR_RANGE(r, i) {
        T a, b;
        phys::godunov(rho[i], rho[i - s], c[i], c[i - s], u[i], u[i - s], p[i], p[i - s], a, b);
        st_stream(us + i, a);
        st_stream(ps + i, b);
    }
}

// ---- a5: acoustic_GAD! (ref src/riemann_schemes.jl:55-104) -----------------------------------------
template <int LIM, typename T>
__global__ void __launch_bounds__(kBlock)
k_acoustic_GAD(armon_range r, int64_t s, T dt, T dx, T* __restrict__ us,
               T* __restrict__ ps, const T* __restrict__ rho, const T* __restrict__ u,
               const T* __restrict__ p, const T* __restrict__ c)
{
    ARMON_FOR_RANGE(r, i) {
        const T rho_mm = rho[i - 2 * s], c_mm = c[i - 2 * s], u_mm = u[i - 2 * s], p_mm = p[i - 2 * s];
        const T rho_m = rho[i - s], c_m = c[i - s], u_m = u[i - s], p_m = p[i - s];
        const T rho_i = rho[i], c_i = c[i], u_i = u[i], p_i = p[i];
        const T rho_p = rho[i + s], c_p = c[i + s], u_p = u[i + s], p_p = p[i + s];
        T us_m, ps_m, us_0, ps_0, us_p, ps_p, a, b;
        phys::godunov(rho_m, rho_mm, c_m, c_mm, u_m, u_mm, p_m, p_mm, us_m, ps_m);
        phys::godunov(rho_i, rho_m, c_i, c_m, u_i, u_m, p_i, p_m, us_0, ps_0);
        phys::godunov(rho_p, rho_i, c_p, c_i, u_p, u_i, p_p, p_i, us_p, ps_p);
        phys::gad_flux<LIM>(dt, dx, rho_m, c_m, u_m, p_m, rho_i, c_i, u_i, p_i,
                            us_m, ps_m, us_0, ps_0, us_p, ps_p, a, b);
        us[i] = a;
        ps[i] = b;
    }
}

// The reference's kernel above solves every interface three times (once as "its own", twice as a neighbour's:
// ref src/riemann_schemes.jl:63-80). The two forms below compute the same values from the same operands — bit for bit
// the same results — but solve each interface ONCE and hand the solutions to the neighbours:
//  * sweep along x (s == 1): lanes along x, a wave covers 64 consecutive cells of which the outer ones are halo
//    lanes (they only contribute their interface solution); neighbours' cells and solutions come from DPP wavefront
//    shifts. The stencil needs one halo lane on either side (62 fluxes per 64 solves); a wave takes 4 + 4 (fp64) and
//    produces 56 fluxes = 7 whole 64-B sectors, placed on the sectors of the array: a wave that begins and ends its
//    stores inside a sector pays for it (2.34-2.39 -> 2.19 ms at 16384², profiles/r03_gad_forms.txt; the misplaced
//    LOADS cost nothing, profiles/r03_row_pitch_repairs.txt).
//  * sweep along y (s == row pitch of the range): lane ↔ column, each thread walks kGadRows rows with a rolling
//    window of the last two cells and three solutions. kGadRows fluxes per kGadRows + 2 solves.
// No cell outside the reference's own stencil [i - 2s, i + s] of the range is read.
constexpr int kGadValid = 62;
#ifndef ARMON_GAD_ALIGNED
#define ARMON_GAD_ALIGNED 1      // x forms (GAD, second-order advection): 1 = waves that store whole sectors (56 of 64 lanes in fp64), 0 = 62 / 60
#endif
#ifndef ARMON_GAD_ROWS
// rows per thread of the y form (tuning macro). SHORT walks: 8 rows cost 3 extra row loads per 8 — served by the L2, the rows
// were just read by the workgroup below — and put 8 x as many workgroups of a column in flight, so that the device streams a
// narrow window of rows instead of 30 distant fronts: 2.58-2.71 -> 2.25-2.33 ms at 16384² (4, 6, 12, 16, 32 rows: 2.41, 2.34,
// 2.28, 2.44, 2.66; a chunk-major launch with XCD-stable column blocks: no further gain; profiles/r05_staged_y_chunks.txt).
// Round 3 had only tried LONGER walks (64 -> 256 rows: equal).
#define ARMON_GAD_ROWS 8
#endif
constexpr int kGadRows = ARMON_GAD_ROWS;
#ifndef ARMON_GAD_AHEAD
#define ARMON_GAD_AHEAD 1        // rows in flight ahead of the one being computed in the y form (tuning macro)
#endif
constexpr int kGadAhead = ARMON_GAD_AHEAD;

template <int LIM, typename T>
__global__ void __launch_bounds__(kBlock)
k_acoustic_GAD_x(armon_range r, T dt, T dx, T* __restrict__ us, T* __restrict__ ps, const T* __restrict__ rho,
                 const T* __restrict__ u, const T* __restrict__ p, const T* __restrict__ c)
{
    using fused::from_next_lane;
    using fused::from_prev_lane;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#if ARMON_GAD_ALIGNED
    // a wave stores whole 64-B sectors: 64 - S fluxes (S = cells per sector), S/2 halo lanes on either side, the first wave
    // of a row starting on the sector at or below the row's first flux
    constexpr int kS = 64 / (int)sizeof(T), kLeft = kS / 2, kValid = 64 - kS;
    const int64_t w0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * kValid - ((r.col_start + r.row_start) & (kS - 1));
    const int64_t k = w0 + lane - kLeft;
    const bool has_cell = k >= -2 && k <= r.row_len;           // the reference's stencil of the range
    const bool stores = lane >= kLeft && lane < kLeft + kValid && k >= 0 && k < r.row_len;
    if (w0 >= r.row_len) return;                               // whole wave past the row
#else
    constexpr int kLeft = 1;
    const int64_t k = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * kGadValid + lane - 1;   // lanes 0 and 63: halo
    const bool has_cell = k >= -1 && k <= r.row_len;           // the halo lanes may sit one cell outside the row
    const bool stores = lane >= 1 && lane <= kGadValid && k < r.row_len;
    if (((int64_t)blockIdx.x * (kBlock / 64) + wave) * kGadValid >= r.row_len) return;      // whole wave past the row
#endif
    for (int64_t j = blockIdx.y; j < r.col_len; j += gridDim.y) {
        const int64_t i = row_base(r, j) + k;
        T rho_i = 1, c_i = 1, u_i = 0, p_i = 1;
        if (has_cell) { rho_i = rho[i]; c_i = c[i]; u_i = u[i]; p_i = p[i]; }
        T rho_m = from_prev_lane(rho_i), c_m = from_prev_lane(c_i), u_m = from_prev_lane(u_i), p_m = from_prev_lane(p_i);
        if (kLeft == 1 && lane == 0) {                          // no lane to the left: its cell i - 1 from memory
            rho_m = 1; c_m = 1; u_m = 0; p_m = 1;
            if (has_cell) { rho_m = rho[i - 1]; c_m = c[i - 1]; u_m = u[i - 1]; p_m = p[i - 1]; }
        }
        T us_0, ps_0;
#ifdef ARMON_GAD_NOCOMPUTE
        us_0 = rho_m; ps_0 = c_m + u_m + p_m;
#else
        phys::godunov(rho_i, rho_m, c_i, c_m, u_i, u_m, p_i, p_m, us_0, ps_0);        // interface i | i-1, once
#endif
        const T us_m = from_prev_lane(us_0), ps_m = from_prev_lane(ps_0);
        const T us_p = from_next_lane(us_0), ps_p = from_next_lane(ps_0);
        T a, b;
#ifdef ARMON_GAD_NOCOMPUTE       // probe builds: the kernel's memory pattern alone
        a = rho_i + c_i + us_m * 0; b = u_i + p_i + us_p * 0;
#else
        phys::gad_flux<LIM>(dt, dx, rho_m, c_m, u_m, p_m, rho_i, c_i, u_i, p_i, us_m, ps_m, us_0, ps_0, us_p, ps_p, a, b);
#endif
        if (stores) {
            st_stream(us + i, a);
            st_stream(ps + i, b);
        }
    }
}

// Sweep along x, two cells per lane (fp64, even row pitch, 16-B aligned arrays): a wave covers a strip of 128 consecutive
// cells with 16-B accesses and produces the 120 fluxes in its middle (4 halo cells on either side: the stencil needs 2 and 1,
// but 120 cells are 15 whole 64-B sectors, so with the strip origins chosen below every strip's stores start and end on a
// sector of the array, and its loads start 32 B before one — the layout of the fused X sweep). A workgroup
// takes one strip of 4 consecutive rows. Same operands into the same functions as the forms above: the same bits.
// Measured 2-3 % SLOWER than one cell per lane (2.42-2.44 against 2.37 ms at 16384², profiles/r03_gad_forms.txt): the kernel
// runs at 95 % of its own no-arithmetic form either way, so the form is only compiled into A/B builds (-DARMON_GAD_X2=1).
#ifndef ARMON_GAD_X2
#define ARMON_GAD_X2 0
#endif
#if ARMON_GAD_X2
constexpr int kGadStride2 = 120, kGadHalo2 = 4;

template <int LIM>
__global__ void __launch_bounds__(kBlock)
k_acoustic_GAD_x2(armon_range r, double dt, double dx, double* __restrict__ us, double* __restrict__ ps,
                  const double* __restrict__ rho, const double* __restrict__ u, const double* __restrict__ p,
                  const double* __restrict__ c)
{
    using fused::from_next_lane;
    using fused::from_prev_lane;
    typedef double V2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // first flux of this strip, relative to the first cell of a row of the range (strip 0 starts on the sector at or below it)
    const int64_t w0 = (int64_t)blockIdx.x * kGadStride2 - ((r.col_start + r.row_start) & 7);
    const int64_t k0 = w0 - kGadHalo2 + 2 * lane;              // this lane's cells: k0, k0 + 1
    const int64_t lo = w0 > 0 ? w0 : 0;
    const int64_t hi = (w0 + kGadStride2 < r.row_len) ? w0 + kGadStride2 : r.row_len;
    // reads stay inside the reference's stencil of the range: cells -2 .. row_len of a row
    const bool in_a = k0 >= -2 && k0 <= r.row_len, in_b = k0 + 1 >= -2 && k0 + 1 <= r.row_len;
    const bool st_a = k0 >= lo && k0 < hi, st_b = k0 + 1 >= lo && k0 + 1 < hi;
    for (int64_t j = (int64_t)blockIdx.y * (kBlock / 64) + wave; j < r.col_len; j += (int64_t)gridDim.y * (kBlock / 64)) {
        const int64_t i = row_base(r, j) + k0;
        double rho_a = 1, c_a = 1, u_a = 0, p_a = 1, rho_b = 1, c_b = 1, u_b = 0, p_b = 1;
        if (in_a && in_b) {
            const V2 vr = *reinterpret_cast<const V2*>(rho + i), vc = *reinterpret_cast<const V2*>(c + i);
            const V2 vu = *reinterpret_cast<const V2*>(u + i), vp = *reinterpret_cast<const V2*>(p + i);
            rho_a = vr.x; rho_b = vr.y; c_a = vc.x; c_b = vc.y; u_a = vu.x; u_b = vu.y; p_a = vp.x; p_b = vp.y;
        } else {
            if (in_a) { rho_a = rho[i]; c_a = c[i]; u_a = u[i]; p_a = p[i]; }
            if (in_b) { rho_b = rho[i + 1]; c_b = c[i + 1]; u_b = u[i + 1]; p_b = p[i + 1]; }
        }
        // the cell left of a is the previous lane's b (lane 0 reads zeros: its fluxes are halo)
        const double rho_m = from_prev_lane(rho_b), c_m = from_prev_lane(c_b), u_m = from_prev_lane(u_b), p_m = from_prev_lane(p_b);
        double us_a, ps_a, us_b, ps_b;
        phys::godunov(rho_a, rho_m, c_a, c_m, u_a, u_m, p_a, p_m, us_a, ps_a);          // interface a | m
        phys::godunov(rho_b, rho_a, c_b, c_a, u_b, u_a, p_b, p_a, us_b, ps_b);          // interface b | a
        const double us_l = from_prev_lane(us_b), ps_l = from_prev_lane(ps_b);           // interface m | (cell before m)
        const double us_r = from_next_lane(us_a), ps_r = from_next_lane(ps_a);           // interface (cell after b) | b
        double fa_u, fa_p, fb_u, fb_p;
        phys::gad_flux<LIM>(dt, dx, rho_m, c_m, u_m, p_m, rho_a, c_a, u_a, p_a, us_l, ps_l, us_a, ps_a, us_b, ps_b, fa_u, fa_p);
        phys::gad_flux<LIM>(dt, dx, rho_a, c_a, u_a, p_a, rho_b, c_b, u_b, p_b, us_a, ps_a, us_b, ps_b, us_r, ps_r, fb_u, fb_p);
        if (st_a && st_b) {
            __builtin_nontemporal_store(V2{fa_u, fb_u}, reinterpret_cast<V2*>(us + i));
            __builtin_nontemporal_store(V2{fa_p, fb_p}, reinterpret_cast<V2*>(ps + i));
        } else {
            if (st_a) { st_stream(us + i, fa_u); st_stream(ps + i, fa_p); }
            if (st_b) { st_stream(us + i + 1, fb_u); st_stream(ps + i + 1, fb_p); }
        }
    }
}
#endif   // ARMON_GAD_X2

template <int LIM, typename T>
__global__ void __launch_bounds__(kBlock)
k_acoustic_GAD_y(armon_range r, int64_t s, T dt, T dx, T* __restrict__ us, T* __restrict__ ps, const T* __restrict__ rho,
                 const T* __restrict__ u, const T* __restrict__ p, const T* __restrict__ c)
{
    const int64_t k = row_thread<T>(r);
    if (k < 0 || k >= r.row_len) return;
    const int64_t j0 = (int64_t)blockIdx.y * kGadRows;
    const int64_t j1 = (j0 + kGadRows < r.col_len) ? j0 + kGadRows : r.col_len;
    int64_t i = row_base(r, j0) + k;
    // cells j0-2, j0-1, j0 and the solutions of interfaces j0-1 and j0
    T rho_mm = rho[i - 2 * s], c_mm = c[i - 2 * s], u_mm = u[i - 2 * s], p_mm = p[i - 2 * s];
    T rho_m = rho[i - s], c_m = c[i - s], u_m = u[i - s], p_m = p[i - s];
    T rho_i = rho[i], c_i = c[i], u_i = u[i], p_i = p[i];
    T us_m, ps_m, us_0, ps_0;
    phys::godunov(rho_m, rho_mm, c_m, c_mm, u_m, u_mm, p_m, p_mm, us_m, ps_m);
    phys::godunov(rho_i, rho_m, c_i, c_m, u_i, u_m, p_i, p_m, us_0, ps_0);
    // rows j + 1 .. j + kGadAhead are in flight while row j is computed: a row is requested kGadAhead steps before it is
    // consumed (one step of 7 waves on a SIMD is shorter than a loaded HBM latency). No row above j1 is read.
    T q[kGadAhead][4];
#pragma unroll
    for (int a = 0; a < kGadAhead; a++) {
        const int64_t in = i + ((j0 + 1 + a < j1) ? (int64_t)(a + 1) : (j1 - j0)) * s;
        q[a][0] = rho[in]; q[a][1] = c[in]; q[a][2] = u[in]; q[a][3] = p[in];
    }
    for (int64_t j = j0; j < j1; j++, i += s) {
        const int64_t in = i + ((j + 1 + kGadAhead < j1) ? (int64_t)(kGadAhead + 1) : (j1 - j)) * s;
        const T rho_n = rho[in], c_n = c[in], u_n = u[in], p_n = p[in];
        const T rho_p = q[0][0], c_p = q[0][1], u_p = q[0][2], p_p = q[0][3];          // row j + 1
        T us_p, ps_p, a, b;
#ifdef ARMON_GAD_NOCOMPUTE
        us_p = rho_p + c_p; ps_p = u_p + p_p; a = us_m + rho_i + c_i; b = ps_m + u_i + p_i;
#else
        phys::godunov(rho_p, rho_i, c_p, c_i, u_p, u_i, p_p, p_i, us_p, ps_p);         // interface j+1 | j, once
        phys::gad_flux<LIM>(dt, dx, rho_m, c_m, u_m, p_m, rho_i, c_i, u_i, p_i, us_m, ps_m, us_0, ps_0, us_p, ps_p, a, b);
#endif
        st_stream(us + i, a);
        st_stream(ps + i, b);
        rho_m = rho_i; c_m = c_i; u_m = u_i; p_m = p_i;
        rho_i = rho_p; c_i = c_p; u_i = u_p; p_i = p_p;
#pragma unroll
        for (int a2 = 0; a2 + 1 < kGadAhead; a2++) { q[a2][0] = q[a2 + 1][0]; q[a2][1] = q[a2 + 1][1]; q[a2][2] = q[a2 + 1][2]; q[a2][3] = q[a2 + 1][3]; }
        q[kGadAhead - 1][0] = rho_n; q[kGadAhead - 1][1] = c_n; q[kGadAhead - 1][2] = u_n; q[kGadAhead - 1][3] = p_n;
        us_m = us_0; ps_m = ps_0;
        us_0 = us_p; ps_0 = ps_p;
    }
}

// ---- a6: cell_update! (ref src/kernels.jl:58-68) ---------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_cell_update(armon_range r, int64_t s, T dx, T dt, const T* __restrict__ us,
              const T* __restrict__ ps, T* __restrict__ rho, T* __restrict__ ua,
              T* __restrict__ E)
{
    ARMON_FOR_RANGE(r, i) {
        T rho_i = rho[i], ua_i = ua[i], E_i = E[i];
        phys::cell_update(dx, dt, us[i], ps[i], us[i + s], ps[i + s], rho_i, ua_i, E_i);
        rho[i] = rho_i;
        ua[i] = ua_i;
        E[i] = E_i;
    }
}

// ---- a7: advection_first_order! (ref src/projection_schemes.jl:62-78) ------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_advection_first_order(armon_range r, int64_t s, T dt, const T* __restrict__ us,
                        const T* __restrict__ rho, const T* __restrict__ u,
                        const T* __restrict__ v, const T* __restrict__ E,
                        T* __restrict__ a_rho, T* __restrict__ a_urho,
                        T* __restrict__ a_vrho, T* __restrict__ a_Erho)
{
    ARMON_FOR_RANGE(r, is) {
        T disp = dt * us[is];
        int64_t d = (disp > 0) ? is - s : is;
        T rd = rho[d];
        a_rho[is] = disp * (rd);
        a_urho[is] = disp * (rd * u[d]);
        a_vrho[is] = disp * (rd * v[d]);
        a_Erho[is] = disp * (rd * E[d]);
    }
}

// ---- a8: advection_second_order! (ref src/projection_schemes.jl:92-124) ----------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_advection_second_order(armon_range r, int64_t s, T dx, T dt,
                         const T* __restrict__ us, const T* __restrict__ rho,
                         const T* __restrict__ u, const T* __restrict__ v,
                         const T* __restrict__ E, T* __restrict__ a_rho,
                         T* __restrict__ a_urho, T* __restrict__ a_vrho,
                         T* __restrict__ a_Erho)
{
    ARMON_FOR_RANGE(r, is) {
        T disp = dt * us[is];
        T Dxe;
        int64_t d;
        if (disp > 0) {
            Dxe = -(dx - dt * us[is - s]);
            d = is - s;
        } else {
            Dxe = dx + dt * us[is + s];
            d = is;
        }
        T Dxl_m = dx + dt * (us[d] - us[d - s]);
        T Dxl   = dx + dt * (us[d + s] - us[d]);
        T Dxl_p = dx + dt * (us[d + 2 * s] - us[d + s]);
        T r_m = xct::Den<T>(Dxl + Dxl_m).quo(2 * Dxl);
        T r_p = xct::Den<T>(Dxl + Dxl_p).quo(2 * Dxl);

        T rm = rho[d - s], r0 = rho[d], rp = rho[d + s];
        T sl_rho  = phys::slope_minmod(rm, r0, rp, r_m, r_p);
        T sl_urho = phys::slope_minmod(rm * u[d - s], r0 * u[d], rp * u[d + s], r_m, r_p);
        T sl_vrho = phys::slope_minmod(rm * v[d - s], r0 * v[d], rp * v[d + s], r_m, r_p);
        T sl_Erho = phys::slope_minmod(rm * E[d - s], r0 * E[d], rp * E[d + s], r_m, r_p);

        T length_factor = xct::Den<T>(2 * Dxl).quo(Dxe);
        a_rho[is]  = disp * (r0        - sl_rho  * length_factor);
        a_urho[is] = disp * (r0 * u[d] - sl_urho * length_factor);
        a_vrho[is] = disp * (r0 * v[d] - sl_vrho * length_factor);
        a_Erho[is] = disp * (r0 * E[d] - sl_Erho * length_factor);
    }
}

// The reference's kernel above gathers, per interface, three deformed lengths, the limited slopes and the products
// ρu, ρv, ρE of the DONOR cell and its two neighbours (17 loads, ref src/projection_schemes.jl:92-124). The deformed
// length Δx(c) = dx + dt·(uˢ[c+s] − uˢ[c]), the products and the slopes are cell-centred quantities: the two forms
// below evaluate them once per cell — the same expressions on the same operands, hence the same bits — and let the
// interface pick its donor's:
//  * sweep along x: lanes along x (a wave covers 64 cells, 60 interfaces: two halo lanes on each side), neighbours by
//    DPP wavefront shifts — 5 loads per lane;
//  * sweep along y: lane ↔ column, kAdvRows rows per thread with rolling windows — 5 loads per row.
// No cell outside the reference's own stencil (uˢ @ is-2s..is+2s; ρ,u,v,E @ is-2s..is+s) is read.
constexpr int kAdvValid = 60;
#ifndef ARMON_ADV_ROWS
#define ARMON_ADV_ROWS 8         // rows per thread of the y form (tuning macro): as ARMON_GAD_ROWS — 3.31-3.47 -> 3.24-3.31 ms at 16384²
#endif
constexpr int kAdvRows = ARMON_ADV_ROWS;

template <typename T> struct adv_cell { T rho, qu, qv, qE; };

template <typename T>
__device__ __forceinline__ adv_cell<T> adv_load(const T* rho, const T* u, const T* v, const T* E, int64_t i)
{
    const T r = rho[i];
    return adv_cell<T>{r, r * u[i], r * v[i], r * E[i]};
}

// limited slopes of cell C between its neighbours L and R (ref :105-116)
template <typename T>
__device__ __forceinline__ adv_cell<T> adv_slopes(const adv_cell<T>& L, const adv_cell<T>& C, const adv_cell<T>& R,
                                                  T D_L, T D_C, T D_R)
{
    const T r_m = xct::Den<T>(D_C + D_L).quo(2 * D_C);
    const T r_p = xct::Den<T>(D_C + D_R).quo(2 * D_C);
    return adv_cell<T>{phys::slope_minmod(L.rho, C.rho, R.rho, r_m, r_p), phys::slope_minmod(L.qu, C.qu, R.qu, r_m, r_p),
                       phys::slope_minmod(L.qv, C.qv, R.qv, r_m, r_p), phys::slope_minmod(L.qE, C.qE, R.qE, r_m, r_p)};
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_advection_second_order_x(armon_range r, T dx, T dt, const T* __restrict__ us, const T* __restrict__ rho,
                           const T* __restrict__ u, const T* __restrict__ v, const T* __restrict__ E,
                           T* __restrict__ a_rho, T* __restrict__ a_urho, T* __restrict__ a_vrho, T* __restrict__ a_Erho)
{
    using fused::from_next_lane;
    using fused::from_prev_lane;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#if ARMON_GAD_ALIGNED
    // whole-sector stores, as in k_acoustic_GAD_x: 64 - S interfaces per wave, S/2 halo lanes on either side (2 are needed)
    constexpr int kS = 64 / (int)sizeof(T), kLeft = kS / 2, kValid = 64 - kS;
    const int64_t w0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * kValid - ((r.col_start + r.row_start) & (kS - 1));
#else
    constexpr int kLeft = 2, kValid = kAdvValid;                                      // lanes 0,1 and 62,63: halo
    const int64_t w0 = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * kAdvValid;      // first interface of this wave
#endif
    if (w0 >= r.row_len) return;
    const int64_t k = w0 + lane - kLeft;
    const bool has_us = k >= -2 && k <= r.row_len + 1;
    const bool has_cell = k >= -2 && k <= r.row_len;                                  // ρ,u,v,E are read up to is + s only
    const bool stores = lane >= kLeft && lane < kLeft + kValid && k >= 0 && k < r.row_len;
    for (int64_t j = blockIdx.y; j < r.col_len; j += gridDim.y) {
        const int64_t i = row_base(r, j) + k;
        const T us_c = has_us ? us[i] : T(0);
        const adv_cell<T> C = has_cell ? adv_load(rho, u, v, E, i) : adv_cell<T>{1, 0, 0, 1};
        const T us_L = from_prev_lane(us_c), us_R = from_next_lane(us_c);
        const T D_C = dx + dt * (us_R - us_c);
        const T D_L = from_prev_lane(D_C), D_R = from_next_lane(D_C);
        const adv_cell<T> L{from_prev_lane(C.rho), from_prev_lane(C.qu), from_prev_lane(C.qv), from_prev_lane(C.qE)};
        const adv_cell<T> R{from_next_lane(C.rho), from_next_lane(C.qu), from_next_lane(C.qv), from_next_lane(C.qE)};
        const adv_cell<T> sl = adv_slopes(L, C, R, D_L, D_C, D_R);
        const adv_cell<T> slL{from_prev_lane(sl.rho), from_prev_lane(sl.qu), from_prev_lane(sl.qv), from_prev_lane(sl.qE)};
        const T disp = dt * us_c;
        const bool up = disp > 0;
        const T Dxe = up ? -(dx - dt * us_L) : (dx + dt * us_R);
        const T lf = xct::Den<T>(2 * (up ? D_L : D_C)).quo(Dxe);
        if (stores) {
            st_stream(a_rho + i,  disp * ((up ? L.rho : C.rho) - (up ? slL.rho : sl.rho) * lf));
            st_stream(a_urho + i, disp * ((up ? L.qu : C.qu) - (up ? slL.qu : sl.qu) * lf));
            st_stream(a_vrho + i, disp * ((up ? L.qv : C.qv) - (up ? slL.qv : sl.qv) * lf));
            st_stream(a_Erho + i, disp * ((up ? L.qE : C.qE) - (up ? slL.qE : sl.qE) * lf));
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_advection_second_order_y(armon_range r, int64_t s, T dx, T dt, const T* __restrict__ us, const T* __restrict__ rho,
                           const T* __restrict__ u, const T* __restrict__ v, const T* __restrict__ E,
                           T* __restrict__ a_rho, T* __restrict__ a_urho, T* __restrict__ a_vrho, T* __restrict__ a_Erho)
{
    const int64_t k = row_thread<T>(r);
    if (k < 0 || k >= r.row_len) return;
    const int64_t j0 = (int64_t)blockIdx.y * kAdvRows;
    const int64_t j1 = (j0 + kAdvRows < r.col_len) ? j0 + kAdvRows : r.col_len;
    int64_t i = row_base(r, j0) + k;
    T us_L = us[i - s], us_C = us[i], us_R = us[i + s];
    const T us_LL = us[i - 2 * s];
    adv_cell<T> L = adv_load(rho, u, v, E, i - s), C = adv_load(rho, u, v, E, i);
    T D_L = dx + dt * (us_C - us_L), D_C = dx + dt * (us_R - us_C);
    adv_cell<T> slL = adv_slopes(adv_load(rho, u, v, E, i - 2 * s), L, C, dx + dt * (us_L - us_LL), D_L, D_C);
    T us_RR = us[i + 2 * s];                                                            // rows j0 + 2 / j0 + 1
    T rn = rho[i + s], un = u[i + s], vn = v[i + s], En = E[i + s];
    for (int64_t j = j0; j < j1; j++, i += s) {
        const adv_cell<T> R{rn, rn * un, rn * vn, rn * En};
        const T D_R = dx + dt * (us_RR - us_R);
        // the next row's operands are requested before this row's arithmetic
        const int64_t in = (j + 1 < j1) ? i + 2 * s : i + s;
        const T us_n = us[in + s];
        rn = rho[in]; un = u[in]; vn = v[in]; En = E[in];
        const adv_cell<T> sl = adv_slopes(L, C, R, D_L, D_C, D_R);
        const T disp = dt * us_C;
        const bool up = disp > 0;
        const T Dxe = up ? -(dx - dt * us_L) : (dx + dt * us_R);
        const T lf = xct::Den<T>(2 * (up ? D_L : D_C)).quo(Dxe);
        st_stream(a_rho + i,  disp * ((up ? L.rho : C.rho) - (up ? slL.rho : sl.rho) * lf));
        st_stream(a_urho + i, disp * ((up ? L.qu : C.qu) - (up ? slL.qu : sl.qu) * lf));
        st_stream(a_vrho + i, disp * ((up ? L.qv : C.qv) - (up ? slL.qv : sl.qv) * lf));
        st_stream(a_Erho + i, disp * ((up ? L.qE : C.qE) - (up ? slL.qE : sl.qE) * lf));
        L = C; C = R;
        D_L = D_C; D_C = D_R;
        us_L = us_C; us_C = us_R; us_R = us_RR; us_RR = us_n;
        slL = sl;
    }
}

// ---- a9: euler_projection! (ref src/projection_schemes.jl:23-41) -----------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_euler_projection(armon_range r, int64_t s, T dx, T dt, const T* __restrict__ us,
                   T* __restrict__ rho, T* __restrict__ u, T* __restrict__ v,
                   T* __restrict__ E, const T* __restrict__ a_rho,
                   const T* __restrict__ a_urho, const T* __restrict__ a_vrho,
                   const T* __restrict__ a_Erho)
{
    ARMON_FOR_RANGE(r, i) {
        T rho_i = rho[i], u_i = u[i], v_i = v[i], E_i = E[i];
        phys::euler_projection(dx, dt, us[i], us[i + s], a_rho[i], a_rho[i + s], a_urho[i],
                               a_urho[i + s], a_vrho[i], a_vrho[i + s], a_Erho[i], a_Erho[i + s],
                               rho_i, u_i, v_i, E_i);
        rho[i] = rho_i;
        u[i] = u_i;
        v[i] = v_i;
        E[i] = E_i;
    }
}

// ---- a10: boundary_conditions! (ref src/halo_exchange.jl:2-29) -------------------------------------
// One thread per border cell (linear over the strip), looping over the ghost layers.
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_boundary_conditions(armon_range r, int64_t incr, int nghost, T u_factor, T v_factor,
                      T* __restrict__ rho, T* __restrict__ u, T* __restrict__ v,
                      T* __restrict__ p, T* __restrict__ c, T* __restrict__ g,
                      T* __restrict__ E)
{
    const int64_t n = r.col_len * r.row_len;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t j = t / r.row_len, k = t - j * r.row_len;
        int64_t i = row_base(r, j) + k;
        int64_t ig = i + incr;
        for (int l = 0; l < nghost; l++) {
            rho[ig] = rho[i];
            u[ig] = u[i] * u_factor;
            v[ig] = v[i] * v_factor;
            p[ig] = p[i];
            c[ig] = c[i];
            g[ig] = g[i];
            E[ig] = E[i];
            i -= incr;
            ig += incr;
        }
    }
}

// ---- a13: pack_to_array! / unpack_from_array! (ref src/halo_exchange.jl:187-216) -------------------
constexpr int kMaxPackVars = 8;
template <typename T> struct pack_vars { T* v[kMaxPackVars]; };   // passed by value: no device-side pointer table

template <bool PACK, typename T>
__device__ __forceinline__ void pack_range(const armon_range& r, int nghost, int64_t face, T* __restrict__ array, int nvars,
                                           const pack_vars<T>& vars)
{
    const int64_t n = r.col_len * r.row_len;
    for (int64_t itr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; itr < n;
         itr += (int64_t)gridDim.x * blockDim.x) {
        int64_t j = itr / r.row_len, k = itr - j * r.row_len;
        int64_t idx = row_base(r, j) + k;
        int64_t i = itr / nghost, i_g = itr - i * nghost;   // divrem(itr, ghosts)
        int64_t i_arr = (i_g * face + i) * nvars;
        for (int v = 0; v < nvars; v++) {
            if (PACK) array[i_arr + v] = vars.v[v][idx];
            else vars.v[v][idx] = array[i_arr + v];
        }
    }
}

template <bool PACK, typename T>
__global__ void __launch_bounds__(kBlock)
k_pack(armon_range r, int nghost, int64_t face, T* __restrict__ array, int nvars, pack_vars<T> vars)
{
    pack_range<PACK, T>(r, nghost, face, array, nvars, vars);
}

// the two sides of an axis in ONE launch (blockIdx.y = side): what the multi-GPU exchange packs / unpacks per sweep
// (csrc/multi_gpu.hip) — same layout, one dependent launch less on the transfer stream's chain
template <bool PACK, typename T>
__global__ void __launch_bounds__(kBlock)
k_pack_pair(armon_range r0, armon_range r1, int nghost, int64_t face, T* __restrict__ array0, T* __restrict__ array1, int nvars,
            pack_vars<T> vars)
{
    if (blockIdx.y == 0) pack_range<PACK, T>(r0, nghost, face, array0, nvars, vars);
    else pack_range<PACK, T>(r1, nghost, face, array1, nvars, vars);
}

// ---- a16: init_test (ref src/kernels.jl:71-145, src/tests.jl:59-121) -------------------------------
template <typename T> struct block_ptrs {
    T *x, *y, *rho, *u, *v, *E, *p, *c, *g, *us, *ps, *work_1, *work_2, *work_3, *work_4, *mask;
};

template <typename T> struct two_state { T hi_rho, lo_rho, hi_E, lo_E, hi_u, lo_u, hi_v, lo_v; };

template <typename T>
__device__ __forceinline__ bool region_high(int test, T x, T y, T sedov_r)
{
    switch (test) {
    case ARMON_TEST_SOD:       return x <= T(0.5);
    case ARMON_TEST_SOD_Y:     return y <= T(0.5);
    case ARMON_TEST_SOD_CIRC:  return (x - T(0.5)) * (x - T(0.5)) + (y - T(0.5)) * (y - T(0.5)) <= T(0.09);
    case ARMON_TEST_BIZARRIUM: return x <= T(0.5);
    case ARMON_TEST_SEDOV:     return x * x + y * y <= sedov_r * sedov_r;
    default:                   return false;
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_init_test(armon_range r, int test, int64_t row_length, int64_t nx, int64_t ny, int nghost,
            int64_t gpos_x, int64_t gpos_y, int64_t gN_x, T ox, T oy, T dXx, T dXy,
            T sedov_r, two_state<T> tp, block_ptrs<T> d)
{
    ARMON_FOR_RANGE(r, i) {
        int64_t Iy = i / row_length;
        int64_t Ix = i - Iy * row_length - nghost + 1;
        Iy = Iy - nghost + 1;
        int64_t gx = Ix + gpos_x - 1, gy = Iy + gpos_y - 1;
        T x = (T)gx * dXx + ox;
        T y = (T)gy * dXy + oy;
        d.x[i] = x;
        d.y[i] = y;
        bool ghost = !(Ix >= 1 && Ix <= nx && Iy >= 1 && Iy <= ny);
        // (the vectors outside the state may be absent — the reference's `vars_to_zero` list, src/kernels.jl:142-144,188-191:
        // a host that runs the fused sweeps never reads us, ps, work_1..4, mask, and c, g only on its first cycle)
        if (d.mask) d.mask[i] = ghost ? T(0.) : T(1.);
        if (test == ARMON_TEST_DEBUG_INDEXES) {
            T gi = (T)(gx + gy * gN_x + 1);
            d.rho[i] = gi; d.E[i] = gi; d.u[i] = gi; d.v[i] = gi;
            if (d.p) d.p[i] = gi;
            if (d.c) d.c[i] = gi;
            if (d.g) d.g[i] = gi;
        } else {
            bool hi = region_high(test, x + dXx / 2, y + dXy / 2, sedov_r);
            d.rho[i] = hi ? tp.hi_rho : tp.lo_rho;
            d.E[i] = hi ? tp.hi_E : tp.lo_E;
            d.u[i] = hi ? tp.hi_u : tp.lo_u;
            d.v[i] = hi ? tp.hi_v : tp.lo_v;
            if (d.p) d.p[i] = 0.;
            if (d.c) d.c[i] = 0.;
            if (d.g) d.g[i] = 0.;
        }
        if (d.us) d.us[i] = 0.;
        if (d.ps) d.ps[i] = 0.;
        if (d.work_1) d.work_1[i] = 0.;
        if (d.work_2) d.work_2[i] = 0.;
        if (d.work_3) d.work_3[i] = 0.;
        if (d.work_4) d.work_4[i] = 0.;
    }
}

inline void linear_grid(const armon_ctx* ctx, int64_t n, dim3& grid, dim3& block)
{
    block = dim3(kBlock);
    int64_t g = (n + kBlock - 1) / kBlock;
    int64_t cap = (int64_t)ctx->n_cu * 8;
    grid = dim3((unsigned)(g < cap ? (g < 1 ? 1 : g) : cap));
}

}  // namespace

#define ARMON_CHECK_CTX_RANGE(ctx, r)                                                  \
    ARMON_REQUIRE((ctx) != nullptr, "ctx is NULL");                                    \
    ARMON_REQUIRE(range_ok(r), "invalid range (col %lld:%lld:%lld row %lld+%lld)",     \
                  (long long)(r).col_start, (long long)(r).col_step, (long long)(r).col_len, \
                  (long long)(r).row_start, (long long)(r).row_len);                   \
    if (range_empty(r)) return ARMON_OK

// ---- implementations, generic in the working precision ---------------------------------------------------
namespace {

template <typename T>
int perfect_gas_EOS_impl(armon_ctx* ctx, armon_range r, T gamma, const T* rho, const T* E, const T* u, const T* v,
                         T* p, T* c, T* g)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(rho && E && u && v && p && c && g, "NULL array");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_perfect_gas_EOS<T>, grid, block, 0, ctx->stream, r, gamma, rho, E, u, v, p, c, g);
    return check_launch("perfect_gas_EOS");
}

template <typename T>
int bizarrium_EOS_impl(armon_ctx* ctx, armon_range r, const T* rho, const T* u, const T* v, const T* E,
                       T* p, T* c, T* g)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(rho && E && u && v && p && c && g, "NULL array");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_bizarrium_EOS<T>, grid, block, 0, ctx->stream, r, rho, u, v, E, p, c, g);
    return check_launch("bizarrium_EOS");
}

template <typename T>
int acoustic_impl(armon_ctx* ctx, armon_range r, int64_t s, T* us, T* ps, const T* rho, const T* ua, const T* p, const T* c)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(us && ps && rho && ua && p && c, "NULL array");
    ARMON_REQUIRE(s > 0, "stride must be positive");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_acoustic<T>, grid, block, 0, ctx->stream, r, s, us, ps, rho, ua, p, c);
    return check_launch("acoustic");
}

#if ARMON_GAD_X2
template <int LIM>
void launch_gad_x2(dim3 grid, dim3 block, hipStream_t st, armon_range r, double dt, double dx, double* us, double* ps,
                   const double* rho, const double* ua, const double* p, const double* c)
{
    hipLaunchKernelGGL((k_acoustic_GAD_x2<LIM>), grid, block, 0, st, r, dt, dx, us, ps, rho, ua, p, c);
}
template <int LIM>
void launch_gad_x2(dim3, dim3, hipStream_t, armon_range, float, float, float*, float*, const float*, const float*, const float*,
                   const float*)
{
}   // fp32 keeps the one-cell-per-lane form (form 3 is never chosen for it)
#endif

template <typename T>
int acoustic_GAD_impl(armon_ctx* ctx, armon_range r, int64_t s, T dt, T dx, T* us, T* ps, const T* rho, const T* ua,
                      const T* p, const T* c, int limiter)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(us && ps && rho && ua && p && c, "NULL array");
    ARMON_REQUIRE(s > 0, "stride must be positive");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    // form of the kernel: 1 = lanes along the sweep (s == 1), 2 = march along the sweep (s == row pitch), 0 = the
    // reference's own shape (any other stride)
    int form = (s == 1) ? 1 : ((s == r.col_step && r.col_len > 1) ? 2 : 0);
    if (form == 1) grid.x = (unsigned)(((r.row_len + kGadValid - 1) / kGadValid + kBlock / 64 - 1) / (kBlock / 64));   // waves of 62 fluxes
#if ARMON_GAD_ALIGNED
    if (form == 1) {
        const int64_t S = 64 / (int64_t)sizeof(T), valid = 64 - S;
        grid.x = (unsigned)(((r.row_len + ((r.col_start + r.row_start) & (S - 1)) + valid - 1) / valid + kBlock / 64 - 1) / (kBlock / 64));
    }
#endif
#if ARMON_GAD_X2
    if (form == 1 && sizeof(T) == 8 && r.col_step % 2 == 0 &&
        ((uintptr_t)us | (uintptr_t)ps | (uintptr_t)rho | (uintptr_t)ua | (uintptr_t)p | (uintptr_t)c) % 16 == 0) {
        form = 3;                                                    // strips of 120 fluxes, a workgroup = 4 rows of one strip
        grid.x = (unsigned)((r.row_len + ((r.col_start + r.row_start) & 7) + kGadStride2 - 1) / kGadStride2);
        const int64_t gy = (r.col_len + kBlock / 64 - 1) / (kBlock / 64);
        grid.y = (unsigned)(gy < 65535 ? gy : 65535);
    }
#endif
    if (form == 2) grid.y = (unsigned)((r.col_len + kGadRows - 1) / kGadRows);
#if ARMON_GAD_X2
#define ARMON_GAD_X2_LAUNCH(LIM) launch_gad_x2<LIM>(grid, block, ctx->stream, r, dt, dx, us, ps, rho, ua, p, c)
#else
#define ARMON_GAD_X2_LAUNCH(LIM) (void)0
#endif
#define ARMON_GAD_LAUNCH(LIM)                                                                                              \
    do {                                                                                                                   \
        if (form == 3)                                                                                                     \
            ARMON_GAD_X2_LAUNCH(LIM);                                                                                      \
        else if (form == 1)                                                                                                \
            hipLaunchKernelGGL((k_acoustic_GAD_x<LIM, T>), grid, block, 0, ctx->stream, r, dt, dx, us, ps, rho, ua, p, c);  \
        else if (form == 2)                                                                                                \
            hipLaunchKernelGGL((k_acoustic_GAD_y<LIM, T>), grid, block, 0, ctx->stream, r, s, dt, dx, us, ps, rho, ua, p, c); \
        else                                                                                                               \
            hipLaunchKernelGGL((k_acoustic_GAD<LIM, T>), grid, block, 0, ctx->stream, r, s, dt, dx, us, ps, rho, ua, p, c); \
    } while (0)
    switch (limiter) {
    case ARMON_LIMITER_NONE: ARMON_GAD_LAUNCH(ARMON_LIMITER_NONE); break;
    case ARMON_LIMITER_MINMOD: ARMON_GAD_LAUNCH(ARMON_LIMITER_MINMOD); break;
    case ARMON_LIMITER_SUPERBEE: ARMON_GAD_LAUNCH(ARMON_LIMITER_SUPERBEE); break;
    default:
        ARMON_REQUIRE(false, "unknown limiter tag %d", limiter);
    }
#undef ARMON_GAD_LAUNCH
    return check_launch("acoustic_GAD");
}

template <typename T>
int cell_update_impl(armon_ctx* ctx, armon_range r, int64_t s, T dx, T dt, const T* us, const T* ps, T* rho, T* ua, T* E)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(us && ps && rho && ua && E, "NULL array");
    ARMON_REQUIRE(s > 0, "stride must be positive");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_cell_update<T>, grid, block, 0, ctx->stream, r, s, dx, dt, us, ps, rho, ua, E);
    return check_launch("cell_update");
}

template <typename T>
int advection_first_order_impl(armon_ctx* ctx, armon_range r, int64_t s, T dt, const T* us, const T* rho, const T* u,
                               const T* v, const T* E, T* a_rho, T* a_urho, T* a_vrho, T* a_Erho)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(us && rho && u && v && E && a_rho && a_urho && a_vrho && a_Erho, "NULL array");
    ARMON_REQUIRE(s > 0, "stride must be positive");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_advection_first_order<T>, grid, block, 0, ctx->stream, r, s, dt, us, rho, u, v, E,
                       a_rho, a_urho, a_vrho, a_Erho);
    return check_launch("advection_first_order");
}

template <typename T>
int advection_second_order_impl(armon_ctx* ctx, armon_range r, int64_t s, T dx, T dt, const T* us, const T* rho,
                                const T* u, const T* v, const T* E, T* a_rho, T* a_urho, T* a_vrho, T* a_Erho)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(us && rho && u && v && E && a_rho && a_urho && a_vrho && a_Erho, "NULL array");
    ARMON_REQUIRE(s > 0, "stride must be positive");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    if (s == 1) {                                         // lanes along the sweep: 60 interfaces per wave
        grid.x = (unsigned)(((r.row_len + kAdvValid - 1) / kAdvValid + kBlock / 64 - 1) / (kBlock / 64));
#if ARMON_GAD_ALIGNED
        {
            const int64_t S = 64 / (int64_t)sizeof(T), valid = 64 - S;
            grid.x = (unsigned)(((r.row_len + ((r.col_start + r.row_start) & (S - 1)) + valid - 1) / valid + kBlock / 64 - 1) / (kBlock / 64));
        }
#endif
        hipLaunchKernelGGL(k_advection_second_order_x<T>, grid, block, 0, ctx->stream, r, dx, dt, us, rho, u,
                           v, E, a_rho, a_urho, a_vrho, a_Erho);
    } else if (s == r.col_step && r.col_len > 1) {        // march along the sweep
        grid.y = (unsigned)((r.col_len + kAdvRows - 1) / kAdvRows);
        hipLaunchKernelGGL(k_advection_second_order_y<T>, grid, block, 0, ctx->stream, r, s, dx, dt, us, rho, u,
                           v, E, a_rho, a_urho, a_vrho, a_Erho);
    } else {
        hipLaunchKernelGGL(k_advection_second_order<T>, grid, block, 0, ctx->stream, r, s, dx, dt, us, rho, u,
                           v, E, a_rho, a_urho, a_vrho, a_Erho);
    }
    return check_launch("advection_second_order");
}

template <typename T>
int euler_projection_impl(armon_ctx* ctx, armon_range r, int64_t s, T dx, T dt, const T* us, T* rho, T* u, T* v, T* E,
                          const T* a_rho, const T* a_urho, const T* a_vrho, const T* a_Erho)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(us && rho && u && v && E && a_rho && a_urho && a_vrho && a_Erho, "NULL array");
    ARMON_REQUIRE(s > 0, "stride must be positive");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_euler_projection<T>, grid, block, 0, ctx->stream, r, s, dx, dt, us, rho, u, v, E,
                       a_rho, a_urho, a_vrho, a_Erho);
    return check_launch("euler_projection");
}

template <typename T>
int boundary_conditions_impl(armon_ctx* ctx, armon_range r, int64_t incr, int nghost, T u_factor, T v_factor,
                             T* rho, T* u, T* v, T* p, T* c, T* g, T* E)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(rho && u && v && p && c && g && E, "NULL array");
    ARMON_REQUIRE(incr != 0 && nghost > 0, "invalid incr/nghost");
    dim3 grid, block;
    linear_grid(ctx, r.col_len * r.row_len, grid, block);
    hipLaunchKernelGGL(k_boundary_conditions<T>, grid, block, 0, ctx->stream, r, incr, nghost, u_factor,
                       v_factor, rho, u, v, p, c, g, E);
    return check_launch("boundary_conditions");
}

template <typename T>
int pack_impl(armon_ctx* ctx, armon_range r, int nghost, int64_t face, T* array, int nvars, const T* const* vars, bool pack)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(array && vars, "NULL array");
    ARMON_REQUIRE(nvars > 0 && nvars <= kMaxPackVars, "nvars must be in [1,%d]", kMaxPackVars);
    ARMON_REQUIRE(nghost > 0 && face > 0, "invalid nghost/face");
    ARMON_REQUIRE(r.col_len * r.row_len <= face * nghost, "range larger than face*nghost");
    pack_vars<T> table = {};
    for (int v = 0; v < nvars; v++) {
        ARMON_REQUIRE(vars[v] != nullptr, "NULL array in vars[%d]", v);
        table.v[v] = const_cast<T*>(vars[v]);
    }
    dim3 grid, block;
    linear_grid(ctx, r.col_len * r.row_len, grid, block);
    if (pack)
        hipLaunchKernelGGL((k_pack<true, T>), grid, block, 0, ctx->stream, r, nghost, face, array, nvars, table);
    else
        hipLaunchKernelGGL((k_pack<false, T>), grid, block, 0, ctx->stream, r, nghost, face, array, nvars, table);
    return check_launch(pack ? "pack_to_array" : "unpack_from_array");
}

template <typename T>
int unpack_impl(armon_ctx* ctx, armon_range r, int nghost, int64_t face, const T* array, int nvars, T* const* vars)
{
    return pack_impl<T>(ctx, r, nghost, face, const_cast<T*>(array), nvars, const_cast<const T* const*>(vars), false);
}

template <typename T, typename BD>
int init_test_impl(armon_ctx* ctx, armon_range r, int test, int64_t row_length, int64_t col_length, int nghost,
                   const int64_t global_pos[2], const int64_t global_N[2], const T origin[2], const T dX[2],
                   T sedov_r, const BD* d)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(global_pos && global_N && origin && dX && d, "NULL argument");
    ARMON_REQUIRE(test >= ARMON_TEST_SOD && test <= ARMON_TEST_DEBUG_INDEXES, "unknown test tag %d", test);
    static_assert(sizeof(BD) == 16 * sizeof(T*), "block data = 16 pointers");
    T* const* arrs = reinterpret_cast<T* const*>(d);
    for (int k = 0; k < 6; k++) ARMON_REQUIRE(arrs[k] != nullptr, "NULL array in block data (field %d: x, y, rho, u, v, E are required)", k);
    block_ptrs<T> bp;
    memcpy(&bp, d, sizeof(bp));
    // ref src/tests.jl:84-121
    two_state<T> tp;
    switch (test) {
    case ARMON_TEST_BIZARRIUM:
        tp = { T(1.42857142857e+4), T(10000.), T(4.48657821135e+6), T(0.5 * (250. * 250.)), T(0.), T(250.), T(0.), T(0.) };
        break;
    case ARMON_TEST_SEDOV:
        // ref src/tests.jl:112: T((1/1.033)^5 / (π * p.r^2)) — π·r² in T, the quotient in Float64
        tp = { T(1.), T(1.), T(pow(1. / 1.033, 5) / (double)(T(M_PI) * (sedov_r * sedov_r))), T(2.5e-14), T(0.), T(0.), T(0.), T(0.) };
        break;
    default:
        tp = { T(1.), T(0.125), T(2.5), T(2.0), T(0.), T(0.), T(0.), T(0.) };
    }
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_init_test<T>, grid, block, 0, ctx->stream, r, test, row_length,
                       row_length - 2 * nghost, col_length - 2 * nghost, nghost, global_pos[0],
                       global_pos[1], global_N[0], origin[0], origin[1], dX[0], dX[1], sedov_r, tp, bp);
    return check_launch("init_test");
}

}  // namespace

// Both sides of an axis in one launch, for the exchange of csrc/multi_gpu.hip (declared in common.hpp). The two ranges have
// the same shape (the low and the high border / ghost domain of one axis).
namespace armon {
template <typename T>
int pack_pair(armon_ctx* ctx, const armon_range r[2], int nghost, int64_t face, T* const array[2], int nvars, T* const* vars, bool pack)
{
    ARMON_REQUIRE(ctx && array[0] && array[1] && vars, "NULL argument");
    ARMON_REQUIRE(nvars > 0 && nvars <= kMaxPackVars && nghost > 0 && face > 0, "invalid nvars / nghost / face");
    ARMON_REQUIRE(range_ok(r[0]) && range_ok(r[1]) && r[0].col_len == r[1].col_len && r[0].row_len == r[1].row_len &&
                  r[0].col_len * r[0].row_len <= face * nghost, "the two sides of an axis must have the same, valid shape");
    if (range_empty(r[0])) return ARMON_OK;
    pack_vars<T> table = {};
    for (int v = 0; v < nvars; v++) {
        ARMON_REQUIRE(vars[v] != nullptr, "NULL array in vars[%d]", v);
        table.v[v] = vars[v];
    }
    dim3 grid, block;
    linear_grid(ctx, r[0].col_len * r[0].row_len, grid, block);
    grid.y = 2;
    if (pack)
        hipLaunchKernelGGL((k_pack_pair<true, T>), grid, block, 0, ctx->stream, r[0], r[1], nghost, face, array[0], array[1], nvars, table);
    else
        hipLaunchKernelGGL((k_pack_pair<false, T>), grid, block, 0, ctx->stream, r[0], r[1], nghost, face, array[0], array[1], nvars, table);
    return check_launch(pack ? "pack_pair" : "unpack_pair");
}
template int pack_pair<double>(armon_ctx*, const armon_range[2], int, int64_t, double* const[2], int, double* const*, bool);
template int pack_pair<float>(armon_ctx*, const armon_range[2], int, int64_t, float* const[2], int, float* const*, bool);
}  // namespace armon

// ---- C ABI: every entry point exists for fp64 (reference names) and fp32 (`_f32` suffix) -------------------------
#define ARMON_EXPORT(name, impl, PARAMS, ARGS)                                            \
    int armon_hip_##name(PARAMS(double)) { return impl<double> ARGS; }                    \
    int armon_hip_##name##_f32(PARAMS(float)) { return impl<float> ARGS; }

extern "C" {

#define P_PG(T) armon_ctx* ctx, armon_range r, T gamma, const T* rho, const T* E, const T* u, const T* v, T* p, T* c, T* g
ARMON_EXPORT(perfect_gas_EOS, perfect_gas_EOS_impl, P_PG, (ctx, r, gamma, rho, E, u, v, p, c, g))

#define P_BZ(T) armon_ctx* ctx, armon_range r, const T* rho, const T* u, const T* v, const T* E, T* p, T* c, T* g
ARMON_EXPORT(bizarrium_EOS, bizarrium_EOS_impl, P_BZ, (ctx, r, rho, u, v, E, p, c, g))

#define P_AC(T) armon_ctx* ctx, armon_range r, int64_t s, T* us, T* ps, const T* rho, const T* ua, const T* p, const T* c
ARMON_EXPORT(acoustic, acoustic_impl, P_AC, (ctx, r, s, us, ps, rho, ua, p, c))

#define P_GAD(T) armon_ctx* ctx, armon_range r, int64_t s, T dt, T dx, T* us, T* ps, const T* rho, const T* ua, const T* p, const T* c, int limiter
ARMON_EXPORT(acoustic_GAD, acoustic_GAD_impl, P_GAD, (ctx, r, s, dt, dx, us, ps, rho, ua, p, c, limiter))

#define P_CU(T) armon_ctx* ctx, armon_range r, int64_t s, T dx, T dt, const T* us, const T* ps, T* rho, T* ua, T* E
ARMON_EXPORT(cell_update, cell_update_impl, P_CU, (ctx, r, s, dx, dt, us, ps, rho, ua, E))

#define P_A1(T) armon_ctx* ctx, armon_range r, int64_t s, T dt, const T* us, const T* rho, const T* u, const T* v, const T* E, T* a_rho, T* a_urho, T* a_vrho, T* a_Erho
ARMON_EXPORT(advection_first_order, advection_first_order_impl, P_A1, (ctx, r, s, dt, us, rho, u, v, E, a_rho, a_urho, a_vrho, a_Erho))

#define P_A2(T) armon_ctx* ctx, armon_range r, int64_t s, T dx, T dt, const T* us, const T* rho, const T* u, const T* v, const T* E, T* a_rho, T* a_urho, T* a_vrho, T* a_Erho
ARMON_EXPORT(advection_second_order, advection_second_order_impl, P_A2, (ctx, r, s, dx, dt, us, rho, u, v, E, a_rho, a_urho, a_vrho, a_Erho))

#define P_EP(T) armon_ctx* ctx, armon_range r, int64_t s, T dx, T dt, const T* us, T* rho, T* u, T* v, T* E, const T* a_rho, const T* a_urho, const T* a_vrho, const T* a_Erho
ARMON_EXPORT(euler_projection, euler_projection_impl, P_EP, (ctx, r, s, dx, dt, us, rho, u, v, E, a_rho, a_urho, a_vrho, a_Erho))

#define P_BC(T) armon_ctx* ctx, armon_range r, int64_t incr, int nghost, T u_factor, T v_factor, T* rho, T* u, T* v, T* p, T* c, T* g, T* E
ARMON_EXPORT(boundary_conditions, boundary_conditions_impl, P_BC, (ctx, r, incr, nghost, u_factor, v_factor, rho, u, v, p, c, g, E))

#define P_PK(T) armon_ctx* ctx, armon_range r, int nghost, int64_t face, T* array, int nvars, const T* const* vars
ARMON_EXPORT(pack_to_array, pack_impl, P_PK, (ctx, r, nghost, face, array, nvars, vars, true))

#define P_UP(T) armon_ctx* ctx, armon_range r, int nghost, int64_t face, const T* array, int nvars, T* const* vars
ARMON_EXPORT(unpack_from_array, unpack_impl, P_UP, (ctx, r, nghost, face, array, nvars, vars))

int armon_hip_init_test(armon_ctx* ctx, armon_range r, int test, int64_t row_length, int64_t col_length, int nghost,
                        const int64_t global_pos[2], const int64_t global_N[2], const double origin[2],
                        const double dX[2], double sedov_r, const armon_block_data* d)
{
    return init_test_impl<double>(ctx, r, test, row_length, col_length, nghost, global_pos, global_N, origin, dX, sedov_r, d);
}

int armon_hip_init_test_f32(armon_ctx* ctx, armon_range r, int test, int64_t row_length, int64_t col_length, int nghost,
                            const int64_t global_pos[2], const int64_t global_N[2], const float origin[2],
                            const float dX[2], float sedov_r, const armon_block_data_f32* d)
{
    return init_test_impl<float>(ctx, r, test, row_length, col_length, nghost, global_pos, global_N, origin, dX, sedov_r, d);
}

}  // extern "C"
