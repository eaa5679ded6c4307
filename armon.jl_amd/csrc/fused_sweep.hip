// fused_sweep.hip — armon_hip_sweep: one directional sweep of solver_cycle (ref src/solver.jl:300-316)
// as ONE kernel launch: EOS → boundary mirror → fluxes → cell update → advection → projection, reading
// ρ,u,v,E once and writing them once (64 B per cell instead of the 352 B of the five staged passes).
//
// Both kernels run the register pipeline of sweep_pipeline.hpp; they differ only in how a lane gets the
// cells it marches over:
//  * Y sweep: lane ↔ column. Rows are x-contiguous, so every step of the march is one fully coalesced
//    row segment per wave; no LDS at all. A workgroup owns 256 columns × one run of rows.
//  * X sweep: the march runs along the contiguous axis, so lane ↔ row and the wave transposes through
//    LDS: a 64-row × CH-column tile is loaded with coalesced row segments, each lane walks its row in
//    LDS (odd row pitch → conflict-free column reads), writes the results back in place, and the tile
//    is stored with coalesced row segments again. One wave per workgroup, so the barriers are free.
// Redundant work is confined to the LAG (≤4) cells at both ends of a run.
#include "common.hpp"
#include "sweep_pipeline.hpp"
#include "sweep_spatial.hpp"

#include <type_traits>
#include <utility>

using namespace armon;

namespace {

struct sweep_args {
    int64_t nx, ny, row_len;       // real cells and array pitch (nx + 2g)
    int32_t g;                     // ghost layers
    int32_t bc_low, bc_high;       // mirror BC applied in-kernel on that side of the sweep axis
    int32_t emit;                  // bit 0: write p_out, bit 1: write c_out
    int64_t seg;                   // cells per run along the sweep axis
    int32_t x_kernel;              // X sweep form: 0 spatial (DPP), 1 LDS-transposed march (vector), 2 generic march
    double dt, dx, gamma;
    double fa_low, ft_low, fa_high, ft_high;   // BC factors: axial / transverse velocity
    const double *rho_in, *ua_in, *ut_in, *E_in;    // ua = velocity along the sweep axis
    double *rho_out, *ua_out, *ut_out, *E_out;
    double *p_out, *c_out;
};

// Source index and velocity factors of cell `j` (0-based real coordinate along the sweep axis, may be
// a ghost): physical boundaries mirror the inside (ref src/halo_exchange.jl:2-29), process boundaries
// read the ghost cells filled by the halo exchange.
__device__ __forceinline__ int64_t bc_source(const sweep_args& a, int64_t n, int64_t j, double& fa, double& ft)
{
    fa = 1.;
    ft = 1.;
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

template <int... Is, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F&& f)
{
    (f(std::integral_constant<int, Is>{}), ...);
}

// ---- Y sweep ---------------------------------------------------------------------------------------
constexpr int kYBlock = 256;

template <class PIPE>
__global__ void __launch_bounds__(kYBlock)
k_sweep_y(sweep_args a)
{
    constexpr int LAG = PIPE::LAG;
    constexpr int PF = 4;            // rows in flight per lane (= the unroll of the march)
    const int64_t x = (int64_t)blockIdx.x * kYBlock + threadIdx.x;
    if (x >= a.nx) return;
    const int64_t o0 = (int64_t)blockIdx.y * a.seg;
    const int64_t o1 = (o0 + a.seg < a.ny) ? o0 + a.seg : a.ny;
    const unsigned col = (unsigned)(x + a.g);    // 32-bit lane offset; row bases stay scalar

    PIPE pipe(a.dt, a.dx, a.gamma);

    const int64_t j_begin = o0 - LAG, j_end = o1 + LAG;
    double pr[PF][4];                // prefetch ring: (ρ, ua, ut, E) of rows j .. j+PF-1
    // Loads one row of this lane's column (clamped to the run so that padding steps stay in bounds).
    auto load = [&](auto slot, int64_t j) {
        constexpr int K = decltype(slot)::value;
        double fa, ft;
        const int64_t jc = j < j_end ? j : j_end - 1;
        const int64_t row = (bc_source(a, a.ny, jc, fa, ft) + a.g) * a.row_len;   // uniform
        pr[K][0] = (a.rho_in + row)[col];
        pr[K][1] = (a.ua_in + row)[col] * fa;
        pr[K][2] = (a.ut_in + row)[col] * ft;
        pr[K][3] = (a.E_in + row)[col];
    };
    auto step = [&](auto ph, int64_t j) {
        constexpr int PH = decltype(ph)::value;
        const double rho = pr[PH][0], ua = pr[PH][1], ut = pr[PH][2], E = pr[PH][3];
        load(ph, j + PF);            // refill this slot: PF rows ahead of the march
        double p, c;
        const fused::Out4 out = pipe.template push<true, PH>(rho, ua, ut, E, p, c);
        if (a.emit && j >= o0 && j < o1) {
            const int64_t row = (j + a.g) * a.row_len;
            if (a.emit & 1) (a.p_out + row)[col] = p;
            if (a.emit & 2) (a.c_out + row)[col] = c;
        }
        const int64_t o = j - LAG;
        if (o >= o0 && o < o1) {
            const int64_t row = (o + a.g) * a.row_len;
            (a.rho_out + row)[col] = out.rho;
            (a.ua_out + row)[col] = out.ua;
            (a.ut_out + row)[col] = out.ut;
            (a.E_out + row)[col] = out.E;
        }
    };

    load(std::integral_constant<int, 0>{}, j_begin);
    load(std::integral_constant<int, 1>{}, j_begin + 1);
    load(std::integral_constant<int, 2>{}, j_begin + 2);
    load(std::integral_constant<int, 3>{}, j_begin + 3);
    // The march is unrolled by the 4 phases of the pipeline's history rings.
    for (int64_t j = j_begin; j < j_end; j += 4) {
        step(std::integral_constant<int, 0>{}, j);
        step(std::integral_constant<int, 1>{}, j + 1);
        step(std::integral_constant<int, 2>{}, j + 2);
        step(std::integral_constant<int, 3>{}, j + 3);
    }
}

// ---- X sweep ---------------------------------------------------------------------------------------
constexpr int kXRows = 64;      // one wave: lane ↔ row

template <class PIPE, int CH>
__global__ void __launch_bounds__(kXRows, 2)
k_sweep_x(sweep_args a)
{
    constexpr int LAG = PIPE::LAG;
    constexpr int PITCH = CH + 1;                 // odd pitch in doubles: conflict-free column walks
    constexpr int RPI = kXRows / CH;              // rows covered by one wave-wide row-segment access
    extern __shared__ double tile[];              // [planes][64][PITCH], planes = 4 (+2 when emitting p, c)

    const int lane = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * kXRows;
    const int64_t o0 = (int64_t)blockIdx.x * a.seg;
    const int64_t o1 = (o0 + a.seg < a.nx) ? o0 + a.seg : a.nx;
    const int64_t j_end = o1 + LAG;
    const bool row_ok = (r0 + lane) < a.ny;
    const bool emit = a.emit != 0;

    const int sub_row = lane / CH, sub_col = lane % CH;   // row-segment phase: lane → (row in group, column)
    auto T = [&](int plane, int r, int t) -> double& { return tile[(plane * kXRows + r) * PITCH + t]; };

    PIPE pipe(a.dt, a.dx, a.gamma);

    for (int64_t jb = o0 - LAG; jb < j_end; jb += CH) {
        // -- load phase: 64 rows × CH columns, 8 B per lane, CH*8 B contiguous per row
        {
            const int64_t j = jb + sub_col;
            double fa, ft;
            const int64_t src = bc_source(a, a.nx, j, fa, ft);
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int r = k * RPI + sub_row;
                const int64_t row = r0 + r;
                if (row < a.ny && j < j_end) {
                    const int64_t idx = (row + a.g) * a.row_len + (src + a.g);
                    T(0, r, sub_col) = a.rho_in[idx];
                    T(1, r, sub_col) = a.ua_in[idx] * fa;
                    T(2, r, sub_col) = a.ut_in[idx] * ft;
                    T(3, r, sub_col) = a.E_in[idx];
                }
            }
        }
        __syncthreads();
        // -- march: lane walks its own row through the tile, results overwrite the consumed slots
        if (row_ok) {
            static_for(std::make_integer_sequence<int, CH>{}, [&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const int64_t j = jb + t;
                if (j < j_end) {
                    double p, c;
                    const fused::Out4 out = pipe.template push<false, (t & 3)>(T(0, lane, t), T(1, lane, t), T(2, lane, t), T(3, lane, t), p, c);
                    T(0, lane, t) = out.rho;
                    T(1, lane, t) = out.ua;
                    T(2, lane, t) = out.ut;
                    T(3, lane, t) = out.E;
                    if (emit) {
                        T(4, lane, t) = p;
                        T(5, lane, t) = c;
                    }
                }
            });
        }
        __syncthreads();
        // -- store phase: slot t holds the new state of column jb + t - LAG (and p, c of column jb + t)
        {
            const int64_t j = jb + sub_col;
            const int64_t o = j - LAG;
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int r = k * RPI + sub_row;
                const int64_t row = r0 + r;
                if (row < a.ny && j < j_end) {
                    if (o >= o0) {
                        const int64_t io = (row + a.g) * a.row_len + (o + a.g);
                        a.rho_out[io] = T(0, r, sub_col);
                        a.ua_out[io] = T(1, r, sub_col);
                        a.ut_out[io] = T(2, r, sub_col);
                        a.E_out[io] = T(3, r, sub_col);
                    }
                    if (emit && j >= o0 && j < o1) {
                        const int64_t ij = (row + a.g) * a.row_len + (j + a.g);
                        if (a.emit & 1) a.p_out[ij] = T(4, r, sub_col);
                        if (a.emit & 2) a.c_out[ij] = T(5, r, sub_col);
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- X sweep, spatial form (lanes along x, DPP neighbour exchange) -------------------------------------
// blockDim = (64, kXSRows): one wave per row, kXSRows consecutive rows per workgroup. Each wave walks
// NITER strips of 64*K cells along its row; a strip yields 64*K - 2*HALO new cells.
constexpr int kXSRows = 4;

template <int SCHEME, int LIM, int PROJ, int EOS, bool EXACT, int K>
__global__ void __launch_bounds__(64 * kXSRows)
k_sweep_x_dpp(sweep_args a, int niter)
{
    using SW = fused::SpatialSweep<SCHEME, LIM, PROJ, EOS, EXACT, K>;
    using St = fused::Strip<K>;
    constexpr int LAG = SW::LAG;
    constexpr int HALO = (K == 1) ? LAG : ((LAG + 1) & ~1);   // even for K = 2: strips stay pair-aligned
    constexpr int WIDTH = 64 * K;
    constexpr int STRIDE = WIDTH - 2 * HALO;

    const int lane = threadIdx.x;
    const int64_t row = (int64_t)blockIdx.y * kXSRows + threadIdx.y;
    if (row >= a.ny) return;                       // whole wave
    const int64_t row_off = (row + a.g) * a.row_len + a.g;
    const double* in[4] = {a.rho_in + row_off, a.ua_in + row_off, a.ut_in + row_off, a.E_in + row_off};
    double* out[4] = {a.rho_out + row_off, a.ua_out + row_off, a.ut_out + row_off, a.E_out + row_off};
    const bool vec_ok = (K == 2) && (a.row_len % 2 == 0) && (a.g % 2 == 0);   // uniform

    SW sw{a.dt, a.dx, a.gamma};

    const int64_t w_first = (int64_t)blockIdx.x * niter * STRIDE;
    for (int it = 0; it < niter; it++) {
        const int64_t w0 = w_first + (int64_t)it * STRIDE;    // first cell this strip produces
        if (w0 >= a.nx) break;
        const int64_t cb = w0 - HALO;                         // first cell of the strip
        const int64_t j0 = cb + (int64_t)lane * K;            // this lane's first cell

        St rho, ua, ut, E;
        const bool interior = cb >= 0 && cb + WIDTH <= a.nx;  // uniform: no ghost, no clamping
        if (interior && (K == 1 || vec_ok)) {
            if (K == 2) {
                const double2 r = *reinterpret_cast<const double2*>(in[0] + j0);
                const double2 u = *reinterpret_cast<const double2*>(in[1] + j0);
                const double2 v = *reinterpret_cast<const double2*>(in[2] + j0);
                const double2 e = *reinterpret_cast<const double2*>(in[3] + j0);
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
                double fa, ft;
                const int64_t src = bc_source(a, a.nx, j, fa, ft);
                rho.v[k] = in[0][src];
                ua.v[k] = in[1][src] * fa;
                ut.v[k] = in[2][src] * ft;
                E.v[k] = in[3][src];
            }
        }

        St o_rho, o_u, o_v, o_E, p, cs;
        sw.run(rho, ua, ut, E, o_rho, o_u, o_v, o_E, p, cs);

        // cells this lane may store: inside the strip's valid window and inside the block
        const int64_t hi = (w0 + STRIDE < a.nx) ? w0 + STRIDE : a.nx;
        if (K == 2 && vec_ok && j0 >= w0 && j0 + 1 < hi) {
            *reinterpret_cast<double2*>(out[0] + j0) = double2{o_rho.v[0], o_rho.v[K - 1]};
            *reinterpret_cast<double2*>(out[1] + j0) = double2{o_u.v[0], o_u.v[K - 1]};
            *reinterpret_cast<double2*>(out[2] + j0) = double2{o_v.v[0], o_v.v[K - 1]};
            *reinterpret_cast<double2*>(out[3] + j0) = double2{o_E.v[0], o_E.v[K - 1]};
            if (a.emit) {
                if (a.emit & 1) *reinterpret_cast<double2*>(a.p_out + row_off + j0) = double2{p.v[0], p.v[K - 1]};
                if (a.emit & 2) *reinterpret_cast<double2*>(a.c_out + row_off + j0) = double2{cs.v[0], cs.v[K - 1]};
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int64_t j = j0 + k;
                if (j >= w0 && j < hi) {
                    out[0][j] = o_rho.v[k];
                    out[1][j] = o_u.v[k];
                    out[2][j] = o_v.v[k];
                    out[3][j] = o_E.v[k];
                    if (a.emit & 1) (a.p_out + row_off)[j] = p.v[k];
                    if (a.emit & 2) (a.c_out + row_off)[j] = cs.v[k];
                }
            }
        }
    }
}

// Vectorised X sweep for even row pitch (16-B aligned column pairs): CH = 16 columns per chunk = one
// 128-B line per row and array, 16-B loads/stores, and the NEXT chunk's 32 loads are issued into registers
// before the march over the current chunk starts, so a single wave per SIMD keeps ~32 KB in flight.
constexpr int kXVecChunk = 16;

template <class PIPE, int CH>
__global__ void __launch_bounds__(kXRows, 1)
k_sweep_x_vec(sweep_args a)
{
    constexpr int LAG = PIPE::LAG;
    constexpr int PITCH = CH + 2;                 // even pitch: 16-B aligned pairs for the row phases
    constexpr int LPR = CH / 2;                   // lanes per row in the row phases (16 B each)
    constexpr int RPI = kXRows / LPR;             // rows per wave-wide access
    constexpr int NI = kXRows / RPI;              // accesses per array and chunk
    constexpr bool VEC_STORE = (LAG % 2) == 0;    // output column pairs are 16-B aligned only for even LAG
    __shared__ double2 tile2[4 * kXRows * PITCH / 2];
    double* tile = reinterpret_cast<double*>(tile2);

    const int lane = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * kXRows;
    const int64_t o0 = (int64_t)blockIdx.x * a.seg;
    const int64_t o1 = (o0 + a.seg < a.nx) ? o0 + a.seg : a.nx;
    const int64_t j_begin = o0 - LAG - ((a.g - LAG) & 1);   // (j_begin + g) even → aligned pairs
    const int64_t j_end = o1 + LAG;
    const bool row_ok = (r0 + lane) < a.ny;
    const int sub_row = lane / LPR, sub_pair = lane % LPR;
    const double* in[4] = {a.rho_in, a.ua_in, a.ut_in, a.E_in};
    double* out[4] = {a.rho_out, a.ua_out, a.ut_out, a.E_out};

    auto T = [&](int plane, int r, int t) -> double& { return tile[(plane * kXRows + r) * PITCH + t]; };

    PIPE pipe(a.dt, a.dx, a.gamma);
    double2 R[4][NI];

    // One element with boundary handling (edge chunks only).
    auto elem = [&](int f, int64_t row, int64_t j) -> double {
        double fa, ft;
        const int64_t jc = j < j_end ? j : j_end - 1;
        const int64_t src = bc_source(a, a.nx, jc, fa, ft);
        const double v = in[f][(row + a.g) * a.row_len + (src + a.g)];
        return f == 1 ? v * fa : (f == 2 ? v * ft : v);
    };
    auto load_chunk = [&](int64_t jb) {
        const int64_t j = jb + 2 * sub_pair;
        const bool interior = jb >= 0 && jb + CH <= a.nx;      // uniform
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int64_t row = r0 + k * RPI + sub_row;
            if (row < a.ny) {
                if (interior) {
                    const int64_t idx = (row + a.g) * a.row_len + (j + a.g);
#pragma unroll
                    for (int f = 0; f < 4; f++) R[f][k] = *reinterpret_cast<const double2*>(in[f] + idx);
                } else {
#pragma unroll
                    for (int f = 0; f < 4; f++) R[f][k] = double2{elem(f, row, j), elem(f, row, j + 1)};
                }
            }
        }
    };

    load_chunk(j_begin);
    for (int64_t jb = j_begin; jb < j_end; jb += CH) {
        // -- registers → LDS tile
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int r = k * RPI + sub_row;
#pragma unroll
            for (int f = 0; f < 4; f++) *reinterpret_cast<double2*>(&T(f, r, 2 * sub_pair)) = R[f][k];
        }
        __syncthreads();
        // -- prefetch the next chunk while this one is marched over
        if (jb + CH < j_end) load_chunk(jb + CH);
        // -- march: lane walks its own row through the tile, results overwrite the consumed slots
        if (row_ok) {
#pragma unroll 1
            for (int t0 = 0; t0 < CH; t0 += 4) {
                static_for(std::make_integer_sequence<int, 4>{}, [&](auto phc) {
                    constexpr int PH = decltype(phc)::value;
                    const int t = t0 + PH;
                    if (jb + t < j_end) {
                        double p, c;
                        const fused::Out4 o = pipe.template push<false, PH>(T(0, lane, t), T(1, lane, t), T(2, lane, t), T(3, lane, t), p, c);
                        T(0, lane, t) = o.rho;
                        T(1, lane, t) = o.ua;
                        T(2, lane, t) = o.ut;
                        T(3, lane, t) = o.E;
                    }
                });
            }
        }
        __syncthreads();
        // -- store phase: slot t holds the new state of column jb + t - LAG
        {
            const int64_t o = jb + 2 * sub_pair - LAG;
            const bool interior = VEC_STORE && (jb - LAG >= o0) && (jb + CH - LAG <= o1);   // uniform
#pragma unroll
            for (int k = 0; k < NI; k++) {
                const int r = k * RPI + sub_row;
                const int64_t row = r0 + r;
                if (row < a.ny) {
                    const int64_t io = (row + a.g) * a.row_len + (o + a.g);
                    if (interior) {
#pragma unroll
                        for (int f = 0; f < 4; f++)
                            *reinterpret_cast<double2*>(out[f] + io) = *reinterpret_cast<const double2*>(&T(f, r, 2 * sub_pair));
                    } else {
#pragma unroll
                        for (int e = 0; e < 2; e++) {
                            const int64_t oe = o + e;
                            if (oe >= o0 && oe < o1 && jb + 2 * sub_pair + e < j_end) {
#pragma unroll
                                for (int f = 0; f < 4; f++) out[f][io + e] = T(f, r, 2 * sub_pair + e);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

constexpr int kXChunk = 8;

constexpr int kXSK = 2;          // cells per lane of the spatial X sweep
constexpr int kXSNiter = 8;      // strips per wave

template <class PIPE>
int launch(armon_ctx* ctx, const sweep_args& a, int axis)
{
    if (axis == ARMON_AXIS_X && a.x_kernel == 0) {
        constexpr int halo = (kXSK == 1) ? PIPE::LAG : ((PIPE::LAG + 1) & ~1);
        constexpr int stride = 64 * kXSK - 2 * halo;
        const int64_t per_block = (int64_t)kXSNiter * stride;
        dim3 grid((unsigned)((a.nx + per_block - 1) / per_block), (unsigned)((a.ny + kXSRows - 1) / kXSRows));
        hipLaunchKernelGGL((k_sweep_x_dpp<PIPE::SCHEME, PIPE::LIM, PIPE::PROJ, PIPE::EOS, PIPE::kExact, kXSK>),
                           grid, dim3(64, kXSRows), 0, ctx->stream, a, kXSNiter);
        return check_launch("sweep_x_dpp");
    }
    if (axis == ARMON_AXIS_Y) {
        dim3 grid((unsigned)((a.nx + kYBlock - 1) / kYBlock), (unsigned)((a.ny + a.seg - 1) / a.seg));
        hipLaunchKernelGGL(k_sweep_y<PIPE>, grid, dim3(kYBlock), 0, ctx->stream, a);
        return check_launch("sweep_y");
    }
    dim3 grid((unsigned)((a.nx + a.seg - 1) / a.seg), (unsigned)((a.ny + kXRows - 1) / kXRows));
    const auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec_ok = a.x_kernel == 1 && !a.emit && (a.row_len % 2 == 0) && aligned16(a.rho_in) && aligned16(a.ua_in) &&
                        aligned16(a.ut_in) && aligned16(a.E_in) && aligned16(a.rho_out) && aligned16(a.ua_out) &&
                        aligned16(a.ut_out) && aligned16(a.E_out);
    if (vec_ok) {
        hipLaunchKernelGGL((k_sweep_x_vec<PIPE, kXVecChunk>), grid, dim3(kXRows), 0, ctx->stream, a);
        return check_launch("sweep_x_vec");
    }
    const size_t lds = (size_t)(a.emit ? 6 : 4) * kXRows * (kXChunk + 1) * sizeof(double);
    hipLaunchKernelGGL((k_sweep_x<PIPE, kXChunk>), grid, dim3(kXRows), lds, ctx->stream, a);
    return check_launch("sweep_x");
}

template <int SCHEME, int LIM, int PROJ, int EOS>
int dispatch_exact(armon_ctx* ctx, const sweep_args& a, int axis, bool exact)
{
    if (exact) return launch<fused::Pipe<SCHEME, LIM, PROJ, EOS>>(ctx, a, axis);
    return launch<fused::PipeFast<SCHEME, LIM, PROJ, EOS>>(ctx, a, axis);
}

template <int SCHEME, int LIM, int PROJ>
int dispatch_eos(armon_ctx* ctx, const sweep_args& a, int axis, int eos, bool exact)
{
    if (eos == ARMON_EOS_BIZARRIUM)
        return dispatch_exact<SCHEME, LIM, PROJ, ARMON_EOS_BIZARRIUM>(ctx, a, axis, exact);
    return dispatch_exact<SCHEME, LIM, PROJ, ARMON_EOS_PERFECT_GAS>(ctx, a, axis, exact);
}

template <int SCHEME, int LIM>
int dispatch_proj(armon_ctx* ctx, const sweep_args& a, int axis, int eos, int proj, bool exact)
{
    if (proj == ARMON_PROJECTION_EULER_2ND)
        return dispatch_eos<SCHEME, LIM, ARMON_PROJECTION_EULER_2ND>(ctx, a, axis, eos, exact);
    return dispatch_eos<SCHEME, LIM, ARMON_PROJECTION_EULER>(ctx, a, axis, eos, exact);
}

}  // namespace

extern "C" int armon_hip_sweep(armon_ctx* ctx, const armon_sweep_desc* d)
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
    const int lag = (d->scheme == ARMON_SCHEME_GAD ? 1 : 0) + (d->projection == ARMON_PROJECTION_EULER_2ND ? 1 : 0) + 2;
    ARMON_REQUIRE(d->nghost >= lag, "nghost = %d but this scheme/projection reads %d cells past the block", d->nghost, lag);
    const int64_t n_axis = d->axis == ARMON_AXIS_X ? d->nx : d->ny;
    ARMON_REQUIRE(!(d->bc_low || d->bc_high) || n_axis >= lag,
                  "mirror boundary needs at least %d cells along the sweep axis", lag);
    ARMON_REQUIRE(d->rho_in && d->u_in && d->v_in && d->E_in && d->rho_out && d->u_out && d->v_out && d->E_out,
                  "NULL state array");
    ARMON_REQUIRE(d->rho_in != d->rho_out && d->u_in != d->u_out && d->v_in != d->v_out && d->E_in != d->E_out,
                  "in and out arrays must not alias (ping-pong)");
    const bool exact = d->exact != 0;

    sweep_args a;
    a.nx = d->nx;
    a.ny = d->ny;
    a.row_len = d->nx + 2 * (int64_t)d->nghost;
    a.g = d->nghost;
    a.bc_low = d->bc_low;
    a.bc_high = d->bc_high;
    a.emit = (d->p_out ? 1 : 0) | (d->c_out ? 2 : 0);
    a.dt = d->dt;
    a.dx = d->dx;
    a.gamma = d->gamma;
    const bool X = d->axis == ARMON_AXIS_X;
    a.fa_low = X ? d->u_factor_low : d->v_factor_low;
    a.ft_low = X ? d->v_factor_low : d->u_factor_low;
    a.fa_high = X ? d->u_factor_high : d->v_factor_high;
    a.ft_high = X ? d->v_factor_high : d->u_factor_high;
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
    a.seg = X ? 512 : 128;
    a.x_kernel = d->reserved;      // tuning/testing knob: 0 = default (spatial)

    if (d->scheme == ARMON_SCHEME_GODUNOV)
        return dispatch_proj<ARMON_SCHEME_GODUNOV, ARMON_LIMITER_NONE>(ctx, a, d->axis, d->eos, d->projection, exact);
    switch (d->limiter) {
    case ARMON_LIMITER_MINMOD:
        return dispatch_proj<ARMON_SCHEME_GAD, ARMON_LIMITER_MINMOD>(ctx, a, d->axis, d->eos, d->projection, exact);
    case ARMON_LIMITER_SUPERBEE:
        return dispatch_proj<ARMON_SCHEME_GAD, ARMON_LIMITER_SUPERBEE>(ctx, a, d->axis, d->eos, d->projection, exact);
    default:
        return dispatch_proj<ARMON_SCHEME_GAD, ARMON_LIMITER_NONE>(ctx, a, d->axis, d->eos, d->projection, exact);
    }
}
