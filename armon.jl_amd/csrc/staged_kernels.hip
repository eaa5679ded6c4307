// staged_kernels.hip — one HIP kernel per reference @generic_kernel (SURVEY §8a rows a1–a10, a13, a16).
// These are the drop-in replacements of the generated "main" functions
// (ref src/generic_kernel.jl:825-921): same arguments, same iteration range, same arrays.
//
// Launch shape: a row of the range is x-contiguous, so blockIdx.x/threadIdx.x walk a row (coalesced
// 512 B per wave per array) and blockIdx.y walks rows — no integer division per cell, unlike the
// reference's 1-D ndrange + divrem (ref src/generic_kernel.jl:784-791).
#include "common.hpp"
#include "physics.hpp"

using namespace armon;

namespace {

__device__ __forceinline__ int64_t row_base(const armon_range& r, int64_t j)
{
    return r.col_start + j * r.col_step + r.row_start;
}

#define ARMON_FOR_RANGE(r, i)                                                         \
    const int64_t k_ = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;                 \
    if (k_ < (r).row_len)                                                              \
        for (int64_t j_ = blockIdx.y, i = row_base((r), j_) + k_; j_ < (r).col_len;    \
             j_ += gridDim.y, i = row_base((r), j_) + k_)

// ---- a1: perfect_gas_EOS! (ref src/kernels.jl:4-13) ------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_perfect_gas_EOS(armon_range r, double gamma, const double* __restrict__ rho,
                  const double* __restrict__ E, const double* __restrict__ u,
                  const double* __restrict__ v, double* __restrict__ p, double* __restrict__ c,
                  double* __restrict__ g)
{
    ARMON_FOR_RANGE(r, i) {
        double pi, ci;
        phys::perfect_gas(gamma, rho[i], E[i], u[i], v[i], pi, ci);
        p[i] = pi;
        c[i] = ci;
        g[i] = (1. + gamma) / 2;
    }
}

// ---- a2: bizarrium_EOS! (ref src/kernels.jl:16-55) -------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_bizarrium_EOS(armon_range r, const double* __restrict__ rho, const double* __restrict__ u,
                const double* __restrict__ v, const double* __restrict__ E, double* __restrict__ p,
                double* __restrict__ c, double* __restrict__ g)
{
    ARMON_FOR_RANGE(r, i) {
        double pi, ci, gi;
        phys::bizarrium<true>(rho[i], E[i], u[i], v[i], pi, ci, gi);
        p[i] = pi;
        c[i] = ci;
        g[i] = gi;
    }
}

// ---- a4: acoustic! (ref src/riemann_schemes.jl:33-43) ----------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_acoustic(armon_range r, int64_t s, double* __restrict__ us, double* __restrict__ ps,
           const double* __restrict__ rho, const double* __restrict__ u,
           const double* __restrict__ p, const double* __restrict__ c)
{
    ARMON_FOR_RANGE(r, i) {
        double a, b;
        phys::godunov(rho[i], rho[i - s], c[i], c[i - s], u[i], u[i - s], p[i], p[i - s], a, b);
        us[i] = a;
        ps[i] = b;
    }
}

// ---- a5: acoustic_GAD! (ref src/riemann_schemes.jl:55-104) -----------------------------------------
template <int LIM>
__global__ void __launch_bounds__(kBlock)
k_acoustic_GAD(armon_range r, int64_t s, double dt, double dx, double* __restrict__ us,
               double* __restrict__ ps, const double* __restrict__ rho, const double* __restrict__ u,
               const double* __restrict__ p, const double* __restrict__ c)
{
    ARMON_FOR_RANGE(r, i) {
        const double rho_mm = rho[i - 2 * s], c_mm = c[i - 2 * s], u_mm = u[i - 2 * s], p_mm = p[i - 2 * s];
        const double rho_m = rho[i - s], c_m = c[i - s], u_m = u[i - s], p_m = p[i - s];
        const double rho_i = rho[i], c_i = c[i], u_i = u[i], p_i = p[i];
        const double rho_p = rho[i + s], c_p = c[i + s], u_p = u[i + s], p_p = p[i + s];
        double us_m, ps_m, us_0, ps_0, us_p, ps_p, a, b;
        phys::godunov(rho_m, rho_mm, c_m, c_mm, u_m, u_mm, p_m, p_mm, us_m, ps_m);
        phys::godunov(rho_i, rho_m, c_i, c_m, u_i, u_m, p_i, p_m, us_0, ps_0);
        phys::godunov(rho_p, rho_i, c_p, c_i, u_p, u_i, p_p, p_i, us_p, ps_p);
        phys::gad_flux<LIM>(dt, dx, rho_m, c_m, u_m, p_m, rho_i, c_i, u_i, p_i,
                            us_m, ps_m, us_0, ps_0, us_p, ps_p, a, b);
        us[i] = a;
        ps[i] = b;
    }
}

// ---- a6: cell_update! (ref src/kernels.jl:58-68) ---------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_cell_update(armon_range r, int64_t s, double dx, double dt, const double* __restrict__ us,
              const double* __restrict__ ps, double* __restrict__ rho, double* __restrict__ ua,
              double* __restrict__ E)
{
    ARMON_FOR_RANGE(r, i) {
        double rho_i = rho[i], ua_i = ua[i], E_i = E[i];
        phys::cell_update(dx, dt, us[i], ps[i], us[i + s], ps[i + s], rho_i, ua_i, E_i);
        rho[i] = rho_i;
        ua[i] = ua_i;
        E[i] = E_i;
    }
}

// ---- a7: advection_first_order! (ref src/projection_schemes.jl:62-78) ------------------------------
__global__ void __launch_bounds__(kBlock)
k_advection_first_order(armon_range r, int64_t s, double dt, const double* __restrict__ us,
                        const double* __restrict__ rho, const double* __restrict__ u,
                        const double* __restrict__ v, const double* __restrict__ E,
                        double* __restrict__ a_rho, double* __restrict__ a_urho,
                        double* __restrict__ a_vrho, double* __restrict__ a_Erho)
{
    ARMON_FOR_RANGE(r, is) {
        double disp = dt * us[is];
        int64_t d = (disp > 0) ? is - s : is;
        double rd = rho[d];
        a_rho[is] = disp * (rd);
        a_urho[is] = disp * (rd * u[d]);
        a_vrho[is] = disp * (rd * v[d]);
        a_Erho[is] = disp * (rd * E[d]);
    }
}

// ---- a8: advection_second_order! (ref src/projection_schemes.jl:92-124) ----------------------------
__global__ void __launch_bounds__(kBlock)
k_advection_second_order(armon_range r, int64_t s, double dx, double dt,
                         const double* __restrict__ us, const double* __restrict__ rho,
                         const double* __restrict__ u, const double* __restrict__ v,
                         const double* __restrict__ E, double* __restrict__ a_rho,
                         double* __restrict__ a_urho, double* __restrict__ a_vrho,
                         double* __restrict__ a_Erho)
{
    ARMON_FOR_RANGE(r, is) {
        double disp = dt * us[is];
        double Dxe;
        int64_t d;
        if (disp > 0) {
            Dxe = -(dx - dt * us[is - s]);
            d = is - s;
        } else {
            Dxe = dx + dt * us[is + s];
            d = is;
        }
        double Dxl_m = dx + dt * (us[d] - us[d - s]);
        double Dxl   = dx + dt * (us[d + s] - us[d]);
        double Dxl_p = dx + dt * (us[d + 2 * s] - us[d + s]);
        double r_m = (2 * Dxl) / (Dxl + Dxl_m);
        double r_p = (2 * Dxl) / (Dxl + Dxl_p);

        double rm = rho[d - s], r0 = rho[d], rp = rho[d + s];
        double sl_rho  = phys::slope_minmod(rm, r0, rp, r_m, r_p);
        double sl_urho = phys::slope_minmod(rm * u[d - s], r0 * u[d], rp * u[d + s], r_m, r_p);
        double sl_vrho = phys::slope_minmod(rm * v[d - s], r0 * v[d], rp * v[d + s], r_m, r_p);
        double sl_Erho = phys::slope_minmod(rm * E[d - s], r0 * E[d], rp * E[d + s], r_m, r_p);

        double length_factor = Dxe / (2 * Dxl);
        a_rho[is]  = disp * (r0        - sl_rho  * length_factor);
        a_urho[is] = disp * (r0 * u[d] - sl_urho * length_factor);
        a_vrho[is] = disp * (r0 * v[d] - sl_vrho * length_factor);
        a_Erho[is] = disp * (r0 * E[d] - sl_Erho * length_factor);
    }
}

// ---- a9: euler_projection! (ref src/projection_schemes.jl:23-41) -----------------------------------
__global__ void __launch_bounds__(kBlock)
k_euler_projection(armon_range r, int64_t s, double dx, double dt, const double* __restrict__ us,
                   double* __restrict__ rho, double* __restrict__ u, double* __restrict__ v,
                   double* __restrict__ E, const double* __restrict__ a_rho,
                   const double* __restrict__ a_urho, const double* __restrict__ a_vrho,
                   const double* __restrict__ a_Erho)
{
    ARMON_FOR_RANGE(r, i) {
        double rho_i = rho[i], u_i = u[i], v_i = v[i], E_i = E[i];
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
__global__ void __launch_bounds__(kBlock)
k_boundary_conditions(armon_range r, int64_t incr, int nghost, double u_factor, double v_factor,
                      double* __restrict__ rho, double* __restrict__ u, double* __restrict__ v,
                      double* __restrict__ p, double* __restrict__ c, double* __restrict__ g,
                      double* __restrict__ E)
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
struct pack_vars { double* v[kMaxPackVars]; };   // passed by value: no device-side pointer table

template <bool PACK>
__global__ void __launch_bounds__(kBlock)
k_pack(armon_range r, int nghost, int64_t face, double* __restrict__ array, int nvars, pack_vars vars)
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

// ---- a16: init_test (ref src/kernels.jl:71-145, src/tests.jl:59-121) -------------------------------
struct two_state { double hi_rho, lo_rho, hi_E, lo_E, hi_u, lo_u, hi_v, lo_v; };

__device__ __forceinline__ bool region_high(int test, double x, double y, double sedov_r)
{
    switch (test) {
    case ARMON_TEST_SOD:       return x <= 0.5;
    case ARMON_TEST_SOD_Y:     return y <= 0.5;
    case ARMON_TEST_SOD_CIRC:  return (x - 0.5) * (x - 0.5) + (y - 0.5) * (y - 0.5) <= 0.09;
    case ARMON_TEST_BIZARRIUM: return x <= 0.5;
    case ARMON_TEST_SEDOV:     return x * x + y * y <= sedov_r * sedov_r;
    default:                   return false;
    }
}

__global__ void __launch_bounds__(kBlock)
k_init_test(armon_range r, int test, int64_t row_length, int64_t nx, int64_t ny, int nghost,
            int64_t gpos_x, int64_t gpos_y, int64_t gN_x, double ox, double oy, double dXx, double dXy,
            double sedov_r, two_state tp, armon_block_data d)
{
    ARMON_FOR_RANGE(r, i) {
        int64_t Iy = i / row_length;
        int64_t Ix = i - Iy * row_length - nghost + 1;
        Iy = Iy - nghost + 1;
        int64_t gx = Ix + gpos_x - 1, gy = Iy + gpos_y - 1;
        double x = (double)gx * dXx + ox;
        double y = (double)gy * dXy + oy;
        d.x[i] = x;
        d.y[i] = y;
        bool ghost = !(Ix >= 1 && Ix <= nx && Iy >= 1 && Iy <= ny);
        d.mask[i] = ghost ? 0. : 1.;
        if (test == ARMON_TEST_DEBUG_INDEXES) {
            double gi = (double)(gx + gy * gN_x + 1);
            d.rho[i] = gi; d.E[i] = gi; d.u[i] = gi; d.v[i] = gi; d.p[i] = gi; d.c[i] = gi; d.g[i] = gi;
        } else {
            bool hi = region_high(test, x + dXx / 2, y + dXy / 2, sedov_r);
            d.rho[i] = hi ? tp.hi_rho : tp.lo_rho;
            d.E[i] = hi ? tp.hi_E : tp.lo_E;
            d.u[i] = hi ? tp.hi_u : tp.lo_u;
            d.v[i] = hi ? tp.hi_v : tp.lo_v;
            d.p[i] = 0.; d.c[i] = 0.; d.g[i] = 0.;
        }
        d.us[i] = 0.; d.ps[i] = 0.;
        d.work_1[i] = 0.; d.work_2[i] = 0.; d.work_3[i] = 0.; d.work_4[i] = 0.;
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

extern "C" {

int armon_hip_perfect_gas_EOS(armon_ctx* ctx, armon_range r, double gamma, const double* rho,
                              const double* E, const double* u, const double* v, double* p,
                              double* c, double* g)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(rho && E && u && v && p && c && g, "NULL array");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_perfect_gas_EOS, grid, block, 0, ctx->stream, r, gamma, rho, E, u, v, p, c, g);
    return check_launch("perfect_gas_EOS");
}

int armon_hip_bizarrium_EOS(armon_ctx* ctx, armon_range r, const double* rho, const double* u,
                            const double* v, const double* E, double* p, double* c, double* g)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(rho && E && u && v && p && c && g, "NULL array");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_bizarrium_EOS, grid, block, 0, ctx->stream, r, rho, u, v, E, p, c, g);
    return check_launch("bizarrium_EOS");
}

int armon_hip_acoustic(armon_ctx* ctx, armon_range r, int64_t s, double* us, double* ps,
                       const double* rho, const double* ua, const double* p, const double* c)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(us && ps && rho && ua && p && c, "NULL array");
    ARMON_REQUIRE(s > 0, "stride must be positive");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_acoustic, grid, block, 0, ctx->stream, r, s, us, ps, rho, ua, p, c);
    return check_launch("acoustic");
}

int armon_hip_acoustic_GAD(armon_ctx* ctx, armon_range r, int64_t s, double dt, double dx,
                           double* us, double* ps, const double* rho, const double* ua,
                           const double* p, const double* c, int limiter)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(us && ps && rho && ua && p && c, "NULL array");
    ARMON_REQUIRE(s > 0, "stride must be positive");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    switch (limiter) {
    case ARMON_LIMITER_NONE:
        hipLaunchKernelGGL(k_acoustic_GAD<ARMON_LIMITER_NONE>, grid, block, 0, ctx->stream, r, s, dt, dx, us, ps, rho, ua, p, c);
        break;
    case ARMON_LIMITER_MINMOD:
        hipLaunchKernelGGL(k_acoustic_GAD<ARMON_LIMITER_MINMOD>, grid, block, 0, ctx->stream, r, s, dt, dx, us, ps, rho, ua, p, c);
        break;
    case ARMON_LIMITER_SUPERBEE:
        hipLaunchKernelGGL(k_acoustic_GAD<ARMON_LIMITER_SUPERBEE>, grid, block, 0, ctx->stream, r, s, dt, dx, us, ps, rho, ua, p, c);
        break;
    default:
        ARMON_REQUIRE(false, "unknown limiter tag %d", limiter);
    }
    return check_launch("acoustic_GAD");
}

int armon_hip_cell_update(armon_ctx* ctx, armon_range r, int64_t s, double dx, double dt,
                          const double* us, const double* ps, double* rho, double* ua, double* E)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(us && ps && rho && ua && E, "NULL array");
    ARMON_REQUIRE(s > 0, "stride must be positive");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_cell_update, grid, block, 0, ctx->stream, r, s, dx, dt, us, ps, rho, ua, E);
    return check_launch("cell_update");
}

int armon_hip_advection_first_order(armon_ctx* ctx, armon_range r, int64_t s, double dt,
                                    const double* us, const double* rho, const double* u,
                                    const double* v, const double* E, double* a_rho,
                                    double* a_urho, double* a_vrho, double* a_Erho)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(us && rho && u && v && E && a_rho && a_urho && a_vrho && a_Erho, "NULL array");
    ARMON_REQUIRE(s > 0, "stride must be positive");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_advection_first_order, grid, block, 0, ctx->stream, r, s, dt, us, rho, u, v, E,
                       a_rho, a_urho, a_vrho, a_Erho);
    return check_launch("advection_first_order");
}

int armon_hip_advection_second_order(armon_ctx* ctx, armon_range r, int64_t s, double dx, double dt,
                                     const double* us, const double* rho, const double* u,
                                     const double* v, const double* E, double* a_rho,
                                     double* a_urho, double* a_vrho, double* a_Erho)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(us && rho && u && v && E && a_rho && a_urho && a_vrho && a_Erho, "NULL array");
    ARMON_REQUIRE(s > 0, "stride must be positive");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_advection_second_order, grid, block, 0, ctx->stream, r, s, dx, dt, us, rho, u,
                       v, E, a_rho, a_urho, a_vrho, a_Erho);
    return check_launch("advection_second_order");
}

int armon_hip_euler_projection(armon_ctx* ctx, armon_range r, int64_t s, double dx, double dt,
                               const double* us, double* rho, double* u, double* v, double* E,
                               const double* a_rho, const double* a_urho, const double* a_vrho,
                               const double* a_Erho)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(us && rho && u && v && E && a_rho && a_urho && a_vrho && a_Erho, "NULL array");
    ARMON_REQUIRE(s > 0, "stride must be positive");
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_euler_projection, grid, block, 0, ctx->stream, r, s, dx, dt, us, rho, u, v, E,
                       a_rho, a_urho, a_vrho, a_Erho);
    return check_launch("euler_projection");
}

int armon_hip_boundary_conditions(armon_ctx* ctx, armon_range r, int64_t incr, int nghost,
                                  double u_factor, double v_factor, double* rho, double* u,
                                  double* v, double* p, double* c, double* g, double* E)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(rho && u && v && p && c && g && E, "NULL array");
    ARMON_REQUIRE(incr != 0 && nghost > 0, "invalid incr/nghost");
    dim3 grid, block;
    linear_grid(ctx, r.col_len * r.row_len, grid, block);
    hipLaunchKernelGGL(k_boundary_conditions, grid, block, 0, ctx->stream, r, incr, nghost, u_factor,
                       v_factor, rho, u, v, p, c, g, E);
    return check_launch("boundary_conditions");
}

static int pack_common(armon_ctx* ctx, armon_range r, int nghost, int64_t face, double* array,
                       int nvars, const double* const* vars, bool pack)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(array && vars, "NULL array");
    ARMON_REQUIRE(nvars > 0 && nvars <= kMaxPackVars, "nvars must be in [1,%d]", kMaxPackVars);
    ARMON_REQUIRE(nghost > 0 && face > 0, "invalid nghost/face");
    ARMON_REQUIRE(r.col_len * r.row_len <= face * nghost, "range larger than face*nghost");
    pack_vars table = {};
    for (int v = 0; v < nvars; v++) {
        ARMON_REQUIRE(vars[v] != nullptr, "NULL array in vars[%d]", v);
        table.v[v] = const_cast<double*>(vars[v]);
    }
    dim3 grid, block;
    linear_grid(ctx, r.col_len * r.row_len, grid, block);
    if (pack)
        hipLaunchKernelGGL(k_pack<true>, grid, block, 0, ctx->stream, r, nghost, face, array, nvars, table);
    else
        hipLaunchKernelGGL(k_pack<false>, grid, block, 0, ctx->stream, r, nghost, face, array, nvars, table);
    return check_launch(pack ? "pack_to_array" : "unpack_from_array");
}

int armon_hip_pack_to_array(armon_ctx* ctx, armon_range r, int nghost, int64_t face, double* array,
                            int nvars, const double* const* vars)
{
    return pack_common(ctx, r, nghost, face, array, nvars, vars, true);
}

int armon_hip_unpack_from_array(armon_ctx* ctx, armon_range r, int nghost, int64_t face,
                                const double* array, int nvars, double* const* vars)
{
    return pack_common(ctx, r, nghost, face, const_cast<double*>(array), nvars,
                       const_cast<const double* const*>(vars), false);
}

int armon_hip_init_test(armon_ctx* ctx, armon_range r, int test, int64_t row_length,
                        int64_t col_length, int nghost, const int64_t global_pos[2],
                        const int64_t global_N[2], const double origin[2], const double dX[2],
                        double sedov_r, const armon_block_data* d)
{
    ARMON_CHECK_CTX_RANGE(ctx, r);
    ARMON_REQUIRE(global_pos && global_N && origin && dX && d, "NULL argument");
    ARMON_REQUIRE(test >= ARMON_TEST_SOD && test <= ARMON_TEST_DEBUG_INDEXES, "unknown test tag %d", test);
    const double* const* arrs = reinterpret_cast<const double* const*>(d);
    for (int k = 0; k < 16; k++) ARMON_REQUIRE(arrs[k] != nullptr, "NULL array in block data (field %d)", k);
    // ref src/tests.jl:84-121
    two_state tp;
    switch (test) {
    case ARMON_TEST_BIZARRIUM:
        tp = { 1.42857142857e+4, 10000., 4.48657821135e+6, 0.5 * (250. * 250.), 0., 250., 0., 0. };
        break;
    case ARMON_TEST_SEDOV:
        tp = { 1., 1., pow(1. / 1.033, 5) / (M_PI * (sedov_r * sedov_r)), 2.5e-14, 0., 0., 0., 0. };
        break;
    default:
        tp = { 1., 0.125, 2.5, 2.0, 0., 0., 0., 0. };
    }
    dim3 grid, block;
    range_grid(r, 1, grid, block);
    hipLaunchKernelGGL(k_init_test, grid, block, 0, ctx->stream, r, test, row_length,
                       row_length - 2 * nghost, col_length - 2 * nghost, nghost, global_pos[0],
                       global_pos[1], global_N[0], origin[0], origin[1], dX[0], dX[1], sedov_r, tp, *d);
    return check_launch("init_test");
}

}  // extern "C"
