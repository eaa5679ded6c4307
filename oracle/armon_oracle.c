/*
 * armon_oracle.c — CPU ORACLE (test infrastructure, NOT product code). See armon_oracle.h.
 *
 * Straight fp64 restatement of the reference's per-cell kernels and synchronous solver cycle.
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Compile with -ffp-contract=off so every operation is a single IEEE-754 op (the HIP "exact"
 * kernels are bit-identical to this file); OpenMP only distributes rows over threads, mirroring the
 * reference CPU path `@threaded for j in range.col` + inner SIMD loop
 * (ref src/generic_kernel.jl:150-193).
 */
#include "armon_oracle.h"

#include <math.h>
#include <stddef.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ARMON_ORACLE_F32
#define SQRT sqrtf
#define FABS fabsf
#define HYPOT hypotf
#define POW powf
#define R(x) x##f            /* literal in the working precision */
#else
#define SQRT sqrt
#define FABS fabs
#define HYPOT hypot
#define POW pow
#define R(x) x
#endif

static int g_threads = 1;

void armon_oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int  armon_oracle_get_threads(void) { return g_threads; }

/* Julia's @fastmath max/min: max_fast(x,y) = ifelse(y > x, y, x); min_fast(x,y) = ifelse(y > x, x, y)
 * (kernels are compiled with @fastmath: ref src/generic_kernel.jl:32-36,477-479). */
static inline real mx(real x, real y) { return (y > x) ? y : x; }
static inline real mn(real x, real y) { return (y > x) ? x : y; }

#define ROWS_BEGIN(r)                                                                           \
    _Pragma("omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)")     \
    for (int64_t j_ = 0; j_ < (r).col_len; j_++) {                                              \
        const int64_t base_ = (r).col_start + j_ * (r).col_step + (r).row_start;                \
        _Pragma("omp simd")                                                                     \
        for (int64_t i_ = 0; i_ < (r).row_len; i_++) {                                          \
            const int64_t i = base_ + i_;
#define ROWS_END }}

/* ref src/kernels.jl:4-13 */
void armon_oracle_perfect_gas_EOS(armon_range r, real gamma,
        const real* rho, const real* E, const real* u, const real* v,
        real* p, real* c, real* g)
{
    ROWS_BEGIN(r)
        real e = E[i] - R(0.5) * (u[i] * u[i] + v[i] * v[i]);
        p[i] = (gamma - R(1.)) * rho[i] * e;
        c[i] = SQRT(gamma * p[i] / rho[i]);
        g[i] = (R(1.) + gamma) / 2;
    ROWS_END
}

/* ref src/kernels.jl:16-55 */
void armon_oracle_bizarrium_EOS(armon_range r,
        const real* rho_, const real* u, const real* v, const real* E,
        real* p, real* c, real* g)
{
    const real rho0 = 10000., K0 = 1e+11, Cv0 = 1000., T0 = 300., eps0 = 0., G0 = 1.5, s = 1.5;
    const real q = (real)(-42080895. / 14941154.), rr = (real)(727668333. / 149411540.);
    ROWS_BEGIN(r)
        real rho = rho_[i];
        real x = rho / rho0 - 1;
        real G = G0 * (1 - rho0 / rho);
        real x2 = x * x, x3 = x * x * x;
        real opx = 1 + x, opx2 = opx * opx, opx3 = opx * opx * opx, opx4 = opx2 * opx2;

        real f0 = (1 + (s / 3 - 2) * x + q * x2 + rr * x3) / (1 - s * x);
        real f1 = (s / 3 - 2 + 2 * q * x + 3 * rr * x2 + s * f0) / (1 - s * x);
        real f2 = (2 * q + 6 * rr * x + 2 * s * f1) / (1 - s * x);
        real f3 = (6 * rr + 3 * s * f2) / (1 - s * x);

        real epsk0 = eps0 - Cv0 * T0 * (1 + G) + R(0.5) * (K0 / rho0) * x2 * f0;
        real pk0 = -Cv0 * T0 * G0 * rho0 + R(0.5) * K0 * x * opx2 * (2 * f0 + x * f1);
        real pk0prime = -R(0.5) * K0 * opx3 * rho0 *
                          (2 * (1 + 3 * x) * f0 + 2 * x * (2 + 3 * x) * f1 + x2 * opx * f2);
        real pk0second = R(0.5) * K0 * opx4 * (rho0 * rho0) *
                           (12 * (1 + 2 * x) * f0 + 6 * (1 + 6 * x + 6 * x2) * f1 +
                            6 * x * opx * (1 + 2 * x) * f2 + x2 * opx2 * f3);

        real e = E[i] - R(0.5) * (u[i] * u[i] + v[i] * v[i]);
        real pi = pk0 + G0 * rho0 * (e - epsk0);
        real ci = SQRT(G0 * rho0 * (pi - pk0) - pk0prime) / rho;
        p[i] = pi;
        c[i] = ci;
        g[i] = R(0.5) / (rho * rho * rho * (ci * ci)) * (pk0second + (G0 * rho0) * (G0 * rho0) * (pi - pk0));
    ROWS_END
}

/* ref src/riemann_schemes.jl:21-30 — interface between left cell (i-s) and right cell (i) */
static inline void godunov(real rho_i, real rho_im, real c_i, real c_im,
                           real u_i, real u_im, real p_i, real p_im,
                           real* us, real* ps)
{
    real rc_l = rho_im * c_im;
    real rc_r = rho_i * c_i;
    *us = (rc_l * u_im + rc_r * u_i + (p_im - p_i)) / (rc_l + rc_r);
    *ps = (rc_r * p_im + rc_l * p_i + rc_l * rc_r * (u_im - u_i)) / (rc_l + rc_r);
}

/* ref src/riemann_schemes.jl:33-43 */
void armon_oracle_acoustic(armon_range r, int64_t s, real* us, real* ps,
        const real* rho, const real* u, const real* p, const real* c)
{
    ROWS_BEGIN(r)
        real a, b;
        godunov(rho[i], rho[i - s], c[i], c[i - s], u[i], u[i - s], p[i], p[i - s], &a, &b);
        us[i] = a;
        ps[i] = b;
    ROWS_END
}

/* ref src/limiters.jl:6-8 */
static inline real limiter(real r, int lim)
{
    switch (lim) {
    case ARMON_LIMITER_MINMOD:   return mx(0., mn(1., r));
    case ARMON_LIMITER_SUPERBEE: return mx(mx(0., mn(R(2.) * r, 1.)), mn(r, 2.));
    default:                     return 1.;
    }
}

/* ref src/riemann_schemes.jl:55-104 */
void armon_oracle_acoustic_GAD(armon_range r, int64_t s, real dt, real dx,
        real* us, real* ps,
        const real* rho, const real* u, const real* p, const real* c, int lim)
{
    ROWS_BEGIN(r)
        real us_m, ps_m, us_0, ps_0, us_p, ps_p;
        godunov(rho[i - s], rho[i - 2 * s], c[i - s], c[i - 2 * s],
                u[i - s], u[i - 2 * s], p[i - s], p[i - 2 * s], &us_m, &ps_m);
        godunov(rho[i], rho[i - s], c[i], c[i - s], u[i], u[i - s], p[i], p[i - s], &us_0, &ps_0);
        godunov(rho[i + s], rho[i], c[i + s], c[i], u[i + s], u[i], p[i + s], p[i], &us_p, &ps_p);

        real r_um = (us_p - u[i]) / (us_0 - u[i - s] + R(1e-6));
        real r_pm = (ps_p - p[i]) / (ps_0 - p[i - s] + R(1e-6));
        real r_up = (u[i - s] - us_m) / (u[i] - us_0 + R(1e-6));
        real r_pp = (p[i - s] - ps_m) / (p[i] - ps_0 + R(1e-6));

        r_um = limiter(r_um, lim);
        r_pm = limiter(r_pm, lim);
        r_up = limiter(r_up, lim);
        r_pp = limiter(r_pp, lim);

        real dm_l = rho[i - s] * dx;
        real dm_r = rho[i] * dx;
        real Dm = (dm_l + dm_r) / 2;

        real rc_l = rho[i - s] * c[i - s];
        real rc_r = rho[i] * c[i];
        real theta = R(0.5) * (1 - (rc_l + rc_r) / 2 * (dt / Dm));

        us[i] = us_0 + theta * (r_up * (u[i] - us_0) - r_um * (us_0 - u[i - s]));
        ps[i] = ps_0 + theta * (r_pp * (p[i] - ps_0) - r_pm * (ps_0 - p[i - s]));
    ROWS_END
}

/* ref src/kernels.jl:58-68 */
void armon_oracle_cell_update(armon_range r, int64_t s, real dx, real dt,
        const real* us, const real* ps, real* rho, real* ua, real* E)
{
    ROWS_BEGIN(r)
        real dm = rho[i] * dx;
        rho[i] = dm / (dx + dt * (us[i + s] - us[i]));
        ua[i] += dt / dm * (ps[i] - ps[i + s]);
        E[i] += dt / dm * (ps[i] * us[i] - ps[i + s] * us[i + s]);
    ROWS_END
}

/* ref src/projection_schemes.jl:62-78 */
void armon_oracle_advection_first_order(armon_range r, int64_t s, real dt,
        const real* us, const real* rho, const real* u, const real* v, const real* E,
        real* adv_rho, real* adv_urho, real* adv_vrho, real* adv_Erho)
{
    ROWS_BEGIN(r)
        int64_t is = i, d = i;
        real disp = dt * us[is];
        if (disp > 0) d = is - s;
        adv_rho[is]  = disp * (rho[d]);
        adv_urho[is] = disp * (rho[d] * u[d]);
        adv_vrho[is] = disp * (rho[d] * v[d]);
        adv_Erho[is] = disp * (rho[d] * E[d]);
    ROWS_END
}

/* ref src/projection_schemes.jl:15-20 */
static inline real slope_minmod(real um, real u0, real up, real r_m, real r_p)
{
    real Dp = r_p * (up - u0);
    real Dm = r_m * (u0 - um);
    real sg = (Dp > 0) ? 1. : ((Dp < 0) ? -1. : Dp);   /* Julia sign() */
    return sg * mx(0., mn(sg * Dp, sg * Dm));
}

/* ref src/projection_schemes.jl:92-124 */
void armon_oracle_advection_second_order(armon_range r, int64_t s, real dx, real dt,
        const real* us, const real* rho, const real* u, const real* v, const real* E,
        real* adv_rho, real* adv_urho, real* adv_vrho, real* adv_Erho)
{
    ROWS_BEGIN(r)
        int64_t is = i, d = i;
        real disp = dt * us[is];
        real Dxe;
        if (disp > 0) {
            Dxe = -(dx - dt * us[is - s]);
            d = is - s;
        } else {
            Dxe = dx + dt * us[is + s];
        }

        real Dxl_m = dx + dt * (us[d] - us[d - s]);
        real Dxl   = dx + dt * (us[d + s] - us[d]);
        real Dxl_p = dx + dt * (us[d + 2 * s] - us[d + s]);

        real r_m = (2 * Dxl) / (Dxl + Dxl_m);
        real r_p = (2 * Dxl) / (Dxl + Dxl_p);

        real sl_rho  = slope_minmod(rho[d - s], rho[d], rho[d + s], r_m, r_p);
        real sl_urho = slope_minmod(rho[d - s] * u[d - s], rho[d] * u[d], rho[d + s] * u[d + s], r_m, r_p);
        real sl_vrho = slope_minmod(rho[d - s] * v[d - s], rho[d] * v[d], rho[d + s] * v[d + s], r_m, r_p);
        real sl_Erho = slope_minmod(rho[d - s] * E[d - s], rho[d] * E[d], rho[d + s] * E[d + s], r_m, r_p);

        real length_factor = Dxe / (2 * Dxl);
        adv_rho[is]  = disp * (rho[d]        - sl_rho  * length_factor);
        adv_urho[is] = disp * (rho[d] * u[d] - sl_urho * length_factor);
        adv_vrho[is] = disp * (rho[d] * v[d] - sl_vrho * length_factor);
        adv_Erho[is] = disp * (rho[d] * E[d] - sl_Erho * length_factor);
    ROWS_END
}

/* ref src/projection_schemes.jl:23-41 */
void armon_oracle_euler_projection(armon_range r, int64_t s, real dx, real dt,
        const real* us, real* rho, real* u, real* v, real* E,
        const real* adv_rho, const real* adv_urho, const real* adv_vrho, const real* adv_Erho)
{
    ROWS_BEGIN(r)
        real dX = dx + dt * (us[i + s] - us[i]);
        real t_rho  = (dX * rho[i]        - (adv_rho[i + s]  - adv_rho[i]))  / dx;
        real t_urho = (dX * rho[i] * u[i] - (adv_urho[i + s] - adv_urho[i])) / dx;
        real t_vrho = (dX * rho[i] * v[i] - (adv_vrho[i + s] - adv_vrho[i])) / dx;
        real t_Erho = (dX * rho[i] * E[i] - (adv_Erho[i + s] - adv_Erho[i])) / dx;
        rho[i] = t_rho;
        u[i] = t_urho / t_rho;
        v[i] = t_vrho / t_rho;
        E[i] = t_Erho / t_rho;
    ROWS_END
}

/* ref src/halo_exchange.jl:2-29 */
void armon_oracle_boundary_conditions(armon_range r, int64_t incr, int nghost,
        real u_factor, real v_factor,
        real* rho, real* u, real* v, real* p, real* c, real* g, real* E)
{
    for (int64_t j = 0; j < r.col_len; j++) {
        for (int64_t k = 0; k < r.row_len; k++) {
            int64_t i = r.col_start + j * r.col_step + r.row_start + k;
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
}

/* ref src/halo_exchange.jl:187-200; iteration index: ref src/generic_kernel.jl:784-791 */
void armon_oracle_pack_to_array(armon_range r, int nghost, int64_t face,
        real* array, int nvars, const real* const* vars)
{
    for (int64_t j = 0; j < r.col_len; j++) {
        for (int64_t k = 0; k < r.row_len; k++) {
            int64_t idx = r.col_start + j * r.col_step + r.row_start + k;
            int64_t itr = j * r.row_len + k;           /* 0-based @iter_idx */
            int64_t i = itr / nghost, i_g = itr % nghost;
            int64_t i_arr = (i_g * face + i) * nvars;
            for (int v = 0; v < nvars; v++) array[i_arr + v] = vars[v][idx];
        }
    }
}

/* ref src/halo_exchange.jl:203-216 */
void armon_oracle_unpack_from_array(armon_range r, int nghost, int64_t face,
        const real* array, int nvars, real* const* vars)
{
    for (int64_t j = 0; j < r.col_len; j++) {
        for (int64_t k = 0; k < r.row_len; k++) {
            int64_t idx = r.col_start + j * r.col_step + r.row_start + k;
            int64_t itr = j * r.row_len + k;
            int64_t i = itr / nghost, i_g = itr % nghost;
            int64_t i_arr = (i_g * face + i) * nvars;
            for (int v = 0; v < nvars; v++) vars[v][idx] = array[i_arr + v];
        }
    }
}

/* ref src/reductions.jl:13-53 (mask-less CPU form over the real domain) */
real armon_oracle_dtCFL(armon_range r, real dx, real dy,
        const real* u, const real* v, const real* c)
{
    real res = (real)INFINITY;
    #pragma omp parallel for schedule(static) num_threads(g_threads) reduction(min : res) if (g_threads > 1)
    for (int64_t j = 0; j < r.col_len; j++) {
        const int64_t base = r.col_start + j * r.col_step + r.row_start;
        real row_res = (real)INFINITY;
        for (int64_t k = 0; k < r.row_len; k++) {
            int64_t i = base + k;
            real ax = FABS(mx(FABS(u[i] + c[i]), FABS(u[i] - c[i])));
            real ay = FABS(mx(FABS(v[i] + c[i]), FABS(v[i] - c[i])));
            real cell = mn(dx / ax, dy / ay);
            row_res = mn(row_res, cell);
        }
        res = mn(res, row_res);
    }
    return res;
}

/* ref src/reductions.jl:211-259 */
void armon_oracle_conservation_vars(armon_range r, real ds,
        const real* rho, const real* E, real out[2])
{
    real mass = 0., energy = 0.;
    for (int64_t j = 0; j < r.col_len; j++) {
        const int64_t base = r.col_start + j * r.col_step + r.row_start;
        for (int64_t k = 0; k < r.row_len; k++) {
            int64_t i = base + k;
            mass += rho[i];
            energy += rho[i] * E[i];
        }
    }
    out[0] = mass * ds;
    out[1] = energy * ds;
}

/* ref src/tests.jl:59-63 (on the cell centre) */
static inline int region_high(int test, real x, real y, real sedov_r)
{
    switch (test) {
    case ARMON_TEST_SOD:       return x <= R(0.5);
    case ARMON_TEST_SOD_Y:     return y <= R(0.5);
    case ARMON_TEST_SOD_CIRC:  return (x - R(0.5)) * (x - R(0.5)) + (y - R(0.5)) * (y - R(0.5)) <= R(0.09);
    case ARMON_TEST_BIZARRIUM: return x <= R(0.5);
    case ARMON_TEST_SEDOV:     return x * x + y * y <= sedov_r * sedov_r;
    default:                   return 0;
    }
}

/* ref src/tests.jl:84-121: (high_ρ, low_ρ, high_E, low_E, high_u, low_u, high_v, low_v) */
void armon_oracle_two_state_params(int test, real sedov_r, real out[8])
{
    switch (test) {
    case ARMON_TEST_BIZARRIUM:
        out[0] = 1.42857142857e+4; out[1] = 10000.;
        out[2] = 4.48657821135e+6; out[3] = 0.5 * (250. * 250.);
        out[4] = 0.; out[5] = 250.; out[6] = 0.; out[7] = 0.;
        break;
    case ARMON_TEST_SEDOV:
        out[0] = 1.; out[1] = 1.;
        /* ref src/tests.jl:112: T((1/1.033)^5 / (π * p.r^2)) — π·r² in T, the quotient in Float64 */
        out[2] = (real)(pow(1. / 1.033, 5) / (double)((real)M_PI * (sedov_r * sedov_r))); out[3] = 2.5e-14;
        out[4] = out[5] = out[6] = out[7] = 0.;
        break;
    default: /* Sod family */
        out[0] = 1.; out[1] = 0.125; out[2] = 2.5; out[3] = 2.0;
        out[4] = out[5] = out[6] = out[7] = 0.;
    }
}

/* ref src/kernels.jl:71-145 */
void armon_oracle_init_test(armon_range r, int test, int64_t row_length, int64_t col_length,
        int nghost, const int64_t global_pos[2], const int64_t global_N[2],
        const real origin[2], const real dX[2], real sedov_r, const armon_oracle_block_data* d)
{
    real tp[8];
    armon_oracle_two_state_params(test, sedov_r, tp);
    const int64_t nx = row_length - 2 * nghost, ny = col_length - 2 * nghost;
    for (int64_t j = 0; j < r.col_len; j++) {
        for (int64_t k = 0; k < r.row_len; k++) {
            int64_t i = r.col_start + j * r.col_step + r.row_start + k;
            /* position(bsize, i), 1-based real-cell coordinates: ref src/blocking/blocking.jl:99-104 */
            int64_t Ix = i % row_length - nghost + 1;
            int64_t Iy = i / row_length - nghost + 1;
            /* 0-indexed global position: ref src/kernels.jl:122 (global_pos already 0-based here) */
            int64_t gx = Ix + global_pos[0] - 1;
            int64_t gy = Iy + global_pos[1] - 1;
            d->x[i] = (real)gx * dX[0] + origin[0];
            d->y[i] = (real)gy * dX[1] + origin[1];
            int ghost = !(Ix >= 1 && Ix <= nx && Iy >= 1 && Iy <= ny);
            d->mask[i] = ghost ? 0. : 1.;
            real mx_ = d->x[i] + dX[0] / 2, my_ = d->y[i] + dX[1] / 2;
            if (test == ARMON_TEST_DEBUG_INDEXES) {
                /* ref src/kernels.jl:93-103,135-137: global linear index (1-based) */
                real gi = (real)(gx + gy * global_N[0] + 1);
                d->rho[i] = d->E[i] = d->u[i] = d->v[i] = d->p[i] = d->c[i] = d->g[i] = gi;
            } else {
                int hi = region_high(test, mx_, my_, sedov_r);
                d->rho[i] = hi ? tp[0] : tp[1];
                d->E[i]   = hi ? tp[2] : tp[3];
                d->u[i]   = hi ? tp[4] : tp[5];
                d->v[i]   = hi ? tp[6] : tp[7];
                d->p[i] = d->c[i] = d->g[i] = 0.;
            }
            d->us[i] = d->ps[i] = 0.;
            d->work_1[i] = d->work_2[i] = d->work_3[i] = d->work_4[i] = 0.;
        }
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* Solver: ref src/solver.jl:288-403, src/reductions.jl:164-199, src/solver_state.jl:102-166       */

/* block_domain_range(bsize, bottom_left, top_right): ref src/blocking/blocking.jl:71-85 (0-based) */
static armon_range domain_range(int64_t nx, int64_t ny, int g,
                                int64_t blx, int64_t bly, int64_t trx, int64_t try_)
{
    const int64_t row = nx + 2 * g;
    armon_range r;
    int64_t fx = blx + 1, fy = bly + 1, lx = trx + nx, ly = try_ + ny;   /* 1-based real coords */
    r.col_start = (fy + g - 1) * row;
    r.col_step = row;
    r.col_len = ly - fy + 1;
    r.row_start = fx + g - 1;
    r.row_len = lx - fx + 1;
    return r;
}

static double now_seconds(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ref src/tests.jl:150-211: Dirichlet flag per side L,R,B,T */
static void bc_dirichlet(int test, int out[4])
{
    switch (test) {
    case ARMON_TEST_SOD:       out[0] = 1; out[1] = 1; out[2] = 0; out[3] = 0; break;
    case ARMON_TEST_SOD_Y:     out[0] = 0; out[1] = 0; out[2] = 1; out[3] = 1; break;
    case ARMON_TEST_SOD_CIRC:  out[0] = 1; out[1] = 1; out[2] = 1; out[3] = 1; break;
    case ARMON_TEST_BIZARRIUM: out[0] = 1; out[1] = 0; out[2] = 1; out[3] = 1; break;
    case ARMON_TEST_SEDOV:     out[0] = 0; out[1] = 0; out[2] = 0; out[3] = 0; break;
    default:                   out[0] = 1; out[1] = 1; out[2] = 1; out[3] = 1;
    }
}

static void update_eos(const armon_oracle_run* run, const armon_oracle_block_data* d)
{
    armon_range r = domain_range(run->nx, run->ny, run->nghost, 0, 0, 0, 0);   /* EOS: real cells */
    if (run->test == ARMON_TEST_BIZARRIUM)
        armon_oracle_bizarrium_EOS(r, d->rho, d->u, d->v, d->E, d->p, d->c, d->g);
    else
        armon_oracle_perfect_gas_EOS(r, 7. / 5., d->rho, d->E, d->u, d->v, d->p, d->c, d->g);
}

/* TEST AID (no reference counterpart): periodic ghosts along `axis` — layer l outside the low side holds the cell l in from
 * the high side, and vice versa, for the 7 variables the boundary conditions write (ref src/halo_exchange.jl:2-29). */
static void periodic_ghosts(const armon_oracle_run* run, const armon_oracle_block_data* d, int axis)
{
    const int64_t nx = run->nx, ny = run->ny, g = run->nghost, row = nx + 2 * g;
    real* vars[7] = {d->rho, d->u, d->v, d->p, d->c, d->g, d->E};
    const int64_t n = (axis == ARMON_AXIS_X) ? nx : ny, m = (axis == ARMON_AXIS_X) ? ny : nx;
    const int64_t s = (axis == ARMON_AXIS_X) ? 1 : row, t = (axis == ARMON_AXIS_X) ? row : 1;
    const int64_t first = g * row + g;                       /* first real cell */
    for (int64_t k = 0; k < m; k++)
        for (int64_t l = 0; l < g; l++)
            for (int v = 0; v < 7; v++) {
                vars[v][first + k * t + (-1 - l) * s] = vars[v][first + k * t + (n - 1 - l) * s];
                vars[v][first + k * t + (n + l) * s] = vars[v][first + k * t + l * s];
            }
}

/* one directional sweep: ref src/solver.jl:300-316; ranges ref src/parameters.jl:988-1025 */
static void sweep(const armon_oracle_run* run, const armon_oracle_block_data* d, int axis, real dt)
{
    const int64_t nx = run->nx, ny = run->ny;
    const int g = run->nghost;
    const int64_t row = nx + 2 * g;
    const int64_t s = (axis == ARMON_AXIS_X) ? 1 : row;
    const real dx = run->domain_size[axis] / (real)(axis == ARMON_AXIS_X ? nx : ny);
    const int w = (run->projection == ARMON_PROJECTION_EULER_2ND) ? 2 : 1;
    real* ua = (axis == ARMON_AXIS_X) ? d->u : d->v;
    int dirichlet[4];
    bc_dirichlet(run->test, dirichlet);

    update_eos(run, d);

    /* BC on both sides of the sweep axis: ref src/halo_exchange.jl:32-36,323-354 */
    if (run->periodic[axis]) periodic_ghosts(run, d, axis);
    else for (int hs = 0; hs < 2; hs++) {
        int side = (axis == ARMON_AXIS_X) ? (hs ? ARMON_SIDE_RIGHT : ARMON_SIDE_LEFT)
                                          : (hs ? ARMON_SIDE_TOP : ARMON_SIDE_BOTTOM);
        real uf = 1., vf = 1.;
        if (dirichlet[side]) { if (axis == ARMON_AXIS_X) uf = -1.; else vf = -1.; }
        armon_range br;   /* border_domain(bsize, side): ref src/blocking/blocking.jl:141-165 */
        switch (side) {
        case ARMON_SIDE_LEFT:   br = domain_range(nx, ny, g, 0, 0, 1 - nx, 0); break;
        case ARMON_SIDE_RIGHT:  br = domain_range(nx, ny, g, nx - 1, 0, 0, 0); break;
        case ARMON_SIDE_BOTTOM: br = domain_range(nx, ny, g, 0, 0, 0, 1 - ny); break;
        default:                br = domain_range(nx, ny, g, 0, ny - 1, 0, 0);
        }
        int64_t incr = hs ? s : -s;
        armon_oracle_boundary_conditions(br, incr, g, uf, vf, d->rho, d->u, d->v, d->p, d->c, d->g, d->E);
    }

    armon_range fl, cu, ad, pr;
    if (axis == ARMON_AXIS_X) {
        fl = domain_range(nx, ny, g, -w, 0, w + 1, 0);
        cu = domain_range(nx, ny, g, -w, 0, w, 0);
        ad = domain_range(nx, ny, g, 0, 0, 1, 0);
    } else {
        fl = domain_range(nx, ny, g, 0, -w, 0, w + 1);
        cu = domain_range(nx, ny, g, 0, -w, 0, w);
        ad = domain_range(nx, ny, g, 0, 0, 0, 1);
    }
    pr = domain_range(nx, ny, g, 0, 0, 0, 0);

    if (run->scheme == ARMON_SCHEME_GAD)
        armon_oracle_acoustic_GAD(fl, s, dt, dx, d->us, d->ps, d->rho, ua, d->p, d->c, run->limiter);
    else
        armon_oracle_acoustic(fl, s, d->us, d->ps, d->rho, ua, d->p, d->c);

    armon_oracle_cell_update(cu, s, dx, dt, d->us, d->ps, d->rho, ua, d->E);

    if (run->projection == ARMON_PROJECTION_EULER_2ND)
        armon_oracle_advection_second_order(ad, s, dx, dt, d->us, d->rho, d->u, d->v, d->E,
                                            d->work_1, d->work_2, d->work_3, d->work_4);
    else
        armon_oracle_advection_first_order(ad, s, dt, d->us, d->rho, d->u, d->v, d->E,
                                           d->work_1, d->work_2, d->work_3, d->work_4);

    armon_oracle_euler_projection(pr, s, dx, dt, d->us, d->rho, d->u, d->v, d->E,
                                  d->work_1, d->work_2, d->work_3, d->work_4);
}

int armon_oracle_solve(armon_oracle_run* run, const armon_oracle_block_data* d, int skip_init)
{
    const int64_t nx = run->nx, ny = run->ny;
    const int g = run->nghost;
    const real dX[2] = { run->domain_size[0] / (real)nx, run->domain_size[1] / (real)ny };
    armon_range real_r = domain_range(nx, ny, g, 0, 0, 0, 0);

    if (!skip_init) {
        /* ref src/tests.jl:15-19 */
        real sedov_r = (real)(HYPOT(dX[0], dX[1]) / sqrt(2.));
        armon_range full = domain_range(nx, ny, g, -g, -g, g, g);
        int64_t gpos[2] = { 0, 0 }, gN[2] = { nx, ny };
        armon_oracle_init_test(full, run->test, nx + 2 * g, ny + 2 * g, g, gpos, gN,
                               run->origin, dX, sedov_r, d);
    }

    real cons[2];
    armon_oracle_conservation_vars(real_r, dX[0] * dX[1], d->rho, d->E, cons);
    run->initial_mass = cons[0];
    run->initial_energy = cons[1];

    /* GlobalTimeStep: ref src/solver_state.jl:58-68 */
    int64_t cycle = 0;
    real time = 0., current_dt = run->cst_dt ? run->Dt : 0., next_cycle_dt = (real)INFINITY;
    run->status = 0;

    double t0 = now_seconds();
    while (time < run->maxtime && cycle < run->maxcycle) {           /* ref src/solver.jl:333 */
        if (cycle == 0) update_eos(run, d);                          /* ref src/solver.jl:291-295 */

        /* next_time_step: ref src/reductions.jl:164-199; update_dt!: ref src/solver_state.jl:102-142 */
        if (!run->cst_dt) {
            real local = armon_oracle_dtCFL(real_r, dX[0], dX[1], d->u, d->v, d->c);
            real new_dt = local;
            if (!isfinite(new_dt) || new_dt <= 0) { run->status = ARMON_ERR_INVALID_DT; break; }
            if (current_dt == 0) new_dt = run->cfl * new_dt;
            else new_dt = mn(run->cfl * new_dt, (real)(1.05 * current_dt));
            next_cycle_dt = new_dt;
            if (current_dt == 0) current_dt = next_cycle_dt;
        }

        /* split_axes: ref src/axis_splitting.jl:24-46 */
        int axes[3]; real fac[3]; int n_sweeps = 2;
        int even = (cycle % 2 == 0);
        switch (run->splitting) {
        case ARMON_SPLIT_GODUNOV:
            axes[0] = even ? ARMON_AXIS_X : ARMON_AXIS_Y; axes[1] = even ? ARMON_AXIS_Y : ARMON_AXIS_X;
            fac[0] = fac[1] = 1.; break;
        case ARMON_SPLIT_STRANG:
            n_sweeps = 3;
            axes[0] = axes[2] = even ? ARMON_AXIS_X : ARMON_AXIS_Y; axes[1] = even ? ARMON_AXIS_Y : ARMON_AXIS_X;
            fac[0] = fac[2] = 0.5; fac[1] = 1.; break;
        case ARMON_SPLIT_X_ONLY: n_sweeps = 1; axes[0] = ARMON_AXIS_X; fac[0] = 1.; break;
        case ARMON_SPLIT_Y_ONLY: n_sweeps = 1; axes[0] = ARMON_AXIS_Y; fac[0] = 1.; break;
        default:
            axes[0] = ARMON_AXIS_X; axes[1] = ARMON_AXIS_Y; fac[0] = fac[1] = 1.;
        }
        for (int k = 0; k < n_sweeps; k++) sweep(run, d, axes[k], current_dt * fac[k]);

        /* next_cycle!: ref src/solver_state.jl:145-166 */
        cycle += 1;
        time += current_dt;
        if (run->cst_dt) current_dt = next_cycle_dt = run->Dt;
        else { current_dt = next_cycle_dt; next_cycle_dt = (real)INFINITY; }
    }
    run->solve_seconds = now_seconds() - t0;

    run->final_time = time;
    run->last_dt = current_dt;
    run->cycles = cycle;
    armon_oracle_conservation_vars(real_r, dX[0] * dX[1], d->rho, d->E, cons);
    run->final_mass = cons[0];
    run->final_energy = cons[1];
    return run->status;
}
