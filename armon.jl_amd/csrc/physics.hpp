// physics.hpp — per-cell device functions of the hot path, shared by the staged kernels and the
// fused sweeps. Operation order follows the reference expression by expression so that, built with
// -ffp-contract=off, results are bit-identical to an IEEE evaluation of the reference formulas.
// Everything is templated on the working precision T (double, or float for data_type=Float32,
// ref src/parameters.jl:185); all literals are written in T, like oracle/armon_oracle.c's R(...).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/armon_hip.h"

namespace armon {
namespace phys {

// Julia @fastmath max/min semantics (ref src/generic_kernel.jl:32-36): ifelse(y > x, y, x)
template <typename T> __device__ __forceinline__ T mx(T x, T y) { return (y > x) ? y : x; }
template <typename T> __device__ __forceinline__ T mn(T x, T y) { return (y > x) ? x : y; }

__device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
__device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
__device__ __forceinline__ double abs_(double x) { return fabs(x); }
__device__ __forceinline__ float abs_(float x) { return fabsf(x); }

// ref src/kernels.jl:4-13
template <typename T>
__device__ __forceinline__ void perfect_gas(T gamma, T rho, T E, T u, T v, T& p, T& c)
{
    T e = E - T(0.5) * (u * u + v * v);
    p = (gamma - T(1.)) * rho * e;
    c = sqrt_(gamma * p / rho);
}

// ref src/kernels.jl:16-55. WITH_G: also evaluate f3/pk0second, which only feed `g`.
template <bool WITH_G, typename T>
__device__ __forceinline__ void bizarrium(T rho, T E, T u, T v, T& p, T& c, T& g)
{
    const T rho0 = T(10000.), K0 = T(1e+11), Cv0 = T(1000.), T0 = T(300.), eps0 = T(0.), G0 = T(1.5), s = T(1.5);
    const T q = T(-42080895. / 14941154.), rr = T(727668333. / 149411540.);

    T x = rho / rho0 - 1;
    T G = G0 * (1 - rho0 / rho);
    T x2 = x * x, x3 = x * x * x;
    T opx = 1 + x, opx2 = opx * opx, opx3 = opx * opx * opx;

    T f0 = (1 + (s / 3 - 2) * x + q * x2 + rr * x3) / (1 - s * x);
    T f1 = (s / 3 - 2 + 2 * q * x + 3 * rr * x2 + s * f0) / (1 - s * x);
    T f2 = (2 * q + 6 * rr * x + 2 * s * f1) / (1 - s * x);

    T epsk0 = eps0 - Cv0 * T0 * (1 + G) + T(0.5) * (K0 / rho0) * x2 * f0;
    T pk0 = -Cv0 * T0 * G0 * rho0 + T(0.5) * K0 * x * opx2 * (2 * f0 + x * f1);
    T pk0prime = -T(0.5) * K0 * opx3 * rho0 *
                 (2 * (1 + 3 * x) * f0 + 2 * x * (2 + 3 * x) * f1 + x2 * opx * f2);

    T e = E - T(0.5) * (u * u + v * v);
    p = pk0 + G0 * rho0 * (e - epsk0);
    c = sqrt_(G0 * rho0 * (p - pk0) - pk0prime) / rho;
    if (WITH_G) {
        T opx4 = opx2 * opx2;
        T f3 = (6 * rr + 3 * s * f2) / (1 - s * x);
        T pk0second = T(0.5) * K0 * opx4 * (rho0 * rho0) *
                      (12 * (1 + 2 * x) * f0 + 6 * (1 + 6 * x + 6 * x2) * f1 +
                       6 * x * opx * (1 + 2 * x) * f2 + x2 * opx2 * f3);
        g = T(0.5) / (rho * rho * rho * (c * c)) * (pk0second + (G0 * rho0) * (G0 * rho0) * (p - pk0));
    }
}

// ref src/riemann_schemes.jl:21-30 — interface between the left cell (m = i-s) and the right cell (i)
template <typename T>
__device__ __forceinline__ void godunov(T rho_i, T rho_m, T c_i, T c_m, T u_i, T u_m, T p_i, T p_m, T& us, T& ps)
{
    T rc_l = rho_m * c_m;
    T rc_r = rho_i * c_i;
    us = (rc_l * u_m + rc_r * u_i + (p_m - p_i)) / (rc_l + rc_r);
    ps = (rc_r * p_m + rc_l * p_i + rc_l * rc_r * (u_m - u_i)) / (rc_l + rc_r);
}

// ref src/limiters.jl:6-8
template <int LIM, typename T>
__device__ __forceinline__ T limiter(T r)
{
    if (LIM == ARMON_LIMITER_MINMOD) return mx(T(0.), mn(T(1.), r));
    if (LIM == ARMON_LIMITER_SUPERBEE) return mx(mx(T(0.), mn(T(2.) * r, T(1.))), mn(r, T(2.)));
    return T(1.);
}

// Second-order part of acoustic_GAD! (ref src/riemann_schemes.jl:84-104) given the three first-order
// interface solutions: (us_m,ps_m) at i-s, (us_0,ps_0) at i, (us_p,ps_p) at i+s.
template <int LIM, typename T>
__device__ __forceinline__ void gad_flux(T dt, T dx,
                                         T rho_m, T c_m, T u_m, T p_m,   // cell i-s
                                         T rho_i, T c_i, T u_i, T p_i,   // cell i
                                         T us_m, T ps_m, T us_0, T ps_0, T us_p, T ps_p,
                                         T& us, T& ps)
{
    T r_um = (us_p - u_i) / (us_0 - u_m + T(1e-6));
    T r_pm = (ps_p - p_i) / (ps_0 - p_m + T(1e-6));
    T r_up = (u_m - us_m) / (u_i - us_0 + T(1e-6));
    T r_pp = (p_m - ps_m) / (p_i - ps_0 + T(1e-6));

    r_um = limiter<LIM>(r_um);
    r_pm = limiter<LIM>(r_pm);
    r_up = limiter<LIM>(r_up);
    r_pp = limiter<LIM>(r_pp);

    T dm_l = rho_m * dx;
    T dm_r = rho_i * dx;
    T Dm = (dm_l + dm_r) / 2;

    T rc_l = rho_m * c_m;
    T rc_r = rho_i * c_i;
    T theta = T(0.5) * (1 - (rc_l + rc_r) / 2 * (dt / Dm));

    us = us_0 + theta * (r_up * (u_i - us_0) - r_um * (us_0 - u_m));
    ps = ps_0 + theta * (r_pp * (p_i - ps_0) - r_pm * (ps_0 - p_m));
}

// ref src/kernels.jl:58-68 — (us_i,ps_i) flux on the low side of the cell, (us_n,ps_n) on the high side
template <typename T>
__device__ __forceinline__ void cell_update(T dx, T dt, T us_i, T ps_i, T us_n, T ps_n, T& rho, T& ua, T& E)
{
    T dm = rho * dx;
    rho = dm / (dx + dt * (us_n - us_i));
    ua += dt / dm * (ps_i - ps_n);
    E += dt / dm * (ps_i * us_i - ps_n * us_n);
}

// ref src/projection_schemes.jl:15-20
template <typename T>
__device__ __forceinline__ T slope_minmod(T um, T u0, T up, T r_m, T r_p)
{
    T Dp = r_p * (up - u0);
    T Dm = r_m * (u0 - um);
    T sg = (Dp > 0) ? T(1.) : ((Dp < 0) ? T(-1.) : Dp);  // Julia sign()
    return sg * mx(T(0.), mn(sg * Dp, sg * Dm));
}

// ref src/projection_schemes.jl:23-41
template <typename T>
__device__ __forceinline__ void euler_projection(T dx, T dt, T us_i, T us_n,
                                                 T a_rho_i, T a_rho_n, T a_urho_i, T a_urho_n,
                                                 T a_vrho_i, T a_vrho_n, T a_Erho_i, T a_Erho_n,
                                                 T& rho, T& u, T& v, T& E)
{
    T dX = dx + dt * (us_n - us_i);
    T t_rho  = (dX * rho     - (a_rho_n  - a_rho_i))  / dx;
    T t_urho = (dX * rho * u - (a_urho_n - a_urho_i)) / dx;
    T t_vrho = (dX * rho * v - (a_vrho_n - a_vrho_i)) / dx;
    T t_Erho = (dX * rho * E - (a_Erho_n - a_Erho_i)) / dx;
    rho = t_rho;
    u = t_urho / t_rho;
    v = t_vrho / t_rho;
    E = t_Erho / t_rho;
}

// ref src/reductions.jl:13-20 (mask-less form)
template <typename T>
__device__ __forceinline__ T dt_cfl_cell(T u, T v, T c, T dx, T dy)
{
    T ax = abs_(mx(abs_(u + c), abs_(u - c)));
    T ay = abs_(mx(abs_(v + c), abs_(v - c)));
    return mn(dx / ax, dy / ay);
}

}  // namespace phys
}  // namespace armon
