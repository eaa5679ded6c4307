// physics.hpp — per-cell device functions of the hot path, shared by the staged kernels and the
// fused sweeps. Operation order follows the reference expression by expression so that, built with
// -ffp-contract=off, results are bit-identical to an IEEE evaluation of the reference formulas.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/armon_hip.h"

namespace armon {
namespace phys {

// Julia @fastmath max/min semantics (ref src/generic_kernel.jl:32-36): ifelse(y > x, y, x)
__device__ __forceinline__ double mx(double x, double y) { return (y > x) ? y : x; }
__device__ __forceinline__ double mn(double x, double y) { return (y > x) ? x : y; }

// ref src/kernels.jl:4-13
__device__ __forceinline__ void perfect_gas(double gamma, double rho, double E, double u, double v,
                                            double& p, double& c)
{
    double e = E - 0.5 * (u * u + v * v);
    p = (gamma - 1.) * rho * e;
    c = sqrt(gamma * p / rho);
}

// ref src/kernels.jl:16-55. WITH_G: also evaluate f3/pk0second, which only feed `g`.
template <bool WITH_G>
__device__ __forceinline__ void bizarrium(double rho, double E, double u, double v,
                                          double& p, double& c, double& g)
{
    const double rho0 = 10000., K0 = 1e+11, Cv0 = 1000., T0 = 300., eps0 = 0., G0 = 1.5, s = 1.5;
    const double q = -42080895. / 14941154., rr = 727668333. / 149411540.;

    double x = rho / rho0 - 1;
    double G = G0 * (1 - rho0 / rho);
    double x2 = x * x, x3 = x * x * x;
    double opx = 1 + x, opx2 = opx * opx, opx3 = opx * opx * opx;

    double f0 = (1 + (s / 3 - 2) * x + q * x2 + rr * x3) / (1 - s * x);
    double f1 = (s / 3 - 2 + 2 * q * x + 3 * rr * x2 + s * f0) / (1 - s * x);
    double f2 = (2 * q + 6 * rr * x + 2 * s * f1) / (1 - s * x);

    double epsk0 = eps0 - Cv0 * T0 * (1 + G) + 0.5 * (K0 / rho0) * x2 * f0;
    double pk0 = -Cv0 * T0 * G0 * rho0 + 0.5 * K0 * x * opx2 * (2 * f0 + x * f1);
    double pk0prime = -0.5 * K0 * opx3 * rho0 *
                      (2 * (1 + 3 * x) * f0 + 2 * x * (2 + 3 * x) * f1 + x2 * opx * f2);

    double e = E - 0.5 * (u * u + v * v);
    p = pk0 + G0 * rho0 * (e - epsk0);
    c = sqrt(G0 * rho0 * (p - pk0) - pk0prime) / rho;
    if (WITH_G) {
        double opx4 = opx2 * opx2;
        double f3 = (6 * rr + 3 * s * f2) / (1 - s * x);
        double pk0second = 0.5 * K0 * opx4 * (rho0 * rho0) *
                           (12 * (1 + 2 * x) * f0 + 6 * (1 + 6 * x + 6 * x2) * f1 +
                            6 * x * opx * (1 + 2 * x) * f2 + x2 * opx2 * f3);
        g = 0.5 / (rho * rho * rho * (c * c)) * (pk0second + (G0 * rho0) * (G0 * rho0) * (p - pk0));
    }
}

// ref src/riemann_schemes.jl:21-30 — interface between the left cell (m = i-s) and the right cell (i)
__device__ __forceinline__ void godunov(double rho_i, double rho_m, double c_i, double c_m,
                                        double u_i, double u_m, double p_i, double p_m,
                                        double& us, double& ps)
{
    double rc_l = rho_m * c_m;
    double rc_r = rho_i * c_i;
    us = (rc_l * u_m + rc_r * u_i + (p_m - p_i)) / (rc_l + rc_r);
    ps = (rc_r * p_m + rc_l * p_i + rc_l * rc_r * (u_m - u_i)) / (rc_l + rc_r);
}

// ref src/limiters.jl:6-8
template <int LIM>
__device__ __forceinline__ double limiter(double r)
{
    if (LIM == ARMON_LIMITER_MINMOD) return mx(0., mn(1., r));
    if (LIM == ARMON_LIMITER_SUPERBEE) return mx(mx(0., mn(2. * r, 1.)), mn(r, 2.));
    return 1.;
}

// Second-order part of acoustic_GAD! (ref src/riemann_schemes.jl:84-104) given the three first-order
// interface solutions: (us_m,ps_m) at i-s, (us_0,ps_0) at i, (us_p,ps_p) at i+s.
template <int LIM>
__device__ __forceinline__ void gad_flux(double dt, double dx,
                                         double rho_m, double c_m, double u_m, double p_m,   // cell i-s
                                         double rho_i, double c_i, double u_i, double p_i,   // cell i
                                         double us_m, double ps_m, double us_0, double ps_0,
                                         double us_p, double ps_p,
                                         double& us, double& ps)
{
    double r_um = (us_p - u_i) / (us_0 - u_m + 1e-6);
    double r_pm = (ps_p - p_i) / (ps_0 - p_m + 1e-6);
    double r_up = (u_m - us_m) / (u_i - us_0 + 1e-6);
    double r_pp = (p_m - ps_m) / (p_i - ps_0 + 1e-6);

    r_um = limiter<LIM>(r_um);
    r_pm = limiter<LIM>(r_pm);
    r_up = limiter<LIM>(r_up);
    r_pp = limiter<LIM>(r_pp);

    double dm_l = rho_m * dx;
    double dm_r = rho_i * dx;
    double Dm = (dm_l + dm_r) / 2;

    double rc_l = rho_m * c_m;
    double rc_r = rho_i * c_i;
    double theta = 0.5 * (1 - (rc_l + rc_r) / 2 * (dt / Dm));

    us = us_0 + theta * (r_up * (u_i - us_0) - r_um * (us_0 - u_m));
    ps = ps_0 + theta * (r_pp * (p_i - ps_0) - r_pm * (ps_0 - p_m));
}

// ref src/kernels.jl:58-68 — (us_i,ps_i) flux on the low side of the cell, (us_n,ps_n) on the high side
__device__ __forceinline__ void cell_update(double dx, double dt, double us_i, double ps_i,
                                            double us_n, double ps_n,
                                            double& rho, double& ua, double& E)
{
    double dm = rho * dx;
    rho = dm / (dx + dt * (us_n - us_i));
    ua += dt / dm * (ps_i - ps_n);
    E += dt / dm * (ps_i * us_i - ps_n * us_n);
}

// ref src/projection_schemes.jl:15-20
__device__ __forceinline__ double slope_minmod(double um, double u0, double up, double r_m, double r_p)
{
    double Dp = r_p * (up - u0);
    double Dm = r_m * (u0 - um);
    double sg = (Dp > 0) ? 1. : ((Dp < 0) ? -1. : Dp);  // Julia sign()
    return sg * mx(0., mn(sg * Dp, sg * Dm));
}

// ref src/projection_schemes.jl:23-41
__device__ __forceinline__ void euler_projection(double dx, double dt, double us_i, double us_n,
                                                 double a_rho_i, double a_rho_n,
                                                 double a_urho_i, double a_urho_n,
                                                 double a_vrho_i, double a_vrho_n,
                                                 double a_Erho_i, double a_Erho_n,
                                                 double& rho, double& u, double& v, double& E)
{
    double dX = dx + dt * (us_n - us_i);
    double t_rho  = (dX * rho     - (a_rho_n  - a_rho_i))  / dx;
    double t_urho = (dX * rho * u - (a_urho_n - a_urho_i)) / dx;
    double t_vrho = (dX * rho * v - (a_vrho_n - a_vrho_i)) / dx;
    double t_Erho = (dX * rho * E - (a_Erho_n - a_Erho_i)) / dx;
    rho = t_rho;
    u = t_urho / t_rho;
    v = t_vrho / t_rho;
    E = t_Erho / t_rho;
}

// ref src/reductions.jl:13-20 (mask-less form)
__device__ __forceinline__ double dt_cfl_cell(double u, double v, double c, double dx, double dy)
{
    double ax = fabs(mx(fabs(u + c), fabs(u - c)));
    double ay = fabs(mx(fabs(v + c), fabs(v - c)));
    return mn(dx / ax, dy / ay);
}

}  // namespace phys
}  // namespace armon
