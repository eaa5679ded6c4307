// physics.hpp — per-cell device functions of the hot path, shared by the staged kernels and the
// fused sweeps. Operation order follows the reference expression by expression so that, built with
// -ffp-contract=off, results are bit-identical to an IEEE evaluation of the reference formulas; the
// divisions and square roots go through xct:: (below): correctly rounded, hence the same bits, but
// without the range scaling / fix-up of the compiler's expansion and with shared denominators.
// Everything is templated on the working precision T (double, or float for data_type=Float32,
// ref src/parameters.jl:185); all literals are written in T, like oracle/armon_oracle.c's R(...).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/armon_hip.h"

namespace armon {

// ======================================================================================================
// Correctly rounded division and square root at speed (exact flavour)
// ======================================================================================================
// The compiler's IEEE a/b is v_div_scale ×2, v_rcp_f64, two Newton steps, quotient, remainder, v_div_fmas,
// v_div_fixup: 11 instructions, 20 times per cell. The scale / fix-up instructions only act when an operand or
// the quotient leaves ~2^±500 (or is 0, inf, NaN); everything the solver divides is within 1e±30, and there the
// remaining operations — reproduced below in the same order — give the same, correctly rounded bits (asserted by
// every bit-parity test against the oracle, which divides with the CPU's IEEE instruction). A prepared denominator
// (reciprocal refined once: 5 instructions) then serves every quotient that shares it for 3 instructions each:
// the two Godunov quotients (ref src/riemann_schemes.jl:27-28), the three u,v,E of the projection, its four /dx,
// the two dt/dm of the cell update, consecutive slope ratios.
namespace xct {

template <typename T> struct Den;

// fp32: the compiler's own expansion (it toggles the denormal mode around its FMAs; nothing to share cheaply)
template <> struct Den<float> {
    float b;
    __device__ __forceinline__ Den() : b(1.f) {}
    __device__ __forceinline__ explicit Den(float b_) : b(b_) {}
    __device__ __forceinline__ float quo(float a) const { return a / b; }
    __device__ __forceinline__ float quo_t(float a) const { return a / b; }
    __device__ __forceinline__ Den twice() const { return Den(2.f * b); }
};

#ifdef ARMON_XCT_PLAIN   // A/B builds: the compiler's IEEE expansion everywhere (tools/build_variant.sh)
template <> struct Den<double> {
    double b;
    __device__ __forceinline__ Den() : b(1.) {}
    __device__ __forceinline__ explicit Den(double b_) : b(b_) {}
    __device__ __forceinline__ double quo(double a) const { return a / b; }
    __device__ __forceinline__ double quo_t(double a) const { return a / b; }
    __device__ __forceinline__ Den twice() const { return Den(2. * b); }
};
__device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
#else
template <> struct Den<double> {
    double b, r;                                     // denominator and its reciprocal (≤ 1 ulp)
    __device__ __forceinline__ Den() : b(1.), r(1.) {}
    __device__ __forceinline__ explicit Den(double b_) : b(b_)
    {
        double y = __builtin_amdgcn_rcp(b_);
        double e = __builtin_fma(-b_, y, 1.0);
        y = __builtin_fma(y, e, y);
        e = __builtin_fma(-b_, y, 1.0);
        r = __builtin_fma(y, e, y);
    }
    __device__ __forceinline__ double quo(double a) const       // a / b
    {
        const double q = a * r;
        const double rem = __builtin_fma(-b, q, a);
        return __builtin_fma(rem, r, q);
    }
    // a / b where the quotient may fall below the normal range — the velocities of a fluid at rest, their fluxes and slope
    // ratios (Sedov's far field carries 1e-320). There quo()'s remainder step underflows and the result can be one unit of
    // the subnormal grid (4.9e-324) off an IEEE division (1 run in 800 of tests/test_gpu_random_shapes.py), so the rare tiny
    // quotient goes through the IEEE expansion instead: the exact flavour holds EVERY bit of the oracle, subnormal values
    // included, for 5 % of its VALU-bound sweeps (seven guarded quotients per cell; 85 -> 80 Gcells/s at 16384²; until
    // round 4 only the A/B library was built this way). -DARMON_LOOSE_SUBNORMAL: the unguarded form, for measurements.
    __device__ __forceinline__ double quo_t(double a) const
    {
        const double q = quo(a);
#ifndef ARMON_LOOSE_SUBNORMAL
        // (an exact zero — the transverse velocity of Sod, a fluid at rest — is the same zero either way and stays on the fast path)
        if (__builtin_expect(__builtin_fabs(q) < 0x1p-960 && a != 0., 0)) return a / b;
#endif
        return q;
    }
    __device__ __forceinline__ Den twice() const                // the denominator 2·b (exact scaling)
    {
        Den d;
        d.b = 2. * b;
        d.r = 0.5 * r;
        return d;
    }
};

__device__ __forceinline__ double sqrt_(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return (x == 0.) ? x : g;
}
#endif
__device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }

}  // namespace xct

namespace phys {

using xct::Den;

// Julia @fastmath max/min semantics (ref src/generic_kernel.jl:32-36): ifelse(y > x, y, x)
template <typename T> __device__ __forceinline__ T mx(T x, T y) { return (y > x) ? y : x; }
template <typename T> __device__ __forceinline__ T mn(T x, T y) { return (y > x) ? x : y; }

// Running maximum of NON-NEGATIVE values in which a NaN sticks — what the dt/CFL reductions accumulate with. The
// reference raises on the first non-finite time step (ref src/solver_state.jl:123-124), so a NaN sound speed or velocity
// in ONE cell must reach the host, while mx() above drops a NaN operand on either side. For non-negative IEEE values the
// order of the bit patterns as unsigned integers is the order of the values, with +inf above every finite value and every
// (sign-cleared) NaN above +inf: an unsigned maximum of the bits is the same maximum, NaN wins, and stays.
__device__ __forceinline__ double amax(double acc, double val)
{
    const unsigned long long a = (unsigned long long)__double_as_longlong(acc), b = (unsigned long long)__double_as_longlong(val);
    return __longlong_as_double((long long)(b > a ? b : a));
}
__device__ __forceinline__ float amax(float acc, float val)
{
    const unsigned a = __float_as_uint(acc), b = __float_as_uint(val);
    return __uint_as_float(b > a ? b : a);
}
// minimum in which a NaN on either side wins (folds of the dt/CFL reductions)
template <typename T> __device__ __forceinline__ T mn_nan(T x, T y) { return (y < x || y != y) ? y : x; }

__device__ __forceinline__ double sqrt_(double x) { return xct::sqrt_(x); }
__device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
__device__ __forceinline__ double abs_(double x) { return fabs(x); }
__device__ __forceinline__ float abs_(float x) { return fabsf(x); }

// ref src/kernels.jl:4-13
template <typename T>
__device__ __forceinline__ void perfect_gas(T gamma, T rho, T E, T u, T v, T& p, T& c)
{
    T e = E - T(0.5) * (u * u + v * v);
    p = (gamma - T(1.)) * rho * e;
    c = sqrt_(Den<T>(rho).quo(gamma * p));
}

// ref src/kernels.jl:16-55. WITH_G: also evaluate f3/pk0second, which only feed `g`.
template <bool WITH_G, typename T>
__device__ __forceinline__ void bizarrium(T rho, T E, T u, T v, T& p, T& c, T& g)
{
    const T rho0 = T(10000.), K0 = T(1e+11), Cv0 = T(1000.), T0 = T(300.), eps0 = T(0.), G0 = T(1.5), s = T(1.5);
    const T q = T(-42080895. / 14941154.), rr = T(727668333. / 149411540.);

    const Den<T> d_rho(rho);
    T x = Den<T>(rho0).quo(rho) - 1;
    T G = G0 * (1 - d_rho.quo(rho0));
    T x2 = x * x, x3 = x * x * x;
    T opx = 1 + x, opx2 = opx * opx, opx3 = opx * opx * opx;

    const Den<T> d_sx(1 - s * x);
    T f0 = d_sx.quo(1 + (s / 3 - 2) * x + q * x2 + rr * x3);
    T f1 = d_sx.quo(s / 3 - 2 + 2 * q * x + 3 * rr * x2 + s * f0);
    T f2 = d_sx.quo(2 * q + 6 * rr * x + 2 * s * f1);

    T epsk0 = eps0 - Cv0 * T0 * (1 + G) + T(0.5) * (K0 / rho0) * x2 * f0;
    T pk0 = -Cv0 * T0 * G0 * rho0 + T(0.5) * K0 * x * opx2 * (2 * f0 + x * f1);
    T pk0prime = -T(0.5) * K0 * opx3 * rho0 *
                 (2 * (1 + 3 * x) * f0 + 2 * x * (2 + 3 * x) * f1 + x2 * opx * f2);

    T e = E - T(0.5) * (u * u + v * v);
    p = pk0 + G0 * rho0 * (e - epsk0);
    c = d_rho.quo(sqrt_(G0 * rho0 * (p - pk0) - pk0prime));
    if (WITH_G) {
        T opx4 = opx2 * opx2;
        T f3 = d_sx.quo(6 * rr + 3 * s * f2);
        T pk0second = T(0.5) * K0 * opx4 * (rho0 * rho0) *
                      (12 * (1 + 2 * x) * f0 + 6 * (1 + 6 * x + 6 * x2) * f1 +
                       6 * x * opx * (1 + 2 * x) * f2 + x2 * opx2 * f3);
        g = Den<T>(rho * rho * rho * (c * c)).quo(T(0.5)) * (pk0second + (G0 * rho0) * (G0 * rho0) * (p - pk0));
    }
}

// ref src/riemann_schemes.jl:21-30 — interface between the left cell (m = i-s) and the right cell (i)
template <typename T>
__device__ __forceinline__ void godunov(T rho_i, T rho_m, T c_i, T c_m, T u_i, T u_m, T p_i, T p_m, T& us, T& ps)
{
    T rc_l = rho_m * c_m;
    T rc_r = rho_i * c_i;
    const Den<T> d(rc_l + rc_r);                  // one denominator, two quotients
    us = d.quo_t(rc_l * u_m + rc_r * u_i + (p_m - p_i));
    ps = d.quo(rc_r * p_m + rc_l * p_i + rc_l * rc_r * (u_m - u_i));
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
    T r_um = Den<T>(us_0 - u_m + T(1e-6)).quo_t(us_p - u_i);
    T r_pm = Den<T>(ps_0 - p_m + T(1e-6)).quo(ps_p - p_i);
    T r_up = Den<T>(u_i - us_0 + T(1e-6)).quo_t(u_m - us_m);
    T r_pp = Den<T>(p_i - ps_0 + T(1e-6)).quo(p_m - ps_m);

    r_um = limiter<LIM>(r_um);
    r_pm = limiter<LIM>(r_pm);
    r_up = limiter<LIM>(r_up);
    r_pp = limiter<LIM>(r_pp);

    T dm_l = rho_m * dx;
    T dm_r = rho_i * dx;
    T Dm = (dm_l + dm_r) / 2;

    T rc_l = rho_m * c_m;
    T rc_r = rho_i * c_i;
    T theta = T(0.5) * (1 - (rc_l + rc_r) / 2 * Den<T>(Dm).quo(dt));

    us = us_0 + theta * (r_up * (u_i - us_0) - r_um * (us_0 - u_m));
    ps = ps_0 + theta * (r_pp * (p_i - ps_0) - r_pm * (ps_0 - p_m));
}

// ref src/kernels.jl:58-68 — (us_i,ps_i) flux on the low side of the cell, (us_n,ps_n) on the high side
template <typename T>
__device__ __forceinline__ void cell_update(T dx, T dt, T us_i, T ps_i, T us_n, T ps_n, T& rho, T& ua, T& E)
{
    T dm = rho * dx;
    rho = Den<T>(dx + dt * (us_n - us_i)).quo(dm);
    const T dt_dm = Den<T>(dm).quo(dt);
    ua += dt_dm * (ps_i - ps_n);
    E += dt_dm * (ps_i * us_i - ps_n * us_n);
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
    const Den<T> d_dx(dx);
    T t_rho  = d_dx.quo(dX * rho     - (a_rho_n  - a_rho_i));
    T t_urho = d_dx.quo_t(dX * rho * u - (a_urho_n - a_urho_i));
    T t_vrho = d_dx.quo_t(dX * rho * v - (a_vrho_n - a_vrho_i));
    T t_Erho = d_dx.quo(dX * rho * E - (a_Erho_n - a_Erho_i));
    const Den<T> d_rho(t_rho);
    rho = t_rho;
    u = d_rho.quo_t(t_urho);
    v = d_rho.quo_t(t_vrho);
    E = d_rho.quo(t_Erho);
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
