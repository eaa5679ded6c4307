// sweep_spatial.hpp — the fused X sweep with lanes laid ALONG the sweep axis.
//
// For an X sweep the stencil runs along the contiguous axis, so a wave owns a strip of 64·K consecutive
// cells of one row (K cells per lane) and every stage of the sweep is evaluated "in place": a lane
// computes the EOS of its own cells, the interface solve on the low side of each of its cells, the
// limited flux there, the Lagrangian update of its cells, their slopes, the advection flux on their
// low side and finally their projection. What the reference hands from one kernel to the next through
// memory, and what the marching pipeline (sweep_pipeline.hpp) keeps in a lane's registers across steps,
// is here fetched from the neighbouring lane with DPP wavefront shifts (v_mov_b32 wave_shr:1 /
// wave_shl:1, two per double): no LDS, no barrier, fully coalesced loads and stores, and ~1/3 of the
// registers of the marching form, so 4+ waves per SIMD hide the fp64 and memory latencies.
// Cost: the first and last HALO cells of a strip are recomputed by the neighbouring strip.
//
// The arithmetic of each stage is the same code as Pipe / PipeFast (exact: bit-identical to the staged
// kernels; tuned: shared 1-ulp reciprocals + FMAs).
#pragma once

#include "sweep_pipeline.hpp"

namespace armon {
namespace fused {

// value of the previous lane (cell i-1 for K = 1). Lane 0 has no source and reads 0 (bound_ctrl): it
// is a halo lane whose results are discarded, and not keeping its own value saves a register copy per shift.
// ROW = 0: the 64 lanes of the wave are one strip (wave_shr:1 / wave_shl:1); ROW = 1: four independent strips of 16
// lanes each (row_shr:1 / row_shl:1; lanes 0 and 15 of every row read 0) — the form of the narrow boundary strips.
template <int ROW = 0>
__device__ __forceinline__ double from_prev_lane(double x)
{
    constexpr int ctrl = ROW ? 0x111 : 0x138;                          // row_shr:1 : wave_shr:1
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int ROW = 0>
__device__ __forceinline__ float from_prev_lane(float x)
{
    constexpr int ctrl = ROW ? 0x111 : 0x138;
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, true));
}

// value of the next lane; lane 63 (lane 15 of a row) reads 0
template <int ROW = 0>
__device__ __forceinline__ double from_next_lane(double x)
{
    constexpr int ctrl = ROW ? 0x101 : 0x130;                          // row_shl:1 : wave_shl:1
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int ROW = 0>
__device__ __forceinline__ float from_next_lane(float x)
{
    constexpr int ctrl = ROW ? 0x101 : 0x130;
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, true));
}

// Per-lane strip of K consecutive cells with access to the cells just outside it.
template <int K, typename T>
struct Strip {
    T v[K];
    __device__ __forceinline__ T& operator[](int k) { return v[k]; }
    __device__ __forceinline__ T operator[](int k) const { return v[k]; }
};

template <int K, typename T>
struct Shifted {      // v[k-1] for k = 0..K-1 ("left") or v[k+1] ("right")
    T v[K];
    __device__ __forceinline__ T operator[](int k) const { return v[k]; }
};

template <int ROW, int K, typename T>
__device__ __forceinline__ Shifted<K, T> left_of(const Strip<K, T>& s)
{
    Shifted<K, T> r;
    r.v[0] = from_prev_lane<ROW>(s.v[K - 1]);
#pragma unroll
    for (int k = 1; k < K; k++) r.v[k] = s.v[k - 1];
    return r;
}

template <int ROW, int K, typename T>
__device__ __forceinline__ Shifted<K, T> right_of(const Strip<K, T>& s)
{
    Shifted<K, T> r;
    r.v[K - 1] = from_next_lane<ROW>(s.v[0]);
#pragma unroll
    for (int k = 0; k < K - 1; k++) r.v[k] = s.v[k + 1];
    return r;
}

// One X sweep of the K cells held by each lane. In: pre-sweep (ρ, u, v, E). Out: post-sweep state, valid
// for the cells at least LAG cells away from both ends of the wave's strip; p/c: EOS of the cells.
template <int SCHEME, int LIM, int PROJ, int EOS, bool EXACT, int K, typename T = double, int ROW = 0>
struct SpatialSweep {
    using TR = PipeTraits<SCHEME, LIM, PROJ, EOS>;
    static constexpr int S = TR::S, W = TR::W, LAG = TR::LAG;
    using S_ = Strip<K, T>;
    using Sh = Shifted<K, T>;
    // neighbour access of this sweep's strip shape (the whole wave, or rows of 16 lanes)
    static __device__ __forceinline__ Sh left_of(const S_& s) { return fused::left_of<ROW, K, T>(s); }
    static __device__ __forceinline__ Sh right_of(const S_& s) { return fused::right_of<ROW, K, T>(s); }

    T dt, dx, gamma;
    // 1 / dx and dt / dx, formed by the caller (the host, in the run's precision; armon_hip_sweep): a wave lives for one
    // strip, and the two IEEE divisions (two v_rcp_f64 at 4x the issue cost of an FMA + their expansions, ~40 FMA slots of
    // a strip's ~600) were per-wave work. Tuned arithmetic only; the exact flavour prepares its own denominators.
    T inv_dx, dt_dx;

    __device__ __forceinline__ void run(const S_& rho, const S_& u, const S_& v, const S_& E,
                                        S_& o_rho, S_& o_u, S_& o_v, S_& o_E, S_& p, S_& cs) const
    {
        if (EXACT) run_exact(rho, u, v, E, o_rho, o_u, o_v, o_E, p, cs);
        else run_fast(rho, u, v, E, o_rho, o_u, o_v, o_E, p, cs);
    }

    // ---------------------------------------------------------------------------------------------
    __device__ __forceinline__ void run_exact(const S_& rho, const S_& ua, const S_& ut, const S_& E,
                                              S_& o_rho, S_& o_u, S_& o_v, S_& o_E, S_& p, S_& cs) const
    {
        using Den = xct::Den<T>;       // correctly rounded quotients, shared denominators (sweep_pipeline.hpp)
        const Den d_dx(dx);
        S_ rc;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (EOS == ARMON_EOS_BIZARRIUM) {
                T g_unused;
                phys::bizarrium<false>(rho[k], E[k], ua[k], ut[k], p.v[k], cs.v[k], g_unused);
            } else {
                phys::perfect_gas(gamma, rho[k], E[k], ua[k], ut[k], p.v[k], cs.v[k]);
            }
            rc.v[k] = rho[k] * cs[k];
        }
        // first-order solve on the low side of each cell (ref src/riemann_schemes.jl:21-30)
        const Sh rcL = left_of(rc), uL = left_of(ua), pL = left_of(p);
        S_ gus, gps;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const T rc_l = rcL[k], rc_r = rc[k];
            const Den d(rc_l + rc_r);
            gus.v[k] = d.quo_t(rc_l * uL[k] + rc_r * ua[k] + (pL[k] - p[k]));
            gps.v[k] = d.quo(rc_r * pL[k] + rc_l * p[k] + rc_l * rc_r * (uL[k] - ua[k]));
        }
        S_ fus, fps;
        if (S == 1) {
            // acoustic_GAD! (ref src/riemann_schemes.jl:84-104): cells i-s = left, i = own
            const Sh rhoL = left_of(rho);
            const Sh gusL = left_of(gus), gpsL = left_of(gps), gusR = right_of(gus), gpsR = right_of(gps);
#pragma unroll
            for (int k = 0; k < K; k++) {
                const T r_um = phys::limiter<LIM>(Den(gus[k] - uL[k] + T(1e-6)).quo_t(gusR[k] - ua[k]));
                const T r_pm = phys::limiter<LIM>(Den(gps[k] - pL[k] + T(1e-6)).quo(gpsR[k] - p[k]));
                const T r_up = phys::limiter<LIM>(Den(ua[k] - gus[k] + T(1e-6)).quo_t(uL[k] - gusL[k]));
                const T r_pp = phys::limiter<LIM>(Den(p[k] - gps[k] + T(1e-6)).quo(pL[k] - gpsL[k]));
                const T dm_l = rhoL[k] * dx;
                const T dm_r = rho[k] * dx;
                const T Dm = (dm_l + dm_r) / 2;
                const T theta = T(0.5) * (1 - (rcL[k] + rc[k]) / 2 * Den(Dm).quo(dt));
                fus.v[k] = gus[k] + theta * (r_up * (ua[k] - gus[k]) - r_um * (gus[k] - uL[k]));
                fps.v[k] = gps[k] + theta * (r_pp * (p[k] - gps[k]) - r_pm * (gps[k] - pL[k]));
            }
        } else {
            fus = gus;
            fps = gps;
        }
        // Lagrangian update (ref src/kernels.jl:58-68)
        const Sh fusR = right_of(fus), fpsR = right_of(fps);
        S_ l_rho, l_ua, l_E, dxl, q_ua, q_ut, q_E;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const T dm = rho[k] * dx;
            dxl.v[k] = dx + dt * (fusR[k] - fus[k]);
            l_rho.v[k] = Den(dxl[k]).quo(dm);
            const T dt_dm = Den(dm).quo(dt);
            l_ua.v[k] = ua[k] + dt_dm * (fps[k] - fpsR[k]);
            l_E.v[k] = E[k] + dt_dm * (fps[k] * fus[k] - fpsR[k] * fusR[k]);
            q_ua.v[k] = l_rho[k] * l_ua[k];
            q_ut.v[k] = l_rho[k] * ut[k];
            q_E.v[k] = l_rho[k] * l_E[k];
        }
        // advection flux on the low side of each cell
        S_ a0, a1, a2, a3;    // ρ, ρu, ρv, ρE
        const Sh rL = left_of(l_rho), quL = left_of(q_ua), qvL = left_of(q_ut), qEL = left_of(q_E);
        if (W == 1) {
            const Sh rR = right_of(l_rho), quR = right_of(q_ua), qvR = right_of(q_ut), qER = right_of(q_E);
            const Sh dxlL = left_of(dxl), dxlR = right_of(dxl);
            S_ s0, s1, s2, s3;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const T r_m = Den(dxl[k] + dxlL[k]).quo(2 * dxl[k]);
                const T r_p = Den(dxl[k] + dxlR[k]).quo(2 * dxl[k]);
                s0.v[k] = phys::slope_minmod(rL[k], l_rho[k], rR[k], r_m, r_p);
                s1.v[k] = phys::slope_minmod(quL[k], q_ua[k], quR[k], r_m, r_p);
                s2.v[k] = phys::slope_minmod(qvL[k], q_ut[k], qvR[k], r_m, r_p);
                s3.v[k] = phys::slope_minmod(qEL[k], q_E[k], qER[k], r_m, r_p);
            }
            const Sh s0L = left_of(s0), s1L = left_of(s1), s2L = left_of(s2), s3L = left_of(s3);
            const Sh fusL = left_of(fus);
#pragma unroll
            for (int k = 0; k < K; k++) {
                const T disp = dt * fus[k];
                const bool up = disp > 0;
                const T Dxe = up ? -(dx - dt * fusL[k]) : (dx + dt * fusR[k]);
                const T lf = Den(2 * (up ? dxlL[k] : dxl[k])).quo(Dxe);
                a0.v[k] = disp * ((up ? rL[k] : l_rho[k]) - (up ? s0L[k] : s0[k]) * lf);
                a1.v[k] = disp * ((up ? quL[k] : q_ua[k]) - (up ? s1L[k] : s1[k]) * lf);
                a2.v[k] = disp * ((up ? qvL[k] : q_ut[k]) - (up ? s2L[k] : s2[k]) * lf);
                a3.v[k] = disp * ((up ? qEL[k] : q_E[k]) - (up ? s3L[k] : s3[k]) * lf);
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const T disp = dt * fus[k];
                const bool up = disp > 0;
                a0.v[k] = disp * (up ? rL[k] : l_rho[k]);
                a1.v[k] = disp * (up ? quL[k] : q_ua[k]);
                a2.v[k] = disp * (up ? qvL[k] : q_ut[k]);
                a3.v[k] = disp * (up ? qEL[k] : q_E[k]);
            }
        }
        // projection (ref src/projection_schemes.jl:23-41)
        const Sh a0R = right_of(a0), a1R = right_of(a1), a2R = right_of(a2), a3R = right_of(a3);
#pragma unroll
        for (int k = 0; k < K; k++) {
            const T dX = dxl[k];
            const T t_rho  = d_dx.quo(dX * l_rho[k]           - (a0R[k] - a0[k]));
            const T t_urho = d_dx.quo_t(dX * l_rho[k] * l_ua[k] - (a1R[k] - a1[k]));
            const T t_vrho = d_dx.quo_t(dX * l_rho[k] * ut[k]   - (a2R[k] - a2[k]));
            const T t_Erho = d_dx.quo(dX * l_rho[k] * l_E[k]  - (a3R[k] - a3[k]));
            const Den d_rho(t_rho);
            o_rho.v[k] = t_rho;
            o_u.v[k] = d_rho.quo_t(t_urho);
            o_v.v[k] = d_rho.quo_t(t_vrho);
            o_E.v[k] = d_rho.quo(t_Erho);
        }
    }

    // ---------------------------------------------------------------------------------------------
    __device__ __forceinline__ void run_fast(const S_& rho, const S_& ua, const S_& ut, const S_& E,
                                             S_& o_rho, S_& o_u, S_& o_v, S_& o_E, S_& p, S_& cs) const
    {
        using namespace fast;
        const T gm1 = gamma - T(1.), ggm1 = gamma * (gamma - T(1.));
        S_ rc;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (EOS == ARMON_EOS_BIZARRIUM) {
                fast::bizarrium(rho[k], ua[k], ut[k], E[k], p.v[k], cs.v[k]);
            } else {
                const T e = fma_(T(-0.5), fma_(ua[k], ua[k], ut[k] * ut[k]), E[k]);
                p.v[k] = gm1 * rho[k] * e;
                cs.v[k] = sqrt_(ggm1 * e);
            }
            rc.v[k] = rho[k] * cs[k];
        }
        const Sh rcL = left_of(rc), uL = left_of(ua), pL = left_of(p);
        S_ gus, gps, src;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const T rc_l = rcL[k], rc_r = rc[k];
            src.v[k] = rc_l + rc_r;
            const T inv = rcp(src[k]);
            gus.v[k] = fma_(rc_l, uL[k], fma_(rc_r, ua[k], pL[k] - p[k])) * inv;
            gps.v[k] = fma_(rc_r, pL[k], fma_(rc_l, p[k], rc_l * rc_r * (uL[k] - ua[k]))) * inv;
        }
        S_ fus, fps;
        if (S == 1) {
            const Sh rhoL = left_of(rho);
            const Sh gusL = left_of(gus), gpsL = left_of(gps), gusR = right_of(gus), gpsR = right_of(gps);
#pragma unroll
            for (int k = 0; k < K; k++) {
                const T Au = gus[k] - uL[k], Bu = ua[k] - gus[k];
                const T Ap = gps[k] - pL[k], Bp = p[k] - gps[k];
                const T r_um = limiter<LIM>((gusR[k] - ua[k]) * rcp1(Au + T(1e-6)));
                const T r_pm = limiter<LIM>((gpsR[k] - p[k]) * rcp1(Ap + T(1e-6)));
                const T r_up = limiter<LIM>((uL[k] - gusL[k]) * rcp1(Bu + T(1e-6)));
                const T r_pp = limiter<LIM>((pL[k] - gpsL[k]) * rcp1(Bp + T(1e-6)));
                const T theta = fma_(T(-0.5) * src[k] * dt_dx, rcp1(rhoL[k] + rho[k]), T(0.5));
                fus.v[k] = fma_(theta, fma_(r_up, Bu, -r_um * Au), gus[k]);
                fps.v[k] = fma_(theta, fma_(r_pp, Bp, -r_pm * Ap), gps[k]);
            }
        } else {
            fus = gus;
            fps = gps;
        }
        S_ dtu, pu;
#pragma unroll
        for (int k = 0; k < K; k++) {
            dtu.v[k] = dt * fus[k];
            pu.v[k] = fps[k] * fus[k];
        }
        const Sh dtuR = right_of(dtu), fpsR = right_of(fps), puR = right_of(pu);
        S_ l_rho, dxl, hinv, q_ua, q_ut, q_E;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const T dtdm = dt_dx * rcp(rho[k]);
            dxl.v[k] = (dx + dtuR[k]) - dtu[k];
            const T inv_dxl = rcp(dxl[k]);
            hinv.v[k] = T(0.5) * inv_dxl;
            l_rho.v[k] = rho[k] * dx * inv_dxl;
            const T ua_n = fma_(dtdm, fps[k] - fpsR[k], ua[k]);
            const T E_n = fma_(dtdm, pu[k] - puR[k], E[k]);
            q_ua.v[k] = l_rho[k] * ua_n;
            q_ut.v[k] = l_rho[k] * ut[k];
            q_E.v[k] = l_rho[k] * E_n;
        }
        S_ a0, a1, a2, a3;
        const Sh rL = left_of(l_rho), quL = left_of(q_ua), qvL = left_of(q_ut), qEL = left_of(q_E);
        if (W == 1) {
            const Sh rR = right_of(l_rho), quR = right_of(q_ua), qvR = right_of(q_ut), qER = right_of(q_E);
            // r₋ = 2Δx / (Δx + Δx₋), r₊ = 2Δx / (Δx + Δx₊): a cell's Δx + Δx₊ is its right neighbour's Δx₋ + Δx (the same
            // bits: IEEE addition commutes), so every cell inverts ONE sum and takes the other reciprocal from its
            // neighbour (one shift instead of a second v_rcp_f64, which issues at a quarter of an FMA's rate)
            const Sh dxlL = left_of(dxl);
            S_ isum;
#pragma unroll
            for (int k = 0; k < K; k++) isum.v[k] = rcp1(dxl[k] + dxlL[k]);
            const Sh isumR = right_of(isum);
            S_ s0, s1, s2, s3;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const T two_dxl = T(2.) * dxl[k];
                const T r_m = two_dxl * isum[k];
                const T r_p = two_dxl * isumR[k];
                s0.v[k] = minmod(r_p * (rR[k] - l_rho[k]), r_m * (l_rho[k] - rL[k]));
                s1.v[k] = minmod(r_p * (quR[k] - q_ua[k]), r_m * (q_ua[k] - quL[k]));
                s2.v[k] = minmod(r_p * (qvR[k] - q_ut[k]), r_m * (q_ut[k] - qvL[k]));
                s3.v[k] = minmod(r_p * (qER[k] - q_E[k]), r_m * (q_E[k] - qEL[k]));
            }
            const Sh s0L = left_of(s0), s1L = left_of(s1), s2L = left_of(s2), s3L = left_of(s3);
            const Sh dtuL = left_of(dtu), hinvL = left_of(hinv);
#pragma unroll
            for (int k = 0; k < K; k++) {
                const T disp = dtu[k];
                const bool up = disp > 0;
                const T Dxe = up ? (dtuL[k] - dx) : (dx + dtuR[k]);
                const T lf = Dxe * (up ? hinvL[k] : hinv[k]);
                a0.v[k] = disp * fma_(-(up ? s0L[k] : s0[k]), lf, up ? rL[k] : l_rho[k]);
                a1.v[k] = disp * fma_(-(up ? s1L[k] : s1[k]), lf, up ? quL[k] : q_ua[k]);
                a2.v[k] = disp * fma_(-(up ? s2L[k] : s2[k]), lf, up ? qvL[k] : q_ut[k]);
                a3.v[k] = disp * fma_(-(up ? s3L[k] : s3[k]), lf, up ? qEL[k] : q_E[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const T disp = dtu[k];
                const bool up = disp > 0;
                a0.v[k] = disp * (up ? rL[k] : l_rho[k]);
                a1.v[k] = disp * (up ? quL[k] : q_ua[k]);
                a2.v[k] = disp * (up ? qvL[k] : q_ut[k]);
                a3.v[k] = disp * (up ? qEL[k] : q_E[k]);
            }
        }
        const Sh a0R = right_of(a0), a1R = right_of(a1), a2R = right_of(a2), a3R = right_of(a3);
#pragma unroll
        for (int k = 0; k < K; k++) {
            const T dX = dxl[k];
            const T T_rho = fma_(dX, l_rho[k], a0[k] - a0R[k]);
            const T T_u = fma_(dX, q_ua[k], a1[k] - a1R[k]);
            const T T_v = fma_(dX, q_ut[k], a2[k] - a2R[k]);
            const T T_E = fma_(dX, q_E[k], a3[k] - a3R[k]);
            const T inv = rcp(T_rho);
            o_rho.v[k] = T_rho * inv_dx;
            o_u.v[k] = T_u * inv;
            o_v.v[k] = T_v * inv;
            o_E.v[k] = T_E * inv;
        }
    }

};

}  // namespace fused
}  // namespace armon
