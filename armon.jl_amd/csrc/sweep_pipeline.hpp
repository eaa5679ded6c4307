// sweep_pipeline.hpp — the fused directional sweep as a register-resident software pipeline.
//
// One lane marches along the sweep axis, one cell per step, and keeps in registers everything the
// reference passes between its five per-sweep kernels through memory (p, c, uˢ, pˢ, work_1..4 —
// ref src/solver.jl:300-316): EOS → first-order interface solve → GAD limited flux → Lagrangian cell
// update → advection (with minmod slopes) → Euler projection. Each step consumes the state of cell j
// and emits the post-sweep state of cell j - LAG, LAG = 2 + [GAD] + [euler_2nd] = 2..4, i.e. exactly
// the dependency cone i-LAG..i+LAG of the reference. Every interface solve, flux,
// update, slope and advection is computed ONCE (the staged GAD kernel solves each interface three
// times, ref src/riemann_schemes.jl:63-80).
//
// History lives in rings indexed by the step's phase, a template parameter: callers unroll the march,
// every ring index is a compile-time constant and no register is ever moved to "shift" the history.
// The cell ring has 8 slots and doubles as the landing zone of the caller's prefetch: the state of
// cell j is loaded straight into slot j mod 8 (up to 4 steps ahead) and stays there while the cell is
// the "current", "previous" and "second previous" cell of the pipeline. A slot is only reloaded after
// its last use, so no live value is ever copied (a reload overlapping a live value would force the
// compiler to copy registers at the loop back-edge, and those copies wait for the loads in flight).
// The other rings have 4 slots (2 for slopes / advection fluxes), indexed by the phase mod 4.
//
// Two arithmetic flavours with the same interface:
//  * Pipe      — EXACT: the same IEEE operations in the same order as the staged kernels and the
//                reference formulas (bit-identical results; needs -ffp-contract=off).
//  * PipeFast  — tuned: shared 1-ulp reciprocals (v_rcp_f64 + 2 Newton steps) instead of IEEE
//                divisions, explicit FMAs, algebraically equivalent regrouping. Results agree with
//                the exact path to rounding (tests state the tolerance).
#pragma once

#include "physics.hpp"

namespace armon {
namespace fused {

template <typename T> struct Out4 { T rho, ua, ut, E; };

template <int SCHEME_, int LIM_, int PROJ_, int EOS_>
struct PipeTraits {
    static constexpr int SCHEME = SCHEME_, LIM = LIM_, PROJ = PROJ_, EOS = EOS_;
    static constexpr int S = (SCHEME == ARMON_SCHEME_GAD) ? 1 : 0;
    static constexpr int W = (PROJ == ARMON_PROJECTION_EULER_2ND) ? 1 : 0;
    static constexpr int LAG = S + W + 2;
};

// ======================================================================================================
// EXACT pipeline
// ======================================================================================================
template <int SCHEME, int LIM, int PROJ, int EOS, typename T = double>
struct Pipe : PipeTraits<SCHEME, LIM, PROJ, EOS> {
    using real = T;
    using TR = PipeTraits<SCHEME, LIM, PROJ, EOS>;
    static constexpr int S = TR::S, W = TR::W, LAG = TR::LAG;
    static constexpr bool kExact = true;

    using Den = xct::Den<T>;
    struct Cell { T rho, ua, ut, E, p, rc; };                 // pre-sweep state + EOS
    struct Upd { T rho, ua, ut, E, q_ua, q_ut, q_E; Den dxl; };   // Lagrangian (post cell_update) state; dxl prepared as a denominator

    T dt, dx, gamma;
    Den d_dx;                   // the uniform denominator dx
    Den dsum[2];                // (dxl[cu-1] + dxl[cu]) of this / the previous step: r₊ of one cell, r₋ of the next
    Cell c[8];                  // ring of 8: cells j+4 .. j (prefetched), j-1, j-2
    T gus[4], gps[4];      // ring: first-order solutions at interfaces j, j-1, j-2
    T fus[4], fps[4];      // ring: final fluxes at interfaces nf .. nf-3
    Upd l[4];                   // ring: updated cells cu, cu-1, cu-2   (cu = nf - 1)
    T s[2][4];             // minmod slopes of cells cu-1 / cu-2
    T a[2][4];             // advection fluxes at interfaces na / na-1
    T csr[4];              // ring: sound speed of cells j .. j-3 (dt/CFL tracking only)

    // (the two quotients the tuned pipeline takes from its caller are not used here: the exact flavour prepares denominators)
    __device__ __forceinline__ Pipe(T dt_, T dx_, T gamma_, T /*inv_dx*/, T /*dt_dx*/) : Pipe(dt_, dx_, gamma_) {}
    __device__ __forceinline__ Pipe(T dt_, T dx_, T gamma_) : dt(dt_), dx(dx_), gamma(gamma_), d_dx(dx_)
    {
        dsum[0] = dsum[1] = Den(dx_ + dx_);
        // Neutral, finite start values: the first 2*LAG outputs are discarded by the caller.
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = Cell{T(1.), T(0.), T(0.), T(1.), T(1.), T(1.)};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            gus[k] = T(0.); gps[k] = T(1.); fus[k] = T(0.); fps[k] = T(1.); csr[k] = T(1.);
            l[k] = Upd{T(1.), T(0.), T(0.), T(1.), T(0.), T(0.), T(1.), d_dx};
            s[0][k] = s[1][k] = T(0.);
            a[0][k] = a[1][k] = T(0.);
        }
    }

    // Feed cell j (pre-sweep state; Y_AXIS: ua is v and ut is u) and get the post-sweep state of cell
    // j - LAG. PH8 = step index mod 8. p_j, c_j: EOS of cell j, for optional materialisation.
    // c_lag: the (pre-sweep) sound speed of the emitted cell j - LAG, for the fused dt/CFL reduction.
    template <bool Y_AXIS, int PH8>
    __device__ __forceinline__ Out4<T> push(T rho, T ua, T ut, T E, T& p_j, T& c_j, T& c_lag)
    {
        Cell& n = c[PH8 & 7];
        n.rho = rho; n.ua = ua; n.ut = ut; n.E = E;
        return advance<Y_AXIS, PH8>(p_j, c_j, c_lag);
    }

    // Same, with (ρ, ua, ut, E) of cell j already stored in c[PH8 & 7] by the caller's prefetch.
    template <bool Y_AXIS, int PH8>
    __device__ __forceinline__ Out4<T> advance(T& p_j, T& c_j, T& c_lag)
    {
        constexpr int PH = PH8 & 3;
        constexpr int R0 = PH & 3, R1 = (PH - 1) & 3, R2 = (PH - 2) & 3, R3 = (PH - 3) & 3;
        constexpr int P0 = PH & 1, P1 = P0 ^ 1;
        c_lag = csr[(PH - LAG) & 3];      // LAG = 4: this very slot, read before it is overwritten below
        Cell& c0 = c[PH8 & 7];
        const Cell& c1 = c[(PH8 - 1) & 7];
        const Cell& c2 = c[(PH8 - 2) & 7];

        // ---- EOS (ref src/kernels.jl:4-55): e = E - 0.5*(u² + v²) with u, v in the reference's order
        T p, cs;
        {
            const T u = Y_AXIS ? c0.ut : c0.ua, v = Y_AXIS ? c0.ua : c0.ut;
            if (EOS == ARMON_EOS_BIZARRIUM) {
                T g_unused;
                phys::bizarrium<false>(c0.rho, c0.E, u, v, p, cs, g_unused);
            } else {
                phys::perfect_gas(gamma, c0.rho, c0.E, u, v, p, cs);
            }
        }
        p_j = p;
        c_j = cs;
        csr[R0] = cs;
        c0.p = p;
        c0.rc = c0.rho * cs;

        // ---- first-order acoustic solve at interface j (ref src/riemann_schemes.jl:21-30): one denominator
        {
            const T rc_l = c1.rc, rc_r = c0.rc;
            const Den d(rc_l + rc_r);
            gus[R0] = d.quo_t(rc_l * c1.ua + rc_r * c0.ua + (c1.p - c0.p));
            gps[R0] = d.quo(rc_r * c1.p + rc_l * c0.p + rc_l * rc_r * (c1.ua - c0.ua));
        }

        // ---- final flux at interface nf = j - S
        if (S == 1) {
            // acoustic_GAD! at interface i = j-1: cells i-s = c2, i = c1 (ref src/riemann_schemes.jl:84-104)
            const T gus0 = gus[R0], gps0 = gps[R0], gus1 = gus[R1], gps1 = gps[R1], gus2 = gus[R2], gps2 = gps[R2];
            const T r_um = phys::limiter<LIM>(Den(gus1 - c2.ua + T(1e-6)).quo_t(gus0 - c1.ua));
            const T r_pm = phys::limiter<LIM>(Den(gps1 - c2.p + T(1e-6)).quo(gps0 - c1.p));
            const T r_up = phys::limiter<LIM>(Den(c1.ua - gus1 + T(1e-6)).quo_t(c2.ua - gus2));
            const T r_pp = phys::limiter<LIM>(Den(c1.p - gps1 + T(1e-6)).quo(c2.p - gps2));
            const T dm_l = c2.rho * dx;
            const T dm_r = c1.rho * dx;
            const T Dm = (dm_l + dm_r) / 2;
            const T theta = T(0.5) * (1 - (c2.rc + c1.rc) / 2 * Den(Dm).quo(dt));
            fus[R0] = gus1 + theta * (r_up * (c1.ua - gus1) - r_um * (gus1 - c2.ua));
            fps[R0] = gps1 + theta * (r_pp * (c1.p - gps1) - r_pm * (gps1 - c2.p));
        } else {
            fus[R0] = gus[R0];
            fps[R0] = gps[R0];
        }
        const T fus0 = fus[R0], fps0 = fps[R0], fus1 = fus[R1], fps1 = fps[R1], fus2 = fus[R2], fus3 = fus[R3];

        // ---- Lagrangian update of cell cu = nf - 1 (ref src/kernels.jl:58-68)
        {
            const Cell& cc = (S == 1) ? c2 : c1;
            Upd& n = l[R0];
            const T dm = cc.rho * dx;
            n.dxl = Den(dx + dt * (fus0 - fus1));
            n.rho = n.dxl.quo(dm);
            const T dt_dm = Den(dm).quo(dt);
            n.ua = cc.ua + dt_dm * (fps1 - fps0);
            n.ut = cc.ut;
            n.E = cc.E + dt_dm * (fps1 * fus1 - fps0 * fus0);
            n.q_ua = n.rho * n.ua;
            n.q_ut = n.rho * n.ut;
            n.q_E = n.rho * n.E;
        }
        const Upd& l0 = l[R0];
        const Upd& l1 = l[R1];
        const Upd& l2 = l[R2];

        // ---- advection flux at interface na (→ a[P0]; a[P1] holds interface na-1), projection of na-1
        if (W == 1) {
            // slopes of cell cu-1 (ref src/projection_schemes.jl:105-116 evaluated per donor cell);
            // s[P1] still holds the slopes of cell cu-2 from the previous step.
            // r₋ = 2Δx/(Δx + Δx₋), r₊ = 2Δx/(Δx + Δx₊): this cell's Δx + Δx₊ is the next cell's Δx₋ + Δx (same bits:
            // IEEE addition commutes), so each sum is prepared once, by the step that first needs it
            dsum[P0] = Den(l1.dxl.b + l0.dxl.b);
            const T r_m = dsum[P1].quo(2 * l1.dxl.b);
            const T r_p = dsum[P0].quo(2 * l1.dxl.b);
            // reference order of the conserved quantities: ρ, ρu, ρv, ρE
            s[P0][0] = phys::slope_minmod(l2.rho, l1.rho, l0.rho, r_m, r_p);
            s[P0][Y_AXIS ? 2 : 1] = phys::slope_minmod(l2.q_ua, l1.q_ua, l0.q_ua, r_m, r_p);
            s[P0][Y_AXIS ? 1 : 2] = phys::slope_minmod(l2.q_ut, l1.q_ut, l0.q_ut, r_m, r_p);
            s[P0][3] = phys::slope_minmod(l2.q_E, l1.q_E, l0.q_E, r_m, r_p);
            // interface is = cu-1 (ref :92-124): upwind donor cell and its slopes
            const T disp = dt * fus2;
            const bool up = disp > 0;
            const Upd& d = up ? l2 : l1;
            const T Dxe = up ? -(dx - dt * fus3) : (dx + dt * fus1);
            const T lf = d.dxl.twice().quo(Dxe);
            const T q1 = Y_AXIS ? d.q_ut : d.q_ua, q2 = Y_AXIS ? d.q_ua : d.q_ut;
            a[P0][0] = disp * (d.rho - (up ? s[P1][0] : s[P0][0]) * lf);
            a[P0][1] = disp * (q1 - (up ? s[P1][1] : s[P0][1]) * lf);
            a[P0][2] = disp * (q2 - (up ? s[P1][2] : s[P0][2]) * lf);
            a[P0][3] = disp * (d.q_E - (up ? s[P1][3] : s[P0][3]) * lf);
            return project<Y_AXIS, P0>(l2);
        } else {
            // advection_first_order! at interface is = cu (ref src/projection_schemes.jl:62-78)
            const T disp = dt * fus1;
            const Upd& d = (disp > 0) ? l1 : l0;
            const T q1 = Y_AXIS ? d.q_ut : d.q_ua, q2 = Y_AXIS ? d.q_ua : d.q_ut;
            a[P0][0] = disp * d.rho;
            a[P0][1] = disp * q1;
            a[P0][2] = disp * q2;
            a[P0][3] = disp * d.q_E;
            return project<Y_AXIS, P0>(l1);
        }
    }

    // euler_projection! of cell o with A_o = a[P1], A_{o+1} = a[P0] (ref src/projection_schemes.jl:23-41)
    template <bool Y_AXIS, int P0>
    __device__ __forceinline__ Out4<T> project(const Upd& lo) const
    {
        constexpr int P1 = P0 ^ 1;
        const T dX = lo.dxl.b;
        const T u = Y_AXIS ? lo.ut : lo.ua, v = Y_AXIS ? lo.ua : lo.ut;   // reference's (u, v)
        const T t_rho  = d_dx.quo(dX * lo.rho        - (a[P0][0] - a[P1][0]));
        const T t_urho = d_dx.quo_t(dX * lo.rho * u    - (a[P0][1] - a[P1][1]));
        const T t_vrho = d_dx.quo_t(dX * lo.rho * v    - (a[P0][2] - a[P1][2]));
        const T t_Erho = d_dx.quo(dX * lo.rho * lo.E - (a[P0][3] - a[P1][3]));
        Out4<T> o;
        o.rho = t_rho;
        const Den d_rho(t_rho);
        const T un = d_rho.quo_t(t_urho), vn = d_rho.quo_t(t_vrho);
        o.ua = Y_AXIS ? vn : un;
        o.ut = Y_AXIS ? un : vn;
        o.E = d_rho.quo(t_Erho);
        return o;
    }
};

// ======================================================================================================
// tuned pipeline
// ======================================================================================================
namespace fast {

// 1/x to ≈1 ulp: v_rcp_f64 (2^-24) + two Newton steps. No scaling/fixup: operands here are O(1) state.
__device__ __forceinline__ double rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32: 1 ulp

// 1/x to ≈2e-15 (one Newton step): for the reciprocals that only feed second-order correction terms
// (limited GAD ratios, θ, slope ratios), where a relative error of 1e-15 is far below their truncation error.
__device__ __forceinline__ double rcp1(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}
__device__ __forceinline__ float rcp1(float x) { return __builtin_amdgcn_rcpf(x); }

// sqrt(x) to ≈1-2 ulp: v_rsq_f64 + two coupled Newton (Goldschmidt) steps; sqrt(0) = 0.
__device__ __forceinline__ double sqrt_(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-g, h, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-g, h, 0.5);
    g = __builtin_fma(g, r, g);
    return (x == 0.) ? 0. : g;
}
__device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }      // v_sqrt_f32: 1 ulp, no denormal scaling (operands are O(1) energies)

// Two fp32 cells per lane as ONE 64-bit value (the tuned fp32 Y march, k_sweep_y2): elementwise arithmetic on this type is
// v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 — one instruction for both cells — where two scalar pipelines side by side
// left the pairing to the compiler's SLP pass (which found about half of it and paid for the rest in v_mov shuffles).
// Same IEEE operations per component, hence the same bits as the scalar pipeline.
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2v rcp(float2v x) { return float2v{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ float2v rcp1(float2v x) { return rcp(x); }
__device__ __forceinline__ float2v sqrt_(float2v x) { return float2v{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)}; }
__device__ __forceinline__ float2v fma_(float2v a, float2v b, float2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float2v max_(float2v a, float2v b) { return float2v{__builtin_fmaxf(a.x, b.x), __builtin_fmaxf(a.y, b.y)}; }
__device__ __forceinline__ float2v min_(float2v a, float2v b) { return float2v{__builtin_fminf(a.x, b.x), __builtin_fminf(a.y, b.y)}; }
// component-wise choice; `m` is the result of a comparison (bool for a scalar, an integer vector for float2v)
template <class M, class T> __device__ __forceinline__ T sel(M m, T a, T b) { return m ? a : b; }
// the scalar type behind a working type: uniform parameters (dt, dx, γ …) stay scalars (SGPRs) next to vector state
template <typename T> struct scalar_of { using type = T; };
template <> struct scalar_of<float2v> { using type = float; };

__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double max_(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float max_(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double min_(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float min_(float a, float b) { return __builtin_fminf(a, b); }

template <int LIM, typename T>
__device__ __forceinline__ T limiter(T r)
{
    if (LIM == ARMON_LIMITER_MINMOD) return max_(T(0.), min_(T(1.), r));
    if (LIM == ARMON_LIMITER_SUPERBEE) return max_(max_(T(0.), min_(T(2.) * r, T(1.))), min_(r, T(2.)));
    return T(1.);
}

// minmod(a, b) = sign·min(|a|,|b|) when a·b > 0, else 0  (== slope_minmod of ref projection_schemes.jl:15-20)
// = the median of (a, b, 0).
template <typename T>
__device__ __forceinline__ T minmod(T a, T b)
{
    return max_(min_(a, b), min_(max_(a, b), T(0.)));
}

// ref src/kernels.jl:16-55 with the four divisions by (1 - s·x) and the two by ρ shared, the polynomials in Horner
// form and every multiply-add an explicit FMA (the library is built with -ffp-contract=off: a plain a*b + c costs two
// instructions): 62 VALU instructions instead of 94.
template <typename T>
__device__ __forceinline__ void bizarrium(T rho, T ua, T ut, T E, T& p, T& cs)
{
    constexpr double rho0 = 10000., K0 = 1e+11, Cv0 = 1000., T0 = 300., eps0 = 0., G0 = 1.5, s_ = 1.5;
    constexpr double q = -42080895. / 14941154., rr = 727668333. / 149411540., a1 = s_ / 3. - 2.;
    const T inv_rho = rcp(rho);
    const T x = fma_(rho, T(1. / rho0), T(-1.));
    const T G = fma_(T(-G0 * rho0), inv_rho, T(G0));                         // G0 (1 - ρ0/ρ)
    const T x2 = x * x;
    const T opx = T(1.) + x, opx2 = opx * opx, opx3 = opx2 * opx;
    const T inv_d = rcp(fma_(T(-s_), x, T(1.)));
    const T f0 = fma_(fma_(fma_(T(rr), x, T(q)), x, T(a1)), x, T(1.)) * inv_d;                  // (1 + a1 x + q x² + r x³)/(1 - s x)
    const T f1 = fma_(T(s_), f0, fma_(fma_(T(3. * rr), x, T(2. * q)), x, T(a1))) * inv_d;       // (a1 + 2q x + 3r x² + s f0)/(1 - s x)
    const T f2 = fma_(T(2. * s_), f1, fma_(T(6. * rr), x, T(2. * q))) * inv_d;                  // (2q + 6r x + 2s f1)/(1 - s x)
    // ε_k0 = ε0 - Cv0 T0 (1 + G) + ½ (K0/ρ0) x² f0
    const T epsk0 = fma_(T(0.5 * K0 / rho0) * x2, f0, fma_(T(-Cv0 * T0), G, T(eps0 - Cv0 * T0)));
    // p_k0 = -Cv0 T0 G0 ρ0 + ½ K0 x (1+x)² (2 f0 + x f1)
    const T pk0 = fma_(T(0.5 * K0) * x * opx2, fma_(x, f1, T(2.) * f0), T(-Cv0 * T0 * G0 * rho0));
    // p_k0' = -½ K0 (1+x)³ ρ0 (2 (1+3x) f0 + 2x (2+3x) f1 + x² (1+x) f2)
    const T inner = fma_(fma_(T(6.), x, T(2.)), f0, fma_(x * fma_(T(6.), x, T(4.)), f1, x2 * opx * f2));
    const T pk0prime = T(-0.5 * K0 * rho0) * opx3 * inner;
    const T e = fma_(T(-0.5), fma_(ua, ua, ut * ut), E);
    const T w = T(G0 * rho0) * (e - epsk0);                                  // = p - p_k0
    p = pk0 + w;
    cs = sqrt_(fma_(T(G0 * rho0), w, -pk0prime)) * inv_rho;
}


}  // namespace fast

template <int SCHEME, int LIM, int PROJ, int EOS, typename T = double>
struct PipeFast : PipeTraits<SCHEME, LIM, PROJ, EOS> {
    using real = T;
    using TR = PipeTraits<SCHEME, LIM, PROJ, EOS>;
    static constexpr int S = TR::S, W = TR::W, LAG = TR::LAG;
    static constexpr bool kExact = false;

    struct Cell { T rho, ua, ut, E, p, rc; };
    // Lagrangian state: q_* = ρ·(ua, ut, E); hinv = 0.5/dxl; d_* = q(this) - q(previous cell)
    struct Upd { T rho, q_ua, q_ut, q_E, dxl, hinv, d_rho, d_ua, d_ut, d_E; };

    using Sc = typename fast::scalar_of<T>::type;     // uniform parameters
    Sc dt, dx, gamma, inv_dx, dt_dx, gm1, ggm1;
    Cell c[8];
    T gus[4], gps[4], src[4];          // first-order solutions + (rc_l + rc_r) of the interface
    T fps[4], dtu[4], pu[4];           // final flux: pˢ, dt·uˢ, pˢ·uˢ at interfaces nf .. nf-3
    Upd l[4];
    T isum[2];                         // 1 / (dxl[cu] + dxl[cu-1]) of this / the previous step
    T s[2][4];
    T a[2][4];
    T csr[4];

    __device__ __forceinline__ PipeFast(Sc dt_, Sc dx_, Sc gamma_) : PipeFast(dt_, dx_, gamma_, Sc(1.) / dx_, dt_ / dx_) {}
    // inv_dx_ = 1 / dx, dt_dx_ = dt / dx as IEEE quotients in the run's precision (the host forms them once per launch)
    __device__ __forceinline__ PipeFast(Sc dt_, Sc dx_, Sc gamma_, Sc inv_dx_, Sc dt_dx_) : dt(dt_), dx(dx_), gamma(gamma_)
    {
        inv_dx = inv_dx_;
        dt_dx = dt_dx_;
        gm1 = gamma_ - Sc(1.);
        ggm1 = gamma_ * (gamma_ - Sc(1.));
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = Cell{T(1.), T(0.), T(0.), T(1.), T(1.), T(1.)};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            gus[k] = T(0.); gps[k] = T(1.); src[k] = T(2.);
            fps[k] = T(1.); dtu[k] = T(0.); pu[k] = T(0.); csr[k] = T(1.);
            l[k] = Upd{T(1.), T(0.), T(0.), T(1.), T(dx_), T(Sc(0.5) / dx_), T(0.), T(0.), T(0.), T(0.)};
            s[0][k] = s[1][k] = T(0.);
            a[0][k] = a[1][k] = T(0.);
        }
        isum[0] = isum[1] = T(Sc(0.5) / dx_);
    }

    template <bool Y_AXIS, int PH8>
    __device__ __forceinline__ Out4<T> push(T rho, T ua, T ut, T E, T& p_j, T& c_j, T& c_lag)
    {
        Cell& n = c[PH8 & 7];
        n.rho = rho; n.ua = ua; n.ut = ut; n.E = E;
        return advance<Y_AXIS, PH8>(p_j, c_j, c_lag);
    }

    template <bool Y_AXIS, int PH8>
    __device__ __forceinline__ Out4<T> advance(T& p_j, T& c_j, T& c_lag)
    {
        using namespace fast;
        constexpr int PH = PH8 & 3;
        constexpr int R0 = PH & 3, R1 = (PH - 1) & 3, R2 = (PH - 2) & 3, R3 = (PH - 3) & 3;
        constexpr int P0 = PH & 1, P1 = P0 ^ 1;
        c_lag = csr[(PH - LAG) & 3];
        Cell& c0 = c[PH8 & 7];
        const Cell& c1 = c[(PH8 - 1) & 7];
        const Cell& c2 = c[(PH8 - 2) & 7];

        // ---- EOS
        T p, cs;
        if (EOS == ARMON_EOS_BIZARRIUM) {
            fast::bizarrium(c0.rho, c0.ua, c0.ut, c0.E, p, cs);
        } else {
            // p = (γ-1)ρe, c = sqrt(γp/ρ) = sqrt(γ(γ-1)e): no division
            const T e = fma_(T(-0.5), fma_(c0.ua, c0.ua, c0.ut * c0.ut), c0.E);
            p = gm1 * c0.rho * e;
            cs = sqrt_(ggm1 * e);
        }
        p_j = p;
        c_j = cs;
        csr[R0] = cs;
        c0.p = p;
        c0.rc = c0.rho * cs;

        // ---- first-order acoustic solve at interface j: one shared reciprocal
        {
            const T rc_l = c1.rc, rc_r = c0.rc;
            const T sum = rc_l + rc_r;
            const T inv = rcp(sum);
            src[R0] = sum;
            gus[R0] = fma_(rc_l, c1.ua, fma_(rc_r, c0.ua, c1.p - c0.p)) * inv;
            gps[R0] = fma_(rc_r, c1.p, fma_(rc_l, c0.p, rc_l * rc_r * (c1.ua - c0.ua))) * inv;
        }

        // ---- final flux at interface nf = j - S
        T fus0;
        if (S == 1) {
            const T gus0 = gus[R0], gps0 = gps[R0], gus1 = gus[R1], gps1 = gps[R1], gus2 = gus[R2], gps2 = gps[R2];
            const T Au = gus1 - c2.ua, Bu = c1.ua - gus1;     // (uˢ_i - u[i-s]), (u[i] - uˢ_i)
            const T Ap = gps1 - c2.p, Bp = c1.p - gps1;
            const T r_um = limiter<LIM>((gus0 - c1.ua) * rcp1(Au + T(1e-6)));
            const T r_pm = limiter<LIM>((gps0 - c1.p) * rcp1(Ap + T(1e-6)));
            const T r_up = limiter<LIM>((c2.ua - gus2) * rcp1(Bu + T(1e-6)));
            const T r_pp = limiter<LIM>((c2.p - gps2) * rcp1(Bp + T(1e-6)));
            // θ = ½(1 - (rc_l+rc_r)/2 · dt/Dm), Dm = dx(ρ_l+ρ_r)/2  →  ½ - ½·(rc_l+rc_r)·(dt/dx)/(ρ_l+ρ_r)
            const T theta = fma_(T(-0.5) * src[R1] * dt_dx, rcp1(c2.rho + c1.rho), T(0.5));
            fus0 = fma_(theta, fma_(r_up, Bu, -r_um * Au), gus1);
            fps[R0] = fma_(theta, fma_(r_pp, Bp, -r_pm * Ap), gps1);
        } else {
            fus0 = gus[R0];
            fps[R0] = gps[R0];
        }
        dtu[R0] = dt * fus0;
        pu[R0] = fps[R0] * fus0;
        const T fps0 = fps[R0], fps1 = fps[R1];

        // ---- Lagrangian update of cell cu = nf - 1
        {
            const Cell& cc = (S == 1) ? c2 : c1;
            Upd& n = l[R0];
            const Upd& prev = l[R1];
            const T dtdm = dt_dx * rcp(cc.rho);             // dt / (ρ dx)
            const T dxl = (dx + dtu[R0]) - dtu[R1];
            const T inv_dxl = rcp(dxl);
            n.dxl = dxl;
            n.hinv = T(0.5) * inv_dxl;
            n.rho = cc.rho * dx * inv_dxl;
            const T ua_n = fma_(dtdm, fps1 - fps0, cc.ua);
            const T E_n = fma_(dtdm, pu[R1] - pu[R0], cc.E);
            n.q_ua = n.rho * ua_n;
            n.q_ut = n.rho * cc.ut;
            n.q_E = n.rho * E_n;
            n.d_rho = n.rho - prev.rho;
            n.d_ua = n.q_ua - prev.q_ua;
            n.d_ut = n.q_ut - prev.q_ut;
            n.d_E = n.q_E - prev.q_E;
        }
        const Upd& l0 = l[R0];
        const Upd& l1 = l[R1];
        const Upd& l2 = l[R2];

        if (W == 1) {
            // slopes of cell cu-1: r₊ = 2Δx/(Δx+Δx₊), r₋ = 2Δx/(Δx+Δx₋); the second sum is last step's first
            isum[P0] = rcp1(l1.dxl + l0.dxl);
            const T two_dxl = T(2.) * l1.dxl;
            const T r_p = two_dxl * isum[P0], r_m = two_dxl * isum[P1];
            s[P0][0] = minmod(r_p * l0.d_rho, r_m * l1.d_rho);
            s[P0][Y_AXIS ? 2 : 1] = minmod(r_p * l0.d_ua, r_m * l1.d_ua);
            s[P0][Y_AXIS ? 1 : 2] = minmod(r_p * l0.d_ut, r_m * l1.d_ut);
            s[P0][3] = minmod(r_p * l0.d_E, r_m * l1.d_E);
            // interface is = cu-1
            const T disp = dtu[R2];
            const auto up = disp > T(0.);          // donor cell: cu-2 when the interface moves up, else cu-1 (per component)
            const T Dxe = sel(up, dtu[R3] - dx, dx + dtu[R1]);
            const T lf = Dxe * sel(up, l2.hinv, l1.hinv);
            const T d_q1 = Y_AXIS ? sel(up, l2.q_ut, l1.q_ut) : sel(up, l2.q_ua, l1.q_ua);
            const T d_q2 = Y_AXIS ? sel(up, l2.q_ua, l1.q_ua) : sel(up, l2.q_ut, l1.q_ut);
            a[P0][0] = disp * fma_(-sel(up, s[P1][0], s[P0][0]), lf, sel(up, l2.rho, l1.rho));
            a[P0][1] = disp * fma_(-sel(up, s[P1][1], s[P0][1]), lf, d_q1);
            a[P0][2] = disp * fma_(-sel(up, s[P1][2], s[P0][2]), lf, d_q2);
            a[P0][3] = disp * fma_(-sel(up, s[P1][3], s[P0][3]), lf, sel(up, l2.q_E, l1.q_E));
            return project<Y_AXIS, P0>(l2);
        } else {
            const T disp = dtu[R1];
            const auto up = disp > T(0.);
            const T d_q1 = Y_AXIS ? sel(up, l1.q_ut, l0.q_ut) : sel(up, l1.q_ua, l0.q_ua);
            const T d_q2 = Y_AXIS ? sel(up, l1.q_ua, l0.q_ua) : sel(up, l1.q_ut, l0.q_ut);
            a[P0][0] = disp * sel(up, l1.rho, l0.rho);
            a[P0][1] = disp * d_q1;
            a[P0][2] = disp * d_q2;
            a[P0][3] = disp * sel(up, l1.q_E, l0.q_E);
            return project<Y_AXIS, P0>(l1);
        }
    }

    // (dX·q - ΔA)/dx, then u = ρu/ρ …: the 1/dx cancels in the three ratios
    template <bool Y_AXIS, int P0>
    __device__ __forceinline__ Out4<T> project(const Upd& lo) const
    {
        using namespace fast;
        constexpr int P1 = P0 ^ 1;
        const T dX = lo.dxl;
        const T q1 = Y_AXIS ? lo.q_ut : lo.q_ua, q2 = Y_AXIS ? lo.q_ua : lo.q_ut;
        const T T_rho = fma_(dX, lo.rho, a[P1][0] - a[P0][0]);
        const T T_u = fma_(dX, q1, a[P1][1] - a[P0][1]);
        const T T_v = fma_(dX, q2, a[P1][2] - a[P0][2]);
        const T T_E = fma_(dX, lo.q_E, a[P1][3] - a[P0][3]);
        const T inv = rcp(T_rho);
        Out4<T> o;
        o.rho = T_rho * inv_dx;
        const T un = T_u * inv, vn = T_v * inv;
        o.ua = Y_AXIS ? vn : un;
        o.ut = Y_AXIS ? un : vn;
        o.E = T_E * inv;
        return o;
    }

};

}  // namespace fused
}  // namespace armon
