// sweep_pipeline.hpp — the fused directional sweep as a register-resident software pipeline.
//
// One lane marches along the sweep axis, one cell per step, and keeps in registers everything the
// reference passes between its five per-sweep kernels through memory (p, c, uˢ, pˢ, work_1..4 —
// ref src/solver.jl:300-316): EOS → first-order interface solve → GAD limited flux → Lagrangian cell
// update → advection (with minmod slopes) → Euler projection. Each step consumes the state of cell j
// and emits the post-sweep state of cell j - LAG, LAG = stencil(scheme) + stencil(projection) - ... =
// 2..4, i.e. exactly the dependency cone i-LAG..i+LAG of the reference.
// Every interface solve, flux, update, slope and advection is computed ONCE (the staged GAD kernel
// solves each interface three times, ref src/riemann_schemes.jl:63-80).
//
// Arithmetic: the EXACT instantiation evaluates the same IEEE operations in the same order as the
// staged kernels / the reference formulas, so results are bit-identical; the tuned instantiation shares
// reciprocals and uses FMAs (see fast:: below).
#pragma once

#include "physics.hpp"

namespace armon {
namespace fused {

struct Out4 { double rho, ua, ut, E; };

template <int SCHEME, int LIM, int PROJ, int EOS, bool EXACT>
struct Pipe {
    static constexpr int S = (SCHEME == ARMON_SCHEME_GAD) ? 1 : 0;
    static constexpr int W = (PROJ == ARMON_PROJECTION_EULER_2ND) ? 1 : 0;
    static constexpr int LAG = S + W + 2;

    struct Cell { double rho, ua, ut, E, p, rc; };          // pre-sweep state + EOS
    struct Upd { double rho, ua, ut, E, q_ua, q_ut, q_E, dxl; };   // Lagrangian (post cell_update) state

    double dt, dx, gamma;
    Cell c0, c1, c2;                              // cells j, j-1, j-2
    double gus0, gps0, gus1, gps1, gus2, gps2;    // first-order solutions at interfaces j, j-1, j-2
    double fus0, fps0, fus1, fps1, fus2, fus3;    // final fluxes at interfaces nf, nf-1 (+ older uˢ)
    Upd l0, l1, l2;                               // updated cells cu, cu-1, cu-2   (cu = nf - 1)
    double s0[4], s1[4];                          // minmod slopes of cells cu-1, cu-2
    double a0[4], a1[4];                          // advection fluxes at interfaces na, na-1

    __device__ __forceinline__ Pipe(double dt_, double dx_, double gamma_) : dt(dt_), dx(dx_), gamma(gamma_)
    {
        // Neutral, finite start values: the first LAG*2 outputs are discarded by the caller.
        const Cell cz = {1., 0., 0., 1., 1., 1.};
        c0 = c1 = c2 = cz;
        gus0 = gus1 = gus2 = 0.; gps0 = gps1 = gps2 = 1.;
        fus0 = fus1 = fus2 = fus3 = 0.; fps0 = fps1 = 1.;
        const Upd lz = {1., 0., 0., 1., 0., 0., 1., dx_};
        l0 = l1 = l2 = lz;
        for (int k = 0; k < 4; k++) { s0[k] = s1[k] = 0.; a0[k] = a1[k] = 0.; }
    }

    // EOS of the incoming cell; p and c are also returned for optional materialisation.
    __device__ __forceinline__ void eos(double rho, double ua, double ut, double E, double& p, double& c) const
    {
        if (EOS == ARMON_EOS_BIZARRIUM) {
            double g_unused;
            // u² + v² is evaluated as u*u + v*v in the reference: keep (u, v) order for bit parity.
            phys::bizarrium<false>(rho, E, ua, ut, p, c, g_unused);
        } else {
            phys::perfect_gas(gamma, rho, E, ua, ut, p, c);
        }
    }

    // Feed cell j (pre-sweep state; `uv_swapped`: ua is v and ut is u, i.e. a Y sweep) and get the
    // post-sweep state of cell j - LAG.
    template <bool Y_AXIS>
    __device__ __forceinline__ Out4 push(double rho, double ua, double ut, double E, double& p_j, double& c_j)
    {
        // ---- EOS (ref src/kernels.jl:4-55): e = E - 0.5*(u² + v²) with u, v in the reference's order
        double p, c;
        if (Y_AXIS) eos(rho, ut, ua, E, p, c); else eos(rho, ua, ut, E, p, c);
        p_j = p;
        c_j = c;
        c2 = c1;
        c1 = c0;
        c0 = Cell{rho, ua, ut, E, p, rho * c};

        // ---- first-order acoustic solve at interface j (ref src/riemann_schemes.jl:21-30)
        gus2 = gus1; gps2 = gps1;
        gus1 = gus0; gps1 = gps0;
        {
            const double rc_l = c1.rc, rc_r = c0.rc;
            gus0 = (rc_l * c1.ua + rc_r * c0.ua + (c1.p - c0.p)) / (rc_l + rc_r);
            gps0 = (rc_r * c1.p + rc_l * c0.p + rc_l * rc_r * (c1.ua - c0.ua)) / (rc_l + rc_r);
        }

        // ---- final flux at interface nf = j - S
        fus3 = fus2;
        fus2 = fus1;
        fus1 = fus0; fps1 = fps0;
        if (S == 1) {
            // acoustic_GAD! at interface i = j-1: cells i-s = c2, i = c1 (ref src/riemann_schemes.jl:84-104)
            const double r_um = phys::limiter<LIM>((gus0 - c1.ua) / (gus1 - c2.ua + 1e-6));
            const double r_pm = phys::limiter<LIM>((gps0 - c1.p) / (gps1 - c2.p + 1e-6));
            const double r_up = phys::limiter<LIM>((c2.ua - gus2) / (c1.ua - gus1 + 1e-6));
            const double r_pp = phys::limiter<LIM>((c2.p - gps2) / (c1.p - gps1 + 1e-6));
            const double dm_l = c2.rho * dx;
            const double dm_r = c1.rho * dx;
            const double Dm = (dm_l + dm_r) / 2;
            const double theta = 0.5 * (1 - (c2.rc + c1.rc) / 2 * (dt / Dm));
            fus0 = gus1 + theta * (r_up * (c1.ua - gus1) - r_um * (gus1 - c2.ua));
            fps0 = gps1 + theta * (r_pp * (c1.p - gps1) - r_pm * (gps1 - c2.p));
        } else {
            fus0 = gus0;
            fps0 = gps0;
        }

        // ---- Lagrangian update of cell cu = nf - 1 (ref src/kernels.jl:58-68)
        l2 = l1;
        l1 = l0;
        {
            const Cell& cc = (S == 1) ? c2 : c1;
            const double dm = cc.rho * dx;
            const double dxl = dx + dt * (fus0 - fus1);
            l0.dxl = dxl;
            l0.rho = dm / dxl;
            l0.ua = cc.ua + dt / dm * (fps1 - fps0);
            l0.ut = cc.ut;
            l0.E = cc.E + dt / dm * (fps1 * fus1 - fps0 * fus0);
            l0.q_ua = l0.rho * l0.ua;
            l0.q_ut = l0.rho * l0.ut;
            l0.q_E = l0.rho * l0.E;
        }

        // ---- advection flux at interface na, projection of cell na - 1
        for (int k = 0; k < 4; k++) a1[k] = a0[k];
        Out4 out;
        if (W == 1) {
            // slopes of cell cu-1 (ref src/projection_schemes.jl:105-116 evaluated per donor cell)
            for (int k = 0; k < 4; k++) s1[k] = s0[k];
            const double r_m = (2 * l1.dxl) / (l1.dxl + l2.dxl);
            const double r_p = (2 * l1.dxl) / (l1.dxl + l0.dxl);
            // reference order of the four conserved quantities: ρ, ρu, ρv, ρE
            if (Y_AXIS) {
                s0[0] = phys::slope_minmod(l2.rho, l1.rho, l0.rho, r_m, r_p);
                s0[1] = phys::slope_minmod(l2.q_ut, l1.q_ut, l0.q_ut, r_m, r_p);
                s0[2] = phys::slope_minmod(l2.q_ua, l1.q_ua, l0.q_ua, r_m, r_p);
                s0[3] = phys::slope_minmod(l2.q_E, l1.q_E, l0.q_E, r_m, r_p);
            } else {
                s0[0] = phys::slope_minmod(l2.rho, l1.rho, l0.rho, r_m, r_p);
                s0[1] = phys::slope_minmod(l2.q_ua, l1.q_ua, l0.q_ua, r_m, r_p);
                s0[2] = phys::slope_minmod(l2.q_ut, l1.q_ut, l0.q_ut, r_m, r_p);
                s0[3] = phys::slope_minmod(l2.q_E, l1.q_E, l0.q_E, r_m, r_p);
            }
            // interface is = cu-1 (ref :92-124): upwind donor cell and its slopes
            const double disp = dt * fus2;
            const bool up = disp > 0;
            const Upd& d = up ? l2 : l1;
            const double Dxe = up ? -(dx - dt * fus3) : (dx + dt * fus1);
            const double lf = Dxe / (2 * d.dxl);
            const double q1 = Y_AXIS ? d.q_ut : d.q_ua, q2 = Y_AXIS ? d.q_ua : d.q_ut;
            a0[0] = disp * (d.rho - (up ? s1[0] : s0[0]) * lf);
            a0[1] = disp * (q1 - (up ? s1[1] : s0[1]) * lf);
            a0[2] = disp * (q2 - (up ? s1[2] : s0[2]) * lf);
            a0[3] = disp * (d.q_E - (up ? s1[3] : s0[3]) * lf);
            out = project<Y_AXIS>(l2);
        } else {
            // advection_first_order! at interface is = cu (ref src/projection_schemes.jl:62-78)
            const double disp = dt * fus1;
            const Upd& d = (disp > 0) ? l1 : l0;
            const double q1 = Y_AXIS ? d.q_ut : d.q_ua, q2 = Y_AXIS ? d.q_ua : d.q_ut;
            a0[0] = disp * d.rho;
            a0[1] = disp * q1;
            a0[2] = disp * q2;
            a0[3] = disp * d.q_E;
            out = project<Y_AXIS>(l1);
        }
        return out;
    }

    // euler_projection! of cell o with A_o = a1, A_{o+1} = a0 (ref src/projection_schemes.jl:23-41)
    template <bool Y_AXIS>
    __device__ __forceinline__ Out4 project(const Upd& l) const
    {
        const double dX = l.dxl;
        const double u = Y_AXIS ? l.ut : l.ua, v = Y_AXIS ? l.ua : l.ut;   // reference's (u, v)
        const double t_rho  = (dX * l.rho       - (a0[0] - a1[0])) / dx;
        const double t_urho = (dX * l.rho * u   - (a0[1] - a1[1])) / dx;
        const double t_vrho = (dX * l.rho * v   - (a0[2] - a1[2])) / dx;
        const double t_Erho = (dX * l.rho * l.E - (a0[3] - a1[3])) / dx;
        Out4 o;
        o.rho = t_rho;
        const double un = t_urho / t_rho, vn = t_vrho / t_rho;
        o.ua = Y_AXIS ? vn : un;
        o.ut = Y_AXIS ? un : vn;
        o.E = t_Erho / t_rho;
        return o;
    }
};

}  // namespace fused
}  // namespace armon
