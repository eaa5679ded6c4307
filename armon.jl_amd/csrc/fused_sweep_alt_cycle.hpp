// fused_sweep_alt_cycle.hpp — armon_hip_cycle_xy, the entry point of the whole-cycle kernels of the A/B build
// (-DARMON_ALT_KERNELS; fused_sweep_alt_kernels.hpp). Included by fused_sweep_impl.hpp at file scope, after armon_hip_sweep.

extern "C" int ARMON_CYCLE_FN(armon_ctx* ctx, const ARMON_SWEEP_DESC* x, const ARMON_SWEEP_DESC* y)
{
    ARMON_REQUIRE(ctx && x && y, "NULL argument");
    ARMON_REQUIRE(x->axis == ARMON_AXIS_X && y->axis == ARMON_AXIS_Y, "expected an X descriptor then a Y descriptor");
    ARMON_REQUIRE(x->scheme == y->scheme && x->limiter == y->limiter && x->projection == y->projection && x->eos == y->eos &&
                  x->exact == y->exact && x->nghost == y->nghost && x->nx == y->nx && x->ny == y->ny && x->gamma == y->gamma,
                  "the two sweeps of a cycle must share scheme, limiter, projection, EOS, arithmetic and block shape");
    ARMON_REQUIRE(x->scheme == ARMON_SCHEME_GAD && x->limiter == ARMON_LIMITER_MINMOD && x->projection == ARMON_PROJECTION_EULER_2ND &&
                  x->eos == ARMON_EOS_PERFECT_GAS && !x->exact, "whole-cycle kernel: GAD + minmod + euler_2nd, perfect gas, tuned arithmetic only");
    ARMON_REQUIRE(x->nx > 0 && x->ny > 0 && x->nx < (1ll << 30) && x->ny < (1ll << 30), "invalid block %lld x %lld", (long long)x->nx, (long long)x->ny);
    ARMON_REQUIRE(x->out_hi == 0 && !x->p_out && !x->c_out && !x->dt_cfl_out, "partial X sweeps / X outputs are not available in the whole-cycle kernel");
    ARMON_REQUIRE(x->x_kernel == 0 || x->x_kernel == 4, "unknown form %d of the whole-cycle kernel", x->x_kernel);
    ARMON_REQUIRE(!x->dt_state && !y->dt_state, "the whole-cycle kernels do not read a device-resident time step (dt_state)");
    ARMON_REQUIRE(x->rho_in && x->u_in && x->v_in && x->E_in && y->rho_out && y->u_out && y->v_out && y->E_out, "NULL state array");
    ARMON_REQUIRE(x->rho_in != y->rho_out && x->u_in != y->u_out && x->v_in != y->v_out && x->E_in != y->E_out, "in and out arrays must not alias");
    constexpr int lag = 4;
    ARMON_REQUIRE(x->nghost >= lag && x->nx >= lag && x->ny >= lag, "needs at least %d ghost layers and cells per axis", lag);
    const bool track = y->dt_cfl_out != nullptr;
    ARMON_REQUIRE(!track || (y->cfl_dx > 0 && y->cfl_dy > 0), "dt_cfl_out needs cfl_dx, cfl_dy > 0");

    sweep_args a;
    a.nx = x->nx;
    a.ny = x->ny;
    a.row_len = x->nx + 2 * (int64_t)x->nghost;
    a.g = x->nghost;
    a.bc_low = x->bc_low;
    a.bc_high = x->bc_high;
    a.emit = y->p_out ? 1 : 0;
    a.dt = (real)x->dt;
    a.dx = (real)x->dx;
    a.gamma = (real)x->gamma;
    a.inv_dx = real(1) / a.dx;
    a.dt_dx = a.dt / a.dx;
    a.fa_low = (real)x->u_factor_low;     // X sweep: axial = u, transverse = v
    a.ft_low = (real)x->v_factor_low;
    a.fa_high = (real)x->u_factor_high;
    a.ft_high = (real)x->v_factor_high;
    a.bc_low_t = y->bc_low;
    a.bc_high_t = y->bc_high;
    a.tu_low = (real)y->u_factor_low;
    a.tv_low = (real)y->v_factor_low;
    a.tu_high = (real)y->u_factor_high;
    a.tv_high = (real)y->v_factor_high;
    a.rho_in = x->rho_in; a.ua_in = x->u_in; a.ut_in = x->v_in; a.E_in = x->E_in;
    a.rho_out = y->rho_out; a.ua_out = y->u_out; a.ut_out = y->v_out; a.E_out = y->E_out;
    a.p_out = y->p_out;
    a.c_out = nullptr;
    a.x_kernel = 0;
    a.xshift = 0;
    a.xcd_remap = 0;
    a.o_lo = 0;
    a.o_hi = x->ny;
    if (y->out_hi != 0) {
        ARMON_REQUIRE(y->out_lo >= 0 && y->out_lo < y->out_hi && y->out_hi <= x->ny, "invalid partial sweep");
        a.o_lo = y->out_lo;
        a.o_hi = y->out_hi;
    }
    const bool pc_form = x->x_kernel == 4;                   // 0: one wave does both stages (k_cycle_xy); 4: producer / consumer waves
    a.x_first = pc_form ? -(int64_t)(x->nghost % 8) : 0;     // stores start on a 64-B sector of the ghosted row
    const int64_t waves_x = (x->nx + 55) / 56;
    const int64_t blocks_x = pc_form ? (x->nx - a.x_first + kPcValid - 1) / kPcValid : (waves_x + 3) / 4;
    {   // run length: same model as the Y march, with this kernel's residency
        const double slots = pc_form ? (double)ctx->n_cu * 2 : (double)ctx->n_cu * ARMON_C_WAVES;
        int best = (int)(x->ny < 32 ? x->ny : 32);
        double best_cost = 1e300;
        for (int64_t nruns = 1; nruns <= x->ny; nruns++) {
            const int64_t seg = (x->ny + nruns - 1) / nruns;
            if (seg < 32) break;
            if ((x->ny + seg - 1) / seg != nruns) continue;
            const double rounds = (double)(blocks_x * nruns) / slots;
            if (rounds < 2.) continue;
            const double cost = 0.5 * (rounds + std::ceil(rounds)) * (double)(seg + 2 * lag);
            if (cost < best_cost) { best_cost = cost; best = (int)seg; }
        }
        a.seg = ctx->tune_y_seg > 0 ? ctx->tune_y_seg : best;
    }
    ARMON_REQUIRE(a.row_len * (int64_t)sizeof(real) * (a.seg + 2 * lag + 16) < (1ll << 32), "block too wide for 32-bit row offsets");
    const int64_t n_out = a.o_hi - a.o_lo;
    dim3 grid((unsigned)blocks_x, (unsigned)((n_out + a.seg - 1) / a.seg));
    const int64_t n_blocks = (int64_t)grid.x * grid.y;
    a.partials = nullptr;
    if (track) {
        int rc = ensure_partials(ctx, (size_t)(2 * n_blocks));
        if (rc != ARMON_OK) return rc;
        a.partials = reinterpret_cast<real*>(ctx->partials);
    }
    constexpr int S = ARMON_SCHEME_GAD, L = ARMON_LIMITER_MINMOD, P = ARMON_PROJECTION_EULER_2ND, E = ARMON_EOS_PERFECT_GAS;
    if (pc_form) {
        if (track)
            hipLaunchKernelGGL((k_cycle_pc<S, L, P, E, true>), grid, dim3(192), 0, ctx->stream, a, (real)y->dt, (real)y->dx);
        else
            hipLaunchKernelGGL((k_cycle_pc<S, L, P, E, false>), grid, dim3(192), 0, ctx->stream, a, (real)y->dt, (real)y->dx);
    } else if (track)
        hipLaunchKernelGGL((k_cycle_xy<S, L, P, E, false, true>), grid, dim3(256), 0, ctx->stream, a, (real)y->dt, (real)y->dx);
    else
        hipLaunchKernelGGL((k_cycle_xy<S, L, P, E, false, false>), grid, dim3(256), 0, ctx->stream, a, (real)y->dt, (real)y->dx);
    int rc = check_launch("cycle_xy");
    if (rc != ARMON_OK || !track) return rc;
    hipLaunchKernelGGL(k_fold_dt, dim3(1), dim3(256), 0, ctx->stream, a.partials, n_blocks, (real)y->cfl_dx, (real)y->cfl_dy, y->dt_cfl_out, y->dt_accumulate);
    return check_launch("fold_dt");
}
