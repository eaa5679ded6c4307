/* native_cycle.c — the hot path driven from plain C through include/armon_hip.h only (no Python, no torch):
 * allocate the reference's 16 BlockData vectors + 4 ping-pong vectors, init_test (Sod), then N cycles of
 * X sweep + Y sweep with the fused dt/CFL reduction and the reference's dt rule (cfl factor, +5 % growth cap,
 * one-cycle lag: ref src/solver_state.jl:102-166). Prints Mcells/s per sweep and mass/energy before and after.
 *
 *   gcc -O2 -I include examples/native_cycle.c -o examples/native_cycle -L armon.jl_amd -larmon_hip \
 *       '-Wl,-rpath,$ORIGIN/../armon.jl_amd' -lm && examples/native_cycle 8192 50
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "armon_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != ARMON_OK) { \
    fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, armon_hip_last_error()); return 1; } } while (0)

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 4096;
    const int cycles = argc > 2 ? atoi(argv[2]) : 20;
    const int g = 4;
    const int64_t row = n + 2 * g, cells = row * row;
    const double dx = 1.0 / (double)n, cfl = 0.95;          /* Sod defaults, ref src/tests.jl:32-44 */

    armon_ctx* ctx = NULL;
    CHECK(armon_hip_init(0, NULL, &ctx));
    double* f[20];                                           /* 16 BlockData fields + 4 alternate state vectors */
    for (int k = 0; k < 20; k++) CHECK(armon_hip_malloc(ctx, (size_t)cells * sizeof(double), (void**)&f[k]));
    armon_block_data bd;
    memcpy(&bd, f, sizeof bd);                               /* x,y,rho,u,v,E,p,c,g,us,ps,work_1..4,mask in this order */
    double *rho = f[2], *u = f[3], *v = f[4], *E = f[5], *rho2 = f[16], *u2 = f[17], *v2 = f[18], *E2 = f[19];

    const armon_range full = {0, row, row, 0, row};          /* every cell, ghosts included */
    const armon_range real = {(int64_t)g * row, row, n, g, n};
    const int64_t gpos[2] = {0, 0}, gN[2] = {n, n};
    const double origin[2] = {0., 0.}, dX[2] = {dx, dx};
    CHECK(armon_hip_init_test(ctx, full, ARMON_TEST_SOD, row, row, g, gpos, gN, origin, dX, 0., &bd));
    double cons0[2], cons1[2], dt_cfl;
    CHECK(armon_hip_conservation_vars(ctx, real, dx * dx, rho, E, cons0));
    CHECK(armon_hip_perfect_gas_EOS(ctx, real, 1.4, rho, E, u, v, f[6], f[7], f[8]));
    CHECK(armon_hip_dtCFL(ctx, real, dx, dx, u, v, f[7], &dt_cfl));       /* first step: synchronous */
    double dt = cfl * dt_cfl, next_dt = dt;

    double* dt_dev;                                          /* device scalar written by the fused reduction */
    double* dt_host;                                         /* pinned landing zone, one slot per cycle parity */
    CHECK(armon_hip_malloc(ctx, 2 * sizeof(double), (void**)&dt_dev));
    CHECK(armon_hip_malloc_host(ctx, 2 * sizeof(double), (void**)&dt_host));

    armon_sweep_desc d;
    memset(&d, 0, sizeof d);
    d.scheme = ARMON_SCHEME_GAD; d.limiter = ARMON_LIMITER_MINMOD; d.projection = ARMON_PROJECTION_EULER_2ND;
    d.eos = ARMON_EOS_PERFECT_GAS; d.nghost = g; d.bc_low = d.bc_high = 1; d.nx = d.ny = n; d.dx = dx; d.gamma = 1.4;
    d.cfl_dx = d.cfl_dy = dx;

    /* Placement of the 8 streamed vectors: 8 spare vectors, 12 role assignments timed by the
     * library with these very descriptors; keep the 8 it picks, free the rest. */
    if ((size_t)cells * sizeof(double) >= ((size_t)256 << 20)) {
        void* pool[16] = {rho, u, v, E, rho2, u2, v2, E2};
        int picks[8];
        double times[12];
        for (int k = 8; k < 16; k++) CHECK(armon_hip_malloc(ctx, (size_t)cells * sizeof(double), &pool[k]));
        armon_sweep_desc tx = d, ty = d;
        tx.axis = ARMON_AXIS_X; ty.axis = ARMON_AXIS_Y; tx.dt = ty.dt = 1e-3 * dx;
        tx.u_factor_low = tx.u_factor_high = -1.; tx.v_factor_low = tx.v_factor_high = 1.;
        ty.u_factor_low = ty.u_factor_high = ty.v_factor_low = ty.v_factor_high = 1.;
        ty.dt_cfl_out = dt_dev;
        CHECK(armon_hip_tune_placement(ctx, &tx, &ty, pool, 16, (size_t)cells * sizeof(double), 12, picks, times));
        double* chosen[8];
        for (int k = 0; k < 8; k++) chosen[k] = (double*)pool[picks[k]];
        for (int k = 0; k < 16; k++) {
            int kept = 0;
            for (int j = 0; j < 8; j++) kept |= picks[j] == k;
            if (!kept) CHECK(armon_hip_free(ctx, pool[k]));
        }
        rho = chosen[0]; u = chosen[1]; v = chosen[2]; E = chosen[3];
        rho2 = chosen[4]; u2 = chosen[5]; v2 = chosen[6]; E2 = chosen[7];
        f[2] = rho; f[3] = u; f[4] = v; f[5] = E; f[16] = rho2; f[17] = u2; f[18] = v2; f[19] = E2;
        printf("placement: X+Y ms per try:");
        for (int k = 0; k < 12; k++) printf(" %.2f", times[k]);
        printf("\n");
    }

    CHECK(armon_hip_sync(ctx));
    const double t0 = now();
    for (int c = 0; c < cycles; c++) {
        /* X sweep: Sod is Dirichlet on left/right (u mirrored), FreeFlow on bottom/top (ref src/tests.jl:164-211) */
        d.axis = ARMON_AXIS_X; d.dt = dt;
        d.u_factor_low = d.u_factor_high = -1.; d.v_factor_low = d.v_factor_high = 1.;
        d.rho_in = rho; d.u_in = u; d.v_in = v; d.E_in = E; d.rho_out = rho2; d.u_out = u2; d.v_out = v2; d.E_out = E2;
        d.dt_cfl_out = NULL;
        CHECK(armon_hip_sweep(ctx, &d));
        /* Y sweep, back into the first set, with the CFL step of the resulting state */
        d.axis = ARMON_AXIS_Y;
        d.u_factor_low = d.u_factor_high = 1.; d.v_factor_low = d.v_factor_high = 1.;
        d.rho_in = rho2; d.u_in = u2; d.v_in = v2; d.E_in = E2; d.rho_out = rho; d.u_out = u; d.v_out = v; d.E_out = E;
        d.dt_cfl_out = dt_dev;
        CHECK(armon_hip_sweep(ctx, &d));
        /* post the read-back of that step; pick up the one posted a cycle ago (never drains the stream) */
        CHECK(armon_hip_memcpy_async(ctx, &dt_host[c & 1], dt_dev, sizeof(double), ARMON_MEMCPY_D2H));
        CHECK(armon_hip_event_record(ctx, c & 1));
        if (c > 0) {
            CHECK(armon_hip_event_sync(ctx, (c - 1) & 1));
            const double l = dt_host[(c - 1) & 1];
            if (!(l > 0.) || !isfinite(l)) { fprintf(stderr, "invalid time step at cycle %d\n", c); return 2; }
            next_dt = fmin(cfl * l, 1.05 * dt);              /* update_dt!: ref src/solver_state.jl:102-142 */
        }
        dt = next_dt;                                        /* next_cycle!: ref :145-166 */
    }
    CHECK(armon_hip_sync(ctx));
    const double el = now() - t0;
    CHECK(armon_hip_conservation_vars(ctx, real, dx * dx, rho, E, cons1));
    printf("Sod %lldx%lld, %d cycles: %.3f ms/cycle, %.1f Mcells/s per sweep; mass %.17g -> %.17g, energy %.17g -> %.17g\n",
           (long long)n, (long long)n, cycles, 1e3 * el / cycles, 2. * n * n * cycles / el / 1e6, cons0[0], cons1[0], cons0[1], cons1[1]);
    for (int k = 0; k < 20; k++) armon_hip_free(ctx, f[k]);
    armon_hip_free(ctx, dt_dev);
    armon_hip_free_host(ctx, dt_host);
    armon_hip_destroy(ctx);
    return 0;
}
