/* native_graph.c — a whole run from plain C with NO host-read scalar after its first cycle: the reference's GlobalTimeStep
 * (ref src/solver_state.jl:30-166) lives in device memory (armon_dt_state), the sweeps read their step from it, the fold of
 * the last sweep's dt reduction runs update_dt! + next_cycle! + the time loop's exit test (auto_step), and one captured
 * cycle (X sweep, Y sweep) is replayed with a single call per cycle (armon_hip_graph_*).
 * The same run is first made with the host-driven loop of native_cycle.c; the two must end on the same bits.
 *
 *   gcc -O2 -I include examples/native_graph.c -o examples/native_graph -L armon.jl_amd -larmon_hip \
 *       '-Wl,-rpath,$ORIGIN/../armon.jl_amd' -lm && examples/native_graph 1024 200
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

typedef struct {
    armon_ctx* ctx;
    int64_t n, row;
    double dx;
    double *f[20], *dt_dev, *dt_host;
    armon_block_data bd;
    armon_range full, real;
} run_t;

/* X sweep from set 1 into set 2, Y sweep back into set 1 with the CFL step of the resulting state in *dt_dev.
 * st == NULL: `dt` is the time step; otherwise the step is read from *st on the device and `dt` is its factor. */
static int enqueue_cycle(run_t* r, double dt, armon_dt_state* st)
{
    double *rho = r->f[2], *u = r->f[3], *v = r->f[4], *E = r->f[5], *rho2 = r->f[16], *u2 = r->f[17], *v2 = r->f[18], *E2 = r->f[19];
    armon_sweep_desc d;
    memset(&d, 0, sizeof d);
    d.scheme = ARMON_SCHEME_GAD; d.limiter = ARMON_LIMITER_MINMOD; d.projection = ARMON_PROJECTION_EULER_2ND;
    d.eos = ARMON_EOS_PERFECT_GAS; d.nghost = 4; d.bc_low = d.bc_high = 1; d.nx = d.ny = r->n; d.dx = r->dx; d.gamma = 1.4;
    d.cfl_dx = d.cfl_dy = r->dx; d.dt = dt; d.dt_state = st;
    d.axis = ARMON_AXIS_X;                                   /* Sod: u mirrored on left/right (ref src/tests.jl:164-211) */
    d.u_factor_low = d.u_factor_high = -1.; d.v_factor_low = d.v_factor_high = 1.;
    d.rho_in = rho; d.u_in = u; d.v_in = v; d.E_in = E; d.rho_out = rho2; d.u_out = u2; d.v_out = v2; d.E_out = E2;
    CHECK(armon_hip_sweep(r->ctx, &d));
    d.axis = ARMON_AXIS_Y;
    d.u_factor_low = d.u_factor_high = 1.; d.v_factor_low = d.v_factor_high = 1.;
    d.rho_in = rho2; d.u_in = u2; d.v_in = v2; d.E_in = E2; d.rho_out = rho; d.u_out = u; d.v_out = v; d.E_out = E;
    d.dt_cfl_out = r->dt_dev;
    CHECK(armon_hip_sweep(r->ctx, &d));
    return 0;
}

/* the initial condition and the first time step (synchronous, like the reference's first cycle) */
static int start(run_t* r, double cfl, double* dt0, double cons[2])
{
    const int64_t gpos[2] = {0, 0}, gN[2] = {r->n, r->n};
    const double origin[2] = {0., 0.}, dX[2] = {r->dx, r->dx};
    double l;
    CHECK(armon_hip_init_test(r->ctx, r->full, ARMON_TEST_SOD, r->row, r->row, 4, gpos, gN, origin, dX, 0., &r->bd));
    CHECK(armon_hip_conservation_vars(r->ctx, r->real, r->dx * r->dx, r->f[2], r->f[5], cons));
    CHECK(armon_hip_perfect_gas_EOS(r->ctx, r->real, 1.4, r->f[2], r->f[5], r->f[3], r->f[4], r->f[6], r->f[7], r->f[8]));
    CHECK(armon_hip_dtCFL(r->ctx, r->real, r->dx, r->dx, r->f[3], r->f[4], r->f[7], &l));
    *dt0 = cfl * l;
    return 0;
}

int main(int argc, char** argv)
{
    run_t r;
    memset(&r, 0, sizeof r);
    r.n = argc > 1 ? atoll(argv[1]) : 1024;
    const int cycles = argc > 2 ? atoi(argv[2]) : 100;
    const int g = 4;
    const double cfl = 0.95;
    if (cycles < 2) { fprintf(stderr, "at least 2 cycles\n"); return 1; }
    r.row = r.n + 2 * g;
    r.dx = 1.0 / (double)r.n;
    const int64_t cells = r.row * r.row;
    CHECK(armon_hip_init(0, NULL, &r.ctx));
    for (int k = 0; k < 20; k++) CHECK(armon_hip_malloc(r.ctx, (size_t)cells * sizeof(double), (void**)&r.f[k]));
    memcpy(&r.bd, r.f, sizeof r.bd);
    CHECK(armon_hip_malloc(r.ctx, 2 * sizeof(double), (void**)&r.dt_dev));
    CHECK(armon_hip_malloc_host(r.ctx, 2 * sizeof(double), (void**)&r.dt_host));
    r.full = (armon_range){0, r.row, r.row, 0, r.row};
    r.real = (armon_range){(int64_t)g * r.row, r.row, r.n, g, r.n};

    /* ---- 1. the host-driven loop (native_cycle.c): dt read back a cycle late, update_dt! on the host ---- */
    double dt_host_final = 0., time_host = 0., el_host = 0., cons0[2], cons_host[2];
    /* (run twice, the first time untimed: the first few hundred launches of a process are slower — clocks, the runtime's
     * pools — and would be charged to whichever loop comes first) */
    for (int pass = 0; pass < 2; pass++) {
        double dt, next_dt;
        if (start(&r, cfl, &dt, cons0)) return 1;
        next_dt = dt;
        time_host = 0.;
        CHECK(armon_hip_sync(r.ctx));
        const double t0 = now();
        for (int c = 0; c < cycles; c++) {
            if (enqueue_cycle(&r, dt, NULL)) return 1;
            CHECK(armon_hip_memcpy_async(r.ctx, &r.dt_host[c & 1], r.dt_dev, sizeof(double), ARMON_MEMCPY_D2H));
            CHECK(armon_hip_event_record(r.ctx, c & 1));
            if (c > 0) {
                CHECK(armon_hip_event_sync(r.ctx, (c - 1) & 1));
                next_dt = fmin(cfl * r.dt_host[(c - 1) & 1], 1.05 * dt);
            }
            time_host += dt;
            dt = next_dt;
        }
        CHECK(armon_hip_sync(r.ctx));
        el_host = now() - t0;
        dt_host_final = dt;
    }
    CHECK(armon_hip_conservation_vars(r.ctx, r.real, r.dx * r.dx, r.f[2], r.f[5], cons_host));

    /* ---- 2. the same run with the time step on the device and the cycle replayed from a graph ---- */
    double dt0, cons1[2], cons_graph[2], l0;
    if (start(&r, cfl, &dt0, cons1)) return 1;
    if (enqueue_cycle(&r, dt0, NULL)) return 1;             /* cycle 0: host-driven (it also sizes the library's scratch) */
    CHECK(armon_hip_memcpy(r.ctx, &l0, r.dt_dev, sizeof(double), ARMON_MEMCPY_D2H));
    CHECK(armon_hip_sync(r.ctx));                            /* copies are enqueued on the context's stream */
    armon_dt_state st, *st_dev;
    memset(&st, 0, sizeof st);
    st.current_dt = dt0;                                     /* cycle 1 runs with cycle 0's step: the rule lags by one cycle */
    st.time = dt0;
    st.L_prev = l0;                                          /* CFL step of the state cycle 1 starts from */
    st.cycle = 1;
    st.auto_step = 1; st.cst_dt = 0; st.maxcycle = cycles; st.cfl = cfl; st.maxtime = 1e30; st.Dt = 0.;
    CHECK(armon_hip_malloc(r.ctx, sizeof st, (void**)&st_dev));
    CHECK(armon_hip_memcpy(r.ctx, st_dev, &st, sizeof st, ARMON_MEMCPY_H2D));
    CHECK(armon_hip_sync(r.ctx));
    armon_graph* graph = NULL;
    CHECK(armon_hip_graph_begin(r.ctx));
    const int rc_capture = enqueue_cycle(&r, 1.0, st_dev);   /* recorded, not run: the step is a factor of the state's */
    CHECK(armon_hip_graph_end(r.ctx, &graph));
    if (rc_capture) return 1;
    CHECK(armon_hip_sync(r.ctx));
    const double t0 = now();
    for (int c = 1; c < cycles + 3; c++)                     /* three replays past the end: no-ops once `done` is set */
        CHECK(armon_hip_graph_launch(r.ctx, graph));
    CHECK(armon_hip_sync(r.ctx));
    const double el_graph = now() - t0;
    CHECK(armon_hip_memcpy(r.ctx, &st, st_dev, sizeof st, ARMON_MEMCPY_D2H));
    CHECK(armon_hip_sync(r.ctx));
    CHECK(armon_hip_conservation_vars(r.ctx, r.real, r.dx * r.dx, r.f[2], r.f[5], cons_graph));

    const int same = st.cycle == cycles && st.done == 1 && !st.invalid && st.current_dt == dt_host_final && st.time == time_host &&
                     cons_graph[0] == cons_host[0] && cons_graph[1] == cons_host[1] && cons1[0] == cons0[0] && cons1[1] == cons0[1];
    printf("Sod %lldx%lld, %d cycles: host-driven %.3f ms/cycle, graph replay %.3f ms/cycle\n", (long long)r.n, (long long)r.n,
           cycles, 1e3 * el_host / cycles, 1e3 * el_graph / (cycles - 1));
    printf("host : time %.17g, next dt %.17g, mass %.17g, energy %.17g\n", time_host, dt_host_final, cons_host[0], cons_host[1]);
    printf("graph: time %.17g, next dt %.17g, mass %.17g, energy %.17g, cycle %lld, done %d\n", st.time, st.current_dt,
           cons_graph[0], cons_graph[1], (long long)st.cycle, st.done);
    printf("identical: %s\n", same ? "yes" : "NO");
    armon_hip_graph_destroy(graph);
    for (int k = 0; k < 20; k++) armon_hip_free(r.ctx, r.f[k]);
    armon_hip_free(r.ctx, r.dt_dev);
    armon_hip_free(r.ctx, st_dev);
    armon_hip_free_host(r.ctx, r.dt_host);
    armon_hip_destroy(r.ctx);
    return same ? 0 : 3;
}
