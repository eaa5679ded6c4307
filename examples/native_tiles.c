/* native_tiles.c — the tile-decomposed hot path driven from plain C through include/armon_hip.h only: a px x py grid
 * of tiles in ONE process (armon_hip_mgpu_init; every tile on device 0 unless device ids are given), per sweep
 *   armon_hip_halo_exchange_start -> interior of the fused sweep on the compute stream, while on the tile's transfer
 *   stream: armon_hip_halo_exchange_finish_edge -> boundary strips (armon_hip_mgpu_edge_ctx) -> armon_hip_mgpu_edge_join,
 * the dt/CFL minimum reduced over the tiles on the device (armon_hip_dt_allreduce) and read back one cycle late — or, with
 * a 5th argument "cycle", all of that per cycle by ONE call: armon_hip_mgpu_cycle — the
 * reference's cycle (ref src/solver.jl:288-320) with its MPI exchange (ref src/halo_exchange.jl:229-354) and
 * MPI_Iallreduce (ref src/solver_state.jl:89-111) replaced by the library's own entry points. No host synchronisation
 * inside a cycle. Prints the global mass and energy: they must equal examples/native_cycle's for the same grid and
 * number of cycles, bit for bit (tests/test_native_example.py).
 *
 *   gcc -O2 -I include examples/native_tiles.c -o examples/native_tiles -L armon.jl_amd -larmon_hip \
 *       '-Wl,-rpath,$ORIGIN/../armon.jl_amd' -lm && examples/native_tiles 2048 2 2 20
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "armon_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != ARMON_OK) { \
    fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, armon_hip_last_error()); return 1; } } while (0)

enum { G = 4, LAG = 4, MAX_TILES = 64 };

typedef struct {
    armon_ctx *ctx, *edge;           /* compute stream; transfer ("edge") stream */
    double* edge_dt;                 /* 2 device scalars for the strips' CFL steps */
    int64_t nx, ny, ox, oy;          /* real cells and 0-based global position of the tile */
    int nb[4];                       /* neighbour rank per ARMON_SIDE_*, -1 = physical boundary */
    double* f[20];                   /* 16 BlockData vectors + 4 ping-pong partners */
    double *s[4], *a[4];             /* current state (rho,u,v,E) and its partners */
    double* dt_dev;
} tile;

/* one fused sweep of a sub-range [lo, hi) of the sweep axis (hi == 0: the whole tile), launched on `ctx`; dt_out: where
 * the CFL step of the cells it produced goes (NULL: not wanted) */
static armon_sweep_desc sweep_desc(const tile* t, int axis, double dt, double dx, int64_t lo, int64_t hi, double* dt_out)
{
    armon_sweep_desc d;
    memset(&d, 0, sizeof d);
    d.axis = axis; d.scheme = ARMON_SCHEME_GAD; d.limiter = ARMON_LIMITER_MINMOD; d.projection = ARMON_PROJECTION_EULER_2ND;
    d.eos = ARMON_EOS_PERFECT_GAS; d.nghost = G; d.nx = t->nx; d.ny = t->ny; d.dt = dt; d.dx = dx; d.gamma = 1.4;
    const int s_lo = axis == ARMON_AXIS_X ? ARMON_SIDE_LEFT : ARMON_SIDE_BOTTOM;
    d.bc_low = t->nb[s_lo] < 0; d.bc_high = t->nb[s_lo + 1] < 0;
    /* Sod: Dirichlet left/right (u mirrored with a sign flip), FreeFlow bottom/top (ref src/tests.jl:164-211) */
    d.u_factor_low = d.u_factor_high = axis == ARMON_AXIS_X ? -1. : 1.;
    d.v_factor_low = d.v_factor_high = 1.;
    d.rho_in = t->s[0]; d.u_in = t->s[1]; d.v_in = t->s[2]; d.E_in = t->s[3];
    d.rho_out = t->a[0]; d.u_out = t->a[1]; d.v_out = t->a[2]; d.E_out = t->a[3];
    d.out_lo = lo; d.out_hi = hi;
    if (dt_out) { d.dt_cfl_out = dt_out; d.cfl_dx = d.cfl_dy = dx; }
    return d;
}

static int sweep(tile* t, armon_ctx* ctx, int axis, double dt, double dx, int64_t lo, int64_t hi, double* dt_out)
{
    const armon_sweep_desc d = sweep_desc(t, axis, dt, dx, lo, hi, dt_out);
    return armon_hip_sweep(ctx, &d);
}

int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1024;
    const int px = argc > 2 ? atoi(argv[2]) : 2, py = argc > 3 ? atoi(argv[3]) : 2;
    const int cycles = argc > 4 ? atoi(argv[4]) : 20;
    /* "cycle": every cycle by ONE call (armon_hip_mgpu_cycle: the sequence below enqueued by the library, one host thread
     * per tile, the next CFL step reduced and read back on the transfer streams) instead of call by call */
    const int one_call = argc > 5 && !strcmp(argv[5], "cycle");
    const int nt = px * py;
    if (nt < 1 || nt > MAX_TILES || n / px < 2 * LAG + 1 || n / py < 2 * LAG + 1) { fprintf(stderr, "bad tile grid\n"); return 1; }
    const double dx = 1.0 / (double)n, cfl = 0.95;

    armon_mgpu* group = NULL;
    CHECK(armon_hip_mgpu_init(px, py, NULL, &group));          /* NULL: every tile on device 0 */
    static tile T[MAX_TILES];
    armon_halo_desc halo[MAX_TILES];
    double* dt_ptrs[MAX_TILES];
    for (int r = 0; r < nt; r++) {
        tile* t = &T[r];
        int coords[2];
        t->ctx = armon_hip_mgpu_ctx(group, r);
        t->edge = armon_hip_mgpu_edge_ctx(group, r);
        t->edge_dt = (double*)armon_hip_mgpu_edge_dt(group, r);
        CHECK(armon_hip_mgpu_tile_info(group, r, NULL, coords, t->nb));
        /* partition of ref src/parameters.jl:673-697: N / P cells, the remainder on the last tile of the axis */
        t->nx = n / px + (coords[0] == px - 1 ? n % px : 0);
        t->ny = n / py + (coords[1] == py - 1 ? n % py : 0);
        t->ox = coords[0] * (n / px);
        t->oy = coords[1] * (n / py);
        const int64_t row = t->nx + 2 * G, col = t->ny + 2 * G;
        for (int k = 0; k < 20; k++) CHECK(armon_hip_malloc(t->ctx, (size_t)(row * col) * sizeof(double), (void**)&t->f[k]));
        CHECK(armon_hip_malloc(t->ctx, 2 * sizeof(double), (void**)&t->dt_dev));
        for (int k = 0; k < 4; k++) { t->s[k] = t->f[2 + k]; t->a[k] = t->f[16 + k]; }
        armon_block_data bd;
        memcpy(&bd, t->f, sizeof bd);
        const armon_range full = {0, row, col, 0, row};
        const int64_t gpos[2] = {t->ox, t->oy}, gN[2] = {n, n};
        const double origin[2] = {0., 0.}, dX[2] = {dx, dx};
        CHECK(armon_hip_init_test(t->ctx, full, ARMON_TEST_SOD, row, col, G, gpos, gN, origin, dX, 0., &bd));
        dt_ptrs[r] = t->dt_dev;
    }

    /* conservation sums and the first time step: per tile, folded on the host (once per run) */
    double mass0 = 0., energy0 = 0., dt_cfl = INFINITY;
    for (int r = 0; r < nt; r++) {
        tile* t = &T[r];
        const int64_t row = t->nx + 2 * G;
        const armon_range real = {(int64_t)G * row, row, t->ny, G, t->nx};
        double cons[2], l;
        CHECK(armon_hip_conservation_vars(t->ctx, real, dx * dx, t->s[0], t->s[3], cons));
        CHECK(armon_hip_perfect_gas_EOS(t->ctx, real, 1.4, t->s[0], t->s[3], t->s[1], t->s[2], t->f[6], t->f[7], t->f[8]));
        CHECK(armon_hip_dtCFL(t->ctx, real, dx, dx, t->s[1], t->s[2], t->f[7], &l));
        mass0 += cons[0]; energy0 += cons[1];
        dt_cfl = fmin(dt_cfl, l);
    }
    double dt = cfl * dt_cfl, next_dt = dt;
    double* dt_host;
    CHECK(armon_hip_malloc_host(T[0].ctx, 2 * sizeof(double), (void**)&dt_host));

    for (int c = 0; c < cycles && one_call; c++) {
        armon_cycle_plan plan;
        static armon_tile_cycle tc[MAX_TILES];
        memset(&plan, 0, sizeof plan);
        plan.n_sweeps = 2; plan.emit_dt = 1; plan.overlap = 1;
        plan.axis[0] = ARMON_AXIS_X; plan.axis[1] = ARMON_AXIS_Y; plan.dt[0] = plan.dt[1] = dt;
        plan.next_axis = c + 1 < cycles ? ARMON_AXIS_X : -1;    /* the next cycle's first exchange is posted ahead */
        plan.event_slot = -1;
        plan.dt_host = &dt_host[c & 1]; plan.dt_event_slot = c & 1;             /* an event of tile 0's EDGE context */
        for (int r = 0; r < nt; r++) {                          /* full sweeps from the set that holds the state now */
            tc[r].x = sweep_desc(&T[r], ARMON_AXIS_X, 0., dx, 0, 0, T[r].dt_dev);
            tc[r].y = sweep_desc(&T[r], ARMON_AXIS_Y, 0., dx, 0, 0, T[r].dt_dev);
        }
        CHECK(armon_hip_mgpu_cycle(group, &plan, tc));          /* two sweeps: the state is back in the same set */
        if (c > 0) {
            CHECK(armon_hip_event_sync(T[0].edge, (c - 1) & 1));
            const double l = dt_host[(c - 1) & 1];
            if (!(l > 0.) || !isfinite(l)) { fprintf(stderr, "invalid time step at cycle %d\n", c); return 2; }
            next_dt = fmin(cfl * l, 1.05 * dt);
        }
        dt = next_dt;
    }
    if (one_call) CHECK(armon_hip_mgpu_sync(group));            /* compute AND transfer streams */
    for (int c = 0; c < cycles && !one_call; c++) {
        for (int axis = ARMON_AXIS_X; axis <= ARMON_AXIS_Y; axis++) {
            const int last = axis == ARMON_AXIS_Y;
            for (int r = 0; r < nt; r++) {                     /* what every tile exchanges: its current rho,u,v,E */
                halo[r].nx = T[r].nx; halo[r].ny = T[r].ny; halo[r].nghost = G; halo[r].nvars = 4;
                for (int k = 0; k < 4; k++) halo[r].vars[k] = T[r].s[k];
            }
            CHECK(armon_hip_halo_exchange_start(group, axis, halo));
            for (int r = 0; r < nt; r++) {                     /* interiors: read no ghost cell, overlap the transfers */
                tile* t = &T[r];
                const int s_lo = axis == ARMON_AXIS_X ? ARMON_SIDE_LEFT : ARMON_SIDE_BOTTOM;
                const int64_t na = axis == ARMON_AXIS_X ? t->nx : t->ny;
                const int64_t lo = t->nb[s_lo] >= 0 ? LAG : 0, hi = t->nb[s_lo + 1] >= 0 ? na - LAG : na;
                CHECK(sweep(t, t->ctx, axis, dt, dx, lo, (lo == 0 && hi == na) ? 0 : hi, last ? t->dt_dev : NULL));
            }
            /* on the transfer streams, concurrent with the interiors: unpack, then the LAG-wide strips next to the remote
             * sides (they read the same input state and write cells the interior does not) */
            CHECK(armon_hip_halo_exchange_finish_edge(group, axis, halo));
            for (int r = 0; r < nt; r++) {
                tile* t = &T[r];
                const int s_lo = axis == ARMON_AXIS_X ? ARMON_SIDE_LEFT : ARMON_SIDE_BOTTOM;
                const int64_t na = axis == ARMON_AXIS_X ? t->nx : t->ny;
                if (t->nb[s_lo] >= 0) CHECK(sweep(t, t->edge, axis, dt, dx, 0, LAG, last ? t->edge_dt : NULL));
                if (t->nb[s_lo + 1] >= 0) CHECK(sweep(t, t->edge, axis, dt, dx, na - LAG, na, last ? t->edge_dt + 1 : NULL));
                for (int k = 0; k < 4; k++) { double* tmp = t->s[k]; t->s[k] = t->a[k]; t->a[k] = tmp; }
            }
            /* compute streams wait for their tile's edge work; after the last sweep the strips' CFL steps are folded in */
            CHECK(armon_hip_mgpu_edge_join(group, last ? dt_ptrs : NULL));
        }
        /* global minimum on the device, read back one cycle late from tile 0 (ref src/solver_state.jl:145-166) */
        CHECK(armon_hip_dt_allreduce(group, dt_ptrs));
        CHECK(armon_hip_memcpy_async(T[0].ctx, &dt_host[c & 1], T[0].dt_dev, sizeof(double), ARMON_MEMCPY_D2H));
        CHECK(armon_hip_event_record(T[0].ctx, c & 1));
        if (c > 0) {
            CHECK(armon_hip_event_sync(T[0].ctx, (c - 1) & 1));
            const double l = dt_host[(c - 1) & 1];
            if (!(l > 0.) || !isfinite(l)) { fprintf(stderr, "invalid time step at cycle %d\n", c); return 2; }
            next_dt = fmin(cfl * l, 1.05 * dt);
        }
        dt = next_dt;
    }

    double mass1 = 0., energy1 = 0.;
    for (int r = 0; r < nt; r++) {
        tile* t = &T[r];
        const int64_t row = t->nx + 2 * G;
        const armon_range real = {(int64_t)G * row, row, t->ny, G, t->nx};
        double cons[2];
        CHECK(armon_hip_conservation_vars(t->ctx, real, dx * dx, t->s[0], t->s[3], cons));
        mass1 += cons[0]; energy1 += cons[1];
    }
    printf("Sod %lldx%lld on %dx%d tiles, %d cycles: mass %.17g -> %.17g, energy %.17g -> %.17g\n",
           (long long)n, (long long)n, px, py, cycles, mass0, mass1, energy0, energy1);
    for (int r = 0; r < nt; r++) {
        for (int k = 0; k < 20; k++) armon_hip_free(T[r].ctx, T[r].f[k]);
        armon_hip_free(T[r].ctx, T[r].dt_dev);
    }
    armon_hip_free_host(T[0].ctx, dt_host);
    armon_hip_mgpu_destroy(group);
    return 0;
}
