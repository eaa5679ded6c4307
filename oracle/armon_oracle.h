/*
 * armon_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C fp64 restatement of Armon.jl's direction-split hot path, used only as the checker by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg. The product path
 * (armon.jl_amd/, libarmon_hip.so) never includes, links or calls anything in this directory.
 *
 * Parity status: PINNED — the composition of these functions reproduces the reference's own golden
 * files test/reference_data/ref_{Sod,Sod_y,Sod_circ,Bizarrium,Sedov}_64bits.csv (re-encoded under
 * tests/golden/ by tests/golden/make_golden.py); see tests/test_oracle_golden.py.
 *
 * Kernel signatures mirror include/armon_hip.h one-to-one (host pointers instead of device pointers)
 * so a parity test feeds the same inputs to both.
 */
#ifndef ARMON_ORACLE_H
#define ARMON_ORACLE_H

#include <stdint.h>
#include "../include/armon_hip.h"   /* armon_range, tags */

/* Precision of this build of the oracle: fp64 by default, fp32 with -DARMON_ORACLE_F32 (the reference
 * supports data_type=Float32, ref src/parameters.jl:185; golden files ref_*_32bits.csv). Both builds export
 * the same symbol names; they live in two libraries (libarmon_oracle.so / libarmon_oracle_f32.so). */
#ifdef ARMON_ORACLE_F32
typedef float real;
#else
typedef double real;
#endif

typedef struct {
    real *x, *y, *rho, *u, *v, *E, *p, *c, *g, *us, *ps, *work_1, *work_2, *work_3, *work_4, *mask;
} armon_oracle_block_data;   /* ref src/blocking/blocks.jl:18-35 (BlockData) */

#ifdef __cplusplus
extern "C" {
#endif

void armon_oracle_set_threads(int n);   /* OpenMP threads for the row loops (1 = serial) */
int  armon_oracle_get_threads(void);

void armon_oracle_perfect_gas_EOS(armon_range, real gamma,
        const real* rho, const real* E, const real* u, const real* v,
        real* p, real* c, real* g);
void armon_oracle_bizarrium_EOS(armon_range,
        const real* rho, const real* u, const real* v, const real* E,
        real* p, real* c, real* g);
void armon_oracle_acoustic(armon_range, int64_t s, real* us, real* ps,
        const real* rho, const real* ua, const real* p, const real* c);
void armon_oracle_acoustic_GAD(armon_range, int64_t s, real dt, real dx,
        real* us, real* ps,
        const real* rho, const real* ua, const real* p, const real* c, int limiter);
void armon_oracle_cell_update(armon_range, int64_t s, real dx, real dt,
        const real* us, const real* ps, real* rho, real* ua, real* E);
void armon_oracle_advection_first_order(armon_range, int64_t s, real dt,
        const real* us, const real* rho, const real* u, const real* v, const real* E,
        real* adv_rho, real* adv_urho, real* adv_vrho, real* adv_Erho);
void armon_oracle_advection_second_order(armon_range, int64_t s, real dx, real dt,
        const real* us, const real* rho, const real* u, const real* v, const real* E,
        real* adv_rho, real* adv_urho, real* adv_vrho, real* adv_Erho);
void armon_oracle_euler_projection(armon_range, int64_t s, real dx, real dt,
        const real* us, real* rho, real* u, real* v, real* E,
        const real* adv_rho, const real* adv_urho, const real* adv_vrho, const real* adv_Erho);
void armon_oracle_boundary_conditions(armon_range, int64_t incr, int nghost,
        real u_factor, real v_factor,
        real* rho, real* u, real* v, real* p, real* c, real* g, real* E);
void armon_oracle_pack_to_array(armon_range, int nghost, int64_t face,
        real* array, int nvars, const real* const* vars);
void armon_oracle_unpack_from_array(armon_range, int nghost, int64_t face,
        const real* array, int nvars, real* const* vars);
real armon_oracle_dtCFL(armon_range, real dx, real dy,
        const real* u, const real* v, const real* c);
void armon_oracle_conservation_vars(armon_range, real ds,
        const real* rho, const real* E, real out[2]);
void armon_oracle_init_test(armon_range, int test, int64_t row_length, int64_t col_length,
        int nghost, const int64_t global_pos[2], const int64_t global_N[2],
        const real origin[2], const real dX[2], real sedov_r, const armon_oracle_block_data* data);

/* ---- whole solver on one ghosted block (ref src/solver.jl:288-403) --------------------------- */
enum { ARMON_SPLIT_SEQUENTIAL = 0, ARMON_SPLIT_GODUNOV = 1, ARMON_SPLIT_STRANG = 2,
       ARMON_SPLIT_X_ONLY = 3, ARMON_SPLIT_Y_ONLY = 4 };

typedef struct {
    int32_t test, scheme, limiter, projection, splitting, nghost;
    int64_t nx, ny;
    real  domain_size[2], origin[2];
    real  cfl, maxtime;
    int64_t maxcycle;
    int32_t cst_dt;  real Dt;
    /* outputs */
    real  final_time, last_dt;
    int64_t cycles;
    double solve_seconds;
    real  initial_mass, initial_energy, final_mass, final_energy;
    int32_t status;   /* 0 ok, ARMON_ERR_INVALID_DT */
    /* TEST AID, not in the reference (its domain is never periodic): the ghosts of that axis are filled from the OPPOSITE
     * border instead of the mirror — the independent check of the library's periodic transport runs (the test aid
     * armon_hip_mgpu_set_periodic, tests/test_gpu_transport.py). 0, 0 = the reference's boundary conditions. */
    int32_t periodic[2];
} armon_oracle_run;

/* Allocates nothing: `data` holds 16 caller-provided arrays of (nx+2g)(ny+2g) doubles.
 * Runs init_test then time_loop; when `skip_init` != 0 the arrays are used as they are. */
int armon_oracle_solve(armon_oracle_run* run, const armon_oracle_block_data* data, int skip_init);

#ifdef __cplusplus
}
#endif
#endif
