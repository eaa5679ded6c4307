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
#include "../include/armon_hip.h"   /* armon_range, armon_block_data, tags */

#ifdef __cplusplus
extern "C" {
#endif

void armon_oracle_set_threads(int n);   /* OpenMP threads for the row loops (1 = serial) */
int  armon_oracle_get_threads(void);

void armon_oracle_perfect_gas_EOS(armon_range, double gamma,
        const double* rho, const double* E, const double* u, const double* v,
        double* p, double* c, double* g);
void armon_oracle_bizarrium_EOS(armon_range,
        const double* rho, const double* u, const double* v, const double* E,
        double* p, double* c, double* g);
void armon_oracle_acoustic(armon_range, int64_t s, double* us, double* ps,
        const double* rho, const double* ua, const double* p, const double* c);
void armon_oracle_acoustic_GAD(armon_range, int64_t s, double dt, double dx,
        double* us, double* ps,
        const double* rho, const double* ua, const double* p, const double* c, int limiter);
void armon_oracle_cell_update(armon_range, int64_t s, double dx, double dt,
        const double* us, const double* ps, double* rho, double* ua, double* E);
void armon_oracle_advection_first_order(armon_range, int64_t s, double dt,
        const double* us, const double* rho, const double* u, const double* v, const double* E,
        double* adv_rho, double* adv_urho, double* adv_vrho, double* adv_Erho);
void armon_oracle_advection_second_order(armon_range, int64_t s, double dx, double dt,
        const double* us, const double* rho, const double* u, const double* v, const double* E,
        double* adv_rho, double* adv_urho, double* adv_vrho, double* adv_Erho);
void armon_oracle_euler_projection(armon_range, int64_t s, double dx, double dt,
        const double* us, double* rho, double* u, double* v, double* E,
        const double* adv_rho, const double* adv_urho, const double* adv_vrho, const double* adv_Erho);
void armon_oracle_boundary_conditions(armon_range, int64_t incr, int nghost,
        double u_factor, double v_factor,
        double* rho, double* u, double* v, double* p, double* c, double* g, double* E);
void armon_oracle_pack_to_array(armon_range, int nghost, int64_t face,
        double* array, int nvars, const double* const* vars);
void armon_oracle_unpack_from_array(armon_range, int nghost, int64_t face,
        const double* array, int nvars, double* const* vars);
double armon_oracle_dtCFL(armon_range, double dx, double dy,
        const double* u, const double* v, const double* c);
void armon_oracle_conservation_vars(armon_range, double ds,
        const double* rho, const double* E, double out[2]);
void armon_oracle_init_test(armon_range, int test, int64_t row_length, int64_t col_length,
        int nghost, const int64_t global_pos[2], const int64_t global_N[2],
        const double origin[2], const double dX[2], double sedov_r, const armon_block_data* data);

/* ---- whole solver on one ghosted block (ref src/solver.jl:288-403) --------------------------- */
enum { ARMON_SPLIT_SEQUENTIAL = 0, ARMON_SPLIT_GODUNOV = 1, ARMON_SPLIT_STRANG = 2,
       ARMON_SPLIT_X_ONLY = 3, ARMON_SPLIT_Y_ONLY = 4 };

typedef struct {
    int32_t test, scheme, limiter, projection, splitting, nghost;
    int64_t nx, ny;
    double  domain_size[2], origin[2];
    double  cfl, maxtime;
    int64_t maxcycle;
    int32_t cst_dt;  double Dt;
    /* outputs */
    double  final_time, last_dt;
    int64_t cycles;
    double  solve_seconds;
    double  initial_mass, initial_energy, final_mass, final_energy;
    int32_t status;   /* 0 ok, ARMON_ERR_INVALID_DT */
} armon_oracle_run;

/* Allocates nothing: `data` holds 16 caller-provided arrays of (nx+2g)(ny+2g) doubles.
 * Runs init_test then time_loop; when `skip_init` != 0 the arrays are used as they are. */
int armon_oracle_solve(armon_oracle_run* run, const armon_block_data* data, int skip_init);

#ifdef __cplusplus
}
#endif
#endif
