/*
 * armon_hip.h — C ABI of libarmon_hip.so, the MI355X (gfx950) device backend for the
 * direction-split hot path of Armon.jl.
 *
 * Every entry point below is what a Julia `ccall` (or any FFI) binds; no C++/torch types cross the
 * boundary. The design follows the reference's own precedent for a C-ABI kernel library, the Kokkos
 * extension: symbol = kernel name without `!` (ref src/generic_kernel.jl:616-620), iteration ranges
 * passed first as plain structs (ref src/generic_kernel.jl:576-614, ext/ArmonKokkos.jl:10-30),
 * limiter / test case as C int tags (ref ext/ArmonKokkos.jl:50-69), `flt_size`/`idx_size` sanity
 * exports (ref ext/ArmonKokkos.jl:122-139), errors reported through a return code + message
 * (ref ext/ArmonKokkos.jl:72-76 → `solver_error(:cpp, msg)`, ref src/utils.jl:108).
 *
 * Conventions
 *  - All floating point data is fp64 (`double`) unless the symbol ends in `_f32`.
 *  - All indices/strides are int64_t and 0-BASED on this side of the boundary (Julia subtracts 1).
 *  - Arrays are flat device pointers of `(Nx+2g)*(Ny+2g)` elements, x-contiguous rows
 *    (ref src/blocking/blocks.jl:18-50, src/blocking/blocking.jl:129-131). They are owned by the
 *    caller; the library never frees or retains them beyond the stream-ordered call.
 *  - Every kernel call is ASYNCHRONOUS on the context's stream and returns 0 on success or a
 *    non-zero `armon_status`; `armon_hip_last_error()` gives the message. `armon_hip_sync`
 *    implements `Base.wait(params)` (ref src/parameters.jl:1031-1038).
 *  - One host thread per context.
 */
#ifndef ARMON_HIP_H
#define ARMON_HIP_H

#include <stddef.h>
#include <stdint.h>

#if defined(ARMON_BUILDING_LIB)
#define ARMON_API __attribute__((visibility("default")))
#else
#define ARMON_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------- */
typedef enum {
    ARMON_OK = 0,
    ARMON_ERR_INVALID_ARG = 1,   /* bad range / null pointer / unsupported option          */
    ARMON_ERR_HIP = 2,           /* a HIP runtime call failed (message has hipGetErrorString) */
    ARMON_ERR_NO_DEVICE = 3,     /* no usable gfx950 device                                   */
    ARMON_ERR_INVALID_DT = 4     /* non finite or <= 0 time step (ref src/solver_state.jl:123) */
} armon_status;

/* ---- tags ------------------------------------------------------------------------------------ */
/* ref ext/ArmonKokkos.jl:50-57 (limiter tags), src/limiters.jl:6-8 */
enum { ARMON_LIMITER_NONE = 0, ARMON_LIMITER_MINMOD = 1, ARMON_LIMITER_SUPERBEE = 2 };
/* ref ext/ArmonKokkos.jl:60-69 (test tags), src/tests.jl:5-11; DebugIndexes ref src/tests.jl:195 */
enum { ARMON_TEST_SOD = 0, ARMON_TEST_SOD_Y = 1, ARMON_TEST_SOD_CIRC = 2, ARMON_TEST_BIZARRIUM = 3,
       ARMON_TEST_SEDOV = 4, ARMON_TEST_DEBUG_INDEXES = 5 };
/* ref src/riemann_schemes.jl:5-6 */
enum { ARMON_SCHEME_GODUNOV = 0, ARMON_SCHEME_GAD = 1 };
/* ref src/projection_schemes.jl:4-5 */
enum { ARMON_PROJECTION_EULER = 0, ARMON_PROJECTION_EULER_2ND = 1 };
/* EOS selected by the test case: ref src/kernels.jl:151-161 */
enum { ARMON_EOS_PERFECT_GAS = 0, ARMON_EOS_BIZARRIUM = 1 };
/* ref src/blocking/blocking.jl Axis / Side enums (Axis.X=1,Y=2; Side.Left=1,Right,Bottom,Top) */
enum { ARMON_AXIS_X = 0, ARMON_AXIS_Y = 1 };
enum { ARMON_SIDE_LEFT = 0, ARMON_SIDE_RIGHT = 1, ARMON_SIDE_BOTTOM = 2, ARMON_SIDE_TOP = 3 };
enum { ARMON_MEMCPY_H2D = 1, ARMON_MEMCPY_D2H = 2, ARMON_MEMCPY_D2D = 3 };

/*
 * Iteration range = the reference's DomainRange (ref src/domain_ranges.jl:39-61), 0-based:
 * cell index = col_start + j*col_step + row_start + i,  j in [0,col_len), i in [0,row_len).
 * From Julia: col_start = first(range.col)-1, col_step = step(range.col), col_len = length(range.col),
 *             row_start = first(range.row)-1, row_len = length(range.row).
 */
typedef struct {
    int64_t col_start, col_step, col_len;
    int64_t row_start, row_len;
} armon_range;

typedef struct armon_ctx armon_ctx;

/* ---- sanity / context (ref ext/ArmonKokkos.jl:122-139, src/parameters.jl:751-802,921-926) ---- */
ARMON_API int         armon_hip_flt_size(void);              /* sizeof(double) = 8                         */
ARMON_API int         armon_hip_idx_size(void);              /* sizeof(int64_t) = 8                        */
ARMON_API const char* armon_hip_version(void);
ARMON_API const char* armon_hip_last_error(void);            /* thread-local message of the last failure   */
ARMON_API int         armon_hip_device_count(int* count);

/* `stream` = an existing hipStream_t to enqueue on (e.g. the caller's), or NULL to create one. */
ARMON_API int armon_hip_init(int device_id, void* stream, armon_ctx** ctx);   /* create_device/init_backend */
ARMON_API int armon_hip_destroy(armon_ctx* ctx);
ARMON_API int armon_hip_sync(armon_ctx* ctx);                                  /* Base.wait(params)          */
ARMON_API int armon_hip_device_memory_info(armon_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
ARMON_API int armon_hip_device_name(armon_ctx* ctx, char* buf, size_t buf_len);
ARMON_API void* armon_hip_stream(armon_ctx* ctx);                              /* the hipStream_t in use     */
/* Tuning knobs of the fused sweeps (measurement tools and tests; no reference counterpart). A context reads the
 * environment variables of the same names ONCE, when it is created; this call changes them afterwards.
 * "ARMON_XS_NITER" strips per wave of the X sweep (values > 1 select the multi-strip form with its prefetch buffer, which only
 * the A/B build libarmon_hip_alt.so carries; the product library runs one strip per wave), "ARMON_Y_SEG" rows per run of the Y march (0 = automatic),
 * "ARMON_SWEEP_ALIGN" 0 = unaligned block/strip origins, "ARMON_Y_COLS1" 1 = fp32 Y march with one column per lane,
 * "ARMON_X_XCD" XCD-aware workgroup order of the X sweep: 1 = on, 0 = off, < 0 = automatic (on for fp64, off for fp32, whose
 * workgroups share their lines themselves), "ARMON_X_ROWS" workgroup shape of the X sweep: 1 = one strip of
 * 4 rows, 2 = 4 consecutive strips of one row, 0 = automatic (the former for fp64, the latter for fp32),
 * "ARMON_Y_SX" store exchange of the Y march (rows stored in sector-aligned windows handed over through LDS): 1 = always,
 * 2 = never, 0 = automatic (when the row pitch is not a multiple of a 64-B sector), "ARMON_COPY_NT" the measurement aid
 * armon_hip_stream_copy4 with non-temporal loads (bit 0) / stores (bit 1).
 * None of them changes a result bit. */
ARMON_API int armon_hip_set_tuning(armon_ctx* ctx, const char* knob, int value);
/* the current value of a knob; and, read-only, "Y_RUN_ROWS": rows per run the last automatic choice of the Y march took (0
 * before the first Y sweep of the context) — tests use it to check that a shape really has several runs per column */
ARMON_API int armon_hip_get_tuning(armon_ctx* ctx, const char* knob, int* value);

/* device array type support: V{T,1}(undef, n) / copyto! (ref src/blocking/blocks.jl:36-44,121-143) */
ARMON_API int armon_hip_malloc(armon_ctx* ctx, size_t bytes, void** ptr);
ARMON_API int armon_hip_free(armon_ctx* ctx, void* ptr);
ARMON_API int armon_hip_memcpy(armon_ctx* ctx, void* dst, const void* src, size_t bytes, int kind); /* async */
ARMON_API int armon_hip_memset(armon_ctx* ctx, void* dst, int byte_value, size_t bytes);             /* async */
/* Pinned host memory + truly asynchronous copies: how the host reads the fused dt/CFL scalar one cycle late
 * (the reference consumes a cycle's reduction in the NEXT cycle, src/solver_state.jl:89-99,145-166) without ever
 * draining the stream: memcpy_async into a pinned slot, event_record, and event_sync a cycle later. The host
 * buffer of memcpy_async must come from malloc_host and stay untouched until the copy's event has completed. */
ARMON_API int armon_hip_malloc_host(armon_ctx* ctx, size_t bytes, void** ptr);
ARMON_API int armon_hip_free_host(armon_ctx* ctx, void* ptr);
ARMON_API int armon_hip_memcpy_async(armon_ctx* ctx, void* dst, const void* src, size_t bytes, int kind);

/* Measurement aid (bench.py): plain streaming copy of four arrays into four others in ONE launch —
 * the traffic shape of a fused sweep (4 read + 4 written, 16 B per lane) with no arithmetic. Its rate on
 * the device at hand is the practical ceiling the sweep kernels are compared with, next to the 8 TB/s
 * spec. `bytes` per array, multiple of 16; async on the context's stream. No reference counterpart. */
ARMON_API int armon_hip_stream_copy4(armon_ctx* ctx, const void* const in[4], void* const out[4], size_t bytes);

/* stream timers for benchmarks (hipEvents on the context's stream).
 * event_record/event_elapsed: a pool of ARMON_HIP_MAX_EVENTS reusable events, recorded without any
 * host synchronisation; event_elapsed_ms synchronises on event `b` and returns t(b) - t(a). */
#define ARMON_HIP_MAX_EVENTS 1024
ARMON_API int armon_hip_event_record(armon_ctx* ctx, int slot);
ARMON_API int armon_hip_event_elapsed_ms(armon_ctx* ctx, int a, int b, double* elapsed_ms);
ARMON_API int armon_hip_event_sync(armon_ctx* ctx, int slot);                   /* host waits for that event only */
ARMON_API int armon_hip_timer_start(armon_ctx* ctx);
ARMON_API int armon_hip_timer_stop(armon_ctx* ctx, double* elapsed_ms);        /* synchronises on the stop event */

/* ---- staged kernels: one per reference @generic_kernel ---------------------------------------- */

/* perfect_gas_EOS!(params, data, range, γ)           ref src/kernels.jl:4-13, call :154 */
ARMON_API int armon_hip_perfect_gas_EOS(armon_ctx*, armon_range, double gamma,
        const double* rho, const double* E, const double* u, const double* v,
        double* p, double* c, double* g);

/* bizarrium_EOS!(params, data, range)                ref src/kernels.jl:16-55, call :160 */
ARMON_API int armon_hip_bizarrium_EOS(armon_ctx*, armon_range,
        const double* rho, const double* u, const double* v, const double* E,
        double* p, double* c, double* g);

/* acoustic!(params, data, range, s, uˢ_, pˢ_, uₐ)    ref src/riemann_schemes.jl:33-52 */
ARMON_API int armon_hip_acoustic(armon_ctx*, armon_range, int64_t s, double* us, double* ps,
        const double* rho, const double* ua, const double* p, const double* c);

/* acoustic_GAD!(params, data, range, s, dt, dx, uₐ, lim)   ref src/riemann_schemes.jl:55-113 */
ARMON_API int armon_hip_acoustic_GAD(armon_ctx*, armon_range, int64_t s, double dt, double dx,
        double* us, double* ps,
        const double* rho, const double* ua, const double* p, const double* c, int limiter);

/* cell_update!(params, data, range, s, dx, dt, uₐ)   ref src/kernels.jl:58-68,217-223 */
ARMON_API int armon_hip_cell_update(armon_ctx*, armon_range, int64_t s, double dx, double dt,
        const double* us, const double* ps, double* rho, double* ua, double* E);

/* advection_first_order!(params, data, range, s, dt, a_ρ, a_uρ, a_vρ, a_Eρ)
 *                                                    ref src/projection_schemes.jl:62-89 */
ARMON_API int armon_hip_advection_first_order(armon_ctx*, armon_range, int64_t s, double dt,
        const double* us, const double* rho, const double* u, const double* v, const double* E,
        double* adv_rho, double* adv_urho, double* adv_vrho, double* adv_Erho);

/* advection_second_order!(params, data, range, s, dx, dt, a×4)
 *                                                    ref src/projection_schemes.jl:15-20,92-135 */
ARMON_API int armon_hip_advection_second_order(armon_ctx*, armon_range, int64_t s, double dx, double dt,
        const double* us, const double* rho, const double* u, const double* v, const double* E,
        double* adv_rho, double* adv_urho, double* adv_vrho, double* adv_Erho);

/* euler_projection!(params, data, range, s, dx, dt, a×4)   ref src/projection_schemes.jl:23-52 */
ARMON_API int armon_hip_euler_projection(armon_ctx*, armon_range, int64_t s, double dx, double dt,
        const double* us, double* rho, double* u, double* v, double* E,
        const double* adv_rho, const double* adv_urho, const double* adv_vrho, const double* adv_Erho);

/* boundary_conditions!(params, data, range, bsize, axis, side, u_factor, v_factor)
 * ref src/halo_exchange.jl:2-36. `range` = border_domain(bsize, side) (one strip of real cells);
 * `incr` = ±stride_along(bsize, axis), signed towards the edge (ref :8-10); `nghost` = ghosts(bsize). */
ARMON_API int armon_hip_boundary_conditions(armon_ctx*, armon_range, int64_t incr, int nghost,
        double u_factor, double v_factor,
        double* rho, double* u, double* v, double* p, double* c, double* g, double* E);

/* pack_to_array!/unpack_from_array!(params, range, bsize, side, array, vars)
 * ref src/halo_exchange.jl:187-216. `range` = border_domain / ghost_domain (single_strip=false);
 * `face` = real_face_size(bsize, side); buffer index (i_g*face + i)*nvars + v with
 * (i, i_g) = divrem(iter, nghost), iter = 0-based position in the range (row-major).
 * `vars` = HOST array of `nvars` device pointers (comm_vars: ρ,u,v,E,p,c,g  ref src/blocking/blocks.jl:50). */
ARMON_API int armon_hip_pack_to_array(armon_ctx*, armon_range, int nghost, int64_t face,
        double* array, int nvars, const double* const* vars);
ARMON_API int armon_hip_unpack_from_array(armon_ctx*, armon_range, int nghost, int64_t face,
        const double* array, int nvars, double* const* vars);

/* dtCFL_kernel(params, state, blk, Δx)::T            ref src/reductions.jl:2-110
 * Min over `range` (real cells) of min(dx/max|u±c|, dy/max|v±c|). Two forms:
 *  _async: result left in device memory `*result_dev` (one double), no host sync;
 *  plain : synchronises and returns the value in `*result_host`. */
ARMON_API int armon_hip_dtCFL_async(armon_ctx*, armon_range, double dx, double dy,
        const double* u, const double* v, const double* c, double* result_dev);
ARMON_API int armon_hip_dtCFL(armon_ctx*, armon_range, double dx, double dy,
        const double* u, const double* v, const double* c, double* result_host);

/* conservation_vars(params, blk)::(T,T)              ref src/reductions.jl:202-323
 * out[0] = ds*Σρ, out[1] = ds*ΣρE over `range`; synchronises. */
ARMON_API int armon_hip_conservation_vars(armon_ctx*, armon_range, double ds,
        const double* rho, const double* E, double out_host[2]);

/* init_test(params, data, range, global_pos, bsize, ΔX, vars_to_zero, test_case)
 * ref src/kernels.jl:71-145,176-207, src/tests.jl:59-121.
 * `range` = full domain (real+ghost); row_length = Nx+2g; global_pos = 0-based global index of the
 * block's first real cell (ref src/kernels.jl:181-182); global_N = global grid (for DebugIndexes);
 * sedov_r (ref src/tests.jl:15-19) is only read for ARMON_TEST_SEDOV. All 16 arrays are written; x, y, rho, u, v, E are
 * required, any of the other ten may be NULL and is then skipped — the reference's `vars_to_zero` argument (ref
 * src/kernels.jl:142-144,188-191): a host that runs the fused sweeps only never reads us, ps, work_1..4 and mask. */
typedef struct {
    double *x, *y, *rho, *u, *v, *E, *p, *c, *g, *us, *ps, *work_1, *work_2, *work_3, *work_4, *mask;
} armon_block_data;   /* ref src/blocking/blocks.jl:18-35 (BlockData) */

ARMON_API int armon_hip_init_test(armon_ctx*, armon_range, int test, int64_t row_length, int64_t col_length,
        int nghost, const int64_t global_pos[2], const int64_t global_N[2],
        const double origin[2], const double dX[2], double sedov_r, const armon_block_data* data);

/* ---- fp32 variants (ref data_type=Float32, src/parameters.jl:185): same kernels, float arrays and scalars ---- */
typedef struct {
    float *x, *y, *rho, *u, *v, *E, *p, *c, *g, *us, *ps, *work_1, *work_2, *work_3, *work_4, *mask;
} armon_block_data_f32;

ARMON_API int armon_hip_perfect_gas_EOS_f32(armon_ctx*, armon_range, float gamma,
        const float* rho, const float* E, const float* u, const float* v, float* p, float* c, float* g);
ARMON_API int armon_hip_bizarrium_EOS_f32(armon_ctx*, armon_range,
        const float* rho, const float* u, const float* v, const float* E, float* p, float* c, float* g);
ARMON_API int armon_hip_acoustic_f32(armon_ctx*, armon_range, int64_t s, float* us, float* ps,
        const float* rho, const float* ua, const float* p, const float* c);
ARMON_API int armon_hip_acoustic_GAD_f32(armon_ctx*, armon_range, int64_t s, float dt, float dx, float* us, float* ps,
        const float* rho, const float* ua, const float* p, const float* c, int limiter);
ARMON_API int armon_hip_cell_update_f32(armon_ctx*, armon_range, int64_t s, float dx, float dt,
        const float* us, const float* ps, float* rho, float* ua, float* E);
ARMON_API int armon_hip_advection_first_order_f32(armon_ctx*, armon_range, int64_t s, float dt,
        const float* us, const float* rho, const float* u, const float* v, const float* E,
        float* adv_rho, float* adv_urho, float* adv_vrho, float* adv_Erho);
ARMON_API int armon_hip_advection_second_order_f32(armon_ctx*, armon_range, int64_t s, float dx, float dt,
        const float* us, const float* rho, const float* u, const float* v, const float* E,
        float* adv_rho, float* adv_urho, float* adv_vrho, float* adv_Erho);
ARMON_API int armon_hip_euler_projection_f32(armon_ctx*, armon_range, int64_t s, float dx, float dt,
        const float* us, float* rho, float* u, float* v, float* E,
        const float* adv_rho, const float* adv_urho, const float* adv_vrho, const float* adv_Erho);
ARMON_API int armon_hip_boundary_conditions_f32(armon_ctx*, armon_range, int64_t incr, int nghost,
        float u_factor, float v_factor, float* rho, float* u, float* v, float* p, float* c, float* g, float* E);
ARMON_API int armon_hip_pack_to_array_f32(armon_ctx*, armon_range, int nghost, int64_t face,
        float* array, int nvars, const float* const* vars);
ARMON_API int armon_hip_unpack_from_array_f32(armon_ctx*, armon_range, int nghost, int64_t face,
        const float* array, int nvars, float* const* vars);
ARMON_API int armon_hip_dtCFL_async_f32(armon_ctx*, armon_range, float dx, float dy,
        const float* u, const float* v, const float* c, float* result_dev);
ARMON_API int armon_hip_dtCFL_f32(armon_ctx*, armon_range, float dx, float dy,
        const float* u, const float* v, const float* c, float* result_host);
ARMON_API int armon_hip_conservation_vars_f32(armon_ctx*, armon_range, float ds,
        const float* rho, const float* E, float out_host[2]);
ARMON_API int armon_hip_init_test_f32(armon_ctx*, armon_range, int test, int64_t row_length, int64_t col_length,
        int nghost, const int64_t global_pos[2], const int64_t global_N[2],
        const float origin[2], const float dX[2], float sedov_r, const armon_block_data_f32* data);

/* ---- fused sweep: EOS → BC (in-tile mirror) → fluxes → cell update → advection → projection ---- */
/* One call = one directional sweep of solver_cycle (ref src/solver.jl:300-316) over one block, reading
 * (ρ,u,v,E) from `in` and writing them to `out` (ping-pong; `in` and `out` must not alias).
 * Algorithmic traffic 64 B per real cell. Optionally materialises p (saved_vars, ref
 * src/blocking/blocks.jl:49) and the per-cell CFL minimum for the next cycle. */
struct armon_dt_state;       /* the time-step state on the device, defined below */
typedef struct {
    int32_t axis;            /* ARMON_AXIS_X / _Y                                               */
    int32_t scheme;          /* ARMON_SCHEME_*                                                  */
    int32_t limiter;         /* ARMON_LIMITER_*  (GAD only)                                     */
    int32_t projection;      /* ARMON_PROJECTION_*                                              */
    int32_t eos;             /* ARMON_EOS_*                                                     */
    int32_t nghost;          /* ghost layers of the arrays (>= scheme·projection stencil)       */
    int32_t bc_low, bc_high; /* 1: physical boundary on the low/high side of `axis` → mirror BC
                                applied in-tile; 0: ghosts already hold neighbour data (halo)   */
    int32_t exact;           /* 1: IEEE division/sqrt, no contraction: bit-identical to the staged kernels and to the
                                CPU restatement of the reference — every bit, subnormal results included
                                (-DARMON_LOOSE_SUBNORMAL at build time trades that last case for 5 %);
                                0: tuned (shared 1-ulp reciprocals + FMA, within the reference's golden tolerance) */
    int32_t x_kernel;        /* X-sweep kernel form: 0 = the one the library runs (lanes along x, DPP
                                shifts, 2 cells per lane). 3 (1 cell per lane) and 2 (LDS-transposed
                                march) are measured alternatives that exist only in the A/B build
                                libarmon_hip_alt.so (-DARMON_ALT_KERNELS); the product library refuses them */
    int64_t nx, ny;          /* real cells of the block                                         */
    double  dt, dx;          /* sweep time step (current_dt·factor) and cell size along axis    */
    double  gamma;           /* perfect gas only                                                */
    double  u_factor_low, v_factor_low, u_factor_high, v_factor_high;   /* BC factors per side  */
    const double *rho_in, *u_in, *v_in, *E_in;
    double *rho_out, *u_out, *v_out, *E_out;
    double *p_out;           /* nullable: EOS pressure of the PRE-sweep state (real cells)      */
    double *c_out;           /* nullable: EOS sound speed of the PRE-sweep state (real cells)   */
    /* Fused dt/CFL reduction of the NEXT cycle (ref src/reductions.jl:2-53, src/solver.jl:298): when
     * dt_cfl_out is non-NULL the sweep also reduces min(cfl_dx/max|u±c|, cfl_dy/max|v±c|) over the real
     * cells, with the post-sweep u, v and the pre-sweep c — exactly what dtCFL_kernel reads at the start
     * of the next cycle when this is the last sweep of a cycle — into *dt_cfl_out (one
     * device double, written by a follow-up fold kernel on the same stream). */
    double *dt_cfl_out;
    double  cfl_dx, cfl_dy;  /* GLOBAL cell sizes along x and y (ref src/reductions.jl:92)        */
    /* Partial sweeps, for overlapping the halo exchange with compute: produce only the cells
     * out_lo <= i < out_hi along the sweep axis (0-based real coordinates; out_hi == 0 means the whole
     * block). The interior [LAG, n-LAG) needs no ghost cell and can run while the halos travel; the two
     * LAG-wide boundary strips follow once the ghosts are in. dt_accumulate != 0: *dt_cfl_out =
     * min(*dt_cfl_out, this launch's value) instead of overwriting it. */
    int64_t out_lo, out_hi;
    int32_t dt_accumulate;
    int32_t reserved;
    /* Device-resident time step (armon_dt_state below), or NULL. When set, the sweep reads the cycle's time step from
     * device memory — its step is `dt` (then a FACTOR: 1, or 0.5 for Strang's half sweeps) x dt_state->current_dt —, does
     * nothing at all once dt_state->done is set, and writes p_out only when dt_state->emit_p is set: what lets a whole
     * cycle be captured once in a hipGraph and replayed without the host reading a scalar (armon_hip_graph_*). */
    struct armon_dt_state* dt_state;
} armon_sweep_desc;

ARMON_API int armon_hip_sweep(armon_ctx*, const armon_sweep_desc*);

/* ---- the dt state machine on the device + graph replay of a cycle -------------------------------------------------------- */
/* GlobalTimeStep (ref src/solver_state.jl:30-166) in device memory: update_dt! (:102-142: validity check, cfl scaling, the
 * 1.05 cap) and next_cycle! (:145-166) run as a one-thread kernel after the last sweep of a cycle, so that a cycle —
 * sweeps, dt reduction, state update, the time_loop's exit test (ref src/solver.jl:350) — has no host-read scalar and can
 * be replayed from a hipGraph. Values are stored as doubles; an fp32 run computes them in float (the _f32 step) exactly as
 * the reference's GlobalTimeStep{Float32} does. */
typedef struct armon_dt_state {
    double  current_dt;      /* time step of the cycle about to run / running                                        */
    double  time;            /* physical time at the start of that cycle                                            */
    double  L_prev;          /* CFL step of the state that cycle starts from (reduced by the previous cycle's last sweep) */
    int64_t cycle;
    int32_t done;            /* the time loop is over (time >= maxtime, cycle >= maxcycle, or an invalid step): sweeps and steps are no-ops */
    int32_t invalid;         /* a non-finite or non-positive step was seen (ref :123-124), at cycle `invalid_cycle`  */
    int32_t emit_p;          /* the cycle about to run is the last one: its last sweep materialises p               */
    int32_t pad;
    int64_t invalid_cycle;
    double  invalid_value;
    /* auto_step != 0: the fold of a sweep's fused dt reduction (dt_cfl_out with dt_state set, dt_accumulate == 0) performs
     * the state-machine step itself, with the constants below — one kernel less per replayed cycle than a separate
     * armon_hip_dt_state_step (a small grid's cycle time is its number of dependent launches). */
    int32_t auto_step, cst_dt;
    int64_t maxcycle;
    double  cfl, maxtime, Dt;
} armon_dt_state;

/* One state-machine step on the context's stream, after the last sweep of a cycle whose dt_cfl_out was `L_new_dev`:
 * next_cycle_dt = min(cfl x L_prev, 1.05 x current_dt) (or Dt with cst_dt), cycle += 1, time += current_dt,
 * current_dt = next_cycle_dt, L_prev = *L_new_dev, then done / emit_p for the next cycle from maxtime and maxcycle. */
ARMON_API int armon_hip_dt_state_step(armon_ctx*, armon_dt_state* state_dev, const double* L_new_dev, double cfl,
                                      double maxtime, int64_t maxcycle, int cst_dt, double Dt);
ARMON_API int armon_hip_dt_state_step_f32(armon_ctx*, armon_dt_state* state_dev, const float* L_new_dev, double cfl,
                                          double maxtime, int64_t maxcycle, int cst_dt, double Dt);

/* Stream capture of whatever the caller enqueues on the context's stream between _begin and _end (sweeps with dt_state,
 * the state step) into an executable graph; _launch replays it on that stream. Everything captured must already have run
 * once outside a capture (the library sizes its scratch on first use and cannot allocate while capturing).
 * The context must own its stream (armon_hip_init with stream = NULL): _begin refuses a caller-supplied or null stream.
 * Invalidation rule: the captured kernels hold the address of the context's reduction scratch, so while a graph of a
 * context is alive that scratch cannot grow — a launch on the same context that needs more of it (a larger block, a
 * first dtCFL) FAILS with ARMON_ERR_INVALID_ARG instead of moving it; run such launches before capturing, or destroy the
 * graph first. A graph should be destroyed before its context; the other order is safe (finalizers of a garbage-collected
 * host): armon_hip_destroy disowns the context's live graphs — their handles can then only be passed to
 * armon_hip_graph_destroy, armon_hip_graph_launch refuses them. */
typedef struct armon_graph armon_graph;
ARMON_API int armon_hip_graph_begin(armon_ctx*);
ARMON_API int armon_hip_graph_end(armon_ctx*, armon_graph** graph);
ARMON_API int armon_hip_graph_launch(armon_ctx*, armon_graph* graph);
ARMON_API int armon_hip_graph_destroy(armon_graph* graph);

/* Whole cycle in one launch (Sequential splitting: X sweep, then Y sweep — ref src/solver.jl:300-316 executed twice)
 * over one block: reads (rho,u,v,E) of x_desc->*_in ONCE and writes y_desc->*_out ONCE; the state between the two
 * sweeps stays in registers (32 B read + 32 B written per cell per CYCLE). x_desc gives the X sweep's dt, dx and
 * boundary; y_desc the Y sweep's, plus p_out (EOS pressure of the state before the Y sweep), dt_cfl_out and an
 * optional row range out_lo/out_hi. Results are the bits of armon_hip_sweep(x_desc) followed by armon_hip_sweep(y_desc).
 * fp64, GAD + minmod + euler_2nd, perfect gas, tuned arithmetic only; a block whose Y sides are remote cannot use it
 * (the rows received from a neighbour would have to be X-swept first). x_desc->x_kernel selects the form: 0 = one wave
 * runs both stages, 4 = producer / consumer waves through LDS. Both are measured alternatives (register-bound, slower
 * than the two launches today: DESIGN.md section 4.2), not what the solver runs: the kernels are compiled only into the
 * A/B build libarmon_hip_alt.so (-DARMON_ALT_KERNELS); in the product library this entry point returns
 * ARMON_ERR_INVALID_ARG. No reference counterpart. */
ARMON_API int armon_hip_cycle_xy(armon_ctx*, const armon_sweep_desc* x_desc, const armon_sweep_desc* y_desc);

/* Placement of the 8 vectors a fused sweep streams (4 read + 4 written). On MI355X the same sweeps run 10-20 %
 * apart depending on where these vectors sit in HBM relative to each other (back-to-back allocations end up at a
 * regular physical spacing that makes the 8 streams collide on channels/banks), so a host
 * should allocate a few spare vectors once and let this call choose: `pool` holds n_pool >= 8 device vectors of
 * `bytes` bytes, pool[0..3] = the current (rho,u,v,E) WITH the state, pool[4..7] = their ping-pong partners, the
 * rest spares. `tries` role assignments (the first = pool[0..7] as given, the others random) are timed with the
 * caller's own descriptors as in a cycle — X reads roles 0..3 and writes 4..7, Y reads 4..7 and writes 0..3; all
 * other fields of x_desc / y_desc are used as they are (dt_cfl_out included), their 8 state pointers are ignored.
 * On return picks[role] = index in `pool` of the vector to use for that role, the state lives in pool[picks[0..3]],
 * every other vector of the pool is free for the caller to release; times_ms (nullable, [tries]) = the best
 * X+Y time of each assignment. Synchronous; needs 4 more vectors of `bytes` transiently. No reference counterpart. */
ARMON_API int armon_hip_tune_placement(armon_ctx*, const armon_sweep_desc* x_desc, const armon_sweep_desc* y_desc,
        void* const* pool, int n_pool, size_t bytes, int tries, int picks[8], double* times_ms);

/* The same choice for a pool that holds NO state yet — call it BEFORE init_test: nothing has to be parked or restored,
 * so the only transient memory is the caller's own spare vectors (n_pool - 8 of them). Candidates are timed on a uniform
 * state the call writes itself (every vector of the pool is overwritten); after 12 draws the search stops early once two
 * of them lie within `tolerance` (e.g. 0.01; 0 = never) of the best seen and a draw >= 7 % slower has been seen too,
 * after at most `tries`. picks[role] as above (roles 0..3:
 * where init_test should put rho,u,v,E; 4..7: their ping-pong partners); *tries_done (nullable) = draws timed. */
ARMON_API int armon_hip_choose_placement(armon_ctx*, const armon_sweep_desc* x_desc, const armon_sweep_desc* y_desc,
        void* const* pool, int n_pool, size_t bytes, int tries, double tolerance, int picks[8], double* times_ms,
        int* tries_done);

/* fp32 descriptor: field for field the fp64 one (see armon_sweep_desc for the meaning of each), with float arrays;
 * the scalars stay double and are rounded to float inside. */
typedef struct {
    int32_t axis, scheme, limiter, projection, eos, nghost;
    int32_t bc_low, bc_high;
    int32_t exact, x_kernel;
    int64_t nx, ny;
    double  dt, dx, gamma;
    double  u_factor_low, v_factor_low, u_factor_high, v_factor_high;
    const float *rho_in, *u_in, *v_in, *E_in;
    float *rho_out, *u_out, *v_out, *E_out;
    float *p_out, *c_out;
    float *dt_cfl_out;
    double  cfl_dx, cfl_dy;
    int64_t out_lo, out_hi;
    int32_t dt_accumulate, reserved;
    struct armon_dt_state* dt_state;
} armon_sweep_desc_f32;

ARMON_API int armon_hip_sweep_f32(armon_ctx*, const armon_sweep_desc_f32*);
ARMON_API int armon_hip_tune_placement_f32(armon_ctx*, const armon_sweep_desc_f32* x_desc, const armon_sweep_desc_f32* y_desc,
        void* const* pool, int n_pool, size_t bytes, int tries, int picks[8], double* times_ms);
ARMON_API int armon_hip_choose_placement_f32(armon_ctx*, const armon_sweep_desc_f32* x_desc, const armon_sweep_desc_f32* y_desc,
        void* const* pool, int n_pool, size_t bytes, int tries, double tolerance, int picks[8], double* times_ms,
        int* tries_done);


/* ---- multi-GPU: tile-decomposed grid, halo exchange and dt reduction on the device ------------------------------ */
/* Replaces the reference's MPI path: start_exchange / finish_exchange (ref src/halo_exchange.jl:229-283), the side
 * selection of block_ghost_exchange (ref :323-354: the two sides ALONG the sweep axis, all ghost layers, no corners),
 * the MPI_Iallreduce(MIN) of the time step (ref src/solver_state.jl:89-111, src/utils.jl:126-143) and the cartesian
 * topology of init_MPI (ref src/parameters.jl:441-447: rank r at coords (r / py, r % py), non-periodic).
 *
 * A group is a px x py grid of tiles. Each local tile owns an armon_ctx (its compute stream: enqueue that tile's
 * kernels there) and a transfer stream. An exchange is pack -> (event) -> transfer -> (event) -> unpack, ordered on
 * the device only: NO host synchronisation between pack and unpack, so whatever the caller enqueues on a tile's
 * compute stream between _start and _finish (the interior of a fused sweep) overlaps with the transfers.
 *  - armon_hip_mgpu_init      : every tile lives in this process, tile r on device device_ids[r] (devices may repeat;
 *                               NULL = all on device 0). Faces travel as peer-to-peer copies (xGMI between GPUs).
 *  - armon_hip_mgpu_init_rank : one process per GPU; this process owns tile `rank` only. Faces travel as RCCL
 *                               send/recv pairs, the dt minimum as an RCCL all-reduce. `id` = ARMON_MGPU_ID_BYTES bytes
 *                               from armon_hip_mgpu_unique_id on rank 0, broadcast by the host's own launcher
 *                               (MPI_Bcast in Armon.jl, the torch.distributed store in bench.py). Collective.
 *                               `stream`: compute stream to adopt, or NULL. */
typedef struct armon_mgpu armon_mgpu;
#define ARMON_MGPU_ID_BYTES 256

ARMON_API int armon_hip_mgpu_init(int px, int py, const int* device_ids, armon_mgpu** group);
ARMON_API int armon_hip_mgpu_unique_id(void* id);
ARMON_API int armon_hip_mgpu_init_rank(int px, int py, int rank, int device_id, void* stream, const void* id,
                                       armon_mgpu** group);
/* armon_hip_mgpu_init_rank in two steps, for hosts that want to agree on readiness in between (bench.py does, over its
 * launcher's group): _prepare_rank is everything LOCAL to the process (RCCL symbols, context, streams, scratch) and may
 * fail on one rank alone; _connect is the collective ncclCommInitRank of the two communicators — every rank enters it, or
 * none (a rank that skipped it after a local failure would leave the others blocked in it). */
ARMON_API int armon_hip_mgpu_prepare_rank(int px, int py, int rank, int device_id, void* stream, armon_mgpu** group);
ARMON_API int armon_hip_mgpu_connect(armon_mgpu* group, const void* id);
ARMON_API int armon_hip_mgpu_destroy(armon_mgpu* group);
/* Test aids for the TRANSPORT (no reference counterpart: the reference's process grid is never periodic, ref
 * src/parameters.jl:432). _set_periodic: out-of-grid neighbours wrap around along that axis, so a 1 x 1 rank group is its
 * own neighbour on every side and ONE GPU executes the real ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd of
 * armon_hip_halo_exchange_start (call it before the first exchange; the caller's sweeps must then treat those sides as
 * remote: bc_low = bc_high = 0). _force_peer_copy: an in-process group moves its faces (and the dt scalars) with
 * hipMemcpyPeerAsync even when both tiles sit on the same device. */
ARMON_API int armon_hip_mgpu_set_periodic(armon_mgpu* group, int periodic_x, int periodic_y);
ARMON_API int armon_hip_mgpu_force_peer_copy(armon_mgpu* group, int on);
/* Test aid: from now on, with probability 1/2 each, the group's stream operations (packs, transfers, unpacks, the edge
 * join, the dt reduction) are preceded by a busy-wait kernel of up to max_delay_us on their stream — timings a single
 * GPU never produces by itself; results must not change. 0 switches it off. seed 0 = default sequence. */
ARMON_API int armon_hip_mgpu_set_chaos(armon_mgpu* group, unsigned max_delay_us, uint64_t seed);
ARMON_API int armon_hip_mgpu_n_local(armon_mgpu* group);                       /* tiles owned by this process */
ARMON_API armon_ctx* armon_hip_mgpu_ctx(armon_mgpu* group, int local_tile);    /* that tile's context (owned by the group) */
/* rank, cartesian coords and the neighbour rank per ARMON_SIDE_* (-1 = physical boundary, MPI.PROC_NULL) */
ARMON_API int armon_hip_mgpu_tile_info(armon_mgpu* group, int local_tile, int* rank, int coords[2], int neighbours[4]);

/* What one local tile exchanges: fused path (rho,u,v,E); staged path comm_vars (rho,u,v,E,p,c,g), ref
 * src/blocking/blocks.jl:50. The wire format is pack_to_array!'s (ref src/halo_exchange.jl:187-216). */
typedef struct {
    int64_t nx, ny;          /* real cells of the tile                              */
    int32_t nghost, nvars;   /* ghost layers; number of vectors (1..8)              */
    void*   vars[8];         /* device vectors of (nx+2g)*(ny+2g) elements          */
} armon_halo_desc;

/* `tiles` = one descriptor per LOCAL tile, in local tile order. _start packs and posts the transfers of the two
 * sides along `axis` of every local tile that has a neighbour there; _finish makes each compute stream wait for its
 * receives and unpacks them into the ghost cells; armon_hip_halo_exchange = _start + _finish. Physical sides are
 * left alone (mirror boundary condition: armon_hip_boundary_conditions, or bc_low/bc_high of the fused sweep). */
ARMON_API int armon_hip_halo_exchange_start(armon_mgpu*, int axis, const armon_halo_desc* tiles);
ARMON_API int armon_hip_halo_exchange_finish(armon_mgpu*, int axis, const armon_halo_desc* tiles);
ARMON_API int armon_hip_halo_exchange(armon_mgpu*, int axis, const armon_halo_desc* tiles);
ARMON_API int armon_hip_halo_exchange_start_f32(armon_mgpu*, int axis, const armon_halo_desc* tiles);
ARMON_API int armon_hip_halo_exchange_finish_f32(armon_mgpu*, int axis, const armon_halo_desc* tiles);
ARMON_API int armon_hip_halo_exchange_f32(armon_mgpu*, int axis, const armon_halo_desc* tiles);

/* Edge stream. What follows the receive of a sweep's faces — the unpack and the LAG-wide strips next to the remote
 * sides — reads the same input state as the interior sweep and writes cells the interior does not, so it can run on the
 * tile's TRANSFER stream while the interior runs on the compute stream, instead of in a chain of small launches after
 * it (the interior reads no ghost cell and never writes the strips' cells):
 *     armon_hip_halo_exchange_start -> armon_hip_sweep(ctx, interior) -> armon_hip_halo_exchange_finish_edge
 *     -> armon_hip_sweep(edge_ctx, strip) per remote side, dt_cfl_out = edge_dt + side index, dt_accumulate = 0
 *     -> armon_hip_mgpu_edge_join(group, dt_dev or NULL)
 * _edge_ctx: the tile's context on its transfer stream (owned by the group). _edge_dt: 2 device scalars of the run's
 * precision for the strips' CFL steps. _edge_join: every compute stream waits for its tile's edge work; with dt_dev,
 * dt_dev[k] = min(dt_dev[k], the tile's two edge scalars) on the compute stream, and the edge scalars are reset.
 * Replaces the interconnect-free part of ref src/halo_exchange.jl:256-283 (finish_exchange + the boundary blocks). */
ARMON_API armon_ctx* armon_hip_mgpu_edge_ctx(armon_mgpu* group, int local_tile);
ARMON_API void* armon_hip_mgpu_edge_dt(armon_mgpu* group, int local_tile);
ARMON_API int armon_hip_halo_exchange_finish_edge(armon_mgpu*, int axis, const armon_halo_desc* tiles);
ARMON_API int armon_hip_halo_exchange_finish_edge_f32(armon_mgpu*, int axis, const armon_halo_desc* tiles);
ARMON_API int armon_hip_mgpu_edge_join(armon_mgpu*, double* const* dt_dev);
ARMON_API int armon_hip_mgpu_edge_join_f32(armon_mgpu*, float* const* dt_dev);

/* Global minimum of one device scalar per local tile (dt_cfl_out of the fused sweep / armon_hip_dtCFL_async), in
 * place, ordered on the compute streams: afterwards every dt_dev[k] holds the minimum over ALL tiles of the group.
 * The reference consumes it one cycle later (ref src/solver_state.jl:145-166): nothing is waited for on the host. */
ARMON_API int armon_hip_dt_allreduce(armon_mgpu*, double* const* dt_dev);
ARMON_API int armon_hip_dt_allreduce_f32(armon_mgpu*, float* const* dt_dev);

/* A whole solver cycle of every LOCAL tile in one call — solver_cycle (ref src/solver.jl:288-320) on the fused path:
 * for each sweep of the cycle (ref split_axes, src/axis_splitting.jl:24-46) the halo exchange along its axis (ref
 * block_ghost_exchange, src/halo_exchange.jl:323-354), the interior of the sweep while the faces travel, the unpack and the
 * LAG-wide strips on the transfer stream, the join; after the last sweep the global minimum of the next CFL step (ref
 * src/solver_state.jl:89-111) — the calls above in the order the edge-stream protocol prescribes, enqueued natively. A
 * group of several local tiles is driven by ONE HOST THREAD PER TILE (created by the first cycle; ARMON_MGPU_THREADS=0 or
 * armon_hip_mgpu_set_threads(group, 0) keeps everything in the calling thread): eight devices fed from one thread cost
 * eight times the enqueue time of one, more than the 0.75 ms of GPU work of the tile of a 16384² grid on 8 GPUs.
 * Nothing is waited for on the host; the call returns when everything is enqueued.
 *  plan  : n_sweeps (1..3), axis[s] and dt[s] of each sweep (dt = current_dt x the splitting's factor, rounded as the run's
 *          precision rounds it), emit_p (last cycle of a run: the last sweep materialises p, ref saved_vars), emit_dt (the
 *          last sweep reduces the next CFL step and the group takes its minimum — folded, reduced and copied to dt_host on
 *          the TRANSFER streams: the compute streams go from this cycle's last sweep to the next cycle's first without
 *          waiting for it, as the reference's MPI_Iallreduce is waited for one cycle later; afterwards every tile's
 *          dt_cfl_out holds the minimum, ordered on that tile's transfer stream; 0 with cst_dt), overlap (0 = unpack and
 *          sweep in order on the compute stream), next_axis = axis of the NEXT cycle's first sweep, whose exchange is
 *          posted at the end of this call and consumed by the next one (-1 = none; a posted exchange nobody consumes is
 *          completed by armon_hip_mgpu_drain), event_slot >= 0: events event_slot + 2s / + 2s + 1 of event_ctx's pool
 *          are recorded on local tile 0's compute stream around sweep s (armon_hip_event_elapsed_ms reads them), -1 = none.
 *  tiles : per local tile the descriptors of a FULL X and a FULL Y sweep of that tile as armon_hip_sweep takes them, with
 *          *_in = the vectors that hold the state when the call is made and *_out = their partners (the same two sets in x
 *          and y); p_out / dt_cfl_out name where emit_p / emit_dt put their results; dt, bc_low / bc_high, out_lo / out_hi,
 *          dt_accumulate are set by the library (the sides from the group's topology). After the call the state is in the
 *          *_out set when n_sweeps is odd, in the *_in set when it is even.
 * Environment knobs of a group, read when it is created (measurement tools; none changes a result bit): ARMON_MGPU_THREADS=0
 * (no host threads), ARMON_MGPU_XFER_PRIORITY=normal|lowest (priority of the transfer streams; default lowest when a tile has
 * its device to itself), ARMON_MGPU_PACK=compute (halo packs in front of the interior on the compute stream, as until round
 * 4), ARMON_MGPU_TIMING=1 (host time of tile 0's steps inside the call, printed when the group is destroyed). */
typedef struct {
    int32_t n_sweeps, emit_p, emit_dt, overlap;
    int32_t axis[4];             /* [0 .. n_sweeps) used */
    double  dt[4];
    int32_t next_axis, event_slot;
    armon_ctx* event_ctx;        /* whose event pool event_slot names: a context on local tile 0's compute stream; NULL = that tile's own */
    void*   dt_host;             /* emit_dt: pinned host slot (one element of the run's precision, armon_hip_malloc_host) the   */
    int32_t dt_event_slot;       /* global minimum is copied to, and the event of local tile 0's EDGE context (armon_hip_mgpu_edge_ctx)
                                    recorded behind that copy (-1 = none): armon_hip_event_sync(edge_ctx, slot), then read *dt_host */
    int32_t reserved;
} armon_cycle_plan;
typedef struct { armon_sweep_desc x, y; } armon_tile_cycle;
typedef struct { armon_sweep_desc_f32 x, y; } armon_tile_cycle_f32;
ARMON_API int armon_hip_mgpu_cycle(armon_mgpu*, const armon_cycle_plan* plan, const armon_tile_cycle* tiles);
ARMON_API int armon_hip_mgpu_cycle_f32(armon_mgpu*, const armon_cycle_plan* plan, const armon_tile_cycle_f32* tiles);
ARMON_API int armon_hip_mgpu_drain(armon_mgpu*, const armon_tile_cycle* tiles);      /* tiles: descriptors whose *_in hold the state */
ARMON_API int armon_hip_mgpu_drain_f32(armon_mgpu*, const armon_tile_cycle_f32* tiles);
ARMON_API int armon_hip_mgpu_sync(armon_mgpu*);       /* host waits for the compute AND transfer streams of every local tile */
ARMON_API int armon_hip_mgpu_set_threads(armon_mgpu*, int on);                        /* 1 / 0; < 0 = by the environment */

/* Host-value all-reduce over the PROCESSES of the group (the conservation sums, ref src/reductions.jl:317-320);
 * op 0 = sum, 1 = min; count <= 16; synchronous; the caller folds its own local tiles first. */
ARMON_API int armon_hip_mgpu_allreduce_host(armon_mgpu*, int op, int count, double* values);

/* border_domain(bsize, side) / ghost_domain(bsize, side) with all ghost layers (ref src/blocking/blocking.jl:141-187,
 * single_strip=false) and real_face_size (ref :210) of a tile of nx x ny real cells, 0-based; outputs nullable. */
ARMON_API int armon_hip_halo_ranges(int64_t nx, int64_t ny, int nghost, int side, armon_range* border,
                                    armon_range* ghost, int64_t* face);

#ifdef __cplusplus
}
#endif
#endif /* ARMON_HIP_H */
