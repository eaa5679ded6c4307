// common.hpp — context, error reporting and launch helpers shared by every translation unit of
// libarmon_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/armon_hip.h"

struct armon_graph;
struct armon_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    // scratch owned by the context (reduction partials, device scalars, pointer tables)
    double* partials = nullptr;      // [partials_cap]
    size_t partials_cap = 0;
    double* scalars = nullptr;       // [16] device doubles: results of reductions
    double* host_scalars = nullptr;  // [16] pinned host mirror
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    hipEvent_t* ev_pool = nullptr;   // [ARMON_HIP_MAX_EVENTS], created on first use
    int n_cu = 256;
    // Tuning knobs of the fused sweeps, read from the environment ONCE (armon_hip_init) or set with
    // armon_hip_set_tuning (A/B tools): nothing on the launch path looks at the environment.
    int tune_xs_niter = 0;           // ARMON_XS_NITER: strips per wave of the X sweep (0 = built-in)
    int tune_y_seg = 0;              // ARMON_Y_SEG: rows per run of the Y march (0 = y_run_length's choice)
    int tune_align = 1;              // ARMON_SWEEP_ALIGN: 0 = unaligned block / strip origins
    int tune_y_cols1 = 0;            // ARMON_Y_COLS1: fp32 Y march with one column per lane
    int tune_x_xcd = -1;             // ARMON_X_XCD: XCD-aware workgroup placement in the X sweep (-1 = by precision: fp64 on, fp32 off)
    int tune_x_rows = 0;             // ARMON_X_ROWS: workgroup of the X sweep = 1: one strip of 4 rows, 2: 4 strips of one row, 0: by precision
    int tune_copy_nt = 0;            // ARMON_COPY_NT: the measurement aid armon_hip_stream_copy4 with nt loads (bit 0) / stores (bit 1)
    int tune_y_sx = 0;               // ARMON_Y_SX: store exchange of the Y march = 1: always, 2: never, 0: when the row pitch is not a multiple of a sector
    // y_run_length's last answer (it depends on the shape only)
    int64_t seg_nx = -1, seg_ny = -1;
    int seg_lag = -1, seg_cols = -1, seg_value = 0;
    // graphs captured on this context bake the address of `partials` in: it must not move while one is alive
    int live_graphs = 0;
    armon_graph* graphs = nullptr;   // the live ones (intrusive list): a context destroyed first disowns them, see context.hip
    bool capturing = false;
};

namespace armon {

void set_error(const char* fmt, ...);

inline int fail_hip(hipError_t e, const char* what)
{
    set_error("%s: %s", what, hipGetErrorString(e));
    return ARMON_ERR_HIP;
}

#define ARMON_HIP_TRY(expr)                                             \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) return ::armon::fail_hip(e_, #expr);      \
    } while (0)

#define ARMON_REQUIRE(cond, ...)                                        \
    do {                                                                \
        if (!(cond)) {                                                  \
            ::armon::set_error(__VA_ARGS__);                            \
            return ARMON_ERR_INVALID_ARG;                               \
        }                                                               \
    } while (0)

// Check the kernel launch itself (bad configuration shows up here, not at the next sync).
inline int check_launch(const char* name)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", name, hipGetErrorString(e));
        return ARMON_ERR_HIP;
    }
    return ARMON_OK;
}

inline bool range_ok(const armon_range& r)
{
    // row_start may be negative (ghost_domain of a first side shifts the row range below 1): only the
    // first cell index has to be valid.
    return r.col_len >= 0 && r.row_len >= 0 && r.col_start + r.row_start >= 0 && r.col_step >= 0;
}
inline bool range_empty(const armon_range& r) { return r.col_len == 0 || r.row_len == 0; }

constexpr int kBlock = 256;  // 4 waves of 64

// 2-D launch shape for a range: x covers a row (x-contiguous cells → coalesced), y covers rows.
inline void range_grid(const armon_range& r, int cells_per_thread, dim3& grid, dim3& block)
{
    block = dim3(kBlock, 1, 1);
    int64_t per_block = (int64_t)kBlock * cells_per_thread;
    int64_t gx = (r.row_len + 15 + per_block - 1) / per_block;      // + the sector-alignment shift of the row walks
    int64_t gy = r.col_len < 65535 ? r.col_len : 65535;
    grid = dim3((unsigned)gx, (unsigned)gy, 1);
}

int ensure_partials(armon_ctx* ctx, size_t n);
// staged_kernels.hip: pack_to_array! / unpack_from_array! of BOTH sides of an axis in one launch (the multi-GPU exchange)
template <typename T>
int pack_pair(armon_ctx* ctx, const armon_range r[2], int nghost, int64_t face, T* const array[2], int nvars, T* const* vars, bool pack);
void disown_graphs(armon_ctx* ctx);      // dt_state.hip

}  // namespace armon
