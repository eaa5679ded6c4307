// reductions.hip — dt/CFL min-reduction and conservation sums.
//
// The reference reduces with `mapreduce` over the full linear range masked by `mask`
// (ref src/reductions.jl:79-87) or with a two-step pass through work_1 (:70-78). Here the range is the
// real domain only (like the CPU form, :23-53): pass 1 = one partial per workgroup (per-lane running
// value → 64-lane wavefront shuffle tree → LDS tree over the 4 waves), pass 2 = one workgroup folds the
// partials in a fixed order. No atomics: the sums are reproducible run to run, the min is exact.
#include "common.hpp"
#include "physics.hpp"
#include "reduce.hpp"

using namespace armon;

namespace {

using red::kWave;
using red::op_min;
using red::op_sum;

template <typename OP>
__device__ __forceinline__ double block_reduce(double v, double* lds)
{
    return red::block_reduce<OP, kBlock / kWave>(v, lds, threadIdx.x);
}

// ---- a11: dtCFL (ref src/reductions.jl:2-53) -------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_dtCFL_partial(armon_range r, double dx, double dy, const double* __restrict__ u,
                const double* __restrict__ v, const double* __restrict__ c,
                double* __restrict__ partials)
{
    __shared__ double lds[kBlock / kWave];
    double acc = INFINITY;
    for (int64_t j = blockIdx.y; j < r.col_len; j += gridDim.y) {
        const int64_t base = r.col_start + j * r.col_step + r.row_start;
        for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < r.row_len;
             k += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = base + k;
            acc = phys::mn(acc, phys::dt_cfl_cell(u[i], v[i], c[i], dx, dy));
        }
    }
    double res = block_reduce<op_min>(acc, lds);
    if (threadIdx.x == 0) partials[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = res;
}

template <typename OP, int NOUT>
__global__ void __launch_bounds__(kBlock)
k_fold(const double* __restrict__ partials, int64_t n, double scale, double* __restrict__ out)
{
    __shared__ double lds[kBlock / kWave];
#pragma unroll
    for (int o = 0; o < NOUT; o++) {
        double acc = OP::id();
        for (int64_t k = threadIdx.x; k < n; k += blockDim.x) acc = OP::f(acc, partials[o * n + k]);
        double res = block_reduce<OP>(acc, lds);
        if (threadIdx.x == 0) out[o] = res * scale;
    }
}

// ---- a15: conservation_vars (ref src/reductions.jl:202-259) ----------------------------------------
__global__ void __launch_bounds__(kBlock)
k_conservation_partial(armon_range r, const double* __restrict__ rho, const double* __restrict__ E,
                       double* __restrict__ partials, int64_t n_partials)
{
    __shared__ double lds[kBlock / kWave];
    double mass = 0., energy = 0.;
    for (int64_t j = blockIdx.y; j < r.col_len; j += gridDim.y) {
        const int64_t base = r.col_start + j * r.col_step + r.row_start;
        for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < r.row_len;
             k += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = base + k;
            mass += rho[i];
            energy += rho[i] * E[i];
        }
    }
    double m = block_reduce<op_sum>(mass, lds);
    double e = block_reduce<op_sum>(energy, lds);
    if (threadIdx.x == 0) {
        const int64_t b = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
        partials[b] = m;
        partials[n_partials + b] = e;
    }
}

// Grid for a reduction over a range: enough workgroups to fill the chip (≈8 per CU), never more than
// the range needs; each workgroup strides over rows (y) and row chunks (x).
inline void reduce_grid(const armon_ctx* ctx, const armon_range& r, dim3& grid)
{
    int64_t gx = (r.row_len + kBlock - 1) / kBlock;
    if (gx > 64) gx = 64;
    int64_t target = (int64_t)ctx->n_cu * 8;
    int64_t gy = (target + gx - 1) / gx;
    if (gy > r.col_len) gy = r.col_len;
    if (gy < 1) gy = 1;
    grid = dim3((unsigned)gx, (unsigned)gy, 1);
}

}  // namespace

extern "C" {

int armon_hip_dtCFL_async(armon_ctx* ctx, armon_range r, double dx, double dy, const double* u,
                          const double* v, const double* c, double* result_dev)
{
    ARMON_REQUIRE(ctx != nullptr, "ctx is NULL");
    ARMON_REQUIRE(range_ok(r) && !range_empty(r), "dtCFL needs a non-empty range");
    ARMON_REQUIRE(u && v && c && result_dev, "NULL array");
    dim3 grid;
    reduce_grid(ctx, r, grid);
    const int64_t n = (int64_t)grid.x * grid.y;
    int rc = ensure_partials(ctx, (size_t)n * 2);
    if (rc != ARMON_OK) return rc;
    hipLaunchKernelGGL(k_dtCFL_partial, grid, dim3(kBlock), 0, ctx->stream, r, dx, dy, u, v, c, ctx->partials);
    rc = check_launch("dtCFL_partial");
    if (rc != ARMON_OK) return rc;
    hipLaunchKernelGGL((k_fold<op_min, 1>), dim3(1), dim3(kBlock), 0, ctx->stream, ctx->partials, n, 1.0, result_dev);
    return check_launch("dtCFL_fold");
}

int armon_hip_dtCFL(armon_ctx* ctx, armon_range r, double dx, double dy, const double* u,
                    const double* v, const double* c, double* result_host)
{
    ARMON_REQUIRE(ctx && result_host, "NULL argument");
    int rc = armon_hip_dtCFL_async(ctx, r, dx, dy, u, v, c, ctx->scalars);
    if (rc != ARMON_OK) return rc;
    ARMON_HIP_TRY(hipMemcpyAsync(ctx->host_scalars, ctx->scalars, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ARMON_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *result_host = ctx->host_scalars[0];
    return ARMON_OK;
}

int armon_hip_conservation_vars(armon_ctx* ctx, armon_range r, double ds, const double* rho,
                                const double* E, double out_host[2])
{
    ARMON_REQUIRE(ctx && out_host, "NULL argument");
    ARMON_REQUIRE(range_ok(r), "invalid range");
    ARMON_REQUIRE(rho && E, "NULL array");
    if (range_empty(r)) { out_host[0] = out_host[1] = 0.; return ARMON_OK; }
    dim3 grid;
    reduce_grid(ctx, r, grid);
    const int64_t n = (int64_t)grid.x * grid.y;
    int rc = ensure_partials(ctx, (size_t)n * 2);
    if (rc != ARMON_OK) return rc;
    hipLaunchKernelGGL(k_conservation_partial, grid, dim3(kBlock), 0, ctx->stream, r, rho, E, ctx->partials, n);
    rc = check_launch("conservation_partial");
    if (rc != ARMON_OK) return rc;
    hipLaunchKernelGGL((k_fold<op_sum, 2>), dim3(1), dim3(kBlock), 0, ctx->stream, ctx->partials, n, ds, ctx->scalars + 2);
    rc = check_launch("conservation_fold");
    if (rc != ARMON_OK) return rc;
    ARMON_HIP_TRY(hipMemcpyAsync(ctx->host_scalars + 2, ctx->scalars + 2, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ARMON_HIP_TRY(hipStreamSynchronize(ctx->stream));
    out_host[0] = ctx->host_scalars[2];
    out_host[1] = ctx->host_scalars[3];
    return ARMON_OK;
}

}  // extern "C"
