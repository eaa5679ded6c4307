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
using red::op_max;
using red::op_min;
using red::op_sum;

template <typename OP, typename T>
__device__ __forceinline__ T block_reduce(T v, T* lds)
{
    return red::block_reduce<OP, kBlock / kWave>(v, lds, threadIdx.x);
}

// ---- a11: dtCFL (ref src/reductions.jl:2-53) -------------------------------------------------------
// min over the cells of min(dx / a, dy / b), a = max(|u+c|, |u-c|), b = max(|v+c|, |v-c|). A correctly rounded division
// is monotone in its divisor, so that minimum is min(dx / max a, dy / max b) — the same bits with two divisions per
// LAUNCH instead of two per cell (the fused sweeps reduce their CFL step the same way, fused_sweep_impl.hpp).
// Two cells per lane with 16-B loads (8-B for fp32) wherever a row of the range starts on such a boundary in all three arrays
// (every row of an even-pitched block does), one cell per lane otherwise. max(|u + c|, |u - c|) is evaluated as |u| + |c|: the
// same bits (the two candidates are fl(|u| + |c|) and |fl(|u| - |c|)| in some order; rounding is monotone and symmetric).
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_dtCFL_partial(armon_range r, int rows_per_block, const T* __restrict__ u, const T* __restrict__ v, const T* __restrict__ c,
                T* __restrict__ partials)
{
    typedef T V2 __attribute__((ext_vector_type(2)));
    __shared__ T lds[kBlock / kWave];
    T au = T(0.), av = T(0.);
    auto add = [&](T uu, T vv, T cc) {
        // amax: a NaN in one cell sticks (the reference's reduction ends in `Invalid time step`, ref src/solver_state.jl:123)
        au = phys::amax(au, phys::abs_(uu) + phys::abs_(cc));
        av = phys::amax(av, phys::abs_(vv) + phys::abs_(cc));
    };
    // workgroup (bx, by) reads rows_per_block CONSECUTIVE rows, and consecutive workgroups consecutive memory: the three
    // streams are walked in address order (a grid that strides over the rows keeps 64 distant rows open at once)
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
    const int64_t j_lo = (int64_t)blockIdx.y * rows_per_block;
    const int64_t j_hi = j_lo + rows_per_block < r.col_len ? j_lo + rows_per_block : r.col_len;
    for (int64_t j = j_lo; j < j_hi; j++) {
        const int64_t base = r.col_start + j * r.col_step + r.row_start;
        constexpr uintptr_t mask = 2 * sizeof(T) - 1;
        const bool pairs = (((uintptr_t)(u + base) | (uintptr_t)(v + base) | (uintptr_t)(c + base)) & mask) == 0;   // uniform
        if (pairs) {
            const int64_t np = r.row_len >> 1;
            for (int64_t k = tid; k < np; k += nthreads) {
                const V2 uu = *reinterpret_cast<const V2*>(u + base + 2 * k);
                const V2 vv = *reinterpret_cast<const V2*>(v + base + 2 * k);
                const V2 cc = *reinterpret_cast<const V2*>(c + base + 2 * k);
                add(uu.x, vv.x, cc.x);
                add(uu.y, vv.y, cc.y);
            }
            if ((r.row_len & 1) && tid == 0) add(u[base + r.row_len - 1], v[base + r.row_len - 1], c[base + r.row_len - 1]);
        } else {
            for (int64_t k = tid; k < r.row_len; k += nthreads) add(u[base + k], v[base + k], c[base + k]);
        }
    }
    const T ru = block_reduce<op_max>(au, lds);
    const T rv = block_reduce<op_max>(av, lds);
    if (threadIdx.x == 0) {
        const int64_t b = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
        partials[2 * b] = ru;
        partials[2 * b + 1] = rv;
    }
}

// max over the n partial pairs, then the two divisions; n = 0: dx / 0 = +inf, the minimum over no cell
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_dtCFL_fold(const T* __restrict__ partials, int64_t n, T dx, T dy, T* __restrict__ out)
{
    __shared__ T lds[kBlock / kWave];
    T au = T(0.), av = T(0.);
    for (int64_t k = threadIdx.x; k < n; k += blockDim.x) {
        au = phys::amax(au, partials[2 * k]);
        av = phys::amax(av, partials[2 * k + 1]);
    }
    au = block_reduce<op_max>(au, lds);
    av = block_reduce<op_max>(av, lds);
    if (threadIdx.x == 0) out[0] = phys::mn_nan(dx / au, dy / av);
}

template <typename OP, int NOUT, typename T>
__global__ void __launch_bounds__(kBlock)
k_fold(const T* __restrict__ partials, int64_t n, T scale, T* __restrict__ out)
{
    __shared__ T lds[kBlock / kWave];
#pragma unroll
    for (int o = 0; o < NOUT; o++) {
        T acc = OP::template id<T>();
        for (int64_t k = threadIdx.x; k < n; k += blockDim.x) acc = OP::f(acc, partials[o * n + k]);
        T res = block_reduce<OP>(acc, lds);
        if (threadIdx.x == 0) out[o] = res * scale;
    }
}

// ---- a15: conservation_vars (ref src/reductions.jl:202-259) ----------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_conservation_partial(armon_range r, const T* __restrict__ rho, const T* __restrict__ E,
                       T* __restrict__ partials, int64_t n_partials)
{
    __shared__ T lds[kBlock / kWave];
    T mass = 0., energy = 0.;
    for (int64_t j = blockIdx.y; j < r.col_len; j += gridDim.y) {
        const int64_t base = r.col_start + j * r.col_step + r.row_start;
        for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < r.row_len;
             k += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = base + k;
            mass += rho[i];
            energy += rho[i] * E[i];
        }
    }
    T m = block_reduce<op_sum>(mass, lds);
    T e = block_reduce<op_sum>(energy, lds);
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
    if (gx < 1) gx = 1;
    int64_t target = (int64_t)ctx->n_cu * 8;
    int64_t gy = (target + gx - 1) / gx;
    if (gy > r.col_len) gy = r.col_len;
    if (gy < 1) gy = 1;
    grid = dim3((unsigned)gx, (unsigned)gy, 1);
}

}  // namespace

namespace {

template <typename T>
int dtCFL_async_impl(armon_ctx* ctx, armon_range r, T dx, T dy, const T* u, const T* v, const T* c, T* result_dev)
{
    ARMON_REQUIRE(ctx != nullptr, "ctx is NULL");
    ARMON_REQUIRE(range_ok(r), "invalid range");
    ARMON_REQUIRE(u && v && c && result_dev, "NULL array");
    // the minimum over no cell is +inf (the reference's mapreduce has init = Inf, ref src/reductions.jl:79-87): an empty
    // range folds zero partials
    dim3 grid(1, 1, 1);
    int rows_per_block = 1;
    if (!range_empty(r)) {
        // two cells per thread across a row; as many consecutive rows per workgroup as keep the partials under 32768 pairs
        const int64_t gx = (r.row_len + 2 * kBlock - 1) / (2 * kBlock);
        int64_t rpb = (gx * r.col_len + 32767) / 32768;
        if (rpb < 4) rpb = 4;
        rows_per_block = (int)rpb;
        grid = dim3((unsigned)gx, (unsigned)((r.col_len + rpb - 1) / rpb), 1);
    }
    const int64_t n = range_empty(r) ? 0 : (int64_t)grid.x * grid.y;
    int rc = ensure_partials(ctx, (size_t)(n > 0 ? n : 1) * 2);
    if (rc != ARMON_OK) return rc;
    T* partials = reinterpret_cast<T*>(ctx->partials);
    if (n > 0) {
        hipLaunchKernelGGL(k_dtCFL_partial<T>, grid, dim3(kBlock), 0, ctx->stream, r, rows_per_block, u, v, c, partials);
        rc = check_launch("dtCFL_partial");
        if (rc != ARMON_OK) return rc;
    }
    hipLaunchKernelGGL(k_dtCFL_fold<T>, dim3(1), dim3(kBlock), 0, ctx->stream, partials, n, dx, dy, result_dev);
    return check_launch("dtCFL_fold");
}

template <typename T>
int dtCFL_impl(armon_ctx* ctx, armon_range r, T dx, T dy, const T* u, const T* v, const T* c, T* result_host)
{
    ARMON_REQUIRE(ctx && result_host, "NULL argument");
    T* dev = reinterpret_cast<T*>(ctx->scalars);
    T* host = reinterpret_cast<T*>(ctx->host_scalars);
    int rc = dtCFL_async_impl<T>(ctx, r, dx, dy, u, v, c, dev);
    if (rc != ARMON_OK) return rc;
    ARMON_HIP_TRY(hipMemcpyAsync(host, dev, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    ARMON_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *result_host = host[0];
    return ARMON_OK;
}

template <typename T>
int conservation_vars_impl(armon_ctx* ctx, armon_range r, T ds, const T* rho, const T* E, T out_host[2])
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
    T* partials = reinterpret_cast<T*>(ctx->partials);
    T* dev = reinterpret_cast<T*>(ctx->scalars) + 4;
    T* host = reinterpret_cast<T*>(ctx->host_scalars) + 4;
    hipLaunchKernelGGL(k_conservation_partial<T>, grid, dim3(kBlock), 0, ctx->stream, r, rho, E, partials, n);
    rc = check_launch("conservation_partial");
    if (rc != ARMON_OK) return rc;
    hipLaunchKernelGGL((k_fold<op_sum, 2, T>), dim3(1), dim3(kBlock), 0, ctx->stream, partials, n, ds, dev);
    rc = check_launch("conservation_fold");
    if (rc != ARMON_OK) return rc;
    ARMON_HIP_TRY(hipMemcpyAsync(host, dev, 2 * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    ARMON_HIP_TRY(hipStreamSynchronize(ctx->stream));
    out_host[0] = host[0];
    out_host[1] = host[1];
    return ARMON_OK;
}

}  // namespace

#define ARMON_EXPORT(name, impl, PARAMS, ARGS)                                            \
    int armon_hip_##name(PARAMS(double)) { return impl<double> ARGS; }                    \
    int armon_hip_##name##_f32(PARAMS(float)) { return impl<float> ARGS; }

extern "C" {

#define P_DTA(T) armon_ctx* ctx, armon_range r, T dx, T dy, const T* u, const T* v, const T* c, T* result_dev
ARMON_EXPORT(dtCFL_async, dtCFL_async_impl, P_DTA, (ctx, r, dx, dy, u, v, c, result_dev))

#define P_DT(T) armon_ctx* ctx, armon_range r, T dx, T dy, const T* u, const T* v, const T* c, T* result_host
ARMON_EXPORT(dtCFL, dtCFL_impl, P_DT, (ctx, r, dx, dy, u, v, c, result_host))

#define P_CV(T) armon_ctx* ctx, armon_range r, T ds, const T* rho, const T* E, T out_host[2]
ARMON_EXPORT(conservation_vars, conservation_vars_impl, P_CV, (ctx, r, ds, rho, E, out_host))

}  // extern "C"
