// context.hip — context management, device memory and error reporting of libarmon_hip.so.
// Mirrors the backend hooks of the reference: create_device / init_backend
// (ref src/parameters.jl:751-778), device_array_type allocation + copyto!
// (ref src/blocking/blocks.jl:36-44,121-143), Base.wait (ref src/parameters.jl:1031-1038),
// device_memory_info (ref src/parameters.jl:921-926, ext/ArmonAMDGPU.jl:25-27).
#include "common.hpp"

#include <cstdlib>

namespace armon {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int ensure_partials(armon_ctx* ctx, size_t n)
{
    if (n <= ctx->partials_cap) return ARMON_OK;
    // A graph captured on this context holds the OLD address in its kernel arguments (and a capture in progress can neither
    // synchronise nor allocate): growing now would leave it with a dangling pointer. The caller runs the larger launch
    // once BEFORE capturing, or destroys the graph first (include/armon_hip.h, armon_hip_graph_*).
    ARMON_REQUIRE(!ctx->capturing, "the reduction scratch would have to grow (%zu -> %zu doubles) inside a stream capture: "
                  "run this launch once before armon_hip_graph_begin", ctx->partials_cap, n);
    ARMON_REQUIRE(ctx->live_graphs == 0, "the reduction scratch would have to grow (%zu -> %zu doubles) while %d captured graph(s) of "
                  "this context point into it: destroy them first, or run the larger launch before capturing",
                  ctx->partials_cap, n, ctx->live_graphs);
    // Grow only between kernels of the same stream: the old buffer may still be read by queued work.
    ARMON_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->partials) ARMON_HIP_TRY(hipFree(ctx->partials));
    ctx->partials = nullptr;
    ctx->partials_cap = 0;
    ARMON_HIP_TRY(hipMalloc((void**)&ctx->partials, n * sizeof(double)));
    ctx->partials_cap = n;
    return ARMON_OK;
}

}  // namespace armon

using namespace armon;

extern "C" {

int armon_hip_flt_size(void) { return (int)sizeof(double); }
int armon_hip_idx_size(void) { return (int)sizeof(int64_t); }
const char* armon_hip_version(void) { return "armon_hip 0.1.0 (gfx950)"; }
const char* armon_hip_last_error(void) { return g_error; }

int armon_hip_device_count(int* count)
{
    ARMON_REQUIRE(count, "count is NULL");
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return fail_hip(e, "hipGetDeviceCount");
    }
    return ARMON_OK;
}

int armon_hip_init(int device_id, void* stream, armon_ctx** out)
{
    ARMON_REQUIRE(out, "ctx out pointer is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        set_error("no HIP device available (%s)", e != hipSuccess ? hipGetErrorString(e) : "count = 0");
        return ARMON_ERR_NO_DEVICE;
    }
    ARMON_REQUIRE(device_id >= 0 && device_id < n, "device_id %d out of range [0,%d)", device_id, n);
    ARMON_HIP_TRY(hipSetDevice(device_id));

    hipDeviceProp_t prop;
    ARMON_HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library holds gfx950 code objects only", device_id, prop.gcnArchName);
        return ARMON_ERR_NO_DEVICE;
    }

    armon_ctx* ctx = new armon_ctx();
    ctx->device = device_id;
    ctx->n_cu = prop.multiProcessorCount;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
        ctx->owns_stream = false;
    } else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; return fail_hip(e, "hipStreamCreate"); }
        ctx->owns_stream = true;
    }
    if ((e = hipMalloc((void**)&ctx->scalars, 16 * sizeof(double))) != hipSuccess ||
        (e = hipHostMalloc((void**)&ctx->host_scalars, 16 * sizeof(double), hipHostMallocDefault)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_start)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_stop)) != hipSuccess) {
        armon_hip_destroy(ctx);
        return fail_hip(e, "context allocation");
    }
    int rc = ensure_partials(ctx, 8192);
    if (rc != ARMON_OK) { armon_hip_destroy(ctx); return rc; }
    for (const char* knob : {"ARMON_XS_NITER", "ARMON_Y_SEG", "ARMON_SWEEP_ALIGN", "ARMON_Y_COLS1", "ARMON_X_XCD", "ARMON_X_ROWS", "ARMON_Y_SX", "ARMON_COPY_NT"}) {
        const char* v = getenv(knob);
        if (v && *v) (void)armon_hip_set_tuning(ctx, knob, atoi(v));
    }
    *out = ctx;
    return ARMON_OK;
}

int armon_hip_set_tuning(armon_ctx* ctx, const char* knob, int value)
{
    ARMON_REQUIRE(ctx && knob, "NULL argument");
    if (!strcmp(knob, "ARMON_XS_NITER")) ctx->tune_xs_niter = value > 0 ? value : 0;
    else if (!strcmp(knob, "ARMON_Y_SEG")) ctx->tune_y_seg = value > 0 ? value : 0;
    else if (!strcmp(knob, "ARMON_SWEEP_ALIGN")) ctx->tune_align = value < 0 ? 1 : (value != 0);
    else if (!strcmp(knob, "ARMON_Y_COLS1")) ctx->tune_y_cols1 = value > 0;
    else if (!strcmp(knob, "ARMON_X_XCD")) ctx->tune_x_xcd = value < 0 ? -1 : (value > 0);
    else if (!strcmp(knob, "ARMON_X_ROWS")) ctx->tune_x_rows = (value == 1 || value == 2) ? value : 0;
    else if (!strcmp(knob, "ARMON_Y_SX")) ctx->tune_y_sx = (value == 1 || value == 2) ? value : 0;
    else if (!strcmp(knob, "ARMON_COPY_NT")) ctx->tune_copy_nt = value & 3;
    else ARMON_REQUIRE(false, "unknown tuning knob '%s'", knob);
    return ARMON_OK;
}

int armon_hip_get_tuning(armon_ctx* ctx, const char* knob, int* value)
{
    ARMON_REQUIRE(ctx && knob && value, "NULL argument");
    if (!strcmp(knob, "ARMON_XS_NITER")) *value = ctx->tune_xs_niter;
    else if (!strcmp(knob, "ARMON_Y_SEG")) *value = ctx->tune_y_seg;
    else if (!strcmp(knob, "ARMON_SWEEP_ALIGN")) *value = ctx->tune_align;
    else if (!strcmp(knob, "ARMON_Y_COLS1")) *value = ctx->tune_y_cols1;
    else if (!strcmp(knob, "ARMON_X_XCD")) *value = ctx->tune_x_xcd;
    else if (!strcmp(knob, "ARMON_X_ROWS")) *value = ctx->tune_x_rows;
    else if (!strcmp(knob, "ARMON_Y_SX")) *value = ctx->tune_y_sx;
    else if (!strcmp(knob, "ARMON_COPY_NT")) *value = ctx->tune_copy_nt;
    else if (!strcmp(knob, "Y_RUN_ROWS")) *value = ctx->tune_y_seg > 0 ? ctx->tune_y_seg : ctx->seg_value;
    else ARMON_REQUIRE(false, "unknown tuning knob '%s'", knob);
    return ARMON_OK;
}

int armon_hip_destroy(armon_ctx* ctx)
{
    if (!ctx) return ARMON_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    disown_graphs(ctx);              // graphs still alive lose their executables here (their handles stay valid to destroy)
    if (ctx->partials) (void)hipFree(ctx->partials);
    if (ctx->scalars) (void)hipFree(ctx->scalars);
    if (ctx->host_scalars) (void)hipHostFree(ctx->host_scalars);
    if (ctx->ev_pool) {
        for (int i = 0; i < ARMON_HIP_MAX_EVENTS; i++)
            if (ctx->ev_pool[i]) (void)hipEventDestroy(ctx->ev_pool[i]);
        delete[] ctx->ev_pool;
    }
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return ARMON_OK;
}

int armon_hip_sync(armon_ctx* ctx)
{
    ARMON_REQUIRE(ctx, "ctx is NULL");
    ARMON_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ARMON_OK;
}

void* armon_hip_stream(armon_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int armon_hip_device_memory_info(armon_ctx* ctx, size_t* free_bytes, size_t* total_bytes)
{
    ARMON_REQUIRE(ctx && free_bytes && total_bytes, "NULL argument");
    ARMON_HIP_TRY(hipSetDevice(ctx->device));
    ARMON_HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
    return ARMON_OK;
}

int armon_hip_device_name(armon_ctx* ctx, char* buf, size_t buf_len)
{
    ARMON_REQUIRE(ctx && buf && buf_len > 0, "NULL argument");
    hipDeviceProp_t prop;
    ARMON_HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    // some boxes report an empty marketing name: fall back to the architecture
    snprintf(buf, buf_len, "%s (%s, %d CUs)", prop.name[0] ? prop.name : "AMD GPU gfx950", prop.gcnArchName, prop.multiProcessorCount);
    return ARMON_OK;
}

int armon_hip_malloc(armon_ctx* ctx, size_t bytes, void** ptr)
{
    ARMON_REQUIRE(ctx && ptr, "NULL argument");
    *ptr = nullptr;
    if (bytes == 0) return ARMON_OK;
    ARMON_HIP_TRY(hipSetDevice(ctx->device));
    ARMON_HIP_TRY(hipMalloc(ptr, bytes));
    return ARMON_OK;
}

int armon_hip_free(armon_ctx* ctx, void* ptr)
{
    ARMON_REQUIRE(ctx, "ctx is NULL");
    if (!ptr) return ARMON_OK;
    ARMON_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ARMON_HIP_TRY(hipFree(ptr));
    return ARMON_OK;
}

int armon_hip_memcpy(armon_ctx* ctx, void* dst, const void* src, size_t bytes, int kind)
{
    ARMON_REQUIRE(ctx, "ctx is NULL");
    if (bytes == 0) return ARMON_OK;
    ARMON_REQUIRE(dst && src, "NULL pointer in memcpy");
    hipMemcpyKind k;
    switch (kind) {
    case ARMON_MEMCPY_H2D: k = hipMemcpyHostToDevice; break;
    case ARMON_MEMCPY_D2H: k = hipMemcpyDeviceToHost; break;
    case ARMON_MEMCPY_D2D: k = hipMemcpyDeviceToDevice; break;
    default: ARMON_REQUIRE(false, "unknown memcpy kind %d", kind);
    }
    ARMON_HIP_TRY(hipMemcpyAsync(dst, src, bytes, k, ctx->stream));
    // Pageable host memory: the copy is staged, and the caller may reuse the host buffer at once.
    if (k != hipMemcpyDeviceToDevice) ARMON_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ARMON_OK;
}

int armon_hip_malloc_host(armon_ctx* ctx, size_t bytes, void** ptr)
{
    ARMON_REQUIRE(ctx && ptr, "NULL argument");
    *ptr = nullptr;
    if (bytes == 0) return ARMON_OK;
    ARMON_HIP_TRY(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    return ARMON_OK;
}

int armon_hip_free_host(armon_ctx* ctx, void* ptr)
{
    ARMON_REQUIRE(ctx, "ctx is NULL");
    if (ptr) ARMON_HIP_TRY(hipHostFree(ptr));
    return ARMON_OK;
}

int armon_hip_memcpy_async(armon_ctx* ctx, void* dst, const void* src, size_t bytes, int kind)
{
    ARMON_REQUIRE(ctx, "ctx is NULL");
    if (bytes == 0) return ARMON_OK;
    ARMON_REQUIRE(dst && src, "NULL pointer in memcpy_async");
    hipMemcpyKind k;
    switch (kind) {
    case ARMON_MEMCPY_H2D: k = hipMemcpyHostToDevice; break;
    case ARMON_MEMCPY_D2H: k = hipMemcpyDeviceToHost; break;
    case ARMON_MEMCPY_D2D: k = hipMemcpyDeviceToDevice; break;
    default: ARMON_REQUIRE(false, "unknown memcpy kind %d", kind);
    }
    ARMON_HIP_TRY(hipMemcpyAsync(dst, src, bytes, k, ctx->stream));
    return ARMON_OK;
}

int armon_hip_memset(armon_ctx* ctx, void* dst, int byte_value, size_t bytes)
{
    ARMON_REQUIRE(ctx, "ctx is NULL");
    if (bytes == 0) return ARMON_OK;
    ARMON_REQUIRE(dst, "NULL pointer in memset");
    ARMON_HIP_TRY(hipMemsetAsync(dst, byte_value, bytes, ctx->stream));
    return ARMON_OK;
}

}  // extern "C"

namespace {
struct copy4_args { const double2* in[4]; double2* out[4]; };

typedef double vdouble2 __attribute__((ext_vector_type(2)));
// NT bit 0: non-temporal loads, bit 1: non-temporal stores (ARMON_COPY_NT, armon_hip_set_tuning; the measurement aid takes the
// fastest form the device has, so that "the same-device copy" stays a ceiling: profiles/r05_ab_nt.txt)
template <int NT>
__global__ void __launch_bounds__(256) k_stream_copy4(copy4_args p, size_t n2)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    vdouble2 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const vdouble2* q = reinterpret_cast<const vdouble2*>(p.in[k] + i);
        v[k] = (NT & 1) ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        vdouble2* q = reinterpret_cast<vdouble2*>(p.out[k] + i);
        if (NT & 2) __builtin_nontemporal_store(v[k], q);
        else *q = v[k];
    }
}
}  // namespace

extern "C" {

int armon_hip_stream_copy4(armon_ctx* ctx, const void* const in[4], void* const out[4], size_t bytes)
{
    ARMON_REQUIRE(ctx && in && out, "NULL argument");
    ARMON_REQUIRE(bytes % 16 == 0 && bytes / 16 / 256 < (1ull << 31), "bytes = %zu: multiple of 16 expected", bytes);
    if (bytes == 0) return ARMON_OK;
    copy4_args p;
    for (int k = 0; k < 4; k++) {
        ARMON_REQUIRE(in[k] && out[k], "NULL array");
        p.in[k] = static_cast<const double2*>(in[k]);
        p.out[k] = static_cast<double2*>(out[k]);
    }
    const size_t n2 = bytes / 16;
    const dim3 grid((unsigned)((n2 + 255) / 256)), block(256);
    switch (ctx->tune_copy_nt & 3) {
    case 0: hipLaunchKernelGGL(k_stream_copy4<0>, grid, block, 0, ctx->stream, p, n2); break;
    case 1: hipLaunchKernelGGL(k_stream_copy4<1>, grid, block, 0, ctx->stream, p, n2); break;
    case 2: hipLaunchKernelGGL(k_stream_copy4<2>, grid, block, 0, ctx->stream, p, n2); break;
    default: hipLaunchKernelGGL(k_stream_copy4<3>, grid, block, 0, ctx->stream, p, n2);
    }
    return check_launch("stream_copy4");
}

int armon_hip_timer_start(armon_ctx* ctx)
{
    ARMON_REQUIRE(ctx, "ctx is NULL");
    ARMON_HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    return ARMON_OK;
}

int armon_hip_timer_stop(armon_ctx* ctx, double* elapsed_ms)
{
    ARMON_REQUIRE(ctx && elapsed_ms, "NULL argument");
    ARMON_HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
    ARMON_HIP_TRY(hipEventSynchronize(ctx->ev_stop));
    float ms = 0.f;
    ARMON_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
    *elapsed_ms = (double)ms;
    return ARMON_OK;
}

int armon_hip_event_record(armon_ctx* ctx, int slot)
{
    ARMON_REQUIRE(ctx, "ctx is NULL");
    ARMON_REQUIRE(slot >= 0 && slot < ARMON_HIP_MAX_EVENTS, "event slot %d out of range", slot);
    if (!ctx->ev_pool) ctx->ev_pool = new hipEvent_t[ARMON_HIP_MAX_EVENTS]();
    if (!ctx->ev_pool[slot]) ARMON_HIP_TRY(hipEventCreate(&ctx->ev_pool[slot]));
    ARMON_HIP_TRY(hipEventRecord(ctx->ev_pool[slot], ctx->stream));
    return ARMON_OK;
}

int armon_hip_event_sync(armon_ctx* ctx, int slot)
{
    ARMON_REQUIRE(ctx, "ctx is NULL");
    ARMON_REQUIRE(ctx->ev_pool && slot >= 0 && slot < ARMON_HIP_MAX_EVENTS && ctx->ev_pool[slot],
                  "event %d was not recorded", slot);
    ARMON_HIP_TRY(hipEventSynchronize(ctx->ev_pool[slot]));
    return ARMON_OK;
}

int armon_hip_event_elapsed_ms(armon_ctx* ctx, int a, int b, double* elapsed_ms)
{
    ARMON_REQUIRE(ctx && elapsed_ms, "NULL argument");
    ARMON_REQUIRE(ctx->ev_pool && a >= 0 && b >= 0 && a < ARMON_HIP_MAX_EVENTS && b < ARMON_HIP_MAX_EVENTS &&
                  ctx->ev_pool[a] && ctx->ev_pool[b], "events %d/%d were not recorded", a, b);
    ARMON_HIP_TRY(hipEventSynchronize(ctx->ev_pool[b]));
    float ms = 0.f;
    ARMON_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_pool[a], ctx->ev_pool[b]));
    *elapsed_ms = (double)ms;
    return ARMON_OK;
}

}  // extern "C"
