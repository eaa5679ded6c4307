// Probe (round 4): issue cost of the fp64 vector instructions the fused sweeps are made of, on gfx950.
// One kernel per instruction: a loop of 8 x 16 independent instances (8 register chains), timed with s_memtime by one
// wave per SIMD and by four (the X sweep's occupancy). Prints cycles per instruction per wave at 1 wave/SIMD and the
// SIMD's cycles per instruction at 4 waves/SIMD (device-wide wall time x clock / instructions per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITER = 2000, UNROLL = 16, CH = 8;

#define OP1(name, text)                                                                                           \
    __global__ __launch_bounds__(256) void k_##name(double* out, long long* cyc, double a0)                      \
    {                                                                                                             \
        double r[CH];                                                                                             \
        for (int k = 0; k < CH; k++) r[k] = a0 + k + threadIdx.x * 1e-3;                                          \
        const long long t0 = __builtin_readcyclecounter();                                                        \
        for (int it = 0; it < ITER; it++) {                                                                       \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                                  \
                _Pragma("unroll") for (int k = 0; k < CH; k++) asm volatile(text : "+v"(r[k]) : "v"(a0));        \
            }                                                                                                     \
        }                                                                                                         \
        const long long t1 = __builtin_readcyclecounter();                                                        \
        double s = 0;                                                                                             \
        for (int k = 0; k < CH; k++) s += r[k];                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                           \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                          \
    }

#define OP32(name, text)                                                                                          \
    __global__ __launch_bounds__(256) void k_##name(double* out, long long* cyc, double a0d)                     \
    {                                                                                                             \
        float r[CH];                                                                                              \
        const float a0 = (float)a0d;                                                                              \
        for (int k = 0; k < CH; k++) r[k] = a0 + k + threadIdx.x * 1e-3f;                                         \
        const long long t0 = __builtin_readcyclecounter();                                                        \
        for (int it = 0; it < ITER; it++) {                                                                       \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                                  \
                _Pragma("unroll") for (int k = 0; k < CH; k++) asm volatile(text : "+v"(r[k]) : "v"(a0));        \
            }                                                                                                     \
        }                                                                                                         \
        const long long t1 = __builtin_readcyclecounter();                                                        \
        float s = 0;                                                                                              \
        for (int k = 0; k < CH; k++) s += r[k];                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                           \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                          \
    }

OP1(fma, "v_fma_f64 %0, %0, %1, %1")
OP1(mul, "v_mul_f64 %0, %0, %1")
OP1(add, "v_add_f64 %0, %0, %1")
OP1(max, "v_max_f64 %0, %0, %1")
OP1(min, "v_min_f64 %0, %0, %1")
OP1(rcp, "v_rcp_f64 %0, %0")
OP1(rsq, "v_rsq_f64 %0, %0")
OP1(sqrt, "v_sqrt_f64 %0, %0")
OP1(mov64, "v_mov_b64 %0, %1")
OP32(fma32, "v_fma_f32 %0, %0, %1, %1")
OP32(rcp32, "v_rcp_f32 %0, %0")
OP32(dpp, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
OP32(dpprow, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
OP32(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
OP32(cndmask_s, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
OP32(cndmask_c, "v_cndmask_b32_e64 %0, %0, 1.0, vcc")
OP32(cndmask_ab, "v_cndmask_b32_e32 %0, %1, %0, vcc")
OP32(cmp_cnd, "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
OP1(cmp, "v_cmp_gt_f64 vcc, %0, %1")
OP1(pkfma, "v_pk_fma_f32 %0, %0, %1, %1")
OP1(pkmul, "v_pk_mul_f32 %0, %0, %1")
OP1(div_scale, "v_div_scale_f64 %0, vcc, %0, %1, %0")
OP1(div_fmas, "v_div_fmas_f64 %0, %0, %1, %1")
OP1(div_fixup, "v_div_fixup_f64 %0, %0, %1, %1")
OP1(frexp, "v_frexp_mant_f64 %0, %0")
OP1(ldexp, "v_ldexp_f64 %0, %0, 1")

typedef void (*kern_t)(double*, long long*, double);

static void measure(const char* name, kern_t k, double* out, long long* cyc, double mhz)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double n_inst = (double)ITER * UNROLL * CH;
    double res[2] = {0, 0}, wall[2] = {0, 0};
    const int wg_threads[2] = {64, 256};
    for (int m = 0; m < 2; m++) {
        // 1024 workgroups of 64 lanes = one wave per SIMD; 1024 of 256 = 4 waves per CU... use 256 CUs x 4 SIMDs:
        // 256-lane workgroups spread their 4 waves over the 4 SIMDs, so 4 waves per SIMD needs 4 workgroups per CU
        const int blocks = m == 0 ? 1024 : 1024;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(blocks), dim3(wg_threads[m]), 0, 0, out, cyc, 1.0000001);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long h[8];
        CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
        res[m] = (double)h[0] / n_inst;
        wall[m] = ms;
    }
    // m = 1: 1024 workgroups x 4 waves over 1024 SIMDs = 4 waves per SIMD
    const double simd_cyc = wall[1] * 1e-3 * mhz * 1e6 / (4 * n_inst);
    printf("%-10s  1 wave/SIMD: %6.2f cyc/inst (s_memtime ticks x clock ratio not applied)   4 waves/SIMD: wall %.3f ms = %5.2f SIMD cycles per instruction\n",
           name, res[0], wall[1], simd_cyc);
    fflush(stdout);
}

int main()
{
    double* out; long long* cyc;
    CK(hipMalloc(&out, 1024 * 256 * 8)); CK(hipMalloc(&cyc, 1024 * 8));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const double mhz = prop.clockRate / 1e3;
    printf("device %s, %d CUs, clockRate %.0f MHz (the instruction counts below assume the clock stays there)\n", prop.gcnArchName,
           prop.multiProcessorCount, mhz);
#define M(name) measure(#name, k_##name, out, cyc, mhz);
    M(fma) M(fma) M(mul) M(add) M(max) M(min) M(rcp) M(rsq) M(sqrt) M(mov64) M(fma32) M(rcp32) M(dpp) M(dpprow) M(cndmask) M(cndmask_s) M(cndmask_c) M(cndmask_ab) M(cmp_cnd) M(cmp)
    M(pkfma) M(pkmul) M(div_scale) M(div_fmas) M(div_fixup) M(frexp) M(ldexp)
    return 0;
}
