// Probe: the practical HBM ceiling for the fused sweep's traffic pattern on this device — 4 arrays read,
// 4 arrays written, (16384+8)^2 doubles each (17.2 GB per launch) — as a plain streaming kernel, with NF
// dependent fp64 FMAs per element mixed in (0 = pure copy) to find where the VALU work stops hiding
// behind the memory stream. Prints ms and TB/s per variant (median of 10 launches).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct ptrs { const double2* in[4]; double2* out[4]; };

// NT bit 0: nontemporal loads, bit 1: nontemporal stores
template <int NT>
__global__ __launch_bounds__(256) void k_stream_nt(ptrs p, size_t n2)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    double2 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (NT & 1) { v[k].x = __builtin_nontemporal_load(&p.in[k][i].x); v[k].y = __builtin_nontemporal_load(&p.in[k][i].y); }
        else v[k] = p.in[k][i];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (NT & 2) { __builtin_nontemporal_store(v[k].x, &p.out[k][i].x); __builtin_nontemporal_store(v[k].y, &p.out[k][i].y); }
        else p.out[k][i] = v[k];
    }
}

template <int NF>
__global__ __launch_bounds__(256) void k_stream(ptrs p, size_t n2, double a, double b)
{
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        double2 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = p.in[k][i];
        // NF dependent FMAs per element, spread over the 8 doubles this lane holds (4 chains of 2)
#pragma unroll
        for (int f = 0; f < NF; f++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k].x = __builtin_fma(v[k].x, a, b);
                v[k].y = __builtin_fma(v[k].y, a, b);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) p.out[k][i] = v[k];
    }
}

template <int NF>
static int run(ptrs p, size_t n2, int blocks, const char* tag)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int it = 0; it < 12; it++) {
        CK(hipEventRecord(e0));
        k_stream<NF><<<blocks, 256>>>(p, n2, 1.0000001, 1e-9);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (it >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    double med = ms[ms.size() / 2];
    double bytes = 8.0 * n2 * 16;
    printf("%-28s blocks %6d  fma/cell %4d  median %.3f ms  min %.3f ms  %.2f TB/s (median)\n", tag, blocks, 4 * NF, med, ms[0],
           bytes / med / 1e9);
    return 0;
}

int main()
{
    const size_t row = 16384 + 8, n = row * row, n2 = n / 2;
    ptrs p;
    for (int k = 0; k < 4; k++) {
        double *a, *b;
        CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
        CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
        p.in[k] = (const double2*)a; p.out[k] = (double2*)b;
    }
    CK(hipDeviceSynchronize());
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int blocks = (int)((n2 + 255) / 256);
        for (int rep = 0; rep < 2; rep++)
        for (int nt = 0; nt < 4; nt++) {
            std::vector<float> ms;
            for (int it = 0; it < 9; it++) {
                CK(hipEventRecord(e0));
                if (nt == 0) k_stream_nt<0><<<blocks, 256>>>(p, n2);
                if (nt == 1) k_stream_nt<1><<<blocks, 256>>>(p, n2);
                if (nt == 2) k_stream_nt<2><<<blocks, 256>>>(p, n2);
                if (nt == 3) k_stream_nt<3><<<blocks, 256>>>(p, n2);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                if (it >= 2) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("full-grid copy, nt loads %d nt stores %d: median %.3f ms  %.2f TB/s\n", nt & 1, (nt >> 1) & 1, ms[ms.size() / 2],
                   8.0 * n2 * 16 / ms[ms.size() / 2] / 1e9);
        }
    }
    int blocks_list[] = {256 * 8, 256 * 16, 256 * 64, (int)((n2 + 255) / 256)};
    for (int b : blocks_list) if (run<0>(p, n2, b, "copy 4in/4out")) return 1;
    if (run<8>(p, n2, 256 * 16, "copy + fma")) return 1;
    if (run<25>(p, n2, 256 * 16, "copy + fma")) return 1;
    if (run<50>(p, n2, 256 * 16, "copy + fma")) return 1;
    if (run<100>(p, n2, 256 * 16, "copy + fma")) return 1;
    if (run<150>(p, n2, 256 * 16, "copy + fma")) return 1;
    if (run<200>(p, n2, 256 * 16, "copy + fma")) return 1;
    // hipMemcpyDtoD of one array pair for reference
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; it++) {
        CK(hipEventRecord(e0));
        CK(hipMemcpyAsync((void*)p.out[0], (const void*)p.in[0], n * 8, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        printf("hipMemcpy D2D 1 array: %.3f ms  %.2f TB/s (read+write)\n", t, 2.0 * n * 8 / t / 1e9);
    }
    return 0;
}
