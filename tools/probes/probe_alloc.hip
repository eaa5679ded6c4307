// Probe: streaming rate of the 4-in/4-out copy as a function of HOW the 8 arrays were allocated
// (fresh process per mode):  A  8 x hipMalloc(size)      B  8 x hipMalloc(4 GiB)     C  one slab, stride ≡ 4 MiB mod 16
//                            D  slab allocated and freed first, then 8 x hipMalloc(size)
//                            E  8 x hipMalloc(size + k * 2 MiB padding), argv[2] = k
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct ptrs { const double2* in[4]; double2* out[4]; };
__global__ __launch_bounds__(256) void k_copy(ptrs p, size_t n2)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    double2 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = p.in[k][i];
#pragma unroll
    for (int k = 0; k < 4; k++) __builtin_nontemporal_store(v[k].x, &p.out[k][i].x), __builtin_nontemporal_store(v[k].y, &p.out[k][i].y);
}
int main(int argc, char** argv)
{
    const char mode = argc > 1 ? argv[1][0] : 'A';
    const size_t pad = argc > 2 ? (size_t)atol(argv[2]) : 0;
    const size_t row = 16384 + 8, n = row * row, n2 = n / 2, bytes = n * 8, MiB = 1u << 20;
    char* a[8];
    if (mode == 'C') {
        const size_t S = 2052 * MiB;
        char* slab; CK(hipMalloc(&slab, 8 * S));
        for (int k = 0; k < 8; k++) a[k] = slab + k * S;
    } else {
        if (mode == 'D') { char* slab; CK(hipMalloc(&slab, 8 * 2060 * MiB)); CK(hipMemset(slab, 0, 8 * 2060 * MiB)); CK(hipDeviceSynchronize()); CK(hipFree(slab)); }
        // F: irregular paddings (triangular numbers x pad x 2 MiB), G: pseudo-random paddings below pad x 64 MiB
        static const size_t tri[8] = {0, 1, 3, 6, 10, 15, 21, 28};
        unsigned long long rng = 0x9E3779B97F4A7C15ull;
        for (int k = 0; k < 8; k++) {
            size_t extra = pad * 2 * MiB;
            if (mode == 'F') extra = tri[k] * pad * 2 * MiB;
            if (mode == 'G') { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; extra = (rng % (pad * 32)) * 2 * MiB; }
            CK(hipMalloc(&a[k], mode == 'B' ? 4096 * MiB : bytes + extra));
        }
    }
    for (int k = 0; k < 8; k++) CK(hipMemset(a[k], 0, bytes));
    ptrs p;
    for (int k = 0; k < 4; k++) { p.in[k] = (const double2*)a[k]; p.out[k] = (double2*)a[4 + k]; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int it = 0; it < 13; it++) {
        CK(hipEventRecord(e0));
        k_copy<<<(unsigned)((n2 + 255) / 256), 256>>>(p, n2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (it >= 3) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("mode %c pad %zu: median %.3f ms  %.2f TB/s   VA gaps (MiB):", mode, pad, ms[ms.size() / 2], 8.0 * n2 * 16 / ms[ms.size() / 2] / 1e9);
    for (int k = 1; k < 8; k++) printf(" %.1f", (double)(a[k] - a[k - 1]) / MiB);
    printf("\n");
    return 0;
}
