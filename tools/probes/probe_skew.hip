// Probe: does the relative placement of the 8 arrays of a sweep (4 read + 4 written) matter for the streaming
// rate? One slab, 8 sub-arrays of (16384+8)^2 doubles carved at base + k * (size rounded up to 2 MiB + skew),
// 4-in/4-out full-grid copy timed for several skews; then the same with 8 separate hipMalloc calls, twice.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
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

static int timeit(const char* tag, ptrs p, size_t n2)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int it = 0; it < 8; it++) {
        CK(hipEventRecord(e0));
        k_copy<<<(unsigned)((n2 + 255) / 256), 256>>>(p, n2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (it >= 3) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-60s median %.3f ms  min %.3f  %.2f TB/s\n", tag, ms[ms.size() / 2], ms[0], 8.0 * n2 * 16 / ms[ms.size() / 2] / 1e9);
    fflush(stdout);
    return 0;
}

int main()
{
    const size_t row = 16384 + 8, n = row * row, n2 = n / 2, bytes = n * 8;
    const size_t MiB2 = 2u << 20;
    const size_t base_stride = (bytes + MiB2 - 1) / MiB2 * MiB2;
    const size_t max_skew = 34 * MiB2;
    char* slab;
    CK(hipMalloc(&slab, 8 * (base_stride + max_skew) + 40 * MiB2));
    CK(hipMemset(slab, 0, 8 * (base_stride + max_skew) + 40 * MiB2));
    char tag[96];
    // random per-array offsets (multiples of 64 KiB below 64 MiB) on top of the regular stride
    unsigned long long rng = 88172645463325252ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    for (int trial = 0; trial < 300; trial++) {
        size_t off[8];
        for (int k = 0; k < 8; k++) off[k] = trial == 0 ? 0 : (size_t)(next() % 1024) * 65536;
        ptrs p;
        const size_t S = base_stride + 32 * MiB2;      // room for the offsets
        for (int k = 0; k < 4; k++) {
            p.in[k] = (const double2*)(slab + (size_t)k * S + off[k]);
            p.out[k] = (double2*)(slab + (size_t)(4 + k) * S + off[4 + k]);
        }
        snprintf(tag, sizeof tag, "rand %zu %zu %zu %zu | %zu %zu %zu %zu", off[0] >> 16, off[1] >> 16, off[2] >> 16, off[3] >> 16,
                 off[4] >> 16, off[5] >> 16, off[6] >> 16, off[7] >> 16);
        if (timeit(tag, p, n2)) return 1;
    }
    CK(hipFree(slab));
    for (int rep = 0; rep < 3; rep++) {
        void* a[8];
        for (int k = 0; k < 8; k++) { CK(hipMalloc(&a[k], bytes)); CK(hipMemset(a[k], 0, bytes)); }
        ptrs p;
        for (int k = 0; k < 4; k++) { p.in[k] = (const double2*)a[k]; p.out[k] = (double2*)a[4 + k]; }
        snprintf(tag, sizeof tag, "8 x hipMalloc (gap a1-a0 = %td B)", (char*)a[1] - (char*)a[0]);
        if (timeit(tag, p, n2)) return 1;
        if (rep == 1) { void* junk; CK(hipMalloc(&junk, 37 * MiB2 + 4096)); }   // perturb the allocator for the last round
        for (int k = 0; k < 8; k++) CK(hipFree(a[k]));
    }
    return 0;
}
