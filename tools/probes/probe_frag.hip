// Probe: is the placement effect a TRANSLATION effect — do the slow allocations differ from the fast ones in
// how many page-table walks they need? M vectors are allocated one by one; per vector, with the TLBs flushed by a sweep
// over another large buffer, one 8-byte load per 2 MiB (and per 64 KiB) is timed; then the 4-in/4-out copy of random
// 8-subsets. Output: per-vector touch times, then "copy_ms  sum_touch_us  idx..." per draw.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_touch(const char* __restrict__ p, size_t stride, size_t count, double* sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const double v = *(const double*)(p + i * stride);
    if (v == 1.2345e300) *sink = v;
}
struct ptrs { const double2* in[4]; double2* out[4]; };
__global__ __launch_bounds__(256) void k_copy4(ptrs p, size_t n2)
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
    const int M = argc > 1 ? atoi(argv[1]) : 24, draws = argc > 2 ? atoi(argv[2]) : 120;
    const size_t MiB = 1ull << 20, row = 16384 + 8, n = row * row, n2 = n / 2, bytes = n * 8;
    std::vector<char*> a(M);
    for (int k = 0; k < M; k++) { CK(hipMalloc(&a[k], bytes)); CK(hipMemset(a[k], 0, bytes)); }
    const size_t thrash_bytes = 24ull << 30;
    char* thrash; CK(hipMalloc(&thrash, thrash_bytes)); CK(hipMemset(thrash, 0, thrash_bytes));
    double* sink; CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto touch_us = [&](const char* p, size_t stride) {
        const size_t count = bytes / stride;
        std::vector<float> t;
        for (int rep = 0; rep < 9; rep++) {
            const size_t tc = thrash_bytes / (2 * MiB);
            k_touch<<<(unsigned)((tc + 255) / 256), 256>>>(thrash, 2 * MiB, tc, sink);      // evict the translations
            (void)hipEventRecord(e0);
            k_touch<<<(unsigned)((count + 255) / 256), 256>>>(p, stride, count, sink);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        return t[t.size() / 2];
    };
    std::vector<float> q2m(M), q64k(M);
    printf("# vector  VA/2MiB   touch per 2 MiB (us)   touch per 64 KiB (us)\n");
    for (int k = 0; k < M; k++) {
        q2m[k] = touch_us(a[k], 2 * MiB);
        q64k[k] = touch_us(a[k], 64 * 1024);
        printf("V %2d  %llx  %8.1f  %8.1f\n", k, (unsigned long long)a[k] >> 21, q2m[k], q64k[k]);
    }
    fflush(stdout);
    unsigned long long rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    std::vector<int> idx(M);
    for (int t = 0; t < draws; t++) {
        for (int k = 0; k < M; k++) idx[k] = k;
        for (int k = 0; k < 8; k++) std::swap(idx[k], idx[k + next() % (M - k)]);
        ptrs p;
        for (int k = 0; k < 4; k++) { p.in[k] = (const double2*)a[idx[k]]; p.out[k] = (double2*)a[idx[4 + k]]; }
        float best = 1e9f;
        for (int it = 0; it < 4; it++) {
            (void)hipEventRecord(e0);
            k_copy4<<<(unsigned)((n2 + 255) / 256), 256>>>(p, n2);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (it > 0) best = std::min(best, ms);
        }
        float s2 = 0, s64 = 0;
        for (int k = 0; k < 8; k++) { s2 += q2m[idx[k]]; s64 += q64k[idx[k]]; }
        printf("D %.3f  %8.1f %8.1f ", best, s2, s64);
        for (int k = 0; k < 8; k++) printf(" %d", idx[k]);
        printf("\n");
    }
    return 0;
}
