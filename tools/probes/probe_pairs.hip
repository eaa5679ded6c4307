// Probe: which of M separately allocated arrays go well together as the 8 streams of a 4-in/4-out copy?
// Random subsets (4 in, 4 out) are timed; output is one line per trial: "time_ms in0 in1 in2 in3 out0 out1 out2 out3".
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
    const int M = argc > 1 ? atoi(argv[1]) : 24, trials = argc > 2 ? atoi(argv[2]) : 300;
    const size_t row = 16384 + 8, n = row * row, n2 = n / 2, bytes = n * 8;
    std::vector<char*> a(M);
    for (int k = 0; k < M; k++) { CK(hipMalloc(&a[k], bytes)); CK(hipMemset(a[k], 0, bytes)); }
    printf("# VA/2MiB:"); for (int k = 0; k < M; k++) printf(" %llx", (unsigned long long)a[k] >> 21); printf("\n");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    std::vector<int> idx(M);
    for (int t = 0; t < trials; t++) {
        for (int k = 0; k < M; k++) idx[k] = k;
        if (t >= M / 8)                                  // first trials: the consecutive groups of 8, as allocated
            for (int k = 0; k < 8; k++) std::swap(idx[k], idx[k + next() % (M - k)]);
        else
            for (int k = 0; k < 8; k++) idx[k] = 8 * t + k;
        ptrs p;
        for (int k = 0; k < 4; k++) { p.in[k] = (const double2*)a[idx[k]]; p.out[k] = (double2*)a[idx[4 + k]]; }
        float best = 1e9;
        for (int it = 0; it < 4; it++) {
            CK(hipEventRecord(e0));
            k_copy<<<(unsigned)((n2 + 255) / 256), 256>>>(p, n2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) best = std::min(best, ms);
        }
        printf("%.3f", best); for (int k = 0; k < 8; k++) printf(" %d", idx[k]); printf("\n");
    }
    return 0;
}
