// Probe: is the HBM placement effect visible on a SINGLE stream? One large slab is cut into 2-GiB chunks;
// for every chunk: read-only rate, write-only rate, copy chunk -> next chunk, and the 4-in/4-out copy over the 8 chunks
// starting there (the sweeps' traffic shape). usage: probe_regions [slab GiB = 192]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ p, size_t n2, double* sink)
{
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 1.2345e300) *sink = acc;
}
__global__ __launch_bounds__(256) void k_write(double2* __restrict__ p, size_t n2)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) { __builtin_nontemporal_store(1.0, &p[i].x); __builtin_nontemporal_store(2.0, &p[i].y); }
}
__global__ __launch_bounds__(256) void k_copy1(const double2* __restrict__ a, double2* __restrict__ b, size_t n2)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) { double2 v = a[i]; __builtin_nontemporal_store(v.x, &b[i].x); __builtin_nontemporal_store(v.y, &b[i].y); }
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
    const size_t GiB = 1ull << 30, slab_gib = argc > 1 ? (size_t)atol(argv[1]) : 192;
    const size_t chunk = 2 * GiB + (16 << 20), n2 = (2 * GiB) / 16;     // chunk stride ≡ 0 mod 16 MiB, 2 GiB used of each
    const int K = (int)(slab_gib * GiB / chunk);
    char* slab; CK(hipMalloc(&slab, (size_t)K * chunk));
    CK(hipMemset(slab, 0, (size_t)K * chunk));
    double* sink; CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned blocks = (unsigned)((n2 + 255) / 256);
    auto timeit = [&](auto launch) {
        float best = 1e9f;
        for (int it = 0; it < 4; it++) {
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float t; (void)hipEventElapsedTime(&t, e0, e1);
            if (it > 0) best = std::min(best, t);
        }
        return best;
    };
    // per-role offsets below 16 MiB (multiples of 64 KiB), so that the 8 streams of copy4 are not all in phase
    const size_t off[8] = {0, 13 << 16, 101 << 16, 47 << 16, 200 << 16, 77 << 16, 158 << 16, 31 << 16};
    printf("# chunk  GiB     read TB/s  write TB/s  copy1 TB/s  copy4(8 chunks from here) ms\n");
    for (int c = 0; c < K; c++) {
        double2* p = (double2*)(slab + (size_t)c * chunk);
        const float tr = timeit([&] { k_read<<<4096, 256>>>(p, n2, sink); });
        const float tw = timeit([&] { k_write<<<blocks, 256>>>(p, n2); });
        float tc = 0, t4 = 0;
        if (c + 1 < K) tc = timeit([&] { k_copy1<<<blocks, 256>>>(p, (double2*)(slab + (size_t)(c + 1) * chunk), n2); });
        if (c + 8 <= K) {
            ptrs q;
            const size_t n2s = n2 - (16u << 20) / 16;           // room for the offsets
            for (int k = 0; k < 4; k++) {
                q.in[k] = (const double2*)(slab + (size_t)(c + k) * chunk + off[k]);
                q.out[k] = (double2*)(slab + (size_t)(c + 4 + k) * chunk + off[4 + k]);
            }
            t4 = timeit([&] { k_copy4<<<(unsigned)((n2s + 255) / 256), 256>>>(q, n2s); });
        }
        const double gb = 2.0 * GiB / 1e9;
        printf("%5d %6.1f   %8.2f   %8.2f   %8.2f   %8.3f\n", c, c * (double)chunk / GiB, gb / tr, gb / tw, tc ? 2 * gb / tc : 0., t4);
        fflush(stdout);
    }
    return 0;
}
