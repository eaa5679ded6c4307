// Probe: the 4-in/4-out copy (the sweeps' traffic shape) with its 8 streams at a UNIFORM stride T inside one slab, as a
// function of T (16-MiB steps from 1 to 6 GiB, then 1-MiB steps around 2 GiB), at two bases 5 GiB apart — which strides
// are fast whatever the base? 1 GiB - 32 MiB per stream. usage: probe_stride [mode: 0 = both scans]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
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
int main()
{
    const size_t GiB = 1ull << 30, MiB = 1ull << 20;
    const size_t n = 1 * GiB - 32 * MiB, n2 = n / 16, slab_bytes = 56 * GiB;
    char* slab; CK(hipMalloc(&slab, slab_bytes));
    CK(hipMemset(slab, 0, slab_bytes));
    char* A0 = (char*)(((size_t)slab + GiB - 1) & ~(GiB - 1));
    printf("# slab VA %p, A %p\n", (void*)slab, (void*)A0);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned blocks = (unsigned)((n2 + 255) / 256);
    auto measure = [&](char* A, size_t T, bool interleave) {
        if ((size_t)(A - slab) + 7 * T + n > slab_bytes) return -1.f;
        ptrs q;
        // roles in allocation order (in0..3, out0..3), or reads on even / writes on odd positions
        for (int k = 0; k < 4; k++) {
            q.in[k] = (const double2*)(A + (size_t)(interleave ? 2 * k : k) * T);
            q.out[k] = (double2*)(A + (size_t)(interleave ? 2 * k + 1 : 4 + k) * T);
        }
        float best = 1e9f;
        for (int it = 0; it < 4; it++) {
            (void)hipEventRecord(e0); k_copy4<<<blocks, 256>>>(q, n2); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float t; (void)hipEventElapsedTime(&t, e0, e1);
            if (it > 0) best = std::min(best, t);
        }
        return best;
    };
    const double gb = 8.0 * n / 1e9;
    auto line = [&](const char* tag, size_t T) {
        const float a = measure(A0, T, false), b = measure(A0 + 5 * GiB, T, false), c = measure(A0 + 1 * GiB, T, false), d = measure(A0, T, true);
        if (a < 0) return;
        printf("%s T = %9.3f MiB  base A: %5.2f   A+5GiB: %5.2f   A+1GiB: %5.2f   interleaved roles at A: %5.2f TB/s\n", tag, (double)T / MiB,
               gb / a, b > 0 ? gb / b : 0., c > 0 ? gb / c : 0., gb / d);
        fflush(stdout);
    };
    for (size_t T = 1 * GiB; T <= 6 * GiB; T += 16 * MiB) line("S16", T);
    for (size_t T = 2 * GiB - 32 * MiB; T <= 2 * GiB + 96 * MiB; T += MiB) line("S1 ", T);
    for (size_t T = 2 * GiB; T <= 2 * GiB + 4 * MiB; T += 64 * 1024) line("S64K", T);
    return 0;
}
