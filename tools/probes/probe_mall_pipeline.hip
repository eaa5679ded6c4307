// Probe for the banded X -> Y pipeline: does a consumer that trails a producer by one band of rows
// read the intermediate state from the Infinity Cache instead of HBM, and do the intermediate's WRITES still cost HBM
// bandwidth? Traffic shape of the two sweeps, no arithmetic: pass 1 copies 4 arrays A -> B (the X sweep), pass 2 copies
// B -> C (the Y sweep), 16384² doubles per array.
//   full      : pass 1 over the whole grid, then pass 2 (today's two launches)                 34.4 GB through HBM
//   banded/1s : bands of R rows, pass1(b) then pass2(b) on ONE stream, B a full-size array
//   banded/2s : pass1 on stream 1, pass2(b) on stream 2 after the event of pass1(b) — concurrent kernels
//   ring      : B is a ring of 3 bands (its lines are rewritten while still cached)
//   direct    : A -> C in one pass (the floor if the intermediate were free): 17.2 GB
// ST: store flavour of pass 1 (0 plain, 1 nt), LD: load flavour of pass 2.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct ptrs { const double2* in[4]; double2* out[4]; };

// copies n2 double2 per array starting at in[k] + ioff / out[k] + ooff
template <int NT_LD, int NT_ST>
__global__ __launch_bounds__(256) void k_copy(ptrs p, size_t ioff, size_t ooff, size_t n2)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    double2 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double2* q = p.in[k] + ioff + i;
        if (NT_LD) { v[k].x = __builtin_nontemporal_load(&q->x); v[k].y = __builtin_nontemporal_load(&q->y); }
        else v[k] = *q;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        double2* q = p.out[k] + ooff + i;
        if (NT_ST) { __builtin_nontemporal_store(v[k].x, &q->x); __builtin_nontemporal_store(v[k].y, &q->y); }
        else *q = v[k];
    }
}

template <int NT_LD, int NT_ST>
static void launch(hipStream_t s, ptrs p, size_t ioff, size_t ooff, size_t n2)
{
    hipLaunchKernelGGL((k_copy<NT_LD, NT_ST>), dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, p, ioff, ooff, n2);
}

int main(int argc, char** argv)
{
    const size_t N = argc > 1 ? (size_t)atol(argv[1]) : 16384;
    const size_t row2 = N / 2;                       // double2 per row
    const size_t n2 = N * row2;
    double2 *A[4], *B[4], *C[4], *Rg[4];
    const size_t ring_rows = 3 * 1024;               // room for 3 bands of up to 1024 rows
    for (int k = 0; k < 4; k++) {
        CK(hipMalloc(&A[k], n2 * 16)); CK(hipMalloc(&B[k], n2 * 16)); CK(hipMalloc(&C[k], n2 * 16));
        CK(hipMalloc(&Rg[k], ring_rows * row2 * 16));
        CK(hipMemset(A[k], 0, n2 * 16)); CK(hipMemset(B[k], 0, n2 * 16)); CK(hipMemset(C[k], 0, n2 * 16));
        CK(hipMemset(Rg[k], 0, ring_rows * row2 * 16));
    }
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, join;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    std::vector<hipEvent_t> ev(2048);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ptrs ab, bc, ac, ar, rc;
    for (int k = 0; k < 4; k++) {
        ab.in[k] = A[k]; ab.out[k] = B[k]; bc.in[k] = B[k]; bc.out[k] = C[k]; ac.in[k] = A[k]; ac.out[k] = C[k];
        ar.in[k] = A[k]; ar.out[k] = Rg[k]; rc.in[k] = Rg[k]; rc.out[k] = C[k];
    }
    auto timeit = [&](const char* tag, auto&& body) {
        std::vector<float> ms;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipEventRecord(e0, s1));
            body();
            CK(hipEventRecord(e1, s1));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (rep) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-46s %7.3f ms (min %7.3f)   %5.2f TB/s of the 34.4 GB two-pass traffic\n", tag, ms[ms.size() / 2], ms[0],
               4 * 16.0 * n2 * 2 / (ms[ms.size() / 2] * 1e-3) / 1e12);
        fflush(stdout);
    };
    timeit("direct A->C (one pass, 17.2 GB)", [&] { launch<0, 1>(s1, ac, 0, 0, n2); });
    timeit("full: A->B then B->C, nt stores", [&] { launch<0, 1>(s1, ab, 0, 0, n2); launch<0, 1>(s1, bc, 0, 0, n2); });
    timeit("full: A->B plain stores, B->C nt stores", [&] { launch<0, 0>(s1, ab, 0, 0, n2); launch<0, 1>(s1, bc, 0, 0, n2); });
    for (size_t R : {32, 64, 128, 256, 512, 1024}) {
        const size_t nb = N / R, band2 = R * row2;
        char tag[128];
        snprintf(tag, sizeof tag, "banded/1s R=%zu (%zu MB/band) plain st", R, band2 * 16 * 4 >> 20);
        timeit(tag, [&] {
            for (size_t b = 0; b < nb; b++) { launch<0, 0>(s1, ab, b * band2, b * band2, band2); launch<0, 1>(s1, bc, b * band2, b * band2, band2); }
        });
        snprintf(tag, sizeof tag, "banded/1s R=%zu nt st", R);
        timeit(tag, [&] {
            for (size_t b = 0; b < nb; b++) { launch<0, 1>(s1, ab, b * band2, b * band2, band2); launch<0, 1>(s1, bc, b * band2, b * band2, band2); }
        });
        snprintf(tag, sizeof tag, "banded/2s R=%zu plain st", R);
        timeit(tag, [&] {
            for (size_t b = 0; b < nb; b++) {
                launch<0, 0>(s1, ab, b * band2, b * band2, band2);
                CK(hipEventRecord(ev[b], s1));
                CK(hipStreamWaitEvent(s2, ev[b], 0));
                launch<0, 1>(s2, bc, b * band2, b * band2, band2);
            }
            CK(hipEventRecord(join, s2));
            CK(hipStreamWaitEvent(s1, join, 0));
        });
        snprintf(tag, sizeof tag, "ring/2s R=%zu (3 bands) plain st", R);
        timeit(tag, [&] {
            for (size_t b = 0; b < nb; b++) {
                if (b >= 3) CK(hipStreamWaitEvent(s1, ev[1024 + b - 3], 0));      // slot consumed
                launch<0, 0>(s1, ar, b * band2, (b % 3) * band2, band2);
                CK(hipEventRecord(ev[b], s1));
                CK(hipStreamWaitEvent(s2, ev[b], 0));
                launch<0, 1>(s2, rc, (b % 3) * band2, b * band2, band2);
                CK(hipEventRecord(ev[1024 + b], s2));
            }
            CK(hipEventRecord(join, s2));
            CK(hipStreamWaitEvent(s1, join, 0));
        });
        snprintf(tag, sizeof tag, "ring/1s R=%zu (1 band) plain st", R);
        timeit(tag, [&] {
            for (size_t b = 0; b < nb; b++) { launch<0, 0>(s1, ar, b * band2, 0, band2); launch<0, 1>(s1, rc, 0, b * band2, band2); }
        });
    }
    return 0;
}
