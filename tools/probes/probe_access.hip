// Probe: how fast can the fused sweeps' ACCESS PATTERNS move their 64 B/cell, with no arithmetic at all?
// Arrays are the solver's: (16384+8)^2 doubles, 4 read + 4 written, real cells start 4 doubles into a row.
//   ypat<PF,VEC>: the Y march — lane <-> VEC adjacent columns, rows marched in runs of `seg` (+2*LAG halo rows),
//                 PF rows prefetched into a register ring; `shift` moves the block origin left so that a block's
//                 row segment starts on a 128-B line (shift = 4 cells) instead of 32 B into it (shift = 0).
//   xpat<W>:      the X strips — a wave reads 128 cells of a row (16 B per lane), stores the inner 128-2*H, walks
//                 `niter` strips; `a0` = cell index of the first stored cell of strip 0 (alignment of the stores).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct ptrs { const double* in[4]; double* out[4]; };
constexpr int LAG = 4;

template <int PF, int VEC, int BLOCK>
__global__ __launch_bounds__(BLOCK) void ypat(ptrs p, int nx, int ny, int g, long pitch, int seg, int shift)
{
    typedef double vt __attribute__((ext_vector_type(VEC)));
    const int xr = (blockIdx.x * BLOCK + threadIdx.x) * VEC - shift;
    const bool active = xr >= 0 && xr + VEC <= nx;
    const int x = active ? xr : 0;
    const int o0 = blockIdx.y * seg, o1 = min(o0 + seg, ny);
    const int jb = o0 - LAG, je = o1 + LAG;          // rows read (ghost rows exist: g = 4 = LAG)
    vt ring[8][4];
    const long col = x + g;
    auto load = [&](int slot, int j) {
        const long off = (long)(j + g) * pitch + col;
#pragma unroll
        for (int k = 0; k < 4; k++) ring[slot][k] = *reinterpret_cast<const vt*>(p.in[k] + off);
    };
#pragma unroll
    for (int k = 0; k < PF; k++) load(k, min(jb + k, je - 1));
    for (int t = 0; t < je - jb + 8; t += 8) {
#pragma unroll
        for (int ph = 0; ph < 8; ph++) {
            const int j = jb + t + ph;
            const int o = j - LAG;
            if (o >= o0 && o < o1 && active) {
                const long off = (long)(o + g) * pitch + col;
#pragma unroll
                for (int k = 0; k < 4; k++) *reinterpret_cast<vt*>(p.out[k] + off) = ring[(ph + 4) & 7][k];   // row j-4's slot
            }
            load((ph + PF) & 7, min(j + PF, je - 1));
        }
    }
}

// one wave per row, 4 rows per block; strip it covers cells [w0-H, w0-H+128), stores [w0, w0+128-2H)
template <int H>
__global__ __launch_bounds__(256) void xpat(ptrs p, int nx, int ny, int g, long pitch, int niter, int a0)
{
    typedef double v2 __attribute__((ext_vector_type(2)));
    constexpr int STRIDE = 128 - 2 * H;
    const int lane = threadIdx.x, row = blockIdx.y * 4 + threadIdx.y;
    if (row >= ny) return;
    const long rb = (long)(row + g) * pitch + g;
    const int wf = a0 + blockIdx.x * niter * STRIDE;
    v2 buf[2][4];
    auto load = [&](int b, int it) {
        int j0 = wf + it * STRIDE - H + lane * 2;
        j0 = max(-g, min(j0, nx + g - 2));
#pragma unroll
        for (int k = 0; k < 4; k++) buf[b][k] = *reinterpret_cast<const v2*>(p.in[k] + rb + j0);
    };
    auto exists = [&](int it) { return it < niter && wf + it * STRIDE < nx; };
    auto store = [&](int b, int it) {
        const int w0 = wf + it * STRIDE, j0 = w0 - H + lane * 2, hi = min(w0 + STRIDE, nx);
        if (j0 >= max(w0, 0) && j0 + 1 < hi) {
#pragma unroll
            for (int k = 0; k < 4; k++) *reinterpret_cast<v2*>(p.out[k] + rb + j0) = buf[b][k];
        }
    };
    if (!exists(0)) return;
    load(0, 0);
    for (int it = 0; exists(it); it += 2) {
        if (exists(it + 1)) load(1, it + 1);
        store(0, it);
        if (!exists(it + 1)) break;
        if (exists(it + 2)) load(0, it + 2);
        store(1, it + 1);
    }
}

template <class F>
static int timeit(const char* tag, double bytes, F&& launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int it = 0; it < 9; it++) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (it >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-64s median %.3f ms  min %.3f ms  %.2f TB/s\n", tag, ms[ms.size() / 2], ms[0], bytes / ms[ms.size() / 2] / 1e9);
    fflush(stdout);
    return 0;
}

int main()
{
    const int nx = 16384, ny = 16384, g = 4;
    const long pitch = nx + 2 * g;
    const size_t n = (size_t)pitch * (ny + 2 * g);
    ptrs p;
    for (int k = 0; k < 4; k++) {
        double *a, *b;
        CK(hipMalloc(&a, n * 8 + 4096)); CK(hipMalloc(&b, n * 8 + 4096));
        CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
        p.in[k] = a; p.out[k] = b;
    }
    CK(hipDeviceSynchronize());
    const double bytes = 64.0 * nx * ny;
    char tag[128];
#define YRUN(PF, VEC, BLOCK, seg, shift) { \
        snprintf(tag, sizeof tag, "Y  PF=%d %2dB/lane block=%3d cols=%4d seg=%4d shift=%d", PF, 8 * VEC, BLOCK, BLOCK * VEC, seg, shift); \
        dim3 grid((nx + shift + BLOCK * VEC - 1) / (BLOCK * VEC), (ny + seg - 1) / seg); \
        if (timeit(tag, bytes, [&] { ypat<PF, VEC, BLOCK><<<grid, BLOCK>>>(p, nx, ny, g, pitch, seg, shift); })) return 1; }
    YRUN(4, 1, 256, 128, 0)
    YRUN(4, 1, 256, 128, 4)
    YRUN(4, 1, 256, 256, 0)
    YRUN(4, 1, 256, 512, 4)
    YRUN(2, 1, 256, 128, 4)
    YRUN(4, 1, 64, 128, 0)
    YRUN(4, 1, 64, 128, 4)
    YRUN(4, 1, 128, 128, 4)
    YRUN(4, 1, 512, 128, 4)
    YRUN(4, 1, 1024, 128, 4)
    YRUN(4, 2, 256, 128, 0)
    YRUN(4, 2, 256, 128, 4)
    YRUN(4, 2, 128, 128, 4)
    YRUN(4, 2, 64, 128, 4)
    YRUN(4, 2, 64, 512, 4)
#define XRUN(H, niter, a0) { \
        snprintf(tag, sizeof tag, "X  halo=%d stride=%3d niter=%3d first-store-cell=%d", H, 128 - 2 * H, niter, a0); \
        const int per = niter * (128 - 2 * H); \
        dim3 grid((nx - (a0) + per - 1) / per, (ny + 3) / 4); \
        if (timeit(tag, bytes, [&] { xpat<H><<<grid, dim3(64, 4)>>>(p, nx, ny, g, pitch, niter, a0); })) return 1; }
    XRUN(4, 8, 0)
    XRUN(4, 8, -4)
    XRUN(4, 137, 0)
    XRUN(4, 137, -4)
    XRUN(4, 2, -4)
    XRUN(4, 32, -4)
    XRUN(8, 8, 4)
    XRUN(8, 8, -4)
    XRUN(8, 147, 4)
    XRUN(0, 8, 0)
    XRUN(0, 8, -4)
    XRUN(0, 128, -4)
    return 0;
}
