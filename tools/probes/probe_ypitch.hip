// Probe: the Y march's access pattern (no arithmetic) when the rows of the arrays do NOT start on 64-B sectors.
// 16384 x 16384 doubles, 4 arrays read + 4 written, lane <-> column, 256 columns per workgroup, runs of `seg` rows,
// 4 rows prefetched. The pitch of the read arrays and of the written arrays are set separately (which side pays?), and
// two repairs are timed: LX — every row is LOADED in sector-aligned windows (thread t takes column (t - r) mod 256 of
// the workgroup, r = the row's phase in cells) and handed to its owner through LDS; SX — the same for the STORES.
// One barrier per row serves both (double-buffered LDS). SX=2: after the hand-over wave w stores the workgroup's whole row
// of array w (2 KB contiguous, 16 B per lane) instead of 512 B of each of the four arrays.
//   hipcc -O3 --offload-arch=gfx950 -o probe_ypitch probe_ypitch.hip && ./probe_ypitch
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct ptrs { const double* in[4]; double* out[4]; };
constexpr int LAG = 4, PF = 4, BLOCK = 256;

template <bool LX, int SXM>
__global__ __launch_bounds__(BLOCK) void ypat(ptrs p, int nx, int ny, int g, long pitch_in, long pitch_out, int seg, int shift)
{
    constexpr bool SX = SXM != 0;
    __shared__ __attribute__((aligned(16))) double lds[2][2][4][BLOCK];                    // [buffer][load / store side][array][column of the workgroup]
    const int t = threadIdx.x;
    const int c0 = blockIdx.x * BLOCK - shift;               // first column of the workgroup (may be < 0: ghost side)
    const int o0 = blockIdx.y * seg, o1 = min(o0 + seg, ny);
    const int jb = o0 - LAG, je = o1 + LAG;
    double ring[8][4];
    auto col_ok = [&](int c) { return c >= -g && c < nx + g; };     // inside the ghosted row
    auto load = [&](int slot, int j) {
        const long rb = (long)(j + g) * pitch_in + g + c0;
        int ci = t;
        if (LX) ci = (t - (int)(rb & 7)) & (BLOCK - 1);
#pragma unroll
        for (int k = 0; k < 4; k++) ring[slot][k] = col_ok(c0 + ci) ? p.in[k][rb + ci] : 0.;
    };
#pragma unroll
    for (int k = 0; k < PF; k++) load(k, min(jb + k, je - 1));
    int buf = 0;
    for (int tt = 0; tt < je - jb + 8; tt += 8) {
#pragma unroll
        for (int ph = 0; ph < 8; ph++) {
            const int j = jb + tt + ph;                       // row whose loaded values are consumed now
            const int o = j - LAG;                            // row produced now
            const int jc = min(j, je - 1);
            double cur[4];
#pragma unroll
            for (int k = 0; k < 4; k++) cur[k] = ring[ph & 7][k];
            const long rbi = (long)(jc + g) * pitch_in + g + c0;
            const long rbo = (long)(o + g) * pitch_out + g + c0;
            const bool produce = o >= o0 && o < o1;           // uniform
            if (LX || SX) {
                if (LX) {
                    const int ci = (t - (int)(rbi & 7)) & (BLOCK - 1);
#pragma unroll
                    for (int k = 0; k < 4; k++) lds[buf][0][k][ci] = cur[k];
                }
                if (SX && produce) {
#pragma unroll
                    for (int k = 0; k < 4; k++) lds[buf][1][k][t] = ring[(ph + 4) & 7][k];
                }
                __syncthreads();
                if (LX) {
#pragma unroll
                    for (int k = 0; k < 4; k++) ring[ph & 7][k] = lds[buf][0][k][t];     // the owner's own column
                }
            }
            if (produce && SXM == 2) {
                // wave w stores the whole workgroup row of array w: 2 KB contiguous, 16 B per lane and instruction
                // (sector-aligned pitches only: the row starts on a sector, pairs never straddle the row's ends here)
                typedef double v2 __attribute__((ext_vector_type(2)));
                const int w = t >> 6, lane = t & 63;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int ci = h * 128 + lane * 2;
                    const int c = c0 + ci;
                    if (c >= 0 && c + 1 < nx) {
                        const v2 v = *reinterpret_cast<const v2*>(&lds[buf][1][w][ci]);
                        __builtin_nontemporal_store(v, reinterpret_cast<v2*>(p.out[w] + rbo + ci));
                    } else {
                        if (c >= 0 && c < nx) __builtin_nontemporal_store(lds[buf][1][w][ci], p.out[w] + rbo + ci);
                        if (c + 1 >= 0 && c + 1 < nx) __builtin_nontemporal_store(lds[buf][1][w][ci + 1], p.out[w] + rbo + ci + 1);
                    }
                }
            } else if (produce) {
                int ci = t;
                if (SX) ci = (t - (int)(rbo & 7)) & (BLOCK - 1);
                const int c = c0 + ci;
                if (c >= 0 && c < nx) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const double v = SX ? lds[buf][1][k][ci] : ring[(ph + 4) & 7][k];
                        __builtin_nontemporal_store(v, p.out[k] + rbo + ci);
                    }
                }
            }
            buf ^= 1;
            // the slot of row j - 4 is free after this step: row j + PF goes there ((ph + 4) & 7 == (ph + PF + 8) & 7 for PF = 4)
            load((ph + PF) & 7, min(j + PF, je - 1));
        }
    }
}

template <class F>
int timeit(const char* tag, double bytes, F launch)
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
    printf("%-72s median %.3f ms  min %.3f ms  %.2f TB/s\n", tag, ms[ms.size() / 2], ms[0], bytes / ms[ms.size() / 2] / 1e9);
    fflush(stdout);
    return 0;
}

int main()
{
    const int nx = 16384, ny = 16384, g = 4;
    const long max_pitch = nx + 2 * g + 16;
    const size_t n = (size_t)max_pitch * (ny + 2 * g);
    ptrs p;
    for (int k = 0; k < 4; k++) {
        double *a, *b;
        CK(hipMalloc(&a, n * 8 + 4096)); CK(hipMalloc(&b, n * 8 + 4096));
        CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
        p.in[k] = a; p.out[k] = b;
    }
    CK(hipDeviceSynchronize());
    const double bytes = 64.0 * nx * ny;
    char tag[160];
    const int seg = 1024, shift = 4;
    dim3 grid((nx + shift + BLOCK - 1) / BLOCK, (ny + seg - 1) / seg);
#define RUN(LX, SX, pin, pout) { \
        snprintf(tag, sizeof tag, "Y march  pitch in %ld (mod 8: %ld)  out %ld (mod 8: %ld)  LX=%d SX=%d", (long)(pin), (long)(pin) % 8, (long)(pout), (long)(pout) % 8, (int)LX, (int)SX); \
        if (timeit(tag, bytes, [&] { ypat<LX, SX><<<grid, BLOCK>>>(p, nx, ny, g, pin, pout, seg, shift); })) return 1; }
    const long P0 = nx + 2 * g;
    RUN(false, 0, P0, P0)
    RUN(false, 0, P0 + 4, P0)
    RUN(false, 0, P0, P0 + 4)
    RUN(false, 0, P0 + 4, P0 + 4)
    RUN(false, 0, P0 + 3, P0)
    RUN(false, 0, P0, P0 + 3)
    RUN(false, 0, P0 + 3, P0 + 3)
    RUN(true, 1, P0, P0)
    RUN(true, 0, P0 + 4, P0)
    RUN(false, 1, P0, P0 + 4)
    RUN(true, 1, P0 + 4, P0 + 4)
    RUN(true, 0, P0 + 3, P0)
    RUN(false, 1, P0, P0 + 3)
    RUN(true, 1, P0 + 3, P0 + 3)
    // sector-aligned pitch: plain, per-thread hand-over, one array per wave with 16-B stores (SX=2), twice each
    RUN(false, 0, P0, P0)
    RUN(false, 1, P0, P0)
    RUN(false, 2, P0, P0)
    RUN(false, 0, P0, P0)
    RUN(false, 1, P0, P0)
    RUN(false, 2, P0, P0)
    return 0;
}
