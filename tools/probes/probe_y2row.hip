// Probe: would the Y march stream faster with 16-B accesses? A lane owns ONE column (the register pipeline allows no more in
// fp64), so a 16-B access has to cover two ROWS: lanes 0-31 of a wave load columns (2k, 2k+1) of row j, lanes 32-63 the same
// columns of row j + 1, and two v_permlane32_swap per double would hand every lane both rows of its own column. Here only
// the ACCESS PATTERN is timed (no arithmetic, no swap): 4 arrays in, 4 out, (16384+8)² doubles, runs of `seg` rows with
// 2 x 4 halo rows, row pairs prefetched PF deep — against the march's present pattern (8 B per lane, one row per step).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct ptrs { const double* in[4]; double* out[4]; };
constexpr int LAG = 4;
typedef double v2 __attribute__((ext_vector_type(2)));

// present pattern: lane <-> column, one row per step, PF rows ahead
template <int PF, int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void y1(ptrs p, int nx, int ny, int g, long pitch, int seg, int shift)
{
    const int xr = blockIdx.x * BLOCK + threadIdx.x - shift;
    const bool active = xr >= 0 && xr < nx;
    const int x = active ? xr : 0;
    const int o0 = blockIdx.y * seg, o1 = min(o0 + seg, ny);
    const int jb = o0 - LAG, je = o1 + LAG;
    double ring[8][4];
    const long col = x + g;
    auto load = [&](int slot, int j) {
        const long off = (long)(j + g) * pitch + col;
#pragma unroll
        for (int k = 0; k < 4; k++) ring[slot][k] = p.in[k][off];
    };
#pragma unroll
    for (int k = 0; k < PF; k++) load(k, min(jb + k, je - 1));
    for (int t = 0; t < je - jb + 8; t += 8) {
#pragma unroll
        for (int ph = 0; ph < 8; ph++) {
            const int j = jb + t + ph, o = j - LAG;
            if (o >= o0 && o < o1 && active) {
                const long off = (long)(o + g) * pitch + col;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (NT) __builtin_nontemporal_store(ring[(ph + 4) & 7][k], p.out[k] + off);
                    else p.out[k][off] = ring[(ph + 4) & 7][k];
                }
            }
            load((ph + PF) & 7, min(j + PF, je - 1));
        }
    }
}

// two rows per step: lanes 0-31 row j, lanes 32-63 row j + 1, 16 B (two columns) per lane; a wave still covers 64 columns
template <int PF, int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void y2(ptrs p, int nx, int ny, int g, long pitch, int seg, int shift)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, k2 = lane & 31;
    const int xr = blockIdx.x * BLOCK + wave * 64 + 2 * k2 - shift;       // first of this lane's two columns
    const bool active = xr >= 0 && xr + 2 <= nx;
    const int x = active ? xr : 0;
    const int o0 = blockIdx.y * seg, o1 = min(o0 + seg, ny);
    const int jb = o0 - LAG, je = o1 + LAG;                               // even number of rows (seg even)
    v2 ring[4][4];                                                        // slots of row PAIRS
    const long col = x + g;
    auto load = [&](int slot, int j) {                                    // j = first row of the pair
        const long off = (long)(min(j + half, je - 1) + g) * pitch + col;
#pragma unroll
        for (int k = 0; k < 4; k++) ring[slot][k] = *reinterpret_cast<const v2*>(p.in[k] + off);
    };
#pragma unroll
    for (int k = 0; k < PF; k++) load(k, min(jb + 2 * k, je - 2));
    for (int t = 0; t < je - jb + 8; t += 8) {
#pragma unroll
        for (int ph = 0; ph < 4; ph++) {                                  // 4 row pairs per 8 rows
            const int j = jb + t + 2 * ph, o = j - LAG + half;            // the pair emitted now: rows j - 4, j - 3
            if (o >= o0 && o < o1 && active) {
                const long off = (long)(o + g) * pitch + col;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (NT) __builtin_nontemporal_store(ring[(ph + 2) & 3][k], reinterpret_cast<v2*>(p.out[k] + off));
                    else *reinterpret_cast<v2*>(p.out[k] + off) = ring[(ph + 2) & 3][k];
                }
            }
            load((ph + PF) & 3, min(j + 2 * PF, je - 2));
        }
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
    for (int rep = 0; rep < 2; rep++)
    for (int seg : {530, 256}) {
        dim3 grid((nx + 4 + 255) / 256, (ny + seg - 1) / seg);
        snprintf(tag, sizeof tag, "Y one row/step   8 B/lane PF=4 rows   seg=%d nt stores", seg);
        if (timeit(tag, bytes, [&] { y1<4, 256, true><<<grid, 256>>>(p, nx, ny, g, pitch, seg, 4); })) return 1;
        snprintf(tag, sizeof tag, "Y two rows/step 16 B/lane PF=2 pairs  seg=%d nt stores", seg);
        if (timeit(tag, bytes, [&] { y2<2, 256, true><<<grid, 256>>>(p, nx, ny, g, pitch, seg, 4); })) return 1;
        snprintf(tag, sizeof tag, "Y two rows/step 16 B/lane PF=3 pairs  seg=%d nt stores", seg);
        if (timeit(tag, bytes, [&] { y2<3, 256, true><<<grid, 256>>>(p, nx, ny, g, pitch, seg, 4); })) return 1;
        snprintf(tag, sizeof tag, "Y one row/step   8 B/lane PF=4 rows   seg=%d plain stores", seg);
        if (timeit(tag, bytes, [&] { y1<4, 256, false><<<grid, 256>>>(p, nx, ny, g, pitch, seg, 4); })) return 1;
        snprintf(tag, sizeof tag, "Y two rows/step 16 B/lane PF=2 pairs  seg=%d plain stores", seg);
        if (timeit(tag, bytes, [&] { y2<2, 256, false><<<grid, 256>>>(p, nx, ny, g, pitch, seg, 4); })) return 1;
    }
    return 0;
}
