// Probe (round 4): does an INTERNAL state layout with fewer streams remove the placement lottery of the fused sweeps?
// The sweeps' access patterns without arithmetic (the X strips and the Y march of csrc/fused_sweep_impl.hpp, as in
// probe_access.hip) copy the same 2 x 17.2 GB of a 16384² fp64 block (4 variables read, 4 written, X then Y) through:
//   flat    8 separately allocated vectors, pitch nx + 2g                      (today: 8 streams)
//   rows    one slab per state set, ROW-INTERLEAVED: row j of rho,u,v,E adjacent,
//           addr = slab + ((4 j + var) * pitch + x) * 8, pitch padded          (2 streams; the kernels only need a pitch)
//   rows8   ONE slab for both sets, 8 rows per row index (in 0..3, out 4..7)   (1 stream)
//   rec     one slab per set, AoSoA records of R cells: [rho x R][u x R][v x R][E x R] along a row
// Every layout is allocated SEVERAL times in the same process (all live together, nothing is reused), so one process
// shows the spread between allocations and ten processes (tools/r04/layout_runs.sh) the spread between processes.
// Output: one line per (layout, instance): X ms, Y ms, X+Y ms, TB/s of the pair.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int LAG = 4;
typedef double v2 __attribute__((ext_vector_type(2)));

// Where cell (row, col) of variable k lives: base[k] + row * rpitch + f(col); REC = 0: f(col) = col,
// REC = R: records of R cells, f(col) = (col / R) * 4R + col % R (the variable's offset k * R is folded into base[k]).
struct lay { double* in[4]; double* out[4]; long rpitch; };

template <int REC>
__device__ __forceinline__ long colf(int col)
{
    if (REC == 0) return col;
    return (long)(col / REC) * (4 * REC) + (col % REC);
}

// Y march: lane <-> column, runs of `seg` rows (+2 LAG halo rows), PF rows in flight, nt stores
template <int PF, int BLOCK, int REC, int NV = 4>
__global__ __launch_bounds__(BLOCK) void ypat(lay p, int nx, int ny, int g, int seg, int shift)
{
    const int xr = (int)(blockIdx.x * BLOCK + threadIdx.x) - shift;
    const bool active = xr >= 0 && xr < nx;
    const int x = active ? xr : 0;
    const int o0 = blockIdx.y * seg, o1 = min(o0 + seg, ny);
    const int jb = o0 - LAG, je = o1 + LAG;
    double ring[8][4];
    const long col = colf<REC>(x + g);
    auto load = [&](int slot, int j) {
        const long off = (long)(j + g) * p.rpitch + col;
#pragma unroll
        for (int k = 0; k < NV; k++) ring[slot][k] = p.in[k][off];
    };
#pragma unroll
    for (int k = 0; k < PF; k++) load(k, min(jb + k, je - 1));
    for (int t = 0; t < je - jb + 8; t += 8) {
#pragma unroll
        for (int ph = 0; ph < 8; ph++) {
            const int j = jb + t + ph;
            const int o = j - LAG;
            if (o >= o0 && o < o1 && active) {
                const long off = (long)(o + g) * p.rpitch + col;
#pragma unroll
                for (int k = 0; k < NV; k++) __builtin_nontemporal_store(ring[(ph + 4) & 7][k], p.out[k] + off);
            }
            load((ph + PF) & 7, min(j + PF, je - 1));
        }
    }
}

// X strips: one wave per row and strip (one strip per wave), 4 rows per workgroup; strip s reads cells
// [a0 + 120 s - 4, + 128) of its row with 16 B per lane and stores the inner 120
template <int REC, int NV = 4>
__global__ __launch_bounds__(256) void xpat(lay p, int nx, int ny, int g, int a0)
{
    constexpr int H = 4, STRIDE = 128 - 2 * H;
    const int lane = threadIdx.x, row = blockIdx.y * 4 + threadIdx.y;
    if (row >= ny) return;
    const long rb = (long)(row + g) * p.rpitch;
    const int w0 = a0 + blockIdx.x * STRIDE;
    if (w0 >= nx) return;
    int j0 = w0 - H + lane * 2;
    const int jl = max(-g, min(j0, nx + g - 2));
    const long off = rb + colf<REC>(jl + g);
    v2 buf[4];
#pragma unroll
    for (int k = 0; k < NV; k++) buf[k] = *reinterpret_cast<const v2*>(p.in[k] + off);
    const int hi = min(w0 + STRIDE, nx);
    if (j0 >= max(w0, 0) && j0 + 1 < hi) {
#pragma unroll
        for (int k = 0; k < NV; k++) __builtin_nontemporal_store(buf[k], reinterpret_cast<v2*>(p.out[k] + off));
    }
}

struct timer {
    hipEvent_t e0, e1;
    timer() { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
    template <class F> double med(F&& launch, int reps = 7)
    {
        std::vector<float> ms;
        for (int it = 0; it < reps + 2; it++) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (it >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        return ms[ms.size() / 2];
    }
};

static int nx = 16384, ny = 16384;
constexpr int g = 4;
static int seg = 529;

// plain linear copy of n doubles, 16 B per lane (the 1-in/1-out reference)
__global__ __launch_bounds__(256) void lincopy(const v2* __restrict__ in, v2* __restrict__ out, size_t n2)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) __builtin_nontemporal_store(in[i], out + i);
}

template <int REC, int NV = 4>
static void run_pair(timer& T, const char* tag, lay A2B, lay B2A)
{
    const int shift = 4;
    dim3 gy((nx + shift + 255) / 256, (ny + seg - 1) / seg);
    const int a0 = -4;
    dim3 gx((nx - a0 + 119) / 120, (ny + 3) / 4);
    const double x = T.med([&] { xpat<REC, NV><<<gx, dim3(64, 4)>>>(A2B, nx, ny, g, a0); });
    const double y = T.med([&] { ypat<4, 256, REC, NV><<<gy, 256>>>(B2A, nx, ny, g, seg, shift); });
    // the pair, back to back, as in a cycle
    const double xy = T.med([&] {
        xpat<REC, NV><<<gx, dim3(64, 4)>>>(A2B, nx, ny, g, a0);
        ypat<4, 256, REC, NV><<<gy, 256>>>(B2A, nx, ny, g, seg, shift);
    });
    const double bytes = 2 * 16.0 * NV * nx * ny;
    printf("%-34s X %.3f  Y %.3f  X+Y %.3f ms  %.2f TB/s\n", tag, x, y, xy, bytes / xy / 1e9);
    fflush(stdout);
}

static double* dmalloc(size_t n)
{
    double* p;
    CK(hipMalloc(&p, n * 8));
    CK(hipMemset(p, 0, n * 8));
    return p;
}

// physically contiguous allocation (hipDeviceMallocContiguous); NULL when the runtime refuses the flag or the size
static double* dmalloc_contig(size_t n)
{
    double* p = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&p, n * 8, hipDeviceMallocContiguous);
    if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    CK(hipMemset(p, 0, n * 8));
    return p;
}

int main(int argc, char** argv)
{
    int inst = 3;
    std::string only;
    for (int i = 1; i < argc; i++) {
        if (!strncmp(argv[i], "--n=", 4)) nx = ny = atoi(argv[i] + 4);
        else if (!strncmp(argv[i], "--inst=", 7)) inst = atoi(argv[i] + 7);
        else if (!strncmp(argv[i], "--seg=", 6)) seg = atoi(argv[i] + 6);
        else if (!strncmp(argv[i], "--only=", 7)) only = argv[i] + 7;
    }
    auto want = [&](const char* name) { return only.empty() || only.find(name) != std::string::npos; };
    timer T;
    const long rows = ny + 2 * g;
    char tag[96];

    if (want("flat")) {
        // today's layout; more vectors than roles so that several disjoint sets of 8 exist in one process
        const long pitch = nx + 2 * g;
        std::vector<double*> v;
        for (int k = 0; k < 8 * inst; k++) v.push_back(dmalloc((size_t)pitch * rows + 512));
        for (int i = 0; i < inst; i++) {
            lay A2B, B2A;
            for (int k = 0; k < 4; k++) {
                A2B.in[k] = v[8 * i + k]; A2B.out[k] = v[8 * i + 4 + k];
                B2A.in[k] = v[8 * i + 4 + k]; B2A.out[k] = v[8 * i + k];
            }
            A2B.rpitch = B2A.rpitch = pitch;
            snprintf(tag, sizeof tag, "flat pitch %ld #%d (in order)", pitch, i);
            run_pair<0>(T, tag, A2B, B2A);
            // reads on the even, writes on the odd allocations (round 2's 'fast' assignment)
            for (int k = 0; k < 4; k++) {
                A2B.in[k] = v[8 * i + 2 * k]; A2B.out[k] = v[8 * i + 2 * k + 1];
                B2A.in[k] = v[8 * i + 2 * k + 1]; B2A.out[k] = v[8 * i + 2 * k];
            }
            snprintf(tag, sizeof tag, "flat pitch %ld #%d (even/odd)", pitch, i);
            run_pair<0>(T, tag, A2B, B2A);
        }
        for (double* p : v) CK(hipFree(p));
    }

    if (only.find("flatc") != std::string::npos) {
        // the 8 flat vectors, each PHYSICALLY CONTIGUOUS: is the level then the same for every instance and process?
        const long pitch = nx + 2 * g;
        std::vector<double*> v;
        for (int k = 0; k < 8 * inst; k++) {
            double* p = dmalloc_contig((size_t)pitch * rows + 512);
            if (!p) { printf("flatc: hipExtMallocWithFlags(hipDeviceMallocContiguous) refused vector %d\n", k); break; }
            v.push_back(p);
        }
        for (int i = 0; i + 1 <= (int)v.size() / 8; i++) {
            lay A2B, B2A;
            for (int k = 0; k < 4; k++) {
                A2B.in[k] = v[8 * i + k]; A2B.out[k] = v[8 * i + 4 + k];
                B2A.in[k] = v[8 * i + 4 + k]; B2A.out[k] = v[8 * i + k];
            }
            A2B.rpitch = B2A.rpitch = pitch;
            snprintf(tag, sizeof tag, "flatc #%d (in order) base %%16MiB=%zuK", i, ((size_t)v[8 * i] % (16u << 20)) >> 10);
            run_pair<0>(T, tag, A2B, B2A);
            for (int k = 0; k < 4; k++) {
                A2B.in[k] = v[8 * i + 2 * k]; A2B.out[k] = v[8 * i + 2 * k + 1];
                B2A.in[k] = v[8 * i + 2 * k + 1]; B2A.out[k] = v[8 * i + 2 * k];
            }
            snprintf(tag, sizeof tag, "flatc #%d (even/odd)", i);
            run_pair<0>(T, tag, A2B, B2A);
        }
        for (double* p : v) CK(hipFree(p));
        // ONE contiguous slab holding the 8 vectors back to back at a padded stride
        for (long padMiB : {0L, 2L, 6L, 10L}) {
            const size_t stride = (size_t)pitch * rows + (size_t)padMiB * 131072;
            double* S = dmalloc_contig(8 * stride + 512);
            if (!S) { printf("flatc: contiguous slab of %.1f GB refused\n", 8.0 * stride * 8 / 1e9); break; }
            lay A2B, B2A;
            for (int k = 0; k < 4; k++) {
                A2B.in[k] = S + (2 * k) * stride; A2B.out[k] = S + (2 * k + 1) * stride;
                B2A.in[k] = A2B.out[k]; B2A.out[k] = A2B.in[k];
            }
            A2B.rpitch = B2A.rpitch = pitch;
            snprintf(tag, sizeof tag, "slabc pad %ld MiB (even/odd)", padMiB);
            run_pair<0>(T, tag, A2B, B2A);
            for (int k = 0; k < 4; k++) {
                A2B.in[k] = S + k * stride; A2B.out[k] = S + (4 + k) * stride;
                B2A.in[k] = A2B.out[k]; B2A.out[k] = A2B.in[k];
            }
            snprintf(tag, sizeof tag, "slabc pad %ld MiB (in order)", padMiB);
            run_pair<0>(T, tag, A2B, B2A);
            CK(hipFree(S));
        }
    }

    if (want("rows")) {
        const long pads[] = {0, 8, 24, 56, 120, 248};       // pitch = nx + 2g + pad (16392: not a multiple of 128 B; 16400 is)
        for (long pad : pads) {
            const long pitch = nx + 2 * g + pad;
            std::vector<double*> keep;
            for (int i = 0; i < inst; i++) {
                double* A = dmalloc((size_t)4 * pitch * rows + 512);
                double* B = dmalloc((size_t)4 * pitch * rows + 512);
                keep.push_back(A); keep.push_back(B);
                lay A2B, B2A;
                for (int k = 0; k < 4; k++) {
                    A2B.in[k] = A + k * pitch; A2B.out[k] = B + k * pitch;
                    B2A.in[k] = B + k * pitch; B2A.out[k] = A + k * pitch;
                }
                A2B.rpitch = B2A.rpitch = 4 * pitch;
                snprintf(tag, sizeof tag, "rows pitch %ld #%d", pitch, i);
                run_pair<0>(T, tag, A2B, B2A);
            }
            for (double* p : keep) CK(hipFree(p));
        }
    }

    if (want("rows8")) {
        const long pads[] = {8, 24};
        for (long pad : pads) {
            const long pitch = nx + 2 * g + pad;
            std::vector<double*> keep;
            for (int i = 0; i < inst; i++) {
                double* S = dmalloc((size_t)8 * pitch * rows + 512);
                keep.push_back(S);
                lay A2B, B2A;
                for (int k = 0; k < 4; k++) {
                    A2B.in[k] = S + k * pitch; A2B.out[k] = S + (4 + k) * pitch;
                    B2A.in[k] = S + (4 + k) * pitch; B2A.out[k] = S + k * pitch;
                }
                A2B.rpitch = B2A.rpitch = 8 * pitch;
                snprintf(tag, sizeof tag, "rows8 pitch %ld #%d", pitch, i);
                run_pair<0>(T, tag, A2B, B2A);
            }
            for (double* p : keep) CK(hipFree(p));
        }
    }

    if (want("rec")) {
        // records of 64 / 128 cells along the ghosted row (row padded to whole records)
        for (int R : {64, 128}) {
            const long cells = ((nx + 2 * g + R - 1) / R) * R;
            const long rp = 4 * cells;
            std::vector<double*> keep;
            for (int i = 0; i < inst; i++) {
                double* A = dmalloc((size_t)rp * rows + 512);
                double* B = dmalloc((size_t)rp * rows + 512);
                keep.push_back(A); keep.push_back(B);
                lay A2B, B2A;
                for (int k = 0; k < 4; k++) {
                    A2B.in[k] = A + k * R; A2B.out[k] = B + k * R;
                    B2A.in[k] = B + k * R; B2A.out[k] = A + k * R;
                }
                A2B.rpitch = B2A.rpitch = rp;
                snprintf(tag, sizeof tag, "rec R=%d #%d", R, i);
                if (R == 64) run_pair<64>(T, tag, A2B, B2A);
                else run_pair<128>(T, tag, A2B, B2A);
            }
            for (double* p : keep) CK(hipFree(p));
        }
    }
    if (want("lin")) {
        // E1: the slab pair of the row-interleaved layout copied LINEARLY (1 stream in, 1 out)
        const long pitch = nx + 2 * g + 8;
        const size_t n = (size_t)4 * pitch * rows;
        for (int i = 0; i < inst; i++) {
            double* A = dmalloc(n + 512);
            double* B = dmalloc(n + 512);
            const double ms = T.med([&] { lincopy<<<(unsigned)((n / 2 + 255) / 256), 256>>>((const v2*)A, (v2*)B, n / 2); });
            printf("lin slab->slab #%d                  %.3f ms  %.2f TB/s\n", i, ms, 16.0 * n / ms / 1e9);
            CK(hipFree(A)); CK(hipFree(B));
        }
    }
    if (want("nv")) {
        // E2/E3: the sweeps' patterns on 1 and 2 variables of the flat layout (2 and 4 streams)
        const long pitch = nx + 2 * g;
        std::vector<double*> v;
        for (int k = 0; k < 8; k++) v.push_back(dmalloc((size_t)pitch * rows + 512));
        for (int i = 0; i < 4; i++) {
            lay A2B, B2A;
            for (int k = 0; k < 4; k++) { A2B.in[k] = v[2 * i]; A2B.out[k] = v[2 * i + 1]; B2A.in[k] = v[2 * i + 1]; B2A.out[k] = v[2 * i]; }
            A2B.rpitch = B2A.rpitch = pitch;
            snprintf(tag, sizeof tag, "flat 1 var #%d", i);
            run_pair<0, 1>(T, tag, A2B, B2A);
        }
        for (int i = 0; i < 2; i++) {
            lay A2B, B2A;
            for (int k = 0; k < 4; k++) {
                A2B.in[k] = v[4 * i + 2 * (k & 1)]; A2B.out[k] = v[4 * i + 2 * (k & 1) + 1];
                B2A.in[k] = A2B.out[k]; B2A.out[k] = A2B.in[k];
            }
            A2B.rpitch = B2A.rpitch = pitch;
            snprintf(tag, sizeof tag, "flat 2 vars #%d", i);
            run_pair<0, 2>(T, tag, A2B, B2A);
        }
        for (double* p : v) CK(hipFree(p));
    }
    if (want("blk")) {
        // E5: one slab per set, variables interleaved in groups of G rows: row j of var k at ((j / G) * 4 + k) * G + j % G
        // (only G | rows kept simple: the kernels take a row pitch, so the group structure is emulated with G = all rows
        // of a run being contiguous: here G = whole planes at a padded plane stride = the 'planes in one slab' layout)
        for (long padMiB : {0L, 2L, 5L, 8L, 11L}) {
            const long pitch = nx + 2 * g;
            const size_t plane = (size_t)pitch * rows + padMiB * 131072;
            for (int i = 0; i < inst; i++) {
                double* A = dmalloc(4 * plane + 512);
                double* B = dmalloc(4 * plane + 512);
                lay A2B, B2A;
                for (int k = 0; k < 4; k++) { A2B.in[k] = A + k * plane; A2B.out[k] = B + k * plane; B2A.in[k] = A2B.out[k]; B2A.out[k] = A2B.in[k]; }
                A2B.rpitch = B2A.rpitch = pitch;
                snprintf(tag, sizeof tag, "planes pad %ld MiB #%d", padMiB, i);
                run_pair<0>(T, tag, A2B, B2A);
                CK(hipFree(A)); CK(hipFree(B));
            }
        }
    }
    return 0;
}
