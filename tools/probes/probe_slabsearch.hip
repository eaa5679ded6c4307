// Probe (round 4): with the 8 streamed vectors carved out of ONE PHYSICALLY CONTIGUOUS slab (hipDeviceMallocContiguous) the X+Y time
// of the sweeps' access patterns is the same in every process and on every device (profiles/r04_contiguous_slab.txt): the
// placement lottery is a fixed function of the physical offsets. This tool searches that function for a fast arrangement:
//   phase A  the 8 vectors at a uniform stride, the whole group slid along the slab (base 0 .. --span GiB, step --step MiB)
//   phase B  at the best bases: random fine offsets of each vector (multiples of 1 MiB below 16 MiB), then coordinate descent
// Output: every evaluation (base, offsets, roles, X+Y ms) and the best found.
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


struct cfg { size_t base; int d[8]; int order; };     // offsets of vector k: base + k * V + d[k] MiB; order 0 = in order (0-3 in, 4-7 out), 1 = even / odd

static double* S = nullptr;
static size_t V = 0;
static size_t UNIT = 131072;      // doubles per offset unit (default 1 MiB)
static int NUNIT = 16;

static double eval(timer& T, const cfg& c)
{
    lay A2B, B2A;
    auto vec = [&](int k) { return S + (c.base + (size_t)k * V + (size_t)c.d[k] * UNIT); };
    for (int k = 0; k < 4; k++) {
        const int i = c.order ? 2 * k : k, o = c.order ? 2 * k + 1 : 4 + k;
        A2B.in[k] = vec(i); A2B.out[k] = vec(o); B2A.in[k] = vec(o); B2A.out[k] = vec(i);
    }
    A2B.rpitch = B2A.rpitch = nx + 2 * g;
    const int shift = 4, a0 = -4;
    dim3 gy((nx + shift + 255) / 256, (ny + seg - 1) / seg), gx((nx - a0 + 119) / 120, (ny + 3) / 4);
    return T.med([&] {
        xpat<0, 4><<<gx, dim3(64, 4)>>>(A2B, nx, ny, g, a0);
        ypat<4, 256, 0, 4><<<gy, 256>>>(B2A, nx, ny, g, seg, shift);
    }, 3);
}

static void show(const char* what, const cfg& c, double ms)
{
    printf("%s base %7.3f GiB  d(units) %4d %4d %4d %4d %4d %4d %4d %4d  %s  X+Y %.3f ms\n", what, c.base * 8.0 / (1 << 30), c.d[0], c.d[1], c.d[2], c.d[3],
           c.d[4], c.d[5], c.d[6], c.d[7], c.order ? "even/odd" : "in order", ms);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    double span_gib = 16, step_mib = 256;
    int randoms = 150, tops = 3;
    unsigned long long seed = 12345;
    for (int i = 1; i < argc; i++) {
        if (!strncmp(argv[i], "--span=", 7)) span_gib = atof(argv[i] + 7);
        else if (!strncmp(argv[i], "--step=", 7)) step_mib = atof(argv[i] + 7);
        else if (!strncmp(argv[i], "--random=", 9)) randoms = atoi(argv[i] + 9);
        else if (!strncmp(argv[i], "--tops=", 7)) tops = atoi(argv[i] + 7);
        else if (!strncmp(argv[i], "--seed=", 7)) seed = strtoull(argv[i] + 7, nullptr, 10);
        else if (!strncmp(argv[i], "--unit=", 7)) { UNIT = (size_t)atol(argv[i] + 7) / 8; NUNIT = (int)((16u << 20) / (UNIT * 8)); }
    }
    const long pitch = nx + 2 * g, rows = ny + 2 * g;
    const size_t n = (size_t)pitch * rows;
    V = ((n * 8 + (2u << 20) - 1) / (2u << 20)) * (2u << 20) / 8 + 16 * 131072;          // vector stride in doubles: 2-MiB multiple + 16 MiB of play
    const size_t span = (size_t)(span_gib * (1 << 30)) / 8;
    const size_t total = 8 * V + span + 131072;
    hipError_t e = hipExtMallocWithFlags((void**)&S, total * 8, hipDeviceMallocContiguous);
    if (e != hipSuccess) { printf("contiguous slab of %.1f GB refused: %s\n", total * 8 / 1e9, hipGetErrorString(e)); return 1; }
    printf("# contiguous slab %.2f GB at %p (virtual), vector stride %.3f MiB, base span %.1f GiB\n", total * 8 / 1e9, (void*)S, V * 8.0 / (1 << 20), span_gib);
    timer T;
    std::vector<std::pair<double, cfg>> all;
    // phase A
    for (size_t b = 0; b <= span; b += (size_t)(step_mib * 131072)) {
        for (int order = 0; order < 2; order++) {
            cfg c{b, {0, 0, 0, 0, 0, 0, 0, 0}, order};
            const double ms = eval(T, c);
            show("A", c, ms);
            all.push_back({ms, c});
        }
    }
    std::sort(all.begin(), all.end(), [](auto& a, auto& b) { return a.first < b.first; });
    // phase B
    auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    std::pair<double, cfg> best = all[0];
    for (int t = 0; t < tops && t < (int)all.size(); t++) {
        cfg start = all[t].second;
        std::pair<double, cfg> loc = all[t];
        for (int r = 0; r < randoms; r++) {
            cfg c = start;
            for (int k = 0; k < 8; k++) c.d[k] = (int)(rnd() % NUNIT);
            const double ms = eval(T, c);
            if (ms < loc.first) { loc = {ms, c}; show("B*", c, ms); }
        }
        // coordinate descent from the local best
        bool improved = true;
        while (improved) {
            improved = false;
            for (int k = 0; k < 8; k++) {
                for (int vv = 0; vv < 16; vv++) {
                    const int v = NUNIT <= 16 ? vv : (int)(rnd() % NUNIT);
                    if (v == loc.second.d[k]) continue;
                    cfg c = loc.second;
                    c.d[k] = v;
                    const double ms = eval(T, c);
                    if (ms < loc.first * 0.997) { loc = {ms, c}; improved = true; show("C*", c, ms); }
                }
            }
        }
        show("local best", loc.second, loc.first);
        if (loc.first < best.first) best = loc;
    }
    show("BEST", best.second, best.first);
    // repeatability of the best within the process
    for (int r = 0; r < 3; r++) show("best again", best.second, eval(T, best.second));
    return 0;
}
