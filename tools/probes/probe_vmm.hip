// Probe (round 4): PHYSICAL SCATTER BY CONSTRUCTION. A physically contiguous slab is deterministically on the slow placement level
// whatever the offsets of the 8 streamed vectors in it (probe_slabsearch, profiles/r04_contiguous_slab.txt); separately allocated
// vectors are fast when the driver happens to scatter them well. Here the vectors are built with the virtual-memory API
// (hipMemCreate / hipMemAddressReserve / hipMemMap): a pool of physical chunks of --chunk MiB is created once and mapped
// behind the 8 virtually contiguous vectors in a chosen order —
//   seq         vector k <- chunks [k C, (k+1) C)                       (the contiguous-like baseline)
//   interleave  vector k, chunk j <- pool[8 j + k]
//   rot         vector k <- its own chunks, rotated by k * C / 8        (same chunks as seq, different phase)
//   perm <seed> one pseudo-random permutation of the whole pool dealt out to the vectors in order
// — and the sweeps' access patterns (X strips + Y march, no arithmetic) are timed on them. Same seed, same time in every process?
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


#include <numeric>

__global__ void k_fill_idx(double* p, size_t n, double tag)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = tag + (double)i;
}
__global__ void k_count_bad(const double* a, const double* b, int nx, int ny, int gg, long pitch, unsigned long long* bad)
{
    // real cells only (the patterns copy real cells)
    for (size_t r = blockIdx.x; r < (size_t)ny; r += gridDim.x)
        for (int c = threadIdx.x; c < nx; c += blockDim.x) {
            const size_t i = (r + gg) * pitch + c + gg;
            if (a[i] != b[i]) atomicAdd(bad, 1ull);
        }
}

int main(int argc, char** argv)
{
    double chunk_mib = 2;
    int nperm = 6;
    bool do_va = false, verify = false;
    double scan_span_gib = 0;
    size_t scan_step_mib = 1024, scan_pad_mib = 0;
    unsigned long long hint = 0;
    long one_shift = -1, one_pad = 0;
    for (int i = 1; i < argc; i++) {
        if (!strncmp(argv[i], "--chunk=", 8)) chunk_mib = atof(argv[i] + 8);
        else if (!strncmp(argv[i], "--perms=", 8)) nperm = atoi(argv[i] + 8);
        else if (!strcmp(argv[i], "--va")) do_va = true;
        else if (!strcmp(argv[i], "--verify")) verify = true;
        else if (!strncmp(argv[i], "--scan=", 7)) { do_va = true; scan_span_gib = atof(argv[i] + 7); }
        else if (!strncmp(argv[i], "--scanstep=", 11)) scan_step_mib = (size_t)atol(argv[i] + 11);
        else if (!strncmp(argv[i], "--scanpad=", 10)) scan_pad_mib = (size_t)atol(argv[i] + 10);
        else if (!strncmp(argv[i], "--hint=", 7)) hint = strtoull(argv[i] + 7, nullptr, 16);
        else if (!strncmp(argv[i], "--one=", 6)) { do_va = true; one_shift = atol(argv[i] + 6); const char* c = strchr(argv[i], ':'); if (c) one_pad = atol(c + 1); }
    }
    int dev = 0;
    CK(hipSetDevice(dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t G = (size_t)(chunk_mib * (1 << 20));
    G = ((G + gran - 1) / gran) * gran;
    const long pitch = nx + 2 * g, rows = ny + 2 * g;
    const size_t vbytes = (size_t)pitch * rows * 8 + 4096;
    const size_t C = (vbytes + G - 1) / G;            // chunks per vector
    const size_t M = 8 * C;
    printf("# granularity %zu B, chunk %zu B, %zu chunks per vector, pool of %zu chunks = %.2f GB\n", gran, G, C, M, M * G / 1e9);
    std::vector<hipMemGenericAllocationHandle_t> pool(M);
    for (size_t i = 0; i < M; i++) CK(hipMemCreate(&pool[i], G, &prop, 0));
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, M * G, 0, nullptr, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    timer T;
    char tag[96];
    auto run = [&](const char* name, const std::vector<size_t>& order) {      // order[k * C + j] = pool index behind chunk j of vector k
        for (size_t i = 0; i < M; i++) CK(hipMemMap((char*)va + i * G, G, 0, pool[order[i]], 0));
        CK(hipMemSetAccess(va, M * G, &acc, 1));
        lay A2B, B2A;
        for (int k = 0; k < 4; k++) {
            double* in = (double*)((char*)va + (size_t)k * C * G);
            double* out = (double*)((char*)va + (size_t)(4 + k) * C * G);
            A2B.in[k] = in; A2B.out[k] = out; B2A.in[k] = out; B2A.out[k] = in;
        }
        A2B.rpitch = B2A.rpitch = pitch;
        run_pair<0>(T, name, A2B, B2A);
        // roles even / odd as well
        for (int k = 0; k < 4; k++) {
            double* in = (double*)((char*)va + (size_t)(2 * k) * C * G);
            double* out = (double*)((char*)va + (size_t)(2 * k + 1) * C * G);
            A2B.in[k] = in; A2B.out[k] = out; B2A.in[k] = out; B2A.out[k] = in;
        }
        snprintf(tag, sizeof tag, "%s (even/odd)", name);
        run_pair<0>(T, tag, A2B, B2A);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(va, M * G));
    };
    std::vector<size_t> order(M);
    std::iota(order.begin(), order.end(), 0);
    if (one_shift < 0) run("seq", order);
    for (size_t k = 0; k < 8; k++) for (size_t j = 0; j < C; j++) order[k * C + j] = j * 8 + k;
    if (one_shift < 0) run("interleave", order);
    for (size_t k = 0; k < 8; k++) for (size_t j = 0; j < C; j++) order[k * C + j] = k * C + (j + k * C / 8) % C;
    if (one_shift < 0) run("rot", order);
    for (int s = 1; s <= nperm; s++) {
        std::iota(order.begin(), order.end(), 0);
        unsigned long long x = 0x9E3779B97F4A7C15ull * (unsigned long long)s;
        for (size_t i = M - 1; i > 0; i--) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            std::swap(order[i], order[(size_t)(x % (i + 1))]);
        }
        snprintf(tag, sizeof tag, "perm %d", s);
        run(tag, order);
    }
    // ---- is it the VIRTUAL layout? The same pool, in the same order, mapped with the vectors `pad` apart and the whole group
    // shifted by `shift` inside one large reservation (physical memory unchanged throughout).
    if (do_va) {
        const size_t VSPAN = M * G + 8 * (scan_pad_mib << 20) + ((size_t)(scan_span_gib > 0 ? scan_span_gib + 2 : 40) << 30);
        void* big = nullptr;
        CK(hipMemAddressReserve(&big, VSPAN, 1ull << 30, (void*)hint, 0));
        printf("# VA experiment: reservation %.1f GB at %p (hint %p)\n", VSPAN / 1e9, big, (void*)hint);
        auto run_va = [&](size_t shift, size_t pad) {
            const size_t stride = C * G + pad;
            for (size_t k = 0; k < 8; k++)
                for (size_t j = 0; j < C; j++) CK(hipMemMap((char*)big + shift + k * stride + j * G, G, 0, pool[k * C + j], 0));
            for (size_t k = 0; k < 8; k++) CK(hipMemSetAccess((char*)big + shift + k * stride, C * G, &acc, 1));
            for (int order = 0; order < 2; order++) {
                lay A2B, B2A;
                for (int k = 0; k < 4; k++) {
                    const size_t i = order ? 2 * k : k, o = order ? 2 * k + 1 : 4 + k;
                    double* in = (double*)((char*)big + shift + i * stride);
                    double* out = (double*)((char*)big + shift + o * stride);
                    A2B.in[k] = in; A2B.out[k] = out; B2A.in[k] = out; B2A.out[k] = in;
                }
                A2B.rpitch = B2A.rpitch = pitch;
                snprintf(tag, sizeof tag, "va shift %6.0f MiB pad %5.0f MiB %s", shift / 1048576., pad / 1048576., order ? "e/o" : "ord");
                if (verify) {
                    // the copies really happen: index patterns into the inputs, X pattern A -> B, Y pattern B -> A2 would overwrite, so check A -> B only
                    const size_t nvec = (size_t)pitch * rows;
                    for (int k = 0; k < 4; k++) {
                        k_fill_idx<<<4096, 256>>>(A2B.in[k], nvec, 1e9 * (k + 1));
                        k_fill_idx<<<4096, 256>>>(A2B.out[k], nvec, -1.0);
                    }
                    dim3 gx((nx + 4 + 119) / 120, (ny + 3) / 4);
                    xpat<0, 4><<<gx, dim3(64, 4)>>>(A2B, nx, ny, g, -4);
                    unsigned long long* bad; CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
                    for (int k = 0; k < 4; k++) k_count_bad<<<2048, 256>>>(A2B.in[k], A2B.out[k], nx, ny, g, pitch, bad);
                    unsigned long long hb = 0; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipFree(bad));
                    printf("   verify: %llu real cells differ after the X pattern\n", hb);
                }
                run_pair<0>(T, tag, A2B, B2A);
            }
            CK(hipDeviceSynchronize());
            for (size_t k = 0; k < 8; k++) CK(hipMemUnmap((char*)big + shift + k * stride, C * G));
        };
        const size_t MiB = 1ull << 20;
        if (one_shift >= 0) {
            run_va((size_t)one_shift * MiB, (size_t)one_pad * MiB);
        } else if (scan_span_gib > 0) {
            for (size_t shift = 0; shift <= (size_t)(scan_span_gib * 1024) * MiB; shift += (size_t)scan_step_mib * MiB) run_va(shift, scan_pad_mib * MiB);
        } else {
            for (size_t pad : {0 * MiB, 2 * MiB, 4 * MiB, 6 * MiB, 8 * MiB, 16 * MiB, 32 * MiB, 64 * MiB, 256 * MiB, 1024 * MiB, 2048 * MiB}) run_va(0, pad);
            for (size_t shift : {2 * MiB, 8 * MiB, 64 * MiB, 1024 * MiB, 4096 * MiB, 8192 * MiB, 16384 * MiB}) run_va(shift, 0);
        }
        CK(hipMemAddressFree(big, VSPAN));
    }
    for (size_t i = 0; i < M; i++) CK(hipMemRelease(pool[i]));
    CK(hipMemAddressFree(va, M * G));
    return 0;
}
