// multi_gpu.hip — the tile-decomposed grid natively: armon_hip_mgpu_init / armon_hip_halo_exchange / armon_hip_dt_allreduce.
//
// Replaces the reference's MPI halo path — start_exchange / finish_exchange (ref src/halo_exchange.jl:229-283: pack →
// wait → MPI.Start of persistent Send/Recv → MPI.Wait → unpack), block_ghost_exchange's side selection (ref
// :323-354: only the two sides ALONG the sweep axis, no corners), and the MPI_Iallreduce(MIN) of the time step
// (ref src/solver_state.jl:89-111, src/utils.jl:126-143) — with device-side ordering only:
//
//   compute stream of a tile : e_state ┐        interior sweep …                       ┌ wait e_recv ─ unpack ─ e_unpack ─ strips
//   transfer stream of a tile:         └ wait ─ pack ─ e_pack ─ copy / ncclSend+ncclRecv ┴ e_recv
// (with the edge stream the unpack and the strips stay on the transfer stream and the compute stream only joins: below)
//
// No host synchronisation anywhere between pack and unpack; the host only enqueues. Two transports behind the same
// choreography:
//   * in-process group (armon_hip_mgpu_init): every tile of the px × py grid lives in this process, on the device the
//     caller names (devices may repeat: several tiles on one GPU); a face travels as ONE peer copy
//     (hipMemcpyPeerAsync, or a plain device copy when both tiles sit on the same GPU) issued on the RECEIVER's
//     transfer stream after the sender's pack event. Over xGMI that is a direct point-to-point DMA to the one
//     neighbour that needs it.
//   * one process per GPU (armon_hip_mgpu_init_rank): RCCL send/recv pairs, grouped per sweep, on the tile's
//     transfer stream; the dt minimum is an ncclAllReduce on the compute stream (a second communicator, so the two
//     streams never share one). RCCL is dlopen'ed (the copy already in the process — e.g. torch's — is reused).
//
// Buffer layout on the wire = pack_to_array!'s (ref src/halo_exchange.jl:187-216), produced by the same kernels as
// armon_hip_pack_to_array / armon_hip_unpack_from_array.
#include "common.hpp"
#include "tile_pool.hpp"

#include <cmath>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: every entry point is resolved with dlsym

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace armon;

namespace {

struct rccl_api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

rccl_api g_rccl;

int load_rccl()
{
    if (g_rccl.lib) return ARMON_OK;
    void* h = nullptr;
    // a copy already loaded in the process first (two RCCL instances in one process would each own the GPUs' IPC state)
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (h) break;
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) {
        set_error("RCCL not found: %s", dlerror());
        return ARMON_ERR_HIP;
    }
#define RCCL_SYM(field, sym)                                                     \
    do {                                                                          \
        *(void**)(&g_rccl.field) = dlsym(h, sym);                                 \
        if (!g_rccl.field) {                                                      \
            set_error("RCCL symbol %s missing", sym);                             \
            return ARMON_ERR_HIP;                                                 \
        }                                                                         \
    } while (0)
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    RCCL_SYM(CommInitRank, "ncclCommInitRank");
    RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RCCL_SYM(GroupStart, "ncclGroupStart");
    RCCL_SYM(GroupEnd, "ncclGroupEnd");
    RCCL_SYM(Send, "ncclSend");
    RCCL_SYM(Recv, "ncclRecv");
    RCCL_SYM(AllReduce, "ncclAllReduce");
    RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
    g_rccl.lib = h;
    return ARMON_OK;
}

#define ARMON_RCCL_TRY(expr)                                                                   \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) {                                                                \
            set_error("%s: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "RCCL error"); \
            return ARMON_ERR_HIP;                                                               \
        }                                                                                       \
    } while (0)

constexpr int kSides = 4;
// Bit pattern of an unused edge scalar, neutral for the minimum in BOTH precisions: +inf as a float, 1.4e306 as (half of)
// a double — a group may run fp64 and fp32 problems one after the other.
constexpr unsigned kEdgeNeutral = 0x7F800000u;
inline int opposite(int side) { return side ^ 1; }            // Left<->Right, Bottom<->Top (ARMON_SIDE_* order)
inline int first_side(int axis) { return axis == ARMON_AXIS_X ? ARMON_SIDE_LEFT : ARMON_SIDE_BOTTOM; }

struct tile_t {
    int rank = 0, cx = 0, cy = 0, device = 0;
    armon_ctx* ctx = nullptr;
    bool owns_ctx = true;
    hipStream_t xfer = nullptr;                  // transfer stream
    int nb[kSides] = {-1, -1, -1, -1};           // neighbour rank per side, -1 = physical boundary (MPI.PROC_NULL)
    void* send[kSides] = {};
    void* recv[kSides] = {};
    size_t cap[kSides] = {};                     // bytes of send[s] / recv[s]
    hipEvent_t e_pack[kSides] = {}, e_recv[kSides] = {}, e_unpack[kSides] = {};
    bool rec_recv[kSides] = {}, rec_unpack[kSides] = {};
    size_t inflight[kSides] = {};                // bytes posted by start, 0 = nothing pending
    hipEvent_t e_state = nullptr;                // the state an exchange packs is complete on the compute stream
    hipEvent_t e_intdt = nullptr;                // cycle driver: the last sweep's own CFL step is in the tile's scalar (compute stream)
    hipEvent_t e_dtdone = nullptr;               // cycle driver: that scalar has been reduced and read back (transfer stream)
    bool rec_dtdone = false;
    hipEvent_t e_red = nullptr;                  // dt scalar ready on the compute stream
    // "edge" work — unpack + the LAG-wide strips next to the remote sides — on the TRANSFER stream, concurrent with the
    // interior sweep on the compute stream (they read the same input state and write disjoint cells)
    armon_ctx* edge = nullptr;                   // a context on xfer (own reduction scratch)
    hipEvent_t e_edge = nullptr;                 // edge work of the current sweep done
    double* edge_dt = nullptr;                   // [2] device scalars: the strips' CFL steps (+inf when unused)
    uint64_t chaos_rng = 0x9E3779B97F4A7C15ull;  // test aid, see chaos()
};

}  // namespace

struct armon_mgpu {
    int px = 1, py = 1;
    bool rccl = false;
    bool periodic[2] = {false, false};           // test aid (armon_hip_mgpu_set_periodic): out-of-grid neighbours wrap
    bool force_peer = false;                     // test aid: hipMemcpyPeerAsync even between tiles of one device
    int rank = 0, device = 0;                    // rank mode: kept between _prepare_rank and _connect
    std::vector<tile_t> tiles;                   // local tiles; in-process: all of them, index == rank
    // in-process reductions: gather on tiles[0]'s device
    double* red_buf = nullptr;                   // [n_tiles] (doubles; fp32 runs use the first half of each slot)
    hipEvent_t e_red_done = nullptr;
    // test aid (armon_hip_mgpu_set_chaos): pseudo-random busy-wait kernels in front of the group's own stream operations,
    // to shake the event ordering under timings one GPU never produces by itself
    unsigned chaos_us = 0;
    // RCCL
    ncclComm_t comm_halo = nullptr, comm_red = nullptr;
    double* red_scratch = nullptr;               // device, [16]: host-value all-reduces
    double* red_scratch_host = nullptr;          // pinned
    // armon_hip_mgpu_cycle: the exchange of the NEXT cycle's first sweep posted at the end of a cycle (-1 = none), and the
    // host threads that drive the local tiles (one per tile, created by the first cycle of a group of several tiles)
    int prefetched_axis = -1;
    struct tile_pool* pool = nullptr;
    int use_threads = -1;                        // -1: by the environment (ARMON_MGPU_THREADS, default on), 0 / 1
    // ARMON_MGPU_TIMING=1: host time of tile 0's part of the cycle steps, printed when the group is destroyed
    bool pack_on_compute = false;
    int timing = -1;
    double t_host[8] = {};
    long n_cycles = 0;
};

namespace {

__global__ void k_spin(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();               // constant-rate counter (100 MHz)
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// chaos mode: with probability 1/2, hold `stream` for up to chaos_us microseconds (one sequence per tile: tiles may be
// driven by different host threads)
void chaos(armon_mgpu* g, tile_t& t, hipStream_t stream)
{
    if (!g->chaos_us) return;
    uint64_t& r = t.chaos_rng;
    r ^= r << 13; r ^= r >> 7; r ^= r << 17;
    if (r & 1) return;
    const unsigned long long us = 1 + (r >> 8) % g->chaos_us;
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, stream, us * 100ull);
    (void)hipGetLastError();
}

// ref MPI.Cart_coords / Cart_shift on a px × py grid, last dimension fastest (src/parameters.jl:441-447). The reference's
// grid is never periodic (MPI.Cart_create's default, :432); `periodic` is the test aid of armon_hip_mgpu_set_periodic.
void set_topology(tile_t& t, int rank, int px, int py, const bool periodic[2])
{
    t.rank = rank;
    t.cx = rank / py;
    t.cy = rank % py;
    auto at = [&](int cx, int cy) {
        if (periodic[0]) cx = (cx + px) % px;
        if (periodic[1]) cy = (cy + py) % py;
        return (cx < 0 || cx >= px || cy < 0 || cy >= py) ? -1 : cx * py + cy;
    };
    t.nb[ARMON_SIDE_LEFT] = at(t.cx - 1, t.cy);
    t.nb[ARMON_SIDE_RIGHT] = at(t.cx + 1, t.cy);
    t.nb[ARMON_SIDE_BOTTOM] = at(t.cx, t.cy - 1);
    t.nb[ARMON_SIDE_TOP] = at(t.cx, t.cy + 1);
}

int make_tile_resources(tile_t& t, bool alone_on_device)
{
    ARMON_HIP_TRY(hipSetDevice(t.device));
    // The transfer stream carries small launches (packs, unpacks, LAG-wide strips) beside the interior sweep of the compute
    // stream. At the LOWEST priority the interior's workgroups are placed first: the Y march fills the device with exactly
    // one round of long-lived workgroups, and edge launches that slip in between cost it a second round (4096 x 8192 tile with
    // four remote sides: 0.93 -> 0.80 ms per cycle, profiles/r05_enqueue_and_priority.txt). Only when the tile has its device
    // to itself: tiles that share one (tests, one-GPU rehearsals) would starve each other's transfer streams behind their
    // interiors (8 tiles on one device: 6.3 -> 6.7 ms). ARMON_MGPU_XFER_PRIORITY=normal / lowest overrides.
    {
        int least = 0, greatest = 0;
        const char* v = getenv("ARMON_MGPU_XFER_PRIORITY");
        bool normal = v && *v ? !strcmp(v, "normal") : !alone_on_device;
        // (a runtime without stream priorities is not an error: the transfer stream is then an ordinary one)
        if (!normal && (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
                        hipStreamCreateWithPriority(&t.xfer, hipStreamNonBlocking, least) != hipSuccess)) {
            (void)hipGetLastError();
            t.xfer = nullptr;
            normal = true;
        }
        if (normal) ARMON_HIP_TRY(hipStreamCreateWithFlags(&t.xfer, hipStreamNonBlocking));
    }
    for (int s = 0; s < kSides; s++) {
        ARMON_HIP_TRY(hipEventCreateWithFlags(&t.e_pack[s], hipEventDisableTiming));
        ARMON_HIP_TRY(hipEventCreateWithFlags(&t.e_recv[s], hipEventDisableTiming));
        ARMON_HIP_TRY(hipEventCreateWithFlags(&t.e_unpack[s], hipEventDisableTiming));
    }
    ARMON_HIP_TRY(hipEventCreateWithFlags(&t.e_red, hipEventDisableTiming));
    ARMON_HIP_TRY(hipEventCreateWithFlags(&t.e_state, hipEventDisableTiming));
    ARMON_HIP_TRY(hipEventCreateWithFlags(&t.e_intdt, hipEventDisableTiming));
    ARMON_HIP_TRY(hipEventCreateWithFlags(&t.e_dtdone, hipEventDisableTiming));
    ARMON_HIP_TRY(hipEventCreateWithFlags(&t.e_edge, hipEventDisableTiming));
    int rc = armon_hip_init(t.device, (void*)t.xfer, &t.edge);
    if (rc != ARMON_OK) return rc;
    ARMON_HIP_TRY(hipMalloc((void**)&t.edge_dt, 2 * sizeof(double)));
    const unsigned neutral[4] = {kEdgeNeutral, kEdgeNeutral, kEdgeNeutral, kEdgeNeutral};
    ARMON_HIP_TRY(hipMemcpy(t.edge_dt, neutral, sizeof neutral, hipMemcpyHostToDevice));
    return ARMON_OK;
}

int ensure_face_buffers(armon_mgpu* g, tile_t& t, int side, size_t bytes)
{
    if (bytes <= t.cap[side]) return ARMON_OK;
    // growing is rare (the first exchange of a side, or a wider set of variables later): drain every stream of the group —
    // a neighbour's transfer stream may still be reading the buffer that is about to be freed — then replace
    for (tile_t& o : g->tiles) {
        ARMON_HIP_TRY(hipSetDevice(o.device));
        ARMON_HIP_TRY(hipStreamSynchronize(o.ctx->stream));
        ARMON_HIP_TRY(hipStreamSynchronize(o.xfer));
    }
    ARMON_HIP_TRY(hipSetDevice(t.device));
    if (t.send[side]) ARMON_HIP_TRY(hipFree(t.send[side]));
    if (t.recv[side]) ARMON_HIP_TRY(hipFree(t.recv[side]));
    t.send[side] = t.recv[side] = nullptr;
    t.cap[side] = 0;
    ARMON_HIP_TRY(hipMalloc(&t.send[side], bytes));
    ARMON_HIP_TRY(hipMalloc(&t.recv[side], bytes));
    t.cap[side] = bytes;
    t.rec_recv[side] = t.rec_unpack[side] = false;
    return ARMON_OK;
}

template <typename T> int pack(armon_ctx*, armon_range, int, int64_t, T*, int, const T* const*);
template <> int pack<double>(armon_ctx* c, armon_range r, int g, int64_t face, double* a, int nv, const double* const* v)
{
    return armon_hip_pack_to_array(c, r, g, face, a, nv, v);
}
template <> int pack<float>(armon_ctx* c, armon_range r, int g, int64_t face, float* a, int nv, const float* const* v)
{
    return armon_hip_pack_to_array_f32(c, r, g, face, a, nv, v);
}
template <typename T> int unpack(armon_ctx*, armon_range, int, int64_t, const T*, int, T* const*);
template <> int unpack<double>(armon_ctx* c, armon_range r, int g, int64_t face, const double* a, int nv, double* const* v)
{
    return armon_hip_unpack_from_array(c, r, g, face, a, nv, v);
}
template <> int unpack<float>(armon_ctx* c, armon_range r, int g, int64_t face, const float* a, int nv, float* const* v)
{
    return armon_hip_unpack_from_array_f32(c, r, g, face, a, nv, v);
}

int check_desc(const armon_mgpu* g, int axis, const armon_halo_desc* d)
{
    ARMON_REQUIRE(g && d, "NULL argument");
    ARMON_REQUIRE(axis == ARMON_AXIS_X || axis == ARMON_AXIS_Y, "invalid axis %d", axis);
    for (size_t k = 0; k < g->tiles.size(); k++) {
        ARMON_REQUIRE(d[k].nx > 0 && d[k].ny > 0 && d[k].nghost > 0, "tile %zu: empty tile or no ghost layer", k);
        ARMON_REQUIRE(d[k].nvars >= 1 && d[k].nvars <= 8, "tile %zu: nvars = %d (1..8)", k, d[k].nvars);
        ARMON_REQUIRE((axis == ARMON_AXIS_X ? d[k].nx : d[k].ny) >= d[k].nghost,
                      "tile %zu has fewer cells along the axis than ghost layers (ref src/parameters.jl:684-690)", k);
        for (int v = 0; v < d[k].nvars; v++) ARMON_REQUIRE(d[k].vars[v], "tile %zu: vars[%d] is NULL", k, v);
    }
    return ARMON_OK;
}

// What an exchange_start that fails half-way leaves behind: sides marked in flight whose transfer was never posted. Every
// later start would then refuse ("already started") and a finish would unpack a buffer nothing was received into, so the
// group would be unusable after one recoverable error. Drain the local streams (a pack or a copy of this call may be
// running) and clear the marks of the sides this call touched.
void abandon_start(armon_mgpu* g, int s0)
{
    for (tile_t& t : g->tiles) {
        (void)hipSetDevice(t.device);
        (void)hipStreamSynchronize(t.ctx->stream);
        (void)hipStreamSynchronize(t.xfer);
        for (int s = s0; s < s0 + 2; s++) t.inflight[s] = 0;
    }
    (void)hipGetLastError();
}

// ---- the exchange, tile by tile -------------------------------------------------------------------------------------------
// Every step below works on ONE local tile, so that the group-wide entry points (loops over the tiles in this thread) and
// the cycle driver (armon_hip_mgpu_cycle: one host thread per tile, barriers between the steps) share them. What a step
// needs from ANOTHER tile — its pack / receive events, its send buffer — must have been produced by a step that is
// complete for every tile (loop finished / barrier passed) before it starts: start_check → start_pack → start_move.

// 0. everything that can be refused is refused BEFORE anything is packed or marked: an exchange still in flight, the sizes
//    the two ends of a face expect (in-process: both descriptors are here), and the face buffers (their growth drains
//    every stream of the group and allocates: serial section, never from a tile's thread)
template <typename T>
int start_check(armon_mgpu* g, size_t k, int axis, const armon_halo_desc* d, bool idle_now = true)
{
    const int s0 = first_side(axis);
    tile_t& t = g->tiles[k];
    for (int s = s0; s < s0 + 2; s++) {
        if (t.nb[s] < 0) continue;
        // (idle_now = false: checked for a step that runs later, after other exchanges of the same side — the cycle driver;
        // "in flight" is then checked by the step itself, and a buffer in use is never replaced: see below)
        ARMON_REQUIRE(!idle_now || t.inflight[s] == 0, "halo exchange of side %d already started (finish it first)", s);
        int64_t face;
        int rc = armon_hip_halo_ranges(d[k].nx, d[k].ny, d[k].nghost, s, nullptr, nullptr, &face);
        if (rc != ARMON_OK) return rc;
        const size_t bytes = (size_t)face * d[k].nghost * d[k].nvars * sizeof(T);
        if (!g->rccl) {
            const armon_halo_desc& dn = d[t.nb[s]];
            int64_t nface;
            rc = armon_hip_halo_ranges(dn.nx, dn.ny, dn.nghost, opposite(s), nullptr, nullptr, &nface);
            if (rc != ARMON_OK) return rc;
            const size_t nbytes = (size_t)nface * dn.nghost * dn.nvars * sizeof(T);
            ARMON_REQUIRE(nbytes == bytes, "tiles %d and %d disagree on the size of their common face (%zu vs %zu bytes)",
                          t.rank, g->tiles[t.nb[s]].rank, bytes, nbytes);
        }
        ARMON_REQUIRE(t.inflight[s] == 0 || bytes <= t.cap[s], "the face buffers of side %d would have to grow while an exchange "
                      "of that side is in flight", s);
        rc = ensure_face_buffers(g, t, s, bytes);
        if (rc != ARMON_OK) return rc;
    }
    return ARMON_OK;
}

// 1. pack the remote faces of tile k — on its TRANSFER stream, behind an event that marks the state on the compute stream
//    (everything the caller enqueued there before the exchange was started): the packs then run beside whatever the compute
//    stream does next (the interior of the sweep) instead of in front of it — two small launches less per sweep on the chain
//    that bounds a small tile's cycle (the 4096 x 8192 tile of a 16384² grid on 8 GPUs spends 0.34 ms in an interior).
template <typename T>
int start_pack(armon_mgpu* g, size_t k, int axis, const armon_halo_desc* d, bool& touched)
{
    const int s0 = first_side(axis);
    tile_t& t = g->tiles[k];
    for (int s = s0; s < s0 + 2; s++)
        ARMON_REQUIRE(t.nb[s] < 0 || t.inflight[s] == 0, "halo exchange of side %d already started (finish it first)", s);
    if (t.nb[s0] < 0 && t.nb[s0 + 1] < 0) return ARMON_OK;
    ARMON_HIP_TRY(hipSetDevice(t.device));
    chaos(g, t, t.ctx->stream);
    // (ARMON_MGPU_PACK=compute: the round-4 form, packs in front of the interior on the compute stream — the A/B of
    // tools/r05/enqueue_time.py)
    armon_ctx* pc = g->pack_on_compute ? t.ctx : t.edge;
    if (!g->pack_on_compute) {
        ARMON_HIP_TRY(hipEventRecord(t.e_state, t.ctx->stream));
        ARMON_HIP_TRY(hipStreamWaitEvent(t.xfer, t.e_state, 0));
    }
    const bool both = t.nb[s0] >= 0 && t.nb[s0 + 1] >= 0;
    armon_range border[2];
    int64_t face = 0;
    for (int s = s0; s < s0 + 2; s++) {
        if (t.nb[s] < 0) continue;
        int rc = armon_hip_halo_ranges(d[k].nx, d[k].ny, d[k].nghost, s, &border[s - s0], nullptr, &face);
        if (rc != ARMON_OK) return rc;
        // the previous message of this side must have left send[s]: in-process, the neighbour's copy out of it (its event);
        // RCCL: our own send, earlier on this same stream
        if (!g->rccl) {
            tile_t& n = g->tiles[t.nb[s]];
            if (n.rec_recv[opposite(s)]) ARMON_HIP_TRY(hipStreamWaitEvent(pc->stream, n.e_recv[opposite(s)], 0));
        }
    }
    const size_t bytes = (size_t)face * d[k].nghost * d[k].nvars * sizeof(T);
    chaos(g, t, pc->stream);
    touched = true;
    if (both) {
        // the two faces of the axis in ONE launch
        T* const arrays[2] = {static_cast<T*>(t.send[s0]), static_cast<T*>(t.send[s0 + 1])};
        int rc = pack_pair<T>(pc, border, d[k].nghost, face, arrays, d[k].nvars, reinterpret_cast<T* const*>(d[k].vars), true);
        if (rc != ARMON_OK) return rc;
    }
    for (int s = s0; s < s0 + 2; s++) {
        if (t.nb[s] < 0) continue;
        if (!both) {
            int rc = pack<T>(pc, border[s - s0], d[k].nghost, face, static_cast<T*>(t.send[s]), d[k].nvars,
                             reinterpret_cast<const T* const*>(d[k].vars));
            if (rc != ARMON_OK) return rc;
        }
        ARMON_HIP_TRY(hipEventRecord(t.e_pack[s], pc->stream));
        t.inflight[s] = bytes;
    }
    return ARMON_OK;
}

// 2. move the faces of tile k on its transfer stream (every tile's packs have been enqueued)
template <typename T>
int start_move(armon_mgpu* g, size_t k, int axis)
{
    const int s0 = first_side(axis);
    tile_t& t = g->tiles[k];
    bool any = false;
    for (int s = s0; s < s0 + 2; s++) any = any || t.inflight[s] != 0;
    if (!any) return ARMON_OK;
    ARMON_HIP_TRY(hipSetDevice(t.device));
    chaos(g, t, t.xfer);                         // late faces
    chaos(g, t, t.ctx->stream);                  // or a late interior: the edge work then runs ahead of it
    if (g->rccl) {
        for (int s = s0; s < s0 + 2; s++) {
            if (!t.inflight[s]) continue;
            ARMON_HIP_TRY(hipStreamWaitEvent(t.xfer, t.e_pack[s], 0));
            if (t.rec_unpack[s]) ARMON_HIP_TRY(hipStreamWaitEvent(t.xfer, t.e_unpack[s], 0));   // recv[s] free again
        }
        const ncclDataType_t dt = sizeof(T) == 8 ? ncclDouble : ncclFloat;
        // Sends in side order, receives in the OPPOSITE order: RCCL matches the k-th send to a peer with that peer's
        // k-th receive from us. With two different neighbours the order is irrelevant; when both sides have the SAME
        // peer (a periodic grid of one or two tiles along the axis — armon_hip_mgpu_set_periodic — the rank itself
        // included) our low face must land in the peer's HIGH ghosts, i.e. in the receive it posts second-to-last.
        ARMON_RCCL_TRY(g_rccl.GroupStart());
        for (int s = s0; s < s0 + 2; s++) {
            if (!t.inflight[s]) continue;
            ARMON_RCCL_TRY(g_rccl.Send(t.send[s], t.inflight[s] / sizeof(T), dt, t.nb[s], g->comm_halo, t.xfer));
        }
        for (int s = s0 + 1; s >= s0; s--) {
            if (!t.inflight[s]) continue;
            ARMON_RCCL_TRY(g_rccl.Recv(t.recv[s], t.inflight[s] / sizeof(T), dt, t.nb[s], g->comm_halo, t.xfer));
        }
        ARMON_RCCL_TRY(g_rccl.GroupEnd());
        for (int s = s0; s < s0 + 2; s++) {
            if (!t.inflight[s]) continue;
            ARMON_HIP_TRY(hipEventRecord(t.e_recv[s], t.xfer));
            t.rec_recv[s] = true;
        }
    } else {
        for (int s = s0; s < s0 + 2; s++) {
            if (!t.inflight[s]) continue;
            tile_t& n = g->tiles[t.nb[s]];
            const int os = opposite(s);
            ARMON_HIP_TRY(hipStreamWaitEvent(t.xfer, n.e_pack[os], 0));                          // neighbour packed
            if (t.rec_unpack[s]) ARMON_HIP_TRY(hipStreamWaitEvent(t.xfer, t.e_unpack[s], 0));   // recv[s] free again
            if (n.device == t.device && !g->force_peer)
                ARMON_HIP_TRY(hipMemcpyAsync(t.recv[s], n.send[os], t.inflight[s], hipMemcpyDeviceToDevice, t.xfer));
            else
                ARMON_HIP_TRY(hipMemcpyPeerAsync(t.recv[s], t.device, n.send[os], n.device, t.inflight[s], t.xfer));
            ARMON_HIP_TRY(hipEventRecord(t.e_recv[s], t.xfer));
            t.rec_recv[s] = true;
        }
    }
    return ARMON_OK;
}

template <typename T>
int exchange_start(armon_mgpu* g, int axis, const armon_halo_desc* d)
{
    int rc = check_desc(g, axis, d);
    if (rc != ARMON_OK) return rc;
    const size_t nt = g->tiles.size();
    for (size_t k = 0; k < nt && rc == ARMON_OK; k++) rc = start_check<T>(g, k, axis, d);
    if (rc != ARMON_OK) return rc;
    bool touched = false;
    for (size_t k = 0; k < nt && rc == ARMON_OK; k++) rc = start_pack<T>(g, k, axis, d, touched);
    for (size_t k = 0; k < nt && rc == ARMON_OK; k++) rc = start_move<T>(g, k, axis);
    if (rc != ARMON_OK && touched) abandon_start(g, first_side(axis));     // the message of `rc` stays in last_error
    return rc;
}

// unpack the received faces of tile k into its ghost cells: on the compute stream (which waits for the receive), or on
// the transfer stream, where the unpack simply follows the receive (on_edge)
template <typename T>
int finish_one(armon_mgpu* g, size_t k, int axis, const armon_halo_desc* d, bool on_edge)
{
    const int s0 = first_side(axis);
    tile_t& t = g->tiles[k];
    if (!t.inflight[s0] && !t.inflight[s0 + 1]) return ARMON_OK;
    const bool both = t.inflight[s0] && t.inflight[s0 + 1];
    armon_ctx* c = on_edge ? t.edge : t.ctx;
    armon_range ghost[2];
    int64_t face = 0;
    ARMON_HIP_TRY(hipSetDevice(t.device));
    for (int s = s0; s < s0 + 2; s++) {
        if (!t.inflight[s]) continue;
        int rc = armon_hip_halo_ranges(d[k].nx, d[k].ny, d[k].nghost, s, nullptr, &ghost[s - s0], &face);
        if (rc != ARMON_OK) return rc;
        ARMON_REQUIRE((size_t)face * d[k].nghost * d[k].nvars * sizeof(T) == t.inflight[s],
                      "tile %zu: finish does not match the exchange that was started", k);
        if (!on_edge) ARMON_HIP_TRY(hipStreamWaitEvent(c->stream, t.e_recv[s], 0));
    }
    chaos(g, t, c->stream);
    if (both) {                                  // both sides in ONE launch, behind both receives
        T* const arrays[2] = {static_cast<T*>(t.recv[s0]), static_cast<T*>(t.recv[s0 + 1])};
        int rc = pack_pair<T>(c, ghost, d[k].nghost, face, arrays, d[k].nvars, reinterpret_cast<T* const*>(d[k].vars), false);
        if (rc != ARMON_OK) return rc;
    }
    for (int s = s0; s < s0 + 2; s++) {
        if (!t.inflight[s]) continue;
        if (!both) {
            int rc = unpack<T>(c, ghost[s - s0], d[k].nghost, face, static_cast<const T*>(t.recv[s]), d[k].nvars,
                               reinterpret_cast<T* const*>(d[k].vars));
            if (rc != ARMON_OK) return rc;
        }
        ARMON_HIP_TRY(hipEventRecord(t.e_unpack[s], c->stream));
        t.rec_unpack[s] = true;
        t.inflight[s] = 0;
    }
    return ARMON_OK;
}

template <typename T>
int exchange_finish(armon_mgpu* g, int axis, const armon_halo_desc* d, bool on_edge = false)
{
    int rc = check_desc(g, axis, d);
    for (size_t k = 0; k < g->tiles.size() && rc == ARMON_OK; k++) rc = finish_one<T>(g, k, axis, d, on_edge);
    return rc;
}

template <typename T>
__global__ void k_min_broadcast(T* __restrict__ slots, int stride, int n)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        T m = slots[0];
        for (int k = 1; k < n; k++) m = (slots[k * stride] < m || slots[k * stride] != slots[k * stride]) ? slots[k * stride] : m;
        for (int k = 0; k < n; k++) slots[k * stride] = m;
    }
}

// dst = min(dst, e[0], e[1]), then both scalars neutral again (NaN wins, so that the host's validity check sees it)
template <typename T>
__global__ void k_fold_edge_dt(T* __restrict__ dst, T* __restrict__ e)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        T m = *dst;
        for (int k = 0; k < 2; k++) {
            const T v = e[k];
            m = (v < m || v != v) ? v : m;
        }
        *dst = m;
        unsigned* w = reinterpret_cast<unsigned*>(e);
        for (int k = 0; k < 4; k++) w[k] = kEdgeNeutral;
    }
}

// the compute stream of tile k waits for its edge work; with dt_dev, dt_dev = min(dt_dev, the two edge scalars)
template <typename T>
int edge_join_one(armon_mgpu* g, size_t k, T* dt_dev)
{
    tile_t& t = g->tiles[k];
    ARMON_HIP_TRY(hipSetDevice(t.device));
    chaos(g, t, t.xfer);                         // strips that finish long after the interior
    ARMON_HIP_TRY(hipEventRecord(t.e_edge, t.xfer));
    ARMON_HIP_TRY(hipStreamWaitEvent(t.ctx->stream, t.e_edge, 0));
    if (dt_dev) {
        hipLaunchKernelGGL(k_fold_edge_dt<T>, dim3(1), dim3(64), 0, t.ctx->stream, dt_dev, reinterpret_cast<T*>(t.edge_dt));
        return check_launch("fold_edge_dt");
    }
    return ARMON_OK;
}

template <typename T>
int edge_join(armon_mgpu* g, T* const* dt_dev)
{
    ARMON_REQUIRE(g, "NULL argument");
    for (size_t k = 0; k < g->tiles.size(); k++) {
        if (dt_dev) ARMON_REQUIRE(dt_dev[k], "dt_dev[%zu] is NULL", k);
        int rc = edge_join_one<T>(g, k, dt_dev ? dt_dev[k] : nullptr);
        if (rc != ARMON_OK) return rc;
    }
    return ARMON_OK;
}

template <typename T>
__global__ void k_nan_to_neg_inf(T* __restrict__ x)
{
    const T v = *x;
    if (v != v) *x = -T(INFINITY);
}

// every tile on ONE device (up to 64 of them): lane k reads tile k's scalar, the wave folds, lane k writes the minimum back
constexpr int kMaxDirect = 64;
struct dt_ptrs { void* p[kMaxDirect]; };
template <typename T>
__global__ void __launch_bounds__(64) k_min_broadcast_direct(dt_ptrs d, int n)
{
    const int lane = threadIdx.x;
    T* mine = static_cast<T*>(d.p[lane < n ? lane : 0]);
    T m = *mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_xor(m, off, 64);
        m = (o < m || o != o) ? o : m;          // a NaN anywhere must reach the host's validity check
    }
    if (lane < n) *mine = m;
}

// The global minimum in three steps, each per tile (same rule as the exchange: a step starts when the previous one is
// complete for every tile). RCCL: one all-reduce on the tile's compute stream, in red_post. In-process: every tile records
// "my scalar is ready" (red_post); tile 0 gathers on its transfer stream, folds, scatters (red_root); every compute
// stream waits for the result (red_wait).
// (on_xfer: the cycle driver's form — the scalar is folded, reduced and read back on the tile's TRANSFER stream, so that the
// compute stream goes from a cycle's last sweep to the next cycle's first without waiting for an all-reduce it does not need:
// the reference consumes the result one cycle later, ref src/solver_state.jl:89-99,145-166)
template <typename T>
int red_post(armon_mgpu* g, size_t k, T* dt_dev, bool on_xfer = false)
{
    tile_t& t = g->tiles[k];
    hipStream_t st = on_xfer ? t.xfer : t.ctx->stream;
    ARMON_HIP_TRY(hipSetDevice(t.device));
    if (g->rccl) {
        // ncclMin is free to drop a NaN operand; -inf survives any minimum and fails the host's validity check just as well
        // (ref src/solver_state.jl:123-124: `!isfinite(new_dt) || new_dt <= 0`)
        hipLaunchKernelGGL(k_nan_to_neg_inf<T>, dim3(1), dim3(1), 0, st, dt_dev);
        int rc = check_launch("nan_to_neg_inf");
        if (rc != ARMON_OK) return rc;
        ARMON_RCCL_TRY(g_rccl.AllReduce(dt_dev, dt_dev, 1, sizeof(T) == 8 ? ncclDouble : ncclFloat, ncclMin,
                                        g->comm_red, st));
        return ARMON_OK;
    }
    if (g->tiles.size() == 1) return ARMON_OK;
    chaos(g, t, st);
    ARMON_HIP_TRY(hipEventRecord(t.e_red, st));
    return ARMON_OK;
}

template <typename T>
int red_root(armon_mgpu* g, T* const* dt_dev)
{
    const size_t nt = g->tiles.size();
    if (g->rccl || nt == 1) return ARMON_OK;
    tile_t& root = g->tiles[0];
    ARMON_HIP_TRY(hipSetDevice(root.device));
    chaos(g, root, root.xfer);
    bool one_device = nt <= (size_t)kMaxDirect && !g->force_peer;    // force_peer: take the several-device path below
    for (size_t k = 0; k < nt; k++) one_device = one_device && g->tiles[k].device == root.device;
    for (size_t k = 0; k < nt; k++) ARMON_HIP_TRY(hipStreamWaitEvent(root.xfer, g->tiles[k].e_red, 0));
    if (one_device) {
        // one kernel on tile 0's transfer stream instead of 2·nt serialized copies (the chain between two cycles)
        dt_ptrs d;
        for (size_t k = 0; k < nt; k++) d.p[k] = dt_dev[k];
        hipLaunchKernelGGL(k_min_broadcast_direct<T>, dim3(1), dim3(64), 0, root.xfer, d, (int)nt);
        int rc = check_launch("min_broadcast_direct");
        if (rc != ARMON_OK) return rc;
    } else {
        // several devices: gather on tile 0's device (its transfer stream), fold, scatter back
        constexpr int stride = sizeof(double) / sizeof(T);       // one 8-byte slot per tile
        T* slots = reinterpret_cast<T*>(g->red_buf);
        for (size_t k = 0; k < nt; k++) {
            tile_t& t = g->tiles[k];
            if (t.device == root.device && !g->force_peer)
                ARMON_HIP_TRY(hipMemcpyAsync(slots + k * stride, dt_dev[k], sizeof(T), hipMemcpyDeviceToDevice, root.xfer));
            else
                ARMON_HIP_TRY(hipMemcpyPeerAsync(slots + k * stride, root.device, dt_dev[k], t.device, sizeof(T), root.xfer));
        }
        hipLaunchKernelGGL(k_min_broadcast<T>, dim3(1), dim3(64), 0, root.xfer, slots, stride, (int)nt);
        int rc = check_launch("min_broadcast");
        if (rc != ARMON_OK) return rc;
        for (size_t k = 0; k < nt; k++) {
            tile_t& t = g->tiles[k];
            if (t.device == root.device && !g->force_peer)
                ARMON_HIP_TRY(hipMemcpyAsync(dt_dev[k], slots + k * stride, sizeof(T), hipMemcpyDeviceToDevice, root.xfer));
            else
                ARMON_HIP_TRY(hipMemcpyPeerAsync(dt_dev[k], t.device, slots + k * stride, root.device, sizeof(T), root.xfer));
        }
    }
    ARMON_HIP_TRY(hipEventRecord(g->e_red_done, root.xfer));
    return ARMON_OK;
}

int red_wait(armon_mgpu* g, size_t k, bool on_xfer = false)
{
    if (g->rccl || g->tiles.size() == 1) return ARMON_OK;
    tile_t& t = g->tiles[k];
    ARMON_HIP_TRY(hipSetDevice(t.device));
    ARMON_HIP_TRY(hipStreamWaitEvent(on_xfer ? t.xfer : t.ctx->stream, g->e_red_done, 0));
    return ARMON_OK;
}

template <typename T>
int dt_allreduce(armon_mgpu* g, T* const* dt_dev)
{
    ARMON_REQUIRE(g && dt_dev, "NULL argument");
    const size_t nt = g->tiles.size();
    for (size_t k = 0; k < nt; k++) ARMON_REQUIRE(dt_dev[k], "dt_dev[%zu] is NULL", k);
    int rc = ARMON_OK;
    for (size_t k = 0; k < nt && rc == ARMON_OK; k++) rc = red_post<T>(g, k, dt_dev[k]);
    if (rc == ARMON_OK) rc = red_root<T>(g, dt_dev);
    for (size_t k = 0; k < nt && rc == ARMON_OK; k++) rc = red_wait(g, k);
    return rc;
}


// ---- a whole solver cycle of every local tile in ONE call ------------------------------------------------------------------
// What the host mirror did call by call (multi_tile.py: exchange_start, interior sweep, finish_edge, strips, edge_join, ...,
// dt_allreduce: ≈ 15 library calls per tile and sweep from an interpreter) as one native entry point, so that the enqueue
// cost of a cycle stays well under the 0.75 ms of GPU work of the 8-GPU strong-scaling tile (4096 x 8192). A group of
// several local tiles is driven by one host thread per tile (the HIP runtime serialises a thread's calls; 8 devices fed
// from one thread cost 8 x the enqueue time): the cycle is a list of steps, a step is executed for tile k by thread k,
// and a barrier separates two steps — the rule of the per-tile functions above (the threads: tile_pool.hpp).

template <typename T> struct cycle_traits;
template <> struct cycle_traits<double> {
    using desc = armon_sweep_desc;
    using tile = armon_tile_cycle;
    static int sweep(armon_ctx* c, const desc* d) { return armon_hip_sweep(c, d); }
};
template <> struct cycle_traits<float> {
    using desc = armon_sweep_desc_f32;
    using tile = armon_tile_cycle_f32;
    static int sweep(armon_ctx* c, const desc* d) { return armon_hip_sweep_f32(c, d); }
};

struct host_timer {
    armon_mgpu* g;
    int slot;
    bool on;
    std::chrono::steady_clock::time_point t0;
    host_timer(armon_mgpu* g_, size_t k, int slot_) : g(g_), slot(slot_), on(g_->timing > 0 && k == 0)
    {
        if (on) t0 = std::chrono::steady_clock::now();
    }
    ~host_timer()
    {
        if (on) g->t_host[slot] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
};
const char* const kHostTimerNames[8] = {"pack", "move", "interior / full sweep", "unpack", "strips", "join", "dt reduction", "prefetch"};

bool want_threads(armon_mgpu* g)
{
    if (g->use_threads < 0) {
        const char* v = getenv("ARMON_MGPU_THREADS");
        g->use_threads = (v && *v) ? atoi(v) != 0 : 1;
    }
    return g->use_threads != 0 && g->tiles.size() > 1;
}

int run_steps(armon_mgpu* g, const std::vector<std::function<int(size_t)>>& steps)
{
    if (want_threads(g)) {
        if (!g->pool) g->pool = new tile_pool((int)g->tiles.size(), [] { return std::string(armon_hip_last_error()); });
        int tile = -1;
        std::string message;
        const int rc = g->pool->run(steps, &tile, &message);
        if (rc != ARMON_OK) set_error("tile %d: %s", tile, message.c_str());
        return rc;
    }
    for (const auto& step : steps)
        for (size_t k = 0; k < g->tiles.size(); k++) {
            int rc = step(k);
            if (rc != ARMON_OK) return rc;
        }
    return ARMON_OK;
}

// the descriptor of sweep `s` of the cycle for one tile: the tile's X or Y template with the pointer sets in the roles they
// have after s sweeps, the step of that sweep, the cycle's outputs on the last sweep only, the sides from the topology
template <typename T>
typename cycle_traits<T>::desc sweep_of(const tile_t& t, const armon_cycle_plan& plan, const typename cycle_traits<T>::tile& tc, int s)
{
    typename cycle_traits<T>::desc d = plan.axis[s] == ARMON_AXIS_X ? tc.x : tc.y;
    if (s & 1) {
        const T *a = d.rho_in, *b = d.u_in, *c = d.v_in, *e = d.E_in;
        d.rho_in = d.rho_out; d.u_in = d.u_out; d.v_in = d.v_out; d.E_in = d.E_out;
        d.rho_out = const_cast<T*>(a); d.u_out = const_cast<T*>(b); d.v_out = const_cast<T*>(c); d.E_out = const_cast<T*>(e);
    }
    const bool last = s == plan.n_sweeps - 1;
    d.dt = plan.dt[s];
    if (!(last && plan.emit_p)) d.p_out = d.c_out = nullptr;
    if (!(last && plan.emit_dt)) d.dt_cfl_out = nullptr;
    const int s0 = first_side(plan.axis[s]);
    d.bc_low = t.nb[s0] < 0;
    d.bc_high = t.nb[s0 + 1] < 0;
    d.out_lo = d.out_hi = 0;
    d.dt_accumulate = 0;
    d.dt_state = nullptr;
    return d;
}

template <typename D>
armon_halo_desc halo_of(const D& d)
{
    armon_halo_desc h{};
    h.nx = d.nx;
    h.ny = d.ny;
    h.nghost = d.nghost;
    h.nvars = 4;
    h.vars[0] = (void*)d.rho_in; h.vars[1] = (void*)d.u_in; h.vars[2] = (void*)d.v_in; h.vars[3] = (void*)d.E_in;
    return h;
}

template <typename T>
int mgpu_cycle(armon_mgpu* g, const armon_cycle_plan* plan_, const typename cycle_traits<T>::tile* tcs)
{
    using D = typename cycle_traits<T>::desc;
    ARMON_REQUIRE(g && plan_ && tcs, "NULL argument");
    const armon_cycle_plan plan = *plan_;
    const size_t nt = g->tiles.size();
    ARMON_REQUIRE(plan.n_sweeps >= 1 && plan.n_sweeps <= 3, "a cycle has 1 to 3 sweeps (ref src/axis_splitting.jl:24-46), not %d", plan.n_sweeps);
    for (int s = 0; s < plan.n_sweeps; s++)
        ARMON_REQUIRE(plan.axis[s] == ARMON_AXIS_X || plan.axis[s] == ARMON_AXIS_Y, "invalid axis %d", plan.axis[s]);
    ARMON_REQUIRE(plan.next_axis >= -1 && plan.next_axis <= ARMON_AXIS_Y, "invalid next_axis %d", plan.next_axis);
    ARMON_REQUIRE(plan.event_slot < 0 || plan.event_slot + 2 * plan.n_sweeps <= ARMON_HIP_MAX_EVENTS, "event slots out of the pool");
    ARMON_REQUIRE(plan.dt_event_slot < ARMON_HIP_MAX_EVENTS, "event slots out of the pool");
    ARMON_REQUIRE(!plan.event_ctx || plan.event_ctx->stream == g->tiles[0].ctx->stream,
                  "event_ctx must sit on the first local tile's compute stream");
    for (size_t k = 0; k < nt; k++) {
        ARMON_REQUIRE(!tcs[k].x.dt_state && !tcs[k].y.dt_state, "tile %zu: a device-resident time step cannot drive a tile cycle", k);
        ARMON_REQUIRE(!plan.emit_dt || (tcs[k].x.dt_cfl_out && tcs[k].y.dt_cfl_out), "tile %zu: emit_dt without dt_cfl_out", k);
    }

    // descriptors of every sweep and tile, and of the exchange posted ahead for the next cycle
    std::vector<D> sw(nt * 3);
    std::vector<armon_halo_desc> halo(nt * 4);           // [s * nt + k], s = 3: the prefetch
    std::vector<T*> dt_dev(nt, nullptr);
    for (size_t k = 0; k < nt; k++) {
        for (int s = 0; s < plan.n_sweeps; s++) {
            sw[s * nt + k] = sweep_of<T>(g->tiles[k], plan, tcs[k], s);
            halo[s * nt + k] = halo_of(sw[s * nt + k]);
        }
        if (plan.emit_dt) dt_dev[k] = sw[(plan.n_sweeps - 1) * nt + k].dt_cfl_out;
        if (plan.next_axis >= 0) {                    // the state after the cycle: what its last sweep wrote
            D d = plan.next_axis == ARMON_AXIS_X ? tcs[k].x : tcs[k].y;
            if (plan.n_sweeps & 1) {
                d.rho_in = d.rho_out; d.u_in = d.u_out; d.v_in = d.v_out; d.E_in = d.E_out;
            }
            halo[3 * nt + k] = halo_of(d);
        }
    }
    auto any_remote = [&](int axis) {
        const int s0 = first_side(axis);
        for (const tile_t& t : g->tiles)
            if (t.nb[s0] >= 0 || t.nb[s0 + 1] >= 0) return true;
        return false;
    };
    auto lag_of = [](const D& d) { return 2 + (d.scheme == ARMON_SCHEME_GAD) + (d.projection == ARMON_PROJECTION_EULER_2ND); };
    auto overlapped = [&](size_t k, const D& d) {
        const int64_t n = d.axis == ARMON_AXIS_X ? d.nx : d.ny;
        return plan.overlap && n >= 2 * lag_of(d) + 1 && g->tiles[k].edge != nullptr;
    };

    // serial section: whatever can be refused or has to allocate (face buffers grow on a side's first exchange)
    std::vector<bool> skip_start(3, false);
    for (int s = 0; s < plan.n_sweeps; s++) {
        const int axis = plan.axis[s];
        if (!any_remote(axis)) continue;
        if (s == 0 && g->prefetched_axis == axis) {
            skip_start[0] = true;                     // posted by the previous cycle
            continue;
        }
        ARMON_REQUIRE(!(s == 0 && g->prefetched_axis >= 0), "an exchange along axis %d was posted ahead; this cycle starts along %d "
                      "(armon_hip_mgpu_drain first)", g->prefetched_axis, axis);
        // ("in flight" is checked when the step runs: the side may be in use by the exchange posted ahead, or by an earlier
        // sweep of this very cycle — Strang's X, Y, X)
        int rc = check_desc(g, axis, &halo[s * nt]);
        for (size_t k = 0; k < nt && rc == ARMON_OK; k++) rc = start_check<T>(g, k, axis, &halo[s * nt], false);
        if (rc != ARMON_OK) return rc;
    }
    const bool prefetch = plan.next_axis >= 0 && any_remote(plan.next_axis);
    if (prefetch) {
        int rc = check_desc(g, plan.next_axis, &halo[3 * nt]);
        for (size_t k = 0; k < nt && rc == ARMON_OK; k++) rc = start_check<T>(g, k, plan.next_axis, &halo[3 * nt], false);
        if (rc != ARMON_OK) return rc;
    }
    g->prefetched_axis = -1;

    std::vector<std::function<int(size_t)>> steps;
    std::vector<char> touched(nt, 0);
    for (int s = 0; s < plan.n_sweeps; s++) {
        const int axis = plan.axis[s];
        const bool remote = any_remote(axis);
        const armon_halo_desc* h = &halo[s * nt];
        if (remote && !skip_start[s])
            steps.push_back([=, &touched](size_t k) -> int {
                bool t = false;
                host_timer ht(g, k, 0);
                int rc = start_pack<T>(g, k, axis, h, t);
                touched[k] = touched[k] || t;
                return rc;
            });
        steps.push_back([=, &sw](size_t k) -> int {
            tile_t& t = g->tiles[k];
            const D& d = sw[s * nt + k];
            int rc = ARMON_OK;
            if (remote && !skip_start[s]) {
                host_timer ht(g, k, 1);
                rc = start_move<T>(g, k, axis);
            }
            if (rc != ARMON_OK) return rc;
            ARMON_HIP_TRY(hipSetDevice(t.device));
            const bool timed = plan.event_slot >= 0 && k == 0;
            armon_ctx* ev = plan.event_ctx ? plan.event_ctx : t.ctx;
            // the sweep that writes the tile's CFL scalar must find the previous value consumed (reduced, read back: on the
            // transfer stream, long ago — this wait is the formal dependency, it never stalls)
            if (d.dt_cfl_out && t.rec_dtdone) ARMON_HIP_TRY(hipStreamWaitEvent(t.ctx->stream, t.e_dtdone, 0));
            if (timed && (rc = armon_hip_event_record(ev, plan.event_slot + 2 * s)) != ARMON_OK) return rc;
            const int s0 = first_side(axis);
            const bool lo_r = t.nb[s0] >= 0, hi_r = t.nb[s0 + 1] >= 0;
            if (!lo_r && !hi_r) {
                host_timer ht(g, k, 2);
                rc = cycle_traits<T>::sweep(t.ctx, &d);
            } else if (!overlapped(k, d)) {
                rc = finish_one<T>(g, k, axis, h, false);
                if (rc == ARMON_OK) rc = cycle_traits<T>::sweep(t.ctx, &d);
            } else {
                const int64_t n = axis == ARMON_AXIS_X ? d.nx : d.ny;
                const int lag = lag_of(d);
                D di = d;
                di.out_lo = lo_r ? lag : 0;
                di.out_hi = hi_r ? n - lag : n;
                {
                    host_timer ht(g, k, 2);
                    rc = cycle_traits<T>::sweep(t.ctx, &di);
                }
                if (rc == ARMON_OK) {
                    host_timer ht(g, k, 3);
                    rc = finish_one<T>(g, k, axis, h, true);
                }
                const int64_t ranges[2][2] = {{0, di.out_lo}, {di.out_hi, n}};
                for (int side = 0; side < 2 && rc == ARMON_OK; side++) {
                    if (ranges[side][0] == ranges[side][1]) continue;
                    D ds = d;
                    ds.out_lo = ranges[side][0];
                    ds.out_hi = ranges[side][1];
                    if (ds.dt_cfl_out) ds.dt_cfl_out = reinterpret_cast<T*>(t.edge_dt) + side;
                    host_timer ht(g, k, 4);
                    rc = cycle_traits<T>::sweep(t.edge, &ds);
                }
                if (rc == ARMON_OK) {
                    host_timer ht(g, k, 5);
                    rc = edge_join_one<T>(g, k, (T*)nullptr);          // the CFL scalars are folded on the transfer stream: below
                }
            }
            if (rc == ARMON_OK && timed) rc = armon_hip_event_record(ev, plan.event_slot + 2 * s + 1);
            return rc;
        });
    }
    if (plan.emit_dt) {
        // The next CFL step: tile scalar (compute stream: the last sweep's own fold) → transfer stream: + the strips' two
        // scalars → minimum over the tiles → pinned host slot + event. The compute stream only records that its part is
        // there; it does not wait for any of this.
        steps.push_back([=, &dt_dev](size_t k) -> int {
            host_timer ht(g, k, 6);
            tile_t& t = g->tiles[k];
            ARMON_HIP_TRY(hipSetDevice(t.device));
            ARMON_HIP_TRY(hipEventRecord(t.e_intdt, t.ctx->stream));
            ARMON_HIP_TRY(hipStreamWaitEvent(t.xfer, t.e_intdt, 0));
            chaos(g, t, t.xfer);
            hipLaunchKernelGGL(k_fold_edge_dt<T>, dim3(1), dim3(64), 0, t.xfer, dt_dev[k], reinterpret_cast<T*>(t.edge_dt));
            int rc = check_launch("fold_edge_dt");
            return rc != ARMON_OK ? rc : red_post<T>(g, k, dt_dev[k], true);
        });
        if (!g->rccl && nt > 1) {
            steps.push_back([=, &dt_dev](size_t k) -> int { return k == 0 ? red_root<T>(g, dt_dev.data()) : (int)ARMON_OK; });
        }
        steps.push_back([=, &dt_dev](size_t k) -> int {
            host_timer ht(g, k, 6);
            tile_t& t = g->tiles[k];
            int rc = red_wait(g, k, true);
            if (rc != ARMON_OK) return rc;
            ARMON_HIP_TRY(hipSetDevice(t.device));
            if (k == 0 && plan.dt_host) {
                ARMON_HIP_TRY(hipMemcpyAsync(plan.dt_host, dt_dev[0], sizeof(T), hipMemcpyDeviceToHost, t.xfer));
                if (plan.dt_event_slot >= 0 && (rc = armon_hip_event_record(t.edge, plan.dt_event_slot)) != ARMON_OK) return rc;
            }
            ARMON_HIP_TRY(hipEventRecord(t.e_dtdone, t.xfer));
            t.rec_dtdone = true;
            return ARMON_OK;
        });
    }
    if (prefetch) {
        const int axis = plan.next_axis;
        const armon_halo_desc* h = &halo[3 * nt];
        steps.push_back([=, &touched](size_t k) -> int {
            bool tch = false;
            host_timer ht(g, k, 7);
            int rc = start_pack<T>(g, k, axis, h, tch);
            touched[k] = touched[k] || tch;
            return rc;
        });
        steps.push_back([=](size_t k) -> int {
            host_timer ht(g, k, 7);
            return start_move<T>(g, k, axis);
        });
    }
    if (g->timing < 0) {
        const char* v = getenv("ARMON_MGPU_TIMING");
        g->timing = (v && *v) ? atoi(v) != 0 : 0;
    }
    g->n_cycles++;
    int rc = run_steps(g, steps);
    if (rc != ARMON_OK) {
        // leave the group usable: nothing in flight, no half-posted exchange (the error message stays)
        char keep[512];
        snprintf(keep, sizeof keep, "%s", armon_hip_last_error());
        abandon_start(g, ARMON_SIDE_LEFT);
        abandon_start(g, ARMON_SIDE_BOTTOM);
        set_error("%s", keep);
        return rc;
    }
    if (prefetch) g->prefetched_axis = plan.next_axis;
    return ARMON_OK;
}

// complete an exchange posted ahead that no cycle will consume (end of a run): unpack it where it was meant to go
template <typename T>
int mgpu_drain(armon_mgpu* g, const typename cycle_traits<T>::tile* tcs)
{
    ARMON_REQUIRE(g, "NULL argument");
    if (g->prefetched_axis < 0) return ARMON_OK;
    ARMON_REQUIRE(tcs, "NULL argument");
    const int axis = g->prefetched_axis;
    std::vector<armon_halo_desc> h(g->tiles.size());
    for (size_t k = 0; k < g->tiles.size(); k++) h[k] = halo_of(axis == ARMON_AXIS_X ? tcs[k].x : tcs[k].y);
    g->prefetched_axis = -1;
    return exchange_finish<T>(g, axis, h.data());
}

}  // namespace

extern "C" {

// border_domain / ghost_domain of a side with all `nghost` layers (ref src/blocking/blocking.jl:141-187 with
// single_strip=false, src/blocking/blocking.jl:71-85 for the linear indices), as 0-based armon_range of a tile of
// nx × ny real cells; face = real_face_size(bsize, side) (ref :210).
int armon_hip_halo_ranges(int64_t nx, int64_t ny, int nghost, int side, armon_range* border, armon_range* ghost, int64_t* face)
{
    ARMON_REQUIRE(nx > 0 && ny > 0 && nghost > 0, "empty tile or no ghost layer");
    ARMON_REQUIRE(side >= ARMON_SIDE_LEFT && side <= ARMON_SIDE_TOP, "invalid side %d", side);
    const int64_t g = nghost, row = nx + 2 * g;
    const int64_t origin = row * g + g;                       // first real cell
    armon_range b{}, gh{};
    b.col_step = gh.col_step = row;
    switch (side) {
    case ARMON_SIDE_LEFT:                                     // real columns [0, g) / ghost columns [-g, 0)
        b = {origin, row, ny, 0, g};
        gh = {origin, row, ny, -g, g};
        break;
    case ARMON_SIDE_RIGHT:                                    // real columns [nx-g, nx) / ghost columns [nx, nx+g)
        b = {origin + nx - 1, row, ny, -(g - 1), g};
        gh = {origin + nx - 1, row, ny, 1, g};
        break;
    case ARMON_SIDE_BOTTOM:                                   // real rows [0, g) / ghost rows [-g, 0)
        b = {origin, row, g, 0, nx};
        gh = {origin - g * row, row, g, 0, nx};
        break;
    default:                                                  // Top: real rows [ny-g, ny) / ghost rows [ny, ny+g)
        b = {origin + (ny - g) * row, row, g, 0, nx};
        gh = {origin + ny * row, row, g, 0, nx};
    }
    if (border) *border = b;
    if (ghost) *ghost = gh;
    if (face) *face = (side == ARMON_SIDE_LEFT || side == ARMON_SIDE_RIGHT) ? ny : nx;
    return ARMON_OK;
}

static bool pack_on_compute_env()
{
    const char* v = getenv("ARMON_MGPU_PACK");
    return v && !strcmp(v, "compute");
}

int armon_hip_mgpu_init(int px, int py, const int* device_ids, armon_mgpu** out)
{
    ARMON_REQUIRE(out, "group out pointer is NULL");
    *out = nullptr;
    ARMON_REQUIRE(px >= 1 && py >= 1 && px * py <= 4096, "invalid tile grid %d x %d", px, py);
    const int nt = px * py;
    armon_mgpu* g = new armon_mgpu();
    g->px = px;
    g->py = py;
    g->pack_on_compute = pack_on_compute_env();
    g->tiles.resize(nt);
    int rc = ARMON_OK;
    for (int r = 0; r < nt && rc == ARMON_OK; r++) {
        tile_t& t = g->tiles[r];
        set_topology(t, r, px, py, g->periodic);
        t.device = device_ids ? device_ids[r] : 0;
        rc = armon_hip_init(t.device, nullptr, &t.ctx);
        int sharing = 0;
        for (int q = 0; q < nt; q++) sharing += (device_ids ? device_ids[q] : 0) == t.device;
        if (rc == ARMON_OK) rc = make_tile_resources(t, sharing == 1);
    }
    if (rc == ARMON_OK) {
        // direct peer access between the devices of neighbouring tiles (xGMI); "already enabled" is fine
        for (tile_t& t : g->tiles)
            for (int s = 0; s < kSides; s++) {
                if (t.nb[s] < 0 || g->tiles[t.nb[s]].device == t.device) continue;
                (void)hipSetDevice(t.device);
                hipError_t e = hipDeviceEnablePeerAccess(g->tiles[t.nb[s]].device, 0);
                if (e != hipSuccess) (void)hipGetLastError();      // copies still work, staged by the runtime
            }
        tile_t& root = g->tiles[0];
        hipError_t e = hipSetDevice(root.device);
        if (e == hipSuccess) e = hipMalloc((void**)&g->red_buf, (size_t)nt * sizeof(double));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->e_red_done, hipEventDisableTiming);
        if (e != hipSuccess) rc = fail_hip(e, "group allocation");
    }
    if (rc != ARMON_OK) {
        armon_hip_mgpu_destroy(g);
        return rc;
    }
    *out = g;
    return ARMON_OK;
}

int armon_hip_mgpu_unique_id(void* id)
{
    ARMON_REQUIRE(id, "id is NULL");
    int rc = load_rccl();
    if (rc != ARMON_OK) return rc;
    ncclUniqueId* ids = static_cast<ncclUniqueId*>(id);
    ARMON_RCCL_TRY(g_rccl.GetUniqueId(&ids[0]));
    ARMON_RCCL_TRY(g_rccl.GetUniqueId(&ids[1]));
    return ARMON_OK;
}

// One process per GPU, in two steps so that the host can agree on readiness in between: _prepare_rank does everything that
// is LOCAL to this process (RCCL symbols, context, streams, events, scratch) and can fail on one rank alone; _connect is the
// collective part (ncclCommInitRank of the two communicators) and must be entered by every rank or by none — a rank that
// skipped it after a local failure would leave the others blocked in it for ever.
int armon_hip_mgpu_prepare_rank(int px, int py, int rank, int device_id, void* stream, armon_mgpu** out)
{
    ARMON_REQUIRE(out, "group out pointer is NULL");
    *out = nullptr;
    ARMON_REQUIRE(px >= 1 && py >= 1 && rank >= 0 && rank < px * py, "invalid rank %d of a %d x %d tile grid", rank, px, py);
    int rc = load_rccl();
    if (rc != ARMON_OK) return rc;
    armon_mgpu* g = new armon_mgpu();
    g->px = px;
    g->py = py;
    g->pack_on_compute = pack_on_compute_env();
    g->rccl = true;
    g->rank = rank;
    g->device = device_id;
    g->tiles.resize(1);
    tile_t& t = g->tiles[0];
    set_topology(t, rank, px, py, g->periodic);
    t.device = device_id;
    rc = armon_hip_init(device_id, stream, &t.ctx);
    if (rc == ARMON_OK) rc = make_tile_resources(t, true);
    if (rc == ARMON_OK) {
        hipError_t e = hipMalloc((void**)&g->red_scratch, 16 * sizeof(double));
        if (e == hipSuccess) e = hipHostMalloc((void**)&g->red_scratch_host, 16 * sizeof(double), hipHostMallocDefault);
        if (e != hipSuccess) rc = fail_hip(e, "group allocation");
    }
    if (rc != ARMON_OK) {
        armon_hip_mgpu_destroy(g);
        return rc;
    }
    *out = g;
    return ARMON_OK;
}

int armon_hip_mgpu_connect(armon_mgpu* g, const void* id)
{
    ARMON_REQUIRE(g && id, "NULL argument");
    ARMON_REQUIRE(g->rccl && !g->comm_halo && !g->comm_red, "not a prepared, unconnected rank group");
    ARMON_HIP_TRY(hipSetDevice(g->device));
    const ncclUniqueId* ids = static_cast<const ncclUniqueId*>(id);
    ncclResult_t r = g_rccl.CommInitRank(&g->comm_halo, g->px * g->py, ids[0], g->rank);
    if (r == ncclSuccess) r = g_rccl.CommInitRank(&g->comm_red, g->px * g->py, ids[1], g->rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank: %s", g_rccl.GetErrorString(r));
        return ARMON_ERR_HIP;
    }
    return ARMON_OK;
}

int armon_hip_mgpu_init_rank(int px, int py, int rank, int device_id, void* stream, const void* id, armon_mgpu** out)
{
    ARMON_REQUIRE(out, "group out pointer is NULL");
    *out = nullptr;
    ARMON_REQUIRE(id, "id is NULL");
    armon_mgpu* g = nullptr;
    int rc = armon_hip_mgpu_prepare_rank(px, py, rank, device_id, stream, &g);
    if (rc != ARMON_OK) return rc;
    rc = armon_hip_mgpu_connect(g, id);
    if (rc != ARMON_OK) {
        armon_hip_mgpu_destroy(g);
        return rc;
    }
    *out = g;
    return ARMON_OK;
}

int armon_hip_mgpu_destroy(armon_mgpu* g)
{
    if (!g) return ARMON_OK;
    delete g->pool;                              // joins the tile threads
    g->pool = nullptr;
    if (g->timing > 0 && g->n_cycles > 0) {
        fprintf(stderr, "armon_mgpu host time of tile 0's steps, microseconds per cycle over %ld cycles:", g->n_cycles);
        for (int i = 0; i < 8; i++) fprintf(stderr, " %s %.1f;", kHostTimerNames[i], g->t_host[i] / (double)g->n_cycles);
        fprintf(stderr, "\n");
    }
    for (tile_t& t : g->tiles) {
        (void)hipSetDevice(t.device);
        if (t.ctx && t.ctx->stream) (void)hipStreamSynchronize(t.ctx->stream);
        if (t.xfer) (void)hipStreamSynchronize(t.xfer);
    }
    if (g->comm_halo) (void)g_rccl.CommDestroy(g->comm_halo);
    if (g->comm_red) (void)g_rccl.CommDestroy(g->comm_red);
    for (tile_t& t : g->tiles) {
        (void)hipSetDevice(t.device);
        for (int s = 0; s < kSides; s++) {
            if (t.send[s]) (void)hipFree(t.send[s]);
            if (t.recv[s]) (void)hipFree(t.recv[s]);
            if (t.e_pack[s]) (void)hipEventDestroy(t.e_pack[s]);
            if (t.e_recv[s]) (void)hipEventDestroy(t.e_recv[s]);
            if (t.e_unpack[s]) (void)hipEventDestroy(t.e_unpack[s]);
        }
        if (t.e_red) (void)hipEventDestroy(t.e_red);
        if (t.e_state) (void)hipEventDestroy(t.e_state);
        if (t.e_intdt) (void)hipEventDestroy(t.e_intdt);
        if (t.e_dtdone) (void)hipEventDestroy(t.e_dtdone);
        if (t.e_edge) (void)hipEventDestroy(t.e_edge);
        if (t.edge_dt) (void)hipFree(t.edge_dt);
        if (t.edge) (void)armon_hip_destroy(t.edge);          // before the stream it borrows
        if (t.xfer) (void)hipStreamDestroy(t.xfer);
        if (t.ctx && t.owns_ctx) (void)armon_hip_destroy(t.ctx);
    }
    if (g->red_buf) (void)hipFree(g->red_buf);
    if (g->e_red_done) (void)hipEventDestroy(g->e_red_done);
    if (g->red_scratch) (void)hipFree(g->red_scratch);
    if (g->red_scratch_host) (void)hipHostFree(g->red_scratch_host);
    delete g;
    return ARMON_OK;
}

int armon_hip_mgpu_set_chaos(armon_mgpu* g, unsigned max_delay_us, uint64_t seed)
{
    ARMON_REQUIRE(g, "NULL argument");
    ARMON_REQUIRE(max_delay_us <= 20000, "delays above 20 ms are not a test any more");
    g->chaos_us = max_delay_us;
    for (tile_t& t : g->tiles) {
        t.chaos_rng = (seed ? seed : 0x9E3779B97F4A7C15ull) + 0xD1B54A32D192ED03ull * (uint64_t)t.rank;
        if (!t.chaos_rng) t.chaos_rng = 1;
    }
    return ARMON_OK;
}

// Test aids for the transport: a periodic topology (out-of-grid neighbours wrap around, per axis) makes a 1 x 1 rank group
// its own left/right/bottom/top neighbour, so that ONE GPU executes the real ncclSend/ncclRecv pairs of exchange_start; and
// force_peer_copy sends an in-process group's faces through hipMemcpyPeerAsync even when both tiles share a device. The
// reference's process grid is never periodic (ref src/parameters.jl:432: MPI.Cart_create without `periodic`).
int armon_hip_mgpu_set_periodic(armon_mgpu* g, int periodic_x, int periodic_y)
{
    ARMON_REQUIRE(g, "NULL argument");
    for (const tile_t& t : g->tiles)
        for (int s = 0; s < kSides; s++) ARMON_REQUIRE(t.inflight[s] == 0, "a halo exchange is in flight");
    g->periodic[0] = periodic_x != 0;
    g->periodic[1] = periodic_y != 0;
    for (tile_t& t : g->tiles) set_topology(t, t.rank, g->px, g->py, g->periodic);
    return ARMON_OK;
}

int armon_hip_mgpu_force_peer_copy(armon_mgpu* g, int on)
{
    ARMON_REQUIRE(g, "NULL argument");
    ARMON_REQUIRE(!g->rccl, "peer copies are the transport of an in-process group (armon_hip_mgpu_init)");
    g->force_peer = on != 0;
    return ARMON_OK;
}

int armon_hip_mgpu_n_local(armon_mgpu* g) { return g ? (int)g->tiles.size() : 0; }

armon_ctx* armon_hip_mgpu_ctx(armon_mgpu* g, int local_tile)
{
    if (!g || local_tile < 0 || local_tile >= (int)g->tiles.size()) return nullptr;
    return g->tiles[local_tile].ctx;
}

int armon_hip_mgpu_tile_info(armon_mgpu* g, int local_tile, int* rank, int coords[2], int neighbours[4])
{
    ARMON_REQUIRE(g && local_tile >= 0 && local_tile < (int)g->tiles.size(), "invalid tile %d", local_tile);
    const tile_t& t = g->tiles[local_tile];
    if (rank) *rank = t.rank;
    if (coords) { coords[0] = t.cx; coords[1] = t.cy; }
    if (neighbours) for (int s = 0; s < kSides; s++) neighbours[s] = t.nb[s];
    return ARMON_OK;
}

int armon_hip_halo_exchange_start(armon_mgpu* g, int axis, const armon_halo_desc* tiles) { return exchange_start<double>(g, axis, tiles); }
int armon_hip_halo_exchange_finish(armon_mgpu* g, int axis, const armon_halo_desc* tiles) { return exchange_finish<double>(g, axis, tiles); }
int armon_hip_halo_exchange(armon_mgpu* g, int axis, const armon_halo_desc* tiles)
{
    int rc = exchange_start<double>(g, axis, tiles);
    return rc != ARMON_OK ? rc : exchange_finish<double>(g, axis, tiles);
}
int armon_hip_halo_exchange_start_f32(armon_mgpu* g, int axis, const armon_halo_desc* tiles) { return exchange_start<float>(g, axis, tiles); }
int armon_hip_halo_exchange_finish_f32(armon_mgpu* g, int axis, const armon_halo_desc* tiles) { return exchange_finish<float>(g, axis, tiles); }
int armon_hip_halo_exchange_f32(armon_mgpu* g, int axis, const armon_halo_desc* tiles)
{
    int rc = exchange_start<float>(g, axis, tiles);
    return rc != ARMON_OK ? rc : exchange_finish<float>(g, axis, tiles);
}

// ---- edge stream: unpack and boundary strips concurrent with the interior sweep -----------------------------------------
armon_ctx* armon_hip_mgpu_edge_ctx(armon_mgpu* g, int local_tile)
{
    if (!g || local_tile < 0 || local_tile >= (int)g->tiles.size()) return nullptr;
    return g->tiles[local_tile].edge;
}
void* armon_hip_mgpu_edge_dt(armon_mgpu* g, int local_tile)
{
    if (!g || local_tile < 0 || local_tile >= (int)g->tiles.size()) return nullptr;
    return g->tiles[local_tile].edge_dt;
}
int armon_hip_halo_exchange_finish_edge(armon_mgpu* g, int axis, const armon_halo_desc* tiles) { return exchange_finish<double>(g, axis, tiles, true); }
int armon_hip_halo_exchange_finish_edge_f32(armon_mgpu* g, int axis, const armon_halo_desc* tiles) { return exchange_finish<float>(g, axis, tiles, true); }
int armon_hip_mgpu_edge_join(armon_mgpu* g, double* const* dt_dev) { return edge_join<double>(g, dt_dev); }
int armon_hip_mgpu_edge_join_f32(armon_mgpu* g, float* const* dt_dev) { return edge_join<float>(g, dt_dev); }

int armon_hip_mgpu_cycle(armon_mgpu* g, const armon_cycle_plan* plan, const armon_tile_cycle* tiles) { return mgpu_cycle<double>(g, plan, tiles); }
int armon_hip_mgpu_cycle_f32(armon_mgpu* g, const armon_cycle_plan* plan, const armon_tile_cycle_f32* tiles) { return mgpu_cycle<float>(g, plan, tiles); }
int armon_hip_mgpu_drain(armon_mgpu* g, const armon_tile_cycle* tiles) { return mgpu_drain<double>(g, tiles); }
int armon_hip_mgpu_drain_f32(armon_mgpu* g, const armon_tile_cycle_f32* tiles) { return mgpu_drain<float>(g, tiles); }
int armon_hip_mgpu_sync(armon_mgpu* g)
{
    ARMON_REQUIRE(g, "NULL argument");
    for (tile_t& t : g->tiles) {
        ARMON_HIP_TRY(hipSetDevice(t.device));
        ARMON_HIP_TRY(hipStreamSynchronize(t.ctx->stream));
        ARMON_HIP_TRY(hipStreamSynchronize(t.xfer));
    }
    return ARMON_OK;
}

int armon_hip_mgpu_set_threads(armon_mgpu* g, int on)
{
    ARMON_REQUIRE(g, "NULL argument");
    g->use_threads = on < 0 ? -1 : (on != 0);
    return ARMON_OK;
}

int armon_hip_dt_allreduce(armon_mgpu* g, double* const* dt_dev) { return dt_allreduce<double>(g, dt_dev); }
int armon_hip_dt_allreduce_f32(armon_mgpu* g, float* const* dt_dev) { return dt_allreduce<float>(g, dt_dev); }

// Host-value all-reduce over the PROCESSES of the group (diagnostics: the conservation sums of ref
// src/reductions.jl:317-320, Allreduce(SUM)); the caller combines its own local tiles first. op: 0 = sum, 1 = min.
// Synchronous. A no-op for an in-process group (one process owns every tile).
int armon_hip_mgpu_allreduce_host(armon_mgpu* g, int op, int count, double* values)
{
    ARMON_REQUIRE(g && values && count >= 1 && count <= 16, "invalid argument (count = %d, 1..16)", count);
    ARMON_REQUIRE(op == 0 || op == 1, "unknown reduction %d", op);
    if (!g->rccl) return ARMON_OK;
    tile_t& t = g->tiles[0];
    ARMON_HIP_TRY(hipSetDevice(t.device));
    for (int k = 0; k < count; k++) g->red_scratch_host[k] = values[k];
    ARMON_HIP_TRY(hipMemcpyAsync(g->red_scratch, g->red_scratch_host, count * sizeof(double), hipMemcpyHostToDevice, t.ctx->stream));
    ARMON_RCCL_TRY(g_rccl.AllReduce(g->red_scratch, g->red_scratch, (size_t)count, ncclDouble, op == 0 ? ncclSum : ncclMin,
                                    g->comm_red, t.ctx->stream));
    ARMON_HIP_TRY(hipMemcpyAsync(g->red_scratch_host, g->red_scratch, count * sizeof(double), hipMemcpyDeviceToHost, t.ctx->stream));
    ARMON_HIP_TRY(hipStreamSynchronize(t.ctx->stream));
    for (int k = 0; k < count; k++) values[k] = g->red_scratch_host[k];
    return ARMON_OK;
}

}  // extern "C"
