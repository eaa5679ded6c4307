// multi_gpu.hip — the tile-decomposed grid natively: armon_hip_mgpu_init / armon_hip_halo_exchange / armon_hip_dt_allreduce.
//
// Replaces the reference's MPI halo path — start_exchange / finish_exchange (ref src/halo_exchange.jl:229-283: pack →
// wait → MPI.Start of persistent Send/Recv → MPI.Wait → unpack), block_ghost_exchange's side selection (ref
// :323-354: only the two sides ALONG the sweep axis, no corners), and the MPI_Iallreduce(MIN) of the time step
// (ref src/solver_state.jl:89-111, src/utils.jl:126-143) — with device-side ordering only:
//
//   compute stream of a tile : pack ─ e_pack ┐        interior sweep …        ┌ wait e_recv ─ unpack ─ e_unpack ─ strips
//   transfer stream of a tile:               └ wait ─ copy / ncclSend+ncclRecv ┴ e_recv
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

#include <cmath>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: every entry point is resolved with dlsym

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
    hipEvent_t e_red = nullptr;                  // dt scalar ready on the compute stream
    // "edge" work — unpack + the LAG-wide strips next to the remote sides — on the TRANSFER stream, concurrent with the
    // interior sweep on the compute stream (they read the same input state and write disjoint cells)
    armon_ctx* edge = nullptr;                   // a context on xfer (own reduction scratch)
    hipEvent_t e_edge = nullptr;                 // edge work of the current sweep done
    double* edge_dt = nullptr;                   // [2] device scalars: the strips' CFL steps (+inf when unused)
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
    uint64_t chaos_rng = 0x9E3779B97F4A7C15ull;
    // RCCL
    ncclComm_t comm_halo = nullptr, comm_red = nullptr;
    double* red_scratch = nullptr;               // device, [16]: host-value all-reduces
    double* red_scratch_host = nullptr;          // pinned
};

namespace {

__global__ void k_spin(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();               // constant-rate counter (100 MHz)
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// chaos mode: with probability 1/2, hold `stream` for up to chaos_us microseconds
void chaos(armon_mgpu* g, hipStream_t stream)
{
    if (!g->chaos_us) return;
    uint64_t& r = g->chaos_rng;
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

int make_tile_resources(tile_t& t)
{
    ARMON_HIP_TRY(hipSetDevice(t.device));
    ARMON_HIP_TRY(hipStreamCreateWithFlags(&t.xfer, hipStreamNonBlocking));
    for (int s = 0; s < kSides; s++) {
        ARMON_HIP_TRY(hipEventCreateWithFlags(&t.e_pack[s], hipEventDisableTiming));
        ARMON_HIP_TRY(hipEventCreateWithFlags(&t.e_recv[s], hipEventDisableTiming));
        ARMON_HIP_TRY(hipEventCreateWithFlags(&t.e_unpack[s], hipEventDisableTiming));
    }
    ARMON_HIP_TRY(hipEventCreateWithFlags(&t.e_red, hipEventDisableTiming));
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

template <typename T>
int exchange_start_impl(armon_mgpu* g, int axis, const armon_halo_desc* d, bool& touched)
{
    const int s0 = first_side(axis);
    const size_t nt = g->tiles.size();
    // 0. everything that can be refused is refused BEFORE anything is packed or marked: the descriptors (check_desc, by the
    //    caller), an exchange still in flight, the sizes the two ends of a face expect (in-process: both descriptors are
    //    here), and the face buffers (their growth drains the streams and allocates)
    for (size_t k = 0; k < nt; k++) {
        tile_t& t = g->tiles[k];
        for (int s = s0; s < s0 + 2; s++) {
            if (t.nb[s] < 0) continue;
            ARMON_REQUIRE(t.inflight[s] == 0, "halo exchange of side %d already started (finish it first)", s);
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
            rc = ensure_face_buffers(g, t, s, bytes);
            if (rc != ARMON_OK) return rc;
        }
    }
    // 1. pack every remote face of every local tile on its compute stream
    for (size_t k = 0; k < nt; k++) {
        tile_t& t = g->tiles[k];
        for (int s = s0; s < s0 + 2; s++) {
            if (t.nb[s] < 0) continue;
            armon_range border;
            int64_t face;
            int rc = armon_hip_halo_ranges(d[k].nx, d[k].ny, d[k].nghost, s, &border, nullptr, &face);
            if (rc != ARMON_OK) return rc;
            const size_t bytes = (size_t)face * d[k].nghost * d[k].nvars * sizeof(T);
            ARMON_HIP_TRY(hipSetDevice(t.device));
            // the previous message of this side must have left send[s] (in-process: the neighbour's copy event; RCCL:
            // our own transfer stream's event, which finish already made the compute stream wait for)
            if (!g->rccl) {
                tile_t& n = g->tiles[t.nb[s]];
                if (n.rec_recv[opposite(s)]) ARMON_HIP_TRY(hipStreamWaitEvent(t.ctx->stream, n.e_recv[opposite(s)], 0));
            }
            chaos(g, t.ctx->stream);
            touched = true;
            rc = pack<T>(t.ctx, border, d[k].nghost, face, static_cast<T*>(t.send[s]), d[k].nvars,
                         reinterpret_cast<const T* const*>(d[k].vars));
            if (rc != ARMON_OK) return rc;
            ARMON_HIP_TRY(hipEventRecord(t.e_pack[s], t.ctx->stream));
            t.inflight[s] = bytes;
        }
    }
    // 2. move the faces on the transfer streams
    for (size_t k = 0; k < nt; k++) {
        tile_t& t = g->tiles[k];
        bool any = false;
        for (int s = s0; s < s0 + 2; s++) any = any || t.inflight[s] != 0;
        if (!any) continue;
        ARMON_HIP_TRY(hipSetDevice(t.device));
        chaos(g, t.xfer);                         // late faces
        chaos(g, t.ctx->stream);                  // or a late interior: the edge work then runs ahead of it
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
    }
    return ARMON_OK;
}

template <typename T>
int exchange_start(armon_mgpu* g, int axis, const armon_halo_desc* d)
{
    int rc = check_desc(g, axis, d);
    if (rc != ARMON_OK) return rc;
    bool touched = false;
    rc = exchange_start_impl<T>(g, axis, d, touched);
    if (rc != ARMON_OK && touched) abandon_start(g, first_side(axis));     // the message of `rc` stays in last_error
    return rc;
}

template <typename T>
int exchange_finish(armon_mgpu* g, int axis, const armon_halo_desc* d, bool on_edge = false)
{
    int rc = check_desc(g, axis, d);
    if (rc != ARMON_OK) return rc;
    const int s0 = first_side(axis);
    for (size_t k = 0; k < g->tiles.size(); k++) {
        tile_t& t = g->tiles[k];
        for (int s = s0; s < s0 + 2; s++) {
            if (!t.inflight[s]) continue;
            armon_range ghost;
            int64_t face;
            rc = armon_hip_halo_ranges(d[k].nx, d[k].ny, d[k].nghost, s, nullptr, &ghost, &face);
            if (rc != ARMON_OK) return rc;
            ARMON_REQUIRE((size_t)face * d[k].nghost * d[k].nvars * sizeof(T) == t.inflight[s],
                          "tile %zu: finish does not match the exchange that was started", k);
            ARMON_HIP_TRY(hipSetDevice(t.device));
            // on the transfer stream the unpack simply follows the receive; on the compute stream it waits for it
            armon_ctx* c = on_edge ? t.edge : t.ctx;
            chaos(g, c->stream);
            if (!on_edge) ARMON_HIP_TRY(hipStreamWaitEvent(c->stream, t.e_recv[s], 0));
            rc = unpack<T>(c, ghost, d[k].nghost, face, static_cast<const T*>(t.recv[s]), d[k].nvars,
                           reinterpret_cast<T* const*>(d[k].vars));
            if (rc != ARMON_OK) return rc;
            ARMON_HIP_TRY(hipEventRecord(t.e_unpack[s], c->stream));
            t.rec_unpack[s] = true;
            t.inflight[s] = 0;
        }
    }
    return ARMON_OK;
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

template <typename T>
int edge_join(armon_mgpu* g, T* const* dt_dev)
{
    ARMON_REQUIRE(g, "NULL argument");
    for (size_t k = 0; k < g->tiles.size(); k++) {
        tile_t& t = g->tiles[k];
        ARMON_HIP_TRY(hipSetDevice(t.device));
        chaos(g, t.xfer);                         // strips that finish long after the interior
        ARMON_HIP_TRY(hipEventRecord(t.e_edge, t.xfer));
        ARMON_HIP_TRY(hipStreamWaitEvent(t.ctx->stream, t.e_edge, 0));
        if (dt_dev) {
            ARMON_REQUIRE(dt_dev[k], "dt_dev[%zu] is NULL", k);
            hipLaunchKernelGGL(k_fold_edge_dt<T>, dim3(1), dim3(64), 0, t.ctx->stream, dt_dev[k], reinterpret_cast<T*>(t.edge_dt));
            int rc = check_launch("fold_edge_dt");
            if (rc != ARMON_OK) return rc;
        }
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

template <typename T>
int dt_allreduce(armon_mgpu* g, T* const* dt_dev)
{
    ARMON_REQUIRE(g && dt_dev, "NULL argument");
    const size_t nt = g->tiles.size();
    for (size_t k = 0; k < nt; k++) ARMON_REQUIRE(dt_dev[k], "dt_dev[%zu] is NULL", k);
    if (g->rccl) {
        tile_t& t = g->tiles[0];
        ARMON_HIP_TRY(hipSetDevice(t.device));
        // ncclMin is free to drop a NaN operand; -inf survives any minimum and fails the host's validity check just as well
        // (ref src/solver_state.jl:123-124: `!isfinite(new_dt) || new_dt <= 0`)
        hipLaunchKernelGGL(k_nan_to_neg_inf<T>, dim3(1), dim3(1), 0, t.ctx->stream, dt_dev[0]);
        {
            int rc = check_launch("nan_to_neg_inf");
            if (rc != ARMON_OK) return rc;
        }
        ARMON_RCCL_TRY(g_rccl.AllReduce(dt_dev[0], dt_dev[0], 1, sizeof(T) == 8 ? ncclDouble : ncclFloat, ncclMin,
                                        g->comm_red, t.ctx->stream));
        return ARMON_OK;
    }
    if (nt == 1) return ARMON_OK;
    tile_t& root = g->tiles[0];
    for (size_t k = 0; k < nt; k++) chaos(g, g->tiles[k].ctx->stream);
    chaos(g, root.xfer);
    bool one_device = nt <= (size_t)kMaxDirect && !g->force_peer;    // force_peer: take the several-device path below
    for (size_t k = 0; k < nt; k++) one_device = one_device && g->tiles[k].device == root.device;
    if (one_device) {
        // one kernel on tile 0's transfer stream instead of 2·nt serialized copies (the chain between two cycles)
        ARMON_HIP_TRY(hipSetDevice(root.device));
        dt_ptrs d;
        for (size_t k = 0; k < nt; k++) {
            tile_t& t = g->tiles[k];
            d.p[k] = dt_dev[k];
            ARMON_HIP_TRY(hipEventRecord(t.e_red, t.ctx->stream));
            ARMON_HIP_TRY(hipStreamWaitEvent(root.xfer, t.e_red, 0));
        }
        hipLaunchKernelGGL(k_min_broadcast_direct<T>, dim3(1), dim3(64), 0, root.xfer, d, (int)nt);
        int rc = check_launch("min_broadcast_direct");
        if (rc != ARMON_OK) return rc;
        ARMON_HIP_TRY(hipEventRecord(g->e_red_done, root.xfer));
        for (size_t k = 0; k < nt; k++) ARMON_HIP_TRY(hipStreamWaitEvent(g->tiles[k].ctx->stream, g->e_red_done, 0));
        return ARMON_OK;
    }
    // several devices: gather on tile 0's device (its transfer stream), fold, scatter back; every compute stream waits
    // for its value only
    constexpr int stride = sizeof(double) / sizeof(T);       // one 8-byte slot per tile
    T* slots = reinterpret_cast<T*>(g->red_buf);
    for (size_t k = 0; k < nt; k++) {
        tile_t& t = g->tiles[k];
        ARMON_HIP_TRY(hipSetDevice(t.device));
        ARMON_HIP_TRY(hipEventRecord(t.e_red, t.ctx->stream));
    }
    ARMON_HIP_TRY(hipSetDevice(root.device));
    for (size_t k = 0; k < nt; k++) {
        tile_t& t = g->tiles[k];
        ARMON_HIP_TRY(hipStreamWaitEvent(root.xfer, t.e_red, 0));
        if (t.device == root.device && !g->force_peer)
            ARMON_HIP_TRY(hipMemcpyAsync(slots + k * stride, dt_dev[k], sizeof(T), hipMemcpyDeviceToDevice, root.xfer));
        else
            ARMON_HIP_TRY(hipMemcpyPeerAsync(slots + k * stride, root.device, dt_dev[k], t.device, sizeof(T), root.xfer));
    }
    hipLaunchKernelGGL(k_min_broadcast<T>, dim3(1), dim3(64), 0, root.xfer, slots, stride, (int)nt);
    {
        int rc = check_launch("min_broadcast");
        if (rc != ARMON_OK) return rc;
    }
    for (size_t k = 0; k < nt; k++) {
        tile_t& t = g->tiles[k];
        if (t.device == root.device && !g->force_peer)
            ARMON_HIP_TRY(hipMemcpyAsync(dt_dev[k], slots + k * stride, sizeof(T), hipMemcpyDeviceToDevice, root.xfer));
        else
            ARMON_HIP_TRY(hipMemcpyPeerAsync(dt_dev[k], t.device, slots + k * stride, root.device, sizeof(T), root.xfer));
    }
    ARMON_HIP_TRY(hipEventRecord(g->e_red_done, root.xfer));
    for (size_t k = 0; k < nt; k++) {
        tile_t& t = g->tiles[k];
        ARMON_HIP_TRY(hipSetDevice(t.device));
        ARMON_HIP_TRY(hipStreamWaitEvent(t.ctx->stream, g->e_red_done, 0));
    }
    return ARMON_OK;
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

int armon_hip_mgpu_init(int px, int py, const int* device_ids, armon_mgpu** out)
{
    ARMON_REQUIRE(out, "group out pointer is NULL");
    *out = nullptr;
    ARMON_REQUIRE(px >= 1 && py >= 1 && px * py <= 4096, "invalid tile grid %d x %d", px, py);
    const int nt = px * py;
    armon_mgpu* g = new armon_mgpu();
    g->px = px;
    g->py = py;
    g->tiles.resize(nt);
    int rc = ARMON_OK;
    for (int r = 0; r < nt && rc == ARMON_OK; r++) {
        tile_t& t = g->tiles[r];
        set_topology(t, r, px, py, g->periodic);
        t.device = device_ids ? device_ids[r] : 0;
        rc = armon_hip_init(t.device, nullptr, &t.ctx);
        if (rc == ARMON_OK) rc = make_tile_resources(t);
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
    g->rccl = true;
    g->rank = rank;
    g->device = device_id;
    g->tiles.resize(1);
    tile_t& t = g->tiles[0];
    set_topology(t, rank, px, py, g->periodic);
    t.device = device_id;
    rc = armon_hip_init(device_id, stream, &t.ctx);
    if (rc == ARMON_OK) rc = make_tile_resources(t);
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
    g->chaos_rng = seed ? seed : 0x9E3779B97F4A7C15ull;
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
