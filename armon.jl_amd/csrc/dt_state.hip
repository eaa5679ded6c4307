// dt_state.hip — the reference's GlobalTimeStep (ref src/solver_state.jl:30-166) as device-resident state, and the replay
// of a captured cycle (hipGraph).
//
// The reference's time loop (ref src/solver.jl:323-403) reads two scalars on the host every cycle: the reduced CFL step
// (update_dt!, :102-142) and, through it, the time that decides whether to go on (:350). On the device path a cycle is two
// to three kernel launches, so for small grids those host round trips and the launch calls themselves are the cycle time.
// Here update_dt! + next_cycle! + the exit test run as ONE one-thread kernel after the last sweep of a cycle, the sweeps
// read their time step from the state (armon_sweep_desc::dt_state), and a whole cycle can be captured once and replayed
// with a single call per cycle; the host looks at the state every few cycles, asynchronously.
#include "common.hpp"
#include "dt_state.hpp"

#include <cmath>

using namespace armon;

struct armon_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    armon_ctx* ctx = nullptr;        // the context whose scratch the captured kernels point into (NULL once it is gone)
    armon_graph* next = nullptr;     // the context's list of live graphs
};

// A context destroyed BEFORE its graphs (finalizers of a garbage-collected host run in any order): the executable graphs go
// with it — their kernels point into its scratch —, the handles the host still holds become empty shells that
// armon_hip_graph_launch refuses and armon_hip_graph_destroy frees without touching the context.
void armon::disown_graphs(armon_ctx* ctx)
{
    for (armon_graph* g = ctx->graphs; g;) {
        armon_graph* next = g->next;
        if (g->exec) (void)hipGraphExecDestroy(g->exec);
        if (g->graph) (void)hipGraphDestroy(g->graph);
        g->exec = nullptr;
        g->graph = nullptr;
        g->ctx = nullptr;
        g->next = nullptr;
        g = next;
    }
    ctx->graphs = nullptr;
    ctx->live_graphs = 0;
}

namespace {

template <typename T>
__global__ void k_dt_state_step(armon_dt_state* __restrict__ st, const T* __restrict__ L_new, T cfl, T maxtime,
                                int64_t maxcycle, int cst_dt, T Dt)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    dt_state_step<T>(st, cst_dt ? T(0) : L_new[0], cfl, maxtime, maxcycle, cst_dt, Dt);
}

template <typename T>
int step_impl(armon_ctx* ctx, armon_dt_state* st, const T* L_new, double cfl, double maxtime, int64_t maxcycle, int cst_dt,
              double Dt)
{
    ARMON_REQUIRE(ctx && st, "NULL argument");
    ARMON_REQUIRE(cst_dt || L_new, "the reduced CFL step (dt_cfl_out of the cycle's last sweep) is needed unless cst_dt");
    hipLaunchKernelGGL(k_dt_state_step<T>, dim3(1), dim3(1), 0, ctx->stream, st, L_new, (T)cfl, (T)maxtime, maxcycle, cst_dt, (T)Dt);
    return check_launch("dt_state_step");
}

}  // namespace

extern "C" {

int armon_hip_dt_state_step(armon_ctx* ctx, armon_dt_state* st, const double* L_new, double cfl, double maxtime,
                            int64_t maxcycle, int cst_dt, double Dt)
{
    return step_impl<double>(ctx, st, L_new, cfl, maxtime, maxcycle, cst_dt, Dt);
}

int armon_hip_dt_state_step_f32(armon_ctx* ctx, armon_dt_state* st, const float* L_new, double cfl, double maxtime,
                                int64_t maxcycle, int cst_dt, double Dt)
{
    return step_impl<float>(ctx, st, L_new, cfl, maxtime, maxcycle, cst_dt, Dt);
}

int armon_hip_graph_begin(armon_ctx* ctx)
{
    ARMON_REQUIRE(ctx, "ctx is NULL");
    // Capture needs a stream of the library's own: the legacy null stream cannot be captured, and capturing a stream the
    // caller handed in (armon_hip_init with a stream) would swallow the caller's own work on it.
    ARMON_REQUIRE(ctx->stream != nullptr && ctx->owns_stream, "graph capture needs a context that owns its stream "
                  "(armon_hip_init with stream = NULL creates one)");
    ARMON_REQUIRE(!ctx->capturing, "a capture is already in progress on this context");
    ARMON_HIP_TRY(hipSetDevice(ctx->device));
    ARMON_HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    return ARMON_OK;
}

int armon_hip_graph_end(armon_ctx* ctx, armon_graph** out)
{
    ARMON_REQUIRE(ctx && out, "NULL argument");
    *out = nullptr;
    hipGraph_t graph = nullptr;
    ctx->capturing = false;           // whatever EndCapture returns, the stream is out of capture mode afterwards
    ARMON_HIP_TRY(hipStreamEndCapture(ctx->stream, &graph));
    ARMON_REQUIRE(graph, "nothing was captured");
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(graph);
        return fail_hip(e, "hipGraphInstantiate");
    }
    armon_graph* g = new armon_graph();
    g->graph = graph;
    g->exec = exec;
    g->ctx = ctx;
    g->next = ctx->graphs;
    ctx->graphs = g;
    ctx->live_graphs++;
    *out = g;
    return ARMON_OK;
}

int armon_hip_graph_launch(armon_ctx* ctx, armon_graph* g)
{
    ARMON_REQUIRE(ctx && g, "NULL argument");
    ARMON_REQUIRE(g->exec && g->ctx == ctx, "this graph was captured on another context, or its context has been destroyed");
    ARMON_HIP_TRY(hipGraphLaunch(g->exec, ctx->stream));
    return ARMON_OK;
}

int armon_hip_graph_destroy(armon_graph* g)
{
    if (!g) return ARMON_OK;
    if (g->ctx) {
        for (armon_graph** p = &g->ctx->graphs; *p; p = &(*p)->next)
            if (*p == g) {
                *p = g->next;
                break;
            }
        if (g->ctx->live_graphs > 0) g->ctx->live_graphs--;
    }
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return ARMON_OK;
}

}  // extern "C"
