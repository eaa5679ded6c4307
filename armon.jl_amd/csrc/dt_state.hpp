// dt_state.hpp — update_dt! + next_cycle! + the time loop's exit test on a device-resident armon_dt_state, for ONE thread
// (ref src/solver_state.jl:102-166, src/solver.jl:350). Shared by the stand-alone step kernel (dt_state.hip) and the fold
// of the fused dt reduction (fused_sweep_impl.hpp, armon_dt_state::auto_step).
#pragma once

#include "common.hpp"

namespace armon {

// All arithmetic in the run's precision T, like GlobalTimeStep{T}: the product cfl·L in T, the 1.05 cap in Float64
// (ref src/solver_state.jl:129: `convert(T, min(params.cfl * new_dt, 1.05 * previous_dt))`), time accumulated in T.
template <typename T>
__device__ __forceinline__ void dt_state_step(armon_dt_state* __restrict__ st, T L_new, T cfl, T maxtime, int64_t maxcycle,
                                              int cst_dt, T Dt)
{
    if (st->done) return;
    const T current = (T)st->current_dt;
    T next;
    if (cst_dt) {
        next = Dt;                                                       // ref next_cycle!, :150-153
    } else {
        const T L = (T)st->L_prev;
        if (!(L == L) || L - L != T(0) || L <= T(0)) {                   // !isfinite(L) || L <= 0   (ref :123-124)
            st->invalid = 1;
            st->invalid_cycle = st->cycle;
            st->invalid_value = (double)L;
            st->done = 1;
            return;
        }
        const double a = (double)(cfl * L), b = 1.05 * (double)current;
        next = (T)(b < a ? b : a);
    }
    const int64_t cycle = st->cycle + 1;
    const T time = (T)((T)st->time + current);
    st->cycle = cycle;
    st->time = (double)time;
    st->current_dt = (double)next;
    if (!cst_dt) st->L_prev = (double)L_new;
    // the time loop's exit test for the cycle that would come next (ref src/solver.jl:350), and whether that cycle is the
    // last one (its last sweep then materialises p: the reference's saved p is the EOS of the state before the last sweep)
    st->done = (time < maxtime && cycle < maxcycle) ? 0 : 1;
    st->emit_p = (cycle + 1 >= maxcycle || (T)(time + next) >= maxtime) ? 1 : 0;
}

}  // namespace armon
