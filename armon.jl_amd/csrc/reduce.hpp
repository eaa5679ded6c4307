// reduce.hpp — wavefront shuffle + LDS tree reductions shared by the reduction and sweep kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "physics.hpp"

namespace armon {
namespace red {

constexpr int kWave = 64;

struct op_min { template <typename T> __device__ static T id() { return T(INFINITY); } template <typename T> __device__ static T f(T a, T b) { return phys::mn(a, b); } };
// maximum of NON-NEGATIVE values, NaN sticks (phys::amax): only the dt/CFL reductions use it, and a NaN in one cell has to
// reach the host's validity check (ref src/solver_state.jl:123-124)
struct op_max { template <typename T> __device__ static T id() { return T(0.); } template <typename T> __device__ static T f(T a, T b) { return phys::amax(a, b); } };
struct op_sum { template <typename T> __device__ static T id() { return T(0.); } template <typename T> __device__ static T f(T a, T b) { return a + b; } };

template <typename OP, typename T>
__device__ __forceinline__ T wave_reduce(T v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v = OP::f(v, __shfl_down(v, off, kWave));
    return v;   // valid in lane 0
}

// Reduce across a workgroup of NWAVES waves; result valid in thread 0. `lds` holds NWAVES values.
template <typename OP, int NWAVES, typename T>
__device__ __forceinline__ T block_reduce(T v, T* lds, int tid)
{
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    v = wave_reduce<OP>(v);
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    T r = OP::template id<T>();
    if (tid == 0) {
#pragma unroll
        for (int w = 0; w < NWAVES; w++) r = OP::f(r, lds[w]);
    }
    __syncthreads();
    return r;
}

}  // namespace red
}  // namespace armon
