// reduce.hpp — wavefront shuffle + LDS tree reductions shared by the reduction and sweep kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "physics.hpp"

namespace armon {
namespace red {

constexpr int kWave = 64;

struct op_min { __device__ static double id() { return INFINITY; } __device__ static double f(double a, double b) { return phys::mn(a, b); } };
struct op_max { __device__ static double id() { return 0.; } __device__ static double f(double a, double b) { return phys::mx(a, b); } };
struct op_sum { __device__ static double id() { return 0.; } __device__ static double f(double a, double b) { return a + b; } };

template <typename OP>
__device__ __forceinline__ double wave_reduce(double v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v = OP::f(v, __shfl_down(v, off, kWave));
    return v;   // valid in lane 0
}

// Reduce across a workgroup of NWAVES waves; result valid in thread 0. `lds` holds NWAVES doubles.
template <typename OP, int NWAVES>
__device__ __forceinline__ double block_reduce(double v, double* lds, int tid)
{
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    v = wave_reduce<OP>(v);
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = OP::id();
    if (tid == 0) {
#pragma unroll
        for (int w = 0; w < NWAVES; w++) r = OP::f(r, lds[w]);
    }
    __syncthreads();
    return r;
}

}  // namespace red
}  // namespace armon
