// Probe: semantics of DPP wave_shr:1 / wave_shl:1 on gfx950 (which lane feeds which).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double dpp(double x, bool shr) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    if (shr) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false); }
    else     { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false); }
    return __hiloint2double(hi, lo);
}
__global__ void k(double* out) {
    double x = 100.0 + threadIdx.x;
    out[threadIdx.x] = dpp(x, true);
    out[64 + threadIdx.x] = dpp(x, false);
}
int main() {
    double* d; double h[128];
    (void)hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_shr:1 lanes 0,1,2,15,16,31,32,63 -> %g %g %g %g %g %g %g %g\n", h[0], h[1], h[2], h[15], h[16], h[31], h[32], h[63]);
    printf("wave_shl:1 lanes 0,1,2,15,16,31,32,63 -> %g %g %g %g %g %g %g %g\n", h[64], h[65], h[66], h[79], h[80], h[95], h[96], h[127]);
    return 0;
}
