// Probe: accuracy of v_rcp_f64 / v_rsq_f64 / v_sqrt_f64 and of Newton-refined forms on gfx950.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k(const double* x, double* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i];
    double r0 = __builtin_amdgcn_rcp(a);
    double e = __builtin_fma(-a, r0, 1.0);
    double r1 = __builtin_fma(r0, e, r0);
    e = __builtin_fma(-a, r1, 1.0);
    double r2 = __builtin_fma(r1, e, r1);
    double s0 = __builtin_amdgcn_rsq(a);
    // one Newton step for rsqrt: s1 = s0 * (1.5 - 0.5*a*s0*s0)  (fma form)
    double h = 0.5 * a * s0;
    double t = __builtin_fma(-h, s0, 0.5);
    double s1 = __builtin_fma(s0, t, s0);
    h = 0.5 * a * s1;
    t = __builtin_fma(-h, s1, 0.5);
    double s2 = __builtin_fma(s1, t, s1);
    double q0 = __builtin_amdgcn_sqrt(a);
    out[i] = r0; out[n + i] = r1; out[2 * n + i] = r2; out[3 * n + i] = s0; out[4 * n + i] = s1;
    out[5 * n + i] = s2; out[6 * n + i] = q0; out[7 * n + i] = a * s2;
}

int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), o(8 * n);
    srand(1);
    for (int i = 0; i < n; i++) x[i] = exp((rand() / (double)RAND_MAX - 0.5) * 40.0);
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 8 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 8 * n * 8, hipMemcpyDeviceToHost);
    const char* names[8] = {"rcp", "rcp+1NR", "rcp+2NR", "rsq", "rsq+1NR", "rsq+2NR", "sqrt_hw", "a*rsq2"};
    for (int k2 = 0; k2 < 8; k2++) {
        double worst = 0;
        for (int i = 0; i < n; i++) {
            long double ref = k2 < 3 ? 1.0L / x[i] : (k2 < 6 ? 1.0L / sqrtl(x[i]) : sqrtl(x[i]));
            double rel = fabs((double)((o[(size_t)k2 * n + i] - ref) / ref));
            if (rel > worst) worst = rel;
        }
        printf("%-8s max rel err %.3e (%.2f ulp)\n", names[k2], worst, worst / 1.11e-16);
    }
    return 0;
}
