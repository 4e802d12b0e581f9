// Accuracy of v_rcp_f64 / v_rsq_f64 on gfx950 (how many Newton steps do the pivot chains need?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // x sweeps [1, 4) finely plus a spread of exponents
    const double x = (1.0 + 3.0 * (double)i / (double)n) * ((i & 7) == 0 ? 1e10 : ((i & 7) == 1 ? 1e-10 : 1.0));
    const double y = __builtin_amdgcn_rcp(x);
    const double z = __builtin_amdgcn_rsq(x);
    out[3 * i] = fabs(y * x - 1.0);                       // relative error of rcp (computed with one rounding of slack)
    out[3 * i + 1] = fabs(z * z * x - 1.0) * 0.5;         // relative error of rsq
    // one quotient refinement: q0 = a*y, r = fma(-x, q0, a), q = fma(r, y, q0) against a/x
    const double a = 0.3 + (double)(i % 1000) * 1e-3;
    const double q0 = a * y, r = __builtin_fma(-x, q0, a), q = __builtin_fma(r, y, q0);
    out[3 * i + 2] = fabs(q - a / x) / (a / x);
}
int main() {
    const int n = 1 << 22;
    double* d; hipMalloc(&d, sizeof(double) * 3 * n);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n);
    double* h = new double[3 * n];
    hipMemcpy(h, d, sizeof(double) * 3 * n, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) { m0 = fmax(m0, h[3 * i]); m1 = fmax(m1, h[3 * i + 1]); m2 = fmax(m2, h[3 * i + 2]); }
    printf("max rel err: v_rcp_f64 %.3e (2^%.1f)  v_rsq_f64 %.3e (2^%.1f)  refined quotient %.3e (2^%.1f)\n", m0, log2(m0), m1,
           log2(m1), m2, log2(m2));
    return 0;
}
