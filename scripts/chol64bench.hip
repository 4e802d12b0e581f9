// Phase timing of the in-LDS 64x64 Cholesky (gsm-vi_amd/csrc/gsmvi_chol64.h) on one workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
__device__ unsigned long long g_stamp[64];
#define CHOL_STAMP(k) do { if (threadIdx.x == 0) g_stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#include "../gsm-vi_amd/csrc/gsmvi_chol64.h"
__global__ __launch_bounds__(256) void k(const double* A, double* R, int* fail) {
    __shared__ double T[64 * TS];
    __shared__ double rinv[64];
    __shared__ int sf;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) T[(e >> 6) * TS + (e & 63)] = A[e];
    if (threadIdx.x < 64) rinv[threadIdx.x] = 1.0;
    __syncthreads();
    CHOL_STAMP(0);
    chol64_lds(T, rinv, 64, &sf);
    CHOL_STAMP(63);
    for (int e = threadIdx.x; e < 64 * 64; e += 256) R[e] = ((e & 63) >= (e >> 6)) ? T[(e >> 6) * TS + (e & 63)] : 0.0;
    if (threadIdx.x == 0) *fail = sf;
}
int main() {
    const int n = 64;
    std::vector<double> A(n * n), Rh(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = (i == j) ? 64.0 : 0.0; for (int k = 0; k < n; ++k) s += sin(0.1 * (i * 7 + k * 3 + 1)) * sin(0.1 * (j * 7 + k * 3 + 1)); A[i * n + j] = s; }
    double *dA, *dR; int* df; hipMalloc(&dA, n * n * 8); hipMalloc(&dR, n * n * 8); hipMalloc(&df, 4);
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    for (int r = 0; r < 50; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dA, dR, df);
    hipDeviceSynchronize();
    unsigned long long st[64]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof st);
    hipMemcpy(Rh.data(), dR, n * n * 8, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += Rh[k * n + i] * Rh[k * n + j]; err = fmax(err, fabs(s - A[i * n + j])); }
    printf("recon err %.2e, total %.2f us\n", err, (st[63] - st[0]) / 100.0);
#ifdef GSMVI_CHOL64_BLOCKED
    for (int kb = 0; kb < 4; ++kb)
        printf(" block %d: diag %.2f  rowsolve %.2f  trailing %.2f us\n", kb, (st[1 + 4 * kb + 1] - st[1 + 4 * kb]) / 100.0,
               (st[1 + 4 * kb + 2] - st[1 + 4 * kb + 1]) / 100.0, (st[1 + 4 * kb + 3] - st[1 + 4 * kb + 2]) / 100.0);
#else
    printf(" load %.2f  pivots %.2f  scale %.2f us\n", (st[1] - st[0]) / 100.0, (st[2] - st[1]) / 100.0, (st[3] - st[2]) / 100.0);
#endif
    return 0;
}
