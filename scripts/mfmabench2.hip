// Whole-chip fp64 throughput by wall clock: MFMA 16x16x4 vs VALU FMA vs both, by waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NM, int NV>   // per iteration: NM MFMAs (NM independent chains), NV VALU FMAs (8 chains)
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    v4d acc[NM > 0 ? NM : 1];
    for (int c = 0; c < (NM > 0 ? NM : 1); ++c) acc[c] = (v4d){0, 0, 0, 0};
    double va[8];
    for (int c = 0; c < 8; ++c) va[c] = c + threadIdx.x;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < NM; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NV; ++c) va[c & 7] = __builtin_fma(va[c & 7], b, a);
    }
    double s = 0;
    for (int c = 0; c < (NM > 0 ? NM : 1); ++c) s += acc[c][0] + acc[c][3];
    for (int c = 0; c < 8; ++c) s += va[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NM, int NV> void run(int wps, int iters) {
    const int grid = 256 * wps;
    double* out; hipMalloc(&out, (size_t)grid * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * 4;
    printf("waves/SIMD=%d  MFMA/iter=%d VALU/iter=%2d : MFMA %6.1f TF  VALU %6.1f TF  (%.2f ms)\n", wps, NM, NV,
           waves * iters * NM * 2048.0 / (ms * 1e-3) / 1e12, waves * iters * NV * 128.0 / (ms * 1e-3) / 1e12, ms);
    hipFree(out);
}
int main() {
    const int it = 20000;
    run<4, 0>(1, it); run<4, 0>(2, it); run<4, 0>(4, it); run<4, 0>(8, it / 2);
    run<0, 16>(1, it); run<0, 16>(2, it); run<0, 16>(4, it); run<0, 16>(8, it / 2);
    run<2, 16>(1, it); run<2, 16>(2, it); run<2, 16>(4, it);
    run<2, 32>(2, it); run<2, 48>(2, it);
    return 0;
}
