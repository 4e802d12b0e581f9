// fp64 MFMA (v_mfma_f64_16x16x4_f64) issue rate, dependent latency and clock on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* cyc, int iters) {
    v4d acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}
// same amount of work as fp64 VALU FMAs (64 lanes x 1 fma per instr = 128 flop)
__global__ __launch_bounds__(256) void kv(double* out, unsigned long long* cyc, int iters) {
    double acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = c;
    double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = __builtin_fma(acc[c], a, b);
    }
    double s = 0;
    for (int c = 0; c < 8; ++c) s += acc[c];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int CH> void run(int grid, int iters) {
    double* out; unsigned long long* cyc; hipMalloc(&out, grid * 256 * 8); hipMalloc(&cyc, grid * 16);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    double n = (double)iters * CH;
    printf("mfma f64 chains=%d grid=%d: %.1f shader-cycles/MFMA, %.1f ns/MFMA -> clock %.2f GHz, chip %.1f TF\n", CH, grid,
           h[0] / n, h[1] * 10.0 / n, (double)h[0] / (h[1] * 10.0), grid * 4 * n * 2048.0 / (h[1] * 10.0) / 1e3);
}
// MFMA and VALU FMAs interleaved in the same wave
__global__ __launch_bounds__(256) void kmix(double* out, unsigned long long* cyc, int iters, int nv) {
    v4d acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    double va[8];
    for (int c = 0; c < 8; ++c) va[c] = c;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
        if (nv >= 8) {
#pragma unroll
            for (int c = 0; c < 8; ++c) va[c] = __builtin_fma(va[c], b, a);
        }
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[1], 0, 0, 0);
        if (nv >= 16) {
#pragma unroll
            for (int c = 0; c < 8; ++c) va[c] = __builtin_fma(va[c], b, a);
        }
    }
    double s = acc[0][0] + acc[1][1];
    for (int c = 0; c < 8; ++c) s += va[c];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}
void runv(int grid) {
    double* out; unsigned long long* cyc; hipMalloc(&out, grid * 256 * 8); hipMalloc(&cyc, grid * 16);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kv, dim3(grid), dim3(256), 0, 0, out, cyc, 2000);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    double n = 2000.0 * 8;
    printf("valu fma f64 grid=%d: %.1f cycles/instr/wave, chip %.1f TF\n", grid, h[0] / n, grid * 4 * n * 128.0 / (h[1] * 10.0) / 1e3);
}
void runmix(int grid, int nv) {
    double* out; unsigned long long* cyc; hipMalloc(&out, grid * 256 * 8); hipMalloc(&cyc, grid * 16);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kmix, dim3(grid), dim3(256), 0, 0, out, cyc, 1000, nv);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("mix grid=%d valu/iter=%d: %.1f cycles per (2 MFMA + %d FMA) iter; MFMA %.1f TF + VALU %.1f TF\n", grid, nv, h[0] / 1000.0, nv,
           grid * 4 * 2000.0 * 2048.0 / (h[1] * 10.0) / 1e3, grid * 4 * 1000.0 * nv * 128.0 / (h[1] * 10.0) / 1e3);
}
int main() {
    runv(256); runv(512); runv(1024); runv(2048);
    run<4>(1024, 500); run<2>(2048, 300);
    runmix(256, 0); runmix(256, 8); runmix(256, 16); runmix(512, 16); runmix(1024, 16);
    run<1>(256, 2000); run<2>(256, 2000); run<4>(256, 1000); run<8>(256, 500); run<2>(1, 2000); run<4>(512, 1000);
    double* out; unsigned long long* cyc; hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 256 * 16);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kv, dim3(256), dim3(256), 0, 0, out, cyc, 2000);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    double n = 2000.0 * 8;
    printf("valu fma f64 (8 chains): %.1f cycles/instr, %.2f ns -> chip %.1f TF (1 wave/SIMD)\n", h[0] / n, h[1] * 10.0 / n,
           256 * 4 * n * 128.0 / (h[1] * 10.0) / 1e3);
    return 0;
}
