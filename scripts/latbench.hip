// Dependent-chain latencies on gfx950 (one wave): v_fma_f64, v_rcp_f64, v_rsq_f64, v_readlane round trip, ds_write->ds_read,
// v_mul_f64.  cycles per op from s_memtime.  build: hipcc --offload-arch=gfx950 -O3 -o scripts/latbench scripts/latbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ unsigned long long g_t[16];
__global__ void k(double* out, double x0, int n) {
    __shared__ double sh[128];
    double x = x0 + threadIdx.x * 1e-9, y = 1.0000001;
    unsigned long long t0, t1;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) { x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); }
    asm volatile("" :: "v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_t[0] = t1 - t0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) { x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); }
    asm volatile("" :: "v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_t[1] = t1 - t0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            unsigned long long u = __double_as_longlong(x);
            unsigned lo = __builtin_amdgcn_readlane((unsigned)u, 5), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), 5);
            double s = __longlong_as_double(((unsigned long long)hi << 32) | lo);
            x = x * s;   // one mul + readlane pair per step
        }
    }
    asm volatile("" :: "v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_t[2] = t1 - t0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { sh[threadIdx.x] = x; x = sh[threadIdx.x ^ 1] * 1.0000001; }
    }
    asm volatile("" :: "v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_t[3] = t1 - t0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) { x = x * y; x = x * y; x = x * y; x = x * y; }
    asm volatile("" :: "v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_t[4] = t1 - t0;
    // independent fma throughput (8 chains)
    double a[8]; for (int c = 0; c < 8; ++c) a[c] = x + c;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = __builtin_fma(a[c], y, 1e-9);
    }
    for (int c = 0; c < 8; ++c) asm volatile("" :: "v"(a[c]));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_t[5] = t1 - t0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) { x = __builtin_amdgcn_rsq(x); x = __builtin_amdgcn_rsq(x); x = __builtin_amdgcn_rsq(x); x = __builtin_amdgcn_rsq(x); }
    asm volatile("" :: "v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_t[6] = t1 - t0;
    float f = (float)x;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) { f = __builtin_fmaf(f, 1.0001f, 1e-9f); f = __builtin_fmaf(f, 1.0001f, 1e-9f); f = __builtin_fmaf(f, 1.0001f, 1e-9f); f = __builtin_fmaf(f, 1.0001f, 1e-9f); }
    asm volatile("" :: "v"(f));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_t[7] = t1 - t0;
    double s = x + f; for (int c = 0; c < 8; ++c) s += a[c];
    out[threadIdx.x] = s;
}
int main() {
    double* o; hipMalloc(&o, 64 * 8);
    const int n = 1000;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, 1.5, n);
    hipDeviceSynchronize();
    unsigned long long t[16]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof t);
    const char* nm[] = {"v_fma_f64 dependent", "v_rcp_f64 dependent", "readlane pair + v_mul_f64", "ds_write_b64 -> ds_read_b64 + mul", "v_mul_f64 dependent", "v_fma_f64 x8 independent (per fma)", "v_rsq_f64 dependent", "v_fma_f32 dependent"};
    for (int i = 0; i < 8; ++i) printf("%-40s %.1f cycles per op\n", nm[i], (double)t[i] / (n * (i == 5 ? 8.0 : 4.0)));
    return 0;
}
