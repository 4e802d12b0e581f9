// The pivot chain of chol64_blk in isolation (one wave): cycles per pivot for variants of the dependent sequence.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ unsigned long long g_t[16];
__device__ __forceinline__ double rl(double v, int lane) {
    unsigned long long u = __double_as_longlong(v);
    unsigned lo = __builtin_amdgcn_readlane((unsigned)u, lane), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), lane);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double rcp2(double d) {
    double y = __builtin_amdgcn_rcp(d);
    y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
    y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
    return y;
}
__global__ void k(double* out, double x0, int n) {
    __shared__ double sh[256];
    double x = x0 + threadIdx.x * 1e-3, w = 0.3 + threadIdx.x * 1e-4;
    unsigned long long t0, t1;
    // A: full chain with readlane of the pivot
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const double d = rl(x, 3); const double t = w * rcp2(d); x = __builtin_fma(-w, t, x + 1.0); }
    }
    asm volatile("" :: "v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_t[0] = t1 - t0;
    // B: same chain, no readlane (pivot = own value)
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const double d = x; const double t = w * rcp2(d); x = __builtin_fma(-w, t, x + 1.0); }
    }
    asm volatile("" :: "v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_t[1] = t1 - t0;
    // C: chain + LDS write of the row and a dependent broadcast read used by an independent FMA
    double z = 1.0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sh[threadIdx.x] = x;
            const double d = rl(x, 3); const double t = w * rcp2(d); x = __builtin_fma(-w, t, x + 1.0);
            z = __builtin_fma(-sh[7], t, z);
        }
    }
    asm volatile("" :: "v"(x), "v"(z));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_t[2] = t1 - t0;
    // D: readlane pair alone, dependent through a v_mov (x = readlane(x) broadcast)
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { x = rl(x, 3) + 0.0; }
    }
    asm volatile("" :: "v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_t[3] = t1 - t0;
    // E: the Newton form with one step only
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const double d = x; double y = __builtin_amdgcn_rcp(d); y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y); x = __builtin_fma(-w, w * y, x + 1.0); }
    }
    asm volatile("" :: "v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_t[4] = t1 - t0;
    out[threadIdx.x] = x + z;
}
int main() {
    double* o; hipMalloc(&o, 64 * 8);
    const int n = 500;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, 50.0, n);
    hipDeviceSynchronize();
    unsigned long long t[16]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof t);
    const char* nm[] = {"A chain with readlane pivot", "B chain, no readlane", "C chain + LDS row write + broadcast read", "D readlane pair + add", "E chain, one Newton step, no readlane"};
    for (int i = 0; i < 5; ++i) printf("%-44s %.1f cycles per pivot\n", nm[i], (double)t[i] / (n * 4.0));
    return 0;
}
