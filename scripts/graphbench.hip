// What does one more kernel in a dependent chain cost inside a replayed hipGraph on gfx950?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_emptyB() {}
__global__ void k_lds(double* o) { __shared__ double l[4096]; l[threadIdx.x] = threadIdx.x; __syncthreads(); if (l[(threadIdx.x + 1) & 255] < 0) o[0] = 1; }
__global__ void k_one_load(const double* __restrict__ a, double* __restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    o[i] = a[i] + 1.0;
}
// 32 blocks x 1024 threads: load 7 values, block-reduce, thread 0 stores (the scalar stage's shape)
__global__ __launch_bounds__(1024) void k_reduce(const double* __restrict__ a, double* __restrict__ o) {
    __shared__ double l[16];
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) s += a[i + (size_t)k * 32768];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) l[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int k = 0; k < 16; ++k) t += l[k]; o[blockIdx.x] = sqrt(t * t + 1.0); }
}
template <typename F> double time_graph(F launch, int n, int reps) {
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch(st);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int r = 0; r < 3; ++r) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    auto t1 = std::chrono::high_resolution_clock::now();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * (double)n);
}
int main() {
    double *a, *o; hipMalloc(&a, 8 << 20); hipMalloc(&o, 8 << 20); hipMemset(a, 0, 8 << 20);
    printf("empty 256 WG      : %.2f us/kernel\n", time_graph([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); }, 200, 50));
    printf("empty 1 WG        : %.2f us/kernel\n", time_graph([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, 200, 50));
    printf("one_load 256 WG   : %.2f us/kernel\n", time_graph([&](hipStream_t s) { hipLaunchKernelGGL(k_one_load, dim3(256), dim3(256), 0, s, a, o); }, 200, 50));
    printf("reduce 32x1024    : %.2f us/kernel\n", time_graph([&](hipStream_t s) { hipLaunchKernelGGL(k_reduce, dim3(32), dim3(1024), 0, s, a, o); }, 200, 50));
    printf("reduce+one_load   : %.2f us/pair\n", 2 * time_graph([&](hipStream_t s) { hipLaunchKernelGGL(k_reduce, dim3(32), dim3(1024), 0, s, a, o); hipLaunchKernelGGL(k_one_load, dim3(256), dim3(256), 0, s, a, o); }, 100, 50) );
    printf("emptyA+emptyB     : %.2f us/pair\n", 2 * time_graph([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); hipLaunchKernelGGL(k_emptyB, dim3(256), dim3(256), 0, s); }, 100, 50));
    printf("empty256+empty1024: %.2f us/pair\n", 2 * time_graph([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); hipLaunchKernelGGL(k_empty, dim3(32), dim3(1024), 0, s); }, 100, 50));
    printf("one_load+one_load  : %.2f us/pair\n", 2 * time_graph([&](hipStream_t s) { hipLaunchKernelGGL(k_one_load, dim3(256), dim3(256), 0, s, a, o); hipLaunchKernelGGL(k_one_load, dim3(256), dim3(256), 0, s, o, a); }, 100, 50));
    printf("reduce+reduce(o,a) : %.2f us/pair\n", 2 * time_graph([&](hipStream_t s) { hipLaunchKernelGGL(k_reduce, dim3(32), dim3(1024), 0, s, a, o); hipLaunchKernelGGL(k_reduce, dim3(32), dim3(1024), 0, s, a, o + 4096); }, 100, 50));
    printf("reduce+empty       : %.2f us/pair\n", 2 * time_graph([&](hipStream_t s) { hipLaunchKernelGGL(k_reduce, dim3(32), dim3(1024), 0, s, a, o); hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); }, 100, 50));
    printf("one_load+empty     : %.2f us/pair\n", 2 * time_graph([&](hipStream_t s) { hipLaunchKernelGGL(k_one_load, dim3(256), dim3(256), 0, s, a, o); hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); }, 100, 50));
    printf("lds+empty          : %.2f us/pair\n", 2 * time_graph([&](hipStream_t s) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 0, s, o); hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); }, 100, 50));
    printf("reduce+one_load(b) : %.2f us/pair\n", 2 * time_graph([&](hipStream_t s) { hipLaunchKernelGGL(k_reduce, dim3(32), dim3(1024), 0, s, a, o); hipLaunchKernelGGL(k_one_load, dim3(256), dim3(256), 0, s, a + 500000, o + 500000); }, 100, 50));
    return 0;
}
