// Calibration microbenchmarks for gfx950: what does a tiny kernel cost, as rocprofv3 sees it?
//   hipcc -O3 --offload-arch=gfx950 scripts/microbench.hip -o scripts/microbench && rocprofv3 --kernel-trace --stats ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ void k_empty() {}
__global__ void k_one_load(const double* __restrict__ a, double* __restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    o[i] = a[i] + 1.0;
}
// copy n doubles with 16-B accesses, one v2d per thread per step, `per` steps per thread (all issued first)
template <int PER>
__global__ __launch_bounds__(256) void k_copy(const v2d* __restrict__ a, v2d* __restrict__ o) {
    v2d t[PER];
    const size_t base = (size_t)blockIdx.x * 256 * PER + threadIdx.x;
#pragma unroll
    for (int p = 0; p < PER; ++p) t[p] = a[base + (size_t)p * 256];
#pragma unroll
    for (int p = 0; p < PER; ++p) o[base + (size_t)p * 256] = t[p];
}
// read-only reduce: every thread loads PER v2d and writes one double per block
template <int PER>
__global__ __launch_bounds__(256) void k_read(const v2d* __restrict__ a, double* __restrict__ o) {
    v2d t[PER];
    const size_t base = (size_t)blockIdx.x * 256 * PER + threadIdx.x;
#pragma unroll
    for (int p = 0; p < PER; ++p) t[p] = a[base + (size_t)p * 256];
    double s = 0;
#pragma unroll
    for (int p = 0; p < PER; ++p) s += t[p].x + t[p].y;
    if (s == 12345.678) o[blockIdx.x] = s;
}

int main() {
    const size_t N = (size_t)1 << 20;            // 8 MB of doubles
    const int NBUF = 24;                         // ring > 256 MiB (in+out)
    std::vector<double*> in(NBUF), out(NBUF);
    for (int i = 0; i < NBUF; ++i) {
        hipMalloc(&in[i], N * 8);
        hipMalloc(&out[i], N * 8);
        hipMemset(in[i], 0, N * 8);
        hipMemset(out[i], 0, N * 8);
    }
    hipDeviceSynchronize();
    for (int rep = 0; rep < 40; ++rep) {
        const int i = rep % NBUF;
        hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0);
        hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, 0);
        hipLaunchKernelGGL(k_one_load, dim3(256), dim3(256), 0, 0, in[i], out[i]);
        hipLaunchKernelGGL(k_copy<8>, dim3(N / 2 / 256 / 8), dim3(256), 0, 0, (const v2d*)in[i], (v2d*)out[i]);   // 256 WGs
        hipLaunchKernelGGL(k_copy<4>, dim3(N / 2 / 256 / 4), dim3(256), 0, 0, (const v2d*)in[i], (v2d*)out[i]);   // 512 WGs
        hipLaunchKernelGGL(k_copy<2>, dim3(N / 2 / 256 / 2), dim3(256), 0, 0, (const v2d*)in[i], (v2d*)out[i]);   // 1024 WGs
        hipLaunchKernelGGL(k_copy<1>, dim3(N / 2 / 256 / 1), dim3(256), 0, 0, (const v2d*)in[i], (v2d*)out[i]);   // 2048 WGs
        hipLaunchKernelGGL(k_read<8>, dim3(N / 2 / 256 / 8), dim3(256), 0, 0, (const v2d*)in[i], out[i]);
        hipLaunchKernelGGL(k_read<2>, dim3(N / 2 / 256 / 2), dim3(256), 0, 0, (const v2d*)in[i], out[i]);
    }
    hipDeviceSynchronize();
    // cache-resident variants (same buffer every time)
    for (int rep = 0; rep < 40; ++rep) {
        hipLaunchKernelGGL(k_copy<4>, dim3(N / 2 / 256 / 4 + 0), dim3(256), 0, 0, (const v2d*)in[0], (v2d*)out[0]);
        hipLaunchKernelGGL(k_read<4>, dim3(N / 2 / 256 / 4), dim3(256), 0, 0, (const v2d*)in[0], out[1]);
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
