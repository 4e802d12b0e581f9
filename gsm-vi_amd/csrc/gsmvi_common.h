// Shared device helpers for the gfx950 GSM/BaM kernels (fp64, wave64, MFMA f64 16x16x4).
#pragma once
#include <hip/hip_runtime.h>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// D(16x16) += A(16x4) * B(4x16).  Lane l holds A[row = l&15][k = l>>4], B[k = l>>4][col = l&15];
// result register r of lane l is D[row = (l>>4) + 4r][col = l&15]   (f64 layout, not the f32 one).
#define GSMVI_MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

#define GSMVI_WAVE 64
#define GSMVI_WG 256

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Deterministic block sum of up to NV values per thread for a 256-thread block.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* lds /* >= 4*NV doubles */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = wave_sum(v[k]);
        if (lane == 0) lds[w * NV + k] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = (lds[k] + lds[NV + k]) + (lds[2 * NV + k] + lds[3 * NV + k]);
    __syncthreads();
}
