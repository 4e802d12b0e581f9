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

// Broadcast of lane Q (0..3, compile-time) of every quad of lanes to the whole quad: two DPP moves (quad_perm), pure VALU --
// __shfl compiles to ds_bpermute_b32, an LDS-pipeline instruction with ~100 cycles of latency on a dependent chain.
template <int Q>
__device__ __forceinline__ double quad_bcast(double v) {
    constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);      // DPP quad_perm: [Q, Q, Q, Q]
    const unsigned long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), ctrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), ctrl, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
template <int>
__device__ __forceinline__ double quad_bcast_rt(double v, int q) {  // run-time lane (still DPP: four-way select)
    const double a = quad_bcast<0>(v), b = quad_bcast<1>(v), c = quad_bcast<2>(v), d = quad_bcast<3>(v);
    return q == 0 ? a : (q == 1 ? b : (q == 2 ? c : d));
}

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
