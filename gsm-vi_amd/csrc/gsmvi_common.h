// Shared device helpers for the gfx950 GSM/BaM kernels (fp64, wave64, MFMA f64 16x16x4).
#pragma once
#include <hip/hip_runtime.h>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// D(16x16) += A(16x4) * B(4x16).  Lane l holds A[row = l&15][k = l>>4], B[k = l>>4][col = l&15];
// result register r of lane l is D[row = (l>>4) + 4r][col = l&15]   (f64 layout, not the f32 one).
#define GSMVI_MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

#define GSMVI_WAVE 64
#define GSMVI_STAMP_WG 512            // workgroups per kernel slot of the timeline diagnostic (8 words each)
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

// One DPP-permuted copy of a double (two 32-bit DPP moves; CTRL is a DPP control word: quad_perm 0x00-0xFF,
// row_half_mirror 0x141, row_mirror 0x140).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// Sum over the 16 lanes of each DPP row (lanes 16k .. 16k+15), result in every lane of the row; fixed order.
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_f64<0xB1>(v);            // quad_perm [1,0,3,2]: neighbour
    v += dpp_f64<0x4E>(v);            // quad_perm [2,3,0,1]: other pair of the quad
    v += dpp_f64<0x141>(v);           // row_half_mirror: the other quad of the half row
    v += dpp_f64<0x140>(v);           // row_mirror: the other half row
    return v;
}
// Sum over the 64 lanes of the wave, result in every lane; fixed order (rows 0..3 added left to right).  Pure VALU + four
// v_readlane pairs: the ds_bpermute butterfly this replaces put six LDS-pipeline round trips on the dependent chain.
__device__ __forceinline__ double wave_sum(double v) {
    v = row16_sum(v);
    const unsigned long long u = __double_as_longlong(v);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)(u & 0xffffffffu), 16 * k);
        const unsigned hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), 16 * k);
        r[k] = __longlong_as_double(((unsigned long long)hi << 32) | lo);
    }
    return ((r[0] + r[1]) + r[2]) + r[3];
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
