// Panel products for 64-row panels (B = 64: BASELINE config 5) in the regime where they are bound by the fp64 MFMA pipe,
// not by HBM.   Round 3.
//
// k_panel_fast / k_panel_t_fast (gsmvi_fast.hip, gsmvi_factor.hip) give a workgroup 16 columns of the output and let its
// eight waves split the contraction index: right for B <= 32, where the D x D operand's HBM stream is the bound and many
// narrow workgroups keep it busy.  At B = 64 the arithmetic intensity is 16 flop per byte of the D x D operand -- 2.1 GFLOP
// per product at D = 4096, 46 us at the 47 TF this chip sustains in fp64 MFMA, against 17 us for the stream -- and the
// narrow tiling pays for every 16 output columns with a re-staging of the whole 64-row left operand through LDS (256
// strips x 2 MB = 0.5 GB of L2 -> LDS traffic per product): 71-80 us measured.  Here a workgroup owns a 64 x 64 output tile:
// the left operand is re-staged once per 64 columns (a quarter of the traffic), each wave owns a 16 x 32 strip of the tile
// with the FULL contraction range of the workgroup (no cross-wave reduction), both operands of a 64-deep chunk pass
// through LDS once and are reused by four (left) / four (right) waves, and the next chunk's global loads are in flight
// during the MFMA chain of the current one.
//   normal     (TR = false): Pp[y][r][j] = sum_{i in K(y)} alpha (A[r][i] - shift[i]) M[i][j]      M: D x ncols
//   transposed (TR = true) : Pp[y][r][m] = sum_{i in K(y)} A[r][i] M[m][i]                          M: ncols x D
// D % 64 == 0, ncols % 64 == 0, even leading dimensions, 16-byte aligned bases (the callers' fast-path conditions).
// Same split-K slab layout as the narrow kernels, so finish passes and slab-summing consumers are unchanged.  The sum
// over the contraction index runs in a different order than in the narrow kernels (results agree to rounding, not bitwise);
// within this kernel the order is fixed, so results are run-to-run identical.
#include "gsmvi_common.h"
#include "gsmvi_ctx.h"
#include <hip/hip_ext.h>

template <bool TR, bool HAS_SHIFT>
__global__ __launch_bounds__(512) void k_panel_wide(int D, int nrows, const double* __restrict__ A, int lda,
                                                    const double* __restrict__ shift, double alpha,
                                                    const double* __restrict__ M, int ldm, double* __restrict__ Pp,
                                                    int kper, int ncols) {
    constexpr int LDA = 66;                        // [row][k]: lanes (row c, k-slot ks) read conflict-free
    constexpr int LDB = TR ? 66 : 80;              // TR: [m][k] like A;  normal: [k][64 columns], stride 16 mod 32 banks
    __shared__ __attribute__((aligned(16))) double As[64 * LDA];
    __shared__ __attribute__((aligned(16))) double Bs[64 * LDB];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int rb = w & 3, cp = w >> 2;             // wave: rows 16 rb .. +15, columns 32 cp .. +31 of the tile
    const int j0 = blockIdx.x * 64, r0 = blockIdx.z * 64;
    const int kbeg = blockIdx.y * kper;
    const int kend = (kbeg + kper < D) ? kbeg + kper : D;
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    v2d ga[4], gs[4], gb[4];
    auto fetch = [&](int k0) {                     // the 64 x 64 chunks of both operands: 4 + 4 16-byte loads per thread
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = q * 512 + tid, row = u >> 5, c2 = (u & 31) * 2;
            const int gr = r0 + row;
            ga[q] = *reinterpret_cast<const v2d*>(A + (size_t)(gr < nrows ? gr : nrows - 1) * lda + k0 + c2);
            if (HAS_SHIFT) gs[q] = *reinterpret_cast<const v2d*>(shift + k0 + c2);
            if (TR) gb[q] = *reinterpret_cast<const v2d*>(M + (size_t)(j0 + row) * ldm + k0 + c2);
            else gb[q] = *reinterpret_cast<const v2d*>(M + (size_t)(k0 + row) * ldm + j0 + c2);
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 64) {     // block-uniform
        if (k0 > kbeg) __syncthreads();            // the previous chunk's operand reads are done
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = q * 512 + tid, row = u >> 5, c2 = (u & 31) * 2;
            v2d v = ga[q];
            if (HAS_SHIFT) { v.x -= gs[q].x; v.y -= gs[q].y; }
            v.x *= alpha; v.y *= alpha;
            if (r0 + row >= nrows) v = (v2d){0.0, 0.0};
            *reinterpret_cast<v2d*>(&As[row * LDA + c2]) = v;
            *reinterpret_cast<v2d*>(&Bs[row * LDB + c2]) = gb[q];
        }
        __syncthreads();
        if (k0 + 64 < kend) fetch(k0 + 64);        // in flight during the MFMA chain below
#pragma unroll
        for (int half = 0; half < 2; ++half) {     // 2 x 8 k-steps: bounds the live operand registers
            double a[8], b0[8], b1[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int kk = 32 * half + 4 * s + ks;
                a[s] = As[(16 * rb + c) * LDA + kk];
                if (TR) {
                    b0[s] = Bs[(32 * cp + c) * LDB + kk];
                    b1[s] = Bs[(32 * cp + 16 + c) * LDB + kk];
                } else {
                    b0[s] = Bs[kk * LDB + 32 * cp + c];
                    b1[s] = Bs[kk * LDB + 32 * cp + 16 + c];
                }
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                acc0 = GSMVI_MFMA_F64(a[s], b0[s], acc0);
                acc1 = GSMVI_MFMA_F64(a[s], b1[s], acc1);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + 16 * rb + ks + 4 * r;
        if (row < nrows) {
            double* o = Pp + ((size_t)blockIdx.y * nrows + row) * ncols + j0 + 32 * cp + c;
            o[0] = acc0[r];
            o[16] = acc1[r];
        }
    }
}

// contraction rows per workgroup and the number of slabs for a D-deep product whose output has `tiles` 64 x 64 tiles:
// two workgroups per CU wanted, at most GSMVI_MAX_KC slabs, whole 64-deep chunks
void gsmvi_panel_wide_split(int D, int tiles, int num_cu, int kc_force, int* kc_out, int* kper_out) {
    const int nch = D / 64;
    int kc = kc_force > 0 ? kc_force : (2 * num_cu + tiles - 1) / tiles;
    if (kc > nch) kc = nch;
    if (kc > GSMVI_MAX_KC) kc = GSMVI_MAX_KC;
    if (kc < 1) kc = 1;
    const int cpw = (nch + kc - 1) / kc;
    *kc_out = (nch + cpw - 1) / cpw;
    *kper_out = cpw * 64;
}

void gsmvi_launch_panel_wide(hipStream_t st, hipEvent_t* ev, bool transposed, int D, int nrows, const double* A, int lda,
                             const double* shift, double alpha, const double* M, int ldm, double* Pp, int kper, int kc,
                             int ncols) {
    const dim3 grid(ncols / 64, kc, (nrows + 63) / 64);
#define PW(TRV, HS)                                                                                                     \
    do {                                                                                                                \
        if (ev)                                                                                                         \
            hipExtLaunchKernelGGL((k_panel_wide<TRV, HS>), grid, dim3(512), 0, st, ev[0], ev[1], 0, D, nrows, A, lda,   \
                                  shift, alpha, M, ldm, Pp, kper, ncols);                                               \
        else                                                                                                            \
            hipLaunchKernelGGL((k_panel_wide<TRV, HS>), grid, dim3(512), 0, st, D, nrows, A, lda, shift, alpha, M, ldm, \
                               Pp, kper, ncols);                                                                        \
    } while (0)
    if (transposed) PW(true, false);
    else if (shift) PW(false, true);
    else PW(false, false);
#undef PW
}
