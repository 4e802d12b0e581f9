// Device Cholesky factorisation S = R^T R (R upper triangular) for gfx950, fp64.
//
// Replaces np.linalg.cholesky inside _check_goodness (gsmvi/gsm_numpy.py:132-146, gsmvi/gsm.py:136-150,
// gsmvi/bam.py:219-233) and provides the sampling factor for x = mu + z R (replacing the SVD inside
// np.random.multivariate_normal, gsm_numpy.py:116).  A failed pivot (<= 0 or NaN) sets *info to
// 1 + pivot index, mirroring "LinAlgError or NaN => not good".
//
// Blocked right-looking algorithm, block size 64, ONE launch per block step (k_potrf_step8 below): every 64 x 64 tile
// workgroup turns the triangular solve of block row k-1 into an MFMA product with W = R_{k-1,k-1}^-T, applies the rank-64
// update to its tile, and the diagonal tile goes on to factor itself in LDS and to publish W_k.
// The strictly lower triangle of R is zero on exit (the sampler's panel product reads all of R).
#include "gsmvi_common.h"
#include "gsmvi_ctx.h"
#include "../../include/gsmvi_hip.h"

#define NB 64

#include "gsmvi_chol64.h"
#include "gsmvi_chol64b.h"

// =====================================================================================
// C = F^T F (Gram matrix of the columns; the covariance a square factor represents).  One workgroup per 64x64 tile
// of the upper triangle, K = D in chunks of 64 staged transposed in LDS ([col][66]); the mirror
// tile is written from the same accumulators, so C is exactly symmetric.  Used once per fit (return value of the
// factor-form fit) and per monitor checkpoint: not on the per-iteration path.
// =====================================================================================
// shift / shift_dev (round 6): C = F^T F + (shift + *shift_dev) I -- the accumulated jitter of a factor-form BaM fit is absorbed by
// re-factorising this matrix (bam.py:198 adds jitter * I to the covariance every iteration; gsmvi_gram_shift_f64).
__global__ __launch_bounds__(256) void k_gram(int D, const double* __restrict__ F, int ldf, double* __restrict__ C,
                                              int ldc, double shift, const double* __restrict__ shift_dev) {
    constexpr int RS = 66;
    __shared__ double FA[64 * RS];
    __shared__ double FB[64 * RS];
    const int ntr = (D + 63) >> 6;
    int ti, tj;
    {
        const int idx = blockIdx.x;
        int t = 0, base = 0;
        while (base + (ntr - t) <= idx) { base += ntr - t; ++t; }
        ti = t;
        tj = t + (idx - base);
    }
    const int I0 = ti * 64, J0 = tj * 64;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int wr = w >> 1, wc = w & 1;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < D; k0 += 64) {
        double va[16], vb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int p = (tid >> 6) + 4 * q, i = tid & 63;
            const int gi = I0 + i, gj = J0 + i, gk = k0 + p;
            va[q] = (gi < D && gk < D) ? F[(size_t)gk * ldf + gi] : 0.0;
            vb[q] = (gj < D && gk < D) ? F[(size_t)gk * ldf + gj] : 0.0;
        }
        if (k0 > 0) __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int p = (tid >> 6) + 4 * q, i = tid & 63;
            FA[i * RS + p] = va[q];
            FB[i * RS + p] = vb[q];
        }
        __syncthreads();
        const double* a0p = FA + (32 * wr + c) * RS + ks;
        const double* a1p = a0p + 16 * RS;
        const double* b0p = FB + (32 * wc + c) * RS + ks;
        const double* b1p = b0p + 16 * RS;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const double a0 = a0p[4 * s], a1 = a1p[4 * s], b0 = b0p[4 * s], b1 = b1p[4 * s];
            acc[0][0] = GSMVI_MFMA_F64(a0, b0, acc[0][0]);
            acc[0][1] = GSMVI_MFMA_F64(a0, b1, acc[0][1]);
            acc[1][0] = GSMVI_MFMA_F64(a1, b0, acc[1][0]);
            acc[1][1] = GSMVI_MFMA_F64(a1, b1, acc[1][1]);
        }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = I0 + 32 * wr + 16 * rt + ks + 4 * r;
                const int col = J0 + 32 * wc + 16 * ct + c;
                if (row < D && col < D) {
                    if (col > row) {
                        C[(size_t)row * ldc + col] = acc[rt][ct][r];
                        C[(size_t)col * ldc + row] = acc[rt][ct][r];
                    } else if (col == row) {
                        C[(size_t)row * ldc + col] = acc[rt][ct][r] + (shift + (shift_dev ? *shift_dev : 0.0));
                    }
                }
            }
}

// =====================================================================================
// Z = (X - mu) R^-1 for a few rows (R upper triangular, R^T R = Sigma): the whitened residuals whose squared
// norm gives log N(x; mu, Sigma) together with sum_i log R_ii (the monitor's log q, gsmvi/monitors.py:107).
// One workgroup per row, residual row in LDS, column-by-column forward substitution (row j of R is contiguous).
// Also returns logdiag[0] = sum_i log R_ii from workgroup 0.  O(D^2) per row with D barriers: monitor use only.
// =====================================================================================
__global__ __launch_bounds__(256) void k_whiten_rows(int D, const double* __restrict__ R, int ldr,
                                                     const double* __restrict__ X, int ldx,
                                                     const double* __restrict__ mu, double* __restrict__ Z, int ldz,
                                                     double* __restrict__ logdiag) {
    extern __shared__ double rres[];             // D doubles
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < D; i += 256) rres[i] = X[(size_t)b * ldx + i] - (mu ? mu[i] : 0.0);
    __syncthreads();
    for (int j = 0; j < D; ++j) {
        const double zj = rres[j] / R[(size_t)j * ldr + j];      // every thread: same value
        __syncthreads();                                          // all reads of rres[j] done before it is rewritten
        if (tid == 0) rres[j] = zj;
        for (int t = j + 1 + tid; t < D; t += 256) rres[t] -= zj * R[(size_t)j * ldr + t];
        __syncthreads();
    }
    for (int i = tid; i < D; i += 256) Z[(size_t)b * ldz + i] = rres[i];
    if (b == 0 && logdiag) {
        double s = 0.0;
        for (int i = tid; i < D; i += 256) s += log(R[(size_t)i * ldr + i]);
        s = wave_sum(s);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) logdiag[0] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

int gsmvi_gram_impl(hipStream_t st, int D, const double* F, int ldf, double* C, int ldc, double shift, const double* shift_dev) {
    const int ntr = (D + 63) / 64;
    hipLaunchKernelGGL(k_gram, dim3(ntr * (ntr + 1) / 2), dim3(256), 0, st, D, F, ldf, C, ldc, shift, shift_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gsmvi_set_error("gram launch failed: %s%s", hipGetErrorString(e), "");
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}

int gsmvi_whiten_impl(hipStream_t st, int D, int nrows, const double* R, int ldr, const double* X, int ldx,
                      const double* mu, double* Z, int ldz, double* logdiag) {
    hipLaunchKernelGGL(k_whiten_rows, dim3(nrows), dim3(256), (size_t)D * sizeof(double), st, D, R, ldr, X, ldx, mu, Z,
                       ldz, logdiag);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gsmvi_set_error("whiten launch failed: %s%s", hipGetErrorString(e), "");
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}

// =====================================================================================
// Step k, one 512-thread workgroup per 64x64 tile (I, J), k <= I <= J, of the trailing matrix:
//   X_I = W_{k-1} B_I,  X_J = W_{k-1} B_J     B = block row k-1 as it stood BEFORE its triangular solve (rowbuf),
//                                             W_{k-1} = R_{k-1,k-1}^-T (wbuf): the solve is an MFMA product
//   T   = A[I][J] - X_I^T X_J                 (A read from S until a tile has been written once, then from R)
//   I == k          : X_J is block (k-1, J) of the factor -> R; the mirror block (J, k-1) is zeroed
//   I == k,  J == k : T = R_kk^T R_kk in LDS (chol64), R_kk -> R, W_k = R_kk^-T -> wbuf
//   I == k,  J >  k : T is block (k, J) of the NEXT unsolved block row -> rowbuf (double-buffered with wbuf by k & 1)
//   I >  k          : T -> R
// The diagonal tile is workgroup 0 of its launch; its chain (2 products, chol64, W) bounds the step.
// Eight waves per tile workgroup:
//   * the three 64^3 products run two waves per SIMD (wave w: rows 32 wr + 16 rr .., columns 32 wc .., one 16 x 32 strip
//     of accumulators; the fp64 MFMA pipe delivers 46 TF chip-wide there against 34 TF with one wave per SIMD);
//   * the diagonal tile is factored by chol64_blk (gsmvi_chol64b.h, round 3): 16-pivot panels in registers without a
//     workgroup barrier per pivot, and W_k = R_kk^-T falls out of the augmented identity columns (round 2 ran a 64-step
//     substitution on four helper waves beside a one-barrier-per-pivot factorisation: 16.6 us per step against ~10.5).
// (Round 1's two-launch step and round 2's four-wave fused step were removed in round 3: 630 / 504 us against 439 us at
// D = 1024; profiles/r02/.)
// =====================================================================================
// nsteps (wave-uniform, <= 16): k-steps of 4 to run -- the solve's left operand W is lower triangular, so the row block
// 16 rb .. 16 rb + 15 of X = W B needs k < 16 (rb + 1) only
__device__ __forceinline__ void potrf_mma64x8(const double* FA, const double* FB, v4d (&acc)[2], int wr, int rr, int wc,
                                              int c, int ks, int nsteps = 16) {
    constexpr int RS = 66;
    const double* ap = FA + (32 * wr + 16 * rr + c) * RS + ks;
    const double* b0p = FB + (32 * wc + c) * RS + ks;
    const double* b1p = b0p + 16 * RS;
#pragma unroll 4
    for (int s = 0; s < nsteps; ++s) {
        const double a = ap[4 * s], b0 = b0p[4 * s], b1 = b1p[4 * s];
        acc[0] = GSMVI_MFMA_F64(a, b0, acc[0]);
        acc[1] = GSMVI_MFMA_F64(a, b1, acc[1]);
    }
}

// The same product with every operand read issued ahead of the first MFMA (NS k-steps, compile time): with two waves per SIMD the
// unroll-by-4 loop above runs the 64^3 product in 1.52 us on the persistent kernel's chain -- 57 cycles per MFMA where the pipe
// needs 32 -- because each group of four steps waits out the LDS latency on its own.  3 NS operand registers (96 VGPRs at NS = 16;
// the persistent kernel has 256).  Same MFMA order per accumulator, hence the same bits.
template <int NS>
__device__ __forceinline__ void potrf_mma64x8_u(const double* FA, const double* FB, v4d (&acc)[2], int rb, int wc, int c, int ks) {
    constexpr int RS = 66;
    const double* ap = FA + (16 * rb + c) * RS + ks;
    const double* b0p = FB + (32 * wc + c) * RS + ks;
    const double* b1p = b0p + 16 * RS;
    double a[NS], b0[NS], b1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        a[s] = ap[4 * s];
        b0[s] = b0p[4 * s];
        b1[s] = b1p[4 * s];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        acc[0] = GSMVI_MFMA_F64(a[s], b0[s], acc[0]);
        acc[1] = GSMVI_MFMA_F64(a[s], b1[s], acc[1]);
    }
}
// row block rb of X = W B with W lower triangular: 4 (rb + 1) k-steps (rb is wave-uniform)
__device__ __forceinline__ void potrf_mma64x8_tri(const double* FA, const double* FB, v4d (&acc)[2], int rb, int wc, int c, int ks) {
    switch (rb) {
    case 0: potrf_mma64x8_u<4>(FA, FB, acc, 0, wc, c, ks); break;
    case 1: potrf_mma64x8_u<8>(FA, FB, acc, 1, wc, c, ks); break;
    case 2: potrf_mma64x8_u<12>(FA, FB, acc, 2, wc, c, ks); break;
    default: potrf_mma64x8_u<16>(FA, FB, acc, 3, wc, c, ks); break;
    }
}

// ---- split form of the early steps of a LARGE matrix (round 3) ----------------------------------------------------------
// In the fused step every tile (I, J) recomputes X_I and X_J: three 64^3 products per tile where one is needed.  That is
// free while a step has fewer tiles than the chip has CUs (the diagonal tile's chain bounds the step), and it is 2/3 of the
// work when it has many: D = 4096 spends 69 GFLOP where the factorisation has 23 (1.46 ms of the 2.67 ms at the measured
// fp64 MFMA rate).  For steps with m = nblk - k >= POTRF_SPLIT_M tile rows the solve runs once per block in its own launch:
//   k_potrf_solve : workgroup j: X_J = W_{k-1} B_J, J = k + j  ->  block (k-1, J) of R, its mirror block zeroed, and X_J^T
//                   (64 x 64, [column][p]) into xbuf for the tiles
//   k_potrf_step8<true> : the tile loads X_I^T, X_J^T from xbuf (L2 hits) and does the one product it owns.
#define POTRF_SPLIT_M 24
__global__ __launch_bounds__(512) void k_potrf_solve(int D, int k, const double* S, int lds, double* R, int ldr,
                                                     const double* rowbuf, int ldrow, const double* wbuf,
                                                     double* __restrict__ xbuf) {
    constexpr int RS = 66;
    __shared__ __attribute__((aligned(16))) double L0[64 * RS], L1[64 * RS];
    const int J0 = (k + blockIdx.x) * NB;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int wr = (w >> 1) & 1, wc = w & 1, rr = w >> 2;
    const double* Bsrc = (k == 1) ? S : rowbuf + (size_t)((k - 1) & 1) * NB * ldrow;
    const int ldb = (k == 1) ? lds : ldrow;
    const double* Wk = wbuf + (size_t)((k - 1) & 1) * NB * NB;
    double vw[8], vj[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) vw[q] = Wk[tid + 512 * q];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int p = (tid >> 6) + 8 * q, gj = J0 + (tid & 63);
        vj[q] = (gj < D) ? Bsrc[(size_t)p * ldb + gj] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + 512 * q;
        L0[(e >> 6) * RS + (e & 63)] = vw[q];
        L1[(tid & 63) * RS + (tid >> 6) + 8 * q] = vj[q];
    }
    __syncthreads();
    v4d acc[2];
    acc[0] = acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
    potrf_mma64x8(L0, L1, acc, wr, rr, wc, c, ks, 4 * (2 * wr + rr + 1));       // X_J = W B_J (W lower triangular)
    const int lrow0 = 32 * wr + 16 * rr + ks;
    double* xo = xbuf + (size_t)(k + blockIdx.x) * NB * NB;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = lrow0 + 4 * r, col = 32 * wc + 16 * ct + c, gc = J0 + col;
            if (gc < D) R[(size_t)((k - 1) * NB + row) * ldr + gc] = acc[ct][r];
            xo[col * NB + row] = acc[ct][r];
        }
    for (int e = tid; e < NB * NB; e += 512) {
        const int jr = e >> 6, p = e & 63;
        if (J0 + jr < D) R[(size_t)(J0 + jr) * ldr + (k - 1) * NB + p] = 0.0;
    }
}

template <bool PRESOLVED>
__global__ __launch_bounds__(512) void k_potrf_step8(int D, int k, const double* S, int lds, double* R, int ldr,
                                                     double* rowbuf, int ldrow, double* wbuf, int* __restrict__ info,
                                                     unsigned long long* __restrict__ stamps,
                                                     const double* __restrict__ xbuf) {
#define PSTAMP(i)                                                                                                  \
    do {                                                                                                           \
        if (stamps && blockIdx.x == 0 && threadIdx.x == 0 && k < 64) stamps[k * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    PSTAMP(0);
    constexpr int RS = 66;
    // one LDS block: the three 64 x 66 staging tiles of the products; the diagonal tile reuses it afterwards as the
    // [tile | W] matrix (64 x 146) and the panel scratch of chol64_blk
    constexpr int ESD = 146;
    constexpr int LDS_DOUBLES = (3 * 64 * RS > 64 * ESD + CHOLB_SCRATCH_DOUBLES(true)) ? 3 * 64 * RS
                                                                                        : 64 * ESD + CHOLB_SCRATCH_DOUBLES(true);
    __shared__ __attribute__((aligned(16))) double Lall[LDS_DOUBLES];
    double* const L0 = Lall;
    double* const L1 = Lall + 64 * RS;
    double* const L2 = Lall + 2 * 64 * RS;
    __shared__ int sh_fail;
    const int nblk = (D + NB - 1) / NB, m = nblk - k;
    int ti, tj;
    {
        const int idx = blockIdx.x;
        int t = 0, base = 0;
        while (base + (m - t) <= idx) { base += m - t; ++t; }
        ti = t;
        tj = t + (idx - base);
    }
    const int I0 = (k + ti) * NB, J0 = (k + tj) * NB;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int wr = (w >> 1) & 1, wc = w & 1, rr = w >> 2;
    const bool first_row = (ti == 0), diag = (ti == 0 && tj == 0), same = (ti == tj);
    const double* Asrc = (k <= 1) ? S : R;
    const int lda = (k <= 1) ? lds : ldr;
    const double* Bsrc = (k == 1) ? S : rowbuf + (size_t)((k - 1) & 1) * NB * ldrow;
    const int ldb = (k == 1) ? lds : ldrow;
    const int lrow0 = 32 * wr + 16 * rr + ks;                     // local row of accumulator register r: lrow0 + 4 r

    double tv[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = I0 + lrow0 + 4 * r, col = J0 + 32 * wc + 16 * ct + c;
            tv[ct][r] = (row < D && col < D) ? Asrc[(size_t)row * lda + col] : ((row == col) ? 1.0 : 0.0);
        }
    v4d acc[2];
    if (PRESOLVED) {
        // X_I^T and X_J^T ([column][p], 64 x 64 each) from the solve launch: straight into the operand tiles
        const double* xi = xbuf + (size_t)(k + ti) * NB * NB;
        const double* xj = xbuf + (size_t)(k + tj) * NB * NB;
        double vi[8], vj[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            vi[q] = xi[tid + 512 * q];
            vj[q] = xj[tid + 512 * q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 512 * q;
            L2[(e >> 6) * RS + (e & 63)] = vi[q];
            L1[(e >> 6) * RS + (e & 63)] = vj[q];
        }
        __syncthreads();
        PSTAMP(2);
        acc[0] = acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
        potrf_mma64x8(L2, L1, acc, wr, rr, wc, c, ks);                // T -= X_I^T X_J
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) tv[ct][r] -= acc[ct][r];
    } else if (k > 0) {
        const double* Wk = wbuf + (size_t)((k - 1) & 1) * NB * NB;
        double vw[8], vi[8], vj[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) vw[q] = Wk[tid + 512 * q];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int p = (tid >> 6) + 8 * q, i = tid & 63;
            const int gi = I0 + i, gj = J0 + i;
            vi[q] = (gi < D) ? Bsrc[(size_t)p * ldb + gi] : 0.0;
            vj[q] = (!same && gj < D) ? Bsrc[(size_t)p * ldb + gj] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + 512 * q;
            L0[(e >> 6) * RS + (e & 63)] = vw[q];
            const int p = (tid >> 6) + 8 * q, i = tid & 63;
            L1[i * RS + p] = vi[q];
        }
        __syncthreads();
        PSTAMP(1);
        acc[0] = acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
        const int tri_steps = 4 * (2 * wr + rr + 1);                 // W[i][p] = 0 for p > i
        potrf_mma64x8(L0, L1, acc, wr, rr, wc, c, ks, tri_steps);   // X_I = W B_I
        __syncthreads();
        PSTAMP(2);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = lrow0 + 4 * r, col = 32 * wc + 16 * ct + c;
                L2[col * RS + row] = acc[ct][r];
                if (first_row && same) {
                    const int gc = J0 + col;
                    if (gc < D) R[(size_t)((k - 1) * NB + row) * ldr + gc] = acc[ct][r];
                }
            }
        if (!same) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int p = (tid >> 6) + 8 * q, i = tid & 63;
                L1[i * RS + p] = vj[q];
            }
            __syncthreads();
            acc[0] = acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
            potrf_mma64x8(L0, L1, acc, wr, rr, wc, c, ks, tri_steps);   // X_J = W B_J
            __syncthreads();
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = lrow0 + 4 * r, col = 32 * wc + 16 * ct + c;
                    L1[col * RS + row] = acc[ct][r];
                    if (first_row) {
                        const int gc = J0 + col;
                        if (gc < D) R[(size_t)((k - 1) * NB + row) * ldr + gc] = acc[ct][r];
                    }
                }
        }
        __syncthreads();
        acc[0] = acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
        potrf_mma64x8(L2, same ? L2 : L1, acc, wr, rr, wc, c, ks);    // T -= X_I^T X_J
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) tv[ct][r] -= acc[ct][r];
        if (first_row) {
            for (int e = tid; e < NB * NB; e += 512) {
                const int jr = e >> 6, p = e & 63;
                if (J0 + jr < D) R[(size_t)(J0 + jr) * ldr + (k - 1) * NB + p] = 0.0;
            }
        }
    }
    if (!diag) {
        double* dst = first_row ? rowbuf + (size_t)(k & 1) * NB * ldrow : R;
        const int ldd = first_row ? ldrow : ldr;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lrow = lrow0 + 4 * r, col = J0 + 32 * wc + 16 * ct + c;
                const int row = first_row ? lrow : I0 + lrow;
                if (I0 + lrow < D && col < D) dst[(size_t)row * ldd + col] = tv[ct][r];
            }
        return;
    }
    // ---- the diagonal tile (k, k): blocked Cholesky of [tile | I] (chol64_blk): R_kk and W_k = R_kk^-T in one pass ----
    const int nb = (D - I0) < NB ? (D - I0) : NB;
    double* const E = Lall;                                       // [64][ESD]: columns 0..63 the tile, 64..127 W
    double* const scr = Lall + 64 * ESD;
    __syncthreads();                                              // everyone is done with the staging tiles
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = lrow0 + 4 * r, j = 32 * wc + 16 * ct + c;
            E[i * ESD + j] = (i < nb && j < nb) ? (j >= i ? tv[ct][r] : 0.0) : (i == j ? 1.0 : 0.0);
        }
    __syncthreads();
    PSTAMP(3);
    chol64_blk<ESD, false, true>(E, scr, nb, &sh_fail);
    PSTAMP(4);
    if (tid == 0 && sh_fail != 0 && *info == 0) *info = I0 + sh_fail;
    for (int e = tid; e < NB * NB; e += 512) {
        const int i = e >> 6, j = e & 63;
        if (i < nb && j < nb) R[(size_t)(I0 + i) * ldr + I0 + j] = (j >= i) ? E[i * ESD + j] : 0.0;
    }
    if (m > 1) {                                                  // the next step's triangular solve: W_k[i][p], row-major
        double* Wk = wbuf + (size_t)(k & 1) * NB * NB;
        for (int e = tid; e < NB * NB; e += 512) Wk[e] = E[(e >> 6) * ESD + 64 + (e & 63)];
    }
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PSTAMP(5); }
}

#undef PSTAMP

__global__ void k_potrf_clear_info(int* info) { *info = 0; }

// =====================================================================================
// ONE persistent launch per factorisation (round 6): the block steps as a task graph with look-ahead.
//
// The launch-per-step form above puts every step's whole chain -- launch boundary (~5 us), tile loads, the two solve products,
// the tile update, the 64 x 64 factorisation (10.3 us) -- on the critical path: 20 us x 16 steps at D = 1024.  Only the
// factorisation of the diagonal tile and the one rank-64 update it still needs are inherently serial.  Here:
//   * workgroup 0 is the CHAIN: iteration c solves block (c-1, c) with the W_{c-1} it has just produced, applies that block's
//     rank-64 update to tile (c, c), factors it (chol64_blk, [T | I] -> [R_cc | W_c]) and publishes W_c.  Everything else tile
//     (c, c) and block (c-1, c) need has been applied by the other workgroups while the chain was factoring tile (c-1, c-1);
//   * the other workgroups take TILES from a ticket counter, block rows in order: a task is tile (I, J) with ALL the rank-64
//     updates it needs -- steps 0 .. I-1 (0 .. I-2 on the diagonal) -- applied in step order with the tile in registers, the two
//     solved blocks of step p + 1 travelling while step p's product runs; then
//       (I, I), (I, I+1)  : the tile goes to R for the chain, which applies the last update / the solve itself;
//       (I, J >= I+2)     : X = W_I T -> block (I, J) of R, in place (the mirror block is zeroed).
//     One product per tile and step (the launch-per-step form recomputes both solves in every tile: three), and the tile is read
//     once (from S) and written once.  (The first version of this kernel took one task per tile AND step: every product paid a
//     ticket, a flag poll, the tile's round trip through memory and a drain -- 10 us per 64^3 product, 1.83 ms at D = 4096
//     where the chain needs 1.17; this form: 1.19 ms, and 15.8 ms at D = 12288 = 84 % of the fp64 MFMA rate the chip sustains.)
//     A task waits only for tiles with smaller tickets or for the chain, which waits only for tiles of its own and earlier rows
//     that precede every tile waiting for it: no cycle, and with at most one workgroup per CU in the grid every workgroup is
//     resident, so every claimed task runs.
//   * hand-offs are flags in the workspace (form R1 below: write-through stores, a drain, a relaxed flag, sc1 loads of everything
//     handed off); every wait is BOUNDED (max_spin polls, ~3 s): on a timeout the
//     waiter raises the abort flag, every workgroup leaves and *info reports D + 1 -- the pool's GPUs are shared, a kernel that
//     can spin forever is not acceptable.
// R must not alias S.  Same arithmetic per tile as the launch-per-step form, same summation order inside every product; the
// ORDER of the rank-64 updates of a tile is the step order in both (T = ((S - P_0) - P_1) - ..., never S - (P_0 + P_1 + ...)),
// so the factor is bit-identical to k_potrf_step8<true>'s (one product per tile).
// =====================================================================================
#define DAG_TICKET 0
#define DAG_ABORT 1
#define DAG_WREADY 2
// Hand-offs (cdna_hip_programming.md, Guideline 16, form R1): every handed-off byte is stored WRITE-THROUGH (sc1: a relaxed
// agent-scope atomic store of the double), every storing wave drains its stores (s_waitcnt vmcnt(0)) in front of the workgroup
// barrier, ONE lane raises the flag with a relaxed agent-scope store; the consumer's lane 0 polls the flag words with relaxed
// agent-scope loads, and EVERY load of handed-off data is an sc1 load (relaxed agent-scope atomic load), so no acquire fence is
// needed.  (The first version used acquire loads in the poll and release / acquire fences around it: each poll invalidated
// the caches, each publish wrote the XCD's L2 back -- 1070 us at D = 1024 against 324 for the launch-per-step form.)
#define DAG_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
// Workgroup barrier that waits for this wave's LDS operations only.  __syncthreads() carries a workgroup fence that drains the
// vector-memory counter: behind the write-through stores of a solved block the chain's next barrier waited ~3 us for memory to
// acknowledge them (measured with the chain's timeline: "products + E" 4.84 us for two 64^3 products).  Nothing in this kernel
// orders GLOBAL data through a workgroup barrier -- flags are atomics, handed-off data is read with sc1 loads, and dag_publish
// drains explicitly -- so every barrier outside chol64_blk is of this kind.
#define DAG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__device__ __forceinline__ int dag_ld(const int* p) { return __hip_atomic_load(p, DAG_RLX); }
__device__ __forceinline__ void dag_st(int* p, int v) { __hip_atomic_store(p, v, DAG_RLX); }
__device__ __forceinline__ double dag_ldd(const double* p) { return __hip_atomic_load(p, DAG_RLX); }
__device__ __forceinline__ void dag_std(double* p, double v) { __hip_atomic_store(p, v, DAG_RLX); }

// workgroup-wide bounded wait for up to three flags (null = none); false = aborted / timed out (block-uniform)
__device__ __forceinline__ bool dag_wait(int* flags, const int* f0, int n0, const int* f1, int n1, const int* f2, int n2,
                                         int* sh, int max_spin) {
    if (threadIdx.x == 0) {
        int ok = 1, spins = 0;
        for (;;) {
            const int v0 = f0 ? dag_ld(f0) : n0, v1 = f1 ? dag_ld(f1) : n1, v2 = f2 ? dag_ld(f2) : n2;   // (one round trip, not three)
            if (v0 >= n0 && v1 >= n1 && v2 >= n2) break;
            if (dag_ld(flags + DAG_ABORT) != 0 || ++spins > max_spin) {
                ok = 0;
                dag_st(flags + DAG_ABORT, 1);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        *sh = ok;
    }
    DAG_BARRIER();
    const int ok = *sh;
    DAG_BARRIER();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       // (no instruction: keeps the compiler from hoisting loads above the poll)
    return ok != 0;
}
// every storing wave drains its write-through stores, then one lane raises the flag
__device__ __forceinline__ void dag_publish(int* f, int v) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DAG_BARRIER();
    if (threadIdx.x == 0) dag_st(f, v);
}

// How many of the steps pp, pp + 1, ... < P have their solved block(s) in R: lane l of wave 0 reads the flag(s) of step pp + l
// (64 consecutive words per operand column), the count of leading ready steps goes to everybody.  Blocks until step pp is there;
// returns the first step NOT known ready, or -1 (aborted / timed out).
__device__ __forceinline__ int dag_upto(int* flags, const int* colI, const int* colJ, int pp, int P, int* sh, int max_spin) {
    if (threadIdx.x < 64) {
        const int q = pp + (int)threadIdx.x;
        int n = 0, spins = 0;
        for (;;) {
            int ok = 0;
            if (q < P) ok = (dag_ld(colI + q) != 0) && (colJ == nullptr || dag_ld(colJ + q) != 0);
            const unsigned long long m = __ballot(ok);
            n = (m == ~0ull) ? 64 : (int)__builtin_ctzll(~m);
            if (n > 0) break;
            const int ab = (threadIdx.x == 0) ? dag_ld(flags + DAG_ABORT) : 0;
            if (__builtin_amdgcn_readfirstlane(ab) != 0 || ++spins > max_spin) {
                n = -1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (threadIdx.x == 0) {
            if (n < 0) dag_st(flags + DAG_ABORT, 1);
            *sh = n;
        }
    }
    DAG_BARRIER();
    const int n = *sh;
    DAG_BARRIER();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return n < 0 ? -1 : pp + n;
}

// The chain's stores that nobody reads in this launch, for block step cblk: the factor's diagonal block (upper triangle of the
// [R | W] matrix chol64_blk left in E, 146 doubles per row) and, for the block the chain solved from S, the mirror block (1, 0).
template <int ESD>
__device__ __forceinline__ void dag_store_own_t(int D, double* R, int ldr, const double* E, int cblk, int tid) {
    const int I0 = cblk * NB;
    const int nb = (D - I0) < NB ? (D - I0) : NB;
    for (int e = tid; e < NB * NB; e += 512) {
        const int i = e >> 6, j = e & 63;
        if (i < nb && j < nb) R[(size_t)(I0 + i) * ldr + I0 + j] = (j >= i) ? E[i * ESD + j] : 0.0;
    }
    if (cblk == 1)                                                // (from block row 1 on the worker that owned tile (c-1, c) zeroes its mirror)
        for (int e = tid; e < NB * NB; e += 512) {
            const int jr = e >> 6, pcol = e & 63;
            if (I0 + jr < D) R[(size_t)(I0 + jr) * ldr + (cblk - 1) * NB + pcol] = 0.0;
        }
}

__global__ __launch_bounds__(512) void k_potrf_dag(int D, const double* S, int lds, double* R, int ldr, double* wbuf, int* flags,
                                                   int* __restrict__ info, int max_spin, unsigned long long* __restrict__ stamps) {
    // timeline diagnostic (knob "timeline" = 2): the chain's thread 0 stamps up to sixteen points (0-7 phases, 8-10 inside the products) of each of its first 64 iterations
#define DSTAMP(i)                                                                                              \
    do {                                                                                                       \
        if (stamps && threadIdx.x == 0 && cI < 64) stamps[cI * 16 + (i)] = __builtin_amdgcn_s_memrealtime();   \
    } while (0)
    constexpr int RS = 66;
    constexpr int ESD = 146;
    // LDS: the [tile | W] matrix of chol64_blk and its panel scratch; the two staging tiles of the products share the matrix's
    // space (dead while they are live); the W tile (left operand of the solves) has its own, so the chain keeps W_{c-1} in LDS
    __shared__ __attribute__((aligned(16))) double Lall[64 * ESD + CHOLB_SCRATCH_DOUBLES(true)];
    __shared__ __attribute__((aligned(16))) double L0[64 * RS];
    double* const L1 = Lall;
    double* const L2 = Lall + 64 * RS;
    __shared__ int sh_fail, sh_w;
    const int nblk = (D + NB - 1) / NB;
    int* const wready = flags + DAG_WREADY;
    int* const xready = wready + nblk;                            // [J][p]: block (p, J) of the factor is in R (one line per column J)
    int* const tstep = xready + nblk * nblk;                      // [I][J]: rank-64 updates applied to the tile a worker left in R for the chain
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int wr = (w >> 1) & 1, wc = w & 1, rr = w >> 2;
    const int lrow0 = 32 * wr + 16 * rr + ks;

    if (blockIdx.x == 0) {
        // ================================ the chain ================================
        for (int cI = 0; cI < nblk; ++cI) {
            const int I0 = cI * NB;
            DSTAMP(0);
            if (cI >= 2) {
                if (!dag_wait(flags, tstep + (cI - 1) * nblk + cI, cI - 1, tstep + cI * nblk + cI, cI - 1, nullptr, 0, &sh_w, max_spin)) {
                    if (tid == 0) *info = D + 1;
                    return;
                }
            }
            DSTAMP(1);                                            // flags seen
            const bool a_s = (cI <= 1);                           // tile (c, c) still in S
            const double* Asrc = a_s ? S : R;
            const int lda = a_s ? lds : ldr;
            // raw loads first, masks after the own stores below: a select right behind a load would put the wait for the loads in
            // front of those stores (measured: 3.0 us for this phase that way)
            double tv[2][4];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = I0 + lrow0 + 4 * r, col = I0 + 32 * wc + 16 * ct + c;
                    const double* ap = Asrc + (size_t)(row < D ? row : 0) * lda + (col < D ? col : 0);
                    tv[ct][r] = a_s ? *ap : dag_ldd(ap);
                }
            if (cI > 0) {
                const bool b_s = (cI == 1);
                const double* Bsrc = b_s ? S : R;
                const int ldb = b_s ? lds : ldr;
                double vi[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int pr = (tid >> 6) + 8 * q, gi = I0 + (tid & 63);
                    const double* bp = Bsrc + (size_t)((cI - 1) * NB + pr) * ldb + (gi < D ? gi : 0);
                    vi[q] = b_s ? *bp : dag_ldd(bp);
                }
                // behind the loads (their ~1.9 us of latency is the chain's): the stores of the PREVIOUS iteration that are nobody's
                // input in this launch -- the factor's diagonal block, still in E, and the mirror block below it
                dag_store_own_t<ESD>(D, R, ldr, Lall, cI - 1, tid);
                if (I0 + (tid & 63) >= D) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) vi[q] = 0.0;
                }
                // W_{c-1} is published HERE, not behind its own drain at the end of the previous iteration: the wait for this
                // iteration's loads is a wait for those older write-through stores too, and W_{c-1}'s readers -- the solves of
                // row c-1 -- feed iteration c+1, a whole iteration away (0.7 us per iteration off the chain)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                DAG_BARRIER();                                  // E has been read: its space becomes the staging tiles
                if (tid == 0) dag_st(wready + (cI - 1), 1);
#pragma unroll
                for (int q = 0; q < 8; ++q) L1[(tid & 63) * RS + (tid >> 6) + 8 * q] = vi[q];
                DAG_BARRIER();                                  // (L0 holds W_{c-1}: written at the end of the previous iteration)
                DSTAMP(2);                                        // both tiles loaded (sc1) and staged
                v4d acc[2];
                acc[0] = acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
                // the solve's row blocks cost 4, 8, 12, 16 k-steps (W is lower triangular): waves w and w + 4 share a SIMD, so they
                // take row blocks {0, 3} and {1, 2} -- 20 k-steps per SIMD where the products' own mapping gives 12 and 28
                const int rbS = rr ? 3 - wr : wr;
                potrf_mma64x8_tri(L0, L1, acc, rbS, wc, c, ks);                              // X = W_{c-1} T_{c-1,c}
                DSTAMP(8);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * rbS + ks + 4 * r, col = 32 * wc + 16 * ct + c, gc = I0 + col;
                        L2[col * RS + row] = acc[ct][r];
                        if (gc < D) dag_std(R + (size_t)((cI - 1) * NB + row) * ldr + gc, acc[ct][r]);
                    }
                DAG_BARRIER();                                                            // L2 (the X^T tile) is complete
                DSTAMP(9);
                acc[0] = acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
                potrf_mma64x8_u<16>(L2, L2, acc, 2 * wr + rr, wc, c, ks);                    // T_cc -= X^T X
                DSTAMP(10);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tv[ct][r] -= acc[ct][r];
            }
            const int nb = (D - I0) < NB ? (D - I0) : NB;
            double* const E = Lall;
            double* const scr = Lall + 64 * ESD;
            DAG_BARRIER();                                      // everyone is done with the staging tiles
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = lrow0 + 4 * r, j = 32 * wc + 16 * ct + c;
                    E[i * ESD + j] = (i < nb && j < nb) ? (j >= i ? tv[ct][r] : 0.0) : (i == j ? 1.0 : 0.0);
                }
            // the solved block (c-1, c) is published here: its write-through stores have had the product and the staging above to
            // land, so the drain in front of the flag costs the chain next to nothing (dag_publish's barrier is the one E needs)
            DSTAMP(3);                                            // solve + update products done, E staged
            if (cI > 0) dag_publish(xready + cI * nblk + (cI - 1), 1);
            else DAG_BARRIER();
            DSTAMP(4);                                            // solved block published
            chol64_blk<ESD, false, 1, true>(E, scr, nb, &sh_fail);   // (look-ahead inside the block: gsmvi_chol64b.h)
            DSTAMP(5);                                            // factorisation done
            if (tid == 0 && sh_fail != 0 && *info == 0) *info = I0 + sh_fail;
            // (Measured and dropped, round 6: the loop rotated so that the two tiles of iteration c+1 are loaded right behind the
            // factorisation and land with the W / factor stores -- 351 us at D = 1024 against 310 for this form and 324 for one
            // launch per step: the earlier wait for their flags sits on the chain, and the rotated body spilled 22 VGPRs.)
            // W_c first (it is what the other workgroups wait for), into the chain's own operand tile and out to the workers
            {
                double* Wk = wbuf + (size_t)cI * NB * NB;
                for (int e = tid; e < NB * NB; e += 512) {
                    const double wv = E[(e >> 6) * ESD + 64 + (e & 63)];
                    L0[(e >> 6) * RS + (e & 63)] = wv;
                    dag_std(Wk + e, wv);
                }
                DSTAMP(6);                                        // W_c copied and on its way (published by the next iteration)
            }
            DSTAMP(7);                                            // (the factor block itself is stored behind the next iteration's loads)
        }
        dag_store_own_t<ESD>(D, R, ldr, Lall, nblk - 1, tid);
        return;
    }
#undef DSTAMP
    // ================================ the workers ================================
    // A task is one TILE (I, J) with ALL the rank-64 updates it needs, applied in step order with the tile in registers.  Ticket
    // order, per block row r:
    //   kind 1  (r, r+1),   1 <= r <= nblk-2 : steps 0 .. r-1, tile -> R, tstep[r][r+1] = r           (the chain solves it)
    //   kind 0  (r+1, r+1), same rows        : steps 0 .. r-1, tile -> R, tstep[r+1][r+1] = r         (the chain applies step r itself)
    //   kind 2  (r, J >= r+2)                : steps 0 .. r-1, then X = W_r T -> block (r, J) of R, mirror block zeroed, xready[J][r]
    // The two tiles chain iteration r + 1 waits for come BEFORE row r's solves: those wait for W_r, which that iteration
    // publishes behind its own wait -- with the diagonal tile (r+1, r+1) at the head of row r + 1 a grid of fewer workgroups than
    // row r has solves would hold them all and leave nobody for it (tests/test_potrf_dag_order.py simulates the order).
    // The operands of step p + 1 (two solved blocks, sc1 loads) travel while step p's product runs; how many steps are ready is
    // polled 64 at a time (one line per operand column: xready is [column][step]), so a tile whose inputs are there runs its
    // products back to back.
    int row = 0, base = 0;
    for (;;) {
        if (tid == 0) sh_w = atomicAdd(flags + DAG_TICKET, 1);
        DAG_BARRIER();
        const int t = sh_w;
        DAG_BARRIER();
        int hasC = 0;
        while (row < nblk) {
            hasC = (row >= 1 && row + 1 < nblk) ? 1 : 0;          // the two tiles chain iteration row + 1 waits for
            const int nG = nblk - row - 2 > 0 ? nblk - row - 2 : 0;
            const int cnt = 2 * hasC + nG;
            if (t - base < cnt) break;
            base += cnt;
            ++row;
        }
        if (row >= nblk) return;
        const int r_ = t - base;
        const int kind = (r_ < hasC) ? 1 : (r_ < 2 * hasC ? 0 : 2);
        const int I = (kind == 0) ? row + 1 : row;
        const int J = (kind == 2) ? row + 2 + (r_ - 2 * hasC) : row + 1;
        const int P = (kind == 0) ? I - 1 : I;                    // rank-64 updates this task applies
        const bool same = (kind == 0);
        const int I0 = I * NB, J0 = J * NB;
        const int gi = I0 + (tid & 63), gj = J0 + (tid & 63);
        double tv[2][4];                                          // the tile, from S (plain loads, in flight while the operands arrive)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rw = I0 + lrow0 + 4 * r, col = J0 + 32 * wc + 16 * ct + c;
                tv[ct][r] = S[(size_t)(rw < D ? rw : 0) * lds + (col < D ? col : 0)];
            }
        double vi[8], vj[8];
        int upto = 0;
        bool have = false;
        for (int pp = 0; pp < P; ++pp) {
            if (!have) {
                if (pp >= upto) {
                    upto = dag_upto(flags, xready + I * nblk, same ? nullptr : xready + J * nblk, pp, P, &sh_w, max_spin);
                    if (upto < 0) return;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const size_t ro = (size_t)(pp * NB + (tid >> 6) + 8 * q) * ldr;
                    vi[q] = dag_ldd(R + ro + (gi < D ? gi : 0));
                    if (!same) vj[q] = dag_ldd(R + ro + (gj < D ? gj : 0));
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                L2[(tid & 63) * RS + (tid >> 6) + 8 * q] = (gi < D) ? vi[q] : 0.0;
                if (!same) L1[(tid & 63) * RS + (tid >> 6) + 8 * q] = (gj < D) ? vj[q] : 0.0;
            }
            DAG_BARRIER();
            have = (pp + 1 < P && pp + 1 < upto);
            if (have) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const size_t ro = (size_t)((pp + 1) * NB + (tid >> 6) + 8 * q) * ldr;
                    vi[q] = dag_ldd(R + ro + (gi < D ? gi : 0));
                    if (!same) vj[q] = dag_ldd(R + ro + (gj < D ? gj : 0));
                }
            }
            v4d acc[2];
            acc[0] = acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
            potrf_mma64x8_u<16>(L2, same ? L2 : L1, acc, 2 * wr + rr, wc, c, ks);          // X_pI^T X_pJ
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) tv[ct][r] -= acc[ct][r];                        // (step order: the launch-per-step form's bits)
            DAG_BARRIER();                                        // the staging tiles are free again
        }
        if (kind != 2) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rw = I0 + lrow0 + 4 * r, col = J0 + 32 * wc + 16 * ct + c;
                    if (rw < D && col < D) dag_std(R + (size_t)rw * ldr + col, tv[ct][r]);
                }
            dag_publish(tstep + I * nblk + J, kind == 0 ? I - 1 : I);
            if (kind == 1)                                        // the mirror of the block the chain is about to solve: off its path
                for (int e = tid; e < NB * NB; e += 512) {
                    const int jr = e >> 6, pcol = e & 63;
                    if (J0 + jr < D) R[(size_t)(J0 + jr) * ldr + I0 + pcol] = 0.0;
                }
            DAG_BARRIER();
        } else {
            // the updated tile, transposed, is the solve's right operand; W_I arrives from the chain
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) L1[(32 * wc + 16 * ct + c) * RS + lrow0 + 4 * r] = tv[ct][r];
            if (!dag_wait(flags, wready + I, 1, nullptr, 0, nullptr, 0, &sh_w, max_spin)) return;
            const double* Wk = wbuf + (size_t)I * NB * NB;
            double vw[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) vw[q] = dag_ldd(Wk + tid + 512 * q);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = tid + 512 * q;
                L0[(e >> 6) * RS + (e & 63)] = vw[q];
            }
            DAG_BARRIER();
            v4d acc[2];
            acc[0] = acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
            const int rbS = rr ? 3 - wr : wr;                     // (row blocks {0, 3} and {1, 2} per SIMD: see the chain)
            potrf_mma64x8_tri(L0, L1, acc, rbS, wc, c, ks);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rw = 16 * rbS + ks + 4 * r, col = 32 * wc + 16 * ct + c, gc = J0 + col;
                    if (gc < D) dag_std(R + (size_t)(I0 + rw) * ldr + gc, acc[ct][r]);
                }
            for (int e = tid; e < NB * NB; e += 512) {
                const int jr = e >> 6, pcol = e & 63;
                if (J0 + jr < D) R[(size_t)(J0 + jr) * ldr + I0 + pcol] = 0.0;
            }
            dag_publish(xready + J * nblk + I, 1);
            DAG_BARRIER();
        }
    }
}

__global__ void k_potrf_dag_clear(int* info, int* flags, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *info = 0;
    if (i < n) flags[i] = 0;
}

int gsmvi_potrf_impl(gsmvi_ctx* ctx, hipStream_t st, int D, const double* S, int lds, double* R, int ldr,
                     int* info_dev) {
    // one launch per block step (two for the early steps of a large matrix); workspace (the idle panel-partial slab): two row
    // buffers, two W blocks and the X^T blocks of the split steps
    const int nblk = (D + NB - 1) / NB, ldrow = nblk * NB;
    double* rowbuf = ctx->pp;
    double* wbuf = rowbuf + (size_t)2 * NB * ldrow;
    // (every size: 72 / 292 / 585 us and 1.19 / 2.35 / 4.99 / 15.8 ms at D = 256 / 1024 / 2048 / 4096 / 6144 / 8192 / 12288 against
    // 77 / 323 / 712 us and 2.01 / 4.9 / 10.2 / 33.6 ms for one launch per step; knob "potrf_dag" = 0 selects the latter)
    if (ctx->tune_potrf_dag && !ctx->tune_no_fast &&
        (!ctx->timeline_stamps(3) || ctx->tune_timeline == 2)) {
        // one persistent launch (k_potrf_dag): W blocks and the flags live where the launch-per-step form keeps its row buffers
        double* wb = ctx->pp;
        int* flags = reinterpret_cast<int*>(wb + (size_t)nblk * NB * NB);
        const int nflags = DAG_WREADY + nblk + 2 * nblk * nblk;
        int ntasks = 0;
        for (int r = 0; r < nblk; ++r)                           // tiles the workers own: (r, r+1) and (r+1, r+1) for 1 <= r <= nblk-2, (r, J >= r+2)
            ntasks += 2 * (r >= 1 && r + 1 < nblk) + (nblk - r - 2 > 0 ? nblk - r - 2 : 0);
        int grid = 1 + ntasks;
        if (grid > ctx->num_cu) grid = ctx->num_cu;              // at most one workgroup per CU: all resident (133 KB of LDS each)
        if (ctx->tune_potrf_workers > 0 && grid > 1 + ctx->tune_potrf_workers) grid = 1 + ctx->tune_potrf_workers;   // (tests: a small grid)
        hipLaunchKernelGGL(k_potrf_dag_clear, dim3((nflags + 255) / 256), dim3(256), 0, st, info_dev, flags, nflags);
        hipLaunchKernelGGL(k_potrf_dag, dim3(grid), dim3(512), 0, st, D, S, lds, R, ldr, wb, flags, info_dev,
                           ctx->tune_potrf_spin > 0 ? ctx->tune_potrf_spin : 2000000,    // (~3 s of polls: far above any scheduling gap of a shared GPU -- eight
                           // processes time-slicing one device in tests/test_gpu_dist.py --, far below a harness's patience)
                           ctx->tune_timeline == 2 ? ctx->timeline_stamps(3) : nullptr);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) {
            gsmvi_set_error("potrf launch failed: %s%s", hipGetErrorString(e2), "");
            return GSMVI_ERR_HIP;
        }
        ctx->potrf_w = wb;
        ctx->potrf_r = R;
        ctx->potrf_w_n = D;
        return GSMVI_OK;
    }
    ctx->potrf_w = nullptr;
    ctx->potrf_w_n = 0;
    hipLaunchKernelGGL(k_potrf_clear_info, dim3(1), dim3(1), 0, st, info_dev);
    double* xbuf = wbuf + (size_t)2 * NB * NB;                    // nblk blocks of 64 x 64: X_J^T of the split steps
    for (int k = 0; k < nblk; ++k) {
        const int m = nblk - k;
        if (k > 0 && m >= (ctx->tune_potrf_split_m > 0 ? ctx->tune_potrf_split_m : POTRF_SPLIT_M) && !ctx->tune_no_fast) {
            hipLaunchKernelGGL(k_potrf_solve, dim3(m), dim3(512), 0, st, D, k, S, lds, R, ldr, rowbuf, ldrow, wbuf, xbuf);
            hipLaunchKernelGGL(k_potrf_step8<true>, dim3(m * (m + 1) / 2), dim3(512), 0, st, D, k, S, lds, R, ldr, rowbuf,
                               ldrow, wbuf, info_dev, ctx->timeline_stamps(3), xbuf);
        } else {
            hipLaunchKernelGGL(k_potrf_step8<false>, dim3(m * (m + 1) / 2), dim3(512), 0, st, D, k, S, lds, R, ldr, rowbuf,
                               ldrow, wbuf, info_dev, ctx->timeline_stamps(3), xbuf);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gsmvi_set_error("potrf launch failed: %s%s", hipGetErrorString(e), "");
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}
