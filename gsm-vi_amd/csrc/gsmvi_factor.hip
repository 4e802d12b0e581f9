// Factor-form GSM update for gfx950 (fp64): the covariance is never formed or factorised.
//
// State: mu (D), Fm (D x D, any square factor with Sigma = Fm^T Fm); whitened draws Z (B x D), samples
// X = mu + Z Fm, scores G = lp_g(X).  Algebra: SURVEY Appendix A.2 (derived from gsmvi/gsm_numpy.py:4-55,
// verified against the imported reference; there is no reference implementation of this form), written
// for row vectors:
//   W = G Fm^T                               (transposed panel product, k_panel_t)
//   ww = w.w, zw = z.w, rho = 0.5 sqrt(1+4(ww+zw^2)) - 0.5, den = 1+rho-zw
//   u = ((w+z) + z (ww+zw)/den)/(1+rho)       dmu_b = u_b Fm,  mu' = mu + mean_b dmu_b
//   Sigma' = Fm^T M Fm,  M = I + Rt^T J Rt,  Rt = [Z; U] (n = 2B rows),  J = (1/B) [[0, I], [I, -I]]
// A factor of M:  Gamma = Rt Rt^T = Rg^T Rg (Cholesky),  I + Rg J Rg^T = T^T T (Cholesky; it exists iff
// M is positive definite -- this IS the reference's accept/revert test, gsm_numpy.py:121-125,132-146,
// on an n x n matrix instead of D x D),  C = I + Rt^T Rg^-1 (T - I) Rg^-T Rt,  C^T C = M, hence
//   Fm' = C Fm = Fm + Rt^T K Rt Fm,   K = Rg^-1 (T - I) Rg^-T.
// Round 3 -- everything of size D is done in the basis [Z; V], V = W + Z (the whitened residual, -> 0 at the fixed point):
//   u_b = beta_b z_b + alpha_b v_b,  alpha = 1/(1+rho),  beta = (w.v / den)/(1+rho),  i.e.  Rt = S [Z; V],
//   S = [[I, 0], [diag(beta), diag(alpha)]].  The per-sample dots are entries of the Gram matrix
//   Gamma1 = [Z; V][Z; V]^T (zz, zv, vv on its diagonals: zw = zv - zz, ww = vv - 2 zv + zz, w.v = vv - zv), so there is no
//   per-sample launch and no pass over D between W and the next product:  Gamma = S Gamma1 S^T (formed inside the
//   small-matrix kernel),  K'' = S^T K S,  Tm = [Z; V] Fm = [X - mu; V Fm],
//   Fm' = Fm + [Z; V]^T (K'' Tm),   mu' = mu + (1/B) sum_b (beta_b (x_b - mu) + alpha_b (v_b Fm)).
//   (V, not W, is the second block: beta z and alpha v are both small near the fixed point, so no large terms cancel
//   in Gamma's U-block -- the rank-revealing rule for dependent rows relies on its rounding floor.)
// Per iteration Fm is read three times (W, V Fm, update) and written once; no O(D^3) work.
// Requires n = 2B <= D (Gamma must be nonsingular) and n <= 128.
#include "gsmvi_common.h"
#include "gsmvi_ctx.h"
#include "gsmvi_chol64.h"
#include "gsmvi_chol64b.h"
#include "gsmvi_chol128.h"
#include "gsmvi_smallgemm.h"
#include "gsmvi_small16.h"
#include "../../include/gsmvi_hip.h"
#include "../../include/gsmvi_hip_debug.h"

// ---- transposed panel product partials: Pp[kc][r][j] = sum_{i in chunk(kc)} A[r][i] M[j][i] ----------
// M has mrows rows of length D (mrows = D for the square factor; a row block of a sharded matrix otherwise);
// the slabs are nrows x mrows.
// Workgroup = 16 rows j of M x CH-column chunks.  Both the A chunk (NR x CH) and the M tile (16 x CH)
// are loaded with coalesced 16-B accesses along i and staged in LDS [row][CH+2]; MFMA A operand =
// A rows, B operand = M rows (B[k][col j] = M[j][k]).  Guarded: any D, any alignment falls back to
// 8-B loads.
template <int MT, int CH>
__global__ __launch_bounds__(256) void k_panel_t(int D, int nrows, const double* __restrict__ A, int lda,
                                                 const double* __restrict__ M, int ldm, double* __restrict__ Pp,
                                                 int chunks_per_wg, int mrows) {
    constexpr int LDG = CH + 2;
    constexpr int NR = 16 * MT;
    constexpr int KW = CH / 4;                    // columns of the chunk per wave
    __shared__ double As[(NR + 16) * LDG];
    double* Ms = As + NR * LDG;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int j0 = blockIdx.x * 16, r0 = blockIdx.z * NR;
    v4d acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int ch = 0; ch < chunks_per_wg; ++ch) {
        const int cbase = (blockIdx.y * chunks_per_wg + ch) * CH;
        if (cbase >= D) break;
        if (ch > 0) __syncthreads();
        for (int e = tid; e < (NR + 16) * CH; e += 256) {
            const int row = e / CH, col = e % CH;
            const int gc = cbase + col;
            double v = 0.0;
            if (gc < D) {
                if (row < NR) {
                    const int gr = r0 + row;
                    if (gr < nrows) v = A[(size_t)gr * lda + gc];
                } else {
                    const int gj = j0 + row - NR;
                    if (gj < mrows) v = M[(size_t)gj * ldm + gc];
                }
            }
            As[row * LDG + col] = v;
        }
        __syncthreads();
        const double* ap = As + c * LDG + KW * w + ks;
        const double* bp = Ms + c * LDG + KW * w + ks;
#pragma unroll 4
        for (int s = 0; s < KW / 4; ++s) {
            const double b = bp[4 * s];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = GSMVI_MFMA_F64(ap[mt * 16 * LDG + 4 * s], b, acc[mt]);
        }
    }
    __syncthreads();
    double* red = As;                              // [4][NR][17]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(w * NR + 16 * mt + ks + 4 * r) * 17 + c] = acc[mt][r];
    __syncthreads();
    for (int idx = tid; idx < NR * 16; idx += 256) {
        const int rr = idx >> 4, cc = idx & 15;
        const int row = r0 + rr, col = j0 + cc;
        if (row < nrows && col < mrows) {
            const double s = (red[(0 * NR + rr) * 17 + cc] + red[(1 * NR + rr) * 17 + cc]) +
                             (red[(2 * NR + rr) * 17 + cc] + red[(3 * NR + rr) * 17 + cc]);
            Pp[((size_t)blockIdx.y * nrows + row) * mrows + col] = s;
        }
    }
}

// ---- fast transposed panel product (D % 64 == 0, even ld, 16-B aligned bases) ---------------------------
// Same decomposition as k_panel_fast: 512 threads, 8 waves split the chunk's columns, every global load of a
// chunk (A rows and the 16 M rows, both contiguous along i) issued in one batch as 16-B accesses, staged in
// LDS [row][CHW+2], MFMA operands pulled to registers before the chain.
template <int MT, int CHW, bool RAG>     // RAG: mrows % 16 != 0 (clamped re-reads of M rows, guarded stores; round 5)
__global__ __launch_bounds__(512) void k_panel_t_fast(int D, int nrows, const double* __restrict__ A, int lda,
                                                      const double* __restrict__ M, int ldm,
                                                      double* __restrict__ Pp, int chunks_per_wg, int mrows) {
    constexpr int LDG = CHW + 2;
    constexpr int NR = 16 * MT;
    constexpr int RW = CHW / 8;                    // columns of the chunk per wave
    constexpr int NST = RW / 4;
    constexpr int U16 = CHW / 2;                   // 16-B units per staged row
    constexpr int UPT = (NR + 16) * U16 / 512;
    static_assert(((NR + 16) * U16) % 512 == 0, "staging units must divide over 512 threads");
    constexpr int SMEM = ((NR + 16) * LDG > 8 * NR * 17) ? (NR + 16) * LDG : 8 * NR * 17;
    __shared__ __attribute__((aligned(16))) double As[SMEM];
    double* Ms = As + NR * LDG;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int j0 = blockIdx.x * 16, r0 = blockIdx.z * NR;
    v4d acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int ch = 0; ch < chunks_per_wg; ++ch) {
        const int cbase = (blockIdx.y * chunks_per_wg + ch) * CHW;
        if (cbase >= D) break;
        v2d st[UPT];
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = q * 512 + tid;
            const int row = u / U16, c16 = u % U16;
            const int col = cbase + 2 * c16;
            const int colc = col < D ? col : 0;
            const double* src;
            bool ok = col < D;
            if (row < NR) {
                const int gr = r0 + row;
                ok = ok && gr < nrows;
                src = A + (size_t)(gr < nrows ? gr : nrows - 1) * lda + colc;
            } else {
                const int gj = j0 + row - NR;                              // (any mrows, round 5: a row beyond the matrix is a clamped
                src = M + (size_t)((RAG && gj >= mrows) ? mrows - 1 : gj) * ldm + colc;   // re-read; it only feeds output columns that are not stored)
            }
            const v2d v = *reinterpret_cast<const v2d*>(src);
            st[q] = ok ? v : (v2d){0.0, 0.0};
        }
        if (ch > 0) __syncthreads();
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = q * 512 + tid;
            *reinterpret_cast<v2d*>(&As[(u / U16) * LDG + 2 * (u % U16)]) = st[q];
        }
        __syncthreads();
        const double* ap = As + c * LDG + RW * w + ks;
        const double* bp = Ms + c * LDG + RW * w + ks;
        double av[MT][NST], bv[NST];
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            bv[s] = bp[4 * s];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt][s] = ap[mt * 16 * LDG + 4 * s];
        }
#pragma unroll
        for (int s = 0; s < NST; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = GSMVI_MFMA_F64(av[mt][s], bv[s], acc[mt]);
    }
    __syncthreads();
    double* red = As;                              // [8][NR][17]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(w * NR + 16 * mt + ks + 4 * r) * 17 + c] = acc[mt][r];
    __syncthreads();
    for (int idx = tid; idx < NR * 16; idx += 512) {
        const int rr = idx >> 4, cc = idx & 15;
        const int row = r0 + rr;
        if (row < nrows && (!RAG || j0 + cc < mrows)) {
            double s = 0.0;
#pragma unroll
            for (int ww = 0; ww < 8; ww += 2) s += red[(ww * NR + rr) * 17 + cc] + red[((ww + 1) * NR + rr) * 17 + cc];
            Pp[((size_t)blockIdx.y * nrows + row) * mrows + j0 + cc] = s;
        }
    }
}

// ---- elementwise stage behind W = G Fm^T: finishes the split-K slabs and builds the D-sized operands -------------
//   w = sum_kc Pp[kc][b];  Rt1 = [Z; V] with V = W + Z (n x D);  top half of Tm1 = [Z; V] Fm, i.e. X - mu.
__global__ __launch_bounds__(256) void k_gsmf_prep(int D, int B, int KC, const double* __restrict__ Z, int ldz,
                                                   const double* __restrict__ X, int ldx,
                                                   const double* __restrict__ mu0, const double* __restrict__ Pp,
                                                   double* __restrict__ Rt1, double* __restrict__ Tm1) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= D) return;
    double pw[GSMVI_MAX_KC];
#pragma unroll
    for (int kc = 0; kc < GSMVI_MAX_KC; ++kc) pw[kc] = Pp[((size_t)(kc < KC ? kc : KC - 1) * B + b) * D + i];
    const double z = Z[(size_t)b * ldz + i], x = X[(size_t)b * ldx + i], m = mu0[i];
    double w = 0.0;
#pragma unroll
    for (int kc = 0; kc < GSMVI_MAX_KC; ++kc) w += (kc < KC) ? pw[kc] : 0.0;
    Rt1[(size_t)b * D + i] = z;
    Rt1[(size_t)(B + b) * D + i] = w + z;
    Tm1[(size_t)b * D + i] = x - m;
}

// ---- Cholesky A = R^T R of one n x n matrix, 64 < n <= 128, in ONE workgroup ----------------------------
// 2 x 2 blocks of 64 on chol64_blk (gsmvi_chol64b.h): the first block row [A11 | A12] is factored AND solved in one call
// (AUG = 2: the columns of A12 ride in the panel waves, [R11 | R12] comes out), A22 -= R12^T R12 runs on the MFMA pipe,
// then A22 is factored.  Round 2 ran chol64_rows_s (one barrier per pivot) with a helper-wave row solve: 46 us per call.
// *info = 1-based index of the first bad pivot (0 = ok); R gets the upper factor with a zero strictly-lower triangle.
// SEMIDEF: the rank-revealing rule for Gram matrices (gsmvi_chol64.h), off when a diagonal entry reaches 2^32.
// ldr: leading dimension of R (n for a stand-alone matrix; the (1, 1) block of an n' x n' factor in the two-level chain);
// info_off > 0: a later diagonal block -- a failure is recorded as info_off + pivot, and only if the earlier blocks passed.
template <bool SEMIDEF>
__global__ __launch_bounds__(512) void k_chol128(int n, const double* __restrict__ A, double* __restrict__ R, int ldr,
                                                 int* info, int info_off) {
    constexpr int ES1 = 146, ES2 = 82;
    __shared__ __attribute__((aligned(16))) double E1[64 * ES1];
    __shared__ __attribute__((aligned(16))) double E2[64 * ES2];
    __shared__ __attribute__((aligned(16))) double scr[CHOLB_SCRATCH_DOUBLES(2)];
    __shared__ int sh_fail[2], sh_moderate;
    const int tid = threadIdx.x, n2 = n - 64;
    if (tid == 0) sh_moderate = 1;
    __syncthreads();
    for (int e0 = tid; e0 < 64 * 128; e0 += 512 * 8) {           // first block row: eight clamped loads per thread in flight
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 512 * u, i = e >> 7, q = e & 127;
            v[u] = A[(size_t)i * n + (q < n ? q : n - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 512 * u, i = e >> 7, q = e & 127;
            double x = (q < n) ? v[u] : 0.0;
            if (q < 64 && q < i) x = 0.0;                        // strictly-lower part of the diagonal block
            if (SEMIDEF && q == i) {
                if (!(x < 4294967296.0)) sh_moderate = 0;
                x -= GSMVI_DEP_TOL * x;
            }
            E1[i * ES1 + q + (q >= 64 ? 0 : 0)] = x;             // columns 64..127 = A12 (zero beyond n)
        }
    }
    for (int e0 = tid; e0 < 64 * 64; e0 += 512 * 8) {            // A22, padded with the identity beyond n2
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 512 * u, i = e >> 6, q = e & 63;
            v[u] = A[(size_t)(64 + (i < n2 ? i : n2 - 1)) * n + 64 + (q < n2 ? q : n2 - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 512 * u, i = e >> 6, q = e & 63;
            double x = (i < n2 && q < n2) ? (q >= i ? v[u] : 0.0) : (i == q ? 1.0 : 0.0);
            if (SEMIDEF && q == i && i < n2) {
                if (!(x < 4294967296.0)) sh_moderate = 0;
                x -= GSMVI_DEP_TOL * x;
            }
            E2[i * ES2 + q] = x;
        }
    }
    __syncthreads();
    const bool moderate = sh_moderate != 0;
    chol64_blk<ES1, SEMIDEF, 2>(E1, scr, 64, &sh_fail[0], moderate);   // [A11 | A12] -> [R11 | R12]
    {   // A22 -= R12^T R12 (upper 16 x 16 blocks, K = 64) on the MFMA pipe
        const int w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
        for (int blk = w; blk < 10; blk += 8) {
            int bi = 0, rem = blk;
            while (rem >= 4 - bi) { rem -= 4 - bi; ++bi; }
            const int bj = bi + rem;
            const double* ap = E1 + ks * ES1 + 64 + 16 * bi + c;
            const double* bp = E1 + ks * ES1 + 64 + 16 * bj + c;
            double a[16], b[16], tv[4];
#pragma unroll
            for (int st = 0; st < 16; ++st) { a[st] = ap[4 * st * ES1]; b[st] = bp[4 * st * ES1]; }
            double* tp = E2 + (16 * bi + ks) * ES2 + 16 * bj + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) tv[r] = tp[4 * r * ES2];
            v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int st = 0; st < 16; st += 2) {
                acc0 = GSMVI_MFMA_F64(a[st], b[st], acc0);
                acc1 = GSMVI_MFMA_F64(a[st + 1], b[st + 1], acc1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * bi + ks + 4 * r, j = 16 * bj + c;
                if (j >= i) tp[4 * r * ES2] = tv[r] - (acc0[r] + acc1[r]);
            }
        }
    }
    __syncthreads();
    chol64_blk<ES2, SEMIDEF, 0>(E2, scr, n2, &sh_fail[1], moderate);
    for (int e = tid; e < n * n; e += 512) {
        const int i = e / n, j = e % n;
        double x = 0.0;
        if (i < 64) x = (j >= i) ? E1[i * ES1 + j] : 0.0;        // [R11 | R12]: columns 64.. sit at E1[:, 64 + (j - 64)]
        else if (j >= i) x = E2[(i - 64) * ES2 + (j - 64)];
        R[(size_t)i * ldr + j] = x;
    }
    if (tid == 0) {
        const int f = sh_fail[0] != 0 ? sh_fail[0] : (sh_fail[1] != 0 ? 64 + sh_fail[1] : 0);
        if (info_off == 0) *info = f;
        else if (f != 0 && *info == 0) *info = info_off + f;
    }
}

// ---- A = R^T R and W = R^-T (lower triangular), 64 < n <= 128, one workgroup: chol128w_body (gsmvi_chol128.h) -----------------
template <bool SEMIDEF>
__global__ __launch_bounds__(512) void k_chol128w(int n, const double* __restrict__ A, double* __restrict__ R,
                                                  double* __restrict__ Wo, int* __restrict__ info) {
    chol128w_body<SEMIDEF>(n, A, n, R, n, Wo, n, info);
}

// ---- one DIAGONAL BLOCK (nb <= 128 rows) of the two-level scheme for 128 < n <= 256: [A_kk | I] -> [R_kk | W_kk], leading
// dimensions for every operand (the blocks live inside the n x n matrices).  BIG: 64 < nb <= 128 (chol128w_body), else one
// chol64_blk call.  info chaining: info_off == 0 writes *info; a later block (info_off = its first row) only records a failure
// when the earlier ones passed.  SEMIDEF: the magnitude guard of the rank-revealing rule looks at the diagonal of the WHOLE
// matrix (dg, dg_n entries, stride dg_stride), as chol128w_body does for its two halves; tol_applied: the caller's product has
// already lowered this block's diagonal by its rounding floor (OpBlkS22).
template <bool SEMIDEF, bool BIG>
__global__ __launch_bounds__(512) void k_cholw_ld(int nb, const double* __restrict__ A, int lda, double* __restrict__ R, int ldr,
                                                  double* __restrict__ Wo, int ldw, int* info, int info_off, int tol_applied,
                                                  const double* __restrict__ dg, int dg_n, int dg_stride) {
    __shared__ int sh_mod, sh_f;
    const int tid = threadIdx.x;
    int moderate_ext = -1;
    if (SEMIDEF && dg) {
        if (tid == 0) sh_mod = 1;
        __syncthreads();
        for (int i = tid; i < dg_n; i += 512)
            if (!(dg[(size_t)i * dg_stride] < 4294967296.0)) sh_mod = 0;
        __syncthreads();
        moderate_ext = sh_mod;
    }
    if constexpr (BIG) {
        chol128w_body<SEMIDEF>(nb, A, lda, R, ldr, Wo, ldw, &sh_f, nullptr, tol_applied != 0, moderate_ext);
    } else {
        constexpr int ES = 146;
        __shared__ __attribute__((aligned(16))) double E[64 * ES];
        __shared__ __attribute__((aligned(16))) double scr1[CHOLB_SCRATCH_DOUBLES(1)];
        __shared__ int sh_moderate;
        if (tid == 0) sh_moderate = 1;
        __syncthreads();
        for (int e = tid; e < 64 * 64; e += 512) {
            const int i = e >> 6, j = e & 63;
            const bool in = i < nb && j < nb;
            double x = in ? (j >= i ? A[(size_t)i * lda + j] : 0.0) : (i == j ? 1.0 : 0.0);
            if (SEMIDEF && in && i == j) {
                if (!(x < 4294967296.0)) sh_moderate = 0;
                if (!tol_applied) x -= GSMVI_DEP_TOL * x;
            }
            E[i * ES + j] = x;
        }
        __syncthreads();
        const bool moderate = moderate_ext < 0 ? sh_moderate != 0 : moderate_ext != 0;
        chol64_blk<ES, SEMIDEF, 1>(E, scr1, nb, &sh_f, moderate);
        for (int e = tid; e < nb * nb; e += 512) {
            const int i = e / nb, j = e - i * nb;
            R[(size_t)i * ldr + j] = (j >= i) ? E[i * ES + j] : 0.0;
            Wo[(size_t)i * ldw + j] = (j <= i) ? E[i * ES + 64 + j] : 0.0;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int f = sh_f;
        if (info_off == 0) *info = f;
        else if (f != 0 && *info == 0) *info = info_off + f;
    }
}

// ---- two diagonal-block jobs in ONE launch (64 < nb <= 128 each): workgroup 0 = job a under the rank-revealing rule,
// workgroup 1 = job b under the plain rule.  Used by the two-level chain (factor_chain_big): [S22 | I] -> [R22 | W22] of Gamma
// beside [A'11 | I] -> [T11 | W_T11] -- A'11 = I + (Rg J Rg^T)_11 needs only the first block row [R11 R12] of Rg.
__global__ __launch_bounds__(512) void k_cholw_pair(cholw_job a, cholw_job b) {
    CHOL128W_LDS(E1, B12, scr, sh_fail, sh_moderate);
    __shared__ int sh_mod, sh_f;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        int moderate_ext = -1;
        if (a.dg) {
            if (tid == 0) sh_mod = 1;
            __syncthreads();
            for (int i = tid; i < a.dg_n; i += 512)
                if (!(a.dg[(size_t)i * a.dg_stride] < 4294967296.0)) sh_mod = 0;
            __syncthreads();
            moderate_ext = sh_mod;
        }
        chol128w_core<true>(E1, B12, scr, sh_fail, &sh_moderate, a.nb, a.A, a.lda, a.R, a.ldr, a.W, a.ldw, &sh_f, nullptr,
                            a.tol_applied != 0, moderate_ext);
    } else {
        chol128w_core<false>(E1, B12, scr, sh_fail, &sh_moderate, b.nb, b.A, b.lda, b.R, b.ldr, b.W, b.ldw, &sh_f);
    }
    __syncthreads();
    if (tid == 0) {
        const int f = sh_f;
        int* info = blockIdx.x == 0 ? a.info : b.info;
        const int off = blockIdx.x == 0 ? a.info_off : b.info_off;
        if (off == 0) *info = f;
        else if (f != 0 && *info == 0) *info = off + f;
    }
}

// ---- F = F0 + Rt^T Fs  (full, non-symmetric rank-n update; F = F0 when *bad) ---------------------------
__global__ __launch_bounds__(256) void k_gsmf_update(int D, int KF, const double* __restrict__ Ft,
                                                     const double* __restrict__ Fs, const double* __restrict__ F0,
                                                     int ldf0, double* __restrict__ F, int ldf,
                                                     const int* __restrict__ bad, int tj0, int ntc) {
    // (tj0, ntc: the window of column tiles this launch updates -- all of them, or the owned column block of a column-sharded fit)
    constexpr int RS = 66;
    __shared__ double FA[64 * RS];
    __shared__ double FB[64 * RS];
    const int ti = blockIdx.x / ntc, tj = tj0 + blockIdx.x % ntc;
    const int I0 = ti * 64, J0 = tj * 64;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int wr = w >> 1, wc = w & 1;
    const int skip = *bad;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int kb = 0; kb < KF && !skip; kb += 64) {
        double va[16], vb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int p = kb + (tid >> 6) + 4 * q, i = tid & 63;
            const int pc = p < KF ? p : KF - 1;
            const int gi = I0 + i, gj = J0 + i;
            const double a = Ft[(size_t)pc * D + (gi < D ? gi : D - 1)];
            const double b = Fs[(size_t)pc * D + (gj < D ? gj : D - 1)];
            va[q] = (p < KF && gi < D) ? a : 0.0;
            vb[q] = (p < KF && gj < D) ? b : 0.0;
        }
        if (kb > 0) __syncthreads();
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
                if (row < D && col < D) F[(size_t)row * ldf + col] = F0[(size_t)row * ldf0 + col] + acc[rt][ct][r];
            }
}

// ---- fast rank-n update (D % 64 == 0, any even n <= 32 NP): F = F0 + Rt^T Fs, the new mean, the revert passthrough ---
// One 512-thread workgroup per 64 x 64 tile (two waves per SIMD for the fp64 MFMA rate); wave w owns the
// 16-row block w >> 1 and the two 16-column blocks of half w & 1.  Every global load of the workgroup -- the F0
// tile in accumulator layout and all n rows of both operand tiles -- is issued in one batch; the operand rows
// then pass through LDS 32 at a time, [k][80] (64 columns + 16 pad: both MFMA operand reads conflict-free).
// Tile row 0 also writes mu = mu0 + sum_b coef[b] Tm[b] (coef = [beta; alpha] / B from the small-matrix kernel: the mean of
// the rows u_b Fm) for its 64 columns, and workgroup 0 counts the revert.  When *bad (the 2B x 2B positive-definite test failed) F = F0 and mu = mu0.
template <int NP, bool RAG>               // RAG: D % 64 != 0 or 2B % 32 != 0 (edge tiles, partial staging pass; round 5)
__global__ __launch_bounds__(512) void k_gsmf_update_fast(int D, int B, const double* __restrict__ Rt,
                                                          const double* __restrict__ Fs,
                                                          const double* __restrict__ F0, int ldf0,
                                                          double* __restrict__ F, int ldf,
                                                          const double* __restrict__ Tm,
                                                          const double* __restrict__ coef,
                                                          const double* __restrict__ mu0, double* __restrict__ mu,
                                                          const int* __restrict__ bad, int* __restrict__ n_reverts, int tj0, int ntc) {
    constexpr int RS = 80, KP = 32;
    __shared__ __attribute__((aligned(16))) double sm[2 * KP * RS];
    // any even D (round 5): edge tiles re-read clamped rows / columns, store what lies inside.  (tj0, ntc): window of column tiles (round 6)
    const int ti = blockIdx.x / ntc, tj = tj0 + blockIdx.x % ntc;
    const int I0 = ti * 64, J0 = tj * 64;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int wr = w >> 1, wc = w & 1;
    // ---- all global loads ----
    double f0[2][4];
    const int frow = I0 + 16 * wr + ks;
    const int fcol = J0 + 32 * wc + c;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = (RAG && frow + 4 * r >= D) ? D - 1 : frow + 4 * r, cq = (RAG && fcol + 16 * blk >= D) ? D - 1 : fcol + 16 * blk;
            f0[blk][r] = F0[(size_t)rr * ldf0 + cq];
        }
    // (any n = 2B <= 32 NP, round 5: rows beyond n are clamped re-reads, zeroed when they are staged; a pass whose rows
    // all lie beyond n is skipped -- np is block-uniform)
    const int n = 2 * B, np = RAG ? (n + KP - 1) / KP : NP;    // (on the grid n = 32 NP: the pass count is a compile-time constant)
    v2d ga[NP][2], gb[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
        if (p < np) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int u = q * 512 + tid, row = KP * p + (u >> 5), c2 = 2 * (u & 31);
                const int rc = (RAG && row >= n) ? n - 1 : row;
                ga[p][q] = *reinterpret_cast<const v2d*>(Rt + (size_t)rc * D + ((RAG && I0 + c2 >= D) ? D - 2 : I0 + c2));
                gb[p][q] = *reinterpret_cast<const v2d*>(Fs + (size_t)rc * D + ((RAG && J0 + c2 >= D) ? D - 2 : J0 + c2));
            }
        }
    const int skip = *bad;
    double msum = 0.0;
    if (ti == 0) {                               // partial weighted column sums of rows g, g + 8, ... of Tm = [X - mu; V Fm]
        const int g = tid >> 6, col = (RAG && J0 + (tid & 63) >= D) ? D - 1 : J0 + (tid & 63);
        for (int b = g; b < 2 * B; b += 8) msum += coef[b] * Tm[(size_t)b * D + col];
    }
    if (blockIdx.x == 0 && tid == 0 && skip && n_reverts) *n_reverts += 1;
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p < np) {
            if (p > 0) __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int u = q * 512 + tid, row = u >> 5, c2 = 2 * (u & 31);
                const bool in = !RAG || KP * p + row < n;
                *reinterpret_cast<v2d*>(sm + row * RS + c2) = in ? ga[p][q] : (v2d){0.0, 0.0};
                *reinterpret_cast<v2d*>(sm + (KP + row) * RS + c2) = in ? gb[p][q] : (v2d){0.0, 0.0};
            }
            __syncthreads();
            double a[8], b0[8], b1[8];
            const double* ap = sm + ks * RS + 16 * wr + c;
            const double* bp = sm + (KP + ks) * RS + 32 * wc + c;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                a[s] = ap[4 * s * RS];
                b0[s] = bp[4 * s * RS];
                b1[s] = bp[4 * s * RS + 16];
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
        if (!RAG || (frow + 4 * r < D && fcol < D)) F[(size_t)(frow + 4 * r) * ldf + fcol] = skip ? f0[0][r] : f0[0][r] + acc0[r];
        if (!RAG || (frow + 4 * r < D && fcol + 16 < D)) F[(size_t)(frow + 4 * r) * ldf + fcol + 16] = skip ? f0[1][r] : f0[1][r] + acc1[r];
    }
    if (ti == 0) {
        __syncthreads();
        sm[tid] = msum;                          // [8][64]
        __syncthreads();
        if (tid < 64 && (!RAG || J0 + tid < D)) {
            double s = 0.0;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += sm[g * 64 + tid];
            mu[J0 + tid] = skip ? mu0[J0 + tid] : mu0[J0 + tid] + s;
        }
    }
}

// ---- everything small in ONE workgroup (n = 2B <= 64): gsmf_small16_body (gsmvi_small16.h) --------------------------------
// Stand-alone launch of the chain; the lean path runs the same body as a RIDER workgroup of the V Fm panel product instead
// (gsmvi_fast.hip, k_panel_fast<.., RIDER>), so that the product and its launch boundary hide behind the chain.
__global__ __launch_bounds__(512) void k_gsmf_small16(int n, int B, const double* __restrict__ Gp, int kcg,
                                                      double* __restrict__ Kmat, double* __restrict__ coef,
                                                      int* __restrict__ bad_out,
                                                      unsigned long long* __restrict__ stamps, int jmode,
                                                      const int* __restrict__ prior_bad, const double* __restrict__ Pi,
                                                      const double* __restrict__ R11g, const double* __restrict__ W11g) {
    __shared__ __attribute__((aligned(16))) double lds[GSMF_SMALL16_LDS];
    gsmf_small16_body(lds, n, B, Gp, kcg, Kmat, coef, bad_out, stamps, jmode, prior_bad, Pi, R11g, W11g);
}

// ---- K'' = (W S)^T (T - I) (W S), W = Rg^-T, for n > 64: W comes out of the Gram matrix's factorisation itself (k_chol128w /
// k_cholw_ld); the products P = (T - I)(W S) and K'' = (W S)^T P -- with the column operation W S applied while W is loaded and
// the chain's accept / revert decision taken by the first of them -- are OpChainP / OpChainK of gsmvi_smallgemm.h.

// ---- new mean and the revert passthrough for the K-matrix path -----------------------------------------
__global__ __launch_bounds__(256) void k_gsmf_mean(int D, int B, const double* __restrict__ Tm,
                                                   const double* __restrict__ coef,
                                                   const double* __restrict__ mu0, double* __restrict__ mu,
                                                   const int* __restrict__ bad, int* __restrict__ n_reverts, int j0, int j1) {
    const int j = j0 + blockIdx.x * 256 + threadIdx.x;     // (j0 .. j1: the window of columns: all of them, or the owned block)
    if (j >= j1) return;
    if (j == j0 && n_reverts && *bad) *n_reverts += 1;     // one launch per update on the stream: no race
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int n = 2 * B;
    int b = 0;
    for (; b + 8 <= n; b += 8) {                   // eight independent loads in flight per trip
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = coef[b + k] * Tm[(size_t)(b + k) * D + j];
        s0 += v[0] + v[4];
        s1 += v[1] + v[5];
        s2 += v[2] + v[6];
        s3 += v[3] + v[7];
    }
    for (; b < n; ++b) s0 += coef[b] * Tm[(size_t)b * D + j];
    mu[j] = (*bad) ? mu0[j] : mu0[j] + ((s0 + s1) + (s2 + s3));
}

// ---- rank-n update with the skinny product folded in (n = 32 NP <= 64, D % 64 == 0): F = F0 + Rt1^T (K'' Tm1) ----------------
// k_gsmf_update_fast needs Fs = K'' Tm1 from a launch of its own (a 64-strip product, ~6 us + a launch boundary at D = 1024).
// Here every 64 x 64 tile workgroup forms ITS 64 columns of Fs itself: K'' (n x n) and the tile's columns of Tm1 = [X - mu; V Fm]
// go to LDS (the V Fm rows summed from the split-K slabs of that product while they are loaded), Fs_tile = K'' Tm1_tile is 32
// MFMAs per wave, and the rank-n update reads it from LDS.  The product is repeated by the D / 64 workgroups of a tile
// column -- 16 x 0.5 MFLOP at D = 1024, cheap beside a dependent launch.  Tile row 0 also writes the mean, workgroup 0 counts
// the revert; *bad => F = F0, mu = mu0.
// n = 2B may be smaller than N = 32 NP (n = 16, BASELINE config 2): rows and columns beyond n are loaded as zeros.
// KCB: compile-time bound of the V Fm slab count (every 16-byte unit costs KCB loads).
template <int NP, int KCB, bool RAG>      // RAG: D % 64 != 0 (edge tiles; round 5)
__global__ __launch_bounds__(512) void k_gsmf_update_fs(int D, int B, const double* __restrict__ Rt,
                                                        const double* __restrict__ Kmat, const double* __restrict__ Tm,
                                                        const double* __restrict__ vf_slabs, int kcv,
                                                        const double* __restrict__ F0, int ldf0, double* __restrict__ F,
                                                        int ldf, const double* __restrict__ coef,
                                                        const double* __restrict__ mu0, double* __restrict__ mu,
                                                        const int* __restrict__ bad, int* __restrict__ n_reverts,
                                                        gsmf_bam_mean bm, int tj0, int ntc) {
    constexpr int N = 32 * NP, RS = 80, KS = N + 2;
    __shared__ __attribute__((aligned(16))) double bufA[N * RS];      // Tm1 tile [k][64 cols], later the Rt1 tile
    __shared__ __attribute__((aligned(16))) double bufB[N * RS];      // Fs tile
    __shared__ __attribute__((aligned(16))) double bufK[N * KS];      // K''
    __shared__ double msm[8 * 64];
    // any even D (round 5): edge tiles as in k_gsmf_update_fast; (tj0, ntc): window of column tiles (round 6)
    const int ti = blockIdx.x / ntc, tj = tj0 + blockIdx.x % ntc;
    const int I0 = ti * 64, J0 = tj * 64;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int wr = w >> 1, wc = w & 1;
    const int skip = *bad;
    if (blockIdx.x == 0 && tid == 0 && skip && n_reverts) *n_reverts += 1;
    // ---- all global loads of the first phase: F0 tile (accumulator layout), K'', the Tm1 tile, the Rt1 tile ----
    double f0[2][4];
    const int frow = I0 + 16 * wr + ks;
    const int fcol = J0 + 32 * wc + c;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = (RAG && frow + 4 * r >= D) ? D - 1 : frow + 4 * r, cq = (RAG && fcol + 16 * blk >= D) ? D - 1 : fcol + 16 * blk;
            f0[blk][r] = F0[(size_t)rr * ldf0 + cq];
        }
    constexpr int KU = N * N / 2 / 512;            // 16-B units of K'' per thread (N = 64: 4, N = 32: 1)
    const int n = 2 * B;
    v2d gk[KU];
#pragma unroll
    for (int q = 0; q < KU; ++q) {
        const int e = (q * 512 + tid) * 2, i = e / N, j = e % N;
        gk[q] = (i < n && j < n) ? *reinterpret_cast<const v2d*>(Kmat + (size_t)i * n + j) : (v2d){0.0, 0.0};
    }
    constexpr int TU = N * 32 / 512;               // 16-B units of an N x 64 tile per thread (N = 64: 4)
    v2d gt[TU], gr[TU];
#pragma unroll
    for (int q = 0; q < TU; ++q) {
        const int u = q * 512 + tid, row = u >> 5, c2 = 2 * (u & 31);
        const int jc = (RAG && J0 + c2 >= D) ? D - 2 : J0 + c2, icq = (RAG && I0 + c2 >= D) ? D - 2 : I0 + c2;
        if (row >= n) {
            gt[q] = gr[q] = (v2d){0.0, 0.0};
            continue;
        }
        if (row >= B && vf_slabs != nullptr) {     // V Fm rows: sum of the kcv slabs (row is wave-uniform for B % 2 == 0)
            v2d t[KCB];
#pragma unroll
            for (int kq = 0; kq < KCB; ++kq)
                t[kq] = *reinterpret_cast<const v2d*>(vf_slabs + ((size_t)(kq < kcv ? kq : kcv - 1) * B + (row - B)) * D + jc);
            v2d a = {0.0, 0.0};
#pragma unroll
            for (int kq = 0; kq < KCB; ++kq)
                if (kq < kcv) { a.x += t[kq].x; a.y += t[kq].y; }
            gt[q] = a;
        } else {
            gt[q] = *reinterpret_cast<const v2d*>(Tm + (size_t)row * D + jc);
        }
        gr[q] = *reinterpret_cast<const v2d*>(Rt + (size_t)row * D + icq);
    }
#pragma unroll
    for (int q = 0; q < KU; ++q) {
        const int e = (q * 512 + tid) * 2, i = e / N, j = e % N;
        bufK[i * KS + j] = gk[q].x;
        bufK[i * KS + j + 1] = gk[q].y;
    }
#pragma unroll
    for (int q = 0; q < TU; ++q) {
        const int u = q * 512 + tid, row = u >> 5, c2 = 2 * (u & 31);
        *reinterpret_cast<v2d*>(bufA + row * RS + c2) = gt[q];
    }
    __syncthreads();
    double msum = 0.0;
    if (ti == 0) {                                 // weighted column sums of Tm1: the mean of the rows u_b Fm
        const int g = tid >> 6, col = tid & 63;
        for (int b = g; b < n; b += 8) msum += coef[b] * bufA[b * RS + col];
    }
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    if (!skip) {
        // Fs_tile = K'' Tm1_tile: wave (wr, wc) -> rows 16 wr.., columns 32 wc.. (N = 32: only wr < 2 has rows)
        if (16 * wr < N) {
            double a[N / 4], b0[N / 4], b1[N / 4];
            const double* ap = bufK + (16 * wr + c) * KS + ks;
            const double* bp = bufA + ks * RS + 32 * wc + c;
#pragma unroll
            for (int st = 0; st < N / 4; ++st) {
                a[st] = ap[4 * st];
                b0[st] = bp[4 * st * RS];
                b1[st] = bp[4 * st * RS + 16];
            }
            v4d p0 = {0.0, 0.0, 0.0, 0.0}, p1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int st = 0; st < N / 4; ++st) {
                p0 = GSMVI_MFMA_F64(a[st], b0[st], p0);
                p1 = GSMVI_MFMA_F64(a[st], b1[st], p1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bufB[(16 * wr + ks + 4 * r) * RS + 32 * wc + c] = p0[r];
                bufB[(16 * wr + ks + 4 * r) * RS + 32 * wc + c + 16] = p1[r];
            }
        }
    }
    __syncthreads();                               // Fs tile complete; every read of the Tm1 tile is done
#pragma unroll
    for (int q = 0; q < TU; ++q) {
        const int u = q * 512 + tid, row = u >> 5, c2 = 2 * (u & 31);
        *reinterpret_cast<v2d*>(bufA + row * RS + c2) = gr[q];
    }
    __syncthreads();
    if (!skip) {
        double a[N / 4], b0[N / 4], b1[N / 4];
        const double* ap = bufA + ks * RS + 16 * wr + c;
        const double* bp = bufB + ks * RS + 32 * wc + c;
#pragma unroll
        for (int st = 0; st < N / 4; ++st) {
            a[st] = ap[4 * st * RS];
            b0[st] = bp[4 * st * RS];
            b1[st] = bp[4 * st * RS + 16];
        }
#pragma unroll
        for (int st = 0; st < N / 4; ++st) {
            acc0 = GSMVI_MFMA_F64(a[st], b0[st], acc0);
            acc1 = GSMVI_MFMA_F64(a[st], b1[st], acc1);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (!RAG || (frow + 4 * r < D && fcol < D)) F[(size_t)(frow + 4 * r) * ldf + fcol] = skip ? f0[0][r] : f0[0][r] + acc0[r];
        if (!RAG || (frow + 4 * r < D && fcol + 16 < D)) F[(size_t)(frow + 4 * r) * ldf + fcol + 16] = skip ? f0[1][r] : f0[1][r] + acc1[r];
    }
    if (ti == 0) {
        msm[tid] = msum;                           // [8][64]
        __syncthreads();
        if (tid < 64 && (!RAG || J0 + tid < D)) {
            double s = 0.0;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += msm[g * 64 + tid];
            if (bm.xbar) {                         // factor-form BaM (bam.py:112): mu0/(1+reg) + r1 (S gbar) + r1 xbar
                const double reg = bm.reg.get(), r1 = reg / (1.0 + reg);
                mu[J0 + tid] = skip ? mu0[J0 + tid] : mu0[J0 + tid] / (1.0 + reg) + bm.sg_r1[J0 + tid] + r1 * bm.xbar[J0 + tid];
            } else
                mu[J0 + tid] = skip ? mu0[J0 + tid] : mu0[J0 + tid] + s;
        }
    }
}

// ---- 64 < n <= 128: Gamma = S Gamma1 S^T and the per-sample coefficients from the Gram slabs ---------------------------
// 256 elements of Gamma per workgroup; every workgroup derives the B coefficient pairs itself (3 kcg loads per sample)
template <int KCT>   // compile-time bound of the slab count: every entry of Gamma1 costs KCT loads
__global__ __launch_bounds__(256) void k_gsmf_gamma_big(int n, int B, const double* __restrict__ Gp, int kcg,
                                                        double* __restrict__ Gam, double* __restrict__ coef,
                                                        double* __restrict__ ab, int jmode,
                                                        const double* __restrict__ R11g = nullptr,
                                                        const double* __restrict__ W11g = nullptr,
                                                        double* __restrict__ Rg = nullptr, double* __restrict__ Wm = nullptr) {
    // R11g (jmode 2, 2B <= 128; round 5): the Gram matrix is block diagonal in the orthogonal basis and its first block comes
    // factored ([R11 | W11], B x B compact).  This launch then also lays out everything of Rg and W = Rg^-T that is not the second
    // diagonal block: the given block and the two zero blocks -- the factorisation that follows takes the second block alone.
    __shared__ double s_alpha[128], s_beta[128];       // B <= 128 (n = 2B <= 256 since round 4)
    const int tid = threadIdx.x;
    auto g1 = [&](int i, int q) {                      // one entry of Gamma1: the kcg slabs, all loads in one batch
        if (kcg == 1) return Gp[(size_t)i * n + q];    // block-uniform: the finished matrix
        double t[KCT];
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc) t[kc] = Gp[(size_t)(kc < kcg ? kc : kcg - 1) * n * n + (size_t)i * n + q];
        double a = 0.0;
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc) a += (kc < kcg) ? t[kc] : 0.0;
        return a;
    };
    if (tid < B) {
        double al = 1.0, be = 0.0;
        if (!jmode) gsmf_coefs(g1(tid, tid), g1(tid, B + tid), g1(B + tid, B + tid), &al, &be);
        s_alpha[tid] = al;
        s_beta[tid] = be;
        if (blockIdx.x == 0) {
            coef[tid] = jmode ? 0.0 : be / (double)B;
            coef[B + tid] = jmode ? 0.0 : al / (double)B;
            ab[tid] = be;                              // for OpChainP / OpChainK (the column operation W S)
            ab[B + tid] = al;
        }
    }
    __syncthreads();
    const int e = blockIdx.x * 256 + tid;
    if (e >= n * n) return;
    const int i = e / n, q = e % n;
    double v;
    if (i < B && q < B) v = g1(i, q);
    else if (i < B) v = s_beta[q - B] * g1(i, q - B) + s_alpha[q - B] * g1(i, q);
    else if (q < B) v = s_beta[i - B] * g1(i - B, q) + s_alpha[i - B] * g1(i, q);
    else {
        const int a = i - B, b = q - B;
        v = s_beta[a] * (s_beta[b] * g1(a, b) + s_alpha[b] * g1(a, q)) +
            s_alpha[a] * (s_beta[b] * g1(i, b) + s_alpha[b] * g1(i, q));
    }
    Gam[e] = v;
    if (R11g && (i < B || q < B)) {
        const bool first = i < B && q < B;
        Rg[e] = (first && q >= i) ? R11g[(size_t)i * B + q] : 0.0;
        Wm[e] = (first && q <= i) ? W11g[(size_t)i * B + q] : 0.0;
    }
}

int gsmvi_potrf_impl(gsmvi_ctx* ctx, hipStream_t st, int D, const double* S, int lds, double* R, int ldr,
                     int* info_dev);

// diagnostic (include/gsmvi_hip_debug.h): the one-workgroup n x n Cholesky kernels of the 2B x 2B chain on caller data
extern "C" int gsmvi_debug_chol128(void* stream, int n, int with_inverse, const double* A, double* R, double* W, int* info_dev) {
    if (n <= 64 || n > 128 || !A || !R || !info_dev || (with_inverse && !W)) {
        gsmvi_set_error("%s: %s", "gsmvi_debug_chol128", "bad argument (64 < n <= 128)");
        return GSMVI_ERR_BAD_ARG;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (with_inverse) hipLaunchKernelGGL(k_chol128w<true>, dim3(1), dim3(512), 0, st, n, A, R, W, info_dev);
    else hipLaunchKernelGGL(k_chol128<false>, dim3(1), dim3(512), 0, st, n, A, R, n, info_dev, 0);
    return hipGetLastError() == hipSuccess ? GSMVI_OK : GSMVI_ERR_HIP;
}

static int chk(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gsmvi_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}

// Pp[kc][B][mrows] partial slabs of A M^T (A: B x D, M: mrows x D); *kc_out slabs.
static int gsmvi_panel_t_product_mt(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* A, int lda, const double* M,
                                    int ldm, int mrows, double* Pp, int* kc_out, int mt_cap);
int gsmvi_panel_t_product(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* A, int lda, const double* M,
                          int ldm, int mrows, double* Pp, int* kc_out) {
    return gsmvi_panel_t_product_mt(ctx, st, D, B, A, lda, M, ldm, mrows, Pp, kc_out, 4);
}
// the same with 32-row blocks at most: half the split-K slabs at the same number of workgroups -- for consumers that sum the
// slabs while they load their operands (k_bam_nmat2)
int gsmvi_panel_t_product_few_slabs(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* A, int lda, const double* M,
                                    int ldm, int mrows, double* Pp, int* kc_out) {
    return gsmvi_panel_t_product_mt(ctx, st, D, B, A, lda, M, ldm, mrows, Pp, kc_out, ctx->tune_gram_mt >= 4 ? 4 : 2);
}
// mt_cap: largest row-block multiple (16 mt_cap rows per workgroup).  The chunk width is 256 columns for mt <= 2 and 128 for
// mt = 4, so capping at 2 halves the number of split-K slabs a small product leaves behind (4 instead of 8 at D = 1024) at
// the same number of workgroups -- what the one-workgroup consumer of the Gram slabs wants.
static int gsmvi_panel_t_product_mt(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* A, int lda, const double* M,
                                    int ldm, int mrows, double* Pp, int* kc_out, int mt_cap) {
    int MT = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    if (MT > mt_cap) MT = mt_cap;
    const int CH = (MT == 4) ? 128 : 256;
    // (any even D since round 5: the staging of both operands zeroes what lies beyond column D)
    const bool fast_t = !ctx->tune_no_fast && D % 2 == 0 && lda % 2 == 0 && ldm % 2 == 0 &&
                        (reinterpret_cast<uintptr_t>(A) & 15u) == 0 && (reinterpret_cast<uintptr_t>(M) & 15u) == 0;
    if (fast_t && MT == 4 && ctx->tune_wide && mrows % 64 == 0 && mrows >= 1024 && D >= 1024 && D % 64 == 0) {
        // 64-row panels of a D-sized product are MFMA-bound: the 64 x 64-tile kernel (gsmvi_wide.hip)
        int kcw = 1, kper = D;
        gsmvi_panel_wide_split(D, (mrows / 64) * ((B + 63) / 64), ctx->num_cu, ctx->tune_wide_kc, &kcw, &kper);
        *kc_out = kcw;
        gsmvi_launch_panel_wide(st, nullptr, true, D, B, A, lda, nullptr, 1.0, M, ldm, Pp, kper, kcw, mrows);
        ctx->path |= GSMVI_PATH_PANEL_WIDE;
        return chk("k_panel_wide");
    }
    const int strips = (mrows + 15) / 16, nchunks = (D + CH - 1) / CH, zb = (B + 16 * MT - 1) / (16 * MT);
    int kc = (2 * ctx->num_cu + strips * zb - 1) / (strips * zb);
    if (kc > nchunks) kc = nchunks;
    if (kc > GSMVI_MAX_KC) kc = GSMVI_MAX_KC;
    if (kc < 1) kc = 1;
    const int cpw = (nchunks + kc - 1) / kc;
    kc = (nchunks + cpw - 1) / cpw;
    *kc_out = kc;
    const dim3 grid(strips, kc, zb);
    ctx->path |= fast_t ? GSMVI_PATH_PANEL_T_FAST : GSMVI_PATH_PANEL_T_GENERIC;
    if (fast_t && mrows % 16 != 0) {
        if (MT == 1) hipLaunchKernelGGL((k_panel_t_fast<1, 256, true>), grid, dim3(512), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
        else if (MT == 2) hipLaunchKernelGGL((k_panel_t_fast<2, 256, true>), grid, dim3(512), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
        else hipLaunchKernelGGL((k_panel_t_fast<4, 128, true>), grid, dim3(512), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
    } else if (fast_t) {
        if (MT == 1) hipLaunchKernelGGL((k_panel_t_fast<1, 256, false>), grid, dim3(512), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
        else if (MT == 2) hipLaunchKernelGGL((k_panel_t_fast<2, 256, false>), grid, dim3(512), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
        else hipLaunchKernelGGL((k_panel_t_fast<4, 128, false>), grid, dim3(512), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
    } else if (MT == 1) hipLaunchKernelGGL((k_panel_t<1, 256>), grid, dim3(256), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
    else if (MT == 2) hipLaunchKernelGGL((k_panel_t<2, 256>), grid, dim3(256), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
    else hipLaunchKernelGGL((k_panel_t<4, 128>), grid, dim3(256), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
    return chk("k_panel_t");
}

// ---- records of the batch-sharded factor path: rec[b] = [ x_b - mu (D) | v_b (D) | (v Fm)_b (D) ], v = w + z ---------
__global__ __launch_bounds__(256) void k_gsmf_pack(int D, int Bl, const double* __restrict__ Rt,
                                                   const double* __restrict__ Tm, double* __restrict__ rec,
                                                   int ldrec) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= D) return;
    double* rb = rec + (size_t)b * ldrec;
    rb[i] = Tm[(size_t)b * D + i];
    rb[D + i] = Rt[(size_t)(Bl + b) * D + i];
    rb[2 * D + i] = Tm[(size_t)(Bl + b) * D + i];
}

// Rt1 = [Z; V] and Tm1 = [X - mu; V Fm] from the replicated draws and ALL records
__global__ __launch_bounds__(256) void k_gsmf_unpack(int D, int B, const double* __restrict__ Z, int ldz,
                                                     const double* __restrict__ rec, int ldrec,
                                                     double* __restrict__ Rt, double* __restrict__ Tm) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= D) return;
    const double* rb = rec + (size_t)b * ldrec;
    Rt[(size_t)b * D + i] = Z[(size_t)b * ldz + i];
    Rt[(size_t)(B + b) * D + i] = rb[D + i];
    Tm[(size_t)b * D + i] = rb[i];
    Tm[(size_t)(B + b) * D + i] = rb[2 * D + i];
}

// Workspace of the factor path: ctx->sg = Rt1 [Z; V] (n x D) | Tm1 [X - mu; V Fm] (n x D) | Fs (n x D); ctx->small = seven
// n x n slots (Gamma, Rg / K'', A', T, P, coefficients, finished Gram matrix); ctx->gram_slabs; ctx->pp = split-K slabs of
// the D-wide products (W, then V Fm).
struct factor_ws {
    double *Rt, *Tm, *Fs, *Gam, *Rg, *Ap, *Tt, *Pm, *coef, *Gam1, *Gp;
    unsigned long long* stamps;
};
static factor_ws factor_carve(gsmvi_ctx* ctx, int D, int n) {
    factor_ws w;
    w.Rt = ctx->fo_Rt ? ctx->fo_Rt : ctx->sg;
    w.Tm = ctx->fo_Tm ? ctx->fo_Tm : w.Rt + (size_t)n * D;
    w.Fs = ctx->fo_Fs ? ctx->fo_Fs : w.Tm + (size_t)n * D;
    w.Gam = ctx->small;
    w.Rg = w.Gam + (size_t)n * n;
    w.Ap = w.Rg + (size_t)n * n;
    w.Tt = w.Ap + (size_t)n * n;
    w.Pm = w.Tt + (size_t)n * n;
    w.coef = w.Pm + (size_t)n * n;                 // [beta; alpha] / B (2B), then [beta; alpha] (2B)
    w.Gam1 = w.coef + (size_t)n * n;
    w.Gp = ctx->gram_slabs;
    w.stamps = (ctx->tune_cov_dbg & 128)
                   ? reinterpret_cast<unsigned long long*>(ctx->gram_slabs + (size_t)GSMVI_MAX_KC * ctx->rmax * ctx->rmax)
                   : nullptr;
    return w;
}

// W = G Fm^T (split-K slabs), finished inside the elementwise stage: Rt1 = [Z; V], top half of Tm1 = X - mu
static int factor_w_prep(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* Z, int ldz, const double* X, int ldx,
                         const double* G, int ldg, const double* mu0, const double* F0, int ldf0) {
    const factor_ws w = factor_carve(ctx, D, 2 * B);
    int kc = 1;
    int rc = gsmvi_panel_t_product(ctx, st, D, B, G, ldg, F0, ldf0, D, ctx->pp, &kc);
    if (rc) return rc;
    hipLaunchKernelGGL(k_gsmf_prep, dim3((D + 255) / 256, B), dim3(256), 0, st, D, B, kc, Z, ldz, X, ldx, mu0, ctx->pp, w.Rt,
                       w.Tm);
    return chk("k_gsmf_prep");
}

// Back half: from Rt1 = [Z; V], Tm1 = [X - mu; V Fm] (n = 2B rows) to (mu, F).  The Gram matrix comes as kcg slabs at Gp
// (kcg = 1: finished); V Fm either finished in Tm1 (vf_slabs == nullptr) or as kcv split-K slabs summed by their consumer.
static int factor_back(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* mu0, const double* F0, int ldf0,
                       double* mu, double* F, int ldf, int* info_dev, int* n_reverts_dev, const double* Gp, int kcg,
                       const double* vf_slabs, int kcv, int jmode = 0, int chain_done = 0, int fork_vf = 0);

// the Gram product Gamma1 = [Z; V][Z; V]^T as split-K slabs (transposed panel product with A = M = Rt1)
static int factor_gram(gsmvi_ctx* ctx, hipStream_t st, int D, int n, const factor_ws& w, int* kcg, int mt_cap = 4) {
    return gsmvi_panel_t_product_mt(ctx, st, D, n, w.Rt, D, w.Rt, D, n, w.Gp, kcg, mt_cap);
}

// V Fm (rows B .. 2B-1 of Rt times Fm) as split-K slabs in ctx->pp, on the context's second stream: forked from `st` here,
// joined by the consumer's hipStreamWaitEvent(st, ctx->ev_join)
static int factor_fork_point(gsmvi_ctx* ctx, hipStream_t st) {
    hipError_t e = hipEventRecord(ctx->ev_fork, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0);
    if (e != hipSuccess) { gsmvi_set_error("%s: %s", "gsmvi_factor_impl", "fork failed"); return GSMVI_ERR_HIP; }
    return GSMVI_OK;
}
static int factor_fork_vf(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* Rt, const double* F0, int ldf0, int* kcv,
                          bool fork_here = true) {
    int rc = fork_here ? factor_fork_point(ctx, st) : GSMVI_OK;
    if (rc) return rc;
    rc = gsmvi_panel_product_nc(ctx, ctx->side, nullptr, D, D, B, Rt + (size_t)B * D, D, nullptr, 1.0, F0, ldf0, ctx->pp, kcv);
    if (rc) return rc;
    if (!ctx->px_used) {
        gsmvi_set_error("%s: %s", "gsmvi_factor_impl", "fast panel kernel expected (internal error)");
        return GSMVI_ERR_UNSUPPORTED;
    }
    if (hipEventRecord(ctx->ev_join, ctx->side) != hipSuccess) {
        gsmvi_set_error("%s: %s", "gsmvi_factor_impl", "join failed");
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}

int gsmvi_factor_impl(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* Z, int ldz, const double* X, int ldx,
                      const double* G, int ldg, const double* mu0, const double* F0, int ldf0, double* mu, double* F,
                      int ldf, int* info_dev, int* n_reverts_dev) {
    const int n = 2 * B;
    const factor_ws w = factor_carve(ctx, D, n);
    int rc = factor_w_prep(ctx, st, D, B, Z, ldz, X, ldx, G, ldg, mu0, F0, ldf0);
    if (rc) return rc;
    int kcg = 1, kcv = 1;
    // Launch diet (round 3).  Where the consumer of V Fm can sum split-K slabs while it loads them (n = 32, 64: the update
    // kernel with the skinny product folded in; n = 128: the fast panel kernel of Fs = K'' Tm1), the V Fm product keeps its
    // slabs and carries the finish of the Gram slabs as a side job of its workgroups (the one-workgroup chain kernel would
    // pull them through a single CU: measured +5.6 us).  Otherwise: product + finish launches, as before.
    // (any batch size since round 5: n <= 64 always; 64 < n <= 128 when B % 16 == 0 -- the K'' Tm product takes its V Fm rows
    // from split-K slabs per wave, and a wave's 16 rows must lie on one side of row B)
    const bool lean = !ctx->tune_no_fast && ctx->tune_direct_out && (n <= 64 || (n <= 128 && B % 16 == 0)) && D % 2 == 0 &&
                      ldf0 % 2 == 0 && ldf % 2 == 0 && (reinterpret_cast<uintptr_t>(F0) & 15u) == 0;
    // Large D, n = 128: the V Fm product (MFMA-bound, 65 us at D = 4096) does not depend on the Gram product and the eight small
    // launches of the 2B x 2B chain (~100 us on a few CUs), so it runs on the context's second stream beside them; the two
    // event edges cost ~10 us, which is why D = 1024 does not fork (measured in round 2: slower there).
    if (lean && n > 64 && ctx->side && ctx->tune_fork_min_D > 0 && D >= ctx->tune_fork_min_D) {
        // Where to fork.  Launched eagerly, the best place is behind the last multi-workgroup kernel in front of the one-workgroup
        // factorisations (inside factor_back, fork_vf = 1): V Fm then overlaps those, and the memory-bound Gram / Gamma kernels run
        // without its competition (313 -> 299 us at D = 4096).  Replayed from a hipGraph that placement is SLOWER (335 us: the
        // graph executor orders the branch in front of the factorisation it should run beside), so under stream capture the
        // fork stays in front of the Gram product (fork_vf = 2: 314 us).
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        const int fork_vf = (cap == hipStreamCaptureStatusActive) ? 2 : 1;
        if (fork_vf == 2 && (rc = factor_fork_point(ctx, st))) return rc;
        if ((rc = factor_gram(ctx, st, D, n, w, &kcg))) return rc;
        // (captured AFTER the Gram node: the graph executor issues nodes in creation order, and a 512-workgroup product issued
        // first delays the Gram product it does not depend on)
        if (fork_vf == 2 && (rc = factor_fork_vf(ctx, st, D, B, w.Rt, F0, ldf0, &kcv, false))) return rc;
        return factor_back(ctx, st, D, B, mu0, F0, ldf0, mu, F, ldf, info_dev, n_reverts_dev, w.Gp, kcg,
                           fork_vf == 2 ? ctx->pp : nullptr, kcv, 0, 0, fork_vf);
    }
    if ((rc = factor_gram(ctx, st, D, n, w, &kcg, (ctx->tune_rider && n <= 64) ? ctx->tune_gram_mt : 4))) return rc;
    if (lean) {
        const bool rider = ctx->tune_rider && n <= 64;
        if (rider) {                               // the 2B x 2B chain rides in the V Fm launch and sums the Gram slabs itself
            gsmvi_panel_extras& px = ctx->px;
            px.rd_on = 1;
            px.rd_n = n;
            px.rd_B = B;
            px.rd_Gp = w.Gp;
            px.rd_kcg = kcg;
            px.rd_Kmat = w.Rg;
            px.rd_coef = w.coef;
            px.rd_bad = info_dev;
            px.rd_stamps = w.stamps;
            px.rd_jmode = 0;
            px.rd_prior = nullptr;
        } else if (kcg > 1) {
            ctx->px.sj_src = w.Gp;
            ctx->px.sj_kc = kcg;
            ctx->px.sj_stride = (size_t)n * n;
            ctx->px.sj_len = n * n;
            ctx->px.sj_dst = w.Gam1;
        }
        // The chain rides (~35 us on one CU): the product beside it is off the critical path, so it may as well run UNSPLIT
        // and write the finished V Fm rows itself (64 workgroups instead of 256, ~3x longer, still hidden) -- the update kernel
        // then loads 16 KB of finished rows per tile instead of four 16 KB slabs.  Round 5, 1024 <= D <= rider_direct_max_D:
        // c3 iteration 77.3 -> 75.4 us, D = 2048: 129.4 -> 127.6; below D = 1024 the product is one chunk anyway and the eager
        // iteration got SLOWER (D = 256: 50 -> 63 us); at D = 4096 the unsplit product would outlast the chain.
        if (rider && D >= 1024 && D <= ctx->tune_rider_direct_max_D) {
            const int kc_user = ctx->tune_panel_kc;
            ctx->tune_panel_kc = 1;
            ctx->px_used = 0;
            rc = gsmvi_panel_product_out(ctx, st, D, D, B, w.Rt + (size_t)B * D, D, nullptr, 1.0, F0, ldf0, nullptr,
                                         w.Tm + (size_t)B * D, D);
            ctx->tune_panel_kc = kc_user;
            if (rc) return rc;
            if (!ctx->px_used) {
                gsmvi_set_error("%s: %s", "gsmvi_factor_impl", "fast panel kernel expected (internal error)");
                return GSMVI_ERR_UNSUPPORTED;
            }
            return factor_back(ctx, st, D, B, mu0, F0, ldf0, mu, F, ldf, info_dev, n_reverts_dev, w.Gp, kcg, nullptr, 0, 0, 1);
        }
        if ((rc = gsmvi_panel_product_nc(ctx, st, nullptr, D, D, B, w.Rt + (size_t)B * D, D, nullptr, 1.0, F0, ldf0, ctx->pp,
                                         &kcv)))
            return rc;
        if (!ctx->px_used) {                       // (cannot happen under `lean`; kept as a guard)
            gsmvi_set_error("%s: %s", "gsmvi_factor_impl", "fast panel kernel expected (internal error)");
            return GSMVI_ERR_UNSUPPORTED;
        }
        if (rider)
            return factor_back(ctx, st, D, B, mu0, F0, ldf0, mu, F, ldf, info_dev, n_reverts_dev, w.Gp, kcg, ctx->pp, kcv, 0, 1);
        return factor_back(ctx, st, D, B, mu0, F0, ldf0, mu, F, ldf, info_dev, n_reverts_dev, kcg > 1 ? w.Gam1 : w.Gp, 1,
                           ctx->pp, kcv);
    }
    if ((rc = gsmvi_panel_product_out(ctx, st, D, D, B, w.Rt + (size_t)B * D, D, nullptr, 1.0, F0, ldf0, nullptr,
                                      w.Tm + (size_t)B * D, D)))
        return rc;
    return factor_back(ctx, st, D, B, mu0, F0, ldf0, mu, F, ldf, info_dev, n_reverts_dev, w.Gp, kcg, nullptr, 0);
}

// Factor-form BaM (gsmvi_bam.hip): Rt = [Vw; Zw] and Tm = Rt Fm (2 Bh rows each) are in the panels named by ctx->fo_Rt / fo_Tm;
// F = F0 + Rt^T K Tm with M = I + Vw^T Vw - Zw^T Zw = C^T C (J = diag(I, -I)).  mu receives mu0: the caller owns BaM's mean.
// Two calls with the caller's Rt Fm product between them: _gram launches the Gram product and arms the NEXT fast panel launch
// with the 2B x 2B chain as its rider workgroup (2B <= 64) or with the finish of the Gram slabs as a side job;
// _back runs whatever that launch did not take.
int gsmvi_factor_signed_gram(gsmvi_ctx* ctx, hipStream_t st, int D, int Bh, int* kcg, int* info_dev, int* rides) {
    const int n = 2 * Bh;
    const factor_ws w = factor_carve(ctx, D, n);
    *rides = (ctx->tune_rider && n <= 64 && !ctx->tune_no_fast) ? 1 : 0;
    int rc = factor_gram(ctx, st, D, n, w, kcg, *rides ? ctx->tune_gram_mt : 4);
    if (rc) return rc;
    if (*rides) {                                  // the chain rides in the caller's next fast panel launch (k_panel_fast<.., RIDER>)
        gsmvi_panel_extras& px = ctx->px;
        px.rd_on = 1;
        px.rd_n = n;
        px.rd_B = Bh;
        px.rd_Gp = w.Gp;
        px.rd_kcg = *kcg;
        px.rd_Kmat = w.Rg;
        px.rd_coef = w.coef;
        px.rd_bad = info_dev;
        px.rd_stamps = w.stamps;
        px.rd_jmode = ctx->chain_pi ? 2 : 1;       // (2: the orthogonal-basis form, dense J' = S'^T diag(I, -I) S' given by Pi)
        px.rd_Pi = ctx->chain_pi;
        px.rd_R11 = ctx->chain_pi ? ctx->chain_r11 : nullptr;
        px.rd_W11 = ctx->chain_pi ? ctx->chain_w11 : nullptr;
        px.rd_prior = ctx->ints + 8;               // the flag of BaM's (B x B) chain
    } else if (*kcg > 1 && n <= 128) {             // (n > 128: no side job -- it would keep the caller's 2B + 1-row product off the
                                                   // 64 x 64-tile kernel, 36 us instead of ~15; k_gsmf_gamma_big sums the slabs itself)
        ctx->px.sj_src = w.Gp;
        ctx->px.sj_kc = *kcg;
        ctx->px.sj_stride = (size_t)n * n;
        ctx->px.sj_len = n * n;
        ctx->px.sj_dst = w.Gam1;
    }
    return GSMVI_OK;
}
// taken = ctx->px_used of the panel launch behind _gram: the chain has run (rides) / the Gram matrix is finished (side job)
// join: the caller's Rt Fm product ran on the context's second stream (ctx->ev_join recorded there): waited for in front of K'' Tm
int gsmvi_factor_signed_back(gsmvi_ctx* ctx, hipStream_t st, int D, int Bh, const double* mu0, const double* F0, int ldf0,
                             double* mu, double* F, int ldf, int* info_dev, int* n_reverts_dev, int kcg, int rides, int taken,
                             int join) {
    const factor_ws w = factor_carve(ctx, D, 2 * Bh);
    const int finished = !rides && taken && 2 * Bh <= 128;
    return factor_back(ctx, st, D, Bh, mu0, F0, ldf0, mu, F, ldf, info_dev, n_reverts_dev, (kcg > 1 && finished) ? w.Gam1 : w.Gp,
                       finished ? 1 : kcg, nullptr, 0, ctx->chain_pi ? 2 : 1, (rides && taken) ? 1 : 0, join ? 2 : 0);
}

// [A | I] -> [R | W] of one n x n matrix, n <= 64, plain positive-definite rule, compact leading dimension n (gsmvi_bam.hip)
// info_off > 0: a failure is ADDED to *info (recorded only when nothing failed before; an earlier flag stays)
int gsmvi_cholw_small(hipStream_t st, int n, const double* A, double* R, double* W, int* info, int info_off) {
    if (n > 64)
        hipLaunchKernelGGL((k_cholw_ld<false, true>), dim3(1), dim3(512), 0, st, n, A, n, R, n, W, n, info, info_off, 0,
                           (const double*)nullptr, 0, 0);
    else
        hipLaunchKernelGGL((k_cholw_ld<false, false>), dim3(1), dim3(512), 0, st, n, A, n, R, n, W, n, info, info_off, 0,
                           (const double*)nullptr, 0, 0);
    return chk("k_cholw_ld");
}

// Batch-sharded form, stage 1: this rank's B_local samples -> records.
int gsmvi_factor_local_impl(gsmvi_ctx* ctx, hipStream_t st, int D, int Bl, const double* Z, int ldz, const double* X,
                            int ldx, const double* G, int ldg, const double* mu0, const double* F0, int ldf0,
                            double* rec, int ldrec) {
    const factor_ws w = factor_carve(ctx, D, 2 * Bl);
    int rc = factor_w_prep(ctx, st, D, Bl, Z, ldz, X, ldx, G, ldg, mu0, F0, ldf0);
    if (rc) return rc;
    if ((rc = gsmvi_panel_product_out(ctx, st, D, D, Bl, w.Rt + (size_t)Bl * D, D, nullptr, 1.0, F0, ldf0, nullptr,
                                      w.Tm + (size_t)Bl * D, D)))
        return rc;
    hipLaunchKernelGGL(k_gsmf_pack, dim3((D + 255) / 256, Bl), dim3(256), 0, st, D, Bl, w.Rt, w.Tm, rec, ldrec);
    return chk("k_gsmf_pack");
}

// Batch-sharded form, stage 2: every replica applies the combined update from ALL B records and the replicated Z.
int gsmvi_factor_apply_impl(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* Z, int ldz, const double* rec,
                            int ldrec, const double* mu0, const double* F0, int ldf0, double* mu, double* F, int ldf,
                            int* info_dev, int* n_reverts_dev) {
    const int n = 2 * B;
    const factor_ws w = factor_carve(ctx, D, n);
    hipLaunchKernelGGL(k_gsmf_unpack, dim3((D + 255) / 256, B), dim3(256), 0, st, D, B, Z, ldz, rec, ldrec, w.Rt, w.Tm);
    int rc = chk("k_gsmf_unpack");
    if (rc) return rc;
    int kcg = 1;
    if ((rc = factor_gram(ctx, st, D, n, w, &kcg))) return rc;
    return factor_back(ctx, st, D, B, mu0, F0, ldf0, mu, F, ldf, info_dev, n_reverts_dev, w.Gp, kcg, nullptr, 0);
}

// Column-sharded form (round 6; SURVEY 8(e) row 3 / (f) 3: the decomposition that divides the D^2 traffic and the D^2 memory
// of a factor-form fit): the rank owns the columns [col0, col0 + ncols) of Fm -- D x ncols, its own leading dimension.
//   sampler   x[:, C] = mu[C] + z Fm[:, C]                      owned slice (gsmvi_sample_cols_f64), all-gathered by the caller
//   W = G Fm^T = sum over ranks of G[:, C] Fm[:, C]^T            partial sums (gsmvi_gsm_rows_stage_f64 on the block), all-reduced
//   this call  Rt = [Z; W + Z], Tm[:, C] = [X - mu; V Fm[:, C]], the Gram matrix and the 2B x 2B chain (replicated: identical
//              inputs, identical arithmetic on every rank), then Fm'[:, C] = Fm[:, C] + Rt^T (K'' Tm[:, C]) and mu'[C]:
//              the update touches the owned block only -- no D x D matrix is ever sent, read or written whole.
// W: the all-reduced B x D product; X: the gathered samples (only its columns C are used); mu0 / mu: full-length vectors,
// entries C read / written.  col0 % 64 == 0 (tile aligned; the last block may be ragged).
int gsmvi_factor_apply_cols_impl(gsmvi_ctx* ctx, hipStream_t st, int D, int B, int col0, int ncols, const double* Z, int ldz,
                                 const double* W, int ldw, const double* X, int ldx, const double* mu0, const double* F0c,
                                 int ldf0, double* mu, double* Fc, int ldf, int* info_dev, int* n_reverts_dev) {
    const int n = 2 * B;
    const factor_ws w = factor_carve(ctx, D, n);
    // Rt = [Z; V], top half of Tm = X - mu0: k_gsmf_prep with the finished W as its one "slab" (needs ldw == D)
    hipLaunchKernelGGL(k_gsmf_prep, dim3((D + 255) / 256, B), dim3(256), 0, st, D, B, 1, Z, ldz, X, ldx, mu0, W, w.Rt, w.Tm);
    int rc = chk("k_gsmf_prep");
    if (rc) return rc;
    // V Fm on the owned columns, finished rows straight into Tm's lower half
    if ((rc = gsmvi_panel_product_out(ctx, st, D, ncols, B, w.Rt + (size_t)B * D, D, nullptr, 1.0, F0c, ldf0, nullptr,
                                      w.Tm + (size_t)B * D + col0, D)))
        return rc;
    int kcg = 1;
    if ((rc = factor_gram(ctx, st, D, n, w, &kcg))) return rc;
    ctx->colwin_0 = col0 / 64;
    ctx->colwin_n = (ncols + 63) / 64;
    rc = factor_back(ctx, st, D, B, mu0, F0c - col0, ldf0, mu, Fc - col0, ldf, info_dev, n_reverts_dev, w.Gp, kcg, nullptr, 0);
    ctx->colwin_0 = ctx->colwin_n = 0;
    return rc;
}

// ---- the 2B x 2B chain for 128 < n <= 256 (round 4; BASELINE config 4 in factor form has 2B = 256) ---------------------------
// Same algebra as the n <= 128 chain (Gamma = Rg^T Rg with W = Rg^-T, A' = I + Rg J Rg^T = T^T T, P = (T - I)(W S),
// K'' = (W S)^T P), built from the pieces that exist: the one-workgroup 128-row factorisation with the inverse factor
// (chol128w_body / chol64_blk, through k_cholw_ld with leading dimensions) for the two diagonal blocks of each matrix and
// the generic small-matrix product (gsmvi_smallgemm.h) for everything between them -- 14 launches, launch- and pivot-chain
// bound.  Slots: S22 / T1 of Gamma's factorisation live in Ap / Tt (A' does not exist yet); T's inverse factor (only its
// first block is used, for the block-row solve) in Gam1 and its S22 in Rg (both dead by then).  Returns K'' in w.Gam.
static int factor_chain_big(gsmvi_ctx* ctx, hipStream_t st, int n, int B, const factor_ws& w, const double* Gp, int kcg,
                            int jmode, const int* prior, int* info_dev) {
    // block split: B + B (n = 2B, 64 < B <= 128: both diagonal blocks on the 128-row one-workgroup kernel, which is what lets
    // independent ones share a launch).  In the factor-form BaM chain (jmode) the first diagonal block Gamma11 = Vw Vw^T is known
    // before the B x B chain and may have been factored beside it (ctx->early_ready).
    // (The same scheme on 64-row blocks for 64 < n <= 128 was measured and dropped: 14 launches instead of 6 cost more than
    // the shorter pivot chains save -- c5 317 us against 288, profiles/r04/c4_pair_ab.txt.)
    const int n1 = B, n2 = n - n1;
    int* info_g = ctx->ints;
    int* info_t = ctx->ints + 1;
    double* coef = w.coef;
    const size_t off = (size_t)n1 * n + n1;        // the (1, 1) block inside an n x n matrix
    const bool early = jmode && ctx->early_ready != 0;
    ctx->early_ready = 0;
    // paired launches (round 4): both diagonal blocks on the 128-row kernel, and the knob on
    const bool pair = ctx->tune_chain_pair && n1 > 64 && n2 > 64;
    // one diagonal block [A | I] -> [R | W] in its own launch
    auto cholw = [&](bool semidef, int nb, const double* A, int lda, double* R, int ldr, double* Wo, int ldw, int* info, int ioff,
                     int tol, const double* dg) {
        const int dgn = dg ? n : 0, dgs = dg ? n + 1 : 0;
#define CW(SD, BG) hipLaunchKernelGGL((k_cholw_ld<SD, BG>), dim3(1), dim3(512), 0, st, nb, A, lda, R, ldr, Wo, ldw, info, ioff, tol, dg, dgn, dgs)
        if (semidef) { if (nb > 64) CW(true, true); else CW(true, false); }
        else { if (nb > 64) CW(false, true); else CW(false, false); }
#undef CW
    };
    if (kcg <= 4) hipLaunchKernelGGL(k_gsmf_gamma_big<4>, dim3((n * n + 255) / 256), dim3(256), 0, st, n, B, Gp, kcg, w.Gam, coef, coef + n, jmode);
        else hipLaunchKernelGGL(k_gsmf_gamma_big<GSMVI_MAX_KC>, dim3((n * n + 255) / 256), dim3(256), 0, st, n, B, Gp, kcg, w.Gam, coef, coef + n, jmode);
    // Gamma = Rg^T Rg (rank-revealing rule), W = Rg^-T -> w.Pm
    if (early) {
        // [Gamma11 | I] -> [R11 | W11] ran as the second workgroup of k_bam_cholw's launch (compact, ld n1); the R12 product
        // reads W11 / R11 there and copies them into the n x n matrices
        const double* R11 = ctx->early + 128 * 128;
        const double* W11 = ctx->early + 2 * 128 * 128;
        const OpBlkR12 op_r12{n1, n2, n1, W11, w.Gam, w.Rg, n1, n, n, n1, R11, n1, w.Pm, n};
        if (jmode == 2) small_gemm_launch2(st, op_r12, OpChainX{B, B, B, R11, ctx->chain_pi, ctx->chain_x, n1});   // X = R11 Pi^T rides along
        else small_gemm_launch(st, op_r12);
    } else {
        // (advisor, round 4) In the factor-form BaM chain (jmode) the early job above guards this block with ITS OWN diagonal
        // (the second block's is not known yet when it runs); the in-chain factorisation does the same, so the accept / revert
        // decision cannot depend on the "chain_pair" knob.  The second block is guarded with the whole diagonal in both modes:
        // an absurd entry anywhere still switches the rank-revealing rule off where it decides (G4, tests/test_gpu_factor.py).
        cholw(true, n1, w.Gam, n, w.Rg, n, w.Pm, n, info_g, 0, 0, jmode ? nullptr : w.Gam);
        const OpBlkR12 op_r12{n1, n2, n1, w.Pm, w.Gam, w.Rg, n, n, n, n1};
        if (jmode == 2) small_gemm_launch2(st, op_r12, OpChainX{B, B, B, w.Rg, ctx->chain_pi, ctx->chain_x, n});
        else small_gemm_launch(st, op_r12);
    }
    // slots while Gamma is factored: S22 and T1 in Gam1 (free until the end of the chain); A'11 -> Ap, T11 -> Tt, T's inverse
    // factor (only its first block exists and is used: the block-row solve) compact in the upper half of the coefficient slot
    double* S22g = w.Gam1;
    double* T1g = w.Gam1 + (size_t)n2 * n2;
    double* Wt = pair ? w.coef + (size_t)n * n / 2 : w.Gam1;   // n1 x n1, ld n1 (unpaired: Gam1 is free again by then)
    const OpBlkS22 op_s22g{n2, n2, n1, w.Rg, w.Gam, S22g, n, n, n1, 1};
    const OpBlkT1 op_t1{n2, n1, n2, w.Pm, w.Rg, T1g, n, n, n1};
    const OpBlkW21 op_w21{n2, n1, n1, T1g, w.Pm, w.Rg, n, n, n1};
    const OpSmallA op_a{n, n, n, w.Rg, info_g, w.Ap, B, jmode, n, ctx->chain_x};   // A' = I + Rg J Rg^T = T^T T (plain rule: this IS the accept test)
    const OpBlkR12 op_r12t{n1, n2, n1, Wt, w.Ap, w.Tt, n1, n, n, n1};
    if (pair) {
        // A'11 = I + (Rg J Rg^T)_11 needs only [R11 R12]: its factorisation runs beside Gamma's second block, one launch.
        // Independent PRODUCTS share launches too (k_small_gemm2): S22 with A'11, T1 with A', W21 with T's R12.
        small_gemm_launch2(st, op_s22g, OpSmallA{n1, n1, n, w.Rg, info_g, w.Ap, B, jmode, n, ctx->chain_x});
        const cholw_job ja{n2, S22g, n2, w.Rg + off, n, w.Pm + off, n, info_g, n1, 1, w.Gam, n, n + 1};
        const cholw_job jb{n1, w.Ap, n, w.Tt, n, Wt, n1, info_t, 0, 0, nullptr, 0, 0};
        hipLaunchKernelGGL(k_cholw_pair, dim3(2), dim3(512), 0, st, ja, jb);
        small_gemm_launch2(st, op_t1, op_a);
        small_gemm_launch2(st, op_w21, op_r12t);
    } else {
        small_gemm_launch(st, op_s22g);
        cholw(true, n2, S22g, n2, w.Rg + off, n, w.Pm + off, n, info_g, n1, 1, w.Gam);
        small_gemm_launch(st, op_t1);
        small_gemm_launch(st, op_w21);
        small_gemm_launch(st, op_a);
        cholw(false, n1, w.Ap, n, w.Tt, n, Wt, n1, info_t, 0, 0, nullptr);
        small_gemm_launch(st, op_r12t);
    }
    small_gemm_launch(st, OpBlkS22{n2, n2, n1, w.Tt, w.Ap, w.Rg, n, n, n1, 0});
    if (n2 > 64)                                   // (the last block needs no inverse factor: the plain 128-row kernel, 32 us against 39)
        hipLaunchKernelGGL(k_chol128<false>, dim3(1), dim3(512), 0, st, n2, w.Rg, w.Tt + off, n, info_t, n1);
    else
        cholw(false, n2, w.Rg, n2, w.Tt + off, n, w.Gam1, n2, info_t, n1, 0, nullptr);
    int rc = chk("k_cholw_ld");
    if (rc) return rc;
    // P = (T - I)(W S) with the accept / revert decision, K'' = (W S)^T P
    small_gemm_launch(st, OpChainP{n, n, n, w.Tt, w.Pm, coef + n, w.Ap, B, info_dev, info_g, info_t, prior});
    small_gemm_launch(st, OpChainK{n, n, n, w.Pm, w.Ap, coef + n, w.Gam, B, info_dev});
    return chk("k_small_gemm");
}

static int factor_back(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* mu0, const double* F0, int ldf0,
                       double* mu, double* F, int ldf, int* info_dev, int* n_reverts_dev, const double* Gp, int kcg,
                       const double* vf_slabs, int kcv, int jmode, int chain_done, int fork_vf) {
    const int n = 2 * B;                           // n is even
    const factor_ws w = factor_carve(ctx, D, n);
    double *Rt = w.Rt, *Tm = w.Tm, *Fs = w.Fs, *coef = w.coef;
    // the window of 64-column tiles the update launches write: all of F, or the owned column block of a column-sharded fit
    // (gsmvi_gsm_factor_apply_cols_f64: F0 / F are then base pointers shifted so that GLOBAL column indices address the block)
    const int cw_n = ctx->colwin_n > 0 ? ctx->colwin_n : (D + 63) / 64, cw_0 = ctx->colwin_n > 0 ? ctx->colwin_0 : 0;
    int* info_g = ctx->ints;
    int* info_t = ctx->ints + 1;
    int rc, kc2 = 1;
    double* Kmat;
    const int* prior = jmode ? ctx->ints + 8 : nullptr;      // the flag of BaM's (B x B) chain

    if (n <= 64) {
        // everything small in one workgroup
        Kmat = w.Rg;
        if (!chain_done) {                         // (chain_done: it ran as the rider workgroup of the caller's panel product)
            hipLaunchKernelGGL(k_gsmf_small16, dim3(1), dim3(512), 0, st, n, B, Gp, kcg, Kmat, coef, info_dev, w.stamps, jmode, prior,
                               jmode == 2 ? ctx->chain_pi : (const double*)nullptr, jmode == 2 ? ctx->chain_r11 : (const double*)nullptr,
                               jmode == 2 ? ctx->chain_w11 : (const double*)nullptr);
            if ((rc = chk("k_gsmf_small16"))) return rc;
        }
    } else if (n > 128) {
        Kmat = w.Gam;
        if ((rc = factor_chain_big(ctx, st, n, B, w, Gp, kcg, jmode, prior, info_dev))) return rc;
    } else {
        // 64 < n <= 128: Gamma, the two n x n Choleskys one workgroup each, W and K in their own kernels
        Kmat = w.Gam;                              // Gamma is dead once Rg exists
        double* Wm = w.Pm;
        // (jmode 2 with the first diagonal block given: see k_gsmf_gamma_big)
        const double* R11g = jmode == 2 ? ctx->chain_r11 : nullptr;
        const double* W11g = jmode == 2 ? ctx->chain_w11 : nullptr;
        if (kcg <= 4) hipLaunchKernelGGL(k_gsmf_gamma_big<4>, dim3((n * n + 255) / 256), dim3(256), 0, st, n, B, Gp, kcg, w.Gam, coef, coef + n, jmode, R11g, W11g, w.Rg, w.Pm);
        else hipLaunchKernelGGL(k_gsmf_gamma_big<GSMVI_MAX_KC>, dim3((n * n + 255) / 256), dim3(256), 0, st, n, B, Gp, kcg, w.Gam, coef, coef + n, jmode, R11g, W11g, w.Rg, w.Pm);
        if (fork_vf == 1) {                        // V Fm on the second stream from here on: beside the one-workgroup kernels below
            if ((rc = factor_fork_vf(ctx, st, D, B, Rt, F0, ldf0, &kcv))) return rc;
            vf_slabs = ctx->pp;
        }
        // Gram matrix: semi-definite rule; W = Rg^-T comes out of the same factorisation (no substitution launch)
        if (R11g) {                                // the second diagonal block alone (B <= 64 rows: one 64-pivot job instead of 128 pivots;
            const size_t o22 = (size_t)B * n + B;  // the magnitude guard still looks at the whole diagonal)
            hipLaunchKernelGGL((k_cholw_ld<true, false>), dim3(1), dim3(512), 0, st, B, w.Gam + o22, n, w.Rg + o22, n, w.Pm + o22, n,
                               info_g, 0, 0, w.Gam, n, n + 1);
        } else if (n > 64) hipLaunchKernelGGL(k_chol128w<true>, dim3(1), dim3(512), 0, st, n, w.Gam, w.Rg, w.Pm, info_g);
        else hipLaunchKernelGGL((k_cholw_ld<true, false>), dim3(1), dim3(512), 0, st, n, w.Gam, n, w.Rg, n, w.Pm, n, info_g, 0, 0,
                                (const double*)nullptr, 0, 0);           // (n <= 64 comes here only with a dense J': jmode 2)
        // A' = I + Rg J Rg^T, then its plain factorisation T -- the accept / revert test.  (Round 4: the three n x n products of
        // this chain run on the generic MFMA block kernel of gsmvi_smallgemm.h, 64 workgroups each; the VALU dot-product
        // kernel k_gsmf_small_a (14.4 us) and the 16-workgroup k_gsmf_gemm128 (16.5 + 13 us) they replace were deleted.)
        if (jmode == 2)                            // dense J' (gsmvi_bam.hip, orthogonal-basis form): X = R11 Pi^T joins Rg's (1, 2) block
            small_gemm_launch(st, OpChainX{B, B, B, w.Rg, ctx->chain_pi, ctx->chain_x, n});
        small_gemm_launch(st, OpSmallA{n, n, n, w.Rg, info_g, w.Ap, B, jmode, n, ctx->chain_x});
        if (n > 64) hipLaunchKernelGGL(k_chol128<false>, dim3(1), dim3(512), 0, st, n, w.Ap, w.Tt, n, info_t, 0);
        else hipLaunchKernelGGL((k_cholw_ld<false, false>), dim3(1), dim3(512), 0, st, n, w.Ap, n, w.Tt, n, w.Gam1, n, info_t, 0, 0,
                                (const double*)nullptr, 0, 0);           // (its inverse factor is not used: Gam1 is free here)
        if ((rc = chk("k_chol128"))) return rc;
        double* Pmat = w.Ap;                       // A' is dead once T exists
        // P = (T - I) (W S), with the chain's accept / revert decision; then K'' = (W S)^T P
        small_gemm_launch(st, OpChainP{n, n, n, w.Tt, Wm, coef + n, Pmat, B, info_dev, info_g, info_t, prior});
        small_gemm_launch(st, OpChainK{n, n, n, Wm, Pmat, coef + n, Kmat, B, info_dev});
        if ((rc = chk("k_small_gemm"))) return rc;
    }
    if (!ctx->tune_no_fast && ctx->tune_direct_out && D % 2 == 0 && n <= 64 && ldf0 % 2 == 0 && ldf % 2 == 0) {
        ctx->path |= GSMVI_PATH_FUPD_FAST;
        // n <= 64: the skinny product Fs = K'' Tm1 is folded into the update kernel (k_gsmf_update_fs): one launch less
        const int ntl = (D + 63) / 64;
#define UFS(NPV, KCBV, RG) hipLaunchKernelGGL((k_gsmf_update_fs<NPV, KCBV, RG>), dim3(ntl * cw_n), dim3(512), 0, st, D, B, Rt, Kmat, Tm, vf_slabs, kcv, F0, ldf0, F, ldf, coef, mu0, mu, info_dev, n_reverts_dev, ctx->bam_mean, cw_0, cw_n)
        ctx->bam_mean_done = ctx->bam_mean.xbar ? 1 : 0;
        if (D % 64 != 0) {
            if (kcv <= 4) { if (n <= 32) UFS(1, 4, true); else UFS(2, 4, true); }
            else { if (n <= 32) UFS(1, GSMVI_MAX_KC, true); else UFS(2, GSMVI_MAX_KC, true); }
        } else {
            if (kcv <= 4) { if (n <= 32) UFS(1, 4, false); else UFS(2, 4, false); }
            else { if (n <= 32) UFS(1, GSMVI_MAX_KC, false); else UFS(2, GSMVI_MAX_KC, false); }
        }
#undef UFS
        return chk("k_gsmf_update_fs");
    }
    // Fs = K'' Tm1 as one skinny GEMM: inner dimension n <= one chunk, so there is exactly one slab, written straight into Fs
    if (fork_vf && hipStreamWaitEvent(st, ctx->ev_join, 0) != hipSuccess) {       // the V Fm slabs come from the side stream (large D)
        gsmvi_set_error("%s: %s", "gsmvi_factor_impl", "hipStreamWaitEvent failed");
        return GSMVI_ERR_HIP;
    }
    if (vf_slabs) {                                // rows B .. 2B-1 of Tm1 = sum of the V Fm slabs; finished into Tm1 on the way
        ctx->px.msl = vf_slabs;
        ctx->px.kcm = kcv;
        ctx->px.mstride = (size_t)B * D;
        ctx->px.ldsl = D;
        ctx->px.msplit = B;
        ctx->px.mfin = Tm + (size_t)B * D;
        ctx->px.ldfin = D;
    }
    const int kc_user = ctx->tune_panel_kc;
    if (n > 128) ctx->tune_panel_kc = 1;           // inner dimension n = two 128-row chunks of one workgroup: ONE slab, written into Fs
    rc = gsmvi_panel_product_nc(ctx, st, nullptr, n, D, n, Kmat, n, nullptr, 1.0, Tm, D, Fs, &kc2);
    ctx->tune_panel_kc = kc_user;
    if (rc) return rc;
    if (kc2 != 1 || (vf_slabs && !ctx->px_used)) {
        gsmvi_set_error("%s: %s", "gsmvi_factor_impl", "K Tm product was split or left the fast kernel (internal error)");
        return GSMVI_ERR_UNSUPPORTED;
    }
    const int nt = (D + 63) / 64;
    if (!ctx->tune_no_fast && D % 2 == 0 && n <= 256) {
        // the fast kernel also writes the mean and counts the revert (any even n <= 256 since round 5)
#define UF(NPV, RG) hipLaunchKernelGGL((k_gsmf_update_fast<NPV, RG>), dim3(nt * cw_n), dim3(512), 0, st, D, B, Rt, Fs, F0, ldf0, F, ldf, Tm, coef, mu0, mu, info_dev, n_reverts_dev, cw_0, cw_n)
        const int npsel = n <= 32 ? 1 : (n <= 64 ? 2 : (n <= 128 ? 4 : 8));       // instantiated pass counts; RAG unless n fills them all
        if (D % 64 != 0 || n != 32 * npsel) { if (n <= 32) UF(1, true); else if (n <= 64) UF(2, true); else if (n <= 128) UF(4, true); else UF(8, true); }
        else { if (n <= 32) UF(1, false); else if (n <= 64) UF(2, false); else if (n <= 128) UF(4, false); else UF(8, false); }
#undef UF
        ctx->path |= GSMVI_PATH_FUPD_FAST;
        return chk("k_gsmf_update_fast");
    }
    ctx->path |= GSMVI_PATH_FUPD_GENERIC;
    {
        const int c0 = 64 * cw_0, c1 = (64 * (cw_0 + cw_n) < D) ? 64 * (cw_0 + cw_n) : D;
        hipLaunchKernelGGL(k_gsmf_mean, dim3((c1 - c0 + 255) / 256), dim3(256), 0, st, D, B, Tm, coef, mu0, mu, info_dev, n_reverts_dev, c0, c1);
    }
    if ((rc = chk("k_gsmf_mean"))) return rc;
    hipLaunchKernelGGL(k_gsmf_update, dim3(nt * cw_n), dim3(256), 0, st, D, n, Rt, Fs, F0, ldf0, F, ldf, info_dev, cw_0, cw_n);
    return chk("k_gsmf_update");
}
