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
//   Fm' = C Fm = Fm + Rt^T Fs,   Fs = Rg^-1 (T - I) Rg^-T Tm,   Tm = Rt Fm = [X - mu; U Fm].
// Per iteration Fm is read three times (W, U Fm, update) and written once; no O(D^3) work.
// Requires n = 2B <= D (Gamma must be nonsingular) and n <= 128.
#include "gsmvi_common.h"
#include "gsmvi_ctx.h"
#include "gsmvi_chol64.h"
#include "../../include/gsmvi_hip.h"

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
template <int MT, int CHW>
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
                src = M + (size_t)(j0 + row - NR) * ldm + colc;            // mrows % 16 == 0: all 16 rows exist
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
        if (row < nrows) {
            double s = 0.0;
#pragma unroll
            for (int ww = 0; ww < 8; ww += 2) s += red[(ww * NR + rr) * 17 + cc] + red[((ww + 1) * NR + rr) * 17 + cc];
            Pp[((size_t)blockIdx.y * nrows + row) * mrows + j0 + cc] = s;
        }
    }
}

// ---- whitened per-sample stage: one 1024-thread workgroup per sample (D <= 16384) -----------------
//   w_b = sum_kc Pp[kc][b];  scalars;  u_b;  writes Rt = [Z; U] (n x D), its transpose Rtt (D x nq) and the
//   top half of Tm = Rt Fm, i.e. X - mu.
template <int EPT>
__global__ __launch_bounds__(1024) void k_gsmf_scalars(int D, int B, int KC, const double* __restrict__ Z, int ldz,
                                                       const double* __restrict__ X, int ldx,
                                                       const double* __restrict__ mu0,
                                                       const double* __restrict__ Pp, double* __restrict__ Rt,
                                                       double* __restrict__ Rtt, int nq, double* __restrict__ Tm) {
    __shared__ double lds[34];
    const int b = blockIdx.x, tid = threadIdx.x;
    double wv[EPT], zv[EPT], xv[EPT], mv[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {                 // all loads of the row in one batch
        const int i = tid + 1024 * e;
        const int ic = i < D ? i : D - 1;
        double t = 0.0;
        for (int kc = 0; kc < KC; ++kc) t += Pp[((size_t)kc * B + b) * D + ic];
        wv[e] = t;
        zv[e] = Z[(size_t)b * ldz + ic];
        xv[e] = X[(size_t)b * ldx + ic];
        mv[e] = mu0[ic];
    }
    double p0 = 0.0, p1 = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        if (tid + 1024 * e < D) {
            p0 += wv[e] * wv[e];
            p1 += zv[e] * wv[e];
        }
    p0 = wave_sum(p0);
    p1 = wave_sum(p1);
    if ((tid & 63) == 0) {
        lds[2 * (tid >> 6)] = p0;
        lds[2 * (tid >> 6) + 1] = p1;
    }
    __syncthreads();
    if (tid == 0) {
        double ww = 0.0, zw = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            ww += lds[2 * k];
            zw += lds[2 * k + 1];
        }
        const double rho = 0.5 * sqrt(1.0 + 4.0 * (ww + zw * zw)) - 0.5;
        const double den = 1.0 + rho - zw;
        lds[32] = 1.0 / (1.0 + rho);
        lds[33] = (ww + zw) / den;
    }
    __syncthreads();
    const double beta = lds[32], cz = lds[33];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + 1024 * e;
        if (i < D) {
            const double z = zv[e];
            const double u = ((wv[e] + z) + z * cz) * beta;
            Rt[(size_t)b * D + i] = z;
            Rt[(size_t)(B + b) * D + i] = u;
            Rtt[(size_t)i * nq + b] = z;
            Rtt[(size_t)i * nq + B + b] = u;
            Tm[(size_t)b * D + i] = xv[e] - mv[e];
        }
    }
}

// LDS[128][130] <- upper triangle of the n x n matrix src (n <= 128), zero below, identity beyond n.
// Clamped unconditional loads, 16 in flight per thread: a guarded load per iteration serialises 64 L2 round trips.
__device__ __forceinline__ void load_upper128(double* Mt, const double* __restrict__ src, int n) {
    const int j = threadIdx.x & 127, jc = j < n ? j : n - 1, ih = threadIdx.x >> 7;
#pragma unroll
    for (int it0 = 0; it0 < 64; it0 += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = ih + 2 * (it0 + u), ic = i < n ? i : n - 1;
            v[u] = src[(size_t)ic * n + jc];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = ih + 2 * (it0 + u);
            Mt[i * 130 + j] = (i < n && j < n) ? (j >= i ? v[u] : 0.0) : (i == j ? 1.0 : 0.0);
        }
    }
}

// ---- A' = I + Rg J Rg^T  (n x n, 64 < n <= 128),  J = (1/B) [[0, I], [I, -I]],  Rg upper triangular ----
//   (Rg J)[i][k] = (1/B) * ( k <  B : Rg[i][B+k]
//                            k >= B : Rg[i][k-B] - Rg[i][k] )
// 16 x 16 outputs per workgroup; the 16 rows of (Rg J) and the 16 rows of Rg it needs are staged in LDS
// ([row][130]: reading one row per lane is conflict-free).
__global__ __launch_bounds__(256) void k_gsmf_small_a(int n, int B, const double* __restrict__ Rg,
                                                      const int* __restrict__ info_g, double* __restrict__ Ap) {
    __shared__ double RJ[16 * 130];
    __shared__ double RR[16 * 130];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    for (int e = tid; e < 16 * 128; e += 256) {
        const int r = e >> 7, k = e & 127;
        const int gi = i0 + r, gj = j0 + r;
        double vj = 0.0, vr = 0.0;
        if (k < n) {
            if (gi < n) vj = (k < B) ? Rg[(size_t)gi * n + B + k] : (Rg[(size_t)gi * n + k - B] - Rg[(size_t)gi * n + k]);
            if (gj < n) vr = Rg[(size_t)gj * n + k];
        }
        RJ[r * 130 + k] = vj;
        RR[r * 130 + k] = vr;
    }
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
    for (int k = 0; k < 128; k += 2) {
        s0 += RJ[ty * 130 + k] * RR[tx * 130 + k];
        s1 += RJ[ty * 130 + k + 1] * RR[tx * 130 + k + 1];
    }
    const int i = i0 + ty, j = j0 + tx;
    if (i < n && j < n) {
        double v = (i == j ? 1.0 : 0.0) + (s0 + s1) / (double)B;
        if (*info_g != 0) v = (i == j) ? -1.0 : 0.0;     // Gamma was singular: force the PD test to fail
        Ap[(size_t)i * n + j] = v;
    }
}

// ---- Cholesky A = R^T R of one n x n matrix, 64 < n <= 128, in ONE workgroup ----------------------------
// The matrix lives in LDS ([128][130], identity beyond n).  2 x 2 blocks of 64: chol64 of A11, the block row
// R12 = R11^-T A12 one column per quad of lanes (as k_potrf_panel), A22 -= R12^T R12 with a 4 x 4 register
// tile per thread, chol64 of A22.  *info = 1-based index of the first bad pivot (0 = ok); R gets the upper
// factor with a zero strictly-lower triangle.
template <bool SEMIDEF>
__global__ __launch_bounds__(512) void k_chol128(int n, const double* __restrict__ A, double* __restrict__ R,
                                                 int* __restrict__ info) {
    // Eight waves (round 2).  Waves 0-3 factor A11 (chol64_rows_s needs exactly 256 threads) while waves 4-7 solve the block
    // row R12 = R11^-T A12 ONE PIVOT BEHIND: substitution step p needs row p of the factor and its pivot only, both published
    // (unscaled, in LDS) by the barrier that opens pivot p; the helpers execute one barrier per step, as in k_potrf_step8.
    // A22 -= R12^T R12 then runs on the MFMA pipe with all eight waves (it was a VALU loop of 64 x 16 FMAs per thread), and
    // the second block is factored by waves 0-3 with the helpers matching its barriers.  55 -> ~37 us per call.
    constexpr int MS = 130;
    __shared__ __attribute__((aligned(16))) double M[128 * MS];
    __shared__ double rinv[128];
    __shared__ int sh_fail[2];
    const int tid = threadIdx.x;
    const bool team = tid < 256;
    if (team) load_upper128(M, A, n);
    if (tid < 128) rinv[tid] = 1.0;
    __syncthreads();
    bool moderate = true;
    if (SEMIDEF) {                                  // rounding-floor shift of the diagonal and the magnitude guard
        if (tid == 0) sh_fail[1] = 1;
        __syncthreads();
        if (tid < 128) {
            const double d = M[tid * MS + tid];
            M[tid * MS + tid] = d - GSMVI_DEP_TOL * d;           // rounding floor of the row, see chol64_rows_s
            if (!(d < 4294967296.0)) sh_fail[1] = 0;
        }
        __syncthreads();
        moderate = sh_fail[1] != 0;
        __syncthreads();
    }
    if (team) chol64_rows_s<MS, SEMIDEF>(M, rinv, 64, &sh_fail[0], moderate);
    else chol128_helper_rowsolve<MS>(M);               // R12 = R11^-T A12, one pivot behind (gsmvi_chol64.h)
    __syncthreads();
    chol128_rank64_update<MS>(M);                       // A22 -= R12^T R12 on the MFMA pipe
    __syncthreads();
    if (team) chol64_rows_s<MS, SEMIDEF>(M + 64 * MS + 64, rinv + 64, n - 64, &sh_fail[1], moderate);
    else chol64_helper_idle<MS>(n - 64);
    for (int e = tid; e < n * n; e += 512) {
        const int i = e / n, j = e % n;
        R[e] = (j >= i) ? M[i * MS + j] : 0.0;
    }
    if (tid == 0) *info = sh_fail[0] != 0 ? sh_fail[0] : (sh_fail[1] != 0 ? 64 + sh_fail[1] : 0);
}

// ---- F = F0 + Rt^T Fs  (full, non-symmetric rank-n update; F = F0 when *bad) ---------------------------
__global__ __launch_bounds__(256) void k_gsmf_update(int D, int KF, const double* __restrict__ Ft,
                                                     const double* __restrict__ Fs, const double* __restrict__ F0,
                                                     int ldf0, double* __restrict__ F, int ldf,
                                                     const int* __restrict__ bad) {
    constexpr int RS = 66;
    __shared__ double FA[64 * RS];
    __shared__ double FB[64 * RS];
    const int ntiles = (D + 63) >> 6;
    const int ti = blockIdx.x / ntiles, tj = blockIdx.x % ntiles;
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

// ---- fast rank-n update (D % 64 == 0, n = 32 NP): F = F0 + Rt^T Fs, the new mean, the revert passthrough ---
// One 512-thread workgroup per 64 x 64 tile (two waves per SIMD for the fp64 MFMA rate); wave w owns the
// 16-row block w >> 1 and the two 16-column blocks of half w & 1.  Every global load of the workgroup -- the F0
// tile in accumulator layout and all n rows of both operand tiles -- is issued in one batch; the operand rows
// then pass through LDS 32 at a time, [k][80] (64 columns + 16 pad: both MFMA operand reads conflict-free).
// Tile row 0 also writes mu = mu0 + mean_b (U Fm)_b (rows B..2B-1 of Tm) for its 64 columns, and workgroup 0
// counts the revert.  When *bad (the 2B x 2B positive-definite test failed) F = F0 and mu = mu0.
template <int NP>
__global__ __launch_bounds__(512) void k_gsmf_update_fast(int D, int B, const double* __restrict__ Rt,
                                                          const double* __restrict__ Fs,
                                                          const double* __restrict__ F0, int ldf0,
                                                          double* __restrict__ F, int ldf,
                                                          const double* __restrict__ Tm,
                                                          const double* __restrict__ mu0, double* __restrict__ mu,
                                                          const int* __restrict__ bad, int* __restrict__ n_reverts) {
    constexpr int RS = 80, KP = 32;
    __shared__ __attribute__((aligned(16))) double sm[2 * KP * RS];
    const int nt = D >> 6;
    const int ti = blockIdx.x / nt, tj = blockIdx.x % nt;
    const int I0 = ti * 64, J0 = tj * 64;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int wr = w >> 1, wc = w & 1;
    // ---- all global loads ----
    double f0[2][4];
    const size_t frow = (size_t)(I0 + 16 * wr + ks);
    const int fcol = J0 + 32 * wc + c;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) f0[blk][r] = F0[(frow + 4 * r) * ldf0 + fcol + 16 * blk];
    v2d ga[NP][2], gb[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int u = q * 512 + tid, row = u >> 5, c2 = 2 * (u & 31);
            ga[p][q] = *reinterpret_cast<const v2d*>(Rt + (size_t)(KP * p + row) * D + I0 + c2);
            gb[p][q] = *reinterpret_cast<const v2d*>(Fs + (size_t)(KP * p + row) * D + J0 + c2);
        }
    const int skip = *bad;
    double msum = 0.0;
    if (ti == 0) {                               // partial column sums of rows B + g, B + g + 8, ... of Tm
        const int g = tid >> 6, col = J0 + (tid & 63);
        for (int b = g; b < B; b += 8) msum += Tm[(size_t)(B + b) * D + col];
    }
    if (blockIdx.x == 0 && tid == 0 && skip && n_reverts) *n_reverts += 1;
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p > 0) __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int u = q * 512 + tid, row = u >> 5, c2 = 2 * (u & 31);
            *reinterpret_cast<v2d*>(sm + row * RS + c2) = ga[p][q];
            *reinterpret_cast<v2d*>(sm + (KP + row) * RS + c2) = gb[p][q];
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
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        F[(frow + 4 * r) * ldf + fcol] = skip ? f0[0][r] : f0[0][r] + acc0[r];
        F[(frow + 4 * r) * ldf + fcol + 16] = skip ? f0[1][r] : f0[1][r] + acc1[r];
    }
    if (ti == 0) {
        __syncthreads();
        sm[tid] = msum;                          // [8][64]
        __syncthreads();
        if (tid < 64) {
            double s = 0.0;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += sm[g * 64 + tid];
            mu[J0 + tid] = skip ? mu0[J0 + tid] : mu0[J0 + tid] + s / (double)B;
        }
    }
}

// Prepares the 64 x 64 LDS Gram matrix M (row stride STR) for the semi-definite rule of chol64_rows_s: lowers every diagonal
// entry by its rounding floor, M_pp <- M_pp (1 - GSMVI_DEP_TOL), and returns (block-uniform) whether every entry is below
// 2^32 and not NaN.
template <int STR>
__device__ __forceinline__ bool diag_prepare(double* M, int* sh_flag) {
    if (threadIdx.x == 0) *sh_flag = 1;
    __syncthreads();
    if (threadIdx.x < 64) {
        const double d = M[threadIdx.x * STR + threadIdx.x];
        M[threadIdx.x * STR + threadIdx.x] = d - GSMVI_DEP_TOL * d;
        if (!(d < 4294967296.0)) *sh_flag = 0;
    }
    __syncthreads();
    return *sh_flag != 0;
}

// ---- everything small in ONE workgroup (n = 2B <= 64), eight waves ---------------------------------------------------
//   Gamma -> Rg (Cholesky) -> A' = I + Rg J Rg^T -> T (Cholesky, the PD test) -> K = Rg^-1 (T - I) Rg^-T.
// All matrices live in LDS ([64][TS], padded with the identity beyond n).  Waves 0-3 are the Cholesky team, waves 4-7 helpers:
//   * W = Rg^-T (64 substitution steps, one column per QUAD of lanes, the pivot value broadcast inside the quad by DPP) runs
//     on the helper waves WHILE the Cholesky team factors A': the helpers execute one barrier per substitution step so that
//     the workgroup barrier counts match (s_barrier counts waves, not program counters);
//   * the three MFMA phases (A', P = (T - I) W, K = W^T P) use all eight waves (two per SIMD: the fp64 MFMA pipe
//     delivers 46 TF chip-wide there against 34 TF with one), row blocks split by parity between the two teams.
// *bad = 1 if either Cholesky fails (NaN, or M not positive definite) and K is then irrelevant.
// NP forward-substitution steps of W = Rg^-T for the calling quad's column, ONE workgroup barrier per step (matches the
// per-pivot barrier of chol64_rows_s running on the other four waves)
template <int NP>
__device__ __forceinline__ void wsubst_steps(double (&x)[16], int sq, const double* Rs, const double* rinv_g) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        __syncthreads();
        const int pr = p >> 2, pq = p & 3;
        const double mine = x[pr] * rinv_g[p];
        if (sq == pq) x[pr] = mine;
        const double xp = quad_bcast_rt<0>(mine, pq);       // DPP, not ds_bpermute: the chain stays off the LDS pipeline
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (4 * r + 3 > p) {
                const int t = sq + 4 * r;
                const double rv = Rs[p * TS + t];
                x[r] -= (t > p) ? rv * xp : 0.0;
            }
    }
}

__global__ __launch_bounds__(512) void k_gsmf_small8(int n, int B, const double* __restrict__ Gam,
                                                     double* __restrict__ Kmat, int* __restrict__ bad_out,
                                                     unsigned long long* __restrict__ stamps) {
#define SMALL_STAMP(k)                                                                      \
    do {                                                                                    \
        if (stamps && threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memrealtime();        \
    } while (0)
    SMALL_STAMP(0);
    __shared__ __attribute__((aligned(16))) double Rs[64 * TS];
    __shared__ __attribute__((aligned(16))) double Ts[64 * TS];
    __shared__ __attribute__((aligned(16))) double Ps[64 * TS];
    __shared__ double rinv_g[64], rinv_t[64];
    __shared__ int fail_g, fail_t;
    const int tid = threadIdx.x;
    const bool team = tid < 256;                       // Cholesky team (waves 0-3) / helpers (waves 4-7)
    {
        double g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = tid + 512 * k, i = e >> 6, q = e & 63;
            g[k] = Gam[(size_t)(i < n ? i : n - 1) * n + (q < n ? q : n - 1)];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = tid + 512 * k, i = e >> 6, q = e & 63;
            Rs[i * TS + q] = (i < n && q < n && q >= i) ? g[k] : (i == q ? 1.0 : 0.0);
        }
    }
    if (tid < 64) rinv_g[tid] = rinv_t[tid] = 1.0;
    __syncthreads();
    SMALL_STAMP(1);
    // Gamma = Rt Rt^T is only positive SEMI-definite when rows of [Z; U] are linearly dependent (an isotropic state on
    // an isotropic target makes every u_b - a_b z_b parallel to mu - m; the exact fixed point makes U = -Z ...): the
    // dependent rows drop out of Rg (zero row, zero diagonal) and get a unit diagonal in the substitution below, which
    // leaves C^T C = I + Rt^T J Rt intact (factor form).
    const bool moderate = diag_prepare<TS>(Rs, &fail_t);
    if (team) chol64_rows_s<TS, true>(Rs, rinv_g, n, &fail_g, moderate);
    else chol64_helper_idle<TS>(n);
    SMALL_STAMP(2);
    for (int e = tid; e < 64 * 64; e += 512) {
        const int i = e >> 6, q = e & 63;
        if (q < i) Rs[i * TS + q] = 0.0;
    }
    if (tid < 64 && rinv_g[tid] == 0.0) rinv_g[tid] = 1.0;
    __syncthreads();
    const int w = tid >> 6, l = tid & 63, cc = l & 15, ks = l >> 4;
    const int wj = w & 3, g2 = w >> 2;                 // column block of this wave, team index (row-block parity)
    {   // A' = I + (Rg J) Rg^T into Ts;  (Rg J)[i][k] = (1/B)(k < B ? Rg[i][B+k] : Rg[i][k-B] - Rg[i][k]), on the MFMA pipe.
        // A' is symmetric (the mirror is written too); Rg is upper triangular, so Rg[j][k] = 0 for k < 16 j: only the
        // k-blocks j..3 contribute.  Wave (wj, g2) computes the blocks (ib, wj), ib <= wj, ib % 2 == g2
        const double invB = 1.0 / (double)B;
        v4d acc[2];
        acc[0] = acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
        const double* brow = Rs + (16 * wj + cc) * TS;
        for (int kb = wj; kb < 4; ++kb) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int k = 16 * kb + 4 * s4 + ks;
                const int k1 = (k < B) ? B + k : k - B;
                const double bv = brow[k];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ib = 2 * h + g2;
                    if (ib <= wj) {
                        const double* arow = Rs + (16 * ib + cc) * TS;
                        const double a1 = arow[k1 < 64 ? k1 : 63], a0 = arow[k];
                        const double a = (k < n) ? ((k < B) ? a1 : a1 - a0) : 0.0;
                        acc[h] = GSMVI_MFMA_F64(a, bv, acc[h]);
                    }
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ib = 2 * h + g2;
            if (ib <= wj) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ib + ks + 4 * r, j = 16 * wj + cc;
                    const double v = (i == j ? 1.0 : 0.0) + ((i < n && j < n) ? acc[h][r] * invB : 0.0);
                    Ts[i * TS + j] = v;
                    if (ib != wj) Ts[j * TS + i] = v;
                }
            }
        }
    }
    __syncthreads();
    SMALL_STAMP(3);
    // ---- Cholesky of A' (the positive-definite test) || W = Rg^-T by substitution ----
    double x[16];
    const int sc = (tid & 255) >> 2, sq = tid & 3;     // helper thread: column sc of W, quad lane sq (rows sq, sq+4, ..)
    if (team) {
        chol64_rows_s<TS>(Ts, rinv_t, n, &fail_t);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = (sq + 4 * r == sc) ? 1.0 : 0.0;
        // one instantiation per pivot count of the factorisation running beside us (16 ceil(n/16)): the loop must be
        // fully unrolled (x[] is indexed with compile-time constants) and free of per-step branches (a predicated
        // body measured 22 us for the pair at n = 64 against 16.6 us straight-line); rows beyond n are identity.
        switch ((n + 15) >> 4) {
            case 1: wsubst_steps<16>(x, sq, Rs, rinv_g); break;
            case 2: wsubst_steps<32>(x, sq, Rs, rinv_g); break;
            case 3: wsubst_steps<48>(x, sq, Rs, rinv_g); break;
            default: wsubst_steps<64>(x, sq, Rs, rinv_g); break;
        }
        __syncthreads();
        __syncthreads();
        __syncthreads();
    }
    SMALL_STAMP(4);
    const int bad = (fail_g != 0) || (fail_t != 0);
    if (tid == 0) *bad_out = bad;
    if (bad) return;                                   // block-uniform
    // every helper has finished reading Rg (its 64 steps precede the last three barriers): Rs <- W, Ts <- T - I
    if (!team) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Rs[(sq + 4 * r) * TS + sc] = x[r];
    } else {
        for (int e = tid; e < 64 * 64; e += 256) {
            const int i = e >> 6, j = e & 63;
            if (j < i) Ts[i * TS + j] = 0.0;
            else if (j == i) Ts[i * TS + j] -= 1.0;
        }
    }
    __syncthreads();
    SMALL_STAMP(5);
    {
        // P[i][j] = sum_k (T - I)[i][k] W[k][j]: k-blocks max(i, j)..3; wave (wj, g2): column block wj, row blocks ib % 2 == g2
        v4d acc[2];
        acc[0] = acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
        for (int kb = wj; kb < 4; ++kb) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int k = 16 * kb + 4 * s4 + ks;
                const double bv = Rs[k * TS + 16 * wj + cc];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ib = 2 * h + g2;
                    if (ib <= kb) acc[h] = GSMVI_MFMA_F64(Ts[(16 * ib + cc) * TS + k], bv, acc[h]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ps[(16 * (2 * h + g2) + ks + 4 * r) * TS + 16 * wj + cc] = acc[h][r];
        __syncthreads();
        // K[i][j] = sum_k W[k][i] P[k][j]: k-blocks i..3
        acc[0] = acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int k = 16 * kb + 4 * s4 + ks;
                const double bv = Ps[k * TS + 16 * wj + cc];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ib = 2 * h + g2;
                    if (ib <= kb) acc[h] = GSMVI_MFMA_F64(Rs[k * TS + 16 * ib + cc], bv, acc[h]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * (2 * h + g2) + ks + 4 * r, j = 16 * wj + cc;
                if (i < n && j < n) Kmat[(size_t)i * n + j] = acc[h][r];
            }
    }
    SMALL_STAMP(6);
#undef SMALL_STAMP
}

// ---- K = Rg^-1 (T - I) Rg^-T = W^T (T - I) W, W = Rg^-T, for 64 < n <= 128 -----------------------------------
// k_gsmf_kmat_big: W by forward substitution.  One workgroup handles 16 columns, SIXTEEN lanes per column (lane
// q of the group owns rows q, q+16, ..); Rg is resident in LDS ([128][130], padded with the identity beyond
// n); each substitution step broadcasts the pivot value inside the 16-lane group.  The two products then run
// on the MFMA pipe (k_gsmf_gemm128), instead of a second and a third 128-step substitution.
__device__ __forceinline__ void kmat_load_lds(double* Mt, double* rinv, const double* __restrict__ src, int n,
                                              bool want_rinv) {
    load_upper128(Mt, src, n);
    __syncthreads();
    if (want_rinv && threadIdx.x < 128) {              // a zero diagonal marks a dependent row of [Z; U]: unit pivot
        const double dg = Mt[threadIdx.x * 130 + threadIdx.x];
        rinv[threadIdx.x] = dg == 0.0 ? 1.0 : 1.0 / dg;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_gsmf_kmat_big(int n, const double* __restrict__ Rg,
                                                       double* __restrict__ Wout,
                                                       const int* __restrict__ info_g,
                                                       const int* __restrict__ info_t, int* __restrict__ bad_out) {
    __shared__ __attribute__((aligned(16))) double Mt[128 * 130];
    __shared__ double rinv[128];
    const int bad = (*info_g != 0) || (*info_t != 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) *bad_out = bad;
    if (bad) return;                                         // block-uniform
    const int tid = threadIdx.x, c = tid >> 4, q = tid & 15, grp = tid & 48;   // grp: first lane of the group in the wave
    const int cg = blockIdx.x * 16 + c;
    double x[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = (q + 16 * r == cg) ? 1.0 : 0.0;
    // The pivot loops are blocked 8 x 16: the outer block index is unrolled (static register indices), the 16
    // pivots inside a block stay a rolled loop -- fully unrolled, the scheduler hoists every LDS read and spills.
    // phase 1: x = Rg^-T e_c   (L = Rg^T, L[t][p] = Rg[p][t]: row p of Rg)
    kmat_load_lds(Mt, rinv, Rg, n, true);
#pragma unroll
    for (int pb = 0; pb < 8; ++pb) {
#pragma unroll 2
        for (int pq = 0; pq < 16; ++pq) {
            const int p = 16 * pb + pq;
            const double mine = x[pb] * rinv[p];
            if (q == pq) x[pb] = mine;
            const double xp = __shfl(mine, grp | pq, 64);
            const double* row = Mt + p * 130 + q;
            x[pb] -= (q > pq) ? row[16 * pb] * xp : 0.0;
#pragma unroll
            for (int r = pb + 1; r < 8; ++r) x[r] -= row[16 * r] * xp;
        }
    }
    // W = Rg^-T (lower triangular), column cg in this group's registers
    if (cg < n) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int t = q + 16 * r;
            if (t < n) Wout[(size_t)t * n + cg] = x[r];
        }
    }
}

// ---- C = op(A) B for n x n matrices (n <= 128), one 16 x 16 block per wave -------------------------------
//   MODE 0:  C = (A - I) B     (A = T upper triangular: P = (T - I) W)
//   MODE 1:  C = A^T B         (A = W lower triangular: K = W^T P)
// All operand fragments of a block are loaded from L2 in one batch, then one MFMA chain of n/4 steps.
template <int MODE>
__global__ __launch_bounds__(256) void k_gsmf_gemm128(int n, const double* __restrict__ A,
                                                      const double* __restrict__ Bm, double* __restrict__ Cm,
                                                      const int* __restrict__ bad) {
    if (*bad) return;
    const int nb = (n + 15) >> 4;
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= nb * nb) return;                                  // wave-uniform
    const int i0 = (blk / nb) * 16, j0 = (blk % nb) * 16;
    const int l = threadIdx.x & 63, cc = l & 15, ks = l >> 4;
    double a[32], b[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const int k = 4 * s + ks;
        const int kc = k < n ? k : n - 1, ic = (i0 + cc) < n ? i0 + cc : n - 1, jc = (j0 + cc) < n ? j0 + cc : n - 1;
        const double av = MODE == 0 ? A[(size_t)ic * n + kc] - (ic == kc ? 1.0 : 0.0) : A[(size_t)kc * n + ic];
        const double bv = Bm[(size_t)kc * n + jc];
        const bool in = k < n && (i0 + cc) < n;
        a[s] = in ? av : 0.0;
        b[s] = (k < n && (j0 + cc) < n) ? bv : 0.0;
    }
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 32; s += 2) {
        acc0 = GSMVI_MFMA_F64(a[s], b[s], acc0);
        acc1 = GSMVI_MFMA_F64(a[s + 1], b[s + 1], acc1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ks + 4 * r, j = j0 + cc;
        if (i < n && j < n) Cm[(size_t)i * n + j] = acc0[r] + acc1[r];
    }
}

// ---- new mean and the revert passthrough for the K-matrix path -----------------------------------------
__global__ __launch_bounds__(256) void k_gsmf_mean(int D, int B, const double* __restrict__ Tm,
                                                   const double* __restrict__ mu0, double* __restrict__ mu,
                                                   const int* __restrict__ bad, int* __restrict__ n_reverts) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= D) return;
    if (j == 0 && n_reverts && *bad) *n_reverts += 1;      // one launch per update on the stream: no race
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = 0;
    for (; b + 8 <= B; b += 8) {                   // eight independent loads in flight per trip
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = Tm[(size_t)(B + b + k) * D + j];
        s0 += v[0] + v[4];
        s1 += v[1] + v[5];
        s2 += v[2] + v[6];
        s3 += v[3] + v[7];
    }
    for (; b < B; ++b) s0 += Tm[(size_t)(B + b) * D + j];
    mu[j] = (*bad) ? mu0[j] : mu0[j] + ((s0 + s1) + (s2 + s3)) / (double)B;
}

int gsmvi_potrf_impl(gsmvi_ctx* ctx, hipStream_t st, int D, const double* S, int lds, double* R, int ldr,
                     int* info_dev);

static int chk(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gsmvi_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}

// Pp[kc][B][mrows] partial slabs of A M^T (A: B x D, M: mrows x D); *kc_out slabs.
int gsmvi_panel_t_product(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* A, int lda, const double* M,
                          int ldm, int mrows, double* Pp, int* kc_out) {
    const int MT = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
    const int CH = (MT == 4) ? 128 : 256;
    const bool fast_t = !ctx->tune_no_fast && D % 64 == 0 && mrows % 16 == 0 && lda % 2 == 0 && ldm % 2 == 0 &&
                        (reinterpret_cast<uintptr_t>(A) & 15u) == 0 && (reinterpret_cast<uintptr_t>(M) & 15u) == 0;
    const int strips = (mrows + 15) / 16, nchunks = (D + CH - 1) / CH, zb = (B + 16 * MT - 1) / (16 * MT);
    int kc = (2 * ctx->num_cu + strips * zb - 1) / (strips * zb);
    if (kc > nchunks) kc = nchunks;
    if (kc > GSMVI_MAX_KC) kc = GSMVI_MAX_KC;
    if (kc < 1) kc = 1;
    const int cpw = (nchunks + kc - 1) / kc;
    kc = (nchunks + cpw - 1) / cpw;
    *kc_out = kc;
    const dim3 grid(strips, kc, zb);
    if (fast_t) {
        if (MT == 1) hipLaunchKernelGGL((k_panel_t_fast<1, 256>), grid, dim3(512), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
        else if (MT == 2) hipLaunchKernelGGL((k_panel_t_fast<2, 256>), grid, dim3(512), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
        else hipLaunchKernelGGL((k_panel_t_fast<4, 128>), grid, dim3(512), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
    } else if (MT == 1) hipLaunchKernelGGL((k_panel_t<1, 256>), grid, dim3(256), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
    else if (MT == 2) hipLaunchKernelGGL((k_panel_t<2, 256>), grid, dim3(256), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
    else hipLaunchKernelGGL((k_panel_t<4, 128>), grid, dim3(256), 0, st, D, B, A, lda, M, ldm, Pp, cpw, mrows);
    return chk("k_panel_t");
}

// ---- records of the batch-sharded factor path: rec[b] = [ x_b - mu (D) | u_b (D) | (u Fm)_b (D) ] ------------
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

// Rt = [Z; U], its transpose Rtt (D x n), Tm = [X - mu; U Fm] from the replicated draws and ALL records
__global__ __launch_bounds__(256) void k_gsmf_unpack(int D, int B, const double* __restrict__ Z, int ldz,
                                                     const double* __restrict__ rec, int ldrec,
                                                     double* __restrict__ Rt, double* __restrict__ Rtt, int nq,
                                                     double* __restrict__ Tm) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= D) return;
    const double* rb = rec + (size_t)b * ldrec;
    const double z = Z[(size_t)b * ldz + i], u = rb[D + i];
    Rt[(size_t)b * D + i] = z;
    Rt[(size_t)(B + b) * D + i] = u;
    Rtt[(size_t)i * nq + b] = z;
    Rtt[(size_t)i * nq + B + b] = u;
    Tm[(size_t)b * D + i] = rb[i];
    Tm[(size_t)(B + b) * D + i] = rb[2 * D + i];
}

// Front half: per-sample stage for B samples.  Fills Rt = [Z; U] (2B x D), Rtt (D x 2B) and Tm = [X - mu; U Fm]
// in the workspace (layout for n = 2B rows).
static int factor_front(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* Z, int ldz, const double* X, int ldx,
                        const double* G, int ldg, const double* mu0, const double* F0, int ldf0) {
    const int n = 2 * B, nq = n;
    double* Rt = ctx->sg;                          // n x D   [Z; U]
    double* Tm = Rt + (size_t)n * D;               // n x D   [X - mu; U Fm]
    double* Fs = Tm + (size_t)n * D;               // n x D
    double* Rtt = Fs + (size_t)n * D;              // D x n
    // W = G Fm^T
    int kc = 1;
    int rc = gsmvi_panel_t_product(ctx, st, D, B, G, ldg, F0, ldf0, D, ctx->pp, &kc);
    if (rc) return rc;
    {
        const int ept = (D + 1023) / 1024;
#define GS(E) hipLaunchKernelGGL(k_gsmf_scalars<E>, dim3(B), dim3(1024), 0, st, D, B, kc, Z, ldz, X, ldx, mu0, ctx->pp, Rt, Rtt, nq, Tm)
        if (ept <= 1) GS(1); else if (ept <= 2) GS(2); else if (ept <= 4) GS(4); else if (ept <= 8) GS(8); else GS(16);
#undef GS
    }
    if ((rc = chk("k_gsmf_scalars"))) return rc;
    // bottom half of Tm: U Fm
    return gsmvi_panel_product_out(ctx, st, D, D, B, Rt + (size_t)B * D, D, nullptr, 1.0, F0, ldf0, nullptr,
                                   Tm + (size_t)B * D, D);
}

static int factor_back(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* mu0, const double* F0, int ldf0,
                       double* mu, double* F, int ldf, int* info_dev, int* n_reverts_dev);

int gsmvi_factor_impl(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* Z, int ldz, const double* X, int ldx,
                      const double* G, int ldg, const double* mu0, const double* F0, int ldf0, double* mu, double* F,
                      int ldf, int* info_dev, int* n_reverts_dev) {
    int rc = factor_front(ctx, st, D, B, Z, ldz, X, ldx, G, ldg, mu0, F0, ldf0);
    if (rc) return rc;
    return factor_back(ctx, st, D, B, mu0, F0, ldf0, mu, F, ldf, info_dev, n_reverts_dev);
}

// Batch-sharded form, stage 1: this rank's B_local samples -> records.
int gsmvi_factor_local_impl(gsmvi_ctx* ctx, hipStream_t st, int D, int Bl, const double* Z, int ldz, const double* X,
                            int ldx, const double* G, int ldg, const double* mu0, const double* F0, int ldf0,
                            double* rec, int ldrec) {
    int rc = factor_front(ctx, st, D, Bl, Z, ldz, X, ldx, G, ldg, mu0, F0, ldf0);
    if (rc) return rc;
    const double* Rt = ctx->sg;
    const double* Tm = Rt + (size_t)2 * Bl * D;
    hipLaunchKernelGGL(k_gsmf_pack, dim3((D + 255) / 256, Bl), dim3(256), 0, st, D, Bl, Rt, Tm, rec, ldrec);
    return chk("k_gsmf_pack");
}

// Batch-sharded form, stage 2: every replica applies the combined update from ALL B records and the replicated Z.
int gsmvi_factor_apply_impl(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* Z, int ldz, const double* rec,
                            int ldrec, const double* mu0, const double* F0, int ldf0, double* mu, double* F, int ldf,
                            int* info_dev, int* n_reverts_dev) {
    const int n = 2 * B;
    double* Rt = ctx->sg;
    double* Tm = Rt + (size_t)n * D;
    double* Fs = Tm + (size_t)n * D;
    double* Rtt = Fs + (size_t)n * D;
    hipLaunchKernelGGL(k_gsmf_unpack, dim3((D + 255) / 256, B), dim3(256), 0, st, D, B, Z, ldz, rec, ldrec, Rt, Rtt, n, Tm);
    int rc = chk("k_gsmf_unpack");
    if (rc) return rc;
    return factor_back(ctx, st, D, B, mu0, F0, ldf0, mu, F, ldf, info_dev, n_reverts_dev);
}

// Back half: from Rt, Rtt, Tm (n = 2B rows) to (mu, F).
static int factor_back(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* mu0, const double* F0, int ldf0,
                       double* mu, double* F, int ldf, int* info_dev, int* n_reverts_dev) {
    const int n = 2 * B, nq = n;                   // n is even
    // workspace carve: ctx->sg holds 4*rmax*max_D doubles (rmax = 2B+8; ws_sizes in gsmvi_abi.hip)
    double* Rt = ctx->sg;                          // n x D   [Z; U]
    double* Tm = Rt + (size_t)n * D;               // n x D   [X - mu; U Fm]
    double* Fs = Tm + (size_t)n * D;               // n x D
    double* Rtt = Fs + (size_t)n * D;              // D x n
    double* Gam = ctx->small;                      // n x n
    double* Rg = Gam + (size_t)n * n;
    double* Ap = Rg + (size_t)n * n;
    double* Tt = Ap + (size_t)n * n;
    int* info_g = ctx->ints;
    int* info_t = ctx->ints + 1;
    int rc, kc2 = 1;
    // Gamma = Rt Rt^T
    if ((rc = gsmvi_panel_product_out(ctx, st, D, n, n, Rt, D, nullptr, 1.0, Rtt, nq, nullptr, Gam, n))) return rc;
    if (n <= 64) {
        // everything small in one workgroup, then Fs = K Tm as one skinny GEMM (K = n)
        double* Kmat = Rg;                         // reuse the n x n slot
        hipLaunchKernelGGL(k_gsmf_small8, dim3(1), dim3(512), 0, st, n, B, Gam, Kmat, info_dev,
                           (ctx->tune_cov_dbg & 128) ? reinterpret_cast<unsigned long long*>(ctx->pp) : nullptr);
        if ((rc = chk("k_gsmf_small"))) return rc;
        // inner dimension n <= one chunk, so there is exactly one slab: it is written straight into Fs
        if ((rc = gsmvi_panel_product_nc(ctx, st, nullptr, n, D, n, Kmat, n, nullptr, 1.0, Tm, D, Fs, &kc2)))
            return rc;
        if (kc2 != 1) {
            gsmvi_set_error("%s: %s", "gsmvi_factor_impl", "K Tm product was split (internal error)");
            return GSMVI_ERR_UNSUPPORTED;
        }
    } else {
        // 64 < n <= 128: the two n x n Choleskys one workgroup each, K in its own kernel, then the same skinny
        // GEMM Fs = K Tm
        double* Kmat = Gam;                        // Gamma is dead once Rg exists
        double* Wm = Ap;                           // A' is dead once T exists
        double* Pm = Tt + (size_t)n * n;           // fifth n x n slot of the small-matrix workspace
        hipLaunchKernelGGL(k_chol128<true>, dim3(1), dim3(512), 0, st, n, Gam, Rg, info_g);   // Gram matrix: semi-definite rule
        hipLaunchKernelGGL(k_gsmf_small_a, dim3((n + 15) / 16, (n + 15) / 16), dim3(256), 0, st, n, B, Rg, info_g, Ap);
        hipLaunchKernelGGL(k_chol128<false>, dim3(1), dim3(512), 0, st, n, Ap, Tt, info_t);
        if ((rc = chk("k_chol128"))) return rc;
        hipLaunchKernelGGL(k_gsmf_kmat_big, dim3((n + 15) / 16), dim3(256), 0, st, n, Rg, Wm, info_g, info_t,
                           info_dev);
        if ((rc = chk("k_gsmf_kmat_big"))) return rc;
        {
            const int nb = (n + 15) / 16, gw = (nb * nb + 3) / 4;
            hipLaunchKernelGGL(k_gsmf_gemm128<0>, dim3(gw), dim3(256), 0, st, n, Tt, Wm, Pm, info_dev);   // P = (T - I) W
            hipLaunchKernelGGL(k_gsmf_gemm128<1>, dim3(gw), dim3(256), 0, st, n, Wm, Pm, Kmat, info_dev); // K = W^T P
            if ((rc = chk("k_gsmf_gemm128"))) return rc;
        }
        // inner dimension n <= one chunk, so there is exactly one slab: it is written straight into Fs
        if ((rc = gsmvi_panel_product_nc(ctx, st, nullptr, n, D, n, Kmat, n, nullptr, 1.0, Tm, D, Fs, &kc2)))
            return rc;
        if (kc2 != 1) {
            gsmvi_set_error("%s: %s", "gsmvi_factor_impl", "K Tm product was split (internal error)");
            return GSMVI_ERR_UNSUPPORTED;
        }
    }
    const int nt = (D + 63) / 64;
    if (!ctx->tune_no_fast && D % 64 == 0 && (n == 32 || n == 64 || n == 128)) {
        // the fast kernel also writes the mean and counts the revert
#define UF(NPV) hipLaunchKernelGGL(k_gsmf_update_fast<NPV>, dim3(nt * nt), dim3(512), 0, st, D, B, Rt, Fs, F0, ldf0, F, ldf, Tm, mu0, mu, info_dev, n_reverts_dev)
        if (n == 32) UF(1); else if (n == 64) UF(2); else UF(4);
#undef UF
        return chk("k_gsmf_update_fast");
    }
    hipLaunchKernelGGL(k_gsmf_mean, dim3((D + 255) / 256), dim3(256), 0, st, D, B, Tm, mu0, mu, info_dev, n_reverts_dev);
    if ((rc = chk("k_gsmf_mean"))) return rc;
    hipLaunchKernelGGL(k_gsmf_update, dim3(nt * nt), dim3(256), 0, st, D, n, Rt, Fs, F0, ldf0, F, ldf, info_dev);
    return chk("k_gsmf_update");
}
