// The 2B x 2B chain of the factor-form update (n = 2B <= 64) as a device function, so that it can run either as its own
// one-workgroup launch (k_gsmf_small16, gsmvi_factor.hip) or as a RIDER workgroup inside a panel-product launch
// (k_panel_fast<.., RIDER>, gsmvi_fast.hip).  512 threads; every thread of the workgroup must call it.
#pragma once
#include "gsmvi_common.h"
#include "gsmvi_ctx.h"
#include "gsmvi_chol64.h"
#include "gsmvi_chol64b.h"

// The per-sample scalars from the Gram matrix Gamma1 = [Z; V][Z; V]^T (zz = Gamma1[b][b], zv = Gamma1[b][B+b],
// vv = Gamma1[B+b][B+b]):  u_b = beta z_b + alpha v_b  (gsm_numpy.py:8-17 in whitened form, see the file header)
__device__ __forceinline__ void gsmf_coefs(double zz, double zv, double vv, double* alpha, double* beta) {
    const double zw = zv - zz, ww = vv - 2.0 * zv + zz, wv = vv - zv;
    const double rho = 0.5 * sqrt(1.0 + 4.0 * (ww + zw * zw)) - 0.5;
    const double den = 1.0 + rho - zw;
    *alpha = 1.0 / (1.0 + rho);
    *beta = (wv / den) / (1.0 + rho);
}

// ---- everything small in ONE workgroup (n = 2B <= 64), eight waves ---------------------------------------------------
//   Gamma -> Rg, W = Rg^-T (blocked Cholesky of [Gamma | I]) -> A' = I + Rg J Rg^T -> T (Cholesky, the PD test)
//   -> K = Rg^-1 (T - I) Rg^-T = W^T (T - I) W.
// Round 3: both factorisations are chol64_blk (gsmvi_chol64b.h: 16-pivot panels in one wave's registers, no workgroup
// barrier on the pivot chain; W comes out of the augmented identity columns, so the 64-step substitution of round 2 is
// gone); round 2's k_gsmf_small8 spent 13 + 15 us of its 39 us in two chol64_rows_s calls.
// LDS: E1 [64][146] = [Gamma -> Rg | W], later P = (T - I) W in its left half; E2 [64][82] = A' -> T.  Padded with the
// identity beyond n.  Gamma = Rt Rt^T is only positive SEMI-definite when rows of [Z; U] are linearly dependent (an
// isotropic state on an isotropic target makes every u_b - a_b z_b parallel to mu - m; the exact fixed point makes U = -Z):
// the dependent rows drop out of Rg (zero row, zero diagonal) and get a unit pivot in W, which leaves
// C^T C = I + Rt^T J Rt intact (factor form); the rule is off (dependent => failure) when a diagonal
// entry of Gamma reaches 2^32 (fixture G4).  *bad_out = 1 if either Cholesky fails (NaN, or M not positive definite);
// K is then irrelevant.
#define GSMF_ES1 146
#define GSMF_ES2 82
// jmode = 1 (factor-form BaM, gsmvi_bam.hip): Gp is the Gram matrix of Rt = [Vw; Zw] itself (S = I), J = diag(I_B, -I_B)
// (M = I + Vw^T Vw - Zw^T Zw), the coefficients come out as zeros (the mean is BaM's own).
// eight entries per thread of the 64 x 64 window of sum_{kc < kcg} slab[kc] (n x n each): every load of every slab in one batch
// (clamped slab / row / column indices), summed in slab order
template <int KCT>
__device__ __forceinline__ void gsmf_sum_gram_slabs(const double* __restrict__ Gp, int kcg, int n, int tid, double (&g)[8]) {
    double t[KCT][8];
#pragma unroll
    for (int kc = 0; kc < KCT; ++kc)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = tid + 512 * k, i = e >> 6, q = e & 63;
            t[kc][k] = Gp[(size_t)(kc < kcg ? kc : kcg - 1) * n * n + (size_t)(i < n ? i : n - 1) * n + (q < n ? q : n - 1)];
        }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double a = 0.0;
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc) a += (kc < kcg) ? t[kc][k] : 0.0;
        g[k] = a;
    }
}

#define GSMF_SMALL16_LDS (64 * GSMF_ES1 + 64 * GSMF_ES2 + CHOLB_SCRATCH_DOUBLES(true))     // doubles: 141.6 KB
__device__ __forceinline__ void gsmf_small16_body(double* __restrict__ lds, int n, int B, const double* __restrict__ Gp,
                                                  int kcg, double* __restrict__ Kmat, double* __restrict__ coef,
                                                  int* __restrict__ bad_out, unsigned long long* __restrict__ stamps,
                                                  int jmode, const int* __restrict__ prior_bad,
                                                  const double* __restrict__ Pi = nullptr,
                                                  const double* __restrict__ R11g = nullptr,
                                                  const double* __restrict__ W11g = nullptr) {
#define SMALL_STAMP(k)                                                                      \
    do {                                                                                    \
        if (stamps && threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memrealtime();        \
    } while (0)
    SMALL_STAMP(0);
    constexpr int ES1 = GSMF_ES1, ES2 = GSMF_ES2;
    double* E1 = lds;                                  // [64][ES1]  (the caller's LDS block, GSMF_SMALL16_LDS doubles, 16-B aligned)
    double* E2 = lds + 64 * ES1;                       // [64][ES2]
    double* scr = E2 + 64 * ES2;                       // chol64_blk scratch, two column sets
    __shared__ double s_alpha[32], s_beta[32];
    __shared__ int fail_g, fail_t, sh_moderate;
    const int tid = threadIdx.x;
    if (tid == 0) sh_moderate = 1;
    {   // Gamma1 = sum of the kcg split-K slabs of the Gram product (n x n each, full matrix) -> E2, raw
        double g[8];
        if (kcg == 1) {                                // block-uniform: the finished matrix (the lean path's side job)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = tid + 512 * k, i = e >> 6, q = e & 63;
                g[k] = Gp[(size_t)(i < n ? i : n - 1) * n + (q < n ? q : n - 1)];
            }
        } else if (kcg <= 2) gsmf_sum_gram_slabs<2>(Gp, kcg, n, tid, g);   // block-uniform; as many loads as there are slabs (rounded up)
        else if (kcg <= 4) gsmf_sum_gram_slabs<4>(Gp, kcg, n, tid, g);
        else gsmf_sum_gram_slabs<GSMVI_MAX_KC>(Gp, kcg, n, tid, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = tid + 512 * k, i = e >> 6, q = e & 63;
            E2[i * ES2 + q] = g[k];
        }
    }
    __syncthreads();
    if (tid < B) {                                     // per-sample scalars (gsm_numpy.py:8-17, whitened; file header)
        double al = 1.0, be = 0.0;
        if (!jmode) gsmf_coefs(E2[tid * ES2 + tid], E2[tid * ES2 + B + tid], E2[(B + tid) * ES2 + B + tid], &al, &be);
        s_alpha[tid] = al;
        s_beta[tid] = be;
        coef[tid] = jmode ? 0.0 : be / (double)B;      // mean: mu' = mu + sum_b coef[b] (x_b - mu) + coef[B + b] (v_b Fm)
        coef[B + tid] = jmode ? 0.0 : al / (double)B;
    }
    __syncthreads();
    // jmode 2 with a GIVEN first block (round 5): in the orthogonal basis [Vw; Zt] the Gram matrix is block diagonal (Zt is
    // orthogonal to the whitened draws by construction; Gamma12 holds rounding only), and its first block Gvv = Vw Vw^T has been
    // factored already, beside BaM's B x B chain ([R11 | W11], bamq_side_body).  The factorisation below then resumes at row B:
    // half the pivots of the chain's first 64-pivot factorisation.  B must be a multiple of the 16-row panel.
    const bool given = jmode == 2 && R11g != nullptr && (B & 15) == 0 && B > 0;
    {   // Gamma = S Gamma1 S^T (upper triangle), S = [[I, 0], [diag(beta), diag(alpha)]]; identity beyond n
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = tid + 512 * k, i = e >> 6, q = e & 63;
            double v = (i == q) ? 1.0 : 0.0;
            if (i < n && q < n && q >= i) {
                if (q < B) v = E2[i * ES2 + q];                                           // Z Z^T
                else if (i < B) {
                    const int b = q - B;
                    v = s_beta[b] * E2[i * ES2 + b] + s_alpha[b] * E2[i * ES2 + q];        // Z U^T
                } else {
                    const int a = i - B, b = q - B;                                      // U U^T
                    v = s_beta[a] * (s_beta[b] * E2[a * ES2 + b] + s_alpha[b] * E2[a * ES2 + q]) +
                        s_alpha[a] * (s_beta[b] * E2[i * ES2 + b] + s_alpha[b] * E2[i * ES2 + q]);
                }
                if (i == q) {                          // rounding floor of the row (gsmvi_chol64.h) and the magnitude guard
                    if (!(v < 4294967296.0)) sh_moderate = 0;
                    v -= GSMVI_DEP_TOL * v;
                }
            } else if (i < n && q < n) v = 0.0;
            if (given && i < B) v = (q < B && q >= i) ? R11g[i * B + q] : 0.0;   // the finished block; Gamma12 = Vw Zt^T is zero
            E1[i * ES1 + q] = v;
        }
        if (given) {                                   // W11 (lower) in the augmented columns of the given rows, zeros beside it
            for (int e = tid; e < B * 64; e += 512) {
                const int i = e >> 6, q = e & 63;
                E1[i * ES1 + 64 + q] = (q <= i) ? W11g[i * B + q] : 0.0;
            }
        }
    }
    __syncthreads();
    SMALL_STAMP(1);
    const bool moderate = sh_moderate != 0;
    chol64_blk<ES1, true, true>(E1, scr, n, &fail_g, moderate, given ? B >> 4 : 0);     // E1 = [Rg | W]
    SMALL_STAMP(2);
    const int w = tid >> 6, l = tid & 63, cc = l & 15, ks = l >> 4;
    const int nblk = (n + 15) >> 4;
    // The three 64^3 products below run one 16 x 16 output block at a time on a wave, the operands of a k-block (four MFMA
    // steps) fetched from LDS into registers one k-block AHEAD of their use; the blocks are dealt to the eight waves by the
    // tables so that every wave has (nearly) the same number of k-blocks -- the triangular structure makes the block costs
    // differ (round 2 gave each wave a fixed column block: up to 2x imbalance, operands read at their point of use).
    // task = 4 * ib + jb + 1 (0 = none)
#define GSMF_T(i, j) (4 * (i) + (j) + 1)
    static constexpr unsigned char TASK_A[8][2] = {   // A'(ib, jb), ib <= jb: cost 4 - jb k-blocks
        {GSMF_T(0, 0), 0}, {GSMF_T(0, 1), 0}, {GSMF_T(1, 1), 0}, {GSMF_T(0, 2), GSMF_T(0, 3)},
        {GSMF_T(1, 2), GSMF_T(1, 3)}, {GSMF_T(2, 2), GSMF_T(2, 3)}, {GSMF_T(3, 3), 0}, {0, 0}};
    static constexpr unsigned char TASK_P[8][3] = {   // P(ib, jb): cost 4 - max(ib, jb)
        {GSMF_T(0, 0), 0, 0}, {GSMF_T(0, 1), GSMF_T(3, 0), 0}, {GSMF_T(1, 0), GSMF_T(3, 1), 0}, {GSMF_T(1, 1), GSMF_T(3, 2), 0},
        {GSMF_T(0, 2), GSMF_T(1, 2), 0}, {GSMF_T(2, 0), GSMF_T(2, 1), 0}, {GSMF_T(2, 2), GSMF_T(0, 3), GSMF_T(1, 3)},
        {GSMF_T(2, 3), GSMF_T(3, 3), 0}};
    static constexpr unsigned char TASK_K[8][2] = {   // K(ib, jb): cost 4 - ib
        {GSMF_T(0, 0), GSMF_T(3, 0)}, {GSMF_T(0, 1), GSMF_T(3, 1)}, {GSMF_T(0, 2), GSMF_T(3, 2)}, {GSMF_T(0, 3), GSMF_T(3, 3)},
        {GSMF_T(1, 0), GSMF_T(2, 0)}, {GSMF_T(1, 1), GSMF_T(2, 1)}, {GSMF_T(1, 2), GSMF_T(2, 2)}, {GSMF_T(1, 3), GSMF_T(2, 3)}};
#undef GSMF_T
    // one output block: acc = sum over the k-blocks kb0 .. nblk-1 of A(ib-rows, k) B(k, jb-cols); fa(k) / fb(k) fetch the
    // operand values of this lane for contraction index k
    auto block_chain = [&](int kb0, auto fa, auto fb) {
        v4d acc = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};   // two chains: a dependent fp64 MFMA waits ~2x its issue time
        double a[4], b[4], an[4], bn[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { a[s4] = fa(16 * kb0 + 4 * s4 + ks); b[s4] = fb(16 * kb0 + 4 * s4 + ks); }
        for (int kb = kb0; kb < nblk; ++kb) {
            if (kb + 1 < nblk) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) { an[s4] = fa(16 * (kb + 1) + 4 * s4 + ks); bn[s4] = fb(16 * (kb + 1) + 4 * s4 + ks); }
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; s4 += 2) {
                acc = GSMVI_MFMA_F64(a[s4], b[s4], acc);
                acc1 = GSMVI_MFMA_F64(a[s4 + 1], b[s4 + 1], acc1);
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) { a[s4] = an[s4]; b[s4] = bn[s4]; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += acc1[r];
        return acc;
    };
    // jmode 2 (round 5, factor-form BaM in the orthogonal basis [Vw; Zt], gsmvi_bam.hip): M = I + Rt^T J' Rt with the DENSE
    // J' = S'^T diag(I, -I) S', S' = [[I, 0], [Pi, I]].  A' = I + Rg J' Rg^T = I + (Rg S'^T) diag(I, -I) (Rg S'^T)^T, and
    // Rg S'^T differs from Rg in its (1, 2) block only: R12 + R11 Pi^T -- still upper triangular.  So that block is updated in
    // place (Rg is read by the A' phase alone; K is built from W = Rg^-T, which is untouched) and everything below runs as jmode 1.
    if (jmode == 2) {                                  // block-uniform
        // Pi (B x B <= 8 KB) through E2 first (the raw Gram matrix there is dead, A' is not written yet): one coalesced pass
        // instead of B dependent-address global loads per entry (that form cost the rider ~8 us)
        for (int e = tid; e < B * B; e += 512) E2[(e / B) * ES2 + (e % B)] = Pi[e];
        __syncthreads();
        double upd[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {                  // B <= 32: at most 1024 entries, two per thread
            const int e = tid + 512 * u, i = e / B, j = e - i * B;
            double a0 = 0.0, a1 = 0.0;
            if (e < B * B) {
                int k = i;
                for (; k + 1 < B; k += 2) {
                    a0 += E1[i * ES1 + k] * E2[j * ES2 + k];
                    a1 += E1[i * ES1 + k + 1] * E2[j * ES2 + k + 1];
                }
                if (k < B) a0 += E1[i * ES1 + k] * E2[j * ES2 + k];
            }
            upd[u] = a0 + a1;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 512 * u, i = e / B, j = e - i * B;
            if (e < B * B) E1[i * ES1 + B + j] += upd[u];     // (reads above touch columns < B only, writes columns >= B)
        }
        __syncthreads();
    }
    // W <- W S (column operations on the right half of E1; the A' phase below reads only the left half, so both share this
    // barrier interval): K'' = S^T K S = (W S)^T (T - I) (W S)
    for (int e = tid; e < 64 * 32; e += 512) {
        const int r = e >> 5, b = e & 31;
        if (b < B) {
            const double wz = E1[r * ES1 + 64 + b], wu = E1[r * ES1 + 64 + B + b];
            E1[r * ES1 + 64 + b] = wz + s_beta[b] * wu;
            E1[r * ES1 + 64 + B + b] = s_alpha[b] * wu;
        }
    }
    {   // A' = I + (Rg J) Rg^T into E2;  (Rg J)[i][k] = (1/B)(k < B ? Rg[i][B+k] : Rg[i][k-B] - Rg[i][k]), on the MFMA pipe.
        // A' is symmetric (the mirror is written too); Rg is upper triangular, so Rg[j][k] = 0 for k < 16 j: only the
        // k-blocks jb..3 contribute to the block (ib, jb), ib <= jb
        const double invB = jmode ? 1.0 : 1.0 / (double)B;
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
            const int task = TASK_A[w][tq];
            if (task == 0) continue;                   // wave-uniform
            const int ib = (task - 1) >> 2, jb = (task - 1) & 3;
            if (jb >= nblk) {                          // beyond n: the identity padding
                continue;
            }
            const double* arow = E1 + (16 * ib + cc) * ES1;
            const double* brow = E1 + (16 * jb + cc) * ES1;
            const v4d acc = block_chain(
                jb,
                [&](int k) {
                    const int k1 = (k < B) ? B + k : k - B;
                    const double a1 = arow[k1 < 64 ? k1 : 63], a0 = arow[k];
                    if (jmode) return (k < n) ? ((k < B) ? a0 : -a0) : 0.0;      // J = diag(I, -I)
                    return (k < n) ? ((k < B) ? a1 : a1 - a0) : 0.0;
                },
                [&](int k) { return brow[k]; });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ib + ks + 4 * r, j = 16 * jb + cc;
                const double v = (i == j ? 1.0 : 0.0) + ((i < n && j < n) ? acc[r] * invB : 0.0);
                E2[i * ES2 + j] = v;
                if (ib != jb) E2[j * ES2 + i] = v;
            }
        }
        // blocks beyond n: identity padding (written by everybody's share)
        for (int e = tid; e < 64 * 64; e += 512) {
            const int i = e >> 6, j = e & 63;
            if ((i >> 4) >= nblk || (j >> 4) >= nblk) E2[i * ES2 + j] = (i == j) ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    SMALL_STAMP(3);
    chol64_blk<ES2, false, false>(E2, scr, n, &fail_t);             // E2 = T (upper): exists iff M is positive definite
    SMALL_STAMP(4);
    const int bad = (fail_g != 0) || (fail_t != 0) || (prior_bad && *prior_bad != 0);   // prior: BaM's own chain failed
    if (tid == 0) *bad_out = bad;
    if (bad) return;                                   // block-uniform
    {
        // P[i][j] = sum_k (T - I)[i][k] W[k][j]: T - I upper (k >= i), W lower (k >= j): k-blocks max(i, j)..3.  P goes into the
        // left half of E1 (Rg is dead; it was last read in the A' phase, two barriers ago).  The blocks of E2 below the
        // diagonal still hold A' and are never read (k-block >= ib).
        const double* Wm = E1 + 64;
        v4d pacc[3];
#pragma unroll
        for (int tq = 0; tq < 3; ++tq) {
            pacc[tq] = (v4d){0.0, 0.0, 0.0, 0.0};
            const int task = TASK_P[w][tq];
            if (task == 0) continue;
            const int ib = (task - 1) >> 2, jb = (task - 1) & 3;
            if (ib >= nblk || jb >= nblk) continue;
            const int i = 16 * ib + cc;
            pacc[tq] = block_chain(
                ib > jb ? ib : jb, [&](int k) { return E2[i * ES2 + k] - (i == k ? 1.0 : 0.0); },
                [&](int k) { return Wm[k * ES1 + 16 * jb + cc]; });
        }
#pragma unroll
        for (int tq = 0; tq < 3; ++tq) {
            const int task = TASK_P[w][tq];
            if (task == 0) continue;
            const int ib = (task - 1) >> 2, jb = (task - 1) & 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) E1[(16 * ib + ks + 4 * r) * ES1 + 16 * jb + cc] = pacc[tq][r];
        }
        __syncthreads();
        SMALL_STAMP(5);
        // K''[i][j] = sum_k W[k][i] P[k][j]: W[k][i] = 0 for k < i: k-blocks i..3
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
            const int task = TASK_K[w][tq];
            const int ib = (task - 1) >> 2, jb = (task - 1) & 3;
            if (ib >= nblk || jb >= nblk) continue;
            const v4d acc = block_chain(
                ib, [&](int k) { return Wm[k * ES1 + 16 * ib + cc]; }, [&](int k) { return E1[k * ES1 + 16 * jb + cc]; });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ib + ks + 4 * r, j = 16 * jb + cc;
                if (i < n && j < n) Kmat[(size_t)i * n + j] = acc[r];
            }
        }
    }
    SMALL_STAMP(6);
#undef SMALL_STAMP
}

