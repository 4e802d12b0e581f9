// BaM (batch-and-match) update for gfx950, fp64.   Reference: gsmvi/bam.py:72-114 (low-rank form),
// which equals gsmvi/bam.py:31-69 (full form) to round-off.
//
// With an exact factor U = Q Q^T of n = B columns (replacing ARPACK svds, bam.py:10-13; the B centred
// rows sqrt(reg/B)(g_b - gbar) span B - 1 dimensions, so an orthonormal -- Helmert -- recombination gives B - 1 rows with the
// same Gram sum; until round 3 the B + 1 rows themselves were used):
//   Qt (n x D) rows  sqrt(reg/B) helmert_k(g), k < B;  sqrt(reg/(1+reg)) gbar            (bam.py:55-59)
//   Vf (n x D) rows  sqrt(reg/B) helmert_k(x), k < B;  sqrt(reg/(1+reg)) (mu0 - xbar)   V = S0 + Vf^T Vf (bam.py:50-53,60)
//   P  = Qt S0                      one pass over S0, fp64-MFMA panel product
//   M1 = Vf Q,  N0 = P Q            (n x n) Gram matrices over D;  A^T = P + M1^T Vf        (bam.py:107)
//   N  = A^T Q = N0 + M1^T M1;  BB = ((N + I/4)^(1/2) + I/2)^2 = N + I/2 + (N + I/4)^(1/2)  (bam.py:108-109)
//   BB = L L^T;  Z = L^-1 A^T (n x D)  =>  A BB^-1 A^T = Z^T Z                              (bam.py:110)
//   S  = V - Z^T Z = S0 + Vf^T Vf - Z^T Z     rank-2n symmetric update, fp64 MFMA           (bam.py:111)
//   mu = mu0/(1+reg) + reg/(1+reg) (S gbar + xbar)                                         (bam.py:112)
// The n x n matrix square root and the n x n Cholesky run on the DEVICE for n <= 129 (gsmvi_bam_small.hip: scaled
// coupled Newton-Schulz iteration + one-workgroup Cholesky; the reference does this step as a host callback,
// jax.pure_callback, bam.py:15-22).  Only the square-root term goes through the iteration
// (N itself enters BB exactly) and BB^-1 is applied through its Cholesky factor L -- by triangular substitution for n > 128, as
// two products with the explicit W = L^-1 (cond(L) = sqrt(cond(BB)) <= ~1e4 on BASELINE config 4) for n <= 128 since round 4 / 5.
// BB^-1 ITSELF is never formed: with cond(N) ~ 1e7 that form loses 3 digits.
// Everything of size D runs in HIP kernels; S0 is read twice and S written once.
#include <cmath>

#include "gsmvi_common.h"
#include "gsmvi_ctx.h"
#include "gsmvi_chol64.h"      // readlane_f64
#include "../../include/gsmvi_hip.h"

// ---- column means and the Helmert factor panels (both BaM forms) ------------------------------------------------------
// Qt (B x D): rows k-1 = sqrt(reg/B) helmert_k(g), k = 1 .. B-1, row B-1 = sqrt(reg/(1+reg)) gbar; Vout the same for the
// source V (the samples X with shift = mu0 in the dense form: last row sqrt(r1)(mu0 - xbar); the whitened draws Z with
// shift = NULL in the factor form: last row -sqrt(r1) zbar), where helmert_k(v) = (sum_{j<k} c_j - k c_k)/sqrt(k(k+1)), c = v - mean.
// Workgroup = 256 / NG columns x NG sample groups (NG = 4 or 16); group g owns the rows k in [k0, k1) (and the samples of that range: two passes
// over them, the second served by L2), its starting prefix sum comes from the other groups' partial sums.  Any B >= 1.
// XT (factor form): a third track -- the same Helmert rows of the SAMPLES X go to Xh (last row -sqrt(r1)(xbar - xshift)).  With
// x_b = mu0 + z_b F0 they are Vw F0, the first n rows of Rt F0: the update's MFMA-bound product then takes the n + 1 rows
// [Zt; r1 h] only (what the GSM factor update does with its records [x - mu0 | v | v F0]).
template <int NG, bool XT = false>   // sample groups per workgroup: 256 / NG columns x NG groups (4 for B <= 32, 16 above: round 4)
__global__ __launch_bounds__(256) void k_bam_stats_h(int D, int B, const double* __restrict__ V, int ldv,
                                                     const double* __restrict__ shift, const double* __restrict__ X,
                                                     int ldx, const double* __restrict__ G, int ldg, bam_reg regs,
                                                     double* __restrict__ xbar, double* __restrict__ gbar,
                                                     double* __restrict__ zerov, double* __restrict__ Qt,
                                                     double* __restrict__ Vout, double* __restrict__ Vout2,
                                                     const double* __restrict__ xshift, double* __restrict__ Xh) {
    const double reg = regs.get();
    constexpr int NC = 256 / NG;
    __shared__ double red[3][NG][NC];
    const int c = threadIdx.x % NC, g = threadIdx.x / NC;
    const int i = blockIdx.x * NC + c, ic = i < D ? i : D - 1;
    const int k0 = 1 + (g * (B - 1)) / NG, k1 = 1 + ((g + 1) * (B - 1)) / NG;   // rows k in [k0, k1)
    const int s0 = g == 0 ? 0 : k0;                                            // samples [s0, k1) are summed by this group
    double sv = 0.0, sg = 0.0, sx = 0.0;
    {
        int b = s0;
        for (; b + 7 < k1; b += 8) {
            double tv[8], tg[8], tx[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                tv[u] = V[(size_t)(b + u) * ldv + ic];
                tg[u] = G[(size_t)(b + u) * ldg + ic];
                tx[u] = X ? X[(size_t)(b + u) * ldx + ic] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { sv += tv[u]; sg += tg[u]; sx += tx[u]; }
        }
        for (; b < k1; ++b) {
            sv += V[(size_t)b * ldv + ic];
            sg += G[(size_t)b * ldg + ic];
            if (X) sx += X[(size_t)b * ldx + ic];
        }
    }
    red[0][g][c] = sv;
    red[1][g][c] = sg;
    red[2][g][c] = sx;
    __syncthreads();
    double tv_ = 0.0, tg_ = 0.0, tx_ = 0.0;                  // fixed order over the groups
#pragma unroll
    for (int q = 0; q < NG; ++q) { tv_ += red[0][q][c]; tg_ += red[1][q][c]; tx_ += red[2][q][c]; }
    const double vb = tv_ / B, gb = tg_ / B;
    const double a = sqrt(reg / B), r1s = sqrt(reg / (1.0 + reg));
    __syncthreads();                                         // the raw sums have been read by everybody: red is reused below
    if (g == 0 && i < D) {
        const double xb = X ? tx_ / B : vb;
        xbar[i] = xb;
        gbar[i] = gb;
        if (zerov) zerov[i] = 0.0;
        const double vl = -r1s * (vb - (shift ? shift[i] : 0.0));
        Qt[(size_t)(B - 1) * D + i] = r1s * gb;
        Vout[(size_t)(B - 1) * D + i] = vl;
        if (Vout2) Vout2[(size_t)(B - 1) * D + i] = vl;
        if (XT) Xh[(size_t)(B - 1) * D + i] = -r1s * (xb - xshift[i]);
    }
    const double xbm = XT ? tx_ / B : 0.0;
    // prefix of the CENTRED values in front of this group's samples: a second pass over the own samples (centred partial sums:
    // sum(raw) - k mean would cancel badly when the mean is large beside the spread), then the groups in front are added up
    {
        double cv = 0.0, cg = 0.0, cx = 0.0;
        int b = s0;
        for (; b + 7 < k1; b += 8) {
            double tv[8], tg[8], tx[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                tv[u] = V[(size_t)(b + u) * ldv + ic];
                tg[u] = G[(size_t)(b + u) * ldg + ic];
                if (XT) tx[u] = X[(size_t)(b + u) * ldx + ic];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                cv += tv[u] - vb;
                cg += tg[u] - gb;
                if (XT) cx += tx[u] - xbm;
            }
        }
        for (; b < k1; ++b) {
            cv += V[(size_t)b * ldv + ic] - vb;
            cg += G[(size_t)b * ldg + ic] - gb;
            if (XT) cx += X[(size_t)b * ldx + ic] - xbm;
        }
        red[0][g][c] = cv;
        red[1][g][c] = cg;
        if (XT) red[2][g][c] = cx;
    }
    __syncthreads();
    double pv = 0.0, pg = 0.0, px = 0.0;
    for (int q = 0; q < g; ++q) {
        pv += red[0][q][c];
        pg += red[1][q][c];
        if (XT) px += red[2][q][c];
    }
    int k = s0;
    for (; k + 7 < k1; k += 8) {
        double tv[8], tg[8], tx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            tv[u] = V[(size_t)(k + u) * ldv + ic];
            tg[u] = G[(size_t)(k + u) * ldg + ic];
            if (XT) tx[u] = X[(size_t)(k + u) * ldx + ic];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = k + u;
            const double cv = tv[u] - vb, cg = tg[u] - gb, cx = XT ? tx[u] - xbm : 0.0;
            if (kk >= 1 && i < D) {
                const double sc = a / sqrt((double)kk * (double)(kk + 1));
                const double ov = sc * (pv - kk * cv);
                Qt[(size_t)(kk - 1) * D + i] = sc * (pg - kk * cg);
                Vout[(size_t)(kk - 1) * D + i] = ov;
                if (Vout2) Vout2[(size_t)(kk - 1) * D + i] = ov;
                if (XT) Xh[(size_t)(kk - 1) * D + i] = sc * (px - kk * cx);
            }
            pv += cv;
            pg += cg;
            if (XT) px += cx;
        }
    }
    for (; k < k1; ++k) {
        const double cv = V[(size_t)k * ldv + ic] - vb, cg = G[(size_t)k * ldg + ic] - gb;
        const double cx = XT ? X[(size_t)k * ldx + ic] - xbm : 0.0;
        if (k >= 1 && i < D) {
            const double sc = a / sqrt((double)k * (double)(k + 1));
            const double ov = sc * (pv - k * cv);
            Qt[(size_t)(k - 1) * D + i] = sc * (pg - k * cg);
            Vout[(size_t)(k - 1) * D + i] = ov;
            if (Vout2) Vout2[(size_t)(k - 1) * D + i] = ov;
            if (XT) Xh[(size_t)(k - 1) * D + i] = sc * (px - k * cx);
        }
        pv += cv;
        pg += cg;
        if (XT) px += cx;
    }
}

// launch: 4 sample groups x 64 columns for small batches (the round-3 shape), 16 x 16 above (B = 128: 16 workgroups of four
// 32-sample groups became 64 workgroups of sixteen 8-sample groups: every group's loads in one batch)
static inline void bam_stats_launch(hipStream_t st, int D, int B, const double* V, int ldv, const double* shift, const double* X,
                                    int ldx, const double* G, int ldg, bam_reg reg, double* xbar, double* gbar, double* zerov,
                                    double* Qt, double* Vout, double* Vout2, const double* xshift = nullptr, double* Xh = nullptr) {
#define STATS(NGV, XTV, GX) hipLaunchKernelGGL((k_bam_stats_h<NGV, XTV>), dim3(GX), dim3(256), 0, st, D, B, V, ldv, shift, X, ldx, G, ldg, reg, xbar, gbar, zerov, Qt, Vout, Vout2, xshift, Xh)
    if (B <= 32) { if (Xh) STATS(4, true, (D + 63) / 64); else STATS(4, false, (D + 63) / 64); }
    else { if (Xh) STATS(16, true, (D + 15) / 16); else STATS(16, false, (D + 15) / 16); }
#undef STATS
}

// ---- Z = L^-1 (P + M1^T Vf), the new mean, and the signed factor panel -------------------------
// One column of D per thread, COLS threads per block (64 for n <= 160, 32 for n <= 320, 16 for n <= 640, 8 for n <= 1024: the LDS of a CU
// bounds n COLS); the thread's Vf column and running Z column live in LDS ([k][COLS], conflict-free).  M1, L are read with wave-uniform indices (scalar loads, L2 hits).
// Ft = [Vf; Z], Fs = [Vf; -Z]  (rows n..2n-1 written here; rows 0..n-1 by k_bam_stats).
template <int COLS>
__global__ __launch_bounds__(COLS) void k_bam_forward(int D, int n, const double* __restrict__ P,
                                                    const double* __restrict__ M1, const double* __restrict__ L,
                                                    const double* __restrict__ Ldinv,
                                                    const double* __restrict__ zg, const double* __restrict__ vg,
                                                    const double* __restrict__ mu0,
                                                    const double* __restrict__ xbar, bam_reg regs,
                                                    double* __restrict__ Ft, double* __restrict__ Fs,
                                                    double* __restrict__ mu) {
    const double reg = regs.get();
    extern __shared__ double sm[];                 // vf[n][COLS], z[n][COLS]
    double* vf = sm;
    double* z = sm + (size_t)n * COLS;
    const int t = threadIdx.x;
    const int i = blockIdx.x * COLS + t;
    const int ic = i < D ? i : D - 1;
    for (int k = 0; k < n; ++k) vf[k * COLS + t] = Ft[(size_t)k * D + ic];
    double dot_v = 0.0, dot_z = 0.0;
    for (int r = 0; r < n; ++r) {
        double a = P[(size_t)r * D + ic];
        for (int k = 0; k < n; ++k) a += M1[(size_t)k * n + r] * vf[k * COLS + t];       // (M1^T vf)_r
        for (int k = 0; k < r; ++k) a -= L[(size_t)r * n + k] * z[k * COLS + t];
        const double zr = a * Ldinv[r];
        z[r * COLS + t] = zr;
        dot_z += zr * zg[r];
        dot_v += vf[r * COLS + t] * vg[r];
        if (i < D) {
            Ft[(size_t)(n + r) * D + i] = zr;
            Fs[(size_t)(n + r) * D + i] = -zr;
        }
    }
    if (i < D) {
        const double r1 = reg / (1.0 + reg);
        const double s0g = P[(size_t)(n - 1) * D + i] / sqrt(r1);        // (S0 gbar)_i = P[n-1][i]/sqrt(r1)
        mu[i] = mu0[i] / (1.0 + reg) + r1 * (s0g + dot_v - dot_z + xbar[i]);
    }
}

// ---- n > 128 (round 6): the same Z as a BLOCKED forward substitution on the MFMA pipe --------------------------------------
// k_bam_forward above runs one column of D per thread through an n-step substitution with O(n^2) scalar loads: 0.8 ms at
// n = 130, 3.4 ms at n = 256, 14 ms at n = 512 (D = 1024) -- 63 .. 90 % of an update whose other launches take ~0.3 ms.  Here,
// with L = Rb^T from the blocked Cholesky and its diagonal blocks' inverses W_rr = L_rr^-1 (what k_potrf_dag's chain produces
// anyway: W_k = R_kk^-T), block row r of 64 rows is
//     Z_r = W_rr ( P_r + (M1^T)_r Vf - sum_{q < r} L_rq Z_q )
// one launch per block row, a workgroup per 16 columns of D: the inner dimension n + 64 r is split over the eight waves
// (operands straight from L2: for a fixed k both M1[k][64 r + i] and Rb[k][64 r + i] are contiguous in i), the partial tiles are
// summed in fixed order through LDS, P_r is added, and the 64 x 64 by 64 x 16 product with W_rr (staged in LDS) finishes the block.
__global__ __launch_bounds__(512) void k_bam_fwd_block(int D, int n, int r, const double* __restrict__ P,
                                                       const double* __restrict__ M1, const double* __restrict__ Rb,
                                                       const double* __restrict__ Wblk, double* Ft, double* __restrict__ Fs) {
    constexpr int RS = 66;
    __shared__ double red[8 * 64 * 17];
    __shared__ double Ws[64 * RS];
    __shared__ double Ts[64 * 17];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int j0 = blockIdx.x * 16, jc = (j0 + c < D) ? j0 + c : D - 1;
    const int r0 = 64 * r, K = n + r0, ksteps = (K + 3) >> 2;
    double wv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) wv[q] = Wblk[(size_t)r * 4096 + tid + 512 * q];
    v4d acc[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[rt] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int s0 = w; s0 < ksteps; s0 += 32) {                    // four k-steps of this wave per trip: 20 loads in flight
        double a[4][4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 4 * (s0 + 8 * u) + ks;
            const bool live = k < K, low = k < n;
            const int kk = live ? (low ? k : k - n) : 0;
            b[u] = Ft[(size_t)(low ? kk : n + kk) * D + jc];
            const double* arow = (low ? M1 : Rb) + (size_t)kk * n + r0;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const int i = 16 * rt + c;
                const double v = arow[(r0 + i < n) ? i : 0];
                a[u][rt] = (live && r0 + i < n) ? (low ? v : -v) : 0.0;
            }
            if (!live) b[u] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt] = GSMVI_MFMA_F64(a[u][rt], b[u], acc[rt]);
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[(w * 64 + 16 * rt + ks + 4 * q) * 17 + c] = acc[rt][q];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + 512 * q;
        Ws[(e >> 6) * RS + (e & 63)] = wv[q];
    }
    __syncthreads();
    for (int e = tid; e < 64 * 16; e += 512) {
        const int row = e >> 4, col = e & 15;
        double t = 0.0;
#pragma unroll
        for (int ww = 0; ww < 8; ww += 2) t += red[(ww * 64 + row) * 17 + col] + red[((ww + 1) * 64 + row) * 17 + col];
        const int gc = (j0 + col < D) ? j0 + col : D - 1;
        Ts[row * 17 + col] = (r0 + row < n) ? t + P[(size_t)(r0 + row) * D + gc] : 0.0;
    }
    __syncthreads();
    if (w < 4) {
        v4d z = {0.0, 0.0, 0.0, 0.0};
        const double* ap = Ws + (16 * w + c) * RS + ks;
#pragma unroll
        for (int s = 0; s < 16; ++s) z = GSMVI_MFMA_F64(ap[4 * s], Ts[(4 * s + ks) * 17 + c], z);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = r0 + 16 * w + ks + 4 * q;
            if (row < n && j0 + c < D) {
                Ft[(size_t)(n + row) * D + j0 + c] = z[q];
                Fs[(size_t)(n + row) * D + j0 + c] = -z[q];
            }
        }
    }
}

// the new mean behind the blocked substitution (bam.py:112): one column of D per thread, the 2n rows of [Vf; Z] streamed
__global__ __launch_bounds__(256) void k_bam_mean_big(int D, int n, const double* __restrict__ P, const double* __restrict__ Ft,
                                                      const double* __restrict__ zg, const double* __restrict__ vg,
                                                      const double* __restrict__ mu0, const double* __restrict__ xbar,
                                                      bam_reg regs, double* __restrict__ mu) {
    const double reg = regs.get();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    double dv0 = 0.0, dv1 = 0.0, dz0 = 0.0, dz1 = 0.0;
    int k = 0;
    for (; k + 4 <= n; k += 4) {
        double v[4], z[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = Ft[(size_t)(k + u) * D + i];
            z[u] = Ft[(size_t)(n + k + u) * D + i];
        }
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
            dv0 += v[u] * vg[k + u];
            dv1 += v[u + 1] * vg[k + u + 1];
            dz0 += z[u] * zg[k + u];
            dz1 += z[u + 1] * zg[k + u + 1];
        }
    }
    for (; k < n; ++k) {
        dv0 += Ft[(size_t)k * D + i] * vg[k];
        dz0 += Ft[(size_t)(n + k) * D + i] * zg[k];
    }
    const double r1 = reg / (1.0 + reg);
    const double s0g = P[(size_t)(n - 1) * D + i] / sqrt(r1);
    mu[i] = mu0[i] / (1.0 + reg) + r1 * (s0g + (dv0 + dv1) - (dz0 + dz1) + xbar[i]);
}

// ---- N = M1^T M1 + sym(N0) and M1^T (n x n), on the device ---------------------------------------
// One 16 x 16 block of N per wave on the MFMA pipe (round 3; it was an n-long scalar loop per element: 20 us at n = 128):
// k runs over the rows of M1 in batches of 64, operands straight from L2 (a wave's 16 lanes read 128 contiguous bytes of a
// row), two accumulator chains.  The transposed block of M1 is written by the same wave.
__global__ __launch_bounds__(256) void k_bam_nmat(int n, const double* __restrict__ M1,
                                                  const double* __restrict__ N0, double* __restrict__ Nm,
                                                  double* __restrict__ M1T) {
    const int nb = (n + 15) >> 4;
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= nb * nb) return;                                  // wave-uniform
    const int i0 = (blk / nb) * 16, j0 = (blk % nb) * 16;
    const int l = threadIdx.x & 63, cc = l & 15, ks = l >> 4;
    const int ic = (i0 + cc) < n ? i0 + cc : n - 1, jc = (j0 + cc) < n ? j0 + cc : n - 1;
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int kb = 0; kb < n; kb += 64) {
        double a[16], b[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k = kb + 4 * s + ks, kc = k < n ? k : n - 1;
            const double av = M1[(size_t)kc * n + ic], bv = M1[(size_t)kc * n + jc];
            a[s] = (k < n && (i0 + cc) < n) ? av : 0.0;
            b[s] = (k < n && (j0 + cc) < n) ? bv : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
            acc0 = GSMVI_MFMA_F64(a[s], b[s], acc0);
            acc1 = GSMVI_MFMA_F64(a[s + 1], b[s + 1], acc1);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ks + 4 * r, j = j0 + cc;
        if (i < n && j < n) {
            Nm[(size_t)i * n + j] = (acc0[r] + acc1[r]) + 0.5 * (N0[(size_t)i * n + j] + N0[(size_t)j * n + i]);
            M1T[(size_t)i * n + j] = M1[(size_t)j * n + i];
        }
    }
}

// ---- the same from the split-K SLABS of the stacked Gram product [N0; M1] (round 4): finish + N in one launch ------------------
// k_panel_finish (4.7 us) + k_bam_nmat (19 us on 16 workgroups at n = 128) were two launches; here one workgroup per 16 x 16
// block of N sums the slabs while it loads its operands (slab element (r, c) of slab q: slabs[q stride + r n + c], rows
// 0 .. n-1 = N0, rows n .. 2n-1 = M1), K = n split over the four waves, partial blocks summed through LDS in a fixed order,
// and writes its blocks of the finished N0 and M1 as well (k_bam_cholw / k_bam_zw read them).  M1^T is not produced: the
// product-form Z (k_bam_zw) reads M1 by columns itself.
// ldp: row length of the slabs (n; 2n when the product also holds [.; Vw] Vw^T -- G11 then receives the finished Vw Vw^T, the
// first diagonal block of the factor-form chain's Gram matrix, which is factored beside k_bam_cholw).
// KCT: compile-time bound of the slab count (4 / 8 / GSMVI_MAX_KC): every entry costs KCT loads, so the bound matters -- with the
// clamp-to-MAX_KC form 304 loads per lane were issued for kc = 8 (10.5 us at n = 128).
template <int KCT>
__global__ __launch_bounds__(256) void k_bam_nmat2(int n, int kc, const double* __restrict__ slabs, long long slab_stride, int ldp,
                                                   double* __restrict__ N0, double* __restrict__ M1,
                                                   double* __restrict__ Nm, double* __restrict__ G11) {
    __shared__ double red[4 * 256];
    const int nb = (n + 15) >> 4;
    const int bi = blockIdx.x / nb, bj = blockIdx.x - bi * nb, i0 = 16 * bi, j0 = 16 * bj;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, cc = l & 15, ks = l >> 4;
    const int ic = (i0 + cc) < n ? i0 + cc : n - 1, jc = (j0 + cc) < n ? j0 + cc : n - 1;
    const int nk = (n + 3) >> 2;
    auto slab_sum = [&](int r, int c) {                      // one finished entry: the kc slabs, all loads in one batch
        double t[KCT];
#pragma unroll
        for (int q = 0; q < KCT; ++q) t[q] = slabs[(size_t)(q < kc ? q : kc - 1) * slab_stride + (size_t)r * ldp + c];
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < KCT; ++q) a += (q < kc) ? t[q] : 0.0;
        return a;
    };
    const int t = threadIdx.x, i = i0 + (t >> 4), j = j0 + (t & 15);
    const int iq = i < n ? i : n - 1, jq = j < n ? j : n - 1;
    const double n0ij = slab_sum(iq, jq), n0ji = slab_sum(jq, iq), m1ij = slab_sum(n + iq, jq);   // (issued first: not a tail)
    const double g11 = G11 ? slab_sum(n + iq, n + jq) : 0.0;
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int u0 = 0; w + 4 * u0 < nk; u0 += 4) {             // four k-steps of this wave per batch (64 slab loads in flight)
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int st = w + 4 * (u0 + u), k = 4 * st + ks, kk = k < n ? k : n - 1;
            const double av = slab_sum(n + kk, ic), bv = slab_sum(n + kk, jc);
            a[u] = (k < n && (i0 + cc) < n) ? av : 0.0;
            b[u] = (k < n && (j0 + cc) < n) ? bv : 0.0;
        }
        acc0 = GSMVI_MFMA_F64(a[0], b[0], acc0);
        acc1 = GSMVI_MFMA_F64(a[1], b[1], acc1);
        acc0 = GSMVI_MFMA_F64(a[2], b[2], acc0);
        acc1 = GSMVI_MFMA_F64(a[3], b[3], acc1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w * 256 + (ks + 4 * r) * 16 + cc] = acc0[r] + acc1[r];
    __syncthreads();
    if (i < n && j < n) {
        const double v = (red[t] + red[256 + t]) + (red[512 + t] + red[768 + t]);
        Nm[(size_t)i * n + j] = v + 0.5 * (n0ij + n0ji);
        N0[(size_t)i * n + j] = n0ij;
        M1[(size_t)i * n + j] = m1ij;
        if (G11) G11[(size_t)i * n + j] = g11;
    }
}

// Orthogonal basis of the factor form with the one-launch chain (n <= 48, gsmvi_bam_factor_impl): what depends on the chain's
// W = L^-1 rides in k_bam_zw's launch.
//   prologue of every workgroup:  vg' = vg - Dm t2   (t2 = W^T W a from k_bam_small48, Dm from its side workgroup; the mean of
//                                 bam.py:112 needs Zw^T zg = Zt^T zg + Vw^T Pi^T zg and the kernel forms Vw^T vg - Z^T zg)
//   one more workgroup (the last): Pi = W Dm^T (n^3 / 2 multiply-adds from LDS), and the flag of Gvv's factorisation joins the
//                                 flag of the chain (a dependent draw reverts the update)
struct bamf_fix {
    const double* Dm;                  // null: none of this (dense form, n > 48, bam_basis = 0)
    const double* t2;
    double* Pi;
    int* info;
    const int* info1;
};
#define BAMB_SN 48
#define BAMB_LS 49
__device__ __forceinline__ void bamf_pi_rider(int n, const bamf_fix& fx, const double* __restrict__ Wt, double* Ws, double* Ms) {
    constexpr int LS = BAMB_LS;
    const int tid = threadIdx.x;
    for (int e = tid; e < n * n; e += 512) {
        const int i = e / n, j = e - i * n;
        Ws[j * LS + i] = Wt[e];                              // Ws[r][k] = W[r][k] = Wt[k][r]
        Ms[i * LS + j] = fx.Dm[e];
    }
    if (tid == 0 && *fx.info == 0 && *fx.info1 != 0) *fx.info = 1000 + *fx.info1;
    __syncthreads();
    for (int e = tid; e < n * n; e += 512) {                 // Pi[i][j] = sum_{k <= i} W[i][k] Dm[j][k]; four partial sums
        const int i = e / n, j = e - i * n;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        int k = 0;
        for (; k + 3 <= i; k += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += Ws[i * LS + k + u] * Ms[j * LS + k + u];
        }
        for (; k <= i; ++k) a[0] += Ws[i * LS + k] * Ms[j * LS + k];
        fx.Pi[e] = (a[0] + a[1]) + (a[2] + a[3]);
    }
}

// ---- Z = W (P + M1^T Vf) with the explicit inverse factor W = L^-1 (k_bam_cholw), the new mean, the signed panel; n <= 128 ----
// (round 4: bam.py:107,110 as two chained MFMA products instead of a panel product + finish + per-column substitution.)
// One workgroup per 16 columns of D, eight waves, wave w owns the 16-row block w of both products:
//   phase 1  T = M1^T Vf_tile (K = n), A = P_tile + T -> LDS           A-operand M1[k][r] straight from L2 (16 consecutive
//            doubles per k), B-operand the Vf tile staged once in LDS ([k][16]: the four k-slots of a step read 512
//            contiguous bytes, conflict-free)
//   phase 2  Z = W A: row block w needs the k-blocks 0 .. w only (W is lower triangular).  W arrives TRANSPOSED (Wt = R^-1,
//            upper): the A-operand W[r][k] = Wt[k][r] is again 16 consecutive doubles per k -- with a row-major W every load
//            instruction touched 16 cache lines and the kernel took 21 us instead of ~9.  Its loads are issued before phase 1.
// zg = W a (bam.py:110 applied to gbar; a from k_bam_bbav) falls out of the same W fragments: four lanes per row, summed by two
// shuffles.  Z goes to rows n .. 2n-1 of Ft and, negated, of Fs (rows 0 .. n-1 hold Vf: k_bam_stats_h).  Mean (bam.py:112):
// mu = mu0/(1+reg) + r1 (S0 gbar + Vf^T vg - Z^T zg + xbar), the two dots from the tiles in LDS / registers, fixed order.
__global__ __launch_bounds__(512) void k_bam_zw(int D, int n, const double* __restrict__ P, const double* __restrict__ M1,
                                                const double* __restrict__ Wt, const double* __restrict__ av,
                                                const double* __restrict__ vg, const double* __restrict__ mu0,
                                                const double* __restrict__ xbar, bam_reg regs, double* __restrict__ Ft,
                                                double* __restrict__ Fs, double* __restrict__ mu, bamf_fix fx) {
    const double reg = regs.get();
    __shared__ __attribute__((aligned(16))) double Vs[128 * 16], As[128 * 16];
    __shared__ double sav[128], szg[128], svg[128], redz[8 * 4 * 16];
    if (fx.Dm && blockIdx.x == (unsigned)((D + 15) / 16)) {      // block-uniform: the workgroup behind the column tiles (n <= 48)
        __shared__ double Prd[2 * BAMB_SN * BAMB_LS];
        bamf_pi_rider(n, fx, Wt, Prd, Prd + BAMB_SN * BAMB_LS);
        return;
    }
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, cc = l & 15, ks = l >> 4;
    const int j0 = blockIdx.x * 16;
    const int nb = (n + 15) >> 4;
    const int jc = (j0 + cc) < D ? j0 + cc : D - 1;
    const bool colin = (j0 + cc) < D;
    const int rA = 16 * w + cc, rAc = rA < n ? rA : n - 1;   // this lane's row as an MFMA A-operand row
    // all global loads of the workgroup first: Vf tile (4 per thread), this wave's M1 and Wt columns, its P block
    double vt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = tid + 512 * u, k = e >> 4, c = e & 15;
        const int kc = k < n ? k : n - 1, jj = (j0 + c) < D ? j0 + c : D - 1;
        const double v = Ft[(size_t)kc * D + jj];
        vt[u] = (k < n && (j0 + c) < D) ? v : 0.0;
    }
    double am[32], aw[32], pv[4];
    if (w < nb) {                                            // wave-uniform
#pragma unroll
        for (int st = 0; st < 32; ++st) {
            const int k = 4 * st + ks, kc = k < n ? k : n - 1;
            const double m = M1[(size_t)kc * n + rAc];
            am[st] = (k < n && rA < n) ? m : 0.0;
        }
#pragma unroll
        for (int st = 0; st < 32; ++st) {
            const int k = 4 * st + ks, kc = k < n ? k : n - 1;
            double x = 0.0;
            if (st < 4 * (w + 1)) x = Wt[(size_t)kc * n + rAc];          // W[rA][k], k <= 16 w + 15
            aw[st] = (k < n && rA < n) ? x : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * w + ks + 4 * r;
            const double x = P[(size_t)(row < n ? row : n - 1) * D + jc];
            pv[r] = (row < n && colin) ? x : 0.0;
        }
    }
    if (tid < 128) {
        sav[tid] = tid < n ? av[tid] : 0.0;
        svg[tid] = tid < n ? vg[tid] : 0.0;
    }
    double vgfix = 0.0;
    if (fx.Dm) {                                             // vg' = vg - Dm t2 (n <= 48: eight lanes per row, six loads each in one
        const int row = tid >> 3, part = tid & 7, rc = row < n ? row : n - 1;   // batch behind every other load of the prologue)
        double dv[6], tv2[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int k = part + 8 * u, kc = k < n ? k : n - 1;
            dv[u] = fx.Dm[(size_t)rc * n + kc];
            tv2[u] = fx.t2[kc];
        }
        double d = 0.0;
#pragma unroll
        for (int u = 0; u < 6; ++u) d += (part + 8 * u < n) ? dv[u] * tv2[u] : 0.0;
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        vgfix = d;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) Vs[tid + 512 * u] = vt[u];
    __syncthreads();
    if (fx.Dm && (tid & 7) == 0 && (tid >> 3) < n) svg[tid >> 3] -= vgfix;   // (read behind two more barriers)
    if (w < nb) {
        {   // zg[rA] = sum_k W[rA][k] a[k]: this lane holds k = 4 st + ks of its row; the row's four lanes are cc, cc + 16, ...
            double z0 = 0.0, z1 = 0.0;
#pragma unroll
            for (int st = 0; st < 32; st += 2) {
                if (st < 4 * (w + 1)) {
                    z0 += aw[st] * sav[4 * st + ks];
                    z1 += aw[st + 1] * sav[4 * st + 4 + ks];
                }
            }
            double z = z0 + z1;
            z += __shfl_xor(z, 16, 64);
            z += __shfl_xor(z, 32, 64);
            if (ks == 0) szg[rA] = z;                        // (rows >= n: zero operands, zero result)
        }
        v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int st = 0; st < 32; st += 2) {
            if (4 * st < n) {                                // block-uniform: k-steps beyond n are skipped
                acc0 = GSMVI_MFMA_F64(am[st], Vs[(4 * st + ks) * 16 + cc], acc0);
                acc1 = GSMVI_MFMA_F64(am[st + 1], Vs[(4 * st + 4 + ks) * 16 + cc], acc1);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) As[(16 * w + ks + 4 * r) * 16 + cc] = pv[r] + (acc0[r] + acc1[r]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) As[(16 * w + ks + 4 * r) * 16 + cc] = 0.0;
        if (ks == 0) szg[rA] = 0.0;
    }
    __syncthreads();
    if (w < nb) {
        v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int st = 0; st < 32; st += 2) {
            if (st < 4 * (w + 1)) {                          // wave-uniform: k-blocks 0 .. w
                acc0 = GSMVI_MFMA_F64(aw[st], As[(4 * st + ks) * 16 + cc], acc0);
                acc1 = GSMVI_MFMA_F64(aw[st + 1], As[(4 * st + 4 + ks) * 16 + cc], acc1);
            }
        }
        double pz = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * w + ks + 4 * r;
            const double z = acc0[r] + acc1[r];
            if (row < n) {
                pz += z * szg[row];
                if (colin) {
                    Ft[(size_t)(n + row) * D + j0 + cc] = z;
                    Fs[(size_t)(n + row) * D + j0 + cc] = -z;
                }
            }
        }
        redz[(w * 4 + ks) * 16 + cc] = pz;
    } else {
        redz[(w * 4 + ks) * 16 + cc] = 0.0;
    }
    __syncthreads();
    {   // Vf^T vg - Z^T zg per column: 32 partials per column (thread = (part, column)), shuffles inside the wave (its four
        // parts), then eight wave partials through LDS -- fixed order; a 16-thread serial loop here cost ~4 us
        const int col = tid & 15, part = tid >> 4;           // part = 4 w + ks
        double d = -redz[part * 16 + col];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = part + 32 * u;
            d += (k < n) ? Vs[k * 16 + col] * svg[k] : 0.0;
        }
        d += __shfl_xor(d, 16, 64);
        d += __shfl_xor(d, 32, 64);
        __syncthreads();                                     // every read of redz is done: it takes the wave partials
        if (ks == 0) redz[w * 16 + col] = d;
    }
    __syncthreads();
    if (tid < 16 && (j0 + tid) < D) {
        double dd = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) dd += redz[q * 16 + tid];
        const int j = j0 + tid;
        const double r1 = reg / (1.0 + reg);
        const double s0g = P[(size_t)(n - 1) * D + j] / sqrt(r1);          // (S0 gbar)_j = P[n-1][j]/sqrt(r1)
        mu[j] = mu0[j] / (1.0 + reg) + r1 * (s0g + dd + xbar[j]);
    }
}

// ---- symmetric low-rank update  S = S0 + Ft^T Fs + jitter I   (Ft, Fs: KF x D row-major) --------
// Ft^T Fs must be symmetric (Fs = K Ft with K symmetric).  One workgroup per 64x64 tile PAIR (I <= J)
// of the upper triangle: W = S0[I,J] + Ft[:,I]^T Fs[:,J] (4 waves of 32x32, fp64 MFMA, factor rows
// staged through LDS in chunks of 64), stores S[I,J] = W and the mirror S[J,I] = W^T through an
// LDS transpose, so S is exactly symmetric (bam.py:199 symmetrises in fit) and S0 is read once.
__global__ __launch_bounds__(256) void k_lowrank_update(int D, int KF, const double* __restrict__ Ft,
                                                        const double* __restrict__ Fs,
                                                        const double* __restrict__ S0, int lds0,
                                                        double* __restrict__ S, int lds, double jitter) {
    constexpr int RS = 66;
    __shared__ double smem[2 * 64 * RS];
    double* FA = smem;
    double* FB = smem + 64 * RS;
    const int nt = (D + 63) >> 6;
    int ti, tj;
    {
        const int idx = blockIdx.x;
        const double q = 2.0 * nt + 1.0;
        int t = (int)((q - sqrt(q * q - 8.0 * (double)idx)) * 0.5);
        if (t < 0) t = 0;
        while (t > 0 && t * nt - (t * (t - 1)) / 2 > idx) --t;
        while ((t + 1) * nt - ((t + 1) * t) / 2 <= idx) ++t;
        ti = t;
        tj = t + (idx - (t * nt - (t * (t - 1)) / 2));
    }
    const int I0 = ti * 64, J0 = tj * 64;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int wr = w >> 1, wc = w & 1;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int kb = 0; kb < KF; kb += 64) {
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
    __syncthreads();                              // factor tiles are dead; smem becomes the 64x65 transpose buffer
    double* LW = smem;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = 32 * wr + 16 * rt + ks + 4 * r, lc = 32 * wc + 16 * ct + c;
                const int row = I0 + lr, col = J0 + lc;
                double v = 0.0;
                if (row < D && col < D) {
                    v = S0[(size_t)row * lds0 + col] + acc[rt][ct][r] + (row == col ? jitter : 0.0);
                    if (ti != tj || col >= row) S[(size_t)row * lds + col] = v;
                }
                LW[lr * 65 + lc] = v;
            }
    __syncthreads();
    // mirror: S[J0 + lr][I0 + lc] = W[lc][lr]; on a diagonal tile pair only the strict lower part
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = 32 * wr + 16 * rt + ks + 4 * r, lc = 32 * wc + 16 * ct + c;
                const int row = J0 + lr, col = I0 + lc;
                if (row < D && col < D && (ti != tj || col < row)) S[(size_t)row * lds + col] = LW[lc * 65 + lr];
            }
}

// ---- fast form of the same update for D % 64 == 0 and KF <= 288 ---------------------------------------------------
// The product Ft^T Fs = Vf^T Vf - Z^T Z is symmetric term by term (row p < n contributes Vf[p][i] Vf[p][j], row
// p >= n contributes -Z[i] Z[j]), so computing EVERY 64 x 64 tile with the same k order gives a bitwise symmetric S
// without the mirror pass -- twice the (small) MFMA work, but 256 workgroups instead of 136 at D = 1024, two waves per
// SIMD, no transposes.  512 threads per tile; wave w owns the 16-row block w >> 1 and two 16-column blocks; every
// global load of the workgroup is issued first (S0 tile in accumulator layout, then all KF rows of both operand
// tiles, rows beyond KF as zeros); operands pass through LDS 32 rows at a time ([k][80], conflict-free for both MFMA
// operands).  Same structure as k_gsmf_update_fast in the factor path.
// Round 5: (i) the workgroup barriers wait for LDS only (s_waitcnt lgkmcnt(0); s_barrier).  __syncthreads() carries a
// workgroup fence that drains the vector-memory counter, and ALL operand loads of the tile are issued up front: the first
// barrier therefore waited for the last row of the last pass, and the load phase (256 KB per CU at KF = 256) and the
// compute phase ran strictly one after the other.  No thread reads global data written by another thread of the workgroup, so
// only LDS needs ordering (the rule of k_gsm_cov_sym_p).  (ii) KP rows per staging pass as a template
// parameter: 64 for KF > 96 (BASELINE config 4: 256 rows in 4 passes instead of 8 -- half the barriers).
template <int NPMAX, int KP, bool RAG>    // RAG: D % 64 != 0 (edge tiles; round 5)
__global__ __launch_bounds__(512) void k_lowrank_update_fast(int D, int KF, const double* __restrict__ Ft,
                                                             const double* __restrict__ Fs,
                                                             const double* __restrict__ S0, int lds0,
                                                             double* __restrict__ S, int lds, double jitter) {
#define LR_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    constexpr int RS = 80;
    constexpr int UQ = KP * 32 / 512;              // 16-byte units per thread, operand and pass (KP = 32: 2, KP = 64: 4)
    __shared__ __attribute__((aligned(16))) double sm[2 * KP * RS];
    const int nt = (D + 63) >> 6, np = (KF + KP - 1) / KP;    // any even D (round 5): edge tiles re-read clamped rows / columns
    const int ti = blockIdx.x / nt, tj = blockIdx.x % nt;
    const int I0 = ti * 64, J0 = tj * 64;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int wr = w >> 1, wc = w & 1;
    double s0[2][4];
    const int frow = I0 + 16 * wr + ks;
    const int fcol = J0 + 32 * wc + c;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = (RAG && frow + 4 * r >= D) ? D - 1 : frow + 4 * r, cq = (RAG && fcol + 16 * blk >= D) ? D - 1 : fcol + 16 * blk;
            s0[blk][r] = S0[(size_t)rr * lds0 + cq];
        }
    v2d ga[NPMAX][UQ], gb[NPMAX][UQ];
#pragma unroll
    for (int p = 0; p < NPMAX; ++p) {
        if (p < np) {                                        // block-uniform
#pragma unroll
            for (int q = 0; q < UQ; ++q) {
                const int u = q * 512 + tid, row = KP * p + (u >> 5), c2 = 2 * (u & 31);
                const int rc = row < KF ? row : KF - 1;
                ga[p][q] = *reinterpret_cast<const v2d*>(Ft + (size_t)rc * D + ((RAG && I0 + c2 >= D) ? D - 2 : I0 + c2));
                gb[p][q] = *reinterpret_cast<const v2d*>(Fs + (size_t)rc * D + ((RAG && J0 + c2 >= D) ? D - 2 : J0 + c2));
            }
        }
    }
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < NPMAX; ++p) {
        if (p < np) {
            if (p > 0) LR_LDS_BARRIER();                     // the previous pass's operand reads are done
#pragma unroll
            for (int q = 0; q < UQ; ++q) {
                const int u = q * 512 + tid, row = u >> 5, c2 = 2 * (u & 31);
                const bool in = KP * p + row < KF;           // (rows beyond KF: clamped re-reads, staged as zeros)
                *reinterpret_cast<v2d*>(sm + row * RS + c2) = in ? ga[p][q] : (v2d){0.0, 0.0};
                *reinterpret_cast<v2d*>(sm + (KP + row) * RS + c2) = in ? gb[p][q] : (v2d){0.0, 0.0};
            }
            LR_LDS_BARRIER();
            const double* ap = sm + ks * RS + 16 * wr + c;
            const double* bp = sm + (KP + ks) * RS + 32 * wc + c;
#pragma unroll
            for (int h = 0; h < KP / 32; ++h) {              // 8 MFMA steps (32 rows) at a time: operands to registers first
                double a[8], b0[8], b1[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    a[s] = ap[(32 * h + 4 * s) * RS];
                    b0[s] = bp[(32 * h + 4 * s) * RS];
                    b1[s] = bp[(32 * h + 4 * s) * RS + 16];
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    acc0 = GSMVI_MFMA_F64(a[s], b0[s], acc0);
                    acc1 = GSMVI_MFMA_F64(a[s], b1[s], acc1);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = frow + 4 * r;
        if (!RAG || (row < D && fcol < D)) S[(size_t)row * lds + fcol] = s0[0][r] + acc0[r] + (row == fcol ? jitter : 0.0);
        if (!RAG || (row < D && fcol + 16 < D)) S[(size_t)row * lds + fcol + 16] = s0[1][r] + acc1[r] + (row == fcol + 16 ? jitter : 0.0);
    }
#undef LR_LDS_BARRIER
}

// The same update for ANY number of factor rows (round 6; KF > 288, i.e. BaM batches beyond 144: until then the guarded kernel
// k_lowrank_update, 74 us at KF = 512): a run-time loop of 64-row passes instead of NPMAX compile-time passes over registers loaded
// up front -- the loads of pass p + 1 go out as soon as pass p has been staged (one register set: it is consumed by the staging),
// LDS-only barriers.  Same tiles, same k order, same stores as k_lowrank_update_fast.
template <bool RAG>
__global__ __launch_bounds__(512) void k_lowrank_update_big(int D, int KF, const double* __restrict__ Ft,
                                                            const double* __restrict__ Fs,
                                                            const double* __restrict__ S0, int lds0,
                                                            double* __restrict__ S, int lds, double jitter) {
#define LR_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    constexpr int RS = 80, KP = 64;
    constexpr int UQ = KP * 32 / 512;
    __shared__ __attribute__((aligned(16))) double sm[2 * KP * RS];
    const int nt = (D + 63) >> 6, np = (KF + KP - 1) / KP;
    const int ti = blockIdx.x / nt, tj = blockIdx.x % nt;
    const int I0 = ti * 64, J0 = tj * 64;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int wr = w >> 1, wc = w & 1;
    double s0[2][4];
    const int frow = I0 + 16 * wr + ks;
    const int fcol = J0 + 32 * wc + c;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = (RAG && frow + 4 * r >= D) ? D - 1 : frow + 4 * r, cq = (RAG && fcol + 16 * blk >= D) ? D - 1 : fcol + 16 * blk;
            s0[blk][r] = S0[(size_t)rr * lds0 + cq];
        }
    v2d ga[UQ], gb[UQ];
    auto issue = [&](int p) {
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            const int u = q * 512 + tid, row = KP * p + (u >> 5), c2 = 2 * (u & 31);
            const int rc = row < KF ? row : KF - 1;
            ga[q] = *reinterpret_cast<const v2d*>(Ft + (size_t)rc * D + ((RAG && I0 + c2 >= D) ? D - 2 : I0 + c2));
            gb[q] = *reinterpret_cast<const v2d*>(Fs + (size_t)rc * D + ((RAG && J0 + c2 >= D) ? D - 2 : J0 + c2));
        }
    };
    issue(0);
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int p = 0; p < np; ++p) {
        if (p > 0) LR_LDS_BARRIER();
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            const int u = q * 512 + tid, row = u >> 5, c2 = 2 * (u & 31);
            const bool in = KP * p + row < KF;
            *reinterpret_cast<v2d*>(sm + row * RS + c2) = in ? ga[q] : (v2d){0.0, 0.0};
            *reinterpret_cast<v2d*>(sm + (KP + row) * RS + c2) = in ? gb[q] : (v2d){0.0, 0.0};
        }
        LR_LDS_BARRIER();
        if (p + 1 < np) issue(p + 1);
        const double* ap = sm + ks * RS + 16 * wr + c;
        const double* bp = sm + (KP + ks) * RS + 32 * wc + c;
#pragma unroll
        for (int h = 0; h < KP / 32; ++h) {
            double a[8], b0[8], b1[8];
#pragma unroll
            for (int sI = 0; sI < 8; ++sI) {
                a[sI] = ap[(32 * h + 4 * sI) * RS];
                b0[sI] = bp[(32 * h + 4 * sI) * RS];
                b1[sI] = bp[(32 * h + 4 * sI) * RS + 16];
            }
#pragma unroll
            for (int sI = 0; sI < 8; ++sI) {
                acc0 = GSMVI_MFMA_F64(a[sI], b0[sI], acc0);
                acc1 = GSMVI_MFMA_F64(a[sI], b1[sI], acc1);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = frow + 4 * r;
        if (!RAG || (row < D && fcol < D)) S[(size_t)row * lds + fcol] = s0[0][r] + acc0[r] + (row == fcol ? jitter : 0.0);
        if (!RAG || (row < D && fcol + 16 < D)) S[(size_t)row * lds + fcol + 16] = s0[1][r] + acc1[r] + (row == fcol + 16 ? jitter : 0.0);
    }
#undef LR_LDS_BARRIER
}

#define HIPCHK(expr)                                                          \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) {                                               \
            gsmvi_set_error("%s failed: %s", #expr, hipGetErrorString(e_));   \
            return GSMVI_ERR_HIP;                                             \
        }                                                                     \
    } while (0)

#include "gsmvi_chol128.h"   // cholw_job
#include "gsmvi_smallgemm.h"
int gsmvi_bam_small_device(gsmvi_ctx* ctx, hipStream_t st, int n, bam_reg reg, const double* Nd, const double* M1,
                           const double* N0, double* scratch, double* Ld, int* info_dev, int* hint_host, int force_kenq,
                           double* Rscr, const cholw_job* beside, const bamq_side* side64, const double* G11);
int gsmvi_bam_small_one_wg(const gsmvi_ctx* ctx, int n);
int gsmvi_panel_t_product_few_slabs(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* A, int lda, const double* M,
                                    int ldm, int mrows, double* Pp, int* kc_out);
int gsmvi_bam_small_nmax();
size_t gsmvi_bam_small_scratch_doubles(int n);
int gsmvi_panel_t_product(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* A, int lda, const double* M,
                          int ldm, int mrows, double* Pp, int* kc_out);
int gsmvi_bam_small_fused_nmax();
int gsmvi_bam_small_fused(gsmvi_ctx* ctx, hipStream_t st, int n, bam_reg reg, const double* slabs, int kc, int ldslab,
                          size_t slab_stride, double* M1, double* Ld, int* info_dev, const bamq_side* side);

#define BAM_NMAT2(KC, NBQ, N_, SL, STR, LDP, N0_, M1_, ND_, G11_)                                                        \
    do {                                                                                                                \
        if ((KC) <= 4)                                                                                                  \
            hipLaunchKernelGGL(k_bam_nmat2<4>, dim3((NBQ) * (NBQ)), dim3(256), 0, st, N_, KC, SL, STR, LDP, N0_, M1_, ND_, G11_); \
        else                                                                                                            \
            hipLaunchKernelGGL(k_bam_nmat2<GSMVI_MAX_KC>, dim3((NBQ) * (NBQ)), dim3(256), 0, st, N_, KC, SL, STR, LDP, N0_, M1_, ND_, \
                               G11_);                                                                                   \
    } while (0)
int gsmvi_bam_impl(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* X, int ldx, const double* G,
                   int ldg, const double* mu0, const double* S0, int lds0, double reg_value, double jitter, double* mu,
                   double* S, int lds, int* info_dev) {
    const bam_reg reg{reg_value, ctx->reg_dev};
    // n = B columns of Q and rows of Vf (round 3: the Helmert recombination of the B centred rows, k_bam_stats_h; the
    // reference's own factorisation has B + 1.  U = Q Q^T and Vf^T Vf -- all the update depends on -- are unchanged)
    const int n = B, n2 = 2 * n;
    // workspace carve (ctx->sg holds 8*rmax*max_D doubles, rmax = 2B+8)
    double* Qt = ctx->sg;                          // n x D
    double* P = Qt + (size_t)n * D;                // n x D
    double* Ft = P + (size_t)n * D;                // [Vf; Z]   2n x D
    double* Fs = Ft + (size_t)n2 * D;              // [Vf; -Z]  2n x D
    double* xbar = Fs + (size_t)n2 * D;
    double* gbar = xbar + D;
    double* N0 = ctx->small;                       // n x n   } stacked [N0; M1] = [P; Vf] Qt^T: ONE transposed panel product
    double* M1 = N0 + (size_t)n * n;               // n x n   }
    double* Ld = M1 + (size_t)n * n;               // n x n, then Ldinv (n), zg (n), vg (n)

    if (n > gsmvi_bam_small_nmax()) {              // checked before anything is enqueued
        gsmvi_set_error("%s: %s", "gsmvi_bam_update_f64", "B > 1024 exceeds the device chain (one entry per thread in k_bam_post_big)");
        return GSMVI_ERR_UNSUPPORTED;
    }
    bam_stats_launch(st, D, B, X, ldx, mu0, (const double*)nullptr, 0, G, ldg, reg, xbar, gbar, (double*)nullptr, Qt, Ft, Fs);
    int kc = 1, rc;
    if ((rc = gsmvi_panel_product_nc(ctx, st, nullptr, D, D, n, Qt, D, nullptr, 1.0, S0, lds0, ctx->pp, &kc))) return rc;
    if ((rc = gsmvi_panel_finish(st, D, n, kc, ctx->pp, nullptr, P, D))) return rc;
    // M1 = Vf Qt^T and N0 = P Qt^T share the right operand; P and Vf (the first n rows of Ft) are adjacent in the workspace,
    // so both Gram matrices come from one 2n-row transposed panel product, finished into the adjacent [N0; M1]
    const bool fused48_d = n <= gsmvi_bam_small_fused_nmax() && !ctx->tune_bam_full;
    if (!fused48_d && n <= 128) {                  // k_bam_nmat2 sums the slabs in its operand loads: fewer slabs
        if ((rc = gsmvi_panel_t_product_few_slabs(ctx, st, D, n2, P, D, Qt, D, n, ctx->pp, &kc))) return rc;
    } else if ((rc = gsmvi_panel_t_product(ctx, st, D, n2, P, D, Qt, D, n, ctx->pp, &kc))) return rc;
    double* Nd = Ld + (size_t)n * n + 3 * n;       // n x n
    double* M1T = Nd + (size_t)n * n;              // n x n
    double* Upk = M1T + (size_t)n * n;             // (n(n+1)/2 doubles once used by the substitution kernel: the layout is kept)
    const double* Ldinv = Ld + (size_t)n * n;
    // n <= 48: the whole small chain in ONE one-workgroup launch (k_bam_small48) + the product-form Z (k_bam_zw; round 5);
    // 48 < n <= 128 (and n <= 48 under the "bam_full" test knob): slab sums + N (k_bam_nmat2), the multi-workgroup Newton-Schulz
    // steps, BB / Cholesky with the inverse factor (k_bam_bbav, k_bam_cholw) and the product-form Z (k_bam_zw); n > 128: blocked
    // multi-workgroup Cholesky and the generic forward substitution.
    const bool fused48 = n <= gsmvi_bam_small_fused_nmax() && !ctx->tune_bam_full;
    const bool use_w = !fused48 && n <= 128;
    if (!fused48 && !ctx->bam_hint_host) {         // pinned, device-visible word for the step-count hint
        if (hipHostMalloc(reinterpret_cast<void**>(&ctx->bam_hint_host), 64, hipHostMallocMapped) == hipSuccess)
            *ctx->bam_hint_host = 0;
        else
            ctx->bam_hint_host = nullptr;
    }
    double* scratch = Upk + (size_t)n * (n + 1) / 2 + 2;
    int* info_p = info_dev ? info_dev : ctx->ints + 8;
    int* hint = ctx->tune_bam_full ? nullptr : ctx->bam_hint_host;
    if (fused48) {
        if ((rc = gsmvi_bam_small_fused(ctx, st, n, reg, ctx->pp, kc, n, (size_t)n2 * n, M1, Ld, info_p, nullptr))) return rc;
        hipLaunchKernelGGL(k_bam_zw, dim3((D + 15) / 16), dim3(512), 0, st, D, n, P, M1, Ld, Ldinv, Ldinv + 2 * n, mu0, xbar,
                           reg, Ft, Fs, mu, bamf_fix{});
    } else if (use_w) {
        const int nbq = (n + 15) / 16;
        BAM_NMAT2(kc, nbq, n, ctx->pp, (long long)n2 * n, n, N0, M1, Nd, (double*)nullptr);
        if ((rc = gsmvi_bam_small_device(ctx, st, n, reg, Nd, M1, N0, scratch, Ld, info_p, hint, ctx->tune_bam_kenq, M1T, nullptr, nullptr, nullptr)))
            return rc;
        // Z = W (P + M1^T Vf) by two chained MFMA products per 16 columns of D, the mean with it (Wt = (L^-1)^T sits in Ld's slot,
        // [a | . | vg] behind it)
        hipLaunchKernelGGL(k_bam_zw, dim3((D + 15) / 16), dim3(512), 0, st, D, n, P, M1, Ld, Ldinv, Ldinv + 2 * n, mu0, xbar,
                           reg, Ft, Fs, mu, bamf_fix{});
    } else {
        if ((rc = gsmvi_panel_finish(st, n, n2, kc, ctx->pp, nullptr, N0, n))) return rc;
        hipLaunchKernelGGL(k_bam_nmat, dim3((((n + 15) / 16) * ((n + 15) / 16) + 3) / 4), dim3(256), 0, st, n, M1, N0, Nd, M1T);
        if ((rc = gsmvi_bam_small_device(ctx, st, n, reg, Nd, M1, N0, scratch, Ld, info_p, hint, ctx->tune_bam_kenq, nullptr, nullptr, nullptr, nullptr)))
            return rc;
        // round 6: blocked substitution on the MFMA pipe when the factorisation left its diagonal blocks' inverses (k_potrf_dag)
        if (ctx->potrf_w && ctx->potrf_w_n == n && !ctx->tune_no_fast) {
            for (int rb = 0; rb < (n + 63) / 64; ++rb)
                hipLaunchKernelGGL(k_bam_fwd_block, dim3((D + 15) / 16), dim3(512), 0, st, D, n, rb, P, M1, ctx->potrf_r, ctx->potrf_w,
                                   Ft, Fs);
            hipLaunchKernelGGL(k_bam_mean_big, dim3((D + 255) / 256), dim3(256), 0, st, D, n, P, Ft, Ldinv + n, Ldinv + 2 * n, mu0,
                               xbar, reg, mu);
        } else {
#define BFW(CV) hipLaunchKernelGGL(k_bam_forward<CV>, dim3((D + CV - 1) / CV), dim3(CV), sizeof(double) * 2 * n * CV, st, D, n, P, M1, Ld, Ldinv, Ldinv + n, Ldinv + 2 * n, mu0, xbar, reg, Ft, Fs, mu)
        if (n <= 160) BFW(64); else if (n <= 320) BFW(32); else if (n <= 640) BFW(16); else BFW(8);   // (LDS: 16 n COLS bytes <= 160 KB)
#undef BFW
        }
    }
    const int nt = (D + 63) / 64;
    ctx->path |= (!ctx->tune_no_fast && D % 2 == 0) ? GSMVI_PATH_LOWRANK_FAST : GSMVI_PATH_LOWRANK_GENERIC;
    if (!ctx->tune_no_fast && D % 2 == 0 && n2 > 288) {          // round 6: any number of factor rows on the MFMA tile kernel
        if (D % 64 != 0) hipLaunchKernelGGL(k_lowrank_update_big<true>, dim3(nt * nt), dim3(512), 0, st, D, n2, Ft, Fs, S0, lds0, S, lds, jitter);
        else hipLaunchKernelGGL(k_lowrank_update_big<false>, dim3(nt * nt), dim3(512), 0, st, D, n2, Ft, Fs, S0, lds0, S, lds, jitter);
    } else if (!ctx->tune_no_fast && D % 2 == 0 && n2 <= 288) {
#define LRU(NPV, KPV, RG) hipLaunchKernelGGL((k_lowrank_update_fast<NPV, KPV, RG>), dim3(nt * nt), dim3(512), 0, st, D, n2, Ft, Fs, S0, lds0, S, lds, jitter)
        // (KF > 96: 32-row staging passes as in round 4; 64-row passes -- half the barriers -- measured equal: knob "lowrank_kp" = 64)
        if (D % 64 != 0) {
            if (n2 <= 96) LRU(3, 32, true); else if (ctx->tune_lowrank_kp == 64) LRU(5, 64, true); else LRU(9, 32, true);
        } else {
            if (n2 <= 96) LRU(3, 32, false); else if (ctx->tune_lowrank_kp == 64) LRU(5, 64, false); else LRU(9, 32, false);
        }
#undef LRU
    } else {
        hipLaunchKernelGGL(k_lowrank_update, dim3(nt * (nt + 1) / 2), dim3(256), 0, st, D, n2, Ft, Fs, S0, lds0, S, lds,
                           jitter);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gsmvi_set_error("BaM launch failed: %s%s", hipGetErrorString(e), "");
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}

// ===== factor-form BaM update: Sigma = F^T F, no D x D covariance and no D^3 factorisation =================================
// (north_star's factor-form extension applied to bam.py:72-114)  With S0 = F0^T F0, x_b = mu0 + z_b F0:
//   U = Q Q^T depends on the scores only through  sum_b (g_b - gbar)(g_b - gbar)^T  and gbar gbar^T, and the centred rows span
//   B - 1 dimensions: an orthonormal (Helmert) recombination of the B centred rows gives B - 1 rows with the same Gram sum,
//   so Q gets n = B columns instead of B + 1 (rows k < B-1: sqrt(reg/B) helmert_k(g), row B-1: sqrt(reg/(1+reg)) gbar), and the
//   same for the sample factor Vf.  BaM's update depends on Q and Vf only through Q Q^T and Vf^T Vf (bam.py:31-69), so S is unchanged.
//   Everything of bam.py:107-111 is then done in WHITENED coordinates (Vf = Vw F0, P = Wq F0, Z = Zw F0):
//     Wq = Qt F0^T,  [N0; M1] = [Wq; Vw] Wq^T,  Zw = L^-1 (Wq + M1^T Vw),   S = F0^T (I + Vw^T Vw - Zw^T Zw) F0,
//   and the factor of M = I + Rt^T J Rt, Rt = [Vw; Zw], J = diag(I, -I) is taken by the SAME 2B x 2B chain as the GSM factor
//   update (gsmvi_factor.hip, jmode): F = F0 + Rt^T K (Rt F0).  The mean (bam.py:112) needs S gbar = (h F0) with
//   h = wg + Vw^T (Vw wg) - Zw^T (Zw wg), wg = F0 gbar: k_bam_zw emits r1 h as an extra row of Rt (its
//   "mean" output with mu0 = xbar = 0), which rides through the Rt F0 product.  Four passes over F0 (Wq, Rt F0, and the
//   read + write of the update), no pass over a covariance.
// mu = mu0/(1+reg) + r1 (S gbar) + r1 xbar, or mu0 on a reverted update (bam.py:112)
__global__ __launch_bounds__(256) void k_bamf_commit(int D, const double* __restrict__ sg_r1, const double* __restrict__ mu0,
                                                     const double* __restrict__ xbar, bam_reg regs,
                                                     const int* __restrict__ bad, double* __restrict__ mu) {
    const double reg = regs.get();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    const double r1 = reg / (1.0 + reg);
    mu[i] = *bad ? mu0[i] : mu0[i] / (1.0 + reg) + sg_r1[i] + r1 * xbar[i];
}

int gsmvi_panel_t_product(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* A, int lda, const double* M,
                          int ldm, int mrows, double* Pp, int* kc_out);
int gsmvi_factor_signed_gram(gsmvi_ctx* ctx, hipStream_t st, int D, int Bh, int* kcg, int* info_dev, int* rides);
int gsmvi_factor_signed_back(gsmvi_ctx* ctx, hipStream_t st, int D, int Bh, const double* mu0, const double* F0, int ldf0,
                             double* mu, double* F, int ldf, int* info_dev, int* n_reverts_dev, int kcg, int rides, int taken,
                             int join);

// ---- the orthogonal-basis form (round 5, knob "bam_basis"): small pieces ----------------------------------------------------------
// Pi = W Dm^T (one 16 x 16 block per workgroup, gsmvi_smallgemm.h) and, in ONE more workgroup of the same launch,
// vg' = vg - Pi^T zg with zg = W a (the true Zw gbar-vector of bam.py:110) and Pi^T = Dm W^T: the mean of bam.py:112 needs
// Zw^T zg = Zt^T zg + Vw^T Pi^T zg, and k_bam_zw forms Vw^T vg - Z^T zg from the Z it computes (Zt here).  n <= 128.
// The three matrix-vector products take a WAVE per row where the row is contiguous (first form: a thread per row walking it
// with stride n -- 24 us at n = 128) and a thread per column where the column is.  The same workgroup joins the flag of Gvv's
// factorisation (info1, when that ran beside the chain on the second stream) to the flag of the chain.
// y[i] = sum_k M[i][k] x[k] (UPPER: k >= i only) for a row-major n x n matrix in global memory, n <= 128, 256 threads: eight
// lanes per row (lane `part` takes k = part, part + 8, ...: the eight lanes of a row read 64 contiguous bytes), 32 rows per pass,
// all 64 loads of a thread in flight at once; the eight partial sums meet in three DPP steps (fixed order).  x, y in LDS.
template <bool UPPER>
__device__ __forceinline__ void bamf_matvec_rows(int n, const double* __restrict__ M, const double* xs, double* ys) {
    const int row0 = threadIdx.x >> 3, part = threadIdx.x & 7;
    double v[4][16];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = row0 + 32 * p, ic = i < n ? i : n - 1;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = part + 8 * u;
            v[p][u] = M[(size_t)ic * n + (k < n ? k : n - 1)];
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = row0 + 32 * p;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
            const int k = part + 8 * u;
            a0 += (k < n && (!UPPER || k >= i)) ? v[p][u] * xs[k < n ? k : 0] : 0.0;
            a1 += (k + 8 < n && (!UPPER || k + 8 >= i)) ? v[p][u + 1] * xs[k + 8 < n ? k + 8 : 0] : 0.0;
        }
        double sacc = a0 + a1;
        sacc += dpp_f64<0xB1>(sacc);              // neighbour
        sacc += dpp_f64<0x4E>(sacc);              // other pair of the quad
        sacc += dpp_f64<0x141>(sacc);             // row_half_mirror: the other quad of the eight
        if (part == 0 && i < n) ys[i] = sacc;
    }
}
// t2 = W^T (W a) for the vg' correction below: zg[r] = sum_{k <= r} Wt[k][r] a[k] (thread (h, r) walks half a column, coalesced
// over r), then t2[k] = sum_{r >= k} Wt[k][r] zg[r].  256 threads, n <= 128; sa, zg, part: LDS scratch (128, 128, 256); t2 -> LDS.
__device__ __forceinline__ void bamf_t2_body(int n, const double* __restrict__ Wt, const double* __restrict__ av, double* sa,
                                             double* zg, double* part, double* t2) {
    const int tid = threadIdx.x;
    if (tid < 128) sa[tid] = tid < n ? av[tid] : 0.0;
    __syncthreads();
    {
        const int r = tid & 127, h = tid >> 7, rc = r < n ? r : n - 1;
        double v[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            const int k = 64 * h + u;
            v[u] = Wt[(size_t)(k < n ? k : n - 1) * n + rc];
        }
        double z0 = 0.0, z1 = 0.0;
#pragma unroll
        for (int u = 0; u < 64; u += 2) {
            const int k = 64 * h + u;
            z0 += (k <= r && k < n) ? v[u] * sa[k] : 0.0;
            z1 += (k + 1 <= r && k + 1 < n) ? v[u + 1] * sa[k + 1] : 0.0;
        }
        part[tid] = z0 + z1;
    }
    __syncthreads();
    if (tid < 128) zg[tid] = tid < n ? part[tid] + part[tid + 128] : 0.0;
    __syncthreads();
    bamf_matvec_rows<true>(n, Wt, zg, t2);
    __syncthreads();
}
// T = W11 M1 (one 16 x 16 block per workgroup) and, in one more workgroup, t2 (n > 64: this launch follows k_bam_cholw's)
__global__ __launch_bounds__(256) void k_bamf_t_t2(OpBasisT op, int nblk, const double* __restrict__ Wt,
                                                   const double* __restrict__ av, double* __restrict__ t2g) {
    __shared__ double As[16 * SMALLGEMM_SA], Bs[256 * SMALLGEMM_SB], red[4 * 256];
    if ((int)blockIdx.x < nblk) {
        small_gemm_block(op, (int)blockIdx.x, As, Bs, red);
        return;
    }
    const int n = op.m;
    bamf_t2_body(n, Wt, av, As, As + 128, As + 384, As + 256);
    if ((int)threadIdx.x < n) t2g[threadIdx.x] = As[256 + threadIdx.x];
}
// All global loads of a product are issued before its first multiply (a load per iteration behind a dependent reduction costs
// an L2 round trip per row: 37 us for this workgroup at n = 128 when first written; a wave per row with 64 wave-wide
// reductions per wave still 20 us): the two row-access products take EIGHT lanes per row (bamf_matvec_rows).
__global__ __launch_bounds__(256) void k_bamf_pi_vg(OpBasisPi op, int nblk, const double* __restrict__ av, double* __restrict__ vg,
                                                    int* info, const int* info1, const double* __restrict__ t2g) {
    __shared__ double As[16 * SMALLGEMM_SA], Bs[256 * SMALLGEMM_SB], red[4 * 256];
    if ((int)blockIdx.x < nblk) {
        small_gemm_block(op, (int)blockIdx.x, As, Bs, red);
        return;
    }
    const int n = op.m, tid = threadIdx.x;
    double *t2 = As + 256, *part = As + 384;
    if (tid == 0 && info1 && *info == 0 && *info1 != 0) *info = 1000 + *info1;
    if (t2g) {                                     // (t2 came with the T launch)
        if (tid < 128) t2[tid] = tid < n ? t2g[tid] : 0.0;
        __syncthreads();
    } else
        bamf_t2_body(n, op.Wt, av, As, As + 128, part, t2);
    bamf_matvec_rows<false>(n, op.Dm, t2, part);  // vg'[i] = vg[i] - sum_k Dm[i][k] t2[k]
    __syncthreads();
    if (tid < n) vg[tid] -= part[tid];
}

// [A | I] -> [R | W] of one n x n matrix (n <= 128) in its own launch: the first diagonal block Gvv = Vw Vw^T; plain
// positive-definite rule (dependent draws are a failure, not a drop)
int gsmvi_cholw_small(hipStream_t st, int n, const double* A, double* R, double* W, int* info, int info_off);

int gsmvi_bam_factor_impl(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* Z, int ldz, const double* X, int ldx,
                          const double* G, int ldg, const double* mu0, const double* F0, int ldf0, double reg_value, double* mu,
                          double* F, int ldf, int* info_dev, int* n_reverts_dev) {
    const bam_reg reg{reg_value, ctx->reg_dev};
    const int n = B, n2 = 2 * n;
    // workspace (ctx->sg holds 8 rmax max_D doubles, rmax = 2B + 8): 10 n + 5 rows of D
    double* Qt = ctx->sg;                          // n x D
    double* Wq = Qt + (size_t)n * D;               // n x D      } [Wq; Vw]: left operand of the Gram product
    double* Ft = Wq + (size_t)n * D;               // [Vw; Zw; r1 h]  (2n + 1) x D: Rt of the factor chain + the mean's row
    double* T1 = Ft + (size_t)(n2 + 1) * D;        // rows n .. 2n-1 hold -Zw (k_bam_zw's Fs)
    double* Tm = T1 + (size_t)n2 * D;              // (2n + 1) x D = Ft F0
    double* Fsf = Tm + (size_t)(n2 + 1) * D;       // 2n x D: K'' Tm of the generic update path
    double* xbar = Fsf + (size_t)n2 * D;
    double* gbar = xbar + D;
    double* zerov = gbar + D;
    double* N0 = ctx->small;                       // n x n   } stacked [N0; M1] = [Wq; Vw] Wq^T: ONE transposed panel product
    double* M1 = N0 + (size_t)n * n;
    double* Ld = M1 + (size_t)n * n;               // n x n, then Ldinv (n), zg (n), vg (n)
    double* Nd = Ld + (size_t)n * n + 3 * n;
    double* M1T = Nd + (size_t)n * n;
    double* Upk = M1T + (size_t)n * n;
    double* scratch = Upk + (size_t)n * (n + 1) / 2 + 2;
    const double* Ldinv = Ld + (size_t)n * n;
    int* info_bam = ctx->ints + 8;
    int kc = 1, rc;
    // n <= 48: one one-workgroup launch for the small chain (k_bam_small48); above: the chain of the
    // dense form (k_bam_nmat2, Newton-Schulz steps, k_bam_bbav, k_bam_cholw) and the product-form Zw (k_bam_zw).
    // Zw = L^-1 (Wq + M1^T Vw); the mean output of either kernel (mu0 = xbar = 0) is r1 (wg + Vw^T vg - Zw^T zg) = r1 h, row 2n of Ft
    // Orthogonal-basis form (round 5, "bam_basis"): Rt = [Vw; Zt] with Zt the part of Zw orthogonal to the whitened draws -- no
    // linearly dependent rows when the fit sits on the fixed point of a Gaussian target.  It needs the explicit
    // W = L^-1 (every chain variant delivers it since round 5) and the factor of Gvv = Vw Vw^T (beside the chain: DESIGN 4.7).
    const bool basis = ctx->tune_bam_basis != 0 && n <= 128 && ctx->basis != nullptr;
    const bool fused48 = n <= gsmvi_bam_small_fused_nmax() && !ctx->tune_bam_full;

    // (Tm rows 0 .. n-1 = Vw F0 = the Helmert rows of the samples themselves: the stats kernel's third track)
    bam_stats_launch(st, D, B, Z, ldz, (const double*)nullptr, X, ldx, G, ldg, reg, xbar, gbar, zerov, Qt, Ft, (double*)nullptr, mu0, Tm);
    if ((rc = gsmvi_panel_t_product(ctx, st, D, n, Qt, D, F0, ldf0, D, ctx->pp, &kc))) return rc;
    if ((rc = gsmvi_panel_finish(st, D, n, kc, ctx->pp, nullptr, Wq, D))) return rc;
    // Two-level 2B x 2B chain ahead (64 < B <= 128): its first diagonal block Gamma11 = Vw Vw^T does not depend on the B x B
    // chain, so the Gram product below is taken against [Wq; Vw] (2n columns instead of n: the slabs then hold Vw Vw^T as well)
    // and [Gamma11 | I] -> [R11 | W11] runs as the second workgroup of k_bam_cholw's launch (ctx->early; factor_chain_big).
    ctx->early_ready = 0;
    const bool early = !fused48 && n > 64 && n <= 128 && (ctx->tune_chain_pair || basis) && ctx->early;
    const int gcols = (early || basis) ? n2 : n;
    if (!fused48 && n <= 128) {                    // k_bam_nmat2 sums the slabs in its operand loads: fewer slabs
        if ((rc = gsmvi_panel_t_product_few_slabs(ctx, st, D, n2, Wq, D, Wq, D, gcols, ctx->pp, &kc))) return rc;
    } else if ((rc = gsmvi_panel_t_product(ctx, st, D, n2, Wq, D, Wq, D, gcols, ctx->pp, &kc))) return rc;
    if (fused48) {
        // orthogonal basis with the one-launch chain: everything that does not need the chain's L is the launch's second
        // workgroup (bamq_side), what does rides in k_bam_zw's launch (bamf_fix)
        const size_t q2 = (size_t)(ctx->rmax / 2) * (ctx->rmax / 2);
        double* M1p = ctx->basis + q2;
        double* Dm = ctx->basis + 2 * q2;
        double* Pi = ctx->basis + 3 * q2;
        double* t2 = ctx->basis;                   // (the T block: T lives in the side workgroup's LDS here)
        double* R11s = ctx->early + 128 * 128;     // (the slots the n > 64 path uses for the same block)
        double* W11s = ctx->early + 2 * 128 * 128;
        const bamq_side side{M1p, Dm, t2, ctx->ints + 10, R11s, W11s};
        if ((rc = gsmvi_bam_small_fused(ctx, st, n, reg, ctx->pp, kc, gcols, (size_t)n2 * gcols, M1, Ld, info_bam,
                                        basis ? &side : nullptr)))
            return rc;
        ctx->chain_pi = basis ? Pi : nullptr;
        ctx->chain_x = basis ? ctx->basis + 4 * q2 : nullptr;
        ctx->chain_r11 = (basis && ctx->tune_bam_basis != 3) ? R11s : nullptr;   // (bam_basis = 3: the chain factors Gamma11 itself, A/B)
        ctx->chain_w11 = (basis && ctx->tune_bam_basis != 3) ? W11s : nullptr;
        const bamf_fix fx = basis ? bamf_fix{Dm, t2, Pi, info_bam, ctx->ints + 10} : bamf_fix{};
        hipLaunchKernelGGL(k_bam_zw, dim3((D + 15) / 16 + (basis ? 1 : 0)), dim3(512), 0, st, D, n, Wq, basis ? M1p : M1, Ld, Ldinv,
                           Ldinv + 2 * n, zerov, zerov, reg, Ft, T1, Ft + (size_t)n2 * D, fx);
    } else {
        if (!ctx->bam_hint_host) {
            if (hipHostMalloc(reinterpret_cast<void**>(&ctx->bam_hint_host), 64, hipHostMallocMapped) == hipSuccess)
                *ctx->bam_hint_host = 0;
            else
                ctx->bam_hint_host = nullptr;
        }
        const int nbq = (n + 15) / 16;
        double* G11 = ctx->early;                  // n x n each, compact
        double* R11 = ctx->early + 128 * 128;
        double* W11 = ctx->early + 2 * 128 * 128;
        BAM_NMAT2(kc, nbq, n, ctx->pp, (long long)n2 * gcols, gcols, N0, M1, Nd, (early || basis) ? G11 : (double*)nullptr);
        const size_t q2 = (size_t)(ctx->rmax / 2) * (ctx->rmax / 2);
        double* Tb = ctx->basis;                   // n x n each: T, M1', Dm = M1 - M1', Pi, X
        double* M1p = Tb + q2;
        double* Dm = M1p + q2;
        double* Pi = Dm + q2;
        // Orthogonal basis: Gvv = Vw Vw^T, its factor [R11 | W11] and M1' = -Gvv^-1 M1 do not depend on BaM's B x B chain.
        //   n <= 64 (the iteration is ONE workgroup, k_bam_ns64): all of it is the second workgroup of that launch (bamq_side_body);
        //   n  > 64: the factorisation is the second workgroup of k_bam_cholw's launch (the 2B x 2B chain takes [R11 | W11] as its
        //            first block), the two products T = W11 M1, M1' = -W11^T T follow it.
        // (Measured and dropped: the three pieces on the context's second stream beside the Newton-Schulz launches.  An update
        // timed alone gained 15 us at (1024, 128), but back-to-back updates -- a fit -- LOST 8 us, 503 against 495 us: the two
        // cross-queue waits cost more than the 50 us of work they hide once both queues are busy.  scripts/bamf_trace.py.)
        const bool side64 = basis && gsmvi_bam_small_one_wg(ctx, n);
        int* info_side = ctx->ints + 10;           // n <= 64: joined to BaM's flag by k_bamf_pi_vg (a dependent draw reverts the update)
        const bamq_side sd{M1p, Dm, nullptr, info_side, R11, W11};       // ([R11 | W11] also for the 2B x 2B chain: its first block)
        // (the magnitude guard of the rank-revealing rule sees this block's own diagonal: the second block's is not known yet)
        const cholw_job beside{n, G11, n, R11, n, W11, n, ctx->ints, 0, 0, nullptr, 0, 0, basis ? 1 : 0};
        if ((rc = gsmvi_bam_small_device(ctx, st, n, reg, Nd, M1, N0, scratch, Ld, info_bam,
                                         ctx->tune_bam_full ? nullptr : ctx->bam_hint_host, ctx->tune_bam_kenq, M1T,
                                         early ? &beside : nullptr, side64 ? &sd : nullptr, G11)))
            return rc;
        ctx->early_ready = early ? 1 : 0;
        const double* M1z = M1;                    // the n x n matrix k_bam_zw multiplies Vw with: M1, or M1' in the orthogonal basis
        ctx->chain_pi = nullptr;
        if (basis) {
            const int nblk = ((n + 15) >> 4) * ((n + 15) >> 4);
            double* t2g = Pi + 2 * q2;             // n doubles behind X
            if (!side64) {
                if (!early && (rc = gsmvi_cholw_small(st, n, G11, R11, W11, info_side, 0))) return rc;
                hipLaunchKernelGGL(k_bamf_t_t2, dim3(nblk + 1), dim3(256), 0, st, OpBasisT{n, n, n, W11, M1, Tb, n}, nblk, Ld, Ldinv,
                                   t2g);
                small_gemm_launch(st, OpBasisM1p{n, n, n, W11, Tb, M1, M1p, Dm, n});
            }
            hipLaunchKernelGGL(k_bamf_pi_vg, dim3(nblk + 1), dim3(256), 0, st, OpBasisPi{n, n, n, Ld, Dm, Pi}, nblk, Ldinv,
                               const_cast<double*>(Ldinv) + 2 * n, info_bam, early ? (const int*)nullptr : info_side,
                               side64 ? (const double*)nullptr : t2g);
            M1z = M1p;
            ctx->chain_pi = Pi;
            ctx->chain_x = Pi + q2;
            if (n <= 64 && ctx->tune_bam_basis != 3) {   // 2B <= 128: the chain takes [R11 | W11] as given (side64: written by the side
                ctx->chain_r11 = R11;                    // workgroup; else by gsmvi_cholw_small above)
                ctx->chain_w11 = W11;
            }
        }
        hipLaunchKernelGGL(k_bam_zw, dim3((D + 15) / 16), dim3(512), 0, st, D, n, Wq, M1z, Ld, Ldinv, Ldinv + 2 * n, zerov, zerov,
                           reg, Ft, T1, Ft + (size_t)n2 * D, bamf_fix{});
    }
    ctx->fo_Rt = Ft;
    ctx->fo_Tm = Tm;
    ctx->fo_Fs = Fsf;
    ctx->bam_mean = gsmf_bam_mean{xbar, Tm + (size_t)n2 * D, reg};   // 2B <= 64: the update kernel writes BaM's mean itself
    ctx->bam_mean_done = 0;
    int kcg = 1, rides = 0;
    // Large D, 64 < 2B <= 128 (the 2B x 2B chain is six small launches, ~115 us on a few CUs): the product Rt F0 (MFMA-bound,
    // 92 us at D = 4096) depends on Ft only, so it runs on the context's second stream beside the Gram product and the chain and is
    // joined in front of K'' Tm -- what the GSM factor update does with V Fm (gsmvi_factor.hip; same threshold: the two event
    // edges cost ~10 us).  (4096, 64): 684 -> 595 us eager, 679 -> 654 replayed; (3072, 40): 523 -> 476.  Not for 2B > 128: there
    // the 257-row product fills every CU for ~180 us and the chain's twenty small launches queue behind its workgroups (983 us
    // against 971 at (4096, 128)).
    const bool fork_tm = n2 > 64 && n2 <= 128 && ctx->side && ctx->tune_fork_min_D > 0 && D >= ctx->tune_fork_min_D && !ctx->tune_no_fast;
    if (fork_tm) {
        hipError_t fe = hipEventRecord(ctx->ev_fork, st);
        if (fe == hipSuccess) fe = hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0);
        if (fe != hipSuccess) { gsmvi_set_error("%s: %s", "gsmvi_bam_factor_impl", "fork failed"); rc = GSMVI_ERR_HIP; }
        if (!rc) rc = gsmvi_panel_product_out(ctx, ctx->side, D, D, n + 1, Ft + (size_t)n * D, D, nullptr, 1.0, F0, ldf0, nullptr, Tm + (size_t)n * D, D);
        if (!rc && hipEventRecord(ctx->ev_join, ctx->side) != hipSuccess) {
            gsmvi_set_error("%s: %s", "gsmvi_bam_factor_impl", "join failed");
            rc = GSMVI_ERR_HIP;
        }
        if (!rc) rc = gsmvi_factor_signed_gram(ctx, st, D, n, &kcg, info_dev, &rides);
        ctx->px = gsmvi_panel_extras();            // nothing rides in a panel launch here: the chain sums the Gram slabs itself
        if (!rc)
            rc = gsmvi_factor_signed_back(ctx, st, D, n, mu0, F0, ldf0, mu, F, ldf, info_dev, n_reverts_dev, kcg, 0, 0, 1);
    } else {
        rc = gsmvi_factor_signed_gram(ctx, st, D, n, &kcg, info_dev, &rides);   // Gram slabs of [Vw; Zw]; the 2B x 2B chain rides in ...
        // ... this product; with the chain riding (~35 us on one CU) it is off the critical path: unsplit, finished output, no
        // finish launch (gsmvi_factor_impl does the same with V Fm)
        const int kc_user = ctx->tune_panel_kc;
        if (rides && D >= 1024 && D <= ctx->tune_rider_direct_max_D) ctx->tune_panel_kc = 1;
        if (!rc) rc = gsmvi_panel_product_out(ctx, st, D, D, n + 1, Ft + (size_t)n * D, D, nullptr, 1.0, F0, ldf0, nullptr, Tm + (size_t)n * D, D);
        ctx->tune_panel_kc = kc_user;
        const int taken = ctx->px_used;
        ctx->px = gsmvi_panel_extras();
        if (!rc)
            rc = gsmvi_factor_signed_back(ctx, st, D, n, mu0, F0, ldf0, mu, F, ldf, info_dev, n_reverts_dev, kcg, rides, taken, 0);
    }
    ctx->fo_Rt = ctx->fo_Tm = ctx->fo_Fs = nullptr;
    ctx->chain_pi = nullptr;
    ctx->chain_x = nullptr;
    ctx->chain_r11 = ctx->chain_w11 = nullptr;
    const bool mean_done = ctx->bam_mean_done != 0;
    ctx->bam_mean = gsmf_bam_mean{nullptr, nullptr, {0.0, nullptr}};
    ctx->bam_mean_done = 0;
    if (rc) return rc;
    if (!mean_done)
    hipLaunchKernelGGL(k_bamf_commit, dim3((D + 255) / 256), dim3(256), 0, st, D, Tm + (size_t)n2 * D, mu0, xbar, reg, info_dev,
                       mu);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gsmvi_set_error("BaM (factor form) launch failed: %s%s", hipGetErrorString(e), "");
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}

hipError_t gsmvi_bam_prepare() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bam_forward<64>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bam_forward<32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bam_forward<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bam_forward<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
    return e;
}
