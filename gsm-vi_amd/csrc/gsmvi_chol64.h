// In-LDS Cholesky of a 64 x 64 block (shared by gsmvi_potrf.hip and gsmvi_factor.hip).
#pragma once
#include "gsmvi_common.h"
#ifndef CHOL_STAMP
#define CHOL_STAMP(k) ((void)0)     // scripts/chol64bench.hip defines it to time the phases
#endif

#define TS 66
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)(u & 0xffffffffu), lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), lane);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// Factor the nb x nb (nb <= 64) upper block held in T[64][STR] (LDS, padded with identity beyond nb);
// rinv[p] = 1/R[p][p].  *sh_fail = 1-based local index of the first bad pivot (0 = ok).
// Blocked in four 16-column steps.  Per step: (1) wave 0 factors the 16x16 diagonal block entirely in
// registers -- lane j holds column j, pivots and multipliers are broadcast with v_readlane, so the 16
// sequential pivots cost no LDS round trip and no barrier; (2) the 16 x (rest) block row is solved
// one column per thread; (3) the trailing block gets its rank-16 update.  Three barriers per step.
template <int STR>
__device__ __forceinline__ void chol64_lds_s(double* T, double* rinv, int nb, int* sh_fail) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) *sh_fail = 0;
    __syncthreads();
    for (int kb = 0; kb < 4; ++kb) {
        const int k0 = 16 * kb;
        if (k0 >= nb) break;                                     // block-uniform
        CHOL_STAMP(1 + 4 * kb);
        if (w == 0) {
            const int j = lane & 15;
            double col[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) col[i] = T[(k0 + i) * STR + k0 + j];
            int fail = 0;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const double d = readlane_f64(col[p], p);
                const bool ok = d > 0.0 && d < 1.7976931348623157e308;   // false for NaN, <= 0, inf
                if (!ok && fail == 0) fail = k0 + p + 1;
                // serial pivot chain: hardware v_rsq_f64 estimate + Newton steps (full fp64 accuracy,
                // ~10 dependent ops instead of the library rsqrt's ~30), then r = d * rsqrt(d)
                const double dd = ok ? d : 1.0;
                double y = __builtin_amdgcn_rsq(dd);              // ~2^-26 relative error
                y = y * (1.5 - 0.5 * dd * y * y);                 // Newton: error^2 -> fp64 precision
                y = y * (1.5 - 0.5 * dd * y * y);                 // second step kept: v_rsq_f64 accuracy is not documented
                const double ri = ok ? y : 0.0;
                const double r = ok ? d * ri : 1.0;
                col[p] = (j == p) ? r : col[p] * ri;             // row p of the factor (entries j > p matter)
                if (lane == p) rinv[k0 + p] = ri;
#pragma unroll
                for (int i = p + 1; i < 16; ++i) {
                    const double tpi = readlane_f64(col[p], i);  // R[p][i]
                    col[i] -= tpi * col[p];
                }
            }
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (i <= j) T[(k0 + i) * STR + k0 + j] = col[i];
            }
            if (lane == 0 && fail != 0 && k0 + 0 < nb && *sh_fail == 0 && fail <= nb) *sh_fail = fail;
        }
        __syncthreads();
        CHOL_STAMP(2 + 4 * kb);
        const int rest0 = k0 + 16;                               // first column to the right
        // (2) block row: solve R_dd^T x = T[k0..k0+15][c] for every column c >= rest0
        for (int cc = rest0 + tid; cc < 64; cc += 256) {
            double x[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = T[(k0 + i) * STR + cc];
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                x[p] *= rinv[k0 + p];
#pragma unroll
                for (int i = p + 1; i < 16; ++i) x[i] -= T[(k0 + p) * STR + k0 + i] * x[p];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) T[(k0 + i) * STR + cc] = x[i];
        }
        __syncthreads();
        CHOL_STAMP(3 + 4 * kb);
        if (rest0 >= 64) break;                                  // last block: no trailing matrix (block-uniform)
        // (3) trailing update: T[i][q] -= sum_p T[k0+p][i] T[k0+p][q], rest0 <= i <= q < 64.
        // 16 x 16 threads, each owns up to 3 x 3 elements (i = rest0+ty+16a, q = rest0+tx+16b); the
        // operand values are read in one batch per p so the LDS latency is paid once, not per FMA.
        {
            const int ty = tid >> 4, tx = tid & 15;
            double acc[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) acc[a][b] = 0.0;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                double ra[3], rb[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const int i = rest0 + ty + 16 * a, q = rest0 + tx + 16 * a;
                    ra[a] = T[(k0 + p) * STR + (i < 64 ? i : 63)];
                    rb[a] = T[(k0 + p) * STR + (q < 64 ? q : 63)];
                }
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) acc[a][b] += ra[a] * rb[b];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const int i = rest0 + ty + 16 * a, q = rest0 + tx + 16 * b;
                    if (i < 64 && q < 64 && q >= i) T[i * STR + q] -= acc[a][b];
                }
        }
        __syncthreads();
        CHOL_STAMP(4 + 4 * kb);
    }
}


// Register-resident variant with the same contract.  Thread (ty, tx) of the 16 x 16 grid keeps the 4 x 4 set
// S[ty+16a][tx+16b] in registers for the whole factorisation; pivot p costs ONE barrier: the threads holding
// row p publish it (unscaled) into T[p][.], everybody reads the diagonal d_p, the row entries of its own
// rows and columns, and applies S[i][j] -= (S[p][i] / d_p) S[p][j].  The register slot that contains row
// p + 1 is updated first and that row is published before the rest of the update is issued, so the next
// pivot's LDS round trip overlaps the bulk of the arithmetic.  Only entries with j >= i > p are ever consumed,
// so finished rows and the lower triangle may be overwritten with garbage freely.  The published rows are
// scaled by d_p^-1/2 at the end (R[p][j] = S_p[p][j] / sqrt(d_p)).
// (Measured on MI355X, one workgroup: 12 us against 23 us for the blocked variant above; taking four pivots per
// barrier with a redundant 4 x 4 factor in every thread was no faster -- the chain is instruction latency:
// dropping the barrier altogether only takes 11.2 -> 9.1 us, the Newton steps cost nothing.)
// SEMIDEF = true factors a positive SEMI-definite matrix (a Gram matrix G = Rt Rt^T whose rows may be linearly
// dependent).  The CALLER has lowered every diagonal entry by its rounding floor, G_pp <- G_pp (1 - GSMVI_DEP_TOL)
// (a relative perturbation of 64 eps, scale-free per row, so rows of tiny norm are fine); a pivot <= 0 then means
// "not above the rounding floor of its row": it cannot be told from zero in fp64 -- and a tiny POSITIVE noise pivot is
// as harmful as a negative one (1 / sqrt(1e-34) in the factor: scripts/dbg_dep128.py), hence the shift instead of a
// plain sign test.  (Comparing against a per-pivot floor read from LDS inside the loop put that read on the pivot chain:
// 12.2 -> 14.8 us per 64 x 64 block.)  Such a row is DEPENDENT: its row of the factor and its diagonal come out as
// ZERO, rinv[p] = 0, the pivot is skipped in the elimination, no failure is reported (R^T R still equals G to
// rounding).  allow_dep (block-uniform) = false turns the dependent verdict into a FAILURE instead: the callers pass
// "every diagonal entry is of moderate size" (max_p G_pp < 2^32), because dropping a row perturbs the represented matrix
// by up to sqrt(GSMVI_DEP_TOL) |row| -- harmless for whitened draws (|z|^2 ~ D), meaningless for |z| ~ 1e10 (fixture
// G4).  NaN / inf pivots fail in every mode.  SEMIDEF = false is the plain positive-definite test (pivot <= 0 fails).
#define GSMVI_DEP_TOL 1.4210854715202004e-14        /* 64 eps */
template <int STR, bool SEMIDEF = false>
__device__ __forceinline__ void chol64_rows_s(double* T, double* rinv, int nb, int* sh_fail, bool allow_dep = true) {
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    double s[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) s[a][b] = T[(ty + 16 * a) * STR + tx + 16 * b];
    int fail = 0;
    CHOL_STAMP(1);
    // row 0 is already in T
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
        if (16 * pb >= nb) break;                                 // block-uniform; rows beyond nb are identity
#pragma unroll 1
        for (int pq = 0; pq < 16; ++pq) {
            const int p = 16 * pb + pq;
            const double* row = T + p * STR;
            __syncthreads();
            const double d = row[p];
            double ri[4], rj[4];
#pragma unroll
            for (int a = pb; a < 4; ++a) ri[a] = row[ty + 16 * a];
#pragma unroll
            for (int b = pb; b < 4; ++b) rj[b] = row[tx + 16 * b];
            const bool ok = d > 0.0 && d < 1.7976931348623157e308;   // false for NaN, <= 0, inf
            const bool dep = SEMIDEF && allow_dep && d <= 0.0 && d > -1.7976931348623157e308;   // dependent row
            if (!ok && !dep && fail == 0) fail = p + 1;
            const double dd = ok ? d : 1.0;
            double y = __builtin_amdgcn_rcp(dd);
            y = __builtin_fma(y, __builtin_fma(-dd, y, 1.0), y);
            y = __builtin_fma(y, __builtin_fma(-dd, y, 1.0), y);
            const double dinv = ok ? y : 0.0;
            {
                const double t = ri[pb] * dinv;
#pragma unroll
                for (int b = pb; b < 4; ++b) s[pb][b] -= t * rj[b];
            }
            if (ty == pq + 1) {                                   // publish row p + 1 (same register slot)
#pragma unroll
                for (int b = pb; b < 4; ++b) T[(p + 1) * STR + tx + 16 * b] = s[pb][b];
            }
#pragma unroll
            for (int a = pb + 1; a < 4; ++a) {
                const double t = ri[a] * dinv;
#pragma unroll
                for (int b = pb; b < 4; ++b) s[a][b] -= t * rj[b];
            }
        }
        if (pb < 3 && ty == 0) {                                  // first row of the next block of 16
#pragma unroll
            for (int b = pb + 1; b < 4; ++b) T[16 * (pb + 1) * STR + tx + 16 * b] = s[pb + 1][b];
        }
    }
    __syncthreads();
    CHOL_STAMP(2);
    if (tid < 64) {
        const double d = T[tid * STR + tid];
        const bool ok = d > 0.0 && d < 1.7976931348623157e308;
        const double dd = ok ? d : 1.0;
        double y = __builtin_amdgcn_rsq(dd);
        y = y * (1.5 - 0.5 * dd * y * y);
        y = y * (1.5 - 0.5 * dd * y * y);
        rinv[tid] = ok ? y : 0.0;
    }
    if (tid == 0) *sh_fail = (fail != 0 && fail <= nb) ? fail : 0;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = ty + 16 * a, j = tx + 16 * b;
            if (j >= i) {
                const double r = rinv[i];
                const double v = T[i * STR + j];
                T[i * STR + j] = (j == i) ? (r != 0.0 ? v * r : (SEMIDEF ? 0.0 : 1.0)) : v * r;
            }
        }
    __syncthreads();
    CHOL_STAMP(3);
}

// Waves of the workgroup that do NOT take part in chol64_rows_s (which needs exactly 256 threads) execute this instead:
// the same number of workgroup barriers (one per pivot, 16 ceil(nb/16), plus the three trailing ones) -- s_barrier counts
// waves, not program counters.  A helper that has work to do between the barriers replaces this loop with its own.
template <int STR>
__device__ __forceinline__ void chol64_helper_idle(int nb) {
#pragma unroll 1
    for (int pb = 0; pb < 4; ++pb) {
        if (16 * pb >= nb) break;
#pragma unroll 1
        for (int pq = 0; pq < 16; ++pq) __syncthreads();
    }
    __syncthreads();
    __syncthreads();
    __syncthreads();
}

// ---- shared pieces of the 2 x 2-block Cholesky of an n x n matrix, 64 < n <= 128, held in LDS as M[128][MS] -------------
// (k_chol128 of the factor path, k_bam_chol_out of BaM; both run eight waves: waves 0-3 call chol64_rows_s.)
// Helper waves (threads 256..511) while waves 0-3 factor A11 = M[0:64][0:64]: the block row R12 = R11^-T A12 ONE PIVOT
// BEHIND the factorisation.  Substitution step p needs row p of the factor and its pivot only; chol64_rows_s has published
// both (unscaled) by the barrier that opens pivot p.  One workgroup barrier per step + the three trailing ones = the barrier
// count of chol64_rows_s with nb = 64.  Column colq of A12 per quad of lanes; lane q owns the row pairs {8r + 2q, 8r + 2q + 1}
// (ds_read_b128 of the published row); pivot broadcast by DPP.  A dropped / failed pivot (d <= 0, NaN) contributes a zero row,
// as the sequential solve did through rinv[p] = 0.
template <int MS>
__device__ __forceinline__ void chol128_helper_rowsolve(double* M) {
    const int st = threadIdx.x - 256, colq = st >> 2, q = st & 3;
    double x[16];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        x[2 * r] = M[(8 * r + 2 * q) * MS + 64 + colq];
        x[2 * r + 1] = M[(8 * r + 2 * q + 1) * MS + 64 + colq];
    }
    double ri_prev = 0.0;
#pragma unroll
    for (int p = 0; p <= 64; ++p) {
        double ri = 0.0;
        if (p < 64) {
            __syncthreads();                                      // opens pivot p: row p and d_p are final
            const double d = M[p * MS + p];
            const bool ok = d > 0.0 && d < 1.7976931348623157e308;
            const double dd = ok ? d : 1.0;
            double y = __builtin_amdgcn_rsq(dd);
            y = y * (1.5 - 0.5 * dd * y * y);
            y = y * (1.5 - 0.5 * dd * y * y);
            ri = ok ? y : 0.0;
        }
        if (p > 0) {
            const int ps = p - 1;
            const int pr = 2 * (ps >> 3) + (ps & 1), pq = (ps >> 1) & 3;
            const double mine = x[pr] * ri_prev;
            if (q == pq) x[pr] = mine;
            const double xp = quad_bcast_rt<0>(mine, pq) * ri_prev;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (8 * r + 7 > ps) {
                    const int t = 8 * r + 2 * q;
                    const v2d rv = *reinterpret_cast<const v2d*>(&M[ps * MS + t]);
                    x[2 * r] -= (t > ps) ? rv.x * xp : 0.0;
                    x[2 * r + 1] -= (t + 1 > ps) ? rv.y * xp : 0.0;
                }
        }
        ri_prev = ri;
    }
    __syncthreads();
    __syncthreads();
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {                                 // every helper is past its last read of the A12 block
        M[(8 * r + 2 * q) * MS + 64 + colq] = x[2 * r];
        M[(8 * r + 2 * q + 1) * MS + 64 + colq] = x[2 * r + 1];
    }
}

// A22 -= R12^T R12 on the MFMA pipe, all eight waves: the 10 upper 16 x 16 blocks (bi <= bj), operands straight from M
// (A[k][i] = M[k][64 + i]: the lanes of one k-slot read 16 consecutive doubles).  Call between two workgroup barriers.
template <int MS>
__device__ __forceinline__ void chol128_rank64_update(double* M) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 15, ks = l >> 4;
    for (int blk = w; blk < 10; blk += 8) {
        int bi = 0, rem = blk;
        while (rem >= 4 - bi) { rem -= 4 - bi; ++bi; }
        const int bj = bi + rem;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        const double* ap = M + ks * MS + 64 + 16 * bi + c;
        const double* bp = M + ks * MS + 64 + 16 * bj + c;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = GSMVI_MFMA_F64(ap[4 * s * MS], bp[4 * s * MS], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * bi + ks + 4 * r, j = 16 * bj + c;
            if (j >= i) M[(64 + i) * MS + 64 + j] -= acc[r];
        }
    }
}
