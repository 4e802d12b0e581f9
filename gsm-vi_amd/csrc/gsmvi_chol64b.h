// Blocked in-LDS Cholesky of a 64 x 64 block WITHOUT a workgroup barrier on the pivot chain (round 3).
//
// chol64_rows_s (gsmvi_chol64.h) pays one LDS round trip and one s_barrier per pivot: ~190 ns x 64 pivots = 12-15 us for
// one 64 x 64 factorisation on one CU.  Here the matrix is processed in four block rows of 16:
//   panel    : ONE wave holds the 16 x (columns to the right) block row, a column per lane (16 registers), and runs the 16
//              pivots in registers: the pivot and the next row's multiplier are broadcast with v_readlane, the other
//              multipliers come back from a wave-private LDS row as broadcast reads (no barrier: a wave's LDS operations
//              execute in order); the elimination is the unscaled (LDL^T) form, so the reciprocal 1/d_p (v_rcp_f64 + two
//              Newton steps) is the only long operation on the pivot chain, and the rows are scaled by 1/sqrt(d_p) at the end.
//              Because the lanes to the right of the diagonal block are eliminated with it, the block row comes out SOLVED
//              (R_kj = R_kk^-T A_kj): there is no separate triangular solve.
//   trailing : the rank-16 updates A_ij -= R_ki^T R_kj of the remaining block rows on the MFMA pipe, all eight waves.
// Two workgroup barriers per block step (8 in all instead of 64 + 3).
// AUG = true factors the augmented matrix [A | I]: the same row operations turn the identity into W = R^-T (lower
// triangular, E[:, 64:128]), which the callers need as an explicit matrix (the triangular solves of the factor path and of
// the blocked D x D Cholesky are MFMA products with W).  The augmented columns ride in the lanes of a second panel wave
// that repeats the diagonal block's arithmetic (bit-identical multipliers), and in the same trailing products.
//
// Layout: E[64][ES] in LDS, row-major; columns 0..63 = A (upper triangle + diagonal valid on entry; everything strictly
// below the diagonal BLOCKS must be zero on entry if the caller wants a clean upper factor; padded with the identity
// beyond nb), columns 64..127 = W (AUG; initialised here).  ES % 4 == 2 and (2 ES) % 64 == 36 (ES = 146, 82) make both
// operand shapes of the MFMA phases bank-conflict free.
// Semantics = chol64_rows_s: *sh_fail = 1-based index of the first bad pivot (<= 0, NaN, inf) or 0; SEMIDEF (the caller has
// lowered the diagonal by its rounding floor, see gsmvi_chol64.h) turns a pivot <= 0 into a DROPPED row -- zero row and
// zero diagonal in R, unit pivot in W -- unless allow_dep is false.
#pragma once
#include "gsmvi_common.h"
#include "gsmvi_chol64.h"

__device__ __forceinline__ double rcp_nr2(double d) {             // 1/d to fp64 accuracy: v_rcp_f64 (~2^-23) + 2 Newton steps
    double y = __builtin_amdgcn_rcp(d);
    y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
    y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
    return y;
}
__device__ __forceinline__ double rsq_nr2(double d) {             // 1/sqrt(d)
    double y = __builtin_amdgcn_rsq(d);
    y = y * (1.5 - 0.5 * d * y * y);
    y = y * (1.5 - 0.5 * d * y * y);
    return y;
}

// LDS scratch of chol64_blk: per panel wave the 16 x 64 unscaled pivot rows and 16 row scales (AUG: two panel waves)
#define CHOLB_SCRATCH_PER_WAVE (16 * 64 + 16)
#define CHOLB_SCRATCH_DOUBLES(AUGV) (((AUGV) ? 2 : 1) * CHOLB_SCRATCH_PER_WAVE)

#ifndef CHOLB_STAMP
#define CHOLB_STAMP(k) ((void)0)
#endif

template <int ES, bool SEMIDEF, bool AUG>
__device__ __forceinline__ void chol64_blk(double* E, double* scratch, int nb, int* sh_fail, bool allow_dep = true) {
    static_assert(ES % 2 == 0, "rows must stay 16-byte aligned");
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, g = l >> 4;
    const int nblk = (nb + 15) >> 4;
    if (AUG) {
        for (int e = tid; e < 64 * 64; e += 512) {
            const int i = e >> 6, q = e & 63;
            E[i * ES + 64 + q] = (i == q) ? 1.0 : 0.0;
        }
    }
    if (tid == 0) *sh_fail = 0;
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < nblk; ++k) {                              // block-uniform
        const int k0 = 16 * k;
        CHOLB_STAMP(1 + 2 * k);
        // ---- panel: block row k, columns to the right of (and including) the diagonal block, plus the W columns 0 .. 16k+15.
        // Column groups of 16 in this order: A-groups k .. nblk-1, then W-groups 0 .. k.  Wave 0 takes the first four (its
        // lanes 0-15 = the diagonal block), wave 1 (AUG) repeats the diagonal block in lanes 0-15 and takes groups 4 .. 6.
        if (w < (AUG ? 2 : 1)) {
            const int nA = nblk - k, nslots = nA + (AUG ? k + 1 : 0);
            const int slot = (w == 0) ? g : (g == 0 ? 0 : 3 + g);
            const bool active = slot < nslots;
            const bool aug = slot >= nA;
            const int col = (active ? (aug ? 64 + 16 * (slot - nA) : 16 * (k + slot)) : k0) + c;
            double v[16];
            {
                const double* ep = E + k0 * ES + col;
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = ep[i * ES];
            }
            // The UNSCALED pivot row goes to LDS the moment it is final (before its reciprocal is known): the multipliers of
            // the elimination are read back from there as lane-uniform ds_read_b128 broadcasts, two per instruction.  (A
            // v_readlane pair per multiplier made the loop issue-bound -- 30 SGPRs per pivot, spilled through v_writelane --
            // 3.0 us per 16 pivots; this form 1 us.)  The rank-1 update is S[i][j] -= S[p][i] (S[p][j] / d_p): the 1/d_p rides
            // on the lane's own entry, so only the next pivot's row waits for the reciprocal.
            double* ur = scratch + w * CHOLB_SCRATCH_PER_WAVE;    // [16][64] unscaled rows, then [16] row scales
            double* rsb = ur + 16 * 64;
            int fail = 0;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                ur[p * 64 + l] = v[p];
                const double d = readlane_f64(v[p], p);           // lane p of this wave = column p of the diagonal block
                const bool ok = d > 0.0 && d < 1.7976931348623157e308;   // false for NaN, <= 0, inf
                const bool dep = SEMIDEF && allow_dep && d <= 0.0 && d > -1.7976931348623157e308;
                if (!ok && !dep && fail == 0) fail = k0 + p + 1;
                // the reciprocal starts from the broadcast pivot itself (no test in front of it on the chain).  A dropped
                // pivot (SEMIDEF) must eliminate nothing; a FAILED pivot makes the whole result irrelevant, so outside
                // SEMIDEF nothing is masked
                const double dinv = rcp_nr2(d);
                const double t = (!SEMIDEF || ok) ? v[p] * dinv : 0.0;
                if (p + 1 < 16) {                                 // the next pivot's row first, its multiplier by v_readlane
                    const double u = readlane_f64(v[p], p + 1);
                    v[p + 1] = __builtin_fma(-u, t, v[p + 1]);
                }
#pragma unroll
                for (int i2 = (p + 2) & ~1; i2 < 16; i2 += 2) {
                    const v2d uu = *reinterpret_cast<const v2d*>(&ur[p * 64 + i2]);
                    if (i2 >= p + 2) v[i2] = __builtin_fma(-uu.x, t, v[i2]);
                    v[i2 + 1] = __builtin_fma(-uu.y, t, v[i2 + 1]);
                }
            }
            {   // row scales 1/sqrt(d_p), all sixteen at once: lane c owns pivot c.  Dropped row: zero in R, unit pivot in W.
                const double dg = ur[c * 64 + c];
                const bool ok = dg > 0.0 && dg < 1.7976931348623157e308;
                const double rs = ok ? rsq_nr2(ok ? dg : 1.0) : 0.0;
                rsb[c] = rs;
#pragma unroll
                for (int p2 = 0; p2 < 16; p2 += 2) {
                    const v2d r2 = *reinterpret_cast<const v2d*>(&rsb[p2]);
                    v[p2] *= (aug && r2.x == 0.0) ? 1.0 : r2.x;
                    v[p2 + 1] *= (aug && r2.y == 0.0) ? 1.0 : r2.y;
                }
            }
            if (active && !(w == 1 && g == 0)) {
                double* ep = E + k0 * ES + col;
                const bool dg = (slot == 0);                      // the diagonal block: its strictly-lower part is zeroed
#pragma unroll
                for (int i = 0; i < 16; ++i) ep[i * ES] = (dg && i > c) ? 0.0 : v[i];
            }
            if (tid == 0 && fail != 0 && fail <= nb && *sh_fail == 0) *sh_fail = fail;
        }
        __syncthreads();
        CHOLB_STAMP(2 + 2 * k);
        if (k + 1 >= nblk) break;                                 // block-uniform: no trailing matrix
        // ---- trailing: for every remaining block row i and every column group j (A-groups i .. nblk-1, W-groups 0 .. k):
        //   E[rows of i][group j] -= R_ki^T E[rows of k][group j],   K = 16 = four fp64 MFMA 16x16x4
        {
            const int nW = AUG ? k + 1 : 0;
            int idx = w;
            for (int i = k + 1; i < nblk; ++i) {                  // wave-uniform walk over the flat product list
                const int cnt = (nblk - i) + nW;
                while (idx < cnt) {
                    const int gcol = (idx < nblk - i) ? 16 * (i + idx) : 64 + 16 * (idx - (nblk - i));
                    const double* ap = E + (k0 + g) * ES + 16 * i + c;
                    const double* bp = E + (k0 + g) * ES + gcol + c;
                    double a[4], b[4], tv[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) { a[s] = ap[4 * s * ES]; b[s] = bp[4 * s * ES]; }
                    double* tp = E + (16 * i + g) * ES + gcol + c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) tv[r] = tp[4 * r * ES];
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = GSMVI_MFMA_F64(a[s], b[s], acc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) tp[4 * r * ES] = tv[r] - acc[r];
                    idx += 8;
                }
                idx -= cnt;
            }
        }
        __syncthreads();
    }
}
