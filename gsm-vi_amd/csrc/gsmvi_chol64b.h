// Blocked in-LDS Cholesky of a 64 x 64 block WITHOUT a workgroup barrier on the pivot chain (round 3).
//
// The barrier-per-pivot factorisation this replaced (chol64_rows_s, deleted in round 4) paid one LDS round trip and one s_barrier per pivot: ~190 ns x 64 pivots = 12-15 us for
// one 64 x 64 factorisation on one CU.  Here the matrix is processed in four block rows of 16:
//   panel    : the 16 x (columns to the right) block row is held a column per lane, and its 16 pivots run in registers: the
//              pivot and the next row's multiplier are broadcast with v_readlane, the other multipliers come back from an
//              LDS copy of the pivot row as broadcast reads (no barrier: a wave's LDS operations execute in order); the
//              elimination is the unscaled (LDL^T) form, so the reciprocal 1/d_p (v_rcp_f64 + two Newton steps) is the only
//              long operation on the pivot chain, and the rows are scaled by 1/sqrt(d_p) at the end.  Because the lanes to
//              the right of the diagonal block are eliminated with it, the block row comes out SOLVED (R_kj = R_kk^-T A_kj):
//              there is no separate triangular solve.
//              One wave running all 16 pivots is ISSUE-bound (~45 instructions = ~290 cycles per pivot, 1.9 us per panel),
//              so the 16 rows are split between TWO waves on two SIMDs: the wave that owns rows 0-7 runs pivots 0-7 and
//              publishes each pivot row (unscaled, and scaled by 1/d_p) with a counter in LDS; the wave that owns rows 8-15
//              applies them to its rows as they appear (it spins on the counter: both waves are resident, the first never
//              waits for the second) and then runs pivots 8-15.  ~25 instructions per pivot on the chain wave.
//   trailing : the rank-16 updates A_ij -= R_ki^T R_kj of the remaining block rows on the MFMA pipe, all eight waves.
// Two workgroup barriers per block step (8 in all instead of 64 + 3).
// AUG = 1 factors the augmented matrix [A | I]: the same row operations turn the identity into W = R^-T (lower
// triangular, E[:, 64:128]), which the callers need as an explicit matrix (the triangular solves of the factor path and of
// the blocked D x D Cholesky are MFMA products with W).  The augmented columns ride in the lanes of a second pair of panel
// waves that repeats the diagonal block's arithmetic (bit-identical multipliers), and in the same trailing products.
// AUG = 2: the right half E[:, 64:128] holds CALLER DATA C (dense, 64 columns) and comes out as R^-T C -- the block-row
// solve of a larger blocked factorisation (k_chol128: [A11 | A12] -> [R11 | R12]); three pairs of panel waves.
//
// Layout: E[64][ES] in LDS, row-major; columns 0..63 = A (upper triangle + diagonal valid on entry; everything strictly
// below the diagonal BLOCKS must be zero on entry if the caller wants a clean upper factor; padded with the identity
// beyond nb), columns 64..127 = W (AUG; initialised here).  ES % 4 == 2 and (2 ES) % 64 == 36 (ES = 146, 82) make both
// operand shapes of the MFMA phases bank-conflict free.
// Semantics: *sh_fail = 1-based index of the first bad pivot (<= 0, NaN, inf) or 0; SEMIDEF (the caller has
// lowered the diagonal by its rounding floor, see gsmvi_chol64.h) turns a pivot <= 0 into a DROPPED row -- zero row and
// zero diagonal in R, unit pivot in W -- unless allow_dep is false.
// Needs 512 threads (eight waves); every thread of the workgroup must call it.
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

// LDS scratch of chol64_blk, per column set (AUG = 0 / 1 / 2: one / two / three sets): [16][64] unscaled pivot rows, [8][64]
// pivot rows scaled by 1/d_p (pivots 0-7, for the wave that owns rows 8-15), [2][8] row scales, the publish counter
#define CHOLB_SCRATCH_PER_SET (16 * 64 + 8 * 64 + 16 + 2)
#define CHOLB_SCRATCH_DOUBLES(AUGV) (((int)(AUGV) + 1) * CHOLB_SCRATCH_PER_SET)

#ifndef CHOLB_STAMP
#define CHOLB_STAMP(k) ((void)0)
#endif
#ifndef CHOLB_PSTAMP
#define CHOLB_PSTAMP(h, i) ((void)0)       // scripts/chol64b_test.hip: per-pivot stamps inside a panel
#endif

// One wave's half (rows 8H .. 8H+7) of a 16-row panel.  col: this lane's column of E; lanes 0-15 of every panel wave hold the
// diagonal block's columns.  write_back: false for inactive lanes and for the replica of the diagonal block in the second
// column set.  pub: value the publish counter has when this panel starts (monotone over the block steps).
template <int ES, bool SEMIDEF, int H>
__device__ __forceinline__ void cholb_panel_half(double* E, int k0, int col, bool write_back, bool aug, bool diag_lane,
                                                 double* ur, double* tr, double* rsb, volatile int* cnt, int pub,
                                                 bool allow_dep, int* sh_failmin, double (&vkeep)[8]) {
    const int l = threadIdx.x & 63, c = l & 15;
    constexpr int R0 = 8 * H;
    double v[8];
    CHOLB_PSTAMP(H, 0);
    {
        const double* ep = E + (k0 + R0) * ES + col;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ep[i * ES];
    }
#ifdef CHOLB_TEST_CORRUPT_REPLICA
    // test hook: the harness must notice a replica that does NOT hold the diagonal block (validates the test itself)
    if (diag_lane && !write_back) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += 0.25;
    }
#endif
    // Every LDS read below is issued ONE PIVOT before its values are used (explicit software pipelining): a wave issues in
    // order, so an s_waitcnt in front of an elimination FMA also holds back the reciprocal chain of the next pivot behind
    // it -- with the reads issued at their point of use the LDS round trip (~130 cycles) sat on the chain of every pivot.
    if (H == 1) {
        // pivots 0-7 belong to the other wave: apply each one as soon as it is published
        double tq;
        v2d uq[4];
        auto fetch = [&](int p) {
            while (__builtin_amdgcn_readfirstlane(*cnt) < pub + p + 1) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            tq = tr[p * 64 + l];
#pragma unroll
            for (int j = 0; j < 4; ++j) uq[j] = *reinterpret_cast<const v2d*>(&ur[p * 64 + 8 + 2 * j]);
        };
        fetch(0);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const double t = tq;
            v2d uu[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) uu[j] = uq[j];
            if (p + 1 < 8) fetch(p + 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[2 * j] = __builtin_fma(-uu[j].x, t, v[2 * j]);
                v[2 * j + 1] = __builtin_fma(-uu[j].y, t, v[2 * j + 1]);
            }
        }
    }
    CHOLB_PSTAMP(H, 1);
    int fail = 0;
    double t_prev = 0.0;
    v2d u_prev[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) u_prev[j] = (v2d){0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int p = R0 + q;
        CHOLB_PSTAMP(H, 2 + q);
        ur[p * 64 + l] = v[q];                                    // the unscaled pivot row, final
        const double d = readlane_f64(v[q], p);                   // lane p of this wave = column p of the diagonal block
        // the reciprocal starts from the broadcast pivot itself (no test in front of it on the chain).  A dropped pivot
        // (SEMIDEF: d <= 0) must eliminate nothing; a FAILED pivot makes the whole result irrelevant, so outside SEMIDEF
        // nothing is masked.  The pass/fail verdicts are taken from the saved diagonal after the loop, all at once: per pivot
        // they cost six instructions of a wave that issues one every ~6 cycles.
        const double dinv = rcp_nr2(d);
        const double t = (!SEMIDEF || d > 0.0) ? v[q] * dinv : 0.0;
        if (H == 0) {
            tr[p * 64 + l] = t;
            asm volatile("" ::: "memory");                        // program order = LDS order inside one wave
            *cnt = pub + q + 1;
        }
        // rows >= q + 1: the PREVIOUS pivot's update (its multipliers were read from LDS one pivot ago)
        if (q >= 1) {
#pragma unroll
            for (int i = q + 1; i < 8; ++i) {
                const double uv = (i & 1) ? u_prev[i >> 1].y : u_prev[i >> 1].x;
                v[i] = __builtin_fma(-uv, t_prev, v[i]);
            }
        }
        if (q + 1 < 8) {                                          // the next pivot's row: this pivot's update, multiplier by v_readlane
            const double u = readlane_f64(v[q], p + 1);
            v[q + 1] = __builtin_fma(-u, t, v[q + 1]);
        }
        // this pivot's multipliers for the rows >= q + 2, used during the next pivot
#pragma unroll
        for (int j = (q + 2) >> 1; j < 4; ++j) u_prev[j] = *reinterpret_cast<const v2d*>(&ur[p * 64 + R0 + 2 * j]);
        t_prev = t;
    }
    CHOLB_PSTAMP(H, 10);
    {   // row scales 1/sqrt(d_p) of this wave's eight rows at once: lane c owns pivot R0 + (c & 7).  Dropped row: zero in R,
        // unit pivot in W.
        const int pp = R0 + (c & 7);
        const double dg = ur[pp * 64 + pp];
        const bool ok = dg > 0.0 && dg < 1.7976931348623157e308;    // false for NaN, <= 0, inf
        const bool dep = SEMIDEF && allow_dep && dg <= 0.0 && dg > -1.7976931348623157e308;
        const unsigned long long badm = __builtin_amdgcn_ballot_w64(!ok && !dep) & 0xffull;   // lanes 0-7: pivots R0 .. R0+7
        if (badm != 0) fail = k0 + R0 + __builtin_ctzll(badm) + 1;
        const double rs = ok ? rsq_nr2(ok ? dg : 1.0) : 0.0;
        rsb[8 * H + (c & 7)] = rs;
#pragma unroll
        for (int p2 = 0; p2 < 8; p2 += 2) {
            const v2d r2 = *reinterpret_cast<const v2d*>(&rsb[8 * H + p2]);
            v[p2] *= (aug && r2.x == 0.0) ? 1.0 : r2.x;
            v[p2 + 1] *= (aug && r2.y == 0.0) ? 1.0 : r2.y;
        }
    }
    CHOLB_PSTAMP(H, 11);
    // The DIAGONAL block is not written back here: the other column sets hold replicas of it, loaded from E when their waves
    // start this panel, and nothing orders those loads against this wave's end.  chol64_blk writes vkeep back behind the
    // workgroup barrier that follows the panel phase.  (Round 3: a deviation seen about once in 3e5 factor updates inside a
    // busy pipeline, never with the kernel alone on the chip, has not been seen since this change -- DESIGN section 8.)
#ifdef CHOLB_TEST_OLD_WRITEBACK
    // test builds only (scripts/chol64b_test.hip, `make oldwb`): the pre-fix behaviour, set 0 writes its factored diagonal rows
    // back IN PLACE at the end of its half-panel -- the write the replicas' loads were not ordered against
    if (write_back) {
        double* ep = E + (k0 + R0) * ES + col;
#pragma unroll
        for (int i = 0; i < 8; ++i) ep[i * ES] = (diag_lane && R0 + i > c) ? 0.0 : v[i];
    }
#else
    if (write_back && !diag_lane) {
        double* ep = E + (k0 + R0) * ES + col;
#pragma unroll
        for (int i = 0; i < 8; ++i) ep[i * ES] = v[i];
    }
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i) vkeep[i] = v[i];
    if (fail != 0 && l == 0) atomicMin(sh_failmin, fail);
    CHOLB_PSTAMP(H, 12);
}

// k_first > 0 (round 5): the first k_first 16-row panels are GIVEN -- the caller has put the finished rows [R | W] of a
// BLOCK-DIAGONAL matrix there (no coupling between the given rows and the rest: R12 = 0, W21 = 0) and the factorisation resumes
// at panel k_first on the untouched trailing block; the augmented columns of the given rows are the caller's.
// LA (round 6, the persistent Cholesky's chain only): look-ahead inside the block.  Behind panel k only block row k + 1 -- what
// panel k + 1 needs -- is updated by all eight waves; the rows beyond it are updated by the waves the NEXT panel leaves idle,
// beside that panel (it touches block row k + 1 and the scratch only).  Same products, same operands, same order per element.
template <int ES, bool SEMIDEF, int AUG, bool LA = false>
__device__ __forceinline__ void chol64_blk(double* E, double* scratch, int nb, int* sh_fail, bool allow_dep = true,
                                           int k_first = 0) {
    static_assert(ES % 2 == 0, "rows must stay 16-byte aligned");
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, g = l >> 4;
    const int nblk = (nb + 15) >> 4;
    if (AUG == 1) {
        for (int e = tid; e < 64 * 64; e += 512) {
            const int i = e >> 6, q = e & 63;
            if (i >= 16 * k_first) E[i * ES + 64 + q] = (i == q) ? 1.0 : 0.0;
        }
    }
    if (tid == 0) *sh_fail = 0x7fffffff;
    if (tid < AUG + 1)                                            // (the publish counter is monotone over the panels: 8 per panel)
        *reinterpret_cast<volatile int*>(scratch + tid * CHOLB_SCRATCH_PER_SET + 16 * 64 + 8 * 64 + 16) = 8 * k_first;
#ifdef CHOLB_TEST_FORCE_ORDER
    if (tid == 0) *reinterpret_cast<volatile int*>(scratch + 16 * 64 + 8 * 64 + 17) = 0;
#endif
    __syncthreads();
    // products of step ks for the block rows [i_lo, i_hi), dealt to `nwv` waves of which this is number `wv` (wave-uniform walk
    // over the flat product list)
    auto trailing = [&](int ks, int i_lo, int i_hi, int wv, int nwv) {
        const int kk0 = 16 * ks;
        const int nWs = (AUG == 1) ? ks + 1 : (AUG == 2 ? 4 : 0);
        int idx = wv;
        for (int i = i_lo; i < i_hi && i < nblk; ++i) {
            const int cnt = (nblk - i) + nWs;
            while (idx < cnt) {
                const int gcol = (idx < nblk - i) ? 16 * (i + idx) : 64 + 16 * (idx - (nblk - i));
                const double* ap = E + (kk0 + g) * ES + 16 * i + c;
                const double* bp = E + (kk0 + g) * ES + gcol + c;
                double a[4], b[4], tv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) { a[s] = ap[4 * s * ES]; b[s] = bp[4 * s * ES]; }
                double* tp = E + (16 * i + g) * ES + gcol + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) tv[r] = tp[4 * r * ES];
                v4d acc = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};   // two chains of two
                acc = GSMVI_MFMA_F64(a[0], b[0], acc);
                acc1 = GSMVI_MFMA_F64(a[1], b[1], acc1);
                acc = GSMVI_MFMA_F64(a[2], b[2], acc);
                acc1 = GSMVI_MFMA_F64(a[3], b[3], acc1);
#pragma unroll
                for (int r = 0; r < 4; ++r) tp[4 * r * ES] = tv[r] - (acc[r] + acc1[r]);
                idx += nwv;
            }
            idx -= cnt;
        }
    };
#pragma unroll 1
    for (int k = k_first; k < nblk; ++k) {                        // block-uniform
        const int k0 = 16 * k;
        CHOLB_STAMP(1 + 2 * k);
        // ---- panel: block row k, columns to the right of (and including) the diagonal block, plus the right-half columns
        // (W-groups 0 .. k for AUG = 1: the identity's fill-in; all four groups for AUG = 2).  Column groups of 16 in this
        // order: A-groups k .. nblk-1, then the right-half groups.  Column set s (waves 2s, 2s+1) holds the diagonal block in
        // lanes 0-15 (set 0 the original, the others bit-identical replicas) and the groups 3s+1 .. 3s+3 in lanes 16-63.
        // Even wave of a set: rows 0-7, odd wave: rows 8-15.
        const int nW = (AUG == 1) ? k + 1 : (AUG == 2 ? 4 : 0);
        double vk[8];
        if (w < 2 * (AUG + 1)) {
            const int set = w >> 1;
            const int nA = nblk - k, nslots = nA + nW;
            const int slot = (g == 0) ? 0 : 3 * set + g;
            const bool active = slot < nslots;
            const bool aug = slot >= nA;
            const int col = (active ? (aug ? 64 + 16 * (slot - nA) : 16 * (k + slot)) : k0) + c;
            double* ur = scratch + set * CHOLB_SCRATCH_PER_SET;
            double* tr = ur + 16 * 64;
            double* rsb = tr + 8 * 64;
            volatile int* cnt = reinterpret_cast<volatile int*>(rsb + 16);
            const bool wb = active && !(set > 0 && g == 0);
#ifdef CHOLB_TEST_REPLICA_DELAY
            // test hook (scripts/chol64b_test.hip, tests/test_gpu_aux.py): hold the replica sets' waves back for several
            // microseconds, the schedule under which an in-place write-back of the diagonal block by set 0 would be loaded
            // (the branch must be SCALAR: s_sleep is a scalar instruction and ignores the exec mask -- with `if (set > 0)` on
            // a per-lane value the compiler only masked lanes and EVERY panel wave slept, which is why round 3's version of
            // this hook could not reproduce anything)
            if (__builtin_amdgcn_readfirstlane(set) > 0)
                for (int d_ = 0; d_ < CHOLB_TEST_REPLICA_DELAY; ++d_) __builtin_amdgcn_s_sleep(127);
            asm volatile("s_nop 0" ::: "memory");     // the panel's LDS loads must not be scheduled in front of the delay
#endif
#ifdef CHOLB_TEST_FORCE_ORDER
            // test hook: the replica sets' waves load their copy of the diagonal block only AFTER both waves of set 0 have
            // finished this panel (a counter in the spare word of set 0's scratch) -- the latest legal schedule, forced.  With
            // CHOLB_TEST_OLD_WRITEBACK this is exactly the interleaving the round-3 fix removes: the replicas load factored rows.
            volatile int* order_flag = reinterpret_cast<volatile int*>(scratch + 16 * 64 + 8 * 64 + 17);
            if (__builtin_amdgcn_readfirstlane(set) > 0) {
                while (__builtin_amdgcn_readfirstlane(*order_flag) < 2 * (k + 1)) __builtin_amdgcn_s_sleep(1);
                asm volatile("s_nop 0" ::: "memory");
            }
#endif
            if ((w & 1) == 0)
                cholb_panel_half<ES, SEMIDEF, 0>(E, k0, col, wb, aug && AUG == 1, slot == 0, ur, tr, rsb, cnt, 8 * k, allow_dep, sh_fail, vk);
            else
                cholb_panel_half<ES, SEMIDEF, 1>(E, k0, col, wb, aug && AUG == 1, slot == 0, ur, tr, rsb, cnt, 8 * k, allow_dep, sh_fail, vk);
#ifdef CHOLB_TEST_FORCE_ORDER
            asm volatile("" ::: "memory");                        // a wave's LDS operations execute in program order
            if (set == 0 && l == 0) atomicAdd(const_cast<int*>(order_flag), 1);
#endif
        } else if (LA && k > k_first) {
            // the rows beyond this panel's, step k - 1: on the waves the panel leaves idle
            constexpr int NPW = 2 * (AUG + 1);
            trailing(k - 1, k + 1, nblk, w - NPW, 8 - NPW);
        }
        __syncthreads();
#ifndef CHOLB_TEST_OLD_WRITEBACK
        if (w < 2 && g == 0) {                                    // set 0's diagonal-block lanes: the factored block, strictly-lower part zeroed
            double* ep = E + (k0 + 8 * w) * ES + k0 + c;
#pragma unroll
            for (int i = 0; i < 8; ++i) ep[i * ES] = (8 * w + i > c) ? 0.0 : vk[i];
        }
#endif
        CHOLB_STAMP(2 + 2 * k);
        if (k + 1 >= nblk) break;                                 // block-uniform: no trailing matrix
        // ---- trailing: for every remaining block row i and every column group j (A-groups i .. nblk-1, W-groups 0 .. k):
        //   E[rows of i][group j] -= R_ki^T E[rows of k][group j],   K = 16 = four fp64 MFMA 16x16x4
        // (LA: block row k + 1 here, the rows beyond it beside the next panel -- see the top of the loop)
        trailing(k, k + 1, LA ? k + 2 : nblk, w, 8);
        __syncthreads();
    }
    if (tid == 0) {                                               // failures beyond nb are the identity padding: none
        const int f = *sh_fail;
        *sh_fail = (f == 0x7fffffff || f > nb) ? 0 : f;
    }
    __syncthreads();
}
