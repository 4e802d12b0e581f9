// The one-workgroup 128 x 128 Cholesky WITH the inverse factor (64 < n <= 128), shared by the factor path's 2B x 2B chain
// (k_chol128w, gsmvi_factor.hip) and the dense BaM chain (k_bam_cholw, gsmvi_bam_small.hip).  512 threads.
#pragma once
#include "gsmvi_common.h"
#include "gsmvi_chol64.h"
#include "gsmvi_chol64b.h"

// ---- the same factorisation WITH the inverse factor: A = R^T R and W = R^-T (lower triangular), 64 < n <= 128 -------------
// The Gram matrix of the factor path needs W = Rg^-T as an explicit matrix (K'' = (W S)^T (T - I)(W S)).  Round 2 got it from a
// 128-step substitution in a launch of its own (k_gsmf_kmat_big: 25 us).  Here both diagonal blocks are factored as
// [A_kk | I] -> [R_kk | W_kk] (chol64_blk, AUG = 1: +1.8 us each over the plain factorisation) and the off-diagonal blocks are
// MFMA products of things already in LDS:
//   R12 = D' W11 A12           (the row operations that turned I into W11, applied to A12; D' zeroes the rows the
//                               rank-revealing rule dropped, which is what the riding columns of AUG = 2 would hold)
//   A22' = A22 - R12^T R12,    [A22' | I] -> [R22 | W22]
//   W21 = -(W22 R12^T) W11     (block inverse of a triangular matrix)
// One workgroup, E1 [64][146] is used for both factorisations (R11, W11 go to global memory in between; W11 also stays in
// registers, 8 values per thread, for the last product), B12 [64][66] holds R12, then W11.  Same dropped-row convention as k_gsmf_kmat_big's unit pivots.
// (A device function since round 4: k_chol128w of the factor path and k_bam_cholw of the dense BaM chain share it.  A, R, Wo
// must not alias; A is read until the second factorisation starts.  sh_info: optional LDS word that receives the same value as
// *info -- valid for the whole workgroup after the caller's next barrier.)
// Leading dimensions lda / ldr / ldw (the blocks of a larger matrix: the 128 < n <= 256 chain of gsmvi_factor.hip);
// tol_applied: the caller has already lowered A's diagonal by its rounding floor (SEMIDEF; the second diagonal block of the
// two-level scheme); moderate_ext: 0 / 1 overrides the magnitude guard taken from this block's own diagonal (-1: own).
// WT: W is stored TRANSPOSED (Wo[j ldw + i] = W[i][j], i.e. the upper triangular R^-1): what a consumer wants that reads W's
// rows as MFMA A-operands (k_bam_zw: 16 consecutive doubles per k instead of 16 cache lines per load instruction).
// LDS of one factorisation (one set per KERNEL: chol128w_core takes it as arguments, so a kernel whose workgroups run
// different instantiations -- the paired launches of gsmvi_factor.hip / gsmvi_bam_small.hip -- allocates it once).
#define CHOL128W_ES1 146
#define CHOL128W_BS 66
#define CHOL128W_LDS(E1, B12, scr, shf, shm)                                    \
    __shared__ __attribute__((aligned(16))) double E1[64 * CHOL128W_ES1];       \
    __shared__ __attribute__((aligned(16))) double B12[64 * CHOL128W_BS];       \
    __shared__ __attribute__((aligned(16))) double scr[CHOLB_SCRATCH_DOUBLES(1)]; \
    __shared__ int shf[2], shm
template <bool SEMIDEF, bool WT = false>
__device__ __forceinline__ void chol128w_core(double* E1, double* B12, double* scr, int* sh_fail, int* sh_moderate_p, int n,
                                              const double* __restrict__ A, int lda, double* __restrict__ R, int ldr,
                                              double* __restrict__ Wo, int ldw, int* __restrict__ info,
                                              int* sh_info = nullptr, bool tol_applied = false, int moderate_ext = -1) {
    constexpr int ES1 = CHOL128W_ES1, BS = CHOL128W_BS;
    auto widx = [&](int i, int j) -> size_t { return WT ? (size_t)j * ldw + i : (size_t)i * ldw + j; };   // where W[i][j] lives
    int& sh_moderate = *sh_moderate_p;
    const int tid = threadIdx.x, n2 = n - 64;
    const int w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    if (tid == 0) sh_moderate = 1;
    __syncthreads();
    {   // A11 (upper triangle) -> E1 left half; the magnitude guard looks at BOTH diagonal blocks, as k_chol128 does
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 512 * u, i = e >> 6, q = e & 63;
            v[u] = A[(size_t)i * lda + q];
        }
        const double d2 = (tid < n2) ? A[(size_t)(64 + tid) * lda + 64 + tid] : 0.0;
        if (SEMIDEF && !(d2 < 4294967296.0)) sh_moderate = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 512 * u, i = e >> 6, q = e & 63;
            double x = (q < i) ? 0.0 : v[u];
            if (SEMIDEF && q == i) {
                if (!(x < 4294967296.0)) sh_moderate = 0;
                if (!tol_applied) x -= GSMVI_DEP_TOL * x;
            }
            E1[i * ES1 + q] = x;
        }
    }
    __syncthreads();
    const bool moderate = moderate_ext < 0 ? sh_moderate != 0 : moderate_ext != 0;
    chol64_blk<ES1, SEMIDEF, 1>(E1, scr, 64, &sh_fail[0], moderate);   // [A11 | I] -> [R11 | W11]
    // R12 = D' W11 A12: block (ib, jb), k-blocks 0 .. ib (W11 is lower triangular); A12 straight from global memory
    for (int blk = w; blk < 16; blk += 8) {
        const int ib = blk >> 2, jb = blk & 3;
        const int gj = 64 + 16 * jb + c;
        double a[16], b[16];
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int k = 4 * st + ks;
            a[st] = E1[(16 * ib + c) * ES1 + 64 + k];
            b[st] = A[(size_t)k * lda + (gj < n ? gj : n - 1)];
        }
        v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int st = 0; st < 16; st += 2) {
            if (st < 4 * (ib + 1)) {                                   // wave-uniform
                acc0 = GSMVI_MFMA_F64(a[st], gj < n ? b[st] : 0.0, acc0);
                acc1 = GSMVI_MFMA_F64(a[st + 1], gj < n ? b[st + 1] : 0.0, acc1);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ib + ks + 4 * r, j = 16 * jb + c;
            double x = acc0[r] + acc1[r];
            if (E1[i * ES1 + i] == 0.0) x = 0.0;                        // a dropped row of R
            B12[i * BS + j] = x;
            if (64 + j < n) R[(size_t)i * ldr + 64 + j] = x;
        }
    }
    __syncthreads();
    // R11, W11 out (and the two zero blocks); meanwhile the updated A22 in registers: upper 16 x 16 blocks, K = 64
    double wkeep[8];                                                    // W11 stays in registers for the last product (round 4:
#pragma unroll                                                          // it used to be read back from global memory, ~1.5 us)
    for (int u = 0; u < 8; ++u) {
        const int e = tid + 512 * u, i = e >> 6, j = e & 63;
        wkeep[u] = (j <= i) ? E1[i * ES1 + 64 + j] : 0.0;
        R[(size_t)i * ldr + j] = (j >= i) ? E1[i * ES1 + j] : 0.0;
        Wo[widx(i, j)] = wkeep[u];
        if (64 + j < n) Wo[widx(i, 64 + j)] = 0.0;                          // W12 = 0
        if (i < n2) R[(size_t)(64 + i) * ldr + j] = 0.0;                  // R21 = 0
    }
    double t22[2][4];
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
        const int blk = w + 8 * slot;
        if (blk < 10) {
            int bi = 0, rem = blk;
            while (rem >= 4 - bi) { rem -= 4 - bi; ++bi; }
            const int bj = bi + rem;
            double a[16], b[16];
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                a[st] = B12[(4 * st + ks) * BS + 16 * bi + c];
                b[st] = B12[(4 * st + ks) * BS + 16 * bj + c];
            }
            v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int st = 0; st < 16; st += 2) {
                acc0 = GSMVI_MFMA_F64(a[st], b[st], acc0);
                acc1 = GSMVI_MFMA_F64(a[st + 1], b[st + 1], acc1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * bi + ks + 4 * r, j = 16 * bj + c;
                double x = (i == j) ? 1.0 : 0.0;                        // identity beyond n2
                if (i < n2 && j < n2) {
                    x = A[(size_t)(64 + i) * lda + 64 + j];
                    if (SEMIDEF && i == j && !tol_applied) x -= GSMVI_DEP_TOL * x;
                    x -= acc0[r] + acc1[r];
                }
                t22[slot][r] = (j >= i) ? x : 0.0;
            }
        }
    }
    __syncthreads();                                                    // every read of R11 / W11 in E1 is done
    for (int e = tid; e < 64 * 64; e += 512) {                          // blocks below the block diagonal: zero
        const int i = e >> 6, j = e & 63;
        if ((j >> 4) < (i >> 4)) E1[i * ES1 + j] = 0.0;
    }
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
        const int blk = w + 8 * slot;
        if (blk < 10) {
            int bi = 0, rem = blk;
            while (rem >= 4 - bi) { rem -= 4 - bi; ++bi; }
            const int bj = bi + rem;
#pragma unroll
            for (int r = 0; r < 4; ++r) E1[(16 * bi + ks + 4 * r) * ES1 + 16 * bj + c] = t22[slot][r];
        }
    }
    __syncthreads();
    chol64_blk<ES1, SEMIDEF, 1>(E1, scr, n2, &sh_fail[1], moderate);   // [A22' | I] -> [R22 | W22]
    for (int e = tid; e < 64 * 64; e += 512) {
        const int i = e >> 6, j = e & 63;
        if (i < n2 && j < n2) {
            R[(size_t)(64 + i) * ldr + 64 + j] = (j >= i) ? E1[i * ES1 + j] : 0.0;
            Wo[widx(64 + i, 64 + j)] = (j <= i) ? E1[i * ES1 + 64 + j] : 0.0;
        }
    }
    __syncthreads();                                                    // R22 (E1 left half) is out: the half becomes T1
    // T1 = W22 R12^T: (i in block 2, j in block 1), k over block 2; W22[i][k] = 0 for k > i
    {
        double t1[2][4];
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            const int blk = w + 8 * slot, ib = blk >> 2, jb = blk & 3;
            double a[16], b[16];
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int k = 4 * st + ks;
                a[st] = E1[(16 * ib + c) * ES1 + 64 + k];
                b[st] = B12[(16 * jb + c) * BS + k];
            }
            v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int st = 0; st < 16; st += 2) {
                if (st < 4 * (ib + 1)) {
                    acc0 = GSMVI_MFMA_F64(a[st], b[st], acc0);
                    acc1 = GSMVI_MFMA_F64(a[st + 1], b[st + 1], acc1);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) t1[slot][r] = acc0[r] + acc1[r];
        }
        __syncthreads();                                                // all reads of B12 (R12) and of E1's left half are done
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            const int blk = w + 8 * slot, ib = blk >> 2, jb = blk & 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) E1[(16 * ib + ks + 4 * r) * ES1 + 16 * jb + c] = t1[slot][r];
        }
        // W11 (kept in registers since it left E1) into B12
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 512 * u, i = e >> 6, j = e & 63;
            B12[i * BS + j] = wkeep[u];
        }
    }
    __syncthreads();
    // W21 = -T1 W11: (i in block 2, j in block 1), k over block 1; W11[k][j] = 0 for k < j
    for (int blk = w; blk < 16; blk += 8) {
        const int ib = blk >> 2, jb = blk & 3;
        double a[16], b[16];
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int k = 4 * st + ks;
            a[st] = E1[(16 * ib + c) * ES1 + k];
            b[st] = B12[k * BS + 16 * jb + c];
        }
        v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int st = 0; st < 16; st += 2) {
            if (st >= 4 * jb) {                                         // wave-uniform
                acc0 = GSMVI_MFMA_F64(a[st], b[st], acc0);
                acc1 = GSMVI_MFMA_F64(a[st + 1], b[st + 1], acc1);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ib + ks + 4 * r, j = 16 * jb + c;
            if (i < n2) Wo[widx(64 + i, j)] = -(acc0[r] + acc1[r]);
        }
    }
    if (tid == 0) {
        const int f = sh_fail[0] != 0 ? sh_fail[0] : (sh_fail[1] != 0 ? 64 + sh_fail[1] : 0);
        *info = f;
        if (sh_info) *sh_info = f;
    }
}

template <bool SEMIDEF, bool WT = false>
__device__ __forceinline__ void chol128w_body(int n, const double* __restrict__ A, int lda, double* __restrict__ R, int ldr,
                                              double* __restrict__ Wo, int ldw, int* __restrict__ info,
                                              int* sh_info = nullptr, bool tol_applied = false, int moderate_ext = -1) {
    CHOL128W_LDS(E1, B12, scr, sh_fail, sh_moderate);
    chol128w_core<SEMIDEF, WT>(E1, B12, scr, sh_fail, &sh_moderate, n, A, lda, R, ldr, Wo, ldw, info, sh_info, tol_applied,
                               moderate_ext);
}

// One diagonal-block job of a PAIRED launch (round 4): two independent one-workgroup factorisations with the inverse factor,
// 64 < nb <= 128 each, run as the two workgroups of one launch instead of two launches behind each other (each is a
// ~40 us chain of 128 pivots on one CU; 254 CUs idle either way).  info chaining as k_cholw_ld: info_off == 0 writes *info,
// a later block records its failure only when the earlier ones passed.
struct cholw_job {
    int nb;
    const double* A;
    int lda;
    double* R;
    int ldr;
    double* W;
    int ldw;
    int* info;
    int info_off, tol_applied;
    const double* dg;                  // SEMIDEF jobs: diagonal the magnitude guard looks at (dg_n entries, stride dg_stride), or null
    int dg_n, dg_stride;
    int cond_guard;                    // k_bam_cholw_pair's second job: (max / min R_ii)^2 > 1e8 over the kept rows counts as a failure
};
