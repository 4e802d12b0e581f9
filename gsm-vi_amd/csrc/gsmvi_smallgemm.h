// Generic product of SMALL matrices (inner dimension <= 256) with operand and epilogue functors: the pieces between the
// one-workgroup factorisations of the 2B x 2B chain for 2B > 64 (round 4: gsmvi_factor.hip, factor_back / factor_chain_big).
//
//   C(i, j) = sum_{k < K} a(i, k) b(k, j),   i < m, j < p        op.store(i, j, value) writes it where it belongs
//
// One 16 x 16 output block per workgroup (256 threads): both operand tiles (16 x K and K x 16) are staged in LDS by all threads,
// along the direction in which their source is contiguous; K is split over the four waves (wave w takes the k-steps w, w + 4, ...,
// two accumulator chains), the four partial blocks are summed through LDS in a fixed order: deterministic, launch-bound like the
// Newton-Schulz steps of gsmvi_bam_small.hip whose block kernel this generalises.  The functor's a(i, k) / b(k, j) receive
// CLAMPED indices (the kernel zeroes what lies outside); op.skip() lets an op leave the launch early on a device flag.
#pragma once
#include "gsmvi_common.h"

#define SMALLGEMM_SA 260                                     // LDS strides: A [16][260] (260 = 4 mod 32: conflict-free fragment reads),
#define SMALLGEMM_SB 17                                      // B [k][17]
template <class OP>
__device__ __forceinline__ void small_gemm_block(const OP& op, int blk, double* As, double* Bs, double* red) {
    constexpr int SA = SMALLGEMM_SA, SB = SMALLGEMM_SB;
    if (op.skip()) return;
    const int m = op.m, p = op.p, K = op.K;
    const int nbj = (p + 15) >> 4;
    const int bi = blk / nbj, bj = blk - bi * nbj, i0 = 16 * bi, j0 = 16 * bj;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, cc = l & 15, ks = l >> 4;
    const int nk = (K + 3) >> 2;
    // ---- both operand tiles through LDS, loaded along the direction in which the SOURCE is contiguous (OP::A_KMAJOR /
    // OP::B_KMAJOR): an MFMA fragment read straight from a row-major matrix touches 16 cache lines per load instruction
    // (the first version of this kernel did that: 8.7 - 12.9 us per launch at K = 128 ... 256) ----
    const int K4 = 4 * nk;                                   // K rounded up to the MFMA step (<= 256)
    for (int e0 = 0; e0 < 16 * K4; e0 += 256 * 8) {
        double va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u + tid;
            int row, k;
            if (OP::A_KMAJOR) { row = e / K4; k = e - row * K4; } else { k = e >> 4; row = e & 15; }
            const bool in = e < 16 * K4 && k < K && (i0 + row) < m;
            const double v = op.a((i0 + row) < m ? i0 + row : m - 1, k < K ? k : K - 1);
            va[u] = in ? v : 0.0;
            int kb, col;
            if (OP::B_KMAJOR) { col = e / K4; kb = e - col * K4; } else { kb = e >> 4; col = e & 15; }
            const bool inb = e < 16 * K4 && kb < K && (j0 + col) < p;
            const double vv = op.b(kb < K ? kb : K - 1, (j0 + col) < p ? j0 + col : p - 1);
            vb[u] = inb ? vv : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u + tid;
            if (e < 16 * K4) {
                int row, k, kb, col;
                if (OP::A_KMAJOR) { row = e / K4; k = e - row * K4; } else { k = e >> 4; row = e & 15; }
                if (OP::B_KMAJOR) { col = e / K4; kb = e - col * K4; } else { kb = e >> 4; col = e & 15; }
                As[row * SA + k] = va[u];
                Bs[kb * SB + col] = vb[u];
            }
        }
    }
    __syncthreads();
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int st = w; st < nk; st += 8) {                      // wave w: k-steps w, w + 4, ... (two accumulator chains)
        const int k0 = 4 * st + ks;
        acc0 = GSMVI_MFMA_F64(As[cc * SA + k0], Bs[k0 * SB + cc], acc0);
        if (st + 4 < nk) {
            const int k1 = k0 + 16;
            acc1 = GSMVI_MFMA_F64(As[cc * SA + k1], Bs[k1 * SB + cc], acc1);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w * 256 + (ks + 4 * r) * 16 + cc] = acc0[r] + acc1[r];
    __syncthreads();
    const int t = threadIdx.x, i = i0 + (t >> 4), j = j0 + (t & 15);
    const double v = (red[t] + red[256 + t]) + (red[512 + t] + red[768 + t]);
    if (i < m && j < p) op.store(i, j, v);
}

template <class OP>
__global__ __launch_bounds__(256) void k_small_gemm(OP op) {
    __shared__ double As[16 * SMALLGEMM_SA], Bs[256 * SMALLGEMM_SB], red[4 * 256];
    small_gemm_block(op, (int)blockIdx.x, As, Bs, red);
}
// two INDEPENDENT products in one launch (a dependent launch costs ~2 us before its first instruction: scripts/nsbench.hip):
// workgroups [0, nblk1) run op1, the rest op2
template <class OP1, class OP2>
__global__ __launch_bounds__(256) void k_small_gemm2(OP1 op1, OP2 op2, int nblk1) {
    __shared__ double As[16 * SMALLGEMM_SA], Bs[256 * SMALLGEMM_SB], red[4 * 256];
    if ((int)blockIdx.x < nblk1) small_gemm_block(op1, (int)blockIdx.x, As, Bs, red);
    else small_gemm_block(op2, (int)blockIdx.x - nblk1, As, Bs, red);
}

template <class OP>
static inline void small_gemm_launch(hipStream_t st, const OP& op) {
    const int nbi = (op.m + 15) >> 4, nbj = (op.p + 15) >> 4;
    hipLaunchKernelGGL(k_small_gemm<OP>, dim3(nbi * nbj), dim3(256), 0, st, op);
}
template <class OP1, class OP2>
static inline void small_gemm_launch2(hipStream_t st, const OP1& op1, const OP2& op2) {
    const int n1 = ((op1.m + 15) >> 4) * ((op1.p + 15) >> 4), n2 = ((op2.m + 15) >> 4) * ((op2.p + 15) >> 4);
    hipLaunchKernelGGL((k_small_gemm2<OP1, OP2>), dim3(n1 + n2), dim3(256), 0, st, op1, op2, n1);
}

// ---- two-level blocked Cholesky with the inverse factor, A = R^T R, W = R^-T, for an n x n matrix, 128 < n <= 256 -------------
// Blocks of n1 = 128 and n2 = n - 128.  The diagonal blocks are factored by the one-workgroup kernels ([A | I] -> [R | W]); the
// four products between them run here:
//   R12 = D' W11 A12,   A22' = A22 - R12^T R12,   [A22' | I] -> [R22 | W22],   T1 = W22 R12^T,   W21 = -T1 W11
// (D' zeroes the rows the rank-revealing rule dropped, R11[i][i] == 0: what chol128w_body does inside one 128 block).
struct OpBlkR12 {                                  // R[0:n1, n1:n] = D' W11 A12, R[n1:n, 0:n1] = 0
    static constexpr bool A_KMAJOR = true, B_KMAJOR = false;   // which index the SOURCE of a(i, k) / b(k, j) is contiguous in
    int m, p, K;                                   // n1, n2, n1
    const double *W, *A;
    double* R;
    int ldw, lda, ldr, n1;
    // early first block (factor-form BaM: [Gamma11 | I] -> [R11 | W11] was factored beside k_bam_cholw into compact buffers):
    // W above is that W11 (ldw), Rsrc that R11 (ldrs); this launch also copies both into the n x n matrices (R, Wdst)
    const double* Rsrc;
    int ldrs;
    double* Wdst;
    int ldwd;
    __device__ bool skip() const { return false; }
    __device__ double a(int i, int k) const { return W[(size_t)i * ldw + k]; }
    __device__ double b(int k, int j) const { return A[(size_t)k * lda + n1 + j]; }
    __device__ void store(int i, int j, double v) const {
        const double rii = Rsrc ? Rsrc[(size_t)i * ldrs + i] : R[(size_t)i * ldr + i];
        R[(size_t)i * ldr + n1 + j] = (rii == 0.0) ? 0.0 : v;
        R[(size_t)(n1 + j) * ldr + i] = 0.0;       // the block below the diagonal: the factor is read as a full matrix
        if (Rsrc)
            for (int jj = j; jj < n1; jj += p) {
                R[(size_t)i * ldr + jj] = Rsrc[(size_t)i * ldrs + jj];
                Wdst[(size_t)i * ldwd + jj] = W[(size_t)i * ldw + jj];
            }
    }
};
struct OpBlkS22 {                                  // S22 = A22 (diagonal lowered by its rounding floor when semidef) - R12^T R12
    static constexpr bool A_KMAJOR = false, B_KMAJOR = false;   // which index the SOURCE of a(i, k) / b(k, j) is contiguous in
    int m, p, K;                                   // n2, n2, n1
    const double *R, *A;
    double* S;
    int ldr, lda, n1, semidef;
    __device__ bool skip() const { return false; }
    __device__ double a(int i, int k) const { return R[(size_t)k * ldr + n1 + i]; }
    __device__ double b(int k, int j) const { return R[(size_t)k * ldr + n1 + j]; }
    __device__ void store(int i, int j, double v) const {
        double x = A[(size_t)(n1 + i) * lda + n1 + j];
        if (semidef && i == j) x -= GSMVI_DEP_TOL * x;
        S[(size_t)i * m + j] = x - v;
    }
};
struct OpBlkT1 {                                   // T1 = W22 R12^T   (n2 x n1)
    static constexpr bool A_KMAJOR = true, B_KMAJOR = true;   // which index the SOURCE of a(i, k) / b(k, j) is contiguous in
    int m, p, K;                                   // n2, n1, n2
    const double *W, *R;
    double* T1;
    int ldw, ldr, n1;
    __device__ bool skip() const { return false; }
    __device__ double a(int i, int k) const { return W[(size_t)(n1 + i) * ldw + n1 + k]; }
    __device__ double b(int k, int j) const { return R[(size_t)j * ldr + n1 + k]; }
    __device__ void store(int i, int j, double v) const { T1[(size_t)i * p + j] = v; }
};
struct OpBlkW21 {                                  // W[n1:n, 0:n1] = -T1 W11; also the zero blocks W12 and R21
    static constexpr bool A_KMAJOR = true, B_KMAJOR = false;   // which index the SOURCE of a(i, k) / b(k, j) is contiguous in
    int m, p, K;                                   // n2, n1, n1
    const double* T1;
    double *W, *R;
    int ldw, ldr, n1;
    __device__ bool skip() const { return false; }
    __device__ double a(int i, int k) const { return T1[(size_t)i * p + k]; }
    __device__ double b(int k, int j) const { return W[(size_t)k * ldw + j]; }
    __device__ void store(int i, int j, double v) const {
        W[(size_t)(n1 + i) * ldw + j] = -v;
        W[(size_t)j * ldw + n1 + i] = 0.0;
        R[(size_t)(n1 + i) * ldr + j] = 0.0;
    }
};

// ---- the n x n products of the 2B x 2B chain, 64 < n <= 256 -----------------------------------------------------------------
struct OpSmallA {                                  // A' = I + (Rg J) Rg^T;  J = (1/B) [[0, I], [I, -I]] or, jmode, diag(I, -I) unscaled
    static constexpr bool A_KMAJOR = true, B_KMAJOR = true;   // which index the SOURCE of a(i, k) / b(k, j) is contiguous in
    int m, p, K;                                   // n, n, n -- or n1, n1, n: the leading block A'11 alone (it needs only the
    const double* Rg;                              // first block row of Rg; the two-level chain factors it early)
    const int* info_g;
    double* Ap;
    int B, jmode, ld;                              // ld: leading dimension of Rg and A' (n)
    // jmode 2 (round 5, factor-form BaM in the orthogonal basis, gsmvi_bam.hip): the dense signature J' = S'^T diag(I, -I) S',
    // S' = [[I, 0], [Pi, I]], is applied through Rt = Rg S'^T, which differs from Rg in its (1, 2) block only:
    // Rt12 = R12 + X, X = R11 Pi^T (B x B, OpChainX).  Rg itself stays what W = Rg^-T and the block factorisation need.
    const double* X = nullptr;
    __device__ double rt(int i, int k) const {
        const double r = Rg[(size_t)i * ld + k];
        return (jmode == 2 && i < B && k >= B) ? r + X[(size_t)i * B + (k - B)] : r;
    }
    __device__ bool skip() const { return false; }
    __device__ double a(int i, int k) const {
        if (jmode) return (k < B) ? rt(i, k) : -rt(i, k);
        return (k < B) ? Rg[(size_t)i * ld + B + k] : (Rg[(size_t)i * ld + k - B] - Rg[(size_t)i * ld + k]);
    }
    __device__ double b(int k, int j) const { return jmode == 2 ? rt(j, k) : Rg[(size_t)j * ld + k]; }
    __device__ void store(int i, int j, double v) const {
        double x = (i == j ? 1.0 : 0.0) + (jmode ? v : v / (double)B);
        if (*info_g != 0) x = (i == j) ? -1.0 : 0.0;  // Gamma was singular: force the PD test to fail
        Ap[(size_t)i * ld + j] = x;
    }
};
struct OpChainP {                                  // P = (T - I)(W S), with the chain's accept / revert decision
    static constexpr bool A_KMAJOR = true, B_KMAJOR = false;   // which index the SOURCE of a(i, k) / b(k, j) is contiguous in
    int m, p, K;
    const double *T, *W, *ab;
    double* P;
    int B;
    int* bad;
    const int *info_g, *info_t, *prior_bad;
    __device__ bool skip() const {
        const int b = (*info_g != 0) || (*info_t != 0) || (prior_bad && *prior_bad != 0);
        if (blockIdx.x == 0 && threadIdx.x == 0) *bad = b;
        return b != 0;
    }
    __device__ double a(int i, int k) const { return T[(size_t)i * m + k] - (i == k ? 1.0 : 0.0); }
    __device__ double b(int k, int j) const {      // (W S)[k][j] = s1 W[k][j] + s2 W[k][j2]
        const int j2 = j < B ? j + B : j;
        const double s1 = j < B ? 1.0 : 0.0, s2 = ab[j];
        return s1 * W[(size_t)k * m + j] + s2 * W[(size_t)k * m + j2];
    }
    __device__ void store(int i, int j, double v) const { P[(size_t)i * m + j] = v; }
};
struct OpChainK {                                  // K'' = (W S)^T P
    static constexpr bool A_KMAJOR = false, B_KMAJOR = false;   // which index the SOURCE of a(i, k) / b(k, j) is contiguous in
    int m, p, K;
    const double *W, *P, *ab;
    double* Kmat;
    int B;
    const int* bad;
    __device__ bool skip() const { return *bad != 0; }
    __device__ double a(int i, int k) const {
        const int i2 = i < B ? i + B : i;
        const double s1 = i < B ? 1.0 : 0.0, s2 = ab[i];
        return s1 * W[(size_t)k * m + i] + s2 * W[(size_t)k * m + i2];
    }
    __device__ double b(int k, int j) const { return P[(size_t)k * m + j]; }
    __device__ void store(int i, int j, double v) const { Kmat[(size_t)i * m + j] = v; }
};

// ---- factor-form BaM in the basis [Vw; Zt] (round 5; gsmvi_bam.hip, "bam_basis") ---------------------------------------------
// With Gvv = Vw Vw^T = R11^T R11, W11 = R11^-T (the early first block of the chain's Gram matrix), M1 = Vw Wq^T and W^T = (L^-1)^T:
//   T   = W11 M1                 M1' = -Gvv^-1 M1 = -W11^T T          Dm = M1 - M1'
//   Zt  = L^-1 (Wq + M1'^T Vw)   = the part of Zw = L^-1 (Wq + M1^T Vw) orthogonal to the rows of Vw        (k_bam_zw with M1')
//   Pi  = L^-1 Dm^T              Zw = Zt + Pi Vw
//   M   = I + Vw^T Vw - Zw^T Zw = I + [Vw; Zt]^T J' [Vw; Zt],   J' = [[I - Pi^T Pi, -Pi^T], [-Pi, -I]] = S'^T diag(I, -I) S',
//   S' = [[I, 0], [Pi, I]]: the chain applies it through Rg S'^T (OpSmallA jmode 2, gsmf_small16_body), J' is never formed
struct OpBasisT {                                  // T = W11 M1
    static constexpr bool A_KMAJOR = true, B_KMAJOR = false;
    int m, p, K;
    const double *W11, *M1;
    double* T;
    int ldw;
    __device__ bool skip() const { return false; }
    __device__ double a(int i, int k) const { return W11[(size_t)i * ldw + k]; }
    __device__ double b(int k, int j) const { return M1[(size_t)k * p + j]; }
    __device__ void store(int i, int j, double v) const { T[(size_t)i * p + j] = v; }
};
struct OpBasisM1p {                                // M1' = -W11^T T, Dm = M1 - M1'
    static constexpr bool A_KMAJOR = false, B_KMAJOR = false;
    int m, p, K;
    const double *W11, *T, *M1;
    double *M1p, *Dm;
    int ldw;
    __device__ bool skip() const { return false; }
    __device__ double a(int i, int k) const { return W11[(size_t)k * ldw + i]; }
    __device__ double b(int k, int j) const { return T[(size_t)k * p + j]; }
    __device__ void store(int i, int j, double v) const {
        M1p[(size_t)i * p + j] = -v;
        Dm[(size_t)i * p + j] = M1[(size_t)i * p + j] + v;
    }
};
struct OpBasisPi {                                 // Pi = W Dm^T (W = L^-1 given as Wt = W^T)
    static constexpr bool A_KMAJOR = false, B_KMAJOR = true;
    int m, p, K;                                   // n, n, n
    const double *Wt, *Dm;
    double* Pi;
    __device__ bool skip() const { return false; }
    __device__ double a(int i, int k) const { return Wt[(size_t)k * m + i]; }
    __device__ double b(int k, int j) const { return Dm[(size_t)j * m + k]; }
    __device__ void store(int i, int j, double v) const { Pi[(size_t)i * m + j] = v; }
};
struct OpChainX {                                  // X = R11 Pi^T (B x B; R11 upper triangular with leading dimension ldr)
    static constexpr bool A_KMAJOR = true, B_KMAJOR = true;
    int m, p, K;                                   // B, B, B
    const double *R11, *Pi;
    double* X;
    int ldr;
    __device__ bool skip() const { return false; }
    __device__ double a(int i, int k) const { return k >= i ? R11[(size_t)i * ldr + k] : 0.0; }
    __device__ double b(int k, int j) const { return Pi[(size_t)j * m + k]; }
    __device__ void store(int i, int j, double v) const { X[(size_t)i * m + j] = v; }
};
