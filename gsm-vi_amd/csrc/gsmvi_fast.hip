// Guard-free fast paths of the dense-covariance GSM update for gfx950.
//
// Selected by the ABI when D % 64 == 0, every leading dimension is even, every base pointer is
// 16-byte aligned and B is one of {8,16,32,64} (BASELINE configs c2, c3, c5).  Everything else
// runs the guarded generic kernels of gsmvi_kernels.hip (same arithmetic, same reduction order
// inside a kernel family is NOT promised across the two families).
//
// These kernels are latency-bound at D=1024 (8 MB of covariance over 256 CUs = 32 KB per CU), so
// the design rule is: issue every global load of a workgroup in ONE batch, wait once, then MFMA,
// then store.  No data-dependent branch sits between a load and its use.
#include "gsmvi_common.h"
#include <hip/hip_ext.h>

#define GSMVI_LAUNCH(kern, grid, block, shmem, st, ev, ...)                                         \
    do {                                                                                           \
        if (ev)                                                                                    \
            hipExtLaunchKernelGGL(kern, grid, block, shmem, st, (ev)[0], (ev)[1], 0, __VA_ARGS__); \
        else                                                                                       \
            hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__);                         \
    } while (0)

// =====================================================================================
// Panel product partials:  Pp[kc][r][j] = sum_{i in rows(kc)} alpha (A[r][i] - shift[i]) M[i][j]
// Workgroup = 16 columns of M x 256-row chunks; wave w owns rows 64w..64w+63 of the chunk and
// MFMA k-slot ks of step s is row 64w + 4s + ks.  M (the D x D covariance / precision / factor) is
// streamed once from HBM as 128-B row segments straight into registers.  The left operand chunk
// A[:, 256 rows] (16*MT x 256 doubles) is loaded by the whole workgroup with fully coalesced 16-B
// accesses and staged in LDS ([row][258]: conflict-free ds_read_b64 for the MFMA A operand), because
// fragment-shaped loads of it touch 64 cache lines per instruction.
// D % 64 == 0: a wave's 64 rows are all inside or all outside the matrix.
// =====================================================================================
template <int MT, bool HAS_SHIFT>
__global__ __launch_bounds__(256) void k_panel_fast(int D, int nrows, const double* __restrict__ A, int lda,
                                                    const double* __restrict__ shift, double alpha,
                                                    const double* __restrict__ M, int ldm,
                                                    double* __restrict__ Pp, int chunks_per_wg) {
    constexpr int LDG = 258;                       // LDS row stride of the staged A chunk (doubles)
    constexpr int NR = 16 * MT;
    __shared__ __attribute__((aligned(16))) double As[NR * LDG];   // also reused for the reduction
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int j = blockIdx.x * 16 + c;
    const int r0 = blockIdx.z * NR;

    v4d acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4d){0.0, 0.0, 0.0, 0.0};

    for (int ch = 0; ch < chunks_per_wg; ++ch) {
        const int cbase = (blockIdx.y * chunks_per_wg + ch) * 256;          // block-uniform
        if (cbase >= D) break;
        const int wbase = cbase + w * 64;                                   // wave-uniform
        const bool wave_in = wbase < D;
        // ---- every global load of this chunk in one batch ----
        double m[16];
        {
            const double* mp = M + (size_t)((wave_in ? wbase : 0) + ks) * ldm + j;
#pragma unroll
            for (int s = 0; s < 16; ++s) m[s] = mp[(size_t)(4 * s) * ldm];
        }
        v2d ga[8 * MT];                    // 16*MT rows x 128 16-B units = 2048*MT units / 256 threads
        v2d gs[8 * MT];
#pragma unroll
        for (int q = 0; q < 8 * MT; ++q) {
            const int u = q * 256 + tid;
            const int row = u >> 7, c16 = u & 127;
            const int grow = r0 + row;
            const int col = cbase + 2 * c16;
            const bool ok = grow < nrows && col < D;
            ga[q] = *reinterpret_cast<const v2d*>(A + (size_t)(grow < nrows ? grow : nrows - 1) * lda +
                                                  (col < D ? col : 0));
            if (HAS_SHIFT) gs[q] = *reinterpret_cast<const v2d*>(shift + (col < D ? col : 0));
            if (!ok) ga[q] = HAS_SHIFT ? gs[q] : (v2d){0.0, 0.0};          // contributes alpha*(x-x) = 0
        }
        if (ch > 0) __syncthreads();       // previous chunk's MFMA reads of As are done
#pragma unroll
        for (int q = 0; q < 8 * MT; ++q) {
            const int u = q * 256 + tid;
            const int row = u >> 7, c16 = u & 127;
            v2d v = ga[q];
            if (HAS_SHIFT) { v.x -= gs[q].x; v.y -= gs[q].y; }
            v.x *= alpha; v.y *= alpha;
            *reinterpret_cast<v2d*>(&As[row * LDG + 2 * c16]) = v;
        }
        __syncthreads();
        if (wave_in) {
            const double* ap = As + c * LDG + 64 * w + ks;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = GSMVI_MFMA_F64(ap[mt * 16 * LDG + 4 * s], m[s], acc[mt]);
            }
        }
    }

    // cross-wave reduction through LDS (fixed order => deterministic), red[w][row][17]
    __syncthreads();
    double* red = As;                      // 4 * NR * 17 <= NR * 258
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(w * NR + 16 * mt + ks + 4 * r) * 17 + c] = acc[mt][r];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MT; ++k) {
        const int idx = tid + 256 * k;
        const int rr = idx >> 4, cc = idx & 15;
        const int row = r0 + rr;
        if (row < nrows) {
            const double s = (red[(0 * NR + rr) * 17 + cc] + red[(1 * NR + rr) * 17 + cc]) +
                             (red[(2 * NR + rr) * 17 + cc] + red[(3 * NR + rr) * 17 + cc]);
            Pp[((size_t)blockIdx.y * nrows + row) * D + blockIdx.x * 16 + cc] = s;
        }
    }
}

// =====================================================================================
// Per-sample scalars (gsm_numpy.py:8-10,15): one 1024-thread workgroup per sample, every thread
// owns EPT elements of the row; all loads first, one block reduction, thread 0 writes the
// coefficients.  SG_b = sum_kc Pp[kc][b] is written out for the covariance kernel.
// =====================================================================================
template <int EPT, int KCT>
__global__ __launch_bounds__(1024) void k_gsm_scalars_fast(int D, int B, int KC, const double* __restrict__ X,
                                                           int ldx, const double* __restrict__ G, int ldg,
                                                           const double* __restrict__ mu0,
                                                           const double* __restrict__ Pp,
                                                           double* __restrict__ SG, int ldsg,
                                                           double* __restrict__ coef, int ldc,
                                                           double* __restrict__ Xout, int ldxo) {
    __shared__ double lds[32];
    const int b = blockIdx.x, tid = threadIdx.x;
    double pp[EPT][KCT], xv[EPT], gv[EPT], mv0[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + 1024 * e;
        const int ic = i < D ? i : D - 1;
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc)
            pp[e][kc] = Pp[((size_t)(kc < KC ? kc : KC - 1) * B + b) * D + ic];
        xv[e] = X[(size_t)b * ldx + ic];
        gv[e] = G[(size_t)b * ldg + ic];
        mv0[e] = mu0[ic];
    }
    double p0 = 0.0, p1 = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + 1024 * e;
        double sg = 0.0;
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc) sg += (kc < KC) ? pp[e][kc] : 0.0;
        if (i < D) {
            SG[(size_t)b * ldsg + i] = sg;
            if (Xout) Xout[(size_t)b * ldxo + i] = xv[e];
            const double d = mv0[e] - xv[e];
            p0 += gv[e] * sg;
            p1 += d * gv[e];
        }
    }
    p0 = wave_sum(p0);
    p1 = wave_sum(p1);
    const int w = tid >> 6;
    if ((tid & 63) == 0) {
        lds[2 * w] = p0;
        lds[2 * w + 1] = p1;
    }
    __syncthreads();
    if (tid == 0) {
        double gSg = 0.0, mv = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            gSg += lds[2 * k];
            mv += lds[2 * k + 1];
        }
        const double rho = 0.5 * sqrt(1.0 + 4.0 * (gSg + mv * mv)) - 0.5;
        const double den = 1.0 + rho + mv;
        const double cc = (gSg - mv) / den;
        const double beta = 1.0 / (1.0 + rho);
        double* cb = coef + (size_t)b * ldc;
        cb[0] = 1.0 - (1.0 + cc) * beta;
        cb[1] = beta;
        cb[2] = cc;
        cb[3] = rho;
    }
}

// =====================================================================================
// Symmetric rank-2B covariance update (gsm_numpy.py:21-23,50-53).
// One workgroup = one 32x32 tile pair (I,J), I <= J, of the UPPER triangle; it reads S0[I,J]
// once, computes W = S0[I,J] + (1/B) sum_b (d_b[I] d_b[J]^T - e_b[I] e_b[J]^T) with fp64 MFMA
// (one 16x16 tile per wave, K = 2B), stores S[I,J] = W and, through an LDS transpose, the mirror
// S[J,I] = W^T.  S0 must be symmetric (only its upper triangle is read); S is exactly symmetric.
// Bytes moved: 4 D^2 read + 8 D^2 written (algorithmic count of SURVEY 8(d): 16 D^2).
// Factor tiles are built from X, SG, mu0 and coef while staging to LDS ([row][k], stride 2B+2
// doubles: conflict-free ds_read_b64 for both MFMA operands).  Diagonal workgroups also write the
// new mean mu = mu0 + mean_b dmu_b.
// =====================================================================================
template <int SB>
__global__ __launch_bounds__(256) void k_gsm_cov_sym(int D, const double* __restrict__ X, int ldx,
                                                     const double* __restrict__ SG, int ldsg,
                                                     const double* __restrict__ mu0,
                                                     const double* __restrict__ coef, int ldc,
                                                     const double* __restrict__ S0, int lds0,
                                                     double* __restrict__ S, int lds,
                                                     double* __restrict__ mu_out, int dbg) {
    constexpr int KF = 2 * SB;          // MFMA reduction length
    constexpr int RS = KF + 2;          // LDS row stride (doubles)
    constexpr int NIT = SB / 8;         // staging iterations (8 samples x 32 columns per pass)
    __shared__ __attribute__((aligned(16))) double smem[2 * 32 * RS];   // >= 32*33 for every SB
    double* FA = smem;
    double* FB = smem + 32 * RS;

    // upper-triangle tile pair from the linear block index
    const int nt = D >> 5;
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
    const int I0 = ti * 32, J0 = tj * 32;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int wr = w >> 1, wc = w & 1;
    constexpr double invB = 1.0 / (double)SB;

    // ---- every global load of this workgroup, in one batch --------------------------------
    double s0[4];
    const size_t srow = (size_t)(I0 + 16 * wr + ks);
    const int scol = J0 + 16 * wc + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) s0[r] = (dbg & 2) ? 1.0 : S0[(srow + 4 * r) * lds0 + scol];
    __builtin_amdgcn_sched_barrier(0);   // keep the HBM loads of S0 ahead of the L2-resident staging loads

    const int ii = tid & 31, bq = tid >> 5;
    const double mI = mu0[I0 + ii], mJ = mu0[J0 + ii];
    double xI[NIT], gI[NIT], xJ[NIT], gJ[NIT];
    v2d ab[NIT];
    double cc[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int b = bq + 8 * k;
        if (dbg & 4) { xI[k] = gI[k] = xJ[k] = gJ[k] = 0.5; ab[k] = (v2d){0.5, 0.5}; cc[k] = 0.1; continue; }
        xI[k] = X[(size_t)b * ldx + I0 + ii];
        gI[k] = SG[(size_t)b * ldsg + I0 + ii];
        xJ[k] = X[(size_t)b * ldx + J0 + ii];
        gJ[k] = SG[(size_t)b * ldsg + J0 + ii];
        ab[k] = *reinterpret_cast<const v2d*>(coef + (size_t)b * ldc);
        cc[k] = coef[(size_t)b * ldc + 2];
    }

    // ---- factor tiles -> LDS ---------------------------------------------------------------
    double dmu_part = 0.0;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int b = bq + 8 * k;
        const double dI = mI - xI[k], dJ = mJ - xJ[k];
        const double eI = ab[k].x * dI + ab[k].y * gI[k];
        const double eJ = ab[k].x * dJ + ab[k].y * gJ[k];
        FA[ii * RS + b] = dI;
        FA[ii * RS + SB + b] = eI;
        FB[ii * RS + b] = dJ * invB;
        FB[ii * RS + SB + b] = -eJ * invB;
        dmu_part += ab[k].y * ((gI[k] - dI) - cc[k] * dI);
    }
    __syncthreads();

    // ---- MFMA: one 16x16 tile per wave, K = 2B ---------------------------------------------
    const double* ap = FA + (16 * wr + c) * RS + ks;
    const double* bp = FB + (16 * wc + c) * RS + ks;
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KF / 4; s += 2) {       // two independent chains hide the MFMA latency
        if (dbg & 8) break;
        acc0 = GSMVI_MFMA_F64(ap[4 * s], bp[4 * s], acc0);
        acc1 = GSMVI_MFMA_F64(ap[4 * s + 4], bp[4 * s + 4], acc1);
    }
    double wv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wv[r] = s0[r] + (acc0[r] + acc1[r]);

    // ---- direct store S[I,J] ----------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(srow + 4 * r) * lds + scol] = wv[r];

    // ---- mirror store S[J,I] = W^T through LDS, and the new mean on the diagonal -------------
    __syncthreads();                      // everyone is done reading FA / FB
    double* LW = smem;                    // 32 x 33 transpose buffer (aliases the factor tiles)
    if (ti != tj && !(dbg & 1)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) LW[(16 * wr + ks + 4 * r) * 33 + 16 * wc + c] = wv[r];
    } else if (ti == tj) {
        smem[bq * 32 + ii] = dmu_part;
    }
    __syncthreads();
    if (ti != tj && !(dbg & 1)) {
        // element (row j = 16 wr + ks + 4r of J, col i = 16 wc + c of I) = W[i][j]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = LW[(16 * wc + c) * 33 + 16 * wr + ks + 4 * r];
            S[(size_t)(J0 + 16 * wr + ks + 4 * r) * lds + I0 + 16 * wc + c] = v;
        }
    } else if (ti == tj && tid < 32) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += smem[q * 32 + tid];
        mu_out[I0 + tid] = mu0[I0 + tid] + s * invB;
    }
}

// ---- launch helpers ------------------------------------------------------------------------
void gsmvi_launch_panel_fast(hipStream_t st, hipEvent_t* ev, int MT, dim3 grid, int D, int nrows, const double* A,
                             int lda, const double* shift, double alpha, const double* M, int ldm, double* Pp,
                             int chunks_per_wg) {
#define PF(MTV, HS)                                                                                         \
    GSMVI_LAUNCH((k_panel_fast<MTV, HS>), grid, dim3(256), 0, st, ev, D, nrows, A, lda, shift, alpha, M, ldm, \
                 Pp, chunks_per_wg)
    if (shift) {
        if (MT == 1) PF(1, true); else if (MT == 2) PF(2, true); else PF(4, true);
    } else {
        if (MT == 1) PF(1, false); else if (MT == 2) PF(2, false); else PF(4, false);
    }
#undef PF
}

// returns false when (D, KC) has no instantiation
bool gsmvi_launch_gsm_scalars_fast(hipStream_t st, hipEvent_t* ev, int D, int B, int KC, const double* X, int ldx,
                                   const double* G, int ldg, const double* mu0, const double* Pp, double* SG,
                                   int ldsg, double* coef, int ldc, double* Xout, int ldxo) {
    const int ept = (D + 1023) / 1024;
#define SF(E, K)                                                                                              \
    GSMVI_LAUNCH((k_gsm_scalars_fast<E, K>), dim3(B), dim3(1024), 0, st, ev, D, B, KC, X, ldx, G, ldg, mu0, Pp, \
                 SG, ldsg, coef, ldc, Xout, ldxo)
    if (KC > 8 || ept > 4) return false;
    const int kct = KC <= 1 ? 1 : (KC <= 2 ? 2 : (KC <= 4 ? 4 : 8));
    const int e = ept <= 1 ? 1 : (ept <= 2 ? 2 : 4);
    if (e == 1) { if (kct == 1) SF(1, 1); else if (kct == 2) SF(1, 2); else if (kct == 4) SF(1, 4); else SF(1, 8); }
    else if (e == 2) { if (kct == 1) SF(2, 1); else if (kct == 2) SF(2, 2); else if (kct == 4) SF(2, 4); else SF(2, 8); }
    else { if (kct == 1) SF(4, 1); else if (kct == 2) SF(4, 2); else if (kct == 4) SF(4, 4); else SF(4, 8); }
#undef SF
    return true;
}

bool gsmvi_launch_gsm_cov_sym(hipStream_t st, hipEvent_t* ev, int D, int B, const double* X, int ldx,
                              const double* SG, int ldsg, const double* mu0, const double* coef, int ldc,
                              const double* S0, int lds0, double* S, int lds, double* mu_out, int dbg) {
    const int nt = D / 32;
    const dim3 grid(nt * (nt + 1) / 2);
#define CS(SBV)                                                                                             \
    GSMVI_LAUNCH(k_gsm_cov_sym<SBV>, grid, dim3(256), 0, st, ev, D, X, ldx, SG, ldsg, mu0, coef, ldc, S0, lds0, \
                 S, lds, mu_out, dbg)
    switch (B) {
        case 8: CS(8); break;
        case 16: CS(16); break;
        case 32: CS(32); break;
        case 64: CS(64); break;
        default: return false;
    }
#undef CS
    return true;
}
