// gfx950 kernels for the dense-covariance GSM update and its neighbours (score, sampler).
//
// Reference being replaced: gsmvi/gsm_numpy.py:4-55 (per-sample update + batch mean).  The
// algebra is the O(B D^2) form of SURVEY Appendix A.1:
//     SG = G S0                               panel product   (k_panel_partial, fp64 MFMA)
//     gSg_b, mv_b, rho_b, den_b, records       per-sample      (k_gsm_scalars, wave reductions)
//     S  = S0 + (Dm^T Dm - E^T E)/B           rank-2B update  (k_gsm_cov_update, fp64 MFMA)
// Layouts: everything row-major fp64 in HBM; S0 is streamed exactly once per pass.
#include "gsmvi_common.h"
#include <hip/hip_ext.h>

// launch with optional dispatch-timestamp events (profiling mode of the ABI)
#define GSMVI_LAUNCH(kern, grid, block, shmem, st, ev, ...)                                         \
    do {                                                                                           \
        if (ev)                                                                                    \
            hipExtLaunchKernelGGL(kern, grid, block, shmem, st, (ev)[0], (ev)[1], 0, __VA_ARGS__); \
        else                                                                                       \
            hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__);                         \
    } while (0)

// =====================================================================================
// Panel product partials:  Pp[kc][r][j] = sum_{i in row-range(kc)} Ahat[r][i] * M[i][j]
//   Ahat = alpha * (A - 1 shift^T)  (shift may be null).  A: nrows x D, M: D x D.
// Grid: x = column strip of 16, y = kc (row range of M), z = block of 16*MT rows of A.
// A workgroup is 4 waves; wave w owns 64 consecutive rows of M per 256-row chunk.  MFMA k-slot
// `ks` of step `s` is row 16*ks + s of those 64, so every lane's A operand is 16 CONTIGUOUS
// doubles of one row of A (one 128-B line) and M is read as 16 coalesced 128-B row segments.
// =====================================================================================
template <int MT>
__global__ __launch_bounds__(256) void k_panel_partial(int D, int ncols, int nrows, const double* __restrict__ A,
                                                       int lda, const double* __restrict__ shift,
                                                       double alpha, const double* __restrict__ M, int ldm,
                                                       double* __restrict__ Pp, int chunks_per_wg,
                                                       int a_vec_ok) {
    __shared__ double red[4][16 * MT][17];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int j = blockIdx.x * 16 + c;
    const int jc = j < ncols ? j : ncols - 1;
    const int r0 = blockIdx.z * (16 * MT);

    v4d acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4d){0.0, 0.0, 0.0, 0.0};

    for (int ch = 0; ch < chunks_per_wg; ++ch) {
        const int chunk = blockIdx.y * chunks_per_wg + ch;
        if (chunk * 256 >= D) break;                       // block-uniform
        const int base = chunk * 256 + w * 64 + ks * 16;   // first of this lane's 16 rows of M

        double m[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int i = base + s;
            const int ic = i < D ? i : D - 1;
            const double v = M[(size_t)ic * ldm + jc];
            m[s] = (i < D && j < ncols) ? v : 0.0;
        }
        double sh[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int i = base + s;
            sh[s] = (shift != nullptr && i < D) ? shift[i] : 0.0;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = r0 + 16 * mt + c;
            const int rowc = row < nrows ? row : nrows - 1;
            const double rowmask = row < nrows ? alpha : 0.0;
            double a[16];
            const double* ap = A + (size_t)rowc * lda;
            if (a_vec_ok && base + 16 <= D) {
                const v2d* vp = reinterpret_cast<const v2d*>(ap + base);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v2d t = vp[q];
                    a[2 * q] = t.x;
                    a[2 * q + 1] = t.y;
                }
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int i = base + s;
                    const double v = ap[i < D ? i : D - 1];
                    a[s] = i < D ? v : 0.0;
                }
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int i = base + s;
                const double av = (i < D) ? rowmask * (a[s] - sh[s]) : 0.0;
                acc[mt] = GSMVI_MFMA_F64(av, m[s], acc[mt]);
            }
        }
    }

    // cross-wave reduction of the four 64-row partial sums (fixed order => deterministic)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][16 * mt + ks + 4 * r][c] = acc[mt][r];
    __syncthreads();
    for (int idx = tid; idx < 16 * MT * 16; idx += 256) {
        const int rr = idx >> 4, cc = idx & 15;
        const int row = r0 + rr, col = blockIdx.x * 16 + cc;
        if (row < nrows && col < ncols) {
            const double s = (red[0][rr][cc] + red[1][rr][cc]) + (red[2][rr][cc] + red[3][rr][cc]);
            Pp[((size_t)blockIdx.y * nrows + row) * ncols + col] = s;
        }
    }
}

// Out[r][i] = addvec[i] + sum_kc Pp[kc][r][i]      (D here = number of columns of the panel)
// The (at most GSMVI_MAX_KC = 8) slab loads are issued as one batch with clamped indices.
// ncols_out <= D: only the first ncols_out columns are written (slabs whose column count was padded to 16).
__global__ __launch_bounds__(256) void k_panel_finish(int D, int nrows, int KC, const double* __restrict__ Pp,
                                                      const double* __restrict__ addvec,
                                                      double* __restrict__ Out, int ldo, int ncols_out) {
    const int i = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (i >= ncols_out) return;
    double v[8];
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) v[kc] = Pp[((size_t)(kc < KC ? kc : KC - 1) * nrows + r) * D + i];
    const double a = addvec ? addvec[i] : 0.0;
    double s = 0.0;
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) s += (kc < KC) ? v[kc] : 0.0;
    Out[(size_t)r * ldo + i] = s + a;
}

// =====================================================================================
// Per-sample stage of the GSM update (gsm_numpy.py:8-18; one workgroup per sample):
//   SG_b = sum_kc Pp[kc][b]          gSg = g.SG   mv = (mu0-x).g
//   rho = 0.5 sqrt(1+4(gSg+mv^2)) - 0.5,  den = 1+rho+mv,  c = (gSg-mv)/den,  beta = 1/(1+rho)
//   dmu_b = beta ((SG_b - d_b) - c d_b)     (= mu_update of gsm_numpy.py:17)
// and writes the sample's RECORD  rec[b] = [ d_b | e_b | dmu_b ]  (3D doubles, row stride ldrec) with
// d_b = mu0 - x_b and e_b = d_b + dmu_b = mu_b - x_b, i.e. the two factor rows of
// S_update_b = d d^T - e e^T (gsm_numpy.py:21-23).  Records are what the covariance kernel and the
// batch-sharded exchange consume.
// =====================================================================================
__global__ __launch_bounds__(256) void k_gsm_scalars(int D, int B, int KC, const double* __restrict__ X,
                                                     int ldx, const double* __restrict__ G, int ldg,
                                                     const double* __restrict__ mu0,
                                                     const double* __restrict__ Pp, double* __restrict__ rec,
                                                     int ldrec) {
    __shared__ double lds[8];
    __shared__ double sh[2];
    const int b = blockIdx.x;
    double* rb = rec + (size_t)b * ldrec;
    double p[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < D; i += 256) {
        double sg = 0.0;
        for (int kc = 0; kc < KC; ++kc) sg += Pp[((size_t)kc * B + b) * D + i];
        const double g = G[(size_t)b * ldg + i];
        const double d = mu0[i] - X[(size_t)b * ldx + i];
        rb[i] = d;
        rb[D + i] = sg;                      // parked here until the scalars are known
        p[0] += g * sg;
        p[1] += d * g;
    }
    block_sum<2>(p, lds);
    if (threadIdx.x == 0) {
        const double gSg = p[0], mv = p[1];
        const double rho = 0.5 * sqrt(1.0 + 4.0 * (gSg + mv * mv)) - 0.5;
        const double den = 1.0 + rho + mv;
        sh[0] = 1.0 / (1.0 + rho);
        sh[1] = (gSg - mv) / den;
    }
    __syncthreads();
    const double beta = sh[0], c = sh[1];
    for (int i = threadIdx.x; i < D; i += 256) {       // same thread wrote these two entries
        const double d = rb[i], sg = rb[D + i];
        const double dmu = beta * ((sg - d) - c * d);
        rb[D + i] = d + dmu;
        rb[2 * D + i] = dmu;
    }
}

// =====================================================================================
// Rank-2B covariance update, guarded generic version (any D, B, ld, alignment):
//   S = S0 + (1/B) sum_b (d_b d_b^T - e_b e_b^T),  mu = mu0 + (1/B) sum_b dmu_b     (gsm_numpy.py:50-53)
// One workgroup = one 64x64 tile of S, 4 waves of 32x32 (2x2 MFMA tiles).  The factor rows come
// from the records and are staged transposed in LDS ([row][k], padded so the MFMA operand reads
// are bank-conflict free); S0 is prefetched into registers before staging and added to the
// accumulators at the end.  Column tile ct of lane c is column 2c+ct (16 contiguous bytes per lane).
// =====================================================================================
__global__ __launch_bounds__(256) void k_gsm_cov_update(int D, int B, const double* __restrict__ rec, int ldrec,
                                                        const double* __restrict__ mu0,
                                                        const double* __restrict__ S0, int lds0,
                                                        double* __restrict__ S, int lds,
                                                        double* __restrict__ mu_out, int SB, int s_vec_ok,
                                                        int row0, int nrows) {
    // Row-block form: S0/S point at rows [row0, row0+nrows) of the matrix (the full matrix is row0 = 0,
    // nrows = D); tiles cover nrows x D, the records are indexed with global rows.
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int KCH = 2 * SB;
    const int KR = (KCH + 31) & ~31;
    const int RSA = KR + 2, RSB = KR + 1;
    double* FA = smem;
    double* FB = smem + 64 * RSA;

    const int ntiles = (D + 63) >> 6;
    const int ti = blockIdx.x / ntiles, tj = blockIdx.x % ntiles;
    const int I0 = ti * 64, J0 = tj * 64;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int wr = w >> 1, wc = w & 1;
    const double invB = 1.0 / (double)B;

    v2d s0[2][4];
    const int col = J0 + 32 * wc + 2 * c;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = I0 + 32 * wr + 16 * rt + ks + 4 * r;
            if (s_vec_ok && row < nrows && col + 1 < D) {
                s0[rt][r] = *reinterpret_cast<const v2d*>(S0 + (size_t)row * lds0 + col);
            } else {
                v2d t = {0.0, 0.0};
                if (row < nrows && col < D) t.x = S0[(size_t)row * lds0 + col];
                if (row < nrows && col + 1 < D) t.y = S0[(size_t)row * lds0 + col + 1];
                s0[rt][r] = t;
            }
        }

    v4d acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = (v4d){0.0, 0.0, 0.0, 0.0};

    for (int b0 = 0; b0 < B; b0 += SB) {
        const int nb = (B - b0) < SB ? (B - b0) : SB;
        for (int idx = tid; idx < SB * 64; idx += 256) {
            const int bl = idx >> 6, ii = idx & 63, b = b0 + bl;
            double dI = 0.0, eI = 0.0, dJ = 0.0, eJ = 0.0;
            if (bl < nb) {
                const double* rb = rec + (size_t)b * ldrec;
                const int gi = row0 + I0 + ii, gj = J0 + ii;
                if (I0 + ii < nrows) { dI = rb[gi]; eI = rb[D + gi]; }
                if (gj < D) { dJ = rb[gj]; eJ = rb[D + gj]; }
            }
            FA[ii * RSA + bl] = dI;
            FA[ii * RSA + SB + bl] = eI;
            FB[ii * RSB + bl] = dJ * invB;
            FB[ii * RSB + SB + bl] = -eJ * invB;
        }
        __syncthreads();
        const double* a0p = FA + (32 * wr + c) * RSA + ks;
        const double* a1p = a0p + 16 * RSA;
        const double* b0p = FB + (32 * wc + 2 * c) * RSB + ks;
        const double* b1p = b0p + RSB;
        const int nsteps = KCH >> 2;
        for (int s = 0; s < nsteps; ++s) {
            const double a0 = a0p[4 * s], a1 = a1p[4 * s], bb0 = b0p[4 * s], bb1 = b1p[4 * s];
            acc[0][0] = GSMVI_MFMA_F64(a0, bb0, acc[0][0]);
            acc[0][1] = GSMVI_MFMA_F64(a0, bb1, acc[0][1]);
            acc[1][0] = GSMVI_MFMA_F64(a1, bb0, acc[1][0]);
            acc[1][1] = GSMVI_MFMA_F64(a1, bb1, acc[1][1]);
        }
        __syncthreads();
    }

#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = I0 + 32 * wr + 16 * rt + ks + 4 * r;
            v2d o;
            o.x = s0[rt][r].x + acc[rt][0][r];
            o.y = s0[rt][r].y + acc[rt][1][r];
            if (s_vec_ok && row < nrows && col + 1 < D) {
                *reinterpret_cast<v2d*>(S + (size_t)row * lds + col) = o;
            } else {
                if (row < nrows && col < D) S[(size_t)row * lds + col] = o.x;
                if (row < nrows && col + 1 < D) S[(size_t)row * lds + col + 1] = o.y;
            }
        }

    if (ti == 0 && mu_out != nullptr) {            // new mean (all D entries), by the first row of workgroups
        const int gi = J0 + l;
        double part = 0.0;
        if (gi < D)
            for (int b = w; b < B; b += 4) part += rec[(size_t)b * ldrec + 2 * D + gi];
        smem[w * 64 + l] = part;
        __syncthreads();
        if (w == 0 && gi < D)
            mu_out[gi] = mu0[gi] + ((smem[l] + smem[64 + l]) + (smem[128 + l] + smem[192 + l])) * invB;
    }
}

// =====================================================================================
// Commit-or-revert (gsm_numpy.py:121-125), device-side: dst <- src iff *info == 0.
// =====================================================================================
__global__ __launch_bounds__(256) void k_commit(int D, const int* __restrict__ info,
                                                const double* __restrict__ mu_new,
                                                const double* __restrict__ S_new, int lds_new,
                                                double* __restrict__ mu, double* __restrict__ S, int lds,
                                                int* __restrict__ n_reverts) {
    const int bad = *info;
    if (bad != 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && n_reverts) *n_reverts += 1;
        return;
    }
    const size_t n = (size_t)D * D;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
        const size_t r = idx / D, cidx = idx % D;
        S[r * lds + cidx] = S_new[r * lds_new + cidx];
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < D; i += 256) mu[i] = mu_new[i];
}

// ---- launch helpers used by gsmvi_abi.hip --------------------------------------------------
void gsmvi_launch_panel_partial(hipStream_t st, hipEvent_t* ev, int MT, dim3 grid, int D, int ncols, int nrows,
                                const double* A, int lda, const double* shift, double alpha, const double* M,
                                int ldm, double* Pp, int chunks_per_wg, int a_vec_ok) {
    switch (MT) {
        case 1:
            GSMVI_LAUNCH(k_panel_partial<1>, grid, dim3(256), 0, st, ev, D, ncols, nrows, A, lda, shift, alpha, M, ldm, Pp,
                         chunks_per_wg, a_vec_ok);
            break;
        case 2:
            GSMVI_LAUNCH(k_panel_partial<2>, grid, dim3(256), 0, st, ev, D, ncols, nrows, A, lda, shift, alpha, M, ldm, Pp,
                         chunks_per_wg, a_vec_ok);
            break;
        default:
            GSMVI_LAUNCH(k_panel_partial<4>, grid, dim3(256), 0, st, ev, D, ncols, nrows, A, lda, shift, alpha, M, ldm, Pp,
                         chunks_per_wg, a_vec_ok);
            break;
    }
}

void gsmvi_launch_panel_finish(hipStream_t st, hipEvent_t* ev, int D, int nrows, int KC, const double* Pp,
                               const double* addvec, double* Out, int ldo, int ncols_out) {
    GSMVI_LAUNCH(k_panel_finish, dim3((ncols_out + 255) / 256, nrows), dim3(256), 0, st, ev, D, nrows, KC, Pp, addvec,
                 Out, ldo, ncols_out);
}

void gsmvi_launch_gsm_scalars(hipStream_t st, hipEvent_t* ev, int D, int B, int KC, const double* X, int ldx,
                              const double* G, int ldg, const double* mu0, const double* Pp, double* rec,
                              int ldrec) {
    GSMVI_LAUNCH(k_gsm_scalars, dim3(B), dim3(256), 0, st, ev, D, B, KC, X, ldx, G, ldg, mu0, Pp, rec, ldrec);
}

size_t gsmvi_cov_update_lds_bytes(int SB) {
    const int KR = (2 * SB + 31) & ~31;
    return (size_t)64 * (2 * KR + 3) * sizeof(double);
}

hipError_t gsmvi_cov_update_prepare() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_gsm_cov_update),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void gsmvi_launch_gsm_cov_update(hipStream_t st, hipEvent_t* ev, int D, int B, const double* rec, int ldrec,
                                 const double* mu0, const double* S0, int lds0, double* S, int lds, double* mu_out,
                                 int SB, int s_vec_ok, int row0, int nrows) {
    const int nt = (D + 63) / 64, ntr = (nrows + 63) / 64;
    GSMVI_LAUNCH(k_gsm_cov_update, dim3(ntr * nt), dim3(256), gsmvi_cov_update_lds_bytes(SB), st, ev, D, B, rec,
                 ldrec, mu0, S0, lds0, S, lds, mu_out, SB, s_vec_ok, row0, nrows);
}

void gsmvi_launch_commit(hipStream_t st, int D, const int* info, const double* mu_new, const double* S_new,
                         int lds_new, double* mu, double* S, int lds, int* n_reverts) {
    int blocks = (int)(((size_t)D * D + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_commit, dim3(blocks), dim3(256), 0, st, D, info, mu_new, S_new, lds_new, mu, S, lds,
                       n_reverts);
}
