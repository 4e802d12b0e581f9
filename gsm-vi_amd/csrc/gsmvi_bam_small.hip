// BaM's n x n matrix function (n = B since round 3, B + 1 before) on the DEVICE (gsmvi/bam.py:108-110; the reference evaluates it on the host
// through jax.pure_callback + scipy sqrtm, bam.py:15-22).
//
//   BB = ((N + I/4)^(1/2) + I/2)^2 = N + I/2 + (N + I/4)^(1/2),   BB = L L^T,   zg = L^-1 (P gbar + M1^T Vf gbar)
//
// The square root comes from the coupled Newton-Schulz iteration in product form (Higham, Functions of Matrices,
// (6.35); stable, multiplication-only):  with A = N + I/4, s = trace(A),  Y_0 = A / s,  Z_0 = I,
//     M = c^2 Z Y,   T = (3 I - M) / 2,   Y <- c Y T,   Z <- c T Z          =>  Y -> (A/s)^(1/2),  Z -> (A/s)^(-1/2)
// (s = trace(A) >= lambda_max is used as the scale: every workgroup can sum a diagonal by itself)
// Every iterate is a polynomial in A (symmetric, commuting), but the products are taken exactly in the order above:
// that is what makes the iteration stable in floating point.  The eigenvalues mu of Z Y start
// in [1/(4s), 1] (A >= I/4 because N is a Gram matrix) and obey mu <- f(c^2 mu), f(x) = x (3 - x)^2 / 4.  Because
// the lower bound is KNOWN, the scaling c^2 = 3 / (1 + sqrt(l) + l) (Chen & Chow's scaled Newton-Schulz, written
// for mu = sigma^2) is computed from a scalar recurrence alone: small eigenvalues grow ~6.75x per step instead of
// 2.25x, 12-15 steps instead of ~27 at cond(A) ~ 1e7.  The recurrence (and hence the number of steps k*) depends
// on s, which lives on the device: workgroup 0 of step 0 (k_bam_ns_step0) runs it once and stores c_k^2 and k*; the host enqueues the launches of
// k* + 1 steps (k* of the previous call, a pinned word read without synchronising) and the ones beyond k* return at once;
// k_bam_ns_tail makes up for a stale guess.  Two launches per step on many workgroups: M = Z Y, then Y' and Z' together
// (one 16 x 16 block per workgroup, K split over its four waves, fp64 MFMA, operands straight from L2); the first M = Z0 Y0 = Y0
// is never materialised (k_bam_ns_step0).  Round 4 measured a ONE-launch step (every workgroup recomputing the 16-wide panel of M it needs,
// rows of Z staged in LDS): 9.2 us against 2 x 4.55 us -- the n/16-fold recomputation costs what the second launch costs, so it
// was not kept (profiles/r04/c4_chain_ab.txt).  Then, n <= 128: BB (symmetrised) and the factor-independent vectors on many
// workgroups (k_bam_bbav), the Cholesky factorisation WITH the inverse factor in one workgroup (k_bam_cholw: chol64_blk /
// chol128w_body), and Z = L^-1 (...) as an MFMA product in k_bam_zw (gsmvi_bam.hip).  No host synchronisation, no host arithmetic.
#include "gsmvi_common.h"
#include "gsmvi_ctx.h"
#include "gsmvi_chol64.h"
#include "gsmvi_chol64b.h"
#include "gsmvi_chol128.h"
#include "../../include/gsmvi_hip.h"

#define BAMS_NMAX 128                // largest n of the one-workgroup Cholesky k_bam_cholw; above it: blocked potrf + k_bam_post_big
#define BAMS_NBIG 1024               // largest n altogether: k_bam_post_big's substitution owns one entry per thread of its 1024-thread workgroup
                                     // (round 6; 640 until then: LDS of the forward substitution kernel at 16 columns per workgroup -- now 8 above 640)
#define BAMS_LD 144                  // padded leading dimension of the iteration matrices for n <= 128 (9 blocks of 16: n = B + 1 <= 129 until round 3);
                                     // larger n: n rounded up to 16 (passed to the kernels as `ld`)
#define BAMS_KMAX 32                 // launches enqueued; k* <= BAMS_KMAX is checked on the device (else flagged)
// coef layout (doubles): [0..KMAX) c_k^2, [40] k*, [41] s, [42] 1 if s is not finite or the bound did not close in KMAX steps
// (KMAX = 32 scaled steps cover cond(A) up to ~1e20, beyond what fp64 can represent in N + I/4)

// ---- one 16 x 16 block per wave of  C = op(A) op(B)  on BAMS_LD x BAMS_LD matrices ----------------------------------
//   MODE 0: C = A B      MODE 1: C = A T(B)      MODE 2: C = T(A) B,     T(M) = 1.5 I - 0.5 c2 M  applied on the fly.
// No symmetry is assumed: the stability of the coupled iteration depends on the exact products Y T and T Z (with the
// order swapped -- Z T, equal in exact arithmetic -- rounding errors grow and the iteration diverges at
// cond(A) ~ 1e6: checked in numpy and on the device).  A is read by rows (16 rows x 4 consecutive k per k-step:
// 32-B segments, the matrices are L2-resident), B by rows of 16 consecutive columns (128-B segments).
// The partial block of ONE wave: wave w of four takes the k-steps w, w + 4, ... in batches of nine (one batch for ld = 144),
// every load of a batch issued together, two accumulator chains.  Shared by the multi-workgroup step kernels and the tail
// kernel (round 5): the tail used its own single-wave sum, so a call whose stale step-count hint moved steps into the tail
// gave other bits than a call with a fresh hint -- results depended on host / GPU timing (advisor, round 4).
template <int MODE>
__device__ __forceinline__ v4d bams_partial(const double* __restrict__ A, const double* __restrict__ Bm, int i0, int j0, int w,
                                            int nk, double c2, int ld) {
    const int l = threadIdx.x & 63, cc = l & 15, ks = l >> 4;
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int u0 = 0; w + 4 * u0 < nk; u0 += 9) {
        double a[9], b[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int st = w + 4 * (u0 + u);
            const int k = 4 * st + ks;
            const int kc = k < ld ? k : ld - 1;
            const double av = A[(size_t)(i0 + cc) * ld + kc];
            const double bv = Bm[(size_t)kc * ld + j0 + cc];
            a[u] = MODE == 2 ? ((kc == i0 + cc ? 1.5 : 0.0) - 0.5 * c2 * av) : av;
            b[u] = MODE == 1 ? ((kc == j0 + cc ? 1.5 : 0.0) - 0.5 * c2 * bv) : bv;
            if (st >= nk) { a[u] = 0.0; b[u] = 0.0; }
        }
#pragma unroll
        for (int u = 0; u + 1 < 9; u += 2) {
            acc0 = GSMVI_MFMA_F64(a[u], b[u], acc0);
            acc1 = GSMVI_MFMA_F64(a[u + 1], b[u + 1], acc1);
        }
        acc0 = GSMVI_MFMA_F64(a[8], b[8], acc0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc0[r] += acc1[r];
    return acc0;
}

template <int MODE>
__device__ __forceinline__ void bams_block(const double* __restrict__ A, const double* __restrict__ Bm,
                                           double* __restrict__ Out, int blk, int nb, int nk, double c2, double scale,
                                           int ld) {
    // one WORKGROUP per 16 x 16 block; the four waves' partial blocks are summed through LDS in a fixed order
    __shared__ double red[4 * 256];
    const int i0 = (blk / nb) * 16, j0 = (blk % nb) * 16;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, cc = l & 15, ks = l >> 4;
    const v4d p = bams_partial<MODE>(A, Bm, i0, j0, w, nk, c2, ld);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w * 256 + (ks + 4 * r) * 16 + cc] = p[r];
    __syncthreads();
    const int t = threadIdx.x;                               // element (t >> 4, t & 15) of the block
    const double v = (red[t] + red[256 + t]) + (red[512 + t] + red[768 + t]);
    Out[(size_t)(i0 + (t >> 4)) * ld + j0 + (t & 15)] = scale * v;
}

// ---- step 0 WITHOUT a launch in front of it (round 4): s, Y0 = (N + I/4)/s, Z0 = I and M0 = Z0 Y0 = Y0 are not materialised --
// A preparation kernel used to write Y0, Z0 and M0 and run the scalar recurrence, then step 0 read them back: one dependent launch (2 us before
// its first instruction + 1 us for its stores, scripts/nsbench.hip) for 3 ld^2 doubles nobody else reads.  Here every workgroup
// of step 0 sums the diagonal itself (as the prep kernel's workgroups did), forms Y0 and T0 = 1.5 I - 0.5 c0^2 Y0 in its operand
// loads, and Z1 = c0 T0 Z0 = c0 T0 is an elementwise write (the product T0 I adds exact zeros: same bits); workgroup 0 also runs
// the recurrence and writes coef / the host hint for the launches behind it.  Same operations in the same order as
// that pair of launches: bit-identical iterates.  s = trace(N + I/4) >= lambda_max is the scale (every workgroup can sum a
// diagonal by itself); a NaN / inf anywhere in N needs no flag of its own: it propagates through the products into BB, where
// k_bam_cholw rejects it.
__global__ __launch_bounds__(256) void k_bam_ns_step0(int n, int ld, const double* __restrict__ Nm, double* __restrict__ Yo,
                                                      double* __restrict__ Zo, double* __restrict__ coef,
                                                      int* __restrict__ hint_host) {
    __shared__ double red[4 * 256];
    const int tid = threadIdx.x;
    double tr = 0.0;
    for (int i = tid; i < n; i += 256) tr += Nm[(size_t)i * n + i] + 0.25;
    tr = wave_sum(tr);
    if ((tid & 63) == 0) red[tid >> 6] = tr;
    __syncthreads();
    const double s = (red[0] + red[1]) + (red[2] + red[3]);
    const double sinv = 1.0 / s;
    __syncthreads();                                         // red is reused by the block reduction below
    double l0 = 0.25 * sinv;                                 // lower bound of the eigenvalues of Z Y (A >= I/4)
    if (!(l0 > 0.0) || l0 > 1.0) l0 = 1.0;
    const double c2 = (l0 < 0.25) ? 3.0 / (1.0 + sqrt(l0) + l0) : 1.0, c = sqrt(c2);
    if (blockIdx.x == 0 && tid == 0) {                       // the scaling recurrence
        double l = l0;
        const bool s_ok = (s == s) && s > 0.0 && s < 1e300;
        int kstar = BAMS_KMAX + 1;
        for (int k = 0; k < BAMS_KMAX; ++k) {
            const double c2k = (l < 0.25) ? 3.0 / (1.0 + sqrt(l) + l) : 1.0;     // scale only while it pays
            coef[k] = c2k;
            const double x = c2k * l;
            l = x * (3.0 - x) * (3.0 - x) * 0.25;
            if (l > 1.0) l = 1.0;
            if (1.0 - l < 5e-9 && kstar > BAMS_KMAX) kstar = k + 2;              // e -> 0.75 e^2: one more step gives < 1e-16
        }
        coef[42] = (!s_ok || kstar > BAMS_KMAX) ? 1.0 : 0.0;  // cond(A) beyond ~1e12: not reachable in BAMS_KMAX steps
        if (kstar > BAMS_KMAX) kstar = BAMS_KMAX;
        coef[40] = (double)kstar;
        coef[41] = s;
        if (hint_host) *hint_host = kstar;                   // pinned host word: how many steps the NEXT call should enqueue
    }
    const int nb = (n + 15) >> 4, nk = (n + 3) >> 2;
    auto y0 = [&](int i, int k) { return (i < n && k < n) ? (Nm[(size_t)i * n + k] + (i == k ? 0.25 : 0.0)) * sinv : 0.0; };
    if ((int)blockIdx.x >= nb * nb) {                        // Z1 = c0 T0 (Z0 = I), zero outside n x n
        const int blk = blockIdx.x - nb * nb, i = (blk / nb) * 16 + (tid >> 4), j = (blk % nb) * 16 + (tid & 15);
        const double t0 = (i == j ? 1.5 : 0.0) - 0.5 * c2 * y0(i, j);
        Zo[(size_t)i * ld + j] = (i < n && j < n) ? c * t0 : 0.0;
        return;
    }
    // Y1 = c0 Y0 T0: bams_block<1> with the operands formed from N (same batches, same accumulator chains, same reduction)
    const int blk = blockIdx.x, i0 = (blk / nb) * 16, j0 = (blk % nb) * 16;
    const int w = tid >> 6, l = tid & 63, cc = l & 15, ks = l >> 4;
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int u0 = 0; w + 4 * u0 < nk; u0 += 9) {
        double a[9], b[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int st = w + 4 * (u0 + u);
            const int k = 4 * st + ks;
            const int kc = k < ld ? k : ld - 1;
            a[u] = y0(i0 + cc, kc);
            b[u] = (kc == j0 + cc ? 1.5 : 0.0) - 0.5 * c2 * y0(kc, j0 + cc);
            if (st >= nk) { a[u] = 0.0; b[u] = 0.0; }
        }
#pragma unroll
        for (int u = 0; u + 1 < 9; u += 2) {
            acc0 = GSMVI_MFMA_F64(a[u], b[u], acc0);
            acc1 = GSMVI_MFMA_F64(a[u + 1], b[u + 1], acc1);
        }
        acc0 = GSMVI_MFMA_F64(a[8], b[8], acc0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w * 256 + (ks + 4 * r) * 16 + cc] = acc0[r] + acc1[r];
    __syncthreads();
    const double v = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
    Yo[(size_t)(i0 + (tid >> 4)) * ld + j0 + (tid & 15)] = c * v;
}

// M = Z Y (the scaling enters through T in the step kernel)
// (Round 4 tried dropping the early return -- operand loads issued before coef[40] is known, stores predicated: 4.70 / 5.05 us
// per launch against 4.54 / 4.57 with it; kept.)
__global__ __launch_bounds__(256) void k_bam_ns_zy(int n, int ld, int k, const double* __restrict__ Ya,
                                                   const double* __restrict__ Za, const double* __restrict__ Yb,
                                                   const double* __restrict__ Zb, double* __restrict__ Mm,
                                                   const double* __restrict__ coef) {
    if ((double)k >= coef[40] || coef[42] != 0.0) return;
    const double* Y = (k & 1) ? Yb : Ya;
    const double* Z = (k & 1) ? Zb : Za;
    const int nb = (n + 15) >> 4, nk = (n + 3) >> 2;
    bams_block<0>(Z, Y, Mm, blockIdx.x, nb, nk, 0.0, 1.0, ld);
}

// Y' = c Y T and Z' = c T Z (in THIS order), T = 1.5 I - 0.5 c^2 M.  Blocks [0, nb^2) -> Y', the rest -> Z'.
__global__ __launch_bounds__(256) void k_bam_ns_step(int n, int ld, int k, double* __restrict__ Ya, double* __restrict__ Za,
                                                     double* __restrict__ Yb, double* __restrict__ Zb,
                                                     const double* __restrict__ Mm, const double* __restrict__ coef) {
    if ((double)k >= coef[40] || coef[42] != 0.0) return;
    const double c2 = coef[k], c = sqrt(c2);
    const double* Yi = (k & 1) ? Yb : Ya;
    const double* Zi = (k & 1) ? Zb : Za;
    double* Yo = (k & 1) ? Ya : Yb;
    double* Zo = (k & 1) ? Za : Zb;
    const int nb = (n + 15) >> 4, nk = (n + 3) >> 2;
    if ((int)blockIdx.x < nb * nb) bams_block<1>(Yi, Mm, Yo, blockIdx.x, nb, nk, c2, c, ld);         // Y' = c Y T
    else bams_block<2>(Mm, Zi, Zo, blockIdx.x - nb * nb, nb, nk, c2, c, ld);                        // Z' = c T Z
}

// ---- safety net behind the enqueued steps --------------------------------------------------------------------------
// The host enqueues kenq <= BAMS_KMAX multi-workgroup steps, guessed from the k* of the previous call (a pinned host
// word written by k_bam_ns_step0: no synchronisation, possibly stale).  If this call's k* turns out larger, the missing
// steps kenq .. k*-1 are executed HERE by one workgroup -- slow (one CU) but exact, so a stale guess costs time, never
// correctness; normally the kernel returns at once.  Same products, same order, same ping-pong parity.
// Round 5: BIT-IDENTICAL to the multi-workgroup steps.  The 16 waves form four groups of four; a group plays one workgroup of
// k_bam_ns_zy / k_bam_ns_step: wave q of the group computes the partial block of wave q there (bams_partial: same k-steps, same
// batches, same accumulator chains), the four partials meet in the group's LDS slice and are summed in the same order.  So
// (mu, S) no longer depend on how stale the hint was -- what the sharded fits' "replicas stay bit-identical" rests on.
// (the barriers sit OUTSIDE every group-dependent branch: all 16 waves execute the same two barriers per block slot)
__device__ __forceinline__ void bams_group_finish(const v4d& p, bool active, double* Out, int i0, int j0, double scale, int ld,
                                                  double* red /* this group's 4 x 256 */) {
    const int q = (threadIdx.x >> 6) & 3, l = threadIdx.x & 63, cc = l & 15, ks = l >> 4;
    if (active) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[q * 256 + (ks + 4 * r) * 16 + cc] = p[r];
    }
    __syncthreads();
    if (active) {
        const int t = threadIdx.x & 255;
        const double v = (red[t] + red[256 + t]) + (red[512 + t] + red[768 + t]);
        Out[(size_t)(i0 + (t >> 4)) * ld + j0 + (t & 15)] = scale * v;
    }
    __syncthreads();                                         // the slice is reused by the next block
}

__global__ __launch_bounds__(1024) void k_bam_ns_tail(int n, int ld, int kenq, double* Ya, double* Za, double* Yb, double* Zb,
                                                      double* Mm, const double* coef) {
    const int kstar = (int)coef[40];
    if (kenq >= kstar || coef[42] != 0.0) return;            // the usual case: nothing left to do
    __shared__ double red_all[4 * 4 * 256];
    const int nb = (n + 15) >> 4, nk = (n + 3) >> 2, g = threadIdx.x >> 8, q = (threadIdx.x >> 6) & 3;
    const int nb2 = nb * nb;
    double* red = red_all + g * 1024;
    for (int k = kenq; k < kstar; ++k) {
        const double c2 = coef[k], c = sqrt(c2);
        double* Yi = (k & 1) ? Yb : Ya;
        double* Zi = (k & 1) ? Zb : Za;
        double* Yo = (k & 1) ? Ya : Yb;
        double* Zo = (k & 1) ? Za : Zb;
        for (int base = 0; base < nb2; base += 4) {          // M = Z Y   (block-uniform trip count: every wave reaches every barrier)
            const int blk = base + g;
            const bool act = blk < nb2;
            const int i0 = (blk / nb) * 16, j0 = (blk % nb) * 16;
            v4d p = {0.0, 0.0, 0.0, 0.0};
            if (act) p = bams_partial<0>(Zi, Yi, i0, j0, q, nk, 0.0, ld);
            bams_group_finish(p, act, Mm, i0, j0, 1.0, ld, red);
        }
        __threadfence_block();
        __syncthreads();
        for (int base = 0; base < 2 * nb2; base += 4) {      // blocks [0, nb^2): Y' = c Y T, the rest: Z' = c T Z
            const int blk = base + g;
            const bool act = blk < 2 * nb2, isy = blk < nb2;
            const int bq = isy ? blk : blk - nb2;
            const int i0 = (bq / nb) * 16, j0 = (bq % nb) * 16;
            v4d p = {0.0, 0.0, 0.0, 0.0};
            if (act) p = isy ? bams_partial<1>(Yi, Mm, i0, j0, q, nk, c2, ld) : bams_partial<2>(Mm, Zi, i0, j0, q, nk, c2, ld);
            bams_group_finish(p, act, isy ? Yo : Zo, i0, j0, c, ld, red);
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ---- the side workgroup of the one-workgroup chains (n <= 64; orthogonal basis of the factor-form BaM update, gsmvi_bam.hip) ----
// Next to BaM's B x B chain and INDEPENDENT of it the basis needs
//   Gvv = Vw Vw^T,  [Gvv | I] -> [R11 | W11],  T = W11 M1,  M1' = -W11^T T = -Gvv^-1 M1,  Dm = M1 - M1'.
// For n <= 64 the chain is ONE workgroup (k_bam_small48, k_bam_ns64: 30 - 100 us on one CU), so this runs as the SECOND workgroup
// of its launch, ~12 - 25 us on another CU, instead of three launches behind the chain (factorisation 9 - 17 us + two products,
// round-5 first form).  Nothing crosses between the two workgroups: the side workgroup reads Gvv and M1 from what the launch
// BEFORE produced (SRC: the Gram slabs for k_bam_small48, summed in the chain's order so that M1 has the chain's bits; the
// finished matrices of k_bam_nmat2 for k_bam_ns64).  What depends on the chain's L (Pi = L^-1 Dm^T, the correction of vg)
// comes later (k_bam_zw's launch / k_bamf_pi_vg).
#define BAMQ_SIDE_ES 146
#define BAMQ_SIDE_LS 65
#define BAMQ_SIDE_DOUBLES (64 * BAMQ_SIDE_ES + 64 * BAMQ_SIDE_LS)
struct bamq_src_slabs {                                      // element (r, c) of the stacked product = sum of kc slabs
    const double* slabs;
    int kc, ldslab, n;
    long long stride;
    __device__ __forceinline__ double at(int r, int c) const {
        double t[GSMVI_MAX_KC];
#pragma unroll
        for (int k = 0; k < GSMVI_MAX_KC; ++k) t[k] = slabs[(size_t)(k < kc ? k : kc - 1) * stride + (size_t)r * ldslab + c];
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < GSMVI_MAX_KC; ++k) a += (k < kc) ? t[k] : 0.0;
        return a;
    }
    __device__ __forceinline__ double gvv(int i, int j) const { return at(n + i, n + j); }
    __device__ __forceinline__ double m1(int i, int j) const { return at(n + i, j); }
};
struct bamq_src_mats {                                       // finished n x n matrices (k_bam_nmat2)
    const double *G11, *M1;
    int n;
    __device__ __forceinline__ double gvv(int i, int j) const { return G11[(size_t)i * n + j]; }
    __device__ __forceinline__ double m1(int i, int j) const { return M1[(size_t)i * n + j]; }
};
template <class SRC>
__device__ __forceinline__ void bamq_side_body(int n, const SRC& src, double* sm, double* scr, int* sh_f, const bamq_side& sd) {
    constexpr int ES = BAMQ_SIDE_ES, LS = BAMQ_SIDE_LS;
    double* E = sm;                                          // 64 x 146: [Gvv | I] -> [R11 | W11]; then M1 / Dm over R11
    double* Ts = sm + 64 * ES;
    const int tid = threadIdx.x;
    for (int e0 = 0; e0 < 64 * 64; e0 += 512 * 4) {          // Gvv: the upper triangle, identity beyond n
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 512 * u + tid, i = e >> 6, j = e & 63;
            v[u] = src.gvv(i < n ? i : n - 1, j < n ? j : n - 1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 512 * u + tid, i = e >> 6, j = e & 63;
            E[i * ES + j] = (i < n && j < n) ? (j >= i ? v[u] : 0.0) : (i == j ? 1.0 : 0.0);
        }
    }
    __syncthreads();
    chol64_blk<ES, false, 1>(E, scr, n, sh_f);               // (plain rule: dependent draws are a failure, not a drop)
    __syncthreads();
    // ... and so are NEARLY dependent ones.  Everything downstream leans on M1 + Gvv M1' = 0 (Zt orthogonal to the draws; the
    // 2B x 2B chain even takes the Gram matrix as block diagonal), which holds to eps cond(Gvv): for i.i.d. normal draws
    // cond(Gvv) <= ((sqrt(D) + sqrt(B)) / (sqrt(D) - sqrt(B)))^2 < 40 at 2B <= D, while a caller's almost repeated sample would
    // buy a silently inaccurate update.  cond(Gvv) >= (max R_ii / min R_ii)^2: beyond 1e8 the factorisation counts as failed
    // (flag, revert) -- the update then is never less accurate than 1e-8, flagged or not.
    if (tid < 64) {
        const double d = tid < n ? E[tid * ES + tid] : 0.0;
        double dmax = d, dmin = tid < n ? d : 1.7976931348623157e308;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const double a = __shfl_xor(dmax, m, 64), b = __shfl_xor(dmin, m, 64);
            dmax = a > dmax ? a : dmax;
            dmin = b < dmin ? b : dmin;
        }
        if (tid == 0 && *sh_f == 0 && !(dmin > 1e-4 * dmax)) *sh_f = n;
    }
    __syncthreads();
    if (tid == 0) *sd.info1 = *sh_f;
    if (sd.R11) {                                            // for the 2B x 2B chain: its first diagonal block is this one
        for (int e = tid; e < n * n; e += 512) {
            const int i = e / n, j = e - i * n;
            sd.R11[e] = (j >= i) ? E[i * ES + j] : 0.0;
            sd.W11[e] = (j <= i) ? E[i * ES + 64 + j] : 0.0;
        }
        __syncthreads();
    }
    for (int e0 = 0; e0 < 64 * 64; e0 += 512 * 4) {          // M1 over R11 (LDS copy: nobody reads it for n <= 64)
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 512 * u + tid, i = e >> 6, j = e & 63;
            v[u] = src.m1(i < n ? i : n - 1, j < n ? j : n - 1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 512 * u + tid, i = e >> 6, j = e & 63;
            E[i * ES + j] = v[u];
        }
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += 512) {                 // T = W11 M1 (W11 lower: k <= i); four independent partial sums
        const int i = e / n, j = e - i * n;
        const double* wr = E + i * ES + 64;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        int k = 0;
        for (; k + 3 <= i; k += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += wr[k + u] * E[(k + u) * ES + j];
        }
        for (; k <= i; ++k) a[0] += wr[k] * E[k * ES + j];
        Ts[i * LS + j] = (a[0] + a[1]) + (a[2] + a[3]);
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += 512) {                 // M1' = -W11^T T (k >= i), Dm = M1 - M1'
        const int i = e / n, j = e - i * n;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        int k = i;
        for (; k + 3 < n; k += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += E[(k + u) * ES + 64 + i] * Ts[(k + u) * LS + j];
        }
        for (; k < n; ++k) a[0] += E[k * ES + 64 + i] * Ts[k * LS + j];
        const double v = (a[0] + a[1]) + (a[2] + a[3]);
        sd.M1p[e] = -v;
        sd.Dm[e] = E[i * ES + j] + v;
    }
}

// ---- the whole iteration in ONE workgroup for n <= 64 (round 4) ------------------------------------------------------------------
// Y, Z and M = Z Y fit in LDS together ([64][66] doubles each, 101 KB), so a 64 x 64 problem needs no launch per product: the
// multi-workgroup form pays 2 launches x ~4.2 us per step whatever the size (scripts/nsbench.hip: n = 64 costs what n = 128
// costs), 125 us for 15 steps; one CU does the three 64^3 products of a step in ~3.5 us.  Eight waves: M = Z Y two blocks per
// wave; then waves 0-3 own a ROW panel of Y' = c Y T (its A fragments in registers, T from M in LDS) and waves 4-7 a COLUMN
// panel of Z' = c T Z, so both are updated in place behind one barrier.  Same recurrence, same T, same product order (Y T and
// T Z) as the multi-workgroup kernels; no step-count hint and no tail kernel: k* is known where the loop runs.  The final
// iterate goes to the buffer its parity names (k_bam_bbav / k_bam_ns_bb read Y from (k* & 1) ? Yb : Ya), coef[40..42] as usual.
template <bool PAIR>   // PAIR: a second workgroup runs bamq_side_body on the matrices of k_bam_nmat2
__global__ __launch_bounds__(512) void k_bam_ns64(int n, int ld, const double* __restrict__ Nm, double* __restrict__ Ya,
                                                  double* __restrict__ Yb, double* __restrict__ coef, bamq_src_mats src,
                                                  bamq_side sd) {
    constexpr int LS = 66;
    constexpr int SMD = (PAIR && BAMQ_SIDE_DOUBLES > 3 * 64 * LS) ? BAMQ_SIDE_DOUBLES : 3 * 64 * LS;
    __shared__ __attribute__((aligned(16))) double sm[SMD];
    __shared__ __attribute__((aligned(16))) double scr1[PAIR ? CHOLB_SCRATCH_DOUBLES(1) : 2];
    __shared__ double red[8], cf[BAMS_KMAX + 4];
    __shared__ int sh_f;
    if (PAIR && blockIdx.x == 1) {                           // block-uniform
        bamq_side_body(n, src, sm, scr1, &sh_f, sd);
        return;
    }
    double *Ys = sm, *Zs = sm + 64 * LS, *Ms = sm + 2 * 64 * LS;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, cc = l & 15, ks = l >> 4;
    double tr = 0.0;
    if (tid < n) tr = Nm[(size_t)tid * n + tid] + 0.25;
    tr = wave_sum(tr);
    if (l == 0) red[w] = tr;
    __syncthreads();
    const double s = red[0];                                 // n <= 64: the diagonal lives in wave 0
    const double sinv = 1.0 / s;
    for (int e = tid; e < 64 * 64; e += 512) {
        const int i = e >> 6, j = e & 63;
        const bool in = i < n && j < n;
        const double v = in ? (Nm[(size_t)i * n + j] + (i == j ? 0.25 : 0.0)) * sinv : 0.0;
        Ys[i * LS + j] = v;
        Ms[i * LS + j] = v;                                  // M0 = Z0 Y0 = Y0
        Zs[i * LS + j] = (in && i == j) ? 1.0 : 0.0;
    }
    if (tid == 0) {                                          // the scaling recurrence (as k_bam_ns_step0)
        double lq = 0.25 * sinv;
        const bool s_ok = (s == s) && s > 0.0 && s < 1e300;
        if (!(lq > 0.0) || lq > 1.0) lq = 1.0;
        int kstar = BAMS_KMAX + 1;
        for (int k = 0; k < BAMS_KMAX; ++k) {
            const double c2k = (lq < 0.25) ? 3.0 / (1.0 + sqrt(lq) + lq) : 1.0;
            cf[k] = c2k;
            coef[k] = c2k;
            const double x = c2k * lq;
            lq = x * (3.0 - x) * (3.0 - x) * 0.25;
            if (lq > 1.0) lq = 1.0;
            if (1.0 - lq < 5e-9 && kstar > BAMS_KMAX) kstar = k + 2;
        }
        const double failed = (!s_ok || kstar > BAMS_KMAX) ? 1.0 : 0.0;
        if (kstar > BAMS_KMAX) kstar = BAMS_KMAX;
        cf[BAMS_KMAX] = (double)kstar;
        cf[BAMS_KMAX + 1] = failed;
        coef[40] = (double)kstar;
        coef[41] = s;
        coef[42] = failed;
    }
    __syncthreads();
    const int kstar = (int)cf[BAMS_KMAX];
    if (cf[BAMS_KMAX + 1] != 0.0) return;                    // flagged: k_bam_bbav poisons BB
    for (int k = 0; k < kstar; ++k) {
        const double c2 = cf[k], c = sqrt(c2);
        if (k > 0) {                                         // M = Z Y: blocks (bi, 2 bjp), (bi, 2 bjp + 1)
            const int bi = w >> 1, bj = 2 * (w & 1);
            v4d m0 = {0.0, 0.0, 0.0, 0.0}, m1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const double a = Zs[(16 * bi + cc) * LS + 4 * st + ks];
                m0 = GSMVI_MFMA_F64(a, Ys[(4 * st + ks) * LS + 16 * bj + cc], m0);
                m1 = GSMVI_MFMA_F64(a, Ys[(4 * st + ks) * LS + 16 * bj + 16 + cc], m1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Ms[(16 * bi + ks + 4 * r) * LS + 16 * bj + cc] = m0[r];
                Ms[(16 * bi + ks + 4 * r) * LS + 16 * bj + 16 + cc] = m1[r];
            }
            __syncthreads();
        }
        v4d acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = (v4d){0.0, 0.0, 0.0, 0.0};
        if (w < 4) {                                         // Y'(row panel w) = c Y T
            double a[16];
#pragma unroll
            for (int st = 0; st < 16; ++st) a[st] = Ys[(16 * w + cc) * LS + 4 * st + ks];
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int kk = 4 * st + ks;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = 16 * q + cc;
                    const double t = (kk == j ? 1.5 : 0.0) - 0.5 * c2 * Ms[kk * LS + j];
                    acc[q] = GSMVI_MFMA_F64(a[st], t, acc[q]);
                }
            }
        } else {                                             // Z'(column panel w - 4) = c T Z
            const int cp = w - 4;
            double b[16];
#pragma unroll
            for (int st = 0; st < 16; ++st) b[st] = Zs[(4 * st + ks) * LS + 16 * cp + cc];
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int kk = 4 * st + ks;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 16 * q + cc;
                    const double t = (i == kk ? 1.5 : 0.0) - 0.5 * c2 * Ms[i * LS + kk];
                    acc[q] = GSMVI_MFMA_F64(t, b[st], acc[q]);
                }
            }
        }
        __syncthreads();                                     // every read of Y, Z, M of this step is done
        if (w < 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ys[(16 * w + ks + 4 * r) * LS + 16 * q + cc] = c * acc[q][r];
        } else {
            const int cp = w - 4;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) Zs[(16 * q + ks + 4 * r) * LS + 16 * cp + cc] = c * acc[q][r];
        }
        __syncthreads();
    }
    double* Yo = (kstar & 1) ? Yb : Ya;
    for (int e = tid; e < n * n; e += 512) {
        const int i = e / n, j = e - i * n;
        Yo[(size_t)i * ld + j] = Ys[i * LS + j];
    }
}

// BB = N + I/2 + sqrt(s) sym(Y_final)   (n x n, row-major, ld n)
__global__ __launch_bounds__(256) void k_bam_ns_bb(int n, int ld, const double* __restrict__ Nm, const double* __restrict__ Ya,
                                                   const double* __restrict__ Yb, const double* __restrict__ coef,
                                                   double* __restrict__ BBg) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e % n;
    const int kstar = (int)coef[40];
    const double* Y = (kstar & 1) ? Yb : Ya;                 // iterate k* lives in buffer k* & 1
    const double rs = sqrt(coef[41]);
    const double y = 0.5 * (Y[(size_t)i * ld + j] + Y[(size_t)j * ld + i]);
    BBg[e] = (coef[42] != 0.0) ? __longlong_as_double(0x7ff8000000000000LL) : Nm[e] + (i == j ? 0.5 : 0.0) + rs * y;
}

// ---- BB and the vectors that do not depend on its factor, on many workgroups (round 4) ----------------------------------------
//   BB = N + I/2 + sqrt(s) sym(Y_final)     one 16 x 16 block per workgroup; the transposed block of Y passes through LDS, so both
//                                           reads are 128-byte row segments (a one-workgroup version read Y by columns: 8 us)
//   vg = Vf gbar = M1[:, n-1] / r1s,  a = P gbar + M1^T vg (bam.py:107 applied to gbar)      nb extra workgroups, 16 entries each
// Outputs: BBg (n x n); behind the n x n slot of W: [a (n) | (n unused) | vg (n)].  A failed iteration (coef[42]) poisons BB.
__global__ __launch_bounds__(256) void k_bam_bbav(int n, int ld, bam_reg regs, const double* __restrict__ Nm,
                                                  const double* __restrict__ Ya, const double* __restrict__ Yb,
                                                  const double* __restrict__ coef, const double* __restrict__ M1,
                                                  const double* __restrict__ N0, double* __restrict__ BBg,
                                                  double* __restrict__ tail3) {
    const double reg = regs.get();
    const int nb = (n + 15) >> 4, tid = threadIdx.x;
    if ((int)blockIdx.x < nb * nb) {
        __shared__ double Tt[16 * 17];
        const int bi = blockIdx.x / nb, bj = blockIdx.x - bi * nb, r = tid >> 4, c = tid & 15;
        const int kstar = (int)coef[40];
        const double* Y = (kstar & 1) ? Yb : Ya;             // iterate k* lives in buffer k* & 1
        const int i = 16 * bi + r, j = 16 * bj + c;
        const double y1 = Y[(size_t)i * ld + j];             // (padded to ld >= 16 nb: always inside the buffer)
        const double y2 = Y[(size_t)(16 * bj + r) * ld + 16 * bi + c];      // element (r, c) of block (bj, bi)
        const double nv = (i < n && j < n) ? Nm[(size_t)i * n + j] : 0.0;
        Tt[r * 17 + c] = y2;
        __syncthreads();
        if (i < n && j < n) {
            const double x = nv + (i == j ? 0.5 : 0.0) + sqrt(coef[41]) * (0.5 * (y1 + Tt[c * 17 + r]));
            BBg[(size_t)i * n + j] = (coef[42] != 0.0) ? __longlong_as_double(0x7ff8000000000000LL) : x;
        }
        return;
    }
    // workgroups nb^2 .. nb^2 + nb - 1: sixteen entries of a (and of vg) each; sixteen threads per entry, eight loads in flight each
    __shared__ double part[16 * 17];
    const int e = blockIdx.x - nb * nb, p = 16 * e + (tid & 15), q = tid >> 4, pc = p < n ? p : n - 1;
    const double r1s = sqrt(reg / (1.0 + reg));
    double m[8], v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = q + 16 * u, kc = k < n ? k : n - 1;
        m[u] = M1[(size_t)kc * n + pc];
        v[u] = M1[(size_t)kc * n + (n - 1)];                 // vg[k] r1s
    }
    const double n0 = N0[(size_t)pc * n + (n - 1)], vgp = M1[(size_t)pc * n + (n - 1)];
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
        a0 += (q + 16 * u < n) ? m[u] * (v[u] / r1s) : 0.0;
        a1 += (q + 16 * u + 16 < n) ? m[u + 1] * (v[u + 1] / r1s) : 0.0;
    }
    part[q * 17 + (tid & 15)] = a0 + a1;
    __syncthreads();
    if (tid < 16 && p < n) {
        double sum = 0.0;
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) sum += part[qq * 17 + tid];
        tail3[p] = n0 / r1s + sum;
        tail3[2 * n + p] = vgp / r1s;
    }
}

// ---- Cholesky of BB WITH the inverse factor, stored transposed: Wt = R^-1 = (L^-1)^T (upper), ONE workgroup, n <= 128 (round 4) ----
// Replaced round 3's k_bam_chol_out (barrier-per-pivot chol64_rows_s, 73 us): [BB | I] -> [R | W] on chol64_blk (n <= 64: one call;
// 64 < n <= 128: chol128w_body, the 2 x 2 block scheme of the factor path's Gram matrix, 47 us), and what bam.py:110 calls
// solve(BB, .) becomes Z = W (P + M1^T Vf), an MFMA product in k_bam_zw (gsmvi_bam.hip) instead of a 128-step substitution per
// column of D (33 us at D = 1024).  W = L^-1 with cond(L) = sqrt(cond(BB)) <= ~1e4 on BASELINE config 4: the explicit
// triangular inverse changes Z by 1e-12 relative and the update's backward error from 1.3e-17 to 3e-17 (numpy check beside
// the substitution; K8 asserts 1e-14).
// *info != 0 (a failing pivot, NaN in BB) poisons Wt and the vectors behind it with NaN: nothing stale may be applied.
template <bool BIG>   // BIG: 64 < n <= 128 (two block rows), else n <= 64 (the discarded branch's LDS arrays are not instantiated)
__global__ __launch_bounds__(512) void k_bam_cholw(int n, const double* __restrict__ BBg, double* Rg, double* Wt,
                                                   int* __restrict__ info) {
    __shared__ int sh_info;
    const int tid = threadIdx.x;
    if constexpr (BIG) {
        chol128w_body<false, true>(n, BBg, n, Rg, n, Wt, n, info, &sh_info);
    } else {
        constexpr int ES = 146;
        __shared__ __attribute__((aligned(16))) double E[64 * ES];
        __shared__ __attribute__((aligned(16))) double scr1[CHOLB_SCRATCH_DOUBLES(1)];
        __shared__ int sf;
        for (int e = tid; e < 64 * 64; e += 512) {
            const int i = e >> 6, j = e & 63;
            const bool in = i < n && j < n;
            E[i * ES + j] = in ? (j >= i ? BBg[(size_t)i * n + j] : 0.0) : (i == j ? 1.0 : 0.0);
        }
        __syncthreads();
        chol64_blk<ES, false, 1>(E, scr1, n, &sf);
        for (int e = tid; e < n * n; e += 512) {
            const int i = e / n, j = e - i * n;
            Wt[e] = (i <= j) ? E[j * ES + 64 + i] : 0.0;   // Wt[i][j] = W[j][i]
            Rg[e] = (j >= i) ? E[i * ES + j] : 0.0;
        }
        if (tid == 0) { *info = sf; sh_info = sf; }
    }
    __syncthreads();
    if (sh_info != 0) {
        const double qn = __longlong_as_double(0x7ff8000000000000LL);
        for (size_t e = tid; e < (size_t)n * n + 3 * n; e += 512) Wt[e] = qn;
    }
}

// The same with a second, independent job beside it (round 4): workgroup 0 is k_bam_cholw<true>, workgroup 1 factors
// [Gamma11 | I] -> [R11 | W11] of the factor-form chain's Gram matrix under the rank-revealing rule (gsmvi_factor.hip,
// factor_chain_big: Gamma11 = Vw Vw^T is known before the B x B chain starts) -- two ~45 us one-CU pivot chains in the time of one.
__global__ __launch_bounds__(512) void k_bam_cholw_pair(int n, const double* __restrict__ BBg, double* Rg, double* Wt,
                                                        int* __restrict__ info, cholw_job g) {
    CHOL128W_LDS(E1, B12, scr, sh_fail, sh_moderate);
    __shared__ int sh_info;
    const int tid = threadIdx.x;
    if (blockIdx.x == 1) {
        chol128w_core<true, false>(E1, B12, scr, sh_fail, &sh_moderate, g.nb, g.A, g.lda, g.R, g.ldr, g.W, g.ldw, &sh_info);
        __threadfence_block();
        __syncthreads();
        // The same guard as bamq_side_body's: the orthogonal basis of the factor form leans on M1 + Gvv M1' = 0, good to
        // eps cond(Gvv).  Rows the rank-revealing rule DROPPED (exactly repeated draws: zero diagonal) are fine -- the update then
        // equals the dense one --, an ALMOST repeated draw (kept, tiny pivot) is not: cond(Gvv) >= (max / min R_ii)^2 > 1e8
        // counts as a failure (5e-6 off without a flag at cond 1e12, measured).
        __shared__ double gd[2][2];
        if (tid < 128) {
            const double d = tid < g.nb ? g.R[(size_t)tid * g.ldr + tid] : 0.0;
            double dmax = d, dmin = d > 0.0 ? d : 1.7976931348623157e308;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const double a = __shfl_xor(dmax, m, 64), b = __shfl_xor(dmin, m, 64);
                dmax = a > dmax ? a : dmax;
                dmin = b < dmin ? b : dmin;
            }
            if ((tid & 63) == 0) { gd[tid >> 6][0] = dmax; gd[tid >> 6][1] = dmin; }
        }
        __syncthreads();
        if (tid == 0) {
            const double dmax = gd[0][0] > gd[1][0] ? gd[0][0] : gd[1][0], dmin = gd[0][1] < gd[1][1] ? gd[0][1] : gd[1][1];
            int f = sh_info;
            if (g.cond_guard && f == 0 && !(dmin > 1e-4 * dmax)) f = g.nb;
            *g.info = f;
        }
        return;
    }
    chol128w_core<false, true>(E1, B12, scr, sh_fail, &sh_moderate, n, BBg, n, Rg, n, Wt, n, info, &sh_info);
    __syncthreads();
    if (sh_info != 0) {
        const double qn = __longlong_as_double(0x7ff8000000000000LL);
        for (size_t e = tid; e < (size_t)n * n + 3 * n; e += 512) Wt[e] = qn;
    }
}

// ---- n > 129: the small outputs from the blocked Cholesky factor of BB (gsmvi_potrf_impl: BB = R^T R) ------------------
// One workgroup: Ld = R^T (lower), Ldinv, vg = Vf gbar = M1[:, n-1] / r1s, zg = L^-1 (P gbar + M1^T vg) by a column-oriented
// forward substitution (row pp of R is contiguous; eight rows' loads in flight).  A failed factorisation (or a NaN in it)
// poisons every output, as k_bam_cholw does.
__global__ __launch_bounds__(1024) void k_bam_post_big(int n, bam_reg regs, const double* __restrict__ Rb,
                                                       const int* __restrict__ info_p, const double* __restrict__ M1,
                                                       const double* __restrict__ N0, double* __restrict__ Ld,
                                                       int* __restrict__ info, const double* __restrict__ Wblk) {
    const double reg = regs.get();
    __shared__ double sc[BAMS_NBIG + 8], av[BAMS_NBIG + 8];
    __shared__ int sh_bad;
    const int tid = threadIdx.x;
    double* Ldinv = Ld + (size_t)n * n;
    double* zg = Ldinv + n;
    double* vg = zg + n;
    if (tid == 0) sh_bad = (*info_p != 0) ? 1 : 0;
    const double r1s = sqrt(reg / (1.0 + reg));
    for (int p = tid; p < n; p += 1024) sc[p] = M1[(size_t)p * n + (n - 1)] / r1s;
    __syncthreads();
    for (int p = tid; p < n; p += 1024) {
        double a0 = N0[(size_t)p * n + (n - 1)] / r1s, a1 = 0.0;
        int kk = 0;
        for (; kk + 8 <= n; kk += 8) {
            double m[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) m[u] = M1[(size_t)(kk + u) * n + p];
#pragma unroll
            for (int u = 0; u < 8; u += 2) { a0 += m[u] * sc[kk + u]; a1 += m[u + 1] * sc[kk + u + 1]; }
        }
        for (; kk < n; ++kk) a0 += M1[(size_t)kk * n + p] * sc[kk];
        av[p] = a0 + a1;
        const double dg = Rb[(size_t)p * n + p];
        if (!(dg > 0.0) || !(dg < 1.7976931348623157e308)) sh_bad = 1;
    }
    __syncthreads();
    const int bad = sh_bad;
    if (tid == 0) *info = bad;
    if (bad) {
        const double qn = __longlong_as_double(0x7ff8000000000000LL);
        for (size_t e = tid; e < (size_t)n * n + 3 * n; e += 1024) Ld[e] = qn;
        return;
    }
    for (int e = tid; e < n * n; e += 1024) {
        const int i = e / n, j = e % n;                      // L[i][j] = R[j][i], j <= i
        Ld[e] = (j <= i) ? Rb[(size_t)j * n + i] : 0.0;
    }
    for (int p = tid; p < n; p += 1024) { Ldinv[p] = 1.0 / Rb[(size_t)p * n + p]; vg[p] = sc[p]; }
    if (Wblk) {
        // round 6: zg = L^-1 a BLOCKED on the inverses of the diagonal blocks that the factorisation left behind (k_potrf_dag's
        // W_r = R_rr^-T = L_rr^-1, 64 x 64 each): zg_r = W_rr (a_r - sum_{k < 64 r} L[., k] zg[k]), sixteen lanes per row, two barriers
        // per 64 rows instead of two per row (n = 256: 77 us of pivots-with-barriers -> 4 blocks)
        const int row = tid >> 4, sub = tid & 15;                // 64 rows x 16 lanes
        for (int r0 = 0; r0 < n; r0 += 64) {
            const int i = r0 + row;
            double acc_ = 0.0;
            if (i < n)
                for (int k = sub; k < r0; k += 16) acc_ += Rb[(size_t)k * n + i] * av[k];       // L[i][k] = R[k][i]; av[k] = zg[k], k < r0
            acc_ = row16_sum(acc_);
            __syncthreads();                                     // (all reads of av[0 .. r0) by this block's rows are done)
            if (sub == 0) sc[row] = (i < n) ? av[i] - acc_ : 0.0; // the block's right-hand side (sc: vg was copied out above)
            __syncthreads();
            double z = 0.0;
            const double* wrow = Wblk + (size_t)(r0 >> 6) * 4096 + row * 64;
#pragma unroll
            for (int q = 0; q < 4; ++q) z += wrow[sub + 16 * q] * sc[sub + 16 * q];
            z = row16_sum(z);
            __syncthreads();
            if (sub == 0 && i < n) av[i] = z;
            __syncthreads();
        }
        if (tid < n) zg[tid] = av[tid];
        return;
    }
    // zg = L^-1 a: thread p owns a[p] (n <= 1024 threads... n <= BAMS_NBIG); step pp needs R[pp][p], p > pp
    double mine = (tid < n) ? av[tid] : 0.0;
    const int pc = tid < n ? tid : n - 1;
    for (int p0 = 0; p0 < n; p0 += 8) {
        double r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = Rb[(size_t)(p0 + u < n ? p0 + u : n - 1) * n + pc];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int pp = p0 + u;
            if (pp < n) {                                    // block-uniform
                if (tid == pp) av[pp] = mine / r[u];         // r[u] = R[pp][pp] in thread pp
                __syncthreads();
                const double zk = av[pp];
                if (tid > pp && tid < n) mine -= r[u] * zk;
                __syncthreads();
            }
        }
    }
    if (tid < n) zg[tid] = av[tid];
}

// ---- n <= 48: the WHOLE small chain in one workgroup (round 3) ---------------------------------------------------------
// From the split-K slabs of the stacked Gram product [N0; M1] to everything k_bam_zw consumes: slab sum,
// N = M1^T M1 + sym(N0), the scaled Newton-Schulz iteration (same recurrence, same product order as the multi-workgroup steps), BB, its
// Cholesky factor (chol64_blk, gsmvi_chol64b.h) and the small outputs -- one launch instead of finish + nmat + iteration +
// Cholesky (4 launches, 77 + 10 us at n = 32).  fp64 MFMA throughput of ONE CU is the bound of the iteration (a 16x16x4 fp64
// MFMA occupies a SIMD for ~107 cycles): only the nb x nb blocks that exist are computed (the round-2 kernel always ran nine
// waves over a 3 x 3 block grid), dealt to the eight waves so that they spread over the four SIMDs.
// Slab element (r, c) of slab k: slabs[k * slab_stride + r * ldslab + c], rows 0..n-1 = N0, rows n..2n-1 = M1.
// Outputs: M1g (n x n finished M1), Wt = (L^-1)^T (n x n) in Ld's slot with [a | . | vg] behind it, *info.
#define BAMQ_SN 48
#define BAMQ_LD 50
// PAIR: the launch's second workgroup is the side workgroup of the orthogonal basis (bamq_side_body above); the chain
// adds t2 = W^T W a.
template <int NB, bool PAIR>
__global__ __launch_bounds__(512) void k_bam_small48(int n, bam_reg regs, const double* __restrict__ slabs, int kc, int ldslab,
                                                     long long slab_stride, double* __restrict__ M1g,
                                                     double* __restrict__ Ld, int* __restrict__ info,
                                                     unsigned long long* __restrict__ stamps, bamq_side sd) {
    const double reg = regs.get();
#define Q_STAMP(k)                                                                          \
    do {                                                                                    \
        if (stamps && threadIdx.x == 0) stamps[k] = __builtin_amdgcn_s_memrealtime();        \
    } while (0)
    Q_STAMP(0);
    if (stamps && (threadIdx.x & 63) == 0) stamps[8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(2308);   // HW_ID.SIMD_ID
    constexpr int MSZ = BAMQ_SN * BAMQ_LD;                   // 2400 doubles per matrix
    __shared__ __attribute__((aligned(16))) double sm[7 * MSZ];
    __shared__ __attribute__((aligned(16))) double scr[CHOLB_SCRATCH_DOUBLES(1)];
    __shared__ double coefs[BAMS_KMAX + 4], n0c[BAMQ_SN], sc[BAMQ_SN], av[BAMQ_SN], red[8];
    __shared__ int sh_fail, sh_bad;
    static_assert(BAMQ_SIDE_DOUBLES <= 7 * MSZ, "the side workgroup's matrices overlay sm");
    if (PAIR && blockIdx.x == 1) {                           // block-uniform (bamq_side_body above)
        bamq_side_body(n, bamq_src_slabs{slabs, kc, ldslab, n, slab_stride}, sm, scr, &sh_fail, sd);
        return;
    }
    double* Ms = sm + 4 * MSZ;
    double* M1s = sm + 5 * MSZ;
    double* Nm = sm + 6 * MSZ;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, cc = l & 15, ks = l >> 4;
    constexpr int nb = NB, nblocks = NB * NB;              // NB = ceil(n / 16): the block grid is a compile-time shape
    const int nk = (n + 3) >> 2;
    for (int e = tid; e < 7 * MSZ; e += 512) sm[e] = 0.0;
    if (tid == 0) sh_bad = 0;
    __syncthreads();
    // slab sums: N0 -> Ms (temporarily), M1 -> M1s and out
    for (int e0 = 0; e0 < 2 * n * n; e0 += 512 * 4) {
        double t[4][GSMVI_MAX_KC];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int e = e0 + 512 * u + tid;
            if (e >= 2 * n * n) e = 2 * n * n - 1;
            const int r = e / n, c = e - r * n;
#pragma unroll
            for (int k = 0; k < GSMVI_MAX_KC; ++k)
                t[u][k] = slabs[(size_t)(k < kc ? k : kc - 1) * slab_stride + (size_t)r * ldslab + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 512 * u + tid;
            if (e < 2 * n * n) {
                const int r = e / n, c = e - r * n;
                double a = 0.0;
#pragma unroll
                for (int k = 0; k < GSMVI_MAX_KC; ++k) a += (k < kc) ? t[u][k] : 0.0;
                if (r < n) Ms[r * BAMQ_LD + c] = a;
                else {
                    M1s[(r - n) * BAMQ_LD + c] = a;
                    M1g[(size_t)(r - n) * n + c] = a;
                }
            }
        }
    }
    __syncthreads();
    Q_STAMP(1);
    const double r1s = sqrt(reg / (1.0 + reg));
    if (tid < n) {
        n0c[tid] = Ms[tid * BAMQ_LD + n - 1];
        sc[tid] = M1s[tid * BAMQ_LD + n - 1] / r1s;          // vg = Vf gbar = M1[:, n-1] / r1s
    }
    // N = M1^T M1 + sym(N0)
    for (int blk = w; blk < nblocks; blk += 8) {
        const int i0 = 16 * (blk / nb), j0 = 16 * (blk % nb);
        v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        for (int st = 0; st < nk; st += 2) {
            const int k0 = 4 * st + ks, k1 = k0 + 4;         // (k1 may run into the zero padding: rows < 48 + 4 stay inside sm)
            acc0 = GSMVI_MFMA_F64(M1s[k0 * BAMQ_LD + i0 + cc], M1s[k0 * BAMQ_LD + j0 + cc], acc0);
            if (st + 1 < nk) acc1 = GSMVI_MFMA_F64(M1s[k1 * BAMQ_LD + i0 + cc], M1s[k1 * BAMQ_LD + j0 + cc], acc1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + ks + 4 * r, j = j0 + cc;
            if (i < n && j < n) Nm[i * BAMQ_LD + j] = (acc0[r] + acc1[r]) + 0.5 * (Ms[i * BAMQ_LD + j] + Ms[j * BAMQ_LD + i]);
        }
    }
    if (w == 7) {
        // meanwhile, on the last wave: s = trace(N + I/4) = ||M1||_F^2 + trace(N0) + n/4 straight from the operands, and the
        // scalar scaling recurrence (as k_bam_ns_step0) -- ~120 ns per step of one lane, stopped at k*
        double tr = 0.0;
        if (l < n) {
            double t0 = Ms[l * BAMQ_LD + l] + 0.25, t1 = 0.0;
            for (int k = 0; k + 1 < n; k += 2) {
                const double m0 = M1s[k * BAMQ_LD + l], m1 = M1s[(k + 1) * BAMQ_LD + l];
                t0 = __builtin_fma(m0, m0, t0);
                t1 = __builtin_fma(m1, m1, t1);
            }
            if (n & 1) { const double m0 = M1s[(n - 1) * BAMQ_LD + l]; t0 = __builtin_fma(m0, m0, t0); }
            tr = t0 + t1;
        }
        tr = wave_sum(tr);
        if (l == 0) {
            const double s0 = tr;
            red[0] = s0;
            double lb = 0.25 / s0;
            const bool s_ok = (s0 == s0) && s0 > 0.0 && s0 < 1e300;
            if (!(lb > 0.0) || lb > 1.0) lb = 1.0;
            int kst = BAMS_KMAX + 1;
            for (int k = 0; k < BAMS_KMAX; ++k) {
                const double c2 = (lb < 0.25) ? 3.0 / (1.0 + sqrt(lb) + lb) : 1.0;
                coefs[k] = c2;
                const double x = c2 * lb;
                lb = x * (3.0 - x) * (3.0 - x) * 0.25;
                if (lb > 1.0) lb = 1.0;
                if (1.0 - lb < 5e-9 && kst > BAMS_KMAX) kst = k + 2;
                if (k + 1 >= kst) break;                     // coefs[0 .. k*-1] are all the iteration reads
            }
            coefs[BAMS_KMAX + 1] = (!s_ok || kst > BAMS_KMAX) ? 1.0 : 0.0;
            if (kst > BAMS_KMAX) kst = BAMS_KMAX;
            coefs[BAMS_KMAX] = (double)kst;
        }
    }
    __syncthreads();
    const double s = red[0];
    const double sinv = 1.0 / s;
    for (int e = tid; e < BAMQ_SN * BAMQ_SN; e += 512) {
        const int i = e / BAMQ_SN, j = e - i * BAMQ_SN;
        const bool in = i < n && j < n;
        sm[i * BAMQ_LD + j] = in ? (Nm[i * BAMQ_LD + j] + (i == j ? 0.25 : 0.0)) * sinv : 0.0;
        sm[2 * MSZ + i * BAMQ_LD + j] = (in && i == j) ? 1.0 : 0.0;
        Ms[i * BAMQ_LD + j] = 0.0;
    }
    if (tid < n) {                                           // a = P gbar + M1^T vg (bam.py:107 applied to gbar)
        double a0 = n0c[tid] / r1s, a1 = 0.0;
        int kk = 0;
        for (; kk + 1 < n; kk += 2) {
            a0 += M1s[kk * BAMQ_LD + tid] * sc[kk];
            a1 += M1s[(kk + 1) * BAMQ_LD + tid] * sc[kk + 1];
        }
        if (kk < n) a0 += M1s[kk * BAMQ_LD + tid] * sc[kk];
        av[tid] = a0 + a1;
    }
    __syncthreads();
    Q_STAMP(2);
    const int kstar = (int)coefs[BAMS_KMAX];
    const bool failed = coefs[BAMS_KMAX + 1] != 0.0;
    constexpr int NST = 4 * NB;                              // k-steps of a product (the last up to three may be zero padding)
    // One step = two barrier intervals.  A wave owns block w (and wave 0 block 8 when nb = 3).  The Y- and Z-operands of the
    // second interval do not depend on M, so they are fetched during the first one, behind its MFMAs.
    for (int k = 0; k < kstar && !failed; ++k) {
        const double* Y = sm + (k & 1) * MSZ;
        const double* Z = sm + (2 + (k & 1)) * MSZ;
        double* Yo = sm + ((k & 1) ^ 1) * MSZ;
        double* Zo = sm + (2 + ((k & 1) ^ 1)) * MSZ;
        const double c2 = coefs[k], c = sqrt(c2);
        constexpr int NSLOT = (NB * NB > 8) ? 2 : 1;
        double ya[NSLOT][NST], zb[NSLOT][NST];
#pragma unroll
        for (int slot = 0; slot < NSLOT; ++slot) {           // M = Z Y
            const int blk = w + 8 * slot;
            if (blk < nblocks) {                             // wave-uniform
                const int i0 = 16 * (blk / nb), j0 = 16 * (blk % nb);
                double a[NST], b[NST];
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int kk = 4 * st + ks;
                    a[st] = Z[(i0 + cc) * BAMQ_LD + kk];
                    b[st] = Y[kk * BAMQ_LD + j0 + cc];
                }
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int kk = 4 * st + ks;
                    ya[slot][st] = Y[(i0 + cc) * BAMQ_LD + kk];
                    zb[slot][st] = Z[kk * BAMQ_LD + j0 + cc];
                }
                v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int st = 0; st < NST; st += 2) {
                    acc0 = GSMVI_MFMA_F64(a[st], b[st], acc0);
                    acc1 = GSMVI_MFMA_F64(a[st + 1], b[st + 1], acc1);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) Ms[(i0 + ks + 4 * r) * BAMQ_LD + j0 + cc] = acc0[r] + acc1[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int slot = 0; slot < NSLOT; ++slot) {           // Y' = c Y T,  Z' = c T Z,  T = 1.5 I - 0.5 c2 M
            const int blk = w + 8 * slot;
            if (blk < nblocks) {
                const int i0 = 16 * (blk / nb), j0 = 16 * (blk % nb);
                double tb[NST], ta[NST];
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int kk = 4 * st + ks;
                    tb[st] = (kk == j0 + cc ? 1.5 : 0.0) - 0.5 * c2 * Ms[kk * BAMQ_LD + j0 + cc];
                    ta[st] = (kk == i0 + cc ? 1.5 : 0.0) - 0.5 * c2 * Ms[(i0 + cc) * BAMQ_LD + kk];
                }
                v4d ay0 = {0.0, 0.0, 0.0, 0.0}, ay1 = ay0, az0 = ay0, az1 = ay0;
#pragma unroll
                for (int st = 0; st < NST; st += 2) {
                    ay0 = GSMVI_MFMA_F64(ya[slot][st], tb[st], ay0);
                    az0 = GSMVI_MFMA_F64(ta[st], zb[slot][st], az0);
                    ay1 = GSMVI_MFMA_F64(ya[slot][st + 1], tb[st + 1], ay1);
                    az1 = GSMVI_MFMA_F64(ta[st + 1], zb[slot][st + 1], az1);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Yo[(i0 + ks + 4 * r) * BAMQ_LD + j0 + cc] = c * (ay0[r] + ay1[r]);
                    Zo[(i0 + ks + 4 * r) * BAMQ_LD + j0 + cc] = c * (az0[r] + az1[r]);
                }
            }
        }
        __syncthreads();
    }
    Q_STAMP(3);
    // BB = N + I/2 + sqrt(s) sym(Y) -> EW ([BB | I], upper triangle, identity beyond n), then [R | W], W = R^-T = L^-1.
    // Round 5: the factor WITH its inverse (chol64_blk's augmented columns, as k_bam_cholw), so that Z = W (P + M1^T Vf) is the
    // two chained MFMA products of k_bam_zw for n <= 48 too (~8 us) instead of a 16-lanes-per-column substitution (k_bam_forward16,
    // 19 - 20 us at n = 32).  EW (64 x 146) lies over the iteration's buffers from sm + 0: Y and N are read into registers, then
    // a barrier, then EW is written.
    constexpr int EWS = 146;
    static_assert(64 * EWS <= 7 * MSZ, "EW overlays the iteration's matrices");
    double* EW = sm;
    {
        const double* Yf = sm + (kstar & 1) * MSZ;
        const double rs = sqrt(s);
        int nan_in = failed ? 1 : 0;
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 512 * u, i = e >> 6, j = e & 63;
            const bool in = i < n && j < n;
            double x = (i == j) ? 1.0 : 0.0;
            if (in) {
                x = Nm[i * BAMQ_LD + j] + (i == j ? 0.5 : 0.0) + rs * 0.5 * (Yf[i * BAMQ_LD + j] + Yf[j * BAMQ_LD + i]);
                if (!(x == x)) nan_in = 1;
                if (j < i) x = 0.0;
            }
            v[u] = x;
        }
        __syncthreads();                                     // every read of Y and N is done
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 512 * u, i = e >> 6, j = e & 63;
            EW[i * EWS + j] = v[u];
        }
        if (nan_in) sh_bad = 1;
    }
    __syncthreads();
    Q_STAMP(4);
    chol64_blk<EWS, false, 1>(EW, scr, n, &sh_fail);
    Q_STAMP(5);
    // outputs in k_bam_cholw / k_bam_bbav's layout: Wt (n x n, Wt[i][j] = W[j][i]) in Ld's slot, [a | . | vg] behind it
    double* avo = Ld + (size_t)n * n;
    double* vg = avo + 2 * n;
    const int bad = sh_bad || sh_fail != 0;
    if (tid == 0) *info = bad;
    if (bad) {                                               // poison: nothing stale may be applied
        const double qn = __longlong_as_double(0x7ff8000000000000LL);
        for (size_t e = tid; e < (size_t)n * n + 3 * n; e += 512) Ld[e] = qn;
        if (PAIR && tid < n) sd.t2[tid] = qn;
        return;
    }
    for (int e = tid; e < n * n; e += 512) {
        const int i = e / n, j = e - i * n;
        Ld[e] = (i <= j) ? EW[j * EWS + 64 + i] : 0.0;
    }
    if (tid < n) {
        avo[tid] = av[tid];
        vg[tid] = sc[tid];
    }
    if (PAIR) {                                              // t2 = W^T (W a) for the vg correction (k_bam_zw's prologue): two
        double* zgs = n0c;                                   // short dot products per thread straight from EW (n0c is dead)
        if (tid < n) {
            double z0 = 0.0, z1 = 0.0;
            int k = 0;
            for (; k + 1 <= tid; k += 2) {
                z0 += EW[tid * EWS + 64 + k] * av[k];
                z1 += EW[tid * EWS + 64 + k + 1] * av[k + 1];
            }
            if (k <= tid) z0 += EW[tid * EWS + 64 + k] * av[k];
            zgs[tid] = z0 + z1;
        }
        __syncthreads();
        if (tid < n) {
            double t0 = 0.0, t1 = 0.0;
            int r = tid;
            for (; r + 1 < n; r += 2) {
                t0 += EW[r * EWS + 64 + tid] * zgs[r];
                t1 += EW[(r + 1) * EWS + 64 + tid] * zgs[r + 1];
            }
            if (r < n) t0 += EW[r * EWS + 64 + tid] * zgs[r];
            sd.t2[tid] = t0 + t1;
        }
    }
    if (stamps && tid == 0) { stamps[6] = __builtin_amdgcn_s_memrealtime(); stamps[7] = (unsigned long long)kstar; }
#undef Q_STAMP
}

int gsmvi_bam_small_fused_nmax() { return BAMQ_SN; }

// n <= 48: slabs of [N0; M1] in, everything out (see k_bam_small48)
int gsmvi_bam_small_fused(gsmvi_ctx* ctx, hipStream_t st, int n, bam_reg reg, const double* slabs, int kc, int ldslab,
                          size_t slab_stride, double* M1, double* Ld, int* info_dev, const bamq_side* side) {
    unsigned long long* stamps = (ctx->tune_cov_dbg & 256)           // diagnostic (scripts/bam48_timeline.py): phase stamps
                                     ? reinterpret_cast<unsigned long long*>(ctx->gram_slabs + (size_t)GSMVI_MAX_KC * ctx->rmax * ctx->rmax)
                                     : nullptr;
#define SMALL48(NBV, PR) hipLaunchKernelGGL((k_bam_small48<NBV, PR>), dim3(PR ? 2 : 1), dim3(512), 0, st, n, reg, slabs, kc, ldslab, (long long)slab_stride, M1, Ld, info_dev, stamps, PR ? *side : bamq_side{})
    if (side) { if (n <= 16) SMALL48(1, true); else if (n <= 32) SMALL48(2, true); else SMALL48(3, true); }
    else { if (n <= 16) SMALL48(1, false); else if (n <= 32) SMALL48(2, false); else SMALL48(3, false); }
#undef SMALL48
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gsmvi_set_error("BaM small-matrix launch failed: %s%s", hipGetErrorString(e), "");
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}

int gsmvi_potrf_impl(struct gsmvi_ctx* ctx, hipStream_t st, int D, const double* S, int lds, double* R, int ldr,
                     int* info_dev);

// n <= 128: k_bam_bbav forms BB and [a | . | vg] behind W's slot, k_bam_cholw leaves Wt = (L^-1)^T (n x n, upper) in Ld's slot and
// the upper factor in Rscr (scratch); k_bam_zw (gsmvi_bam.hip) consumes Wt.  n > 128: blocked multi-workgroup Cholesky + the
// small outputs of k_bam_post_big (Ld = L, Ldinv, zg, vg) for the generic forward-substitution kernel.
int gsmvi_bam_small_device(gsmvi_ctx* ctx, hipStream_t st, int n, bam_reg reg, const double* Nd, const double* M1,
                           const double* N0, double* scratch, double* Ld, int* info_dev, int* hint_host, int force_kenq,
                           double* Rscr, const cholw_job* beside, const bamq_side* side64, const double* G11) {
    const int ld = n <= BAMS_NMAX ? BAMS_LD : ((n + 15) / 16) * 16;
    const size_t LL = (size_t)ld * ld;
    double* Ya = scratch;
    double* Za = Ya + LL;
    double* Yb = Za + LL;
    double* Zb = Yb + LL;
    double* Mm = Zb + LL;
    double* coef = Mm + LL;
    double* BBg = coef + 64;
    // steps to enqueue: the previous call's k* + 1 when known (pinned host word, read without synchronising), everything
    // otherwise; k_bam_ns_tail makes up for a guess that turns out too small
    int kenq = BAMS_KMAX;
    if (hint_host) {                                         // k* + 1 (round 4; + 2 before): a step beyond k* costs two launches,
        const int h = *reinterpret_cast<volatile int*>(hint_host);   // and k* moves by at most one between neighbouring calls.
        // (+ 0 measured in round 5, knob "bam_hint_slack": the same update back-to-back gains 2 us -- 442 vs 444 at c4 -- but
        // a FIT loses 20 us per iteration, 460 vs 440: k* moves with the regulariser and the one-workgroup tail step is slow.)
        const int slack = ctx->tune_bam_hint_slack;
        if (h > 0 && h + slack < BAMS_KMAX) kenq = h + slack;
        if (kenq < 1) kenq = 1;
    }
    if (force_kenq > 0 && force_kenq < BAMS_KMAX) kenq = force_kenq;       // tests: exercise the safety net
    const int nb = (n + 15) / 16;
    const bool one_wg = n <= 64 && !ctx->tune_bam_full && force_kenq <= 0;   // the whole iteration in one workgroup (k_bam_ns64)
    if (one_wg) {
        if (side64)                                          // (the caller asked gsmvi_bam_small_one_wg first)
            hipLaunchKernelGGL(k_bam_ns64<true>, dim3(2), dim3(512), 0, st, n, ld, Nd, Ya, Yb, coef, bamq_src_mats{G11, M1, n}, *side64);
        else
            hipLaunchKernelGGL(k_bam_ns64<false>, dim3(1), dim3(512), 0, st, n, ld, Nd, Ya, Yb, coef, bamq_src_mats{}, bamq_side{});
        kenq = 0;
    } else
    // step 0 forms s, Y0, Z0 = I and M0 = Y0 in its own operand loads (k_bam_ns_step0): no preparation launch
    hipLaunchKernelGGL(k_bam_ns_step0, dim3(2 * nb * nb), dim3(256), 0, st, n, ld, Nd, Yb, Zb, coef, hint_host);
    for (int k = 1; k < kenq; ++k) {
        hipLaunchKernelGGL(k_bam_ns_zy, dim3(nb * nb), dim3(256), 0, st, n, ld, k, Ya, Za, Yb, Zb, Mm, coef);
        hipLaunchKernelGGL(k_bam_ns_step, dim3(2 * nb * nb), dim3(256), 0, st, n, ld, k, Ya, Za, Yb, Zb, Mm, coef);
    }
    if (!one_wg && kenq < BAMS_KMAX)
        hipLaunchKernelGGL(k_bam_ns_tail, dim3(1), dim3(1024), 0, st, n, ld, kenq, Ya, Za, Yb, Zb, Mm, coef);
    if (n <= BAMS_NMAX) {
        // BB and the factor-independent vectors [a | . | vg] behind W's slot, then the factorisation with the inverse factor
        hipLaunchKernelGGL(k_bam_bbav, dim3(nb * nb + nb), dim3(256), 0, st, n, ld, reg, Nd, Ya, Yb, coef, M1, N0, BBg,
                           Ld + (size_t)n * n);
        if (n > 64 && beside)                               // a second one-workgroup factorisation shares the launch
            hipLaunchKernelGGL(k_bam_cholw_pair, dim3(2), dim3(512), 0, st, n, BBg, Rscr, Ld, info_dev, *beside);
        else if (n > 64) hipLaunchKernelGGL(k_bam_cholw<true>, dim3(1), dim3(512), 0, st, n, BBg, Rscr, Ld, info_dev);
        else hipLaunchKernelGGL(k_bam_cholw<false>, dim3(1), dim3(512), 0, st, n, BBg, Rscr, Ld, info_dev);
    } else {
        // beyond the one-workgroup Cholesky: the blocked multi-workgroup factorisation of the D x D path (its workspace is
        // the idle panel-partial slab), then the small outputs in one workgroup for the generic forward-substitution kernel
        hipLaunchKernelGGL(k_bam_ns_bb, dim3((n * n + 255) / 256), dim3(256), 0, st, n, ld, Nd, Ya, Yb, coef, BBg);
        double* Rb = Mm;                                     // the iterates are dead once BB exists
        int* info_p = ctx->ints + 9;
        int rc = gsmvi_potrf_impl(ctx, st, n, BBg, n, Rb, n, info_p);
        if (rc != GSMVI_OK) return rc;
        hipLaunchKernelGGL(k_bam_post_big, dim3(1), dim3(1024), 0, st, n, reg, Rb, info_p, M1, N0, Ld, info_dev,
                           (ctx->potrf_w && ctx->potrf_w_n == n && !ctx->tune_no_fast) ? ctx->potrf_w : (const double*)nullptr);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gsmvi_set_error("BaM small-matrix launch failed: %s%s", hipGetErrorString(e), "");
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}

// whether gsmvi_bam_small_device runs the iteration as ONE workgroup (k_bam_ns64), i.e. can carry the side workgroup
int gsmvi_bam_small_one_wg(const gsmvi_ctx* ctx, int n) { return n <= 64 && !ctx->tune_bam_full && ctx->tune_bam_kenq <= 0; }
int gsmvi_bam_small_nmax() { return BAMS_NBIG; }
size_t gsmvi_bam_small_scratch_doubles(int n) {
    const size_t ld = n <= BAMS_NMAX ? BAMS_LD : ((n + 15) / 16) * 16;
    return (size_t)5 * ld * ld + 64 + (size_t)n * n;
}
