// Guard-free fast paths of the dense-covariance GSM update for gfx950.
//
// Selected by the ABI when D % 64 == 0, every leading dimension is even, every base pointer is
// 16-byte aligned and B is one of {16,32,64} for the covariance kernel (BASELINE configs c3, c5).  Everything else
// runs the guarded generic kernels of gsmvi_kernels.hip (same arithmetic, same reduction order
// inside a kernel family is NOT promised across the two families).
//
// These kernels are latency-bound at D=1024 (8 MB of covariance over 256 CUs = 32 KB per CU), so
// the design rule is: issue every global load of a workgroup in ONE batch, wait once, then MFMA,
// then store.  No data-dependent branch sits between a load and its use.
#include "gsmvi_common.h"
#include "gsmvi_ctx.h"
#include "gsmvi_small16.h"
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
// Workgroup (512 threads) = 16 columns of M x 256-row chunks; wave w of 8 owns rows 32w..32w+31 of
// the chunk (two waves per SIMD: the fp64 MFMA pipe needs two to reach its ~46 TF) and MFMA k-slot
// ks of step s is row 32w + 4s + ks.  M (the D x D covariance / precision / factor) is
// streamed once from HBM as 128-B row segments straight into registers.  The left operand chunk
// A[:, 256 rows] (16*MT x 256 doubles) is loaded by the whole workgroup with fully coalesced 16-B
// accesses and staged in LDS ([row][258]: conflict-free ds_read_b64 for the MFMA A operand), because
// fragment-shaped loads of it touch 64 cache lines per instruction.
// Any even D (round 5; D % 64 == 0 until round 4): the one wave whose rows straddle row D re-reads row D - 1 for the rows
// beyond it (the A chunk is staged with zeros there).  M has ncols columns (any; ncols = D for the square matrices);
// slabs are nrows x ncols.
// =====================================================================================
// EXTRA = true adds the two optional pieces of gsmvi_panel_extras (gsmvi_ctx.h): right-operand rows taken from split-K slabs
// of a previous product (summed while they are loaded, so that product needs no finish launch), and a slab-summing side
// job shared by all workgroups (finishes the small Gram matrix of the factor path beside this launch's own work).
// RIDER = true (factor path, round 3): the launch carries ONE more workgroup -- x index gridDim.x - 1, (y, z) = (0, 0) -- that
// runs the 2B x 2B chain of the factor update (gsmf_small16_body) while the other workgroups form V Fm: the chain (31 us on one
// CU) does not depend on this product, only the update kernel behind both does, so the product and one launch boundary
// disappear from the iteration's critical path.  The chain's 142 KB of LDS become the launch's LDS size (one workgroup per
// CU for the product: it finishes long before the chain either way).
// RAG (round 5) = the clamped form for D % 64 != 0 or ncols % 16 != 0; RAG = false is the round-4 code, instruction for
// instruction (the clamps and the straddling-wave branch cost 6 - 11 % on the grid shapes when they were unconditional:
// scripts/cov_ab_rounds.py).
template <int MT, bool HAS_SHIFT, int CHW, bool EXTRA, bool RIDER = false, bool RAG = false>
__global__ __launch_bounds__(512) void k_panel_fast(int D, int nrows, const double* __restrict__ A, int lda,
                                                    const double* __restrict__ shift, double alpha,
                                                    const double* __restrict__ M, int ldm,
                                                    double* Pp, int chunks_per_wg, int ncols,
                                                    unsigned long long* __restrict__ stamps, double* __restrict__ Out,
                                                    int ldo, const double* __restrict__ addvec, gsmvi_panel_extras px) {
    // timeline diagnostic: a kernel's slot holds GSMVI_STAMP_WG workgroups x 8 words; workgroups beyond it write nothing
#define PSTAMP(k)                                                                                              \
    do {                                                                                                       \
        const unsigned wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                   \
        if (stamps && threadIdx.x == 0 && wg_ < GSMVI_STAMP_WG)                                                \
            stamps[(size_t)wg_ * 8 + (k)] = __builtin_amdgcn_s_memrealtime();                                  \
    } while (0)
    PSTAMP(0);
    constexpr int LDG = CHW + 2;                   // LDS row stride of the staged A chunk (doubles)
    constexpr int NR = 16 * MT;
    constexpr int RW = CHW / 8;                    // rows of the chunk per wave (8 waves)
    constexpr int NST = RW / 4;                    // MFMA steps per wave and chunk
    constexpr int U16 = CHW / 2;                   // 16-B units per staged row
    constexpr int UPT = NR * U16 / 512;            // staging units per thread
    constexpr int SMEM_P = (NR * LDG > 8 * NR * 17) ? NR * LDG : 8 * NR * 17;
    constexpr int SMEM = (RIDER && GSMF_SMALL16_LDS > SMEM_P) ? GSMF_SMALL16_LDS : SMEM_P;
    __shared__ __attribute__((aligned(16))) double As[SMEM];       // staging buffer, reused for the reduction
    if (RIDER && blockIdx.x == gridDim.x - 1) {                     // block-uniform
        if (blockIdx.y == 0 && blockIdx.z == 0)
            gsmf_small16_body(As, px.rd_n, px.rd_B, px.rd_Gp, px.rd_kcg, px.rd_Kmat, px.rd_coef, px.rd_bad, px.rd_stamps,
                              px.rd_jmode, px.rd_prior, px.rd_Pi, px.rd_R11, px.rd_W11);
        return;
    }
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    // (round 5: any even D and any ncols -- a column beyond the matrix is a clamped re-read whose result is not stored, a
    // row of M beyond D is a clamped re-read multiplied by the zeros the A chunk is staged with beyond D)
    const int jx = blockIdx.x * 16 + c;
    const int j = (RAG && jx >= ncols) ? ncols - 1 : jx;
    const int r0 = blockIdx.z * NR;

    v4d acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4d){0.0, 0.0, 0.0, 0.0};

    for (int ch = 0; ch < chunks_per_wg; ++ch) {
        const int cbase = (blockIdx.y * chunks_per_wg + ch) * CHW;          // block-uniform
        if (cbase >= D) break;
        const int wbase = cbase + w * RW;                                   // wave-uniform: 8 waves x RW rows
        const bool wave_in = wbase < D;
        // ---- every global load of this chunk in one batch: the left-operand chunk FIRST (it comes from L2 / the
        // Infinity Cache and has to pass through LDS and a barrier), the HBM stream of M behind it.  vmcnt counts in
        // issue order, so the staging below waits only for the A loads and MFMA step s only for m[0..s]: the LDS
        // staging and the first MFMA steps overlap the tail of the M stream (measured with the in-kernel timeline,
        // scripts/timeline2.py: the load phase and the MFMA phase used to be strictly serial).
        v2d ga[UPT];
        v2d gs[UPT];
        unsigned okbits = 0;
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = q * 512 + tid;
            const int row = u / U16, c16 = u % U16;
            const int grow = r0 + row;
            const int col = cbase + 2 * c16;
            const bool ok = grow < nrows && col < D;
            ga[q] = *reinterpret_cast<const v2d*>(A + (size_t)(grow < nrows ? grow : nrows - 1) * lda +
                                                  (col < D ? col : 0));
            if (HAS_SHIFT) gs[q] = *reinterpret_cast<const v2d*>(shift + (col < D ? col : 0));
            okbits |= (ok ? 1u : 0u) << q;     // the out-of-range select is applied at staging time: no use of a
        }                                      // loaded value may sit in front of the M stream's issue
        __builtin_amdgcn_sched_barrier(0);
        double m[NST];
        if (EXTRA && px.msl != nullptr && (wave_in ? wbase : 0) + ks >= px.msplit) {
            // wave-uniform when msplit is a multiple of the wave's row count (it is for B % 16 == 0); rows come in steps of 4,
            // so ks >= ... holds for all s of this lane once it holds for s = 0
            // (D % 16 == 0 on this path: the callers' K'' Tm product has D = 2B with B % 16 == 0)
            const double* sp = px.msl + (size_t)((wave_in ? wbase : 0) + ks - px.msplit) * px.ldsl + j;
            // (two compile-time bounds on the slab count, chosen block-uniformly: with the clamp to GSMVI_MAX_KC alone every
            // entry cost 8 loads whatever the count)
            if (px.kcm <= 4) {
                double t[NST][4];
#pragma unroll
                for (int s = 0; s < NST; ++s)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        t[s][q] = sp[(size_t)(q < px.kcm ? q : px.kcm - 1) * px.mstride + (size_t)(4 * s) * px.ldsl];
#pragma unroll
                for (int s = 0; s < NST; ++s) {
                    double a = 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) a += (q < px.kcm) ? t[s][q] : 0.0;
                    m[s] = a;
                }
            } else {
                double t[NST][GSMVI_MAX_KC];
#pragma unroll
                for (int s = 0; s < NST; ++s)
#pragma unroll
                    for (int q = 0; q < GSMVI_MAX_KC; ++q)
                        t[s][q] = sp[(size_t)(q < px.kcm ? q : px.kcm - 1) * px.mstride + (size_t)(4 * s) * px.ldsl];
#pragma unroll
                for (int s = 0; s < NST; ++s) {
                    double a = 0.0;
#pragma unroll
                    for (int q = 0; q < GSMVI_MAX_KC; ++q) a += (q < px.kcm) ? t[s][q] : 0.0;
                    m[s] = a;
                }
            }
            if (px.mfin != nullptr && blockIdx.z == 0 && wave_in) {
                double* fp = px.mfin + (size_t)(wbase + ks - px.msplit) * px.ldfin + j;
#pragma unroll
                for (int s = 0; s < NST; ++s) fp[(size_t)(4 * s) * px.ldfin] = m[s];
            }
        } else if (!RAG || wbase + RW <= D) {      // wave-uniform: all RW rows of this wave exist (!RAG: D % 64 == 0, so a wave
            // beyond row D -- its values are unused -- re-reads rows 0 .. RW - 1, which exist)
            const double* mp = M + (size_t)((wave_in ? wbase : 0) + ks) * ldm + j;
#pragma unroll
            for (int s = 0; s < NST; ++s) m[s] = mp[(size_t)(4 * s) * ldm];
        } else {                                   // the wave that straddles row D (D % RW != 0) and, for D < RW, the waves beyond it:
            // clamped rows.  (Round 6: the waves beyond row D used to take the unclamped branch with base row 0 -- for D < 32 that
            // read rows D .. 31 of a D-row matrix, past the END of the caller's array; harmless inside an allocator block, a
            // memory access fault when the array ends its mapping: found by the whole suite in one process)
            const int rb = wave_in ? wbase : 0;
#pragma unroll
            for (int s = 0; s < NST; ++s) {
                const int r = rb + ks + 4 * s;
                m[s] = M[(size_t)(r < D ? r : D - 1) * ldm + j];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ch > 0) __syncthreads();       // previous chunk's MFMA reads of As are done
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = q * 512 + tid;
            const int row = u / U16, c16 = u % U16;
            v2d v = ga[q];
            if (!((okbits >> q) & 1u)) v = HAS_SHIFT ? gs[q] : (v2d){0.0, 0.0};   // contributes alpha*(x-x) = 0
            if (HAS_SHIFT) { v.x -= gs[q].x; v.y -= gs[q].y; }
            v.x *= alpha; v.y *= alpha;
            *reinterpret_cast<v2d*>(&As[row * LDG + 2 * c16]) = v;
        }
        __syncthreads();
        PSTAMP(1);                         // left operand staged (the M stream may still be in flight)
        if (wave_in) {
            const double* ap = As + c * LDG + RW * w + ks;
            double av[MT][NST];                     // operands to registers first: no LDS round trip per MFMA step
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int s = 0; s < NST; ++s) av[mt][s] = ap[mt * 16 * LDG + 4 * s];
#pragma unroll
            for (int s = 0; s < NST; ++s) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = GSMVI_MFMA_F64(av[mt][s], m[s], acc[mt]);
            }
        }
    }

    // cross-wave reduction through LDS (fixed order => deterministic), red[w][row][17], w = 0..7
    __syncthreads();
    PSTAMP(2);
    double* red = As;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(w * NR + 16 * mt + ks + 4 * r) * 17 + c] = acc[mt][r];
    __syncthreads();
    // Out == nullptr: the slab Pp[kc] is the result (a finish pass or a consumer sums the KC slabs).
    // Out != nullptr (only launched with KC == 1): FINISHED output Out = addvec + slab, no finish pass.
    for (int idx = tid; idx < NR * 16; idx += 512) {
        const int rr = idx >> 4, cc = idx & 15;
        const int row = r0 + rr;
        if (row < nrows && (!RAG || (int)blockIdx.x * 16 + cc < ncols)) {
            double s = 0.0;
#pragma unroll
            for (int ww = 0; ww < 8; ww += 2) s += red[(ww * NR + rr) * 17 + cc] + red[((ww + 1) * NR + rr) * 17 + cc];
            if (Out == nullptr) Pp[((size_t)blockIdx.y * nrows + row) * ncols + blockIdx.x * 16 + cc] = s;
            else Out[(size_t)row * ldo + blockIdx.x * 16 + cc] = s + (addvec ? addvec[blockIdx.x * 16 + cc] : 0.0);
        }
    }
    if (EXTRA && px.sj_src != nullptr) {            // side job: this workgroup's slice of a slab sum
        const int nwg = gridDim.x * gridDim.y * gridDim.z;
        const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const int per = (px.sj_len + nwg - 1) / nwg;
        // (round 5: a LOOP over the slice.  With one element per thread a launch of fewer than sj_len / 512 workgroups left the
        // tail of every slice unsummed -- D < 256 with 2B = 128: 12 - 24 workgroups for 16384 entries -- and the factor update
        // reported a failure it did not have, on every call: GSM.fit(method="auto") at D = 192, B = 64 reverted every iteration)
        const int i_end = (wg + 1) * per < px.sj_len ? (wg + 1) * per : px.sj_len;
        for (int i = wg * per + tid; i < i_end; i += (int)blockDim.x) {
            double t[GSMVI_MAX_KC];
#pragma unroll
            for (int q = 0; q < GSMVI_MAX_KC; ++q) t[q] = px.sj_src[(size_t)(q < px.sj_kc ? q : px.sj_kc - 1) * px.sj_stride + i];
            double a = 0.0;
#pragma unroll
            for (int q = 0; q < GSMVI_MAX_KC; ++q) a += (q < px.sj_kc) ? t[q] : 0.0;
            px.sj_dst[i] = a;
        }
    }
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PSTAMP(3); }
#undef PSTAMP
}

// =====================================================================================
// k_panel_fast with the NEXT chunk's loads in flight (round 6; D >= 2048, rows <= 32, on the grid, plain products).
// At D = 4096 a workgroup of k_panel_fast walks eight 256-row chunks, and each chunk is a serial chain: issue the loads (64 KB of
// left operand from L2, 32 KB of M from HBM) -> wait -> LDS staging -> barrier -> operand reads -> MFMAs -> next chunk's loads.
// Two resident workgroups per CU overlap each other and nothing else: 5.9 us per chunk, 47 - 50 us for the 134 MB of M at
// (4096, 32) = 2.8 TB/s (round-5 verdict, weak 6).  [A first attempt blamed the left operand's re-reads and gave every
// workgroup four 16-column tiles per staged chunk (a quarter of the re-reads): 47.1 us against 47.1 -- the chain, not the bytes.]
// Here the loads of chunk ch + 1 are issued as soon as chunk ch has gone to LDS -- two register sets, alternating -- so they fly
// during chunk ch's operand reads and MFMAs and the barriers around them; the barriers wait for LDS only (s_waitcnt lgkmcnt(0);
// s_barrier: __syncthreads() would drain the prefetch, as in k_gsm_cov_sym_p).  Same arithmetic and the same summation order as
// k_panel_fast with the same chunks_per_wg: bit-identical slabs.
// =====================================================================================
template <int MT, bool HAS_SHIFT>
__global__ __launch_bounds__(512) void k_panel_fast_p(int D, int nrows, const double* __restrict__ A, int lda,
                                                      const double* __restrict__ shift, double alpha,
                                                      const double* __restrict__ M, int ldm, double* __restrict__ Pp,
                                                      int chunks_per_wg, int ncols, double* __restrict__ Out, int ldo,
                                                      const double* __restrict__ addvec) {
#define LDS_BARRIER_P() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    constexpr int CHW = 256;
    constexpr int LDG = CHW + 2;
    constexpr int NR = 16 * MT;
    constexpr int RW = CHW / 8;
    constexpr int NST = RW / 4;
    constexpr int U16 = CHW / 2;
    constexpr int UPT = NR * U16 / 512;
    constexpr int SMEM = (NR * LDG > 8 * NR * 17) ? NR * LDG : 8 * NR * 17;
    __shared__ __attribute__((aligned(16))) double As[SMEM];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int j = blockIdx.x * 16 + c;
    const int r0 = blockIdx.z * NR;
    const int ch0 = blockIdx.y * chunks_per_wg;
    int nch = (D - ch0 * CHW + CHW - 1) / CHW;                  // chunks this workgroup owns (D % 256 == 0 on this path)
    if (nch > chunks_per_wg) nch = chunks_per_wg;
    v4d acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4d){0.0, 0.0, 0.0, 0.0};
    // one set of left-operand registers (they are consumed by the staging, in front of the next issue), two sets for M (chunk ch's
    // rows feed the MFMAs behind the issue of chunk ch + 1's): 2 workgroups per CU stay resident (<= 128 VGPRs)
    v2d ga[UPT], gs[UPT];
    auto issue = [&](int ch, double (&m)[NST]) {
        const int cbase = (ch0 + ch) * CHW;
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = q * 512 + tid;
            const int row = u / U16, c16 = u % U16;
            const int grow = r0 + row;
            ga[q] = *reinterpret_cast<const v2d*>(A + (size_t)(grow < nrows ? grow : nrows - 1) * lda + cbase + 2 * c16);
            if (HAS_SHIFT) gs[q] = *reinterpret_cast<const v2d*>(shift + cbase + 2 * c16);
        }
        __builtin_amdgcn_sched_barrier(0);
        const double* mp = M + (size_t)(cbase + w * RW + ks) * ldm + j;
#pragma unroll
        for (int s = 0; s < NST; ++s) m[s] = mp[(size_t)(4 * s) * ldm];
        __builtin_amdgcn_sched_barrier(0);
    };
    auto chunk = [&](int ch, double (&m)[NST], double (&mn)[NST]) {
        if (ch > 0) LDS_BARRIER_P();                              // the previous chunk's operand reads are done
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = q * 512 + tid;
            const int row = u / U16, c16 = u % U16;
            v2d v = ga[q];
            if (r0 + row >= nrows) v = HAS_SHIFT ? gs[q] : (v2d){0.0, 0.0};
            if (HAS_SHIFT) { v.x -= gs[q].x; v.y -= gs[q].y; }
            v.x *= alpha; v.y *= alpha;
            *reinterpret_cast<v2d*>(&As[row * LDG + 2 * c16]) = v;
        }
        LDS_BARRIER_P();
        if (ch + 1 < nch) issue(ch + 1, mn);                     // next chunk's loads fly during this chunk's operand reads and MFMAs
        const double* ap = As + c * LDG + RW * w + ks;
        double av[MT][NST];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < NST; ++s) av[mt][s] = ap[mt * 16 * LDG + 4 * s];
#pragma unroll
        for (int s = 0; s < NST; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = GSMVI_MFMA_F64(av[mt][s], m[s], acc[mt]);
    };
    double mA[NST], mB[NST];
    if (nch > 0) issue(0, mA);
    for (int ch = 0; ch < nch; ch += 2) {
        chunk(ch, mA, mB);
        if (ch + 1 < nch) chunk(ch + 1, mB, mA);
    }
    __syncthreads();
    double* red = As;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(w * NR + 16 * mt + ks + 4 * r) * 17 + c] = acc[mt][r];
    __syncthreads();
    for (int idx = tid; idx < NR * 16; idx += 512) {
        const int rr = idx >> 4, cc = idx & 15;
        const int row = r0 + rr;
        if (row < nrows) {
            double sm_ = 0.0;
#pragma unroll
            for (int ww = 0; ww < 8; ww += 2) sm_ += red[(ww * NR + rr) * 17 + cc] + red[((ww + 1) * NR + rr) * 17 + cc];
            if (Out == nullptr) Pp[((size_t)blockIdx.y * nrows + row) * ncols + blockIdx.x * 16 + cc] = sm_;
            else Out[(size_t)row * ldo + blockIdx.x * 16 + cc] = sm_ + (addvec ? addvec[blockIdx.x * 16 + cc] : 0.0);
        }
    }
#undef LDS_BARRIER_P
}

// =====================================================================================
// Per-sample stage (gsm_numpy.py:8-18): one 1024-thread workgroup per sample, every thread owns
// EPT elements of the row; all loads first, one block reduction, then the record
// rec[b] = [ d_b | e_b | dmu_b ] is written (see k_gsm_scalars for the algebra).
// =====================================================================================
template <int EPT, int KCT, int NT>
__global__ __launch_bounds__(NT) void k_gsm_scalars_fast(int D, int B, int KC, const double* __restrict__ X,
                                                         int ldx, const double* __restrict__ G, int ldg,
                                                         const double* __restrict__ mu0,
                                                         const double* __restrict__ Pp,
                                                         double* __restrict__ rec, int ldrec,
                                                         unsigned long long* __restrict__ stamps) {
    if (blockIdx.x >= GSMVI_STAMP_WG) stamps = nullptr;          // timeline diagnostic: slot capacity
    if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8] = __builtin_amdgcn_s_memrealtime();
    constexpr int NW = NT / 64;
    __shared__ double lds[2 * NW];
    const int b = blockIdx.x, tid = threadIdx.x;
    double pp[EPT][KCT], xv[EPT], gv[EPT], mv0[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + NT * e;
        const int ic = i < D ? i : D - 1;
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc)
            pp[e][kc] = Pp[((size_t)(kc < KC ? kc : KC - 1) * B + b) * D + ic];
        xv[e] = X[(size_t)b * ldx + ic];
        gv[e] = G[(size_t)b * ldg + ic];
        mv0[e] = mu0[ic];
    }
    double p0 = 0.0, p1 = 0.0, sg[EPT], dd[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + NT * e;
        double t = 0.0;
#pragma unroll
        for (int kc = 0; kc < KCT; ++kc) t += (kc < KC) ? pp[e][kc] : 0.0;
        sg[e] = t;
        dd[e] = mv0[e] - xv[e];
        if (i < D) {
            p0 += gv[e] * t;
            p1 += dd[e] * gv[e];
        }
    }
    if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memrealtime();
    p0 = wave_sum(p0);
    p1 = wave_sum(p1);
    const int w = tid >> 6;
    if ((tid & 63) == 0) {
        lds[2 * w] = p0;
        lds[2 * w + 1] = p1;
    }
    __syncthreads();
    // every thread finishes the reduction and the scalar algebra itself (same operations, same order):
    // cheaper than a second barrier around a one-thread section
    double gSg = 0.0, mv = 0.0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        gSg += lds[2 * k];
        mv += lds[2 * k + 1];
    }
    const double rho = 0.5 * sqrt(1.0 + 4.0 * (gSg + mv * mv)) - 0.5;
    const double den = 1.0 + rho + mv;
    const double beta = 1.0 / (1.0 + rho), c = (gSg - mv) / den;
    double* rb = rec + (size_t)b * ldrec;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + NT * e;
        if (i < D) {
            const double dmu = beta * ((sg[e] - dd[e]) - c * dd[e]);
            rb[i] = dd[e];
            rb[D + i] = dd[e] + dmu;
            rb[2 * D + i] = dmu;
        }
    }
    if (stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// =====================================================================================
// Symmetric rank-2B covariance update (gsm_numpy.py:21-23,50-53) from the per-sample records.
//   W = S0[I,J] + (1/B) ( Dm[:,I]^T Dm[:,J] - E[:,I]^T E[:,J] ),  S[I,J] = W,  S[J,I] = W^T
// One 512-thread workgroup owns the row block I (32 rows) and TWO adjacent column blocks J0, J1 >= I
// of the upper triangle (waves 0-3 compute tile 0, waves 4-7 tile 1: two waves per SIMD, where the
// fp64 MFMA pipe delivers ~46 TF chip-wide instead of ~34 TF with one), so the I tiles are staged
// once and every global load of the workgroup is issued in one batch.  Factor tiles are copied verbatim (16-B loads -> 16-B LDS writes, no arithmetic,
// no transposition) into LDS as [k = sample][32 columns], row stride 48 doubles: the MFMA operand
// reads (lane = column, k-slot = sample) are bank-conflict free for both operands.  The d-part and
// the e-part run as two independent accumulator chains (acc_d - acc_e), which also hides the MFMA
// latency.  S0 must be symmetric (only its upper triangle is read); S comes out exactly symmetric.
// Bytes moved: 4 D^2 read + 8 D^2 written (algorithmic count of SURVEY 8(d): 16 D^2).
// The workgroup whose first tile is diagonal also writes mu = mu0 + mean_b dmu_b.
// =====================================================================================
// RAG (round 5) = edge tiles (D % 32 != 0) and / or fewer than SB samples (B < SB): clamped re-reads, zero-staged sample rows,
// guarded stores.  RAG = false is the round-4 code plus the run-time 1/B (the clamps cost 6 % on the grid shapes when unconditional).
template <int SB, bool RAG>
__global__ __launch_bounds__(512) void k_gsm_cov_sym(int D, int B, double invB, const double* __restrict__ rec, int ldrec,
                                                     const double* __restrict__ mu0,
                                                     const double* __restrict__ S0, int lds0,
                                                     double* __restrict__ S, int lds,
                                                     double* __restrict__ mu_out, int dbg,
                                                     unsigned long long* __restrict__ stamps) {
#define STAMP(k)                                                                          \
    do {                                                                                  \
        if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    // (Round 4 tried LDS-only barriers here -- s_waitcnt lgkmcnt(0); s_barrier instead of __syncthreads(), whose workgroup fence
    // also drains the vector-memory counter: 6.45 against 6.38 us at D = 1024, B = 32, no effect; they matter in the persistent
    // form below, where loads of the next item are in flight across the barriers.)
    if (blockIdx.x >= GSMVI_STAMP_WG) stamps = nullptr;          // timeline diagnostic: slot capacity (D >= 2048 has more workgroups)
    STAMP(0);
    constexpr int RS = 48;                       // LDS row stride (doubles): 32 columns + 16 pad
    constexpr int NPASS = (SB > 32) ? SB / 32 : 1;   // samples are staged 32 at a time: <= 74 KB of LDS, so
    constexpr int SBP = SB / NPASS;              //   two workgroups fit per CU also at B = 64
    constexpr int TILE = SBP * RS;               // one staged tile (per pass)
    constexpr int NU = SB * 16;                  // 16-B units per tile over all samples
    constexpr int UPT = (6 * NU) / 512;          // units per thread over the six tiles
    static_assert((6 * NU) % 512 == 0, "tile units must divide over 512 threads");
    __shared__ __attribute__((aligned(16))) double smem[6 * TILE >= 2 * 32 * 33 ? 6 * TILE : 2 * 32 * 33];
    // staged tiles, in this order: DI, EI, DJ0, EJ0, DJ1, EJ1

    // ---- which tiles: row block ti, column blocks tj0, tj0+1 ------------------------------------
    // Row ti of the upper triangle has m = nt - ti tiles: floor(m/2) two-tile workgroups and, for odd
    // m, one single-tile workgroup (its last tile).  The two-tile workgroups come FIRST in the grid
    // (exactly 256 of them at D = 1024: one per CU); the light single-tile ones are dispatched last
    // and share a CU with a resident workgroup without stretching the kernel's tail.
    const int nt = (D + 31) >> 5;                                // any even D (round 5): edge tiles re-read clamped rows / columns
    const int n_two = ((nt >> 1) * ((nt + 1) >> 1));             // sum_m floor(m/2), m = 1..nt   // and store only what lies inside
    int ti, tj0;
    bool two;
    if ((int)blockIdx.x < n_two) {
        int rem = blockIdx.x;
        ti = 0;
        for (;;) {
            const int inrow = (nt - ti) >> 1;
            if (rem < inrow) break;
            rem -= inrow;
            ++ti;
        }
        tj0 = ti + ((nt - ti) & 1) + 2 * rem;                    // odd tile count: the pairs start right of the diagonal tile
        two = true;
    } else {
        const int k = blockIdx.x - n_two;                        // k-th row with an odd tile count: its DIAGONAL tile (no
        ti = ((nt & 1) ? 0 : 1) + 2 * k;                         // mirror store, so the late single-tile workgroups are light)
        tj0 = ti;
        two = false;
    }
    const bool diag = (tj0 == ti);
    const int I0 = ti * 32, J0 = tj0 * 32;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int t = w >> 2;                        // which tile this wave computes (two waves per SIMD)
    const int wr = (w >> 1) & 1, wc = w & 1;
    const bool mine = (t == 0) || two;           // does this wave's tile exist

    // ---- every global load of this workgroup, in one batch --------------------------------
    // S0 tile of this wave's tile t in STORE layout: 16-B units, unit = (row i, column pair j2), two per thread -- a row of
    // the tile is one 256-B segment.  (Round 2 loaded and stored 8 B per lane in accumulator layout.)
    const int Jt = J0 + 32 * ((t == 1 && two) ? 1 : 0);
    const int tl = tid & 255;
    v2d s0v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int unit = q * 256 + tl, i = unit >> 4, j2 = unit & 15;
        const int gr = (RAG && I0 + i >= D) ? D - 1 : I0 + i, gc = (RAG && Jt + 2 * j2 >= D) ? D - 2 : Jt + 2 * j2;
        s0v[q] = (dbg & 2) ? (v2d){1.0, 1.0} : *reinterpret_cast<const v2d*>(S0 + (size_t)gr * lds0 + gc);
    }
    __builtin_amdgcn_sched_barrier(0);           // keep the HBM loads of S0 ahead of the L2-resident staging loads
    v2d stg[UPT];
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
        const int g = q * 512 + tid;             // global unit over six tiles
        const int tile = g / NU, u = g % NU;     // NU is a power of two
        const int b = u >> 4, c2 = 2 * (u & 15);
        // tile 0,1: I block (d, e); 2,3: J0 block; 4,5: J1 block (= J0 again when there is no second tile)
        const int colbase = (tile < 2) ? I0 : (J0 + ((tile >= 4 && two) ? 32 : 0));
        const int colc = (RAG && colbase + c2 >= D) ? D - 2 : colbase + c2;   // (a column beyond D only feeds outputs that are not stored)
        // the 16 single-tile workgroups (they share a CU with a two-tile one) neither load nor stage a second column
        // block, and their waves 4-7 issue no MFMA: 32 instead of 64 MFMAs per SIMD on those CUs (they were the
        // kernel's 0.9 us tail: profiles/r02/timeline_cold_three_launch.txt).  tile is wave-uniform.
        // (any B <= SB: sample rows b >= B do not exist -- the address is clamped and the unit is zeroed at staging time,
        // so that no use of a loaded value sits between the loads)
        stg[q] = (two || tile < 4) ? *reinterpret_cast<const v2d*>(rec + (size_t)((RAG && b >= B) ? B - 1 : b) * ldrec + (tile & 1) * D + colc)
                                   : (v2d){0.0, 0.0};
    }
    double dmu_part = 0.0;
    if (diag && tid < 256) {                     // dmu tile for the new mean: column = tid & 31, samples tid>>5 + 8k
        double dv[SB / 8];
#pragma unroll
        for (int k = 0; k < SB / 8; ++k) {
            const int b = (tid >> 5) + 8 * k, ci = I0 + (tid & 31);
            dv[k] = rec[(size_t)((RAG && b >= B) ? B - 1 : b) * ldrec + 2 * D + ((RAG && ci >= D) ? D - 1 : ci)];
        }
#pragma unroll
        for (int k = 0; k < SB / 8; ++k) dmu_part += (!RAG || (tid >> 5) + 8 * k < B) ? dv[k] : 0.0;
    }
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(1); }

    // ---- per pass of SBP samples: factor tiles -> LDS verbatim, then MFMA.  One 16x16 block per wave;
    // chain 0 = d-part, chain 1 = e-part; operands go to registers first so the chains issue back to back.
    constexpr int NS = SBP / 4;
    v4d accd = {0.0, 0.0, 0.0, 0.0}, acce = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass > 0) __syncthreads();           // the previous pass's operand reads are done
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int g = q * 512 + tid;
            const int tile = g / NU, u = g % NU;
            const int b = u >> 4;
            if (b / SBP == pass && (two || tile < 4))
                *reinterpret_cast<v2d*>(smem + tile * TILE + (b % SBP) * RS + 2 * (u & 15)) = (!RAG || b < B) ? stg[q] : (v2d){0.0, 0.0};
        }
        __syncthreads();
        if (pass == 0) STAMP(2);
        double ad[NS], ae[NS], bd[NS], be[NS];
        if (mine) {
            const double* adp = smem + ks * RS + 16 * wr + c;
            const double* aep = adp + TILE;
            const double* bdp = smem + (2 + 2 * t) * TILE + ks * RS + 16 * wc + c;
            const double* bep = bdp + TILE;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                ad[s] = adp[4 * s * RS];
                ae[s] = aep[4 * s * RS];
                bd[s] = bdp[4 * s * RS];
                be[s] = bep[4 * s * RS];
            }
        }
        if (mine && !(dbg & 8)) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                accd = GSMVI_MFMA_F64(ad[s], bd[s], accd);
                acce = GSMVI_MFMA_F64(ae[s], be[s], acce);
            }
        }
    }
    // ---- stores through LDS, 16 B per lane: the update tile U_t = (D_I^T D_J - E_I^T E_J)/B goes to LDS in accumulator
    // layout; W_t = S0[I, J_t] + U_t is formed in store layout (row segments of 256 B), stored, written back to LDS, and
    // the mirror S[J_t, I] = W_t^T is read from there by columns (row stride 33: conflict-free both ways)
    __syncthreads();                             // everyone is done reading the factor tiles
    double* LW = smem + t * 32 * 33;             // [2][32 x 33]
    if (mine) {
#pragma unroll
        for (int r = 0; r < 4; ++r) LW[(16 * wr + ks + 4 * r) * 33 + 16 * wc + c] = (accd[r] - acce[r]) * invB;
    }
    if (stamps) STAMP(3);
    __syncthreads();
    if (mine) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int unit = q * 256 + tl, i = unit >> 4, j2 = unit & 15;
            v2d wv2;
            wv2.x = s0v[q].x + LW[i * 33 + 2 * j2];
            wv2.y = s0v[q].y + LW[i * 33 + 2 * j2 + 1];
            if (!RAG || (I0 + i < D && Jt + 2 * j2 < D)) *reinterpret_cast<v2d*>(S + (size_t)(I0 + i) * lds + Jt + 2 * j2) = wv2;
            LW[i * 33 + 2 * j2] = wv2.x;
            LW[i * 33 + 2 * j2 + 1] = wv2.y;
        }
    }
    const bool need = mine && !(t == 0 && diag) && !(dbg & 1);
    __syncthreads();
    if (need) {
        // unit = (row j of J_t, column pair i2 of I): S[J_t + j][I0 + 2 i2 .. + 1] = W_t[2 i2 .. + 1][j]
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int unit = q * 256 + tl, j = unit >> 4, i2 = unit & 15;
            v2d m2;
            m2.x = LW[(2 * i2) * 33 + j];
            m2.y = LW[(2 * i2 + 1) * 33 + j];
            if (!RAG || (Jt + j < D && I0 + 2 * i2 < D)) *reinterpret_cast<v2d*>(S + (size_t)(Jt + j) * lds + I0 + 2 * i2) = m2;
        }
    }
    if (diag) {
        __syncthreads();
        if (tid < 256) smem[tid] = dmu_part;     // [8][32]
        __syncthreads();
        if (tid < 32) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += smem[q * 32 + tid];
            if (!RAG || I0 + tid < D) mu_out[I0 + tid] = mu0[I0 + tid] + s * invB;
        }
    }
    STAMP(4);
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(5); }
#undef STAMP
}

// =====================================================================================
// PERSISTENT form of k_gsm_cov_sym for large D (round 4).  Same tiles, same arithmetic, same results bit for bit: a work ITEM
// is what one workgroup of k_gsm_cov_sym does (row block I of 32 rows x two adjacent column tiles, or a lone diagonal tile).
// At D = 4096 there are 4160 items and the one-item kernel ran them as 4160 workgroups, two resident per CU, each a serial
// chain load (16 KB of S0 from HBM + 48 KB of record tiles from L2) -> LDS -> MFMA -> stores (32 KB): 53.9 us for 201 MB =
// 3.7 TB/s, 0.59 of what a plain copy reaches on the box -- the memory pipes of a CU idle while its workgroups compute.
// Here 2 workgroups per CU stay resident and walk the item list (item = blockIdx.x, + gridDim.x, ...); the NEXT item's global
// loads (8 v2d per thread: two S0 units, six record units) are issued as soon as the current item's record tiles have gone
// to LDS, so they are in flight during the current item's operand reads, MFMAs, LDS round trips and stores (loads and stores
// share the in-order vmcnt counter: the next iteration waits for the loads only, the stores behind them stay in flight).
// =====================================================================================
template <int SB, bool RAG>
__global__ __launch_bounds__(512) void k_gsm_cov_sym_p(int D, int B, double invB, const double* __restrict__ rec, int ldrec,
                                                       const double* __restrict__ mu0,
                                                       const double* __restrict__ S0, int lds0,
                                                       double* __restrict__ S, int lds,
                                                       double* __restrict__ mu_out) {
    // Workgroup barrier that waits for this wave's LDS operations ONLY.  __syncthreads() carries a workgroup-scope fence, which
    // on gfx950 drains the vector-memory counter as well (s_waitcnt vmcnt(0)): every one of the five barriers of an item would
    // wait for the NEXT item's loads issued in front of it -- the first version of this kernel did exactly that and ran 27 %
    // slower than the one-item kernel.  No thread of the workgroup reads global data another thread of it wrote, so the barriers
    // only have to order LDS.
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    constexpr int RS = 48;
    constexpr int NPASS = (SB > 32) ? SB / 32 : 1;
    constexpr int SBP = SB / NPASS;
    constexpr int TILE = SBP * RS;
    constexpr int NU = SB * 16;
    constexpr int UPT = (6 * NU) / 512;
    static_assert((6 * NU) % 512 == 0, "tile units must divide over 512 threads");
    __shared__ __attribute__((aligned(16))) double smem[6 * TILE >= 2 * 32 * 33 ? 6 * TILE : 2 * 32 * 33];
    const int nt = (D + 31) >> 5;                            // (any even D: edge tiles as in k_gsm_cov_sym)
    const int n_two = ((nt >> 1) * ((nt + 1) >> 1));
    const int n_items = n_two + ((nt + 1) >> 1);             // + one lone diagonal tile per row with an odd tile count
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int t = w >> 2;
    const int wr = (w >> 1) & 1, wc = w & 1;
    const int tl = tid & 255;

    struct Item { int ti, tj0; bool two; };
    // incremental decode of the two-tile items (row ti holds (nt - ti) >> 1 of them): the scan continues where the last one stopped
    int scan_ti = 0, scan_base = 0;                          // scan_base = first item index of row scan_ti
    auto decode = [&](int item) -> Item {
        Item it;
        if (item < n_two) {
            for (;;) {
                const int inrow = (nt - scan_ti) >> 1;
                if (item - scan_base < inrow) break;
                scan_base += inrow;
                ++scan_ti;
            }
            it.ti = scan_ti;
            it.tj0 = scan_ti + ((nt - scan_ti) & 1) + 2 * (item - scan_base);
            it.two = true;
        } else {
            const int k = item - n_two;
            it.ti = ((nt & 1) ? 0 : 1) + 2 * k;
            it.tj0 = it.ti;
            it.two = false;
        }
        return it;
    };
    v2d s0v[2], stg[UPT];
    double dmuv[SB / 8];                                     // (kept as loaded: summing here would wait for the loads just issued)
    auto issue_loads = [&](const Item& it, v2d (&s0)[2], double (&dmu)[SB / 8]) {
        const int I0 = it.ti * 32, J0 = it.tj0 * 32;
        const int Jt = J0 + 32 * ((t == 1 && it.two) ? 1 : 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int unit = q * 256 + tl, i = unit >> 4, j2 = unit & 15;
            const int gr = (RAG && I0 + i >= D) ? D - 1 : I0 + i, gc = (RAG && Jt + 2 * j2 >= D) ? D - 2 : Jt + 2 * j2;
            s0[q] = *reinterpret_cast<const v2d*>(S0 + (size_t)gr * lds0 + gc);
        }
        __builtin_amdgcn_sched_barrier(0);                   // the HBM loads of S0 ahead of the L2-resident record loads
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int g = q * 512 + tid;
            const int tile = g / NU, u = g % NU;
            const int b = u >> 4, c2 = 2 * (u & 15);
            const int colbase = (tile < 2) ? I0 : (J0 + ((tile >= 4 && it.two) ? 32 : 0));
            const int colc = (RAG && colbase + c2 >= D) ? D - 2 : colbase + c2;
            stg[q] = (it.two || tile < 4) ? *reinterpret_cast<const v2d*>(rec + (size_t)((RAG && b >= B) ? B - 1 : b) * ldrec + (tile & 1) * D + colc)
                                          : (v2d){0.0, 0.0};
        }
#pragma unroll
        for (int k = 0; k < SB / 8; ++k) dmu[k] = 0.0;
        if (it.tj0 == it.ti && tid < 256) {
#pragma unroll
            for (int k = 0; k < SB / 8; ++k) {
                const int b = (tid >> 5) + 8 * k, ci = I0 + (tid & 31);
                dmu[k] = rec[(size_t)((RAG && b >= B) ? B - 1 : b) * ldrec + 2 * D + ((RAG && ci >= D) ? D - 1 : ci)];
            }
        }
    };
    int item = blockIdx.x;
    if (item >= n_items) return;
    Item cur = decode(item);
    issue_loads(cur, s0v, dmuv);
    while (true) {
        const bool two = cur.two, diag = (cur.tj0 == cur.ti);
        const int I0 = cur.ti * 32, J0 = cur.tj0 * 32;
        const int Jt = J0 + 32 * ((t == 1 && two) ? 1 : 0);
        const bool mine = (t == 0) || two;
        constexpr int NS = SBP / 4;
        v4d accd = {0.0, 0.0, 0.0, 0.0}, acce = {0.0, 0.0, 0.0, 0.0};
        const int nxt = item + gridDim.x;
        Item nx = cur;
        v2d s0n[2] = {s0v[0], s0v[1]};
        double dmun[SB / 8];
#pragma unroll
        for (int k = 0; k < SB / 8; ++k) dmun[k] = 0.0;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (pass > 0) LDS_BARRIER();
#pragma unroll
            for (int q = 0; q < UPT; ++q) {
                const int g = q * 512 + tid;
                const int tile = g / NU, u = g % NU;
                const int b = u >> 4;
                if (b / SBP == pass && (two || tile < 4))
                    *reinterpret_cast<v2d*>(smem + tile * TILE + (b % SBP) * RS + 2 * (u & 15)) = (!RAG || b < B) ? stg[q] : (v2d){0.0, 0.0};
            }
            LDS_BARRIER();
            if (pass == NPASS - 1 && nxt < n_items) {        // the staging registers are free: the next item's loads go out now
                nx = decode(nxt);
                issue_loads(nx, s0n, dmun);
            }
            double ad[NS], ae[NS], bd[NS], be[NS];
            if (mine) {
                const double* adp = smem + ks * RS + 16 * wr + c;
                const double* aep = adp + TILE;
                const double* bdp = smem + (2 + 2 * t) * TILE + ks * RS + 16 * wc + c;
                const double* bep = bdp + TILE;
#pragma unroll
                for (int sI = 0; sI < NS; ++sI) {
                    ad[sI] = adp[4 * sI * RS];
                    ae[sI] = aep[4 * sI * RS];
                    bd[sI] = bdp[4 * sI * RS];
                    be[sI] = bep[4 * sI * RS];
                }
#pragma unroll
                for (int sI = 0; sI < NS; ++sI) {
                    accd = GSMVI_MFMA_F64(ad[sI], bd[sI], accd);
                    acce = GSMVI_MFMA_F64(ae[sI], be[sI], acce);
                }
            }
        }
        LDS_BARRIER();                                     // everyone is done reading the factor tiles
        double* LW = smem + t * 32 * 33;
        if (mine) {
#pragma unroll
            for (int r = 0; r < 4; ++r) LW[(16 * wr + ks + 4 * r) * 33 + 16 * wc + c] = (accd[r] - acce[r]) * invB;
        }
        LDS_BARRIER();
        if (mine) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int unit = q * 256 + tl, i = unit >> 4, j2 = unit & 15;
                v2d wv2;
                wv2.x = s0v[q].x + LW[i * 33 + 2 * j2];
                wv2.y = s0v[q].y + LW[i * 33 + 2 * j2 + 1];
                if (!RAG || (I0 + i < D && Jt + 2 * j2 < D)) *reinterpret_cast<v2d*>(S + (size_t)(I0 + i) * lds + Jt + 2 * j2) = wv2;
                LW[i * 33 + 2 * j2] = wv2.x;
                LW[i * 33 + 2 * j2 + 1] = wv2.y;
            }
        }
        const bool need = mine && !(t == 0 && diag);
        LDS_BARRIER();
        if (need) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int unit = q * 256 + tl, j = unit >> 4, i2 = unit & 15;
                v2d m2;
                m2.x = LW[(2 * i2) * 33 + j];
                m2.y = LW[(2 * i2 + 1) * 33 + j];
                if (!RAG || (Jt + j < D && I0 + 2 * i2 < D)) *reinterpret_cast<v2d*>(S + (size_t)(Jt + j) * lds + I0 + 2 * i2) = m2;
            }
        }
        if (diag) {                                          // (block-uniform)
            LDS_BARRIER();
            if (tid < 256) {
                double dsum = 0.0;
#pragma unroll
                for (int k = 0; k < SB / 8; ++k) dsum += (!RAG || (tid >> 5) + 8 * k < B) ? dmuv[k] : 0.0;
                smem[tid] = dsum;
            }
            LDS_BARRIER();
            if (tid < 32) {
                double sm_ = 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) sm_ += smem[q * 32 + tid];
                if (!RAG || I0 + tid < D) mu_out[I0 + tid] = mu0[I0 + tid] + sm_ * invB;
            }
        }
        if (nxt >= n_items) break;
        LDS_BARRIER();                                     // the LDS tiles of this item are dead: the next item may stage
        item = nxt;
        cur = nx;
        s0v[0] = s0n[0];
        s0v[1] = s0n[1];
#pragma unroll
        for (int k = 0; k < SB / 8; ++k) dmuv[k] = dmun[k];
    }
}

// =====================================================================================
// k_gsm_cov_sym for ANY batch size (round 6; until then B <= 128 only and B = 130 fell to the guarded kernel of
// gsmvi_kernels.hip).  Same items, same tile layout, same store path as k_gsm_cov_sym; the samples come in a RUN-TIME loop of
// 32-sample passes instead of SB / 32 compile-time passes over registers loaded up front:
//   * the six record tiles of a pass are exactly one 16-byte unit per thread and tile (32 samples x 16 units = 512);
//   * two register sets (even / odd passes): the loads of pass p + 2 are issued as soon as pass p has gone to LDS, so they fly
//     during two passes of operand reads and MFMAs (~1 us each on a CU: 16 fp64 MFMAs per wave, two waves per SIMD);
//   * LDS-only barriers (s_waitcnt lgkmcnt(0); s_barrier) as in the persistent form: __syncthreads() would drain the prefetch;
//   * sample rows b >= B of the last pass are clamped re-reads staged as zeros; edge tiles (D % 32 != 0) as RAG = true above.
// MFMA-bound (2 B D^2 flop on the upper triangle: 1.07 GFLOP at D = 1024, B = 512 against 17 MB of covariance), one item per
// workgroup; the mean's dmu tile rides in the same passes on the diagonal workgroups.
// =====================================================================================
__global__ __launch_bounds__(512) void k_gsm_cov_sym_big(int D, int B, double invB, const double* __restrict__ rec, int ldrec,
                                                         const double* __restrict__ mu0,
                                                         const double* __restrict__ S0, int lds0,
                                                         double* __restrict__ S, int lds,
                                                         double* __restrict__ mu_out) {
    constexpr int RS = 48;
    constexpr int TILE = 32 * RS;
    __shared__ __attribute__((aligned(16))) double smem[6 * TILE];
    const int nt = (D + 31) >> 5;
    const int n_two = ((nt >> 1) * ((nt + 1) >> 1));
    int ti, tj0;
    bool two;
    if ((int)blockIdx.x < n_two) {
        int rem = blockIdx.x;
        ti = 0;
        for (;;) {
            const int inrow = (nt - ti) >> 1;
            if (rem < inrow) break;
            rem -= inrow;
            ++ti;
        }
        tj0 = ti + ((nt - ti) & 1) + 2 * rem;
        two = true;
    } else {
        const int k = blockIdx.x - n_two;
        ti = ((nt & 1) ? 0 : 1) + 2 * k;
        tj0 = ti;
        two = false;
    }
    const bool diag = (tj0 == ti);
    const int I0 = ti * 32, J0 = tj0 * 32;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int t = w >> 2;
    const int wr = (w >> 1) & 1, wc = w & 1;
    const bool mine = (t == 0) || two;
    const int Jt = J0 + 32 * ((t == 1 && two) ? 1 : 0);
    const int tl = tid & 255;
    v2d s0v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int unit = q * 256 + tl, i = unit >> 4, j2 = unit & 15;
        const int gr = (I0 + i >= D) ? D - 1 : I0 + i, gc = (Jt + 2 * j2 >= D) ? D - 2 : Jt + 2 * j2;
        s0v[q] = *reinterpret_cast<const v2d*>(S0 + (size_t)gr * lds0 + gc);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int npass = (B + 31) >> 5;
    const int sb = tid >> 4, sc2 = 2 * (tid & 15);               // this thread's unit of every tile: sample row, column pair
    const int dcol = (I0 + (tid & 31) >= D) ? D - 1 : I0 + (tid & 31);
    auto issue = [&](int pass, v2d (&sg)[6], double (&dm)[4]) {
        const int bb = pass * 32 + sb;
        const double* rp = rec + (size_t)(bb < B ? bb : B - 1) * ldrec;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int colbase = (q < 2) ? I0 : (J0 + ((q >= 4 && two) ? 32 : 0));
            const int colc = (colbase + sc2 >= D) ? D - 2 : colbase + sc2;
            sg[q] = (two || q < 4) ? *reinterpret_cast<const v2d*>(rp + (q & 1) * D + colc) : (v2d){0.0, 0.0};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) dm[k] = 0.0;
        if (diag && tid < 256) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int b = pass * 32 + (tid >> 5) + 8 * k;
                dm[k] = rec[(size_t)(b < B ? b : B - 1) * ldrec + 2 * D + dcol];
            }
        }
    };
    v4d accd = {0.0, 0.0, 0.0, 0.0}, acce = {0.0, 0.0, 0.0, 0.0};
    double dmu_part = 0.0;
    auto do_pass = [&](int pass, v2d (&sg)[6], double (&dm)[4]) {
        if (pass > 0) LDS_BARRIER();                             // the previous pass's operand reads are done
        const bool live = pass * 32 + sb < B;
#pragma unroll
        for (int q = 0; q < 6; ++q)
            if (two || q < 4) *reinterpret_cast<v2d*>(smem + q * TILE + sb * RS + sc2) = live ? sg[q] : (v2d){0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k) dmu_part += (pass * 32 + (tid >> 5) + 8 * k < B) ? dm[k] : 0.0;
        LDS_BARRIER();
        if (pass + 2 < npass) issue(pass + 2, sg, dm);           // this register set is free: two passes ahead
        if (mine) {
            double ad[8], ae[8], bd[8], be[8];
            const double* adp = smem + ks * RS + 16 * wr + c;
            const double* aep = adp + TILE;
            const double* bdp = smem + (2 + 2 * t) * TILE + ks * RS + 16 * wc + c;
            const double* bep = bdp + TILE;
#pragma unroll
            for (int sI = 0; sI < 8; ++sI) {
                ad[sI] = adp[4 * sI * RS];
                ae[sI] = aep[4 * sI * RS];
                bd[sI] = bdp[4 * sI * RS];
                be[sI] = bep[4 * sI * RS];
            }
#pragma unroll
            for (int sI = 0; sI < 8; ++sI) {
                accd = GSMVI_MFMA_F64(ad[sI], bd[sI], accd);
                acce = GSMVI_MFMA_F64(ae[sI], be[sI], acce);
            }
        }
    };
    v2d stgA[6], stgB[6];
    double dmA[4], dmB[4];
    issue(0, stgA, dmA);
    if (npass > 1) issue(1, stgB, dmB);
    for (int pass = 0; pass < npass; pass += 2) {
        do_pass(pass, stgA, dmA);
        if (pass + 1 < npass) do_pass(pass + 1, stgB, dmB);
    }
    // ---- stores: as k_gsm_cov_sym (update tile through LDS, S0 added in store layout, mirror by columns) ----
    LDS_BARRIER();
    double* LW = smem + t * 32 * 33;
    if (mine) {
#pragma unroll
        for (int r = 0; r < 4; ++r) LW[(16 * wr + ks + 4 * r) * 33 + 16 * wc + c] = (accd[r] - acce[r]) * invB;
    }
    LDS_BARRIER();
    if (mine) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int unit = q * 256 + tl, i = unit >> 4, j2 = unit & 15;
            v2d wv2;
            wv2.x = s0v[q].x + LW[i * 33 + 2 * j2];
            wv2.y = s0v[q].y + LW[i * 33 + 2 * j2 + 1];
            if (I0 + i < D && Jt + 2 * j2 < D) *reinterpret_cast<v2d*>(S + (size_t)(I0 + i) * lds + Jt + 2 * j2) = wv2;
            LW[i * 33 + 2 * j2] = wv2.x;
            LW[i * 33 + 2 * j2 + 1] = wv2.y;
        }
    }
    const bool need = mine && !(t == 0 && diag);
    LDS_BARRIER();
    if (need) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int unit = q * 256 + tl, j = unit >> 4, i2 = unit & 15;
            v2d m2;
            m2.x = LW[(2 * i2) * 33 + j];
            m2.y = LW[(2 * i2 + 1) * 33 + j];
            if (Jt + j < D && I0 + 2 * i2 < D) *reinterpret_cast<v2d*>(S + (size_t)(Jt + j) * lds + I0 + 2 * i2) = m2;
        }
    }
    if (diag) {
        LDS_BARRIER();
        if (tid < 256) smem[tid] = dmu_part;
        LDS_BARRIER();
        if (tid < 32) {
            double sm_ = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) sm_ += smem[q * 32 + tid];
            if (I0 + tid < D) mu_out[I0 + tid] = mu0[I0 + tid] + sm_ * invB;
        }
    }
}
#undef LDS_BARRIER

// ---- launch helpers ------------------------------------------------------------------------
void gsmvi_launch_panel_fast(hipStream_t st, hipEvent_t* ev, int MT, dim3 grid, int D, int nrows, const double* A,
                             int lda, const double* shift, double alpha, const double* M, int ldm, double* Pp,
                             int chunks_per_wg, int ncols, unsigned long long* stamps, double* Out, int ldo,
                             const double* addvec, const gsmvi_panel_extras* px) {
    const gsmvi_panel_extras none;
    const bool rider = px && px->rd_on && !shift;
    const bool extra = px && (px->msl || px->sj_src || rider);
    const gsmvi_panel_extras pxv = extra ? *px : none;
    const bool rag = D % 64 != 0 || ncols % 16 != 0;           // off the grid: the clamped instantiation (round 5)
    if (grid.x == 0) return;
    if (px && px->w4 && !rider && !extra && MT <= 2) {           // the prefetching form (the caller checked its preconditions)
#define PFP(MTV, HS) GSMVI_LAUNCH((k_panel_fast_p<MTV, HS>), grid, dim3(512), 0, st, ev, D, nrows, A, lda, shift, alpha, M, ldm, \
                                  Pp, chunks_per_wg, ncols, Out, ldo, addvec)
        if (shift) { if (MT == 1) PFP(1, true); else PFP(2, true); }
        else { if (MT == 1) PFP(1, false); else PFP(2, false); }
#undef PFP
        return;
    }
    if (rider) {
        grid.x += 1;
#define PFR(MTV, CW, RG)                                                                                                 \
    GSMVI_LAUNCH((k_panel_fast<MTV, false, CW, true, true, RG>), grid, dim3(512), 0, st, ev, D, nrows, A, lda, shift, alpha, M, \
                 ldm, Pp, chunks_per_wg, ncols, stamps, Out, ldo, addvec, pxv)
        if (rag) { if (MT == 1) PFR(1, 256, true); else if (MT == 2) PFR(2, 256, true); else PFR(4, 128, true); }
        else { if (MT == 1) PFR(1, 256, false); else if (MT == 2) PFR(2, 256, false); else PFR(4, 128, false); }
#undef PFR
        return;
    }
#define PF(MTV, HS, CW, EX, RG)                                                                                          \
    GSMVI_LAUNCH((k_panel_fast<MTV, HS, CW, EX, false, RG>), grid, dim3(512), 0, st, ev, D, nrows, A, lda, shift, alpha, M, ldm, \
                 Pp, chunks_per_wg, ncols, stamps, Out, ldo, addvec, pxv)
#define PFE(MTV, HS, CW)                                                                \
    do {                                                                                \
        if (rag) { if (extra) PF(MTV, HS, CW, true, true); else PF(MTV, HS, CW, false, true); } \
        else { if (extra) PF(MTV, HS, CW, true, false); else PF(MTV, HS, CW, false, false); }   \
    } while (0)
    if (shift) {
        if (MT == 1) PFE(1, true, 256); else if (MT == 2) PFE(2, true, 256); else PFE(4, true, 128);
    } else {
        if (MT == 1) PFE(1, false, 256); else if (MT == 2) PFE(2, false, 256); else PFE(4, false, 128);
    }
#undef PFE
#undef PF
}

// column width of one staged chunk for a given MT (the ABI's chunk arithmetic must agree)
int gsmvi_panel_fast_chunk(int MT) { return MT == 4 ? 128 : 256; }

// returns false when (D, KC) has no instantiation.  nt = threads per sample-workgroup (tuning knob scalars_nt)
bool gsmvi_launch_gsm_scalars_fast(hipStream_t st, hipEvent_t* ev, int D, int B, int KC, const double* X, int ldx,
                                   const double* G, int ldg, const double* mu0, const double* Pp, double* rec,
                                   int ldrec, int nt, unsigned long long* stamps) {
    if (nt != 256 && nt != 512 && nt != 1024) nt = 512;   // 512 measured best at D=1024 (fewer waves to reduce)
    const int ept = (D + nt - 1) / nt;
#define SF(E, K, N)                                                                                              \
    GSMVI_LAUNCH((k_gsm_scalars_fast<E, K, N>), dim3(B), dim3(N), 0, st, ev, D, B, KC, X, ldx, G, ldg, mu0, Pp, rec, \
                 ldrec, stamps)
#define SFK(E, N) do { if (kct == 1) SF(E, 1, N); else if (kct == 2) SF(E, 2, N); else if (kct == 4) SF(E, 4, N); else SF(E, 8, N); } while (0)
#define SFE(N) do { if (e == 1) SFK(1, N); else if (e == 2) SFK(2, N); else if (e == 4) SFK(4, N); else SFK(8, N); } while (0)
    if (KC > 8 || ept > 8) return false;
    const int kct = KC <= 1 ? 1 : (KC <= 2 ? 2 : (KC <= 4 ? 4 : 8));
    const int e = ept <= 1 ? 1 : (ept <= 2 ? 2 : (ept <= 4 ? 4 : 8));
    if (nt == 256) SFE(256); else if (nt == 512) SFE(512); else SFE(1024);
#undef SFE
#undef SFK
#undef SF
    return true;
}

// number of workgroups of k_gsm_cov_sym: row ti of the upper triangle holds ceil((nt - ti)/2)
static int cov_sym_grid(int nt) {
    int n = 0;
    for (int ti = 0; ti < nt; ++ti) n += (nt - ti + 1) / 2;
    return n;
}

bool gsmvi_launch_gsm_cov_sym(hipStream_t st, hipEvent_t* ev, int D, int B, const double* rec, int ldrec,
                              const double* mu0, const double* S0, int lds0, double* S, int lds, double* mu_out,
                              int dbg, unsigned long long* stamps, int num_cu) {
    // Any batch size up to 128 (round 5): the kernel is instantiated for SB = 16, 32, 64, 128 staged sample rows and takes the
    // actual B and 1/B at run time -- rows b >= B are staged as zeros (they add nothing to either accumulator chain), so
    // B = 20 runs the SB = 32 instance at the speed of B = 32.  (Until round 4: B in {16, 32, 64} only, everything else fell to
    // the guarded kernel of gsmvi_kernels.hip.)
    if (B < 1) return false;
    if (B > 128) {          // round 6: any batch size -- the run-time pass loop (k_gsm_cov_sym_big); B <= 128 keeps the instances below
        GSMVI_LAUNCH(k_gsm_cov_sym_big, dim3(cov_sym_grid((D + 31) / 32)), dim3(512), 0, st, ev, D, B, 1.0 / (double)B, rec, ldrec,
                     mu0, S0, lds0, S, lds, mu_out);
        return true;
    }
    const int SB = B <= 16 ? 16 : (B <= 32 ? 32 : (B <= 64 ? 64 : 128));
    const double invB = 1.0 / (double)B;
    const bool rag = D % 32 != 0 || B != SB;                     // off the grid: the clamped instantiation
    // large D: the persistent form (2 resident workgroups per CU walk the item list, the next item's loads in flight during
    // the current item's MFMAs and stores); "cov_dbg" bit 512 keeps the one-item-per-workgroup kernel for A/B runs
    const int n_items = cov_sym_grid((D + 31) / 32);
    if (n_items >= 2048 && SB <= 32 && !stamps && dbg == 0) {    // (B = 64: two staging passes, MFMA-bound -- the one-item kernel is faster there)
        const dim3 pgrid(2 * (num_cu > 0 ? num_cu : 256));       // two resident workgroups per CU of THIS device (a partitioned device has fewer)
#define CSP(SBV, RG) GSMVI_LAUNCH((k_gsm_cov_sym_p<SBV, RG>), pgrid, dim3(512), 0, st, ev, D, B, invB, rec, ldrec, mu0, S0, lds0, S, lds, mu_out)
        if (rag) { if (SB == 16) CSP(16, true); else CSP(32, true); }
        else { if (SB == 16) CSP(16, false); else CSP(32, false); }
        return true;
#undef CSP
    }
    if (dbg & 512) dbg &= ~512;
    const dim3 grid(n_items);
#define CS(SBV, RG)                                                                                         \
    GSMVI_LAUNCH((k_gsm_cov_sym<SBV, RG>), grid, dim3(512), 0, st, ev, D, B, invB, rec, ldrec, mu0, S0, lds0, S, lds, mu_out, dbg, \
                 stamps)
    if (rag) {
        switch (SB) {
            case 16: CS(16, true); break;
            case 32: CS(32, true); break;
            case 64: CS(64, true); break;
            default: CS(128, true); break;
        }
    } else {
        switch (SB) {
            case 16: CS(16, false); break;
            case 32: CS(32, false); break;
            case 64: CS(64, false); break;
            default: CS(128, false); break;
        }
    }
#undef CS
    return true;
}
