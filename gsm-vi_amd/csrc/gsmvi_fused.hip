// Two-launch form of the dense-covariance GSM update for gfx950 (gsm_numpy.py:4-55): the per-sample stage of the
// three-launch path (k_gsm_scalars_fast: its own launch, a 24 B D record write and read) is folded into its
// neighbours.
//
//   k_panel_seam      SG = G S0 as in k_panel_fast (split-K over blockIdx.y), then an in-kernel seam per 16-column
//                     strip: the LAST of the strip's KC workgroups to arrive (one agent-scope ticket per workgroup)
//                     sums the KC partial pieces in fixed order, writes the finished strip of SG and the strip's
//                     partial dots  pd[strip][b] = ( sum_j g_bj SG_bj ,  sum_j (mu0_j - x_bj) g_bj ),  j in strip.
//   k_gsm_cov_fused   every workgroup first reduces the D/16 partial dots of all B samples (fixed order) to
//                     rho_b, beta_b, c_b (gsm_numpy.py:8-10,15), then forms its d- and e-tiles on the fly from X and
//                     SG while staging them in LDS; the rest is k_gsm_cov_sym (gsmvi_fast.hip).
//
// Hand-off inside k_panel_seam: the measured-valid form of MI355X_MICROARCH.md ("Workgroup dispatch, XCD placement &
// inter-workgroup visibility", table row 1): every byte of a piece is stored with an sc1 (write-through) store, every
// storing wave drains with s_waitcnt vmcnt(0), workgroup barrier, ONE lane adds to the strip's counter with an
// agent-scope returning atomic; the workgroup whose add returned KC-1 reads all pieces with sc1 loads after a
// workgroup barrier.  The ABI selects this path only while the panel grid has at most one workgroup per CU (the
// geometry that form was measured in); tests/test_gpu_fused.py runs it under uneven load against the three-launch
// path, which stays the reference (tuning knob fused=0) and the only path of the batch-sharded stages.
#include "gsmvi_common.h"
#include "gsmvi_ctx.h"
#include <hip/hip_ext.h>

#define GSMVI_LAUNCH(kern, grid, block, shmem, st, ev, ...)                                         \
    do {                                                                                           \
        if (ev)                                                                                    \
            hipExtLaunchKernelGGL(kern, grid, block, shmem, st, (ev)[0], (ev)[1], 0, __VA_ARGS__); \
        else                                                                                       \
            hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__);                         \
    } while (0)

__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// sum over the 16 lanes that share l >> 4 (fixed butterfly order; every lane gets the result)
__device__ __forceinline__ double sum16(double v) { return row16_sum(v); }

// =====================================================================================
// SG = G S0 with the per-strip seam.  Grid (D/16 strips, KC, 1), 512 threads; B = 16 MT samples, D % 64 == 0,
// even leading dimensions, 16-byte aligned bases (checked by the ABI).  Same product loop as k_panel_fast.
// =====================================================================================
template <int MT, int CHW>
__global__ __launch_bounds__(512) void k_panel_seam(int D, const double* __restrict__ G, int ldg,
                                                    const double* __restrict__ S0, int lds0,
                                                    const double* __restrict__ X, int ldx,
                                                    const double* __restrict__ mu0, double* Pp,
                                                    double* __restrict__ SG, double* __restrict__ pd,
                                                    unsigned* cnt, int chunks_per_wg,
                                                    unsigned long long* __restrict__ stamps) {
#define PSTAMP(k)                                                                                          \
    do {                                                                                                   \
        if (stamps && threadIdx.x == 0)                                                                    \
            stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    PSTAMP(0);
    constexpr int LDG = CHW + 2;
    constexpr int NR = 16 * MT;                    // = B
    constexpr int RW = CHW / 8;
    constexpr int NST = RW / 4;
    constexpr int U16 = CHW / 2;
    constexpr int UPT = NR * U16 / 512;
    constexpr int EP = (NR * 16 + 511) / 512;      // piece elements per thread
    constexpr int SMEM = (NR * LDG > 8 * NR * 17) ? NR * LDG : 8 * NR * 17;
    __shared__ __attribute__((aligned(16))) double As[SMEM];
    __shared__ unsigned s_ticket;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int j = blockIdx.x * 16 + c;

    v4d acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4d){0.0, 0.0, 0.0, 0.0};
    // operands of the strip's partial dots, fetched up front so that the seam's tail has no dependent load
    double pg[EP], pdv[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) {
        const int idx = tid + 512 * e;
        const int rr = (idx < NR * 16 ? idx : 0) >> 4, col = blockIdx.x * 16 + (idx & 15);
        pg[e] = G[(size_t)rr * ldg + col];
        pdv[e] = mu0[col] - X[(size_t)rr * ldx + col];
    }

    for (int ch = 0; ch < chunks_per_wg; ++ch) {
        const int cbase = (blockIdx.y * chunks_per_wg + ch) * CHW;
        if (cbase >= D) break;
        const int wbase = cbase + w * RW;
        const bool wave_in = wbase < D;
        v2d ga[UPT];
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = q * 512 + tid;
            const int row = u / U16, c16 = u % U16;
            const int col = cbase + 2 * c16;
            ga[q] = *reinterpret_cast<const v2d*>(G + (size_t)row * ldg + (col < D ? col : 0));
            if (col >= D) ga[q] = (v2d){0.0, 0.0};
        }
        __builtin_amdgcn_sched_barrier(0);
        double m[NST];
        {
            const double* mp = S0 + (size_t)((wave_in ? wbase : 0) + ks) * lds0 + j;
#pragma unroll
            for (int s = 0; s < NST; ++s) m[s] = mp[(size_t)(4 * s) * lds0];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ch > 0) __syncthreads();
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = q * 512 + tid;
            const int row = u / U16, c16 = u % U16;
            *reinterpret_cast<v2d*>(&As[row * LDG + 2 * c16]) = ga[q];
        }
        __syncthreads();
        PSTAMP(1);
        if (wave_in) {
            const double* ap = As + c * LDG + RW * w + ks;
            double av[MT][NST];
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

    // cross-wave reduction through LDS (the order of k_panel_fast: pieces are bit-identical to its slabs)
    __syncthreads();
    PSTAMP(2);
    double* red = As;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(w * NR + 16 * mt + ks + 4 * r) * 17 + c] = acc[mt][r];
    __syncthreads();
    const int KC = gridDim.y;
    double sg[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) {
        const int idx = tid + 512 * e;
        sg[e] = 0.0;
        if (idx < NR * 16) {
            const int rr = idx >> 4, cc = idx & 15;
            double s = 0.0;
#pragma unroll
            for (int ww = 0; ww < 8; ww += 2) s += red[(ww * NR + rr) * 17 + cc] + red[((ww + 1) * NR + rr) * 17 + cc];
            sg[e] = s;
            if (KC > 1) st_sc1(&Pp[((size_t)blockIdx.y * NR + rr) * D + blockIdx.x * 16 + cc], s);
        }
    }
    if (KC > 1) {
        // ---- the seam: last arriver of this strip finishes it ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's piece stores have left
        __syncthreads();
        if (tid == 0)
            s_ticket = __hip_atomic_fetch_add(&cnt[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket != (unsigned)(KC - 1)) {
            PSTAMP(3);
            return;
        }
        PSTAMP(4);
        if (tid == 0) __hip_atomic_store(&cnt[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double pv[EP][GSMVI_SEAM_MAX_KC];
#pragma unroll
        for (int e = 0; e < EP; ++e) {
            const int idx = tid + 512 * e;
            const int rr = (idx < NR * 16 ? idx : 0) >> 4, cc = idx & 15;
#pragma unroll
            for (int kc = 0; kc < GSMVI_SEAM_MAX_KC; ++kc)
                pv[e][kc] = ld_sc1(&Pp[((size_t)(kc < KC ? kc : KC - 1) * NR + rr) * D + blockIdx.x * 16 + cc]);
        }
#pragma unroll
        for (int e = 0; e < EP; ++e) {
            double t = 0.0;
#pragma unroll
            for (int kc = 0; kc < GSMVI_SEAM_MAX_KC; ++kc) t += (kc < KC) ? pv[e][kc] : 0.0;   // order of k_gsm_scalars_fast
            sg[e] = t;
        }
    }
    // ---- finished strip: SG and the strip's partial dots (gsm_numpy.py:8-9) ----
#pragma unroll
    for (int e = 0; e < EP; ++e) {
        const int idx = tid + 512 * e;
        const bool on = idx < NR * 16;
        const int rr = (on ? idx : 0) >> 4, cc = idx & 15;
        const int col = blockIdx.x * 16 + cc;
        const double g = pg[e], d = pdv[e];
        if (on) SG[(size_t)rr * D + col] = sg[e];
        const double p0 = sum16(g * sg[e]);
        const double p1 = sum16(d * g);
        if (on && cc == 0) *reinterpret_cast<v2d*>(pd + ((size_t)blockIdx.x * NR + rr) * 2) = (v2d){p0, p1};
    }
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PSTAMP(5); }
#undef PSTAMP
}

// =====================================================================================
// Covariance and mean update from X, SG and the partial dots (see the header; tile geometry, LDS layout, MFMA
// chains, mirror stores: k_gsm_cov_sym).  S0 symmetric (only its upper triangle is read), D % 64 == 0.
// =====================================================================================
template <int SB>
__global__ __launch_bounds__(512) void k_gsm_cov_fused(int D, const double* __restrict__ X, int ldx,
                                                       const double* __restrict__ SG,
                                                       const double* __restrict__ pd,
                                                       const double* __restrict__ mu0,
                                                       const double* __restrict__ S0, int lds0,
                                                       double* __restrict__ S, int lds,
                                                       double* __restrict__ mu_out, int flags,
                                                       unsigned long long* __restrict__ stamps) {
#define STAMP(k)                                                                          \
    do {                                                                                  \
        if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    STAMP(0);
    constexpr int RS = 48;
    constexpr int NPASS = (SB > 32) ? SB / 32 : 1;
    constexpr int SBP = SB / NPASS;
    constexpr int TILE = SBP * RS;
    constexpr int NU = SB * 16;                  // 16-B units per (matrix, column block)
    constexpr int NUNITS = 3 * NU;               // units per matrix over the three column blocks I, J0, J1
    constexpr int UPT = (NUNITS + 511) / 512;
    constexpr int PPT = (SB * 16 + 511) / 512;   // (sample, part) pairs of the partial-dot reduction per thread
    __shared__ __attribute__((aligned(16))) double smem[6 * TILE >= 2 * 32 * 33 ? 6 * TILE : 2 * 32 * 33];
    __shared__ __attribute__((aligned(16))) double coef[2 * SB];       // (beta_b, c_b)

    const int nt = D >> 5;
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
        tj0 = ti + 2 * rem;
        two = true;
    } else {
        const int k = blockIdx.x - n_two;
        ti = ((nt & 1) ? 0 : 1) + 2 * k;
        tj0 = nt - 1;
        two = false;
    }
    const bool diag = (tj0 == ti);
    const int I0 = ti * 32, J0 = tj0 * 32;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 15, ks = l >> 4;
    const int t = w >> 2;
    const int wr = (w >> 1) & 1, wc = w & 1;
    constexpr double invB = 1.0 / (double)SB;
    const bool mine = (t == 0) || two;

    // ---- every global load of this workgroup, in one batch: S0 (HBM), partial dots, X / SG / mu0 tiles ----
    // issue order = wait order (vmcnt counts in order): the partial dots first (the scalar prologue runs while the
    // rest is in flight), then the tiles, the S0 tile -- needed last -- at the end
    const size_t srow = (size_t)(I0 + 16 * wr + ks);
    const int scol = J0 + 32 * ((t == 1 && two) ? 1 : 0) + 16 * wc + c;
    const int nstr = D >> 4;
    v2d pq[PPT][4];
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        const int pr = tid + 512 * p;
        const int b = (pr < SB * 16 ? pr : 0) >> 4, part = pr & 15;
        v2d pv[4];                                                // D <= 1024 on this path: at most 4 strips per lane
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s = part + 16 * k;
            pv[k] = *reinterpret_cast<const v2d*>(pd + ((size_t)(s < nstr ? s : part) * SB + b) * 2);
            if (s >= nstr) pv[k] = (v2d){0.0, 0.0};
        }
        pq[p][0] = pv[0]; pq[p][1] = pv[1]; pq[p][2] = pv[2]; pq[p][3] = pv[3];
    }
    v2d gx[UPT], gsg[UPT], gm[UPT];
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
        const int g = q * 512 + tid;
        const int gc = g < NUNITS ? g : 0;
        const int blk = gc / NU, u = gc % NU;
        const int b = u >> 4, c2 = 2 * (u & 15);
        const int colbase = (blk == 0) ? I0 : (J0 + ((blk == 2 && two) ? 32 : 0));
        gx[q] = *reinterpret_cast<const v2d*>(X + (size_t)b * ldx + colbase + c2);
        gsg[q] = *reinterpret_cast<const v2d*>(SG + (size_t)b * D + colbase + c2);
        gm[q] = *reinterpret_cast<const v2d*>(mu0 + colbase + c2);
    }
    __builtin_amdgcn_sched_barrier(0);
    double s0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) s0[r] = S0[(srow + 4 * r) * lds0 + scol];
    __builtin_amdgcn_sched_barrier(0);
    // ---- per-sample scalars (gsm_numpy.py:8-10,15): every workgroup computes all of them ----
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        const int pr = tid + 512 * p;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a0 += pq[p][k].x;
            a1 += pq[p][k].y;
        }
        const double gSg = sum16(a0), mv = sum16(a1);
        if (pr < SB * 16 && (pr & 15) == 0) {
            const double rho = 0.5 * sqrt(1.0 + 4.0 * (gSg + mv * mv)) - 0.5;
            const double den = 1.0 + rho + mv;
            *reinterpret_cast<v2d*>(&coef[2 * (pr >> 4)]) = (v2d){1.0 / (1.0 + rho), (gSg - mv) / den};
        }
    }
    __syncthreads();
    STAMP(1);

    // ---- d- and e-tiles formed on the fly (gsm_numpy.py:11,17,18,22), staged as in k_gsm_cov_sym ----
    constexpr int NS = SBP / 4;
    v4d accd = {0.0, 0.0, 0.0, 0.0}, acce = {0.0, 0.0, 0.0, 0.0};
    v2d dmu_keep[(NU + 511) / 512];              // dmu of this thread's units of column block I (new mean)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass > 0) __syncthreads();
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int g = q * 512 + tid;
            const int gc = g < NUNITS ? g : 0;
            const int blk = gc / NU, u = gc % NU;
            const int b = u >> 4;
            const v2d bc = *reinterpret_cast<const v2d*>(&coef[2 * b]);
            v2d d, e, dm;
            d.x = gm[q].x - gx[q].x;
            d.y = gm[q].y - gx[q].y;
            dm.x = bc.x * ((gsg[q].x - d.x) - bc.y * d.x);
            dm.y = bc.x * ((gsg[q].y - d.y) - bc.y * d.y);
            e.x = d.x + dm.x;
            e.y = d.y + dm.y;
            if (pass == 0 && blk == 0 && q < (NU + 511) / 512) dmu_keep[q] = dm;
            if (g < NUNITS && b / SBP == pass) {
                double* dst = smem + (2 * blk) * TILE + (b % SBP) * RS + 2 * (u & 15);
                *reinterpret_cast<v2d*>(dst) = d;
                *reinterpret_cast<v2d*>(dst + TILE) = e;
            }
        }
        __syncthreads();
        if (pass == 0) STAMP(2);
        double ad[NS], ae[NS], bd[NS], be[NS];
        {
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
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            accd = GSMVI_MFMA_F64(ad[s], bd[s], accd);
            acce = GSMVI_MFMA_F64(ae[s], be[s], acce);
        }
    }
    double wv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wv[r] = s0[r] + (accd[r] - acce[r]) * invB;
    if (mine) {
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(srow + 4 * r) * lds + scol] = wv[r];
    }
    if (stamps) { asm volatile("" :: "v"(wv[0]), "v"(wv[1]), "v"(wv[2]), "v"(wv[3])); STAMP(3); }

    const bool need = mine && !(t == 0 && diag);
    if (flags & 1) {
        // mirror tile stored straight from the accumulator layout: lane (c, ks), register r holds W[16 wr + ks + 4r]
        // [16 wc + c]; the four lanes ks = 0..3 of one c cover 32 contiguous bytes of row J0 + 32 t + 16 wc + c
        if (need) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                S[(size_t)(J0 + 32 * t + 16 * wc + c) * lds + I0 + 16 * wr + ks + 4 * r] = wv[r];
        }
    } else {
        __syncthreads();
        double* LW = smem + t * 32 * 33;
        if (need) {
#pragma unroll
            for (int r = 0; r < 4; ++r) LW[(16 * wr + ks + 4 * r) * 33 + 16 * wc + c] = wv[r];
        }
        __syncthreads();
        if (need) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = LW[(16 * wc + c) * 33 + 16 * wr + ks + 4 * r];
                S[(size_t)(J0 + 32 * t + 16 * wr + ks + 4 * r) * lds + I0 + 16 * wc + c] = v;
            }
        }
    }
    if (diag) {                                  // new mean: mu0 + mean_b dmu_b, samples summed in order
        __syncthreads();
#pragma unroll
        for (int q = 0; q < (NU + 511) / 512; ++q) {
            const int g = q * 512 + tid;
            if (g < NU) *reinterpret_cast<v2d*>(&smem[(g >> 4) * 32 + 2 * (g & 15)]) = dmu_keep[q];
        }
        __syncthreads();
        if (tid < 32) {
            double s = 0.0;
#pragma unroll 8
            for (int b = 0; b < SB; ++b) s += smem[b * 32 + tid];
            mu_out[I0 + tid] = mu0[I0 + tid] + s * invB;
        }
    }
    STAMP(4);
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(5); }
#undef STAMP
}

// ---- launch helpers ------------------------------------------------------------------------
int gsmvi_panel_seam_chunk(int MT) { return MT == 4 ? 128 : 256; }

bool gsmvi_launch_panel_seam(hipStream_t st, hipEvent_t* ev, int D, int B, int KC, int chunks_per_wg, const double* G,
                             int ldg, const double* S0, int lds0, const double* X, int ldx, const double* mu0,
                             double* Pp, double* SG, double* pd, unsigned* cnt, unsigned long long* stamps) {
    if (KC > GSMVI_SEAM_MAX_KC) return false;
    const dim3 grid(D / 16, KC, 1);
#define PS(MTV, CW)                                                                                              \
    GSMVI_LAUNCH((k_panel_seam<MTV, CW>), grid, dim3(512), 0, st, ev, D, G, ldg, S0, lds0, X, ldx, mu0, Pp, SG, pd, \
                 cnt, chunks_per_wg, stamps)
    switch (B) {
        case 16: PS(1, 256); break;
        case 32: PS(2, 256); break;
        case 64: PS(4, 128); break;
        default: return false;
    }
#undef PS
    return true;
}

static int cov_fused_grid(int nt) {
    int n = 0;
    for (int ti = 0; ti < nt; ++ti) n += (nt - ti + 1) / 2;
    return n;
}

bool gsmvi_launch_gsm_cov_fused(hipStream_t st, hipEvent_t* ev, int D, int B, const double* X, int ldx,
                                const double* SG, const double* pd, const double* mu0, const double* S0, int lds0,
                                double* S, int lds, double* mu_out, int flags, unsigned long long* stamps) {
    const dim3 grid(cov_fused_grid(D / 32));
#define CF(SBV)                                                                                                  \
    GSMVI_LAUNCH(k_gsm_cov_fused<SBV>, grid, dim3(512), 0, st, ev, D, X, ldx, SG, pd, mu0, S0, lds0, S, lds, mu_out, \
                 flags, stamps)
    switch (B) {
        case 16: CF(16); break;
        case 32: CF(32); break;
        case 64: CF(64); break;
        default: return false;
    }
#undef CF
    return true;
}
