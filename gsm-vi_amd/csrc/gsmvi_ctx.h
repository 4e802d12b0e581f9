// Context shared by the ABI translation units (opaque to callers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

#define GSMVI_MAX_KC 8
#define GSMVI_FACTOR_NMAX 256        // largest 2B of the factor-form updates (the 2B x 2B chain of gsmvi_factor.hip)
#define GSMVI_STAMP_WORDS (4 * 4096)   // timeline diagnostic: 4 kernels x 512 workgroups x 8 words

// Optional extras of ONE fast panel-product launch (k_panel_fast), consumed -- and cleared -- by the next product that takes
// the fast kernel; `px_used` tells the caller whether that happened (the guarded kernels ignore them: the caller then runs
// its separate finish passes).
struct gsmvi_panel_extras {
    // rows >= msplit of the right operand M come as split-K slabs: M[row][col] = sum_{q < kcm} msl[q * mstride + (row - msplit) * ldsl + col];
    // the finished rows are also written to mfin (row stride ldfin) by the z == 0 workgroups when mfin != nullptr
    const double* msl = nullptr;
    int kcm = 0;
    size_t mstride = 0;
    int ldsl = 0, msplit = 0;
    double* mfin = nullptr;
    int ldfin = 0;
    // side job shared by all workgroups of the launch: sj_dst[i] = sum_{q < sj_kc} sj_src[q * sj_stride + i], i < sj_len
    const double* sj_src = nullptr;
    int sj_kc = 0;
    size_t sj_stride = 0;
    int sj_len = 0;
    double* sj_dst = nullptr;
    // rider (factor path, n = 2B <= 64): ONE extra workgroup of the launch -- the grid gets one more x index; its (y, z) = (0, 0)
    // workgroup runs the 2B x 2B chain gsmf_small16_body (gsmvi_small16.h) beside the product, the others of that column exit
    int rd_on = 0, rd_n = 0, rd_B = 0, rd_kcg = 0, rd_jmode = 0;
    const double* rd_Gp = nullptr;
    double *rd_Kmat = nullptr, *rd_coef = nullptr;
    int* rd_bad = nullptr;
    unsigned long long* rd_stamps = nullptr;
    const int* rd_prior = nullptr;
    const double* rd_Pi = nullptr;          // jmode 2: the B x B coupling matrix of the orthogonal-basis BaM form (gsmvi_small16.h)
    const double* rd_R11 = nullptr;         // jmode 2, B % 16 == 0: the finished first diagonal block [R11 | W11] of the chain's Gram
    const double* rd_W11 = nullptr;         // matrix (B x B each, compact; Gvv's factor from k_bam_small48's side workgroup)
    int w4 = 0;                             // this launch takes the prefetching form k_panel_fast_p (round 6; set by the product drivers: large D, on the grid)
};

// BaM's regulariser as the kernels take it: by value, or -- so that a captured hipGraph can be replayed with another value
// (bam.py:196 evaluates regf(i) on the host every iteration) -- from a device word read at execution time
// (gsmvi_bam_set_reg_source, include/gsmvi_hip.h).
struct bam_reg {
    double v;
    const double* p;
#if defined(__HIPCC__)
    __device__ __forceinline__ double get() const { return p ? *p : v; }
#endif
};

// BaM's mean inside the factor update kernel (k_gsmf_update_fs): mu = mu0 / (1 + reg) + sg_r1 + r1 xbar (bam.py:112) instead of
// the GSM form mu0 + sum_b coef_b Tm_b; xbar == nullptr: GSM.  The factor-form BaM update sets ctx->bam_mean before the back half
// and finds ctx->bam_mean_done == 1 afterwards when the launch that ran honours it (else k_bamf_commit follows).
struct gsmf_bam_mean {
    const double* xbar;
    const double* sg_r1;               // r1 (S gbar): the last row of Rt F0
    bam_reg reg;
};

struct gsmvi_ctx {
    gsmvi_panel_extras px;     // see above
    int px_used = 0;
    unsigned path = 0;         // GSMVI_PATH_* bits of the kernel families launched since the last reset (gsmvi_last_path)
    int tune_panel_w4_min_D = 2048;    // from this D on, plain products with <= 32 rows on the grid keep the next chunk's loads in flight (k_panel_fast_p; 0: never; A/B runs)
    int colwin_0 = 0, colwin_n = 0;    // column-tile window of the factor update launches (colwin_n = 0: all of F)
    const double* potrf_w = nullptr;   // after a k_potrf_dag factorisation: its W blocks (W_k = R_kk^-T, [nblk][64 x 64], in the slab area:
    const double* potrf_r = nullptr;   //   valid until the next panel product), the factor it wrote and its size; null after the
    int potrf_w_n = 0;                 //   launch-per-step form
    int tune_potrf_dag = 1;       // the factorisation as ONE persistent launch with look-ahead (k_potrf_dag, round 6); 0: one launch per block step (A/B)
    int tune_potrf_workers = 0;   // > 0: at most this many worker workgroups beside k_potrf_dag's chain (tests of the ticket order on a small grid)
    int tune_potrf_spin = 0;      // > 0: polls before a waiting workgroup of k_potrf_dag gives up (tests of the abort path)
    int tune_potrf_split_m = 0;   // > 0: tile rows from which a Cholesky block step runs its row solve as a separate launch (A/B)
    int tune_wide = 1;         // 64-row panels (B = 64) of D-sized products on the 64 x 64-tile kernels of gsmvi_wide.hip
    int tune_wide_kc = 0;      // > 0: force their split-K count (A/B runs)
    int tune_gram_mt = 1;      // row-block cap of the Gram product when the chain rides (fewer split-K slabs for its one CU)
    int tune_rider = 1;        // the 2B x 2B chain rides in the V Fm product's launch (0: its own launch, for A/B runs)
    int device = 0;
    int max_D = 0, max_B = 0;
    int num_cu = 256;
    int rmax = 0;              // panel rows the workspace is sized for
    void* ws = nullptr;        // one hipMalloc
    size_t ws_bytes = 0;
    double* pp = nullptr;      // panel partials [GSMVI_MAX_KC][rmax][max_D]
    double* sg = nullptr;      // [8][rmax][max_D] finished panels (SG, BaM factor panels)
    double *fo_Rt = nullptr, *fo_Tm = nullptr, *fo_Fs = nullptr;   // set only inside the factor-form BaM update: where ITS [Vw; Zw], Tm, Fs panels live
    double* small = nullptr;   // coefficients and small dense matrices
    int* ints = nullptr;       // device ints (flags)
    double* gram_slabs = nullptr;   // [GSMVI_MAX_KC][rmax][rmax] split-K slabs of the factor path's Gram product, + 16 stamp words
    int tune_panel_kc = 0;
    int tune_update_sb = 0;
    int tune_cov_dbg = 0;      // ablation bits for k_gsm_cov_sym (wrong results; timing only)
    int tune_timeline = 0;     // 1 = every fast-path kernel of the dense update writes s_memrealtime stamps (diagnostic)
    unsigned long long* stamps = nullptr;   // [kernel][workgroup][8], allocated by the "timeline" knob
    int tune_direct_out = 1;   // finished outputs of sample / score / U F / Gram products from the product launch when the split-K
                               // count is 1 (no hand-off involved); 0 = always product + k_panel_finish (A/B tests)
    int tune_scalars_nt = 0;   // threads per sample in k_gsm_scalars_fast (256/512/1024; 0 = default)
    int tune_no_fast = 0;      // 1 = force the guarded generic kernels (tests)
    int* bam_hint_host = nullptr;       // pinned word: k* of the last device BaM chain (step-count hint, never synchronised on)
    int tune_bam_kenq = 0;     // > 0: enqueue exactly this many multi-workgroup steps (tests of the tail kernel)
    int tune_bam_hint_slack = 1;   // Newton-Schulz steps enqueued beyond the previous call's k* (round 4: 1; before: 2; 0 measured in round 5)
    int tune_rider_direct_max_D = 2048;   // 1024 <= D <= this: a panel product that carries the chain as its rider runs UNSPLIT (kc = 1,
                                          // finished output): it is hidden behind the ~35 us chain either way (0: never)
    int tune_bam_full = 0;     // 1 = always enqueue every Newton-Schulz step (ignore the hint; tests)
    int tune_lowrank_kp = 0;   // 64: BaM's low-rank update stages 64 rows per pass for KF > 96 (A/B runs: measured equal to 32)
    int tune_chain_pair = 1;   // two-level chain (128 < 2B <= 256): independent one-workgroup factorisations share a launch (0: A/B runs)
    gsmf_bam_mean bam_mean = {nullptr, nullptr, {0.0, nullptr}};
    int bam_mean_done = 0;
    const double* chain_r11 = nullptr;   // [R11 | W11] of Gvv for the one-workgroup 2B x 2B chain (gsmvi_small16.h, jmode 2)
    const double* chain_w11 = nullptr;
    const double* reg_dev = nullptr;   // gsmvi_bam_set_reg_source: BaM's regulariser is read from here at execution time
    int tune_bam_basis = 1;    // 1 (default) = factor-form BaM in the basis [Vw; Zt], Zt = the part of Zw orthogonal to the whitened draws; 0 = [Vw; Zw];
                               // 3 = as 1, but the 2B x 2B chain factors its first diagonal block itself (A/B of the given-block form)
                               // (round 5: no dependent rows at the fixed point of a Gaussian target); 0 = the
                               // round-4 basis [Vw; Zw] (A/B runs)
    double* basis = nullptr;   // workspace of that form: five (R/2)^2 slots -- T, M1', M1 - M1', Pi, X
    const double* chain_pi = nullptr;   // set for ONE 2B x 2B chain (jmode 2): the B x B block Pi of the dense signature J' ...
    double* chain_x = nullptr;          // ... and where X = R11 Pi^T goes (the multi-launch chains)
    double* early = nullptr;   // [Gamma11 | R11 | W11], 128 x 128 each: the first diagonal block of the factor-form BaM chain's Gram
                               // matrix (Vw Vw^T, known before the B x B chain) and its factors, produced beside k_bam_cholw
    int early_ready = 0;       // set by gsmvi_bam_factor_impl when that job was launched; consumed by factor_chain_big
    int profiling = 0;         // when set, the update kernels are launched with dispatch-timestamp events
    hipStream_t side = nullptr;         // second stream of the factor path at large D (V Fm beside the 2B x 2B chain), with its
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // fork / join events (no timing)
    int tune_fork_min_D = 3072;         // smallest D that forks (the two event edges cost ~10 us; 0 = never)
    hipEvent_t ev[8] = {};     // [2*stage], [2*stage+1]: panel, scalars, cov-update, spare
    int ev_valid[4] = {};

    unsigned long long* timeline_stamps(int kernel) {
        return (tune_timeline && stamps) ? stamps + (size_t)kernel * 4096 : nullptr;
    }

    hipEvent_t* stage_events(int stage) {
        if (!profiling) return nullptr;
        ev_valid[stage] = 1;
        return &ev[2 * stage];
    }
};

void gsmvi_panel_wide_split(int D, int tiles, int num_cu, int kc_force, int* kc_out, int* kper_out);
void gsmvi_launch_panel_wide(hipStream_t st, hipEvent_t* ev, bool transposed, int D, int nrows, const double* A, int lda,
                             const double* shift, double alpha, const double* M, int ldm, double* Pp, int kper, int kc,
                             int ncols);
int gsmvi_panel_product_nc(gsmvi_ctx* ctx, hipStream_t st, hipEvent_t* ev, int D, int ncols, int nrows,
                           const double* A, int lda, const double* shift, double alpha, const double* M, int ldm,
                           double* Pp, int* kc_out);
int gsmvi_panel_product_out(gsmvi_ctx* ctx, hipStream_t st, int D, int ncols, int nrows, const double* A, int lda,
                            const double* shift, double alpha, const double* M, int ldm, const double* addvec, double* Out,
                            int ldo);
int gsmvi_panel_finish(hipStream_t st, int ncols, int nrows, int kc, const double* Pp, const double* addvec,
                       double* Out, int ldo);
int gsmvi_panel_finish_cols(hipStream_t st, int ncols_in, int ncols_out, int nrows, int kc, const double* Pp,
                            double* Out, int ldo);
void gsmvi_set_error(const char* fmt, const char* a, const char* b);
int gsmvi_panel_product(gsmvi_ctx* ctx, hipStream_t st, hipEvent_t* ev, int D, int nrows, const double* A, int lda,
                        const double* shift, double alpha, const double* M, int ldm, double* Pp, int* kc_out);

// the side workgroup of k_bam_small48 (gsmvi_bam_small.hip; orthogonal basis of the factor-form BaM update, n <= 48)
struct bamq_side {
    double* M1p;                       // n x n: M1' = -Gvv^-1 M1 (what the substitution multiplies Vw with)
    double* Dm;                        // n x n: M1 - M1'
    double* t2;                        // n: L^-T zg (written by the CHAIN workgroup: wave 0, behind zg)
    int* info1;                        // 0 or the 1-based failing pivot of Gvv's factorisation (dependent draws)
    double* R11;                       // n x n each (compact), or null: Gvv's factor and its inverse transpose for the 2B x 2B chain
    double* W11;
};
