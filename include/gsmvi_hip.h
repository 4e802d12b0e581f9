/*
 * gsmvi_hip.h -- C ABI of the MI355X (gfx950) GSM / BaM update engine.
 *
 * The reference (modichirag/GSM-VI) has no FFI layer: its boundary for this path is the set of
 * pure Python callables listed below.  Each entry point here is what a binding for that callable
 * would bind; the Python mirror in gsm-vi_amd/ calls them through ctypes.
 *
 *   reference callable (file:line, relative to the reference tree)     ->  entry point
 *   gsmvi/gsm_numpy.py:27-55  gsm_update(samples, vs, mu0, S0)          ->  gsmvi_gsm_update_f64
 *   gsmvi/gsm.py:31-58        gsm_update (JAX twin)                     ->  gsmvi_gsm_update_f64
 *   examples/example_gsm_numpy.py:24-29  lp_g of the Gaussian target    ->  gsmvi_gaussian_score_f64
 *   gsmvi/gsm_numpy.py:116    np.random.multivariate_normal(mean,cov,B) ->  gsmvi_sample_f64 (+ gsmvi_potrf_f64)
 *   gsmvi/gsm_numpy.py:105,116 np.random.seed + standard-normal stream  ->  gsmvi_randn_f64 (counter-based)
 *   gsmvi/gsm_numpy.py:132-146 _check_goodness(cov)                     ->  gsmvi_potrf_f64 (info flag)
 *   (no reference twin; gsm_numpy.py:4-55 in factor form)   ->  gsmvi_gsm_factor_update_f64
 *   gsmvi/bam.py:72-114       bam_lowrank_update(samples,vs,mu0,S0,reg) ->  gsmvi_bam_update_f64
 *   gsmvi/bam.py:31-69        bam_update(samples,vs,mu0,S0,reg)         ->  gsmvi_bam_update_f64 (same result, K6)
 *
 * Conventions
 *   - All matrices are row-major float64 in DEVICE memory; `ld*` are leading dimensions in elements.
 *   - Calls are asynchronous on `stream` (a hipStream_t passed as void*); nothing synchronises,
 *     so call sequences can be captured into a hipGraph.
 *   - Inputs are never modified; outputs must not alias inputs (reference updates are pure,
 *     gsm_numpy.py:47-55) unless an entry point says otherwise.
 *   - S0 must be symmetric (it is a covariance); the kernels read it once, by rows.
 *   - Every function returns a gsmvi_status; no C++ exception crosses the ABI.
 *   - One context per stream; calls on one context are not thread-safe (the reference is
 *     single-threaded: gsm_numpy.py:105 uses the global numpy RNG).
 */
#ifndef GSMVI_HIP_H
#define GSMVI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is compiled with -fvisibility=hidden and linked with an export list (csrc/exports.map): exactly the
 * functions declared between this push and its pop are visible (tests/test_abi.py compares `nm -D` with this file). */
#pragma GCC visibility push(default)

#define GSMVI_ABI_VERSION 1

typedef enum gsmvi_status {
    GSMVI_OK = 0,
    GSMVI_ERR_BAD_ARG = 1,    /* NULL pointer, non-positive size, ld < D, aliasing, ...            */
    GSMVI_ERR_HIP = 2,        /* a HIP runtime call failed; see gsmvi_last_error()                 */
    GSMVI_ERR_NO_DEVICE = 3,  /* no gfx950 device visible                                          */
    GSMVI_ERR_WORKSPACE = 4,  /* (D,B) exceeds what the context was created for                    */
    GSMVI_ERR_UNSUPPORTED = 5
} gsmvi_status;

typedef struct gsmvi_ctx gsmvi_ctx;   /* opaque: device id, workspace, launch heuristics */

int gsmvi_abi_version(void);
const char* gsmvi_status_string(int status);
/* Message of the last failing call on this host thread ("" if none). */
const char* gsmvi_last_error(void);
/* Number of HIP devices; GSMVI_ERR_NO_DEVICE (and *n = 0) when there is none. */
int gsmvi_device_count(int* n);

/* Workspace bytes a context allocates for problems up to (max_D, max_B). */
size_t gsmvi_workspace_bytes(int max_D, int max_B);
/* Creates a context on `device` with workspace for D <= max_D, B <= max_B. */
/* gsmvi_create / gsmvi_destroy leave the caller's current HIP device as they found it.  The compute entry points
 * never change the current device either: `stream` must belong to the context's device and that device must be
 * current when they are called (what torch.cuda.set_device / hipSetDevice in the calling loop guarantees). */
int gsmvi_create(gsmvi_ctx** out, int device, int max_D, int max_B);
int gsmvi_destroy(gsmvi_ctx* ctx);
/* Launch-heuristic knobs for tests and A/B measurements: "panel_kc" (split-K count of the panel products; <= 0 = auto),
 * "no_fast" (1 = force the guarded generic kernels), "direct_out" (0 = always product + finish pass), "update_sb",
 * "scalars_nt", "bam_full", "bam_kenq", "rider", "wide", "wide_kc", "gram_mt", "fork_min_D", "potrf_split_m", "chain_pair"
 * (0 = one launch per one-workgroup factorisation of the 128 < 2B <= 256 chain); round 5: "bam_basis" (1 = factor-form BaM in
 * the orthogonal basis [Vw; Zt], default; 0 = the round-4 basis [Vw; Zw]; 3 = as 1 but the 2B x 2B chain factors its first
 * diagonal block itself), "bam_hint_slack" (Newton-Schulz steps enqueued beyond the previous call's count, default 1),
 * "rider_direct_max_D" (largest D at which the panel product carrying the chain as its rider runs unsplit, default 2048),
 * "lowrank_kp" (64 = 64-row staging passes of BaM's low-rank update); round 6: "potrf_dag" (0 = one launch per block step),
 * "potrf_spin" (poll budget of a wait inside k_potrf_dag), "potrf_workers" (cap on its worker workgroups: tests of the ticket
 * order), "panel_w4_min_D"; diagnostics "timeline", "cov_dbg"
 * (see gsmvi_hip_debug.h). */
int gsmvi_set_tuning(gsmvi_ctx* ctx, const char* name, int value);

/*
 * GSM batch update (dense-covariance path).  Replaces gsmvi/gsm_numpy.py:27-55.
 *   X  (B x D, ldx)  samples            G  (B x D, ldg)  scores lp_g(X)
 *   mu0 (D)          current mean       S0 (D x D, lds0) current covariance (symmetric)
 *   mu  (D)          new mean           S  (D x D, lds)  new covariance
 * mu = mu0 + mean_b dmu_b ; S = S0 + mean_b (d_b d_b^T - e_b e_b^T), d_b = mu0 - x_b, e_b = d_b + dmu_b.
 * Three kernels: panel product SG = G S0 (fp64 MFMA), per-sample scalars, rank-2B update (fp64 MFMA).
 * PRECONDITION: S0 is symmetric (a covariance).  For D % 32 == 0 and B in {16, 32, 64} the update kernel reads only
 * the UPPER triangle of S0 and mirrors the result, so S comes out exactly symmetric; for other shapes the generic
 * kernel reads all of S0.  A non-symmetric S0 therefore gives shape-dependent results that differ from
 * gsm_numpy.py:50-53 (S0 + mean): use gsmvi_gsm_update_general_f64 for such an S0 (the Python drop-in gsm_update() does
 * so for host inputs it finds non-symmetric, and on request for device inputs).
 * One context per stream: calls on one gsmvi_ctx share its workspace and must not run concurrently.
 */
int gsmvi_gsm_update_f64(gsmvi_ctx* ctx, void* stream, int D, int B,
                         const double* X, int ldx, const double* G, int ldg,
                         const double* mu0, const double* S0, int lds0,
                         double* mu, double* S, int lds);

/*
 * The same update for an S0 that is not symmetric: the reference's literal semantics S = S0 + mean_b (...) with S0 g_b in
 * the per-sample stage (gsm_numpy.py:7,50-53) for ANY square S0.  Reads all of S0 (transposed panel product + the guarded
 * update kernel); the Python drop-in gsm_update(..., assume_symmetric=False) calls it.  Not a performance path.
 */
int gsmvi_gsm_update_general_f64(gsmvi_ctx* ctx, void* stream, int D, int B,
                                 const double* X, int ldx, const double* G, int ldg,
                                 const double* mu0, const double* S0, int lds0,
                                 double* mu, double* S, int lds);

/*
 * The same update in two stages, for the batch-sharded multi-GPU path (one process per GPU):
 *   local stage : for this rank's B_local samples, panel product + per-sample scalars; writes one
 *                 record per sample  rec[b] = [ d_b (D) | e_b (D) | dmu_b (D) ]  (d_b = mu0 - x_b,
 *                 dmu_b = mu_update of gsm_numpy.py:17, e_b = d_b + dmu_b) with row stride
 *                 ldrec >= gsmvi_gsm_record_len(D) = 3D rounded up to even.  Records of all ranks are
 *                 all-gathered (RCCL) by the caller;
 *   apply       : every replica applies the combined rank-2B update from all B records.
 * gsmvi_gsm_update_f64 == local stage with B_local = B followed by apply.
 */
int gsmvi_gsm_record_len(int D);
int gsmvi_gsm_local_stage_f64(gsmvi_ctx* ctx, void* stream, int D, int B_local,
                              const double* X, int ldx, const double* G, int ldg,
                              const double* mu0, const double* S0, int lds0,
                              double* rec, int ldrec);
int gsmvi_gsm_apply_f64(gsmvi_ctx* ctx, void* stream, int D, int B,
                        const double* rec, int ldrec, const double* mu0,
                        const double* S0, int lds0, double* mu, double* S, int lds);

/*
 * Batch-sharded update over RCCL in ONE call (one process per GPU; SURVEY 8(b): "RCCL-sharded variants taking an
 * ncclComm_t"): local stage on this rank's B_local samples -> ncclAllGather of the records (B_local *
 * gsmvi_gsm_record_len(D) doubles per rank, in place in rec_all) on `stream` -> combined rank-2B update applied by
 * every replica; replicas end bit-identical.  nccl_comm is an ncclComm_t (passed as void* so that this header does
 * not need rccl.h) created by the caller with the RCCL library loaded in its process; this library resolves
 * ncclAllGather from that instance at first use and does not link RCCL itself.  rec_all: caller-owned device buffer
 * of (B_local * nranks) x gsmvi_gsm_record_len(D) doubles.  B_local * nranks must fit the context's max_B.
 * Python callers use torch.distributed instead (gsm-vi_amd/dist.py::sharded_gsm_update), same two stage calls.
 */
int gsmvi_gsm_update_sharded_f64(gsmvi_ctx* ctx, void* stream, void* nccl_comm, int D, int B_local,
                                 const double* X_local, int ldx, const double* G_local, int ldg,
                                 const double* mu0, const double* S0, int lds0, double* rec_all,
                                 double* mu, double* S, int lds);

/*
 * The RCCL library the sharded entry points call into.  By default they resolve ncclAllGather / ncclCommCount /
 * ncclCommUserRank at first use from the RCCL instance already loaded in the process (falling back to librccl.so.1).  A
 * process that holds several RCCL copies (e.g. torch's bundled one beside the system one) passes the dlopen handle of the
 * copy that CREATED its communicators here, before the first sharded call; afterwards the choice is fixed.
 */
int gsmvi_set_rccl_library(void* dl_handle);

/*
 * The same update with the covariance sharded by ROW BLOCKS (the decomposition that divides
 * the HBM-bound passes by the number of GPUs).  A rank owns rows [row0, row0 + nrows) of S0 as an
 * nrows x D row-major block; X, G, mu0 are replicated.
 *   rows stage : SGcols (B x nrows, ldsg) = G S0rows^T, i.e. columns [row0, row0+nrows) of G S0 (S0 symmetric,
 *                gsm_numpy.py:7).  The caller all-gathers the column slices into SG (B x D, contiguous);
 *   records    : per-sample scalars of gsm_numpy.py:8-17 from the gathered SG -> records as above (replicated);
 *   apply rows : Srows = S0rows + (1/B) sum_b (d_b d_b^T - e_b e_b^T)[row0 : row0+nrows, :], and the full new
 *                mean when mu != NULL.
 */
int gsmvi_gsm_rows_stage_f64(gsmvi_ctx* ctx, void* stream, int D, int B, int nrows,
                             const double* G, int ldg, const double* S0rows, int lds0,
                             double* SGcols, int ldsg);
int gsmvi_gsm_records_f64(gsmvi_ctx* ctx, void* stream, int D, int B,
                          const double* X, int ldx, const double* G, int ldg, const double* mu0,
                          const double* SG, double* rec, int ldrec);
int gsmvi_gsm_apply_rows_f64(gsmvi_ctx* ctx, void* stream, int D, int B, int row0, int nrows,
                             const double* rec, int ldrec, const double* mu0,
                             const double* S0rows, int lds0, double* mu, double* Srows, int lds);

/*
 * Factor-form GSM update (BASELINE config 5): the state is a square factor Fm with
 * Sigma = Fm^T Fm; Z (B x D) are the whitened draws behind the samples X = 1 mu0^T + Z F0, G = lp_g(X).
 * Produces (mu, F) with F^T F equal to the covariance gsm_numpy.gsm_update would return for
 * (X, G, mu0, F0^T F0) -- without forming or factorising any D x D covariance: the positive-definite
 * test of gsm_numpy.py:121-125,132-146 becomes a Cholesky of a 2B x 2B matrix.  If that test fails,
 * (mu, F) = (mu0, F0) and *info_dev = 1 (revert); else *info_dev = 0.  Needs 2B <= D and 2B <= 256 (2B <= 64: the chain is one workgroup; <= 128: one-workgroup
 * factorisations; <= 256: two-level blocked, round 4).
 * n_reverts_dev (device int, may be NULL) is incremented on a revert, like gsmvi_commit_f64 does.
 */
int gsmvi_gsm_factor_update_f64(gsmvi_ctx* ctx, void* stream, int D, int B,
                                const double* Z, int ldz, const double* X, int ldx, const double* G, int ldg,
                                const double* mu0, const double* F0, int ldf0,
                                double* mu, double* F, int ldf, int* info_dev, int* n_reverts_dev);

/*
 * The factor-form update in two stages, for the batch-sharded multi-GPU path (BASELINE config 5 on several GPUs;
 * same decomposition as gsmvi_gsm_local_stage_f64 / gsmvi_gsm_apply_f64):
 *   local stage : for this rank's B_local samples (rows of Z, X, G): W = G F0^T, the whitened residual v_b = w_b + z_b
 *                 and v_b F0; one record per sample  rec[b] = [ x_b - mu0 (D) | v_b (D) | v_b F0 (D) ], row stride
 *                 ldrec >= gsmvi_gsm_record_len(D).  Two of the three passes over F0 are divided by the number of
 *                 ranks.  Records of all ranks are all-gathered (RCCL) by the caller;
 *   apply       : every replica holds the same Z (B x D: the draw stream is replicated, gsmvi_randn_f64 is
 *                 counter-based) and all B records, and runs the Gram product of [Z; V], the per-sample scalars (they are
 *                 entries of that Gram matrix), the 2B x 2B positive-definite test and the rank-2B factor update.  Same outputs and revert semantics as gsmvi_gsm_factor_update_f64.
 * gsmvi_gsm_factor_update_f64 == local stage with B_local = B followed by apply.
 */
int gsmvi_gsm_factor_local_stage_f64(gsmvi_ctx* ctx, void* stream, int D, int B_local,
                                     const double* Z, int ldz, const double* X, int ldx, const double* G, int ldg,
                                     const double* mu0, const double* F0, int ldf0, double* rec, int ldrec);
int gsmvi_gsm_factor_apply_f64(gsmvi_ctx* ctx, void* stream, int D, int B,
                               const double* Z, int ldz, const double* rec, int ldrec,
                               const double* mu0, const double* F0, int ldf0,
                               double* mu, double* F, int ldf, int* info_dev, int* n_reverts_dev);

/*
 * The factor-form update batch-sharded over RCCL in ONE call: local stage on this rank's rows -> ncclAllGather of the
 * records (in place in rec_all: (B_local * nranks) x gsmvi_gsm_record_len(D) doubles) -> combined update on every replica.
 * Z_all holds the replicated draws of ALL B = B_local * nranks samples (the draw stream is counter-based; rank r's samples
 * are rows [r B_local, (r + 1) B_local)); X_local, G_local are this rank's rows.  Same outputs and revert semantics as
 * gsmvi_gsm_factor_update_f64; everything is validated before the first launch.
 */
int gsmvi_gsm_factor_update_sharded_f64(gsmvi_ctx* ctx, void* stream, void* nccl_comm, int D, int B_local,
                                        const double* Z_all, int ldz, const double* X_local, int ldx,
                                        const double* G_local, int ldg, const double* mu0, const double* F0, int ldf0,
                                        double* rec_all, double* mu, double* F, int ldf, int* info_dev,
                                        int* n_reverts_dev);

/*
 * Profiling mode (used by bench.py for the roofline line): when on, the three kernels of the GSM
 * update are launched with dispatch-timestamp events; gsmvi_get_profile waits for the last call
 * and returns the kernel durations in milliseconds: ms[0] panel product, ms[1] per-sample
 * scalars, ms[2] covariance update (-1 where a stage did not run).
 */
int gsmvi_set_profiling(gsmvi_ctx* ctx, int on);
int gsmvi_get_profile(gsmvi_ctx* ctx, float* ms, int n);

/*
 * Which kernel families the calls on this context launched since the last reset (round 5).  The tuned kernels need
 * D % 64 == 0, even leading dimensions and 16-byte aligned bases (any batch size); everything else runs the guarded
 * kernels of the same arithmetic (csrc/gsmvi_kernels.hip), about half as fast.  The reference takes any (D, B)
 * (gsm_numpy.py:27-55, bam.py:31-114); a caller that keeps its state padded to a multiple of 64 columns (zero border in
 * mu, X, G, Z, F; identity border on the diagonal of Sigma -- INTEGRATION.md, "Off-grid dimensions") stays on the tuned
 * kernels for every D, and this query is how tests and profiles prove that it did: *bits & GSMVI_PATH_GENERIC_MASK == 0.
 * reset != 0 clears the record after reading it.
 */
#define GSMVI_PATH_PANEL_FAST 0x0001u      /* k_panel_fast: A M products (S0 G, score, sampler, V Fm, ...)          */
#define GSMVI_PATH_PANEL_WIDE 0x0002u      /* k_panel_wide: the same on 64 x 64 tiles (64-row panels, D >= 1024)    */
#define GSMVI_PATH_PANEL_GENERIC 0x0004u   /* k_panel_partial                                                       */
#define GSMVI_PATH_PANEL_T_FAST 0x0008u    /* k_panel_t_fast: A M^T products (W = G F^T, Gram matrices)              */
#define GSMVI_PATH_PANEL_T_GENERIC 0x0010u /* k_panel_t                                                             */
#define GSMVI_PATH_SCALARS_FAST 0x0020u    /* k_gsm_scalars_fast                                                    */
#define GSMVI_PATH_SCALARS_GENERIC 0x0040u /* k_gsm_scalars                                                         */
#define GSMVI_PATH_COV_SYM 0x0080u         /* k_gsm_cov_sym / k_gsm_cov_sym_p: the headline covariance kernel       */
#define GSMVI_PATH_COV_GENERIC 0x0100u     /* k_gsm_cov_update (also: non-symmetric S0, row-block shards)           */
#define GSMVI_PATH_FUPD_FAST 0x0200u       /* k_gsmf_update_fs / k_gsmf_update_fast: F = F0 + Rt^T Fs               */
#define GSMVI_PATH_FUPD_GENERIC 0x0400u    /* k_gsmf_update + k_gsmf_mean                                           */
#define GSMVI_PATH_LOWRANK_FAST 0x0800u    /* k_lowrank_update_fast: BaM's S = S0 + Vf^T Vf - Z^T Z                 */
#define GSMVI_PATH_LOWRANK_GENERIC 0x1000u /* k_lowrank_update                                                      */
#define GSMVI_PATH_GENERIC_MASK (0x0004u | 0x0010u | 0x0040u | 0x0100u | 0x0400u | 0x1000u)
int gsmvi_last_path(gsmvi_ctx* ctx, unsigned* bits, int reset);

/*
 * Where the BaM entry points of this context take the regulariser from (round 5).  The reference evaluates regf(i) on the
 * host in every iteration (bam.py:196) and passes a number; a caller that captures an iteration into a hipGraph needs a
 * value that can change between replays.  reg_dev != NULL: every BaM update launched on this context from now on (dense,
 * factor form, and the sharded forms built on them) reads *reg_dev on the DEVICE when its kernels execute and ignores its `reg`
 * argument (which is still validated: pass any positive number; the value in the word is the caller's to check, reg > 0); the word
 * must stay valid while such work is pending.  NULL (the default) restores the by-value argument.
 */
int gsmvi_bam_set_reg_source(gsmvi_ctx* ctx, const double* reg_dev);

/*
 * Score of the Gaussian target N(m, P^-1) at the rows of X: G = -(X - 1 m^T) P.
 * Replaces the user callback of examples/example_gsm_numpy.py:24-29 (P symmetric precision matrix).
 */
int gsmvi_gaussian_score_f64(gsmvi_ctx* ctx, void* stream, int D, int B,
                             const double* X, int ldx, const double* m,
                             const double* P, int ldp, double* G, int ldg);

/*
 * Upper Cholesky factor R (R^T R = S, R upper triangular, strictly-lower part zeroed) of a
 * symmetric matrix; *info_dev (device int) = 0 if S is positive definite, else 1 + index of the
 * first failing pivot (also set when a NaN is met).  Replaces np.linalg.cholesky inside
 * _check_goodness (gsm_numpy.py:132-146) and supplies the sampling factor.  R must not alias S; only the upper
 * triangle of S is read (plus the full diagonal blocks).  Two launches (a flag-clearing one and the persistent task graph
 * k_potrf_dag, round 6; ceil(D/64) launches before, still behind the knob "potrf_dag" = 0); every wait inside it is bounded:
 * *info_dev = D + 1 reports a wait that ran out of its poll budget (a shared, stalled GPU), never a hang.
 */
int gsmvi_potrf_f64(gsmvi_ctx* ctx, void* stream, int D, const double* S, int lds,
                    double* R, int ldr, int* info_dev);

/*
 * C = F^T F for a square factor F (D x D, any square factor with F^T F = cov): the covariance a factor-form fit
 * returns and hands to its monitor (the reference's fit returns cov, gsm_numpy.py:129; monitors.py:99).  C is
 * exactly symmetric.  Once per fit / per monitor checkpoint, not per iteration.
 */
int gsmvi_gram_f64(gsmvi_ctx* ctx, void* stream, int D, const double* F, int ldf, double* C, int ldc);

/*
 * C = F^T F + (shift + *shift_dev) I (shift_dev may be NULL; a device double read when the kernel executes).  The reference's
 * BaM loop adds jitter * I to the covariance after EVERY update (bam.py:198, default 1e-6); a diagonal shift is not a low-rank
 * change of a square factor, so a factor-form fit carries the shift it owes and absorbs it every few iterations by
 * re-factorising this matrix with gsmvi_potrf_f64 (gsm-vi_amd/bam.py, jitter_every).  shift_dev lets the caller count only
 * ACCEPTED updates without a host synchronisation (a reverted iteration adds no jitter in the reference, bam.py:208-212).
 */
int gsmvi_gram_shift_f64(gsmvi_ctx* ctx, void* stream, int D, const double* F, int ldf, double shift, const double* shift_dev,
                         double* C, int ldc);

/*
 * Whitened residuals Z = (X - 1 mu^T) R^-1 for nrows rows of X and an upper Cholesky factor R (R^T R = cov), and
 * (if logdiag_dev != NULL) logdiag_dev[0] = sum_i log R_ii.  Together they give the row-wise Gaussian log density
 * log N(x; mu, cov) = -1/2 |z|^2 - sum_i log R_ii - D/2 log(2 pi) that the reference's KLMonitor evaluates through
 * numpyro's MultivariateNormal.log_prob (gsmvi/monitors.py:107-113).  mu may be NULL (zero mean).  Monitor use
 * only (D dependent steps per row).  D <= 8192.
 */
int gsmvi_whiten_rows_f64(gsmvi_ctx* ctx, void* stream, int D, int nrows, const double* R, int ldr,
                          const double* X, int ldx, const double* mu, double* Z, int ldz, double* logdiag_dev);

/*
 * Draw samples X = 1 mu^T + Z R for whitened draws Z (B x D) and an upper factor R (R^T R = cov).
 * Replaces np.random.multivariate_normal(mean, cov, size=B) (gsm_numpy.py:116); Z is supplied by
 * the caller (host MT19937 stream in parity mode, device Philox in throughput mode).
 */
int gsmvi_sample_f64(gsmvi_ctx* ctx, void* stream, int D, int B,
                     const double* Z, int ldz, const double* mu, const double* R, int ldr,
                     double* X, int ldx);

/*
 * COLUMN-SHARDED factor-form GSM update (round 6; SURVEY 8(e) row 3 / (f) 3: the decomposition that divides the HBM-bound
 * D^2 passes and the D^2 memory of a fit by the number of ranks).  A rank owns the columns C = [col0, col0 + ncols) of the
 * square factor Fm (Sigma = Fm^T Fm) as a D x ncols block with its own leading dimension, and the entries C of the mean.
 *   gsmvi_sample_cols_f64          Xcols (B x ncols) = mu_cols + Z Fcols: the owned slice of x = mu + z Fm (gsm_numpy.py:116);
 *                                  the caller all-gathers the slices (B ncols doubles per rank).
 *   gsmvi_gsm_rows_stage_f64       on the block (D := ncols, nrows := D, G + col0, Fcols) gives the PARTIAL product
 *                                  G[:, C] Fm[:, C]^T (B x D); the caller all-reduces the partials to W = G Fm^T.
 *   gsmvi_gsm_factor_apply_cols_f64  from the replicated draws Z, the all-reduced W (B x D, contiguous) and the gathered
 *                                  samples X: the 2B x 2B chain of gsmvi_gsm_factor_update_f64 (replicated: identical inputs
 *                                  and arithmetic on every rank, so the accept / revert decision agrees) and the update of the
 *                                  OWNED block alone, Fcols' = Fcols + Rt^T (K'' Tm[:, C]), mu[C].  mu0 / mu are full-length
 *                                  vectors of which entries C are read / written.  col0 % 64 == 0, ncols % 64 == 0 (the last
 *                                  block may be ragged), even D and leading dimensions, 16-byte aligned blocks.
 * Per update a rank reads its block three times and writes it once (32 D ncols bytes) and exchanges B (ncols + D) doubles.
 */
int gsmvi_sample_cols_f64(gsmvi_ctx* ctx, void* stream, int D, int B, int ncols, const double* Z, int ldz,
                          const double* mu_cols, const double* Fcols, int ldf, double* Xcols, int ldx);
int gsmvi_gsm_factor_apply_cols_f64(gsmvi_ctx* ctx, void* stream, int D, int B, int col0, int ncols, const double* Z, int ldz,
                                    const double* W, const double* X, int ldx, const double* mu0, const double* F0cols,
                                    int ldf0, double* mu, double* Fcols, int ldf, int* info_dev, int* n_reverts_dev);

/*
 * Whitened draws: out[0..n) ~ N(0, 1), a pure function of (seed, call, element index) -- counter-based
 * Philox4x32-10 (key = seed, counter = (pair index, call)) + Box-Muller in fp64; see csrc/gsmvi_rng.hip.
 * Replaces the standard-normal stream behind np.random.multivariate_normal (gsm_numpy.py:105,116) in
 * throughput mode; `call` is the fit iteration (the JAX twins likewise derive a fresh sub-key per iteration,
 * gsm.py:117-119).  Stateless, so sharded ranks draw identical Z from the same key.  raw (device uint32,
 * 4 words per element pair, may be NULL) receives the Philox words (tests pin them to Random123 vectors).
 */
int gsmvi_randn_f64(gsmvi_ctx* ctx, void* stream, uint64_t seed, uint64_t call, int64_t n, double* out,
                    uint32_t* raw);

/*
 * The draws of SEVERAL consecutive calls from one launch: out[c * n + i] = element i of draw number call0 + c, c < ncalls --
 * bit-identical to ncalls calls of gsmvi_randn_f64 (a fit loop draws a block of iterations ahead: the stream does not depend
 * on the state).  call_in_dev (device uint64, may be NULL) is added to call0 on the device; call_out_dev (may be NULL, must
 * not alias call_in_dev) receives *call_in_dev + ncalls.  With the two words of a ping-pong pair a launch captured into a
 * hipGraph advances through the stream on every replay.
 */
int gsmvi_randn_batch_f64(gsmvi_ctx* ctx, void* stream, uint64_t seed, uint64_t call0, int ncalls, int64_t n, double* out,
                          const uint64_t* call_in_dev, uint64_t* call_out_dev);

/*
 * Commit-or-revert (gsm_numpy.py:121-125): if *info_dev == 0 copy (mu_new, S_new) over (mu, S),
 * else leave them; *n_reverts_dev is incremented on a revert.  Device-side, no host sync.
 */
int gsmvi_commit_f64(gsmvi_ctx* ctx, void* stream, int D, const int* info_dev,
                     const double* mu_new, const double* S_new, int lds_new,
                     double* mu, double* S, int lds, int* n_reverts_dev);

/*
 * BaM update (gsmvi/bam.py:72-114 with an exact rank-B factor of U; equals bam.py:31-69).
 * Symmetrised output (bam.py:199 does this in fit); jitter is added to the diagonal (bam.py:198).
 * The B x B matrix function of bam.py:108-110 (B + 1 columns in the reference's factorisation of U; an orthonormal
 * recombination of the centred score rows saves one without changing U) -- which the reference evaluates on the host through
 * jax.pure_callback (bam.py:15-22) -- runs on the device (scaled coupled Newton-Schulz square root on the MFMA pipe +
 * a one-workgroup Cholesky for B <= 129, the blocked Cholesky of gsmvi_potrf_f64 up to B = 1024 (round 6; 640 before); csrc/gsmvi_bam_small.hip):
 * no synchronisation, graph-capturable.  For B > 1024 the call returns GSMVI_ERR_UNSUPPORTED before anything is enqueued:
 * there is no host computation in this library.
 * *info_dev = 1 if that small problem was not finite / not positive definite (then mu, S are NaN-poisoned and the
 * caller's accept/revert must reject them).
 */
int gsmvi_bam_update_f64(gsmvi_ctx* ctx, void* stream, int D, int B,
                         const double* X, int ldx, const double* G, int ldg,
                         const double* mu0, const double* S0, int lds0, double reg, double jitter,
                         double* mu, double* S, int lds, int* info_dev);

/*
 * BaM update batch-sharded over RCCL (BASELINE config 4: B = 128 as 16 per GPU): BaM's statistics couple all samples, so the
 * ranks all-gather their (x_b, g_b) rows (two ncclAllGather of B_local x D doubles per rank) into xg_all (caller-owned,
 * 2 x B x D doubles: [X_all | G_all]) and every replica runs gsmvi_bam_update_f64 on the full batch; what is divided is the
 * score evaluation in front of it.  Validated before the first launch.
 */
int gsmvi_bam_update_sharded_f64(gsmvi_ctx* ctx, void* stream, void* nccl_comm, int D, int B_local,
                                 const double* X_local, int ldx, const double* G_local, int ldg,
                                 const double* mu0, const double* S0, int lds0, double reg, double jitter,
                                 double* xg_all, double* mu, double* S, int lds, int* info_dev);

/*
 * BaM update in FACTOR form (north_star's factor-form extension applied to gsmvi/bam.py:72-114): Sigma0 = F0^T F0 in,
 * Sigma = F^T F out, with F^T F equal to the S of gsmvi_bam_update_f64 (jitter = 0) to round-off and the same mean.  No D x D
 * covariance is formed and no D x D factorisation is taken: four passes over F0 and a 2B x 2B chain (the one of
 * gsmvi_gsm_factor_update_f64).  Z (B x D) are the whitened draws of the samples: X = mu0 + Z F0 (the caller's contract, as
 * for the GSM factor update).  Needs 2B <= min(D, 256) (GSMVI_ERR_UNSUPPORTED otherwise, before anything is enqueued).
 * *info_dev != 0 and (mu, F) = (mu0, F0) if BaM's B x B matrix function or the 2B x 2B chain failed (non-finite input, or a
 * Sigma that is not positive definite to working precision); *n_reverts_dev (may be NULL) is then incremented.  Since round 5
 * the update works in the basis [Vw; Zt] (Zt: the part of Zw orthogonal to the whitened draws), which takes the Cholesky factor
 * of Gvv = Vw Vw^T: LINEARLY DEPENDENT draws (a repeated sample; impossible for i.i.d. normal draws, legal in bam.py) make Gvv
 * singular -- the update then either still equals the dense one or is reverted with *info_dev != 0; ALMOST dependent draws
 * ((max R_ii / min R_ii)^2 > 1e8 on the factor's diagonal) are reverted too, because the basis is orthogonal only to
 * eps cond(Gvv).  That ratio is a HEURISTIC LOWER BOUND of cond(Gvv), not the condition number (a graded matrix of Kahan's
 * kind has a far larger one): it catches what it was built for -- one draw nearly repeating another, 5e-6 off without a flag at
 * cond 1e12 before the guard (tests/test_gpu_bam.py) -- and is no guarantee for adversarial draws; method="dense" has no
 * such precondition.
 */
int gsmvi_bam_factor_update_f64(gsmvi_ctx* ctx, void* stream, int D, int B,
                                const double* Z, int ldz, const double* X, int ldx, const double* G, int ldg,
                                const double* mu0, const double* F0, int ldf0, double reg,
                                double* mu, double* F, int ldf, int* info_dev, int* n_reverts_dev);

/*
 * Factor-form BaM update batch-sharded over RCCL (BASELINE config 4, "B = 128 sharded 16/GPU", without a D x D covariance or a
 * D^3 step on any rank; round 4).  Z_all (B x D, B = B_local x ranks) are the whitened draws of ALL samples, replicated (every
 * rank draws the same counter-based stream, gsmvi_randn_f64); X_local / G_local are the samples and scores of this rank's
 * rows [rank B_local, (rank + 1) B_local).  As in gsmvi_bam_update_sharded_f64 the (x_b, g_b) rows are all-gathered into
 * xg_all (caller-owned, 2 x B x D doubles) and every replica runs gsmvi_bam_factor_update_f64 on the full batch: replicas stay
 * bit-identical.  Geometry and the 2B <= min(D, 256) bound are validated before the first launch.
 */
int gsmvi_bam_factor_update_sharded_f64(gsmvi_ctx* ctx, void* stream, void* nccl_comm, int D, int B_local,
                                        const double* Z_all, int ldz, const double* X_local, int ldx,
                                        const double* G_local, int ldg, const double* mu0, const double* F0, int ldf0,
                                        double reg, double* xg_all, double* mu, double* F, int ldf, int* info_dev,
                                        int* n_reverts_dev);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* GSMVI_HIP_H */
