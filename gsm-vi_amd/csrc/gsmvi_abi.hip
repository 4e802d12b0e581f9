// C ABI of the gfx950 GSM/BaM engine (declared in include/gsmvi_hip.h).
// Host side only: argument validation, workspace, launch geometry.  No torch types, no
// exceptions across the boundary, no host synchronisation inside the update entry points.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/gsmvi_hip.h"
#include "../../include/gsmvi_hip_debug.h"

// ---- kernels (gsmvi_kernels.hip / gsmvi_potrf.hip / gsmvi_bam.hip) -------------------------
void gsmvi_launch_panel_partial(hipStream_t st, hipEvent_t* ev, int MT, dim3 grid, int D, int ncols, int nrows,
                                const double* A, int lda, const double* shift, double alpha, const double* M,
                                int ldm, double* Pp, int chunks_per_wg, int a_vec_ok);
void gsmvi_launch_panel_finish(hipStream_t st, hipEvent_t* ev, int D, int nrows, int KC, const double* Pp,
                               const double* addvec, double* Out, int ldo, int ncols_out);
void gsmvi_launch_gsm_scalars(hipStream_t st, hipEvent_t* ev, int D, int B, int KC, const double* X, int ldx,
                              const double* G, int ldg, const double* mu0, const double* Pp, double* rec,
                              int ldrec);
size_t gsmvi_cov_update_lds_bytes(int SB);
hipError_t gsmvi_cov_update_prepare();
hipError_t gsmvi_bam_prepare();
void gsmvi_launch_gsm_cov_update(hipStream_t st, hipEvent_t* ev, int D, int B, const double* rec, int ldrec,
                                 const double* mu0, const double* S0, int lds0, double* S, int lds, double* mu_out,
                                 int SB, int s_vec_ok, int row0, int nrows);
int gsmvi_panel_t_product(struct gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* A, int lda,
                          const double* M, int ldm, int mrows, double* Pp, int* kc_out);
void gsmvi_launch_commit(hipStream_t st, int D, const int* info, const double* mu_new, const double* S_new,
                         int lds_new, double* mu, double* S, int lds, int* n_reverts);
// fast paths (gsmvi_fast.hip)
void gsmvi_launch_panel_fast(hipStream_t st, hipEvent_t* ev, int MT, dim3 grid, int D, int nrows, const double* A,
                             int lda, const double* shift, double alpha, const double* M, int ldm, double* Pp,
                             int chunks_per_wg, int ncols, unsigned long long* stamps, double* Out, int ldo,
                             const double* addvec, const struct gsmvi_panel_extras* px);
bool gsmvi_launch_gsm_scalars_fast(hipStream_t st, hipEvent_t* ev, int D, int B, int KC, const double* X, int ldx,
                                   const double* G, int ldg, const double* mu0, const double* Pp, double* rec,
                                   int ldrec, int nt, unsigned long long* stamps);
bool gsmvi_launch_gsm_cov_sym(hipStream_t st, hipEvent_t* ev, int D, int B, const double* rec, int ldrec,
                              const double* mu0, const double* S0, int lds0, double* S, int lds, double* mu_out,
                              int dbg, unsigned long long* stamps, int num_cu);
int gsmvi_panel_fast_chunk(int MT);
int gsmvi_potrf_impl(struct gsmvi_ctx* ctx, hipStream_t st, int D, const double* S, int lds, double* R, int ldr,
                     int* info_dev);
int gsmvi_gram_impl(hipStream_t st, int D, const double* F, int ldf, double* C, int ldc, double shift, const double* shift_dev);
int gsmvi_whiten_impl(hipStream_t st, int D, int nrows, const double* R, int ldr, const double* X, int ldx,
                      const double* mu, double* Z, int ldz, double* logdiag);
int gsmvi_factor_impl(struct gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* Z, int ldz, const double* X,
                      int ldx, const double* G, int ldg, const double* mu0, const double* F0, int ldf0, double* mu,
                      double* F, int ldf, int* info_dev, int* n_reverts_dev);
int gsmvi_factor_local_impl(struct gsmvi_ctx* ctx, hipStream_t st, int D, int Bl, const double* Z, int ldz,
                            const double* X, int ldx, const double* G, int ldg, const double* mu0, const double* F0,
                            int ldf0, double* rec, int ldrec);
int gsmvi_factor_apply_impl(struct gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* Z, int ldz,
                            const double* rec, int ldrec, const double* mu0, const double* F0, int ldf0, double* mu,
                            double* F, int ldf, int* info_dev, int* n_reverts_dev);
int gsmvi_factor_apply_cols_impl(struct gsmvi_ctx* ctx, hipStream_t st, int D, int B, int col0, int ncols, const double* Z,
                                 int ldz, const double* W, int ldw, const double* X, int ldx, const double* mu0,
                                 const double* F0c, int ldf0, double* mu, double* Fc, int ldf, int* info_dev,
                                 int* n_reverts_dev);
int gsmvi_bam_impl(struct gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* X, int ldx,
                   const double* G, int ldg, const double* mu0, const double* S0, int lds0, double reg,
                   double jitter, double* mu, double* S, int lds, int* info_dev);
int gsmvi_bam_factor_impl(gsmvi_ctx* ctx, hipStream_t st, int D, int B, const double* Z, int ldz, const double* X, int ldx,
                          const double* G, int ldg, const double* mu0, const double* F0, int ldf0, double reg, double* mu,
                          double* F, int ldf, int* info_dev, int* n_reverts_dev);

#include "gsmvi_ctx.h"

static thread_local std::string g_last_error;

void gsmvi_set_error(const char* fmt, const char* a, const char* b) {
    char buf[512];
    snprintf(buf, sizeof buf, fmt, a ? a : "", b ? b : "");
    g_last_error = buf;
}

#define HIP_TRY(expr)                                                     \
    do {                                                                  \
        hipError_t e_ = (expr);                                           \
        if (e_ != hipSuccess) {                                           \
            gsmvi_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); \
            return GSMVI_ERR_HIP;                                         \
        }                                                                 \
    } while (0)

#define BAD_ARG(cond, msg)                          \
    do {                                            \
        if (cond) {                                 \
            gsmvi_set_error("%s: %s", __func__, msg); \
            return GSMVI_ERR_BAD_ARG;               \
        }                                           \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gsmvi_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}

typedef double v2d_abi __attribute__((ext_vector_type(2)));
// calibration copy kernel of gsmvi_debug_stream_copy_f64 (below)
__global__ __launch_bounds__(256) void k_stream_copy(const v2d_abi* __restrict__ src, v2d_abi* __restrict__ dst, size_t n2) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n2; i += stride) {
        v2d_abi a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = (i + 256 * u < n2) ? __builtin_nontemporal_load(src + i + 256 * u) : (v2d_abi){0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + 256 * u < n2) __builtin_nontemporal_store(a[u], dst + i + 256 * u);
    }
}

extern "C" {

int gsmvi_abi_version(void) { return GSMVI_ABI_VERSION; }

const char* gsmvi_status_string(int s) {
    switch (s) {
        case GSMVI_OK: return "ok";
        case GSMVI_ERR_BAD_ARG: return "bad argument";
        case GSMVI_ERR_HIP: return "HIP runtime error";
        case GSMVI_ERR_NO_DEVICE: return "no HIP device";
        case GSMVI_ERR_WORKSPACE: return "problem larger than the context's workspace";
        case GSMVI_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown status";
    }
}

const char* gsmvi_last_error(void) { return g_last_error.c_str(); }

int gsmvi_device_count(int* n) {
    if (!n) return GSMVI_ERR_BAD_ARG;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0) {
        (void)hipGetLastError();
        *n = 0;
        gsmvi_set_error("%s%s", "no HIP device visible: ", e != hipSuccess ? hipGetErrorString(e) : "count 0");
        return GSMVI_ERR_NO_DEVICE;
    }
    *n = c;
    return GSMVI_OK;
}

// Workspace layout (doubles): panel partials [KCmax][Rmax][D], SG [Rmax][D], coef [8][Rmax],
// small-matrix scratch for BaM / potrf, plus a few ints.
static void ws_sizes(int D, int B, size_t* n_pp, size_t* n_sg, size_t* n_small, int* rmax) {
    const int R = 2 * B + 8;                                   // BaM uses up to B+1 panel rows twice
    *rmax = R;
    *n_pp = (size_t)GSMVI_MAX_KC * R * D;
    // potrf parks its factored 64x64 diagonal blocks here: one per block step of max(D, 2B+8)
    const size_t dpad = (size_t)(((D > R ? D : R) + 63) / 64) * 64;
    size_t potrf_scratch = dpad * 64;                          // v1: one factored 64 x 64 block per step
    if (potrf_scratch < 3 * 64 * dpad + 2 * 64 * 64) potrf_scratch = 3 * 64 * dpad + 2 * 64 * 64;   // row buffers + W + X^T blocks
    {   // k_potrf_dag (round 6): one W block per step (64 dpad doubles) + its flags (2 + nblk + 2 nblk^2 ints)
        const size_t nb_ = dpad / 64, dag = 64 * dpad + (2 + nb_ + 2 * nb_ * nb_ + 1) / 2 + 8;
        if (potrf_scratch < dag) potrf_scratch = dag;
    }
    if (*n_pp < potrf_scratch) *n_pp = potrf_scratch;
    // the TRANSPOSED panel products (A M^T: BaM's stacked Gram matrices [P; Vf] Qt^T, the factor path's Rt Rt^T) leave kc slabs of
    // up to R x R doubles, kc <= min(GSMVI_MAX_KC, D / 64): for B >> D that is more than the R x D slabs above (round 5: D = 64,
    // B = 640 wrote 0.82 M doubles into a 0.66 M slab area and returned a wrong update without a flag -- found by a probe, now a test)
    {
        size_t kct = (size_t)(D + 63) / 64;
        if (kct > GSMVI_MAX_KC) kct = GSMVI_MAX_KC;
        if (kct < 1) kct = 1;
        if (*n_pp < kct * (size_t)R * R) *n_pp = kct * (size_t)R * R;
    }
    *n_sg = (size_t)R * D * 8;                                 // SG + BaM factor panels (the factor-form BaM update holds 10 B + 8 rows)
    // + the device chain of BaM's small matrix function: five padded 144 x 144 iterates, coefficients, BB (n <= 129)
    // + the factor path: a seventh R x R slot (finished Gram matrix) and the split-K slabs of the Gram product (+ 16 stamp words)
    // BaM's Newton-Schulz iterates: five ld x ld matrices, ld = 144 for B + 1 <= 129, else B + 1 rounded up to 16 (R/2 + 16 covers it)
    const size_t ldb = (R / 2 + 16 > 144) ? (size_t)(R / 2 + 16) : 144;
    // + three 128 x 128 slots in front of the Gram slabs: the early first block of the factor-form BaM chain (ctx->early)
    // + 2 R^2 doubles behind everything (five (R/2)^2 slots): the orthogonal-basis form of the factor-form BaM update (ctx->basis; round 5)
    *n_small = (size_t)8 * R + (size_t)7 * R * R + 4096 + (size_t)5 * ldb * ldb + 64 + ldb * ldb + 64 + 3 * 128 * 128 +
               (size_t)GSMVI_MAX_KC * R * R + 16 + (size_t)2 * R * R;
}

size_t gsmvi_workspace_bytes(int max_D, int max_B) {
    if (max_D <= 0 || max_B <= 0) return 0;
    size_t a, b, c;
    int r;
    ws_sizes(max_D, max_B, &a, &b, &c, &r);
    return (a + b + c) * sizeof(double) + 256 + 4096;
}

int gsmvi_create(gsmvi_ctx** out, int device, int max_D, int max_B) {
    BAD_ARG(!out, "out is NULL");
    BAD_ARG(max_D <= 0 || max_B <= 0, "max_D and max_B must be positive");
    *out = nullptr;
    int n = 0;
    int st = gsmvi_device_count(&n);
    if (st != GSMVI_OK) return st;
    BAD_ARG(device < 0 || device >= n, "device index out of range");
    int prev_device = 0;                                   // the caller's current device is restored on return
    HIP_TRY(hipGetDevice(&prev_device));
    struct device_guard {
        int prev;
        ~device_guard() { (void)hipSetDevice(prev); }
    } guard{prev_device};
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    gsmvi_ctx* c = new gsmvi_ctx();
    c->device = device;
    c->max_D = max_D;
    c->max_B = max_B;
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    size_t n_pp, n_sg, n_small;
    ws_sizes(max_D, max_B, &n_pp, &n_sg, &n_small, &c->rmax);
    c->ws_bytes = (n_pp + n_sg + n_small) * sizeof(double) + 256 + 4096;
    hipError_t e = hipMalloc(&c->ws, c->ws_bytes);
    if (e != hipSuccess) {
        gsmvi_set_error("hipMalloc of the workspace failed: %s%s", hipGetErrorString(e), "");
        delete c;
        return GSMVI_ERR_HIP;
    }
    c->pp = reinterpret_cast<double*>(c->ws);
    c->sg = c->pp + n_pp;
    c->small = c->sg + n_sg;
    c->ints = reinterpret_cast<int*>(c->small + n_small);
    c->basis = c->small + n_small - (size_t)2 * c->rmax * c->rmax;
    c->gram_slabs = c->basis - ((size_t)GSMVI_MAX_KC * c->rmax * c->rmax + 16);
    c->early = c->gram_slabs - 3 * 128 * 128;
    e = hipMemset(c->ws, 0, c->ws_bytes);
    if (e == hipSuccess) e = gsmvi_cov_update_prepare();
    if (e == hipSuccess) e = gsmvi_bam_prepare();
    for (int k = 0; k < 8 && e == hipSuccess; ++k) e = hipEventCreate(&c->ev[k]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) {
        gsmvi_set_error("context initialisation failed: %s%s", hipGetErrorString(e), "");
        (void)hipFree(c->ws);
        delete c;
        return GSMVI_ERR_HIP;
    }
    *out = c;
    return GSMVI_OK;
}

int gsmvi_destroy(gsmvi_ctx* ctx) {
    if (!ctx) return GSMVI_OK;
    int prev_device = 0;
    (void)hipGetDevice(&prev_device);
    struct device_guard {
        int prev;
        ~device_guard() { (void)hipSetDevice(prev); }
    } guard{prev_device};
    (void)hipSetDevice(ctx->device);
    for (int k = 0; k < 8; ++k)
        if (ctx->ev[k]) (void)hipEventDestroy(ctx->ev[k]);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->bam_hint_host) (void)hipHostFree(ctx->bam_hint_host);
    if (ctx->stamps) (void)hipFree(ctx->stamps);
    hipError_t e = hipFree(ctx->ws);
    delete ctx;
    if (e != hipSuccess) {
        gsmvi_set_error("hipFree failed: %s%s", hipGetErrorString(e), "");
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}

int gsmvi_set_tuning(gsmvi_ctx* ctx, const char* name, int value) {
    BAD_ARG(!ctx || !name, "NULL argument");
    if (!strcmp(name, "panel_kc")) ctx->tune_panel_kc = value;
    else if (!strcmp(name, "update_sb")) ctx->tune_update_sb = value;
    else if (!strcmp(name, "no_fast")) ctx->tune_no_fast = value;
    else if (!strcmp(name, "direct_out")) ctx->tune_direct_out = value;
    else if (!strcmp(name, "rider")) ctx->tune_rider = value;
    else if (!strcmp(name, "potrf_split_m")) ctx->tune_potrf_split_m = value;
    else if (!strcmp(name, "potrf_dag")) ctx->tune_potrf_dag = value;
    else if (!strcmp(name, "panel_w4_min_D")) ctx->tune_panel_w4_min_D = value;
    else if (!strcmp(name, "potrf_spin")) ctx->tune_potrf_spin = value;
    else if (!strcmp(name, "potrf_workers")) ctx->tune_potrf_workers = value;
    else if (!strcmp(name, "wide")) ctx->tune_wide = value;
    else if (!strcmp(name, "wide_kc")) ctx->tune_wide_kc = value;
    else if (!strcmp(name, "fork_min_D")) ctx->tune_fork_min_D = value;
    else if (!strcmp(name, "gram_mt")) ctx->tune_gram_mt = value > 0 ? value : 4;
    else if (!strcmp(name, "bam_full")) ctx->tune_bam_full = value;
    else if (!strcmp(name, "bam_kenq")) ctx->tune_bam_kenq = value;
    else if (!strcmp(name, "bam_hint_slack")) ctx->tune_bam_hint_slack = value;
    else if (!strcmp(name, "rider_direct_max_D")) ctx->tune_rider_direct_max_D = value;
    else if (!strcmp(name, "chain_pair")) ctx->tune_chain_pair = value;
    else if (!strcmp(name, "lowrank_kp")) ctx->tune_lowrank_kp = value;
    else if (!strcmp(name, "bam_basis")) ctx->tune_bam_basis = value;
    else if (!strcmp(name, "scalars_nt")) ctx->tune_scalars_nt = value;
    else if (!strcmp(name, "cov_dbg")) ctx->tune_cov_dbg = value;   // ablation bits, timing experiments only
    else if (!strcmp(name, "timeline")) {                           // whole-update timeline stamps (diagnostic)
        if (value && !ctx->stamps) {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->stamps), GSMVI_STAMP_WORDS * sizeof(unsigned long long)));
            HIP_TRY(hipMemset(ctx->stamps, 0, GSMVI_STAMP_WORDS * sizeof(unsigned long long)));
        }
        ctx->tune_timeline = value;
    }
    else {
        gsmvi_set_error("%s: unknown tuning knob %s", __func__, name);
        return GSMVI_ERR_BAD_ARG;
    }
    return GSMVI_OK;
}

/* Calibration (bench.py, SURVEY 8(d) "attainable peak"): a plain streaming copy of n doubles, 16 bytes per lane, four
 * independent loads in flight per thread, grid-stride over 2048 workgroups -- this library's own yardstick for what a kernel
 * that only moves bytes reaches on the box (torch's copy_ kernel, used until round 3, was slower than the guide's float4
 * copy).  dst and src 16-byte aligned, n even.  Exported by the debug library only. */
int gsmvi_debug_stream_copy_f64(void* stream, double* dst, const double* src, size_t n) {
    BAD_ARG(!dst || !src || n < 2 || (n & 1) || (reinterpret_cast<uintptr_t>(dst) & 15u) || (reinterpret_cast<uintptr_t>(src) & 15u),
            "bad argument (16-byte aligned pointers, even n)");
    hipLaunchKernelGGL(k_stream_copy, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const v2d_abi*>(src), reinterpret_cast<v2d_abi*>(dst), n / 2);
    return hipGetLastError() == hipSuccess ? GSMVI_OK : GSMVI_ERR_HIP;
}

/* Diagnostic: copies n 64-bit words from the start of the panel-partial slab (where the cov_dbg=16
 * build of k_gsm_cov_sym writes its timeline stamps) to host memory.  Not part of the product path. */
int gsmvi_debug_read_stamps(gsmvi_ctx* ctx, unsigned long long* out, int n) {
    BAD_ARG(!ctx || !out || n < 1, "bad argument");
    HIP_TRY(hipDeviceSynchronize());
    if (ctx->tune_timeline && ctx->stamps) {
        BAD_ARG(n > GSMVI_STAMP_WORDS, "more words than the timeline buffer holds");
        HIP_TRY(hipMemcpy(out, ctx->stamps, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost));
        return GSMVI_OK;
    }
    HIP_TRY(hipMemcpy(out, ctx->pp, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost));
    return GSMVI_OK;
}

/* Diagnostic: copy n doubles from a workspace region (0 = panel slabs, 1 = finished panels, 2 = small
 * matrices) starting at element `offset` to host memory.  Tests and debugging only. */
int gsmvi_debug_read_workspace(gsmvi_ctx* ctx, int region, size_t offset, double* out, size_t n) {
    BAD_ARG(!ctx || !out || region < 0 || region > 2, "bad argument");
    const double* base = region == 0 ? ctx->pp : (region == 1 ? ctx->sg : ctx->small);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, base + offset, sizeof(double) * n, hipMemcpyDeviceToHost));
    return GSMVI_OK;
}

// device address of a workspace region (diagnostic scripts view intermediates in place)
int gsmvi_debug_workspace_ptr(gsmvi_ctx* ctx, int region, double** out) {
    BAD_ARG(!ctx || !out || region < 0 || region > 2, "bad argument");
    *out = region == 0 ? ctx->pp : (region == 1 ? ctx->sg : ctx->small);
    return GSMVI_OK;
}

/* Which kernel families ran (GSMVI_PATH_* bits, include/gsmvi_hip.h) on this context since the last reset. */
int gsmvi_last_path(gsmvi_ctx* ctx, unsigned* bits, int reset) {
    BAD_ARG(!ctx || !bits, "NULL argument");
    *bits = ctx->path;
    if (reset) ctx->path = 0;
    return GSMVI_OK;
}

/* BaM's regulariser from a device word read at execution time (graph replay), or by value again (NULL). */
int gsmvi_bam_set_reg_source(gsmvi_ctx* ctx, const double* reg_dev) {
    BAD_ARG(!ctx, "ctx is NULL");
    ctx->reg_dev = reg_dev;
    return GSMVI_OK;
}

int gsmvi_set_profiling(gsmvi_ctx* ctx, int on) {
    BAD_ARG(!ctx, "ctx is NULL");
    ctx->profiling = on ? 1 : 0;
    for (int k = 0; k < 4; ++k) ctx->ev_valid[k] = 0;
    return GSMVI_OK;
}

int gsmvi_get_profile(gsmvi_ctx* ctx, float* ms, int n) {
    BAD_ARG(!ctx || !ms || n < 1, "bad argument");
    for (int k = 0; k < n; ++k) {
        ms[k] = -1.0f;
        if (k < 4 && ctx->ev_valid[k]) {
            HIP_TRY(hipEventSynchronize(ctx->ev[2 * k + 1]));
            HIP_TRY(hipEventElapsedTime(&ms[k], ctx->ev[2 * k], ctx->ev[2 * k + 1]));
        }
    }
    return GSMVI_OK;
}

}  // extern "C"

// ---- panel product driver: partial slabs Pp[kc][nrows][ncols] of alpha (A - shift) M ------------
// A: nrows x D, M: D x ncols (row-major, ldm).  Returns the number of slabs through *kc_out.
int gsmvi_panel_product_nc(gsmvi_ctx* ctx, hipStream_t st, hipEvent_t* ev, int D, int ncols, int nrows,
                           const double* A, int lda, const double* shift, double alpha, const double* M, int ldm,
                           double* Pp, int* kc_out);

// Product with a FINISHED output Out (nrows x ncols, ldo) = addvec + alpha (A - shift) M.  On the fast path with a split-K
// count of 1 the product launch writes Out itself; otherwise product + k_panel_finish.  Same numbers either way.
int gsmvi_panel_product_out(gsmvi_ctx* ctx, hipStream_t st, int D, int ncols, int nrows, const double* A, int lda,
                            const double* shift, double alpha, const double* M, int ldm, const double* addvec, double* Out,
                            int ldo) {
    const int strips = (ncols + 15) / 16;
    const int MT = nrows <= 16 ? 1 : (nrows <= 32 ? 2 : 4);
    const int zblocks = (nrows + 16 * MT - 1) / (16 * MT);
    const int a_vec_ok = (lda % 2 == 0) && aligned16(A);
    const bool fast = !ctx->tune_no_fast && ctx->tune_direct_out && D % 2 == 0 && a_vec_ok &&
                      (!shift || aligned16(shift)) && !(ctx->px.msl && D % 16 != 0);
    if (fast) {
        const int chw = gsmvi_panel_fast_chunk(MT);
        const int nchunks = (D + chw - 1) / chw;
        // the split the plain product would take (two workgroups per CU wanted); the one-launch form is used only when
        // that split is kc == 1 -- never at the price of fewer workgroups (D = 4096: kc = 2, 512 workgroups; forcing
        // kc = 1 there cost 20 % of the product's speed)
        int kc = ctx->tune_panel_kc > 0 ? ctx->tune_panel_kc
                                        : (2 * ctx->num_cu + strips * zblocks - 1) / (strips * zblocks);
        if (kc > nchunks) kc = nchunks;
        if (kc > GSMVI_MAX_KC) kc = GSMVI_MAX_KC;
        if (kc >= 1) {
            const int cpw = (nchunks + kc - 1) / kc;
            kc = (nchunks + cpw - 1) / cpw;
            // only the hand-off-free case (kc == 1): the product writes the finished output itself
            if (kc == 1 && strips * zblocks <= 1024) {
                const gsmvi_panel_extras px = ctx->px;
                ctx->px = gsmvi_panel_extras();
                ctx->px_used = 1;
                gsmvi_launch_panel_fast(st, nullptr, MT, dim3(strips, kc, zblocks), D, nrows, A, lda, shift, alpha, M, ldm,
                                        ctx->pp, cpw, ncols, ctx->timeline_stamps(0), Out, ldo, addvec, &px);
                ctx->path |= GSMVI_PATH_PANEL_FAST;
                return check_launch("k_panel_fast(out)");
            }
        }
    }
    int kc = 1;
    int rc = gsmvi_panel_product_nc(ctx, st, nullptr, D, ncols, nrows, A, lda, shift, alpha, M, ldm, ctx->pp, &kc);
    if (rc != GSMVI_OK) return rc;
    gsmvi_launch_panel_finish(st, nullptr, ncols, nrows, kc, ctx->pp, addvec, Out, ldo, ncols);
    return check_launch("k_panel_finish");
}

int gsmvi_panel_product_nc(gsmvi_ctx* ctx, hipStream_t st, hipEvent_t* ev, int D, int ncols, int nrows,
                           const double* A, int lda, const double* shift, double alpha, const double* M, int ldm,
                           double* Pp, int* kc_out) {
    const int strips = (ncols + 15) / 16;
    const int MT = nrows <= 16 ? 1 : (nrows <= 32 ? 2 : 4);
    const int zblocks = (nrows + 16 * MT - 1) / (16 * MT);
    const int a_vec_ok = (lda % 2 == 0) && aligned16(A);
    // any even inner dimension D and any ncols since round 5 (the kernel clamps); the slab-operand extra needs D % 16 == 0
    const bool fast = !ctx->tune_no_fast && D % 2 == 0 && a_vec_ok && (!shift || aligned16(shift)) && !(ctx->px.msl && D % 16 != 0);
    // 64-row panels of a D-sized product are MFMA-bound: the 64 x 64-tile kernel (gsmvi_wide.hip).  Not with extras: those
    // launches (K'' Tm with slab-summed rows, side jobs, the rider) stay on the narrow kernel.
    if (fast && MT == 4 && ctx->tune_wide && ncols % 64 == 0 && ncols >= 1024 && D >= 1024 && D % 64 == 0 && ldm % 2 == 0 && aligned16(M) &&
        !ctx->px.msl && !ctx->px.sj_src && !ctx->px.rd_on) {
        int kcw = 1, kper = D;
        gsmvi_panel_wide_split(D, (ncols / 64) * zblocks, ctx->num_cu, ctx->tune_wide_kc, &kcw, &kper);
        *kc_out = kcw;
        ctx->px = gsmvi_panel_extras();
        ctx->px_used = 1;
        gsmvi_launch_panel_wide(st, ev, false, D, nrows, A, lda, shift, alpha, M, ldm, Pp, kper, kcw, ncols);
        ctx->path |= GSMVI_PATH_PANEL_WIDE;
        return check_launch("k_panel_wide");
    }
    const int chw = fast ? gsmvi_panel_fast_chunk(MT) : 256;       // rows of M per chunk
    const int nchunks = (D + chw - 1) / chw;
    int kc = ctx->tune_panel_kc > 0 ? ctx->tune_panel_kc
                                    : (2 * ctx->num_cu + strips * zblocks - 1) / (strips * zblocks);
    if (kc > nchunks) kc = nchunks;
    if (kc > GSMVI_MAX_KC) kc = GSMVI_MAX_KC;
    if (kc < 1) kc = 1;
    const int cpw = (nchunks + kc - 1) / kc;
    kc = (nchunks + cpw - 1) / cpw;
    *kc_out = kc;
    gsmvi_panel_extras px = ctx->px;
    ctx->px = gsmvi_panel_extras();                // extras are for ONE launch; px_used reports whether the fast kernel took them
    ctx->px_used = fast ? 1 : 0;
    // round 6: plain products on the grid at large D walk several chunks per workgroup: the prefetching form (k_panel_fast_p)
    if (fast && MT <= 2 && cpw >= 2 && ctx->tune_panel_w4_min_D > 0 && D >= ctx->tune_panel_w4_min_D && D % 256 == 0 &&
        ncols % 16 == 0 && !px.msl && !px.sj_src && !px.rd_on)
        px.w4 = 1;
    if (fast) {
        gsmvi_launch_panel_fast(st, ev, MT, dim3(strips, kc, zblocks), D, nrows, A, lda, shift, alpha, M, ldm, Pp,
                                cpw, ncols, ctx->timeline_stamps(0), nullptr, 0, nullptr, &px);
        ctx->path |= GSMVI_PATH_PANEL_FAST;
        return check_launch("k_panel_fast");
    }
    ctx->path |= GSMVI_PATH_PANEL_GENERIC;
    gsmvi_launch_panel_partial(st, ev, MT, dim3(strips, kc, zblocks), D, ncols, nrows, A, lda, shift, alpha, M,
                               ldm, Pp, cpw, a_vec_ok);
    return check_launch("k_panel_partial");
}

int gsmvi_panel_product(gsmvi_ctx* ctx, hipStream_t st, hipEvent_t* ev, int D, int nrows, const double* A, int lda,
                        const double* shift, double alpha, const double* M, int ldm, double* Pp, int* kc_out) {
    return gsmvi_panel_product_nc(ctx, st, ev, D, D, nrows, A, lda, shift, alpha, M, ldm, Pp, kc_out);
}

// Out (nrows x ncols, ldo) = addvec + sum of the kc slabs
int gsmvi_panel_finish(hipStream_t st, int ncols, int nrows, int kc, const double* Pp, const double* addvec,
                       double* Out, int ldo) {
    gsmvi_launch_panel_finish(st, nullptr, ncols, nrows, kc, Pp, addvec, Out, ldo, ncols);
    return check_launch("k_panel_finish");
}

// slabs with ncols_in columns (padded), only the first ncols_out are summed into Out
int gsmvi_panel_finish_cols(hipStream_t st, int ncols_in, int ncols_out, int nrows, int kc, const double* Pp,
                            double* Out, int ldo) {
    gsmvi_launch_panel_finish(st, nullptr, ncols_in, nrows, kc, Pp, nullptr, Out, ldo, ncols_out);
    return check_launch("k_panel_finish");
}

static int check_common(gsmvi_ctx* ctx, int D, int B, const char* fn) {
    if (!ctx) {
        gsmvi_set_error("%s: %s", fn, "ctx is NULL");
        return GSMVI_ERR_BAD_ARG;
    }
    if (D <= 0 || B <= 0) {
        gsmvi_set_error("%s: %s", fn, "D and B must be positive");
        return GSMVI_ERR_BAD_ARG;
    }
    if (D > ctx->max_D || B > ctx->max_B) {
        gsmvi_set_error("%s: %s", fn, "(D,B) exceeds the context's workspace; create a larger context");
        return GSMVI_ERR_WORKSPACE;
    }
    return GSMVI_OK;
}

extern "C" {

static int gsm_records(gsmvi_ctx* ctx, hipStream_t hs, int D, int B, int kc, const double* X, int ldx,
                       const double* G, int ldg, const double* mu0, const double* Pp, double* rec, int ldrec);

static int gsm_local_stage(gsmvi_ctx* ctx, hipStream_t hs, int D, int B, const double* X, int ldx, const double* G,
                           int ldg, const double* mu0, const double* S0, int lds0, double* rec, int ldrec) {
    int kc = 1;
    int st = gsmvi_panel_product(ctx, hs, ctx->stage_events(0), D, B, G, ldg, nullptr, 1.0, S0, lds0, ctx->pp, &kc);
    if (st != GSMVI_OK) return st;
    return gsm_records(ctx, hs, D, B, kc, X, ldx, G, ldg, mu0, ctx->pp, rec, ldrec);
}

static int gsm_apply(gsmvi_ctx* ctx, hipStream_t hs, int D, int B, const double* rec, int ldrec, const double* mu0,
                     const double* S0, int lds0, double* mu, double* S, int lds) {
    if (!ctx->tune_no_fast && D % 2 == 0 && (ldrec % 2 == 0) && aligned16(rec) && lds0 % 2 == 0 && lds % 2 == 0 &&
        aligned16(S0) && aligned16(S) &&                         // 16-B loads of S0 and stores of S
        gsmvi_launch_gsm_cov_sym(hs, ctx->stage_events(2), D, B, rec, ldrec, mu0, S0, lds0, S, lds, mu,
                                 ctx->tune_cov_dbg,
                                 ctx->timeline_stamps(2) ? ctx->timeline_stamps(2) :
                                 (ctx->tune_cov_dbg & 16) ? reinterpret_cast<unsigned long long*>(ctx->pp) : nullptr,
                                 ctx->num_cu)) {
        ctx->path |= GSMVI_PATH_COV_SYM;
        return check_launch("k_gsm_cov_sym");
    }
    ctx->path |= GSMVI_PATH_COV_GENERIC;
    int SB = ctx->tune_update_sb > 0 ? ctx->tune_update_sb : ((B + 1) & ~1);
    if (SB > 64) SB = 64;
    SB = (SB + 1) & ~1;
    const int s_vec_ok = (lds0 % 2 == 0) && (lds % 2 == 0) && aligned16(S0) && aligned16(S);
    gsmvi_launch_gsm_cov_update(hs, ctx->stage_events(2), D, B, rec, ldrec, mu0, S0, lds0, S, lds, mu, SB, s_vec_ok,
                                0, D);
    return check_launch("k_gsm_cov_update");
}

static int gsm_records(gsmvi_ctx* ctx, hipStream_t hs, int D, int B, int kc, const double* X, int ldx,
                       const double* G, int ldg, const double* mu0, const double* Pp, double* rec, int ldrec) {
    if (!ctx->tune_no_fast && D <= 8192 &&
        gsmvi_launch_gsm_scalars_fast(hs, ctx->stage_events(1), D, B, kc, X, ldx, G, ldg, mu0, Pp, rec, ldrec,
                                      ctx->tune_scalars_nt, ctx->timeline_stamps(1))) {
        ctx->path |= GSMVI_PATH_SCALARS_FAST;
        return check_launch("k_gsm_scalars_fast");
    }
    ctx->path |= GSMVI_PATH_SCALARS_GENERIC;
    gsmvi_launch_gsm_scalars(hs, ctx->stage_events(1), D, B, kc, X, ldx, G, ldg, mu0, Pp, rec, ldrec);
    return check_launch("k_gsm_scalars");
}

int gsmvi_gsm_update_f64(gsmvi_ctx* ctx, void* stream, int D, int B, const double* X, int ldx, const double* G,
                         int ldg, const double* mu0, const double* S0, int lds0, double* mu, double* S, int lds) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!X || !G || !mu0 || !S0 || !mu || !S, "NULL array");
    BAD_ARG(ldx < D || ldg < D || lds0 < D || lds < D, "leading dimension smaller than D");
    BAD_ARG(S == S0 || mu == mu0, "outputs must not alias inputs");
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    const int ldrec = 3 * D + (D & 1);            // even stride keeps every record 16-byte aligned
    st = gsm_local_stage(ctx, hs, D, B, X, ldx, G, ldg, mu0, S0, lds0, ctx->sg, ldrec);
    if (st != GSMVI_OK) return st;
    return gsm_apply(ctx, hs, D, B, ctx->sg, ldrec, mu0, S0, lds0, mu, S, lds);
}

int gsmvi_gsm_record_len(int D) { return 3 * D + (D & 1); }

// The same update for an S0 that is NOT symmetric: keeps the reference's literal semantics S = S0 + mean_b(...) with
// S0 g_b in the per-sample stage (gsm_numpy.py:7,50-53).  Row b of the panel stage is g_b^T S0^T (the transposed panel
// product), and the guarded update kernel reads all of S0 instead of its upper triangle.
int gsmvi_gsm_update_general_f64(gsmvi_ctx* ctx, void* stream, int D, int B, const double* X, int ldx, const double* G,
                                 int ldg, const double* mu0, const double* S0, int lds0, double* mu, double* S, int lds) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!X || !G || !mu0 || !S0 || !mu || !S, "NULL array");
    BAD_ARG(ldx < D || ldg < D || lds0 < D || lds < D, "leading dimension smaller than D");
    BAD_ARG(S == S0 || mu == mu0, "outputs must not alias inputs");
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    const int ldrec = 3 * D + (D & 1);
    int kc = 1;
    st = gsmvi_panel_t_product(ctx, hs, D, B, G, ldg, S0, lds0, D, ctx->pp, &kc);
    if (st != GSMVI_OK) return st;
    st = gsm_records(ctx, hs, D, B, kc, X, ldx, G, ldg, mu0, ctx->pp, ctx->sg, ldrec);
    if (st != GSMVI_OK) return st;
    int SB = (B + 1) & ~1;
    if (SB > 64) SB = 64;
    const int s_vec_ok = (lds0 % 2 == 0) && (lds % 2 == 0) && aligned16(S0) && aligned16(S);
    ctx->path |= GSMVI_PATH_COV_GENERIC;
    gsmvi_launch_gsm_cov_update(hs, nullptr, D, B, ctx->sg, ldrec, mu0, S0, lds0, S, lds, mu, SB, s_vec_ok, 0, D);
    return check_launch("k_gsm_cov_update(general)");
}

int gsmvi_gsm_local_stage_f64(gsmvi_ctx* ctx, void* stream, int D, int B_local, const double* X, int ldx,
                              const double* G, int ldg, const double* mu0, const double* S0, int lds0, double* rec,
                              int ldrec) {
    int st = check_common(ctx, D, B_local, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!X || !G || !mu0 || !S0 || !rec, "NULL array");
    BAD_ARG(ldx < D || ldg < D || lds0 < D || ldrec < 3 * D, "leading dimension too small");
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    return gsm_local_stage(ctx, hs, D, B_local, X, ldx, G, ldg, mu0, S0, lds0, rec, ldrec);
}

int gsmvi_gsm_apply_f64(gsmvi_ctx* ctx, void* stream, int D, int B, const double* rec, int ldrec, const double* mu0,
                        const double* S0, int lds0, double* mu, double* S, int lds) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!rec || !mu0 || !S0 || !mu || !S, "NULL array");
    BAD_ARG(ldrec < 3 * D || lds0 < D || lds < D, "leading dimension too small");
    BAD_ARG(S == S0 || mu == mu0, "outputs must not alias inputs");
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    return gsm_apply(ctx, hs, D, B, rec, ldrec, mu0, S0, lds0, mu, S, lds);
}

int gsmvi_gsm_rows_stage_f64(gsmvi_ctx* ctx, void* stream, int D, int B, int nrows, const double* G, int ldg,
                             const double* S0rows, int lds0, double* SGcols, int ldsg) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!G || !S0rows || !SGcols, "NULL array");
    // (nrows <= the context's max_D, not <= D: the column-sharded factor form calls this with the roles exchanged -- D = the owned
    // columns of the factor, nrows = its D rows -- for the partial product G[:, C] F[:, C]^T; round 6)
    BAD_ARG(nrows <= 0 || nrows > ctx->max_D, "nrows out of range");
    BAD_ARG(ldg < D || lds0 < D || ldsg < nrows, "leading dimension too small");
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    int kc = 1;
    st = gsmvi_panel_t_product(ctx, hs, D, B, G, ldg, S0rows, lds0, nrows, ctx->pp, &kc);
    if (st != GSMVI_OK) return st;
    return gsmvi_panel_finish(hs, nrows, B, kc, ctx->pp, nullptr, SGcols, ldsg);
}

int gsmvi_gsm_records_f64(gsmvi_ctx* ctx, void* stream, int D, int B, const double* X, int ldx, const double* G,
                          int ldg, const double* mu0, const double* SG, double* rec, int ldrec) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!X || !G || !mu0 || !SG || !rec, "NULL array");
    BAD_ARG(ldx < D || ldg < D, "leading dimension smaller than D");
    BAD_ARG(ldrec < gsmvi_gsm_record_len(D), "ldrec smaller than gsmvi_gsm_record_len(D)");
    // the gathered SG (B x D, contiguous) is exactly one partial slab
    return gsm_records(ctx, reinterpret_cast<hipStream_t>(stream), D, B, 1, X, ldx, G, ldg, mu0, SG, rec, ldrec);
}

int gsmvi_gsm_apply_rows_f64(gsmvi_ctx* ctx, void* stream, int D, int B, int row0, int nrows, const double* rec,
                             int ldrec, const double* mu0, const double* S0rows, int lds0, double* mu,
                             double* Srows, int lds) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!rec || !mu0 || !S0rows || !Srows, "NULL array");
    BAD_ARG(row0 < 0 || nrows <= 0 || row0 + nrows > D, "row block out of range");
    BAD_ARG(lds0 < D || lds < D, "leading dimension smaller than D");
    BAD_ARG(ldrec < gsmvi_gsm_record_len(D), "ldrec smaller than gsmvi_gsm_record_len(D)");
    BAD_ARG(Srows == S0rows || (mu && mu == mu0), "outputs must not alias inputs");
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    int SB = ctx->tune_update_sb > 0 ? ctx->tune_update_sb : ((B + 1) & ~1);
    if (SB > 64) SB = 64;
    SB = (SB + 1) & ~1;
    const int s_vec_ok = (lds0 % 2 == 0) && (lds % 2 == 0) && aligned16(S0rows) && aligned16(Srows);
    ctx->path |= GSMVI_PATH_COV_GENERIC;           // (the row-block kernel is the guarded one: a shard's rows need no mirror tiles)
    gsmvi_launch_gsm_cov_update(hs, ctx->stage_events(2), D, B, rec, ldrec, mu0, S0rows, lds0, Srows, lds, mu, SB,
                                s_vec_ok, row0, nrows);
    return check_launch("k_gsm_cov_update(rows)");
}

int gsmvi_gaussian_score_f64(gsmvi_ctx* ctx, void* stream, int D, int B, const double* X, int ldx,
                             const double* m, const double* P, int ldp, double* G, int ldg) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!X || !m || !P || !G, "NULL array");
    BAD_ARG(ldx < D || ldp < D || ldg < D, "leading dimension smaller than D");
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    return gsmvi_panel_product_out(ctx, hs, D, D, B, X, ldx, m, -1.0, P, ldp, nullptr, G, ldg);
}

int gsmvi_sample_f64(gsmvi_ctx* ctx, void* stream, int D, int B, const double* Z, int ldz, const double* mu,
                     const double* R, int ldr, double* X, int ldx) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!Z || !mu || !R || !X, "NULL array");
    BAD_ARG(ldz < D || ldr < D || ldx < D, "leading dimension smaller than D");
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    return gsmvi_panel_product_out(ctx, hs, D, D, B, Z, ldz, nullptr, 1.0, R, ldr, mu, X, ldx);
}

int gsmvi_sample_cols_f64(gsmvi_ctx* ctx, void* stream, int D, int B, int ncols, const double* Z, int ldz,
                          const double* mu_cols, const double* Fcols, int ldf, double* Xcols, int ldx) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!Z || !mu_cols || !Fcols || !Xcols, "NULL array");
    BAD_ARG(ncols <= 0 || ncols > D, "ncols out of range");
    BAD_ARG(ldz < D || ldf < ncols || ldx < ncols, "leading dimension too small");
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    return gsmvi_panel_product_out(ctx, hs, D, ncols, B, Z, ldz, nullptr, 1.0, Fcols, ldf, mu_cols, Xcols, ldx);
}

int gsmvi_gsm_factor_apply_cols_f64(gsmvi_ctx* ctx, void* stream, int D, int B, int col0, int ncols, const double* Z, int ldz,
                                    const double* W, const double* X, int ldx, const double* mu0, const double* F0cols,
                                    int ldf0, double* mu, double* Fcols, int ldf, int* info_dev, int* n_reverts_dev) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!Z || !W || !X || !mu0 || !F0cols || !mu || !Fcols || !info_dev, "NULL argument");
    BAD_ARG(col0 < 0 || ncols <= 0 || col0 + ncols > D, "column block out of range");
    BAD_ARG(col0 % 64 != 0 || (ncols % 64 != 0 && col0 + ncols != D), "column blocks are tile aligned (multiples of 64; the last one may be ragged)");
    BAD_ARG(ldz < D || ldx < D || ldf0 < ncols || ldf < ncols, "leading dimension too small");
    BAD_ARG(Fcols == F0cols || mu == mu0, "outputs must not alias inputs");
    BAD_ARG(D % 2 != 0 || ldf0 % 2 != 0 || ldf % 2 != 0 || !aligned16(F0cols) || !aligned16(Fcols),
            "the column-sharded form takes even D, even leading dimensions and 16-byte aligned blocks");
    if (2 * B > D || 2 * B > GSMVI_FACTOR_NMAX || D > 16384) {
        gsmvi_set_error("%s: %s", __func__, "the factor form needs 2B <= D and 2B <= 256");
        return GSMVI_ERR_UNSUPPORTED;
    }
    return gsmvi_factor_apply_cols_impl(ctx, reinterpret_cast<hipStream_t>(stream), D, B, col0, ncols, Z, ldz, W, D, X, ldx, mu0,
                                        F0cols, ldf0, mu, Fcols, ldf, info_dev, n_reverts_dev);
}

int gsmvi_commit_f64(gsmvi_ctx* ctx, void* stream, int D, const int* info_dev, const double* mu_new,
                     const double* S_new, int lds_new, double* mu, double* S, int lds, int* n_reverts_dev) {
    BAD_ARG(!ctx || !info_dev || !mu_new || !S_new || !mu || !S, "NULL argument");
    BAD_ARG(D <= 0 || lds_new < D || lds < D, "bad size");
    gsmvi_launch_commit(reinterpret_cast<hipStream_t>(stream), D, info_dev, mu_new, S_new, lds_new, mu, S, lds,
                        n_reverts_dev);
    return check_launch("k_commit");
}

int gsmvi_potrf_f64(gsmvi_ctx* ctx, void* stream, int D, const double* S, int lds, double* R, int ldr,
                    int* info_dev) {
    BAD_ARG(!ctx || !S || !R || !info_dev, "NULL argument");
    BAD_ARG(D <= 0 || lds < D || ldr < D, "bad size");
    BAD_ARG(S == R, "the factor must not alias the matrix (tiles of S are read while R is written)");
    if (D > ctx->max_D) {
        gsmvi_set_error("%s: %s", __func__, "D exceeds the context's workspace");
        return GSMVI_ERR_WORKSPACE;
    }
    return gsmvi_potrf_impl(ctx, reinterpret_cast<hipStream_t>(stream), D, S, lds, R, ldr, info_dev);
}

int gsmvi_gram_f64(gsmvi_ctx* ctx, void* stream, int D, const double* F, int ldf, double* C, int ldc) {
    BAD_ARG(!ctx || !F || !C, "NULL argument");
    BAD_ARG(D <= 0 || ldf < D || ldc < D, "bad size");
    BAD_ARG(F == C, "output must not alias the input");
    return gsmvi_gram_impl(reinterpret_cast<hipStream_t>(stream), D, F, ldf, C, ldc, 0.0, nullptr);
}

int gsmvi_gram_shift_f64(gsmvi_ctx* ctx, void* stream, int D, const double* F, int ldf, double shift, const double* shift_dev,
                         double* C, int ldc) {
    BAD_ARG(!ctx || !F || !C, "NULL argument");
    BAD_ARG(D <= 0 || ldf < D || ldc < D, "bad size");
    BAD_ARG(F == C, "output must not alias the input");
    BAD_ARG(!(shift == shift), "shift is NaN");
    return gsmvi_gram_impl(reinterpret_cast<hipStream_t>(stream), D, F, ldf, C, ldc, shift, shift_dev);
}

int gsmvi_whiten_rows_f64(gsmvi_ctx* ctx, void* stream, int D, int nrows, const double* R, int ldr, const double* X,
                          int ldx, const double* mu, double* Z, int ldz, double* logdiag_dev) {
    BAD_ARG(!ctx || !R || !X || !Z, "NULL argument");
    BAD_ARG(D <= 0 || nrows <= 0 || ldr < D || ldx < D || ldz < D, "bad size");
    BAD_ARG(D > 8192, "D > 8192 not supported (the residual row lives in LDS)");
    return gsmvi_whiten_impl(reinterpret_cast<hipStream_t>(stream), D, nrows, R, ldr, X, ldx, mu, Z, ldz, logdiag_dev);
}

int gsmvi_gsm_factor_update_f64(gsmvi_ctx* ctx, void* stream, int D, int B, const double* Z, int ldz, const double* X,
                                int ldx, const double* G, int ldg, const double* mu0, const double* F0, int ldf0,
                                double* mu, double* F, int ldf, int* info_dev, int* n_reverts_dev) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!Z || !X || !G || !mu0 || !F0 || !mu || !F || !info_dev, "NULL argument");
    BAD_ARG(ldz < D || ldx < D || ldg < D || ldf0 < D || ldf < D, "leading dimension smaller than D");
    BAD_ARG(F == F0 || mu == mu0, "outputs must not alias inputs");
    if (2 * B > D || 2 * B > GSMVI_FACTOR_NMAX || D > 16384) {
        gsmvi_set_error("%s: %s", __func__, "the factor form needs 2B <= D and 2B <= 256; use gsmvi_gsm_update_f64");
        return GSMVI_ERR_UNSUPPORTED;
    }
    return gsmvi_factor_impl(ctx, reinterpret_cast<hipStream_t>(stream), D, B, Z, ldz, X, ldx, G, ldg, mu0, F0, ldf0,
                             mu, F, ldf, info_dev, n_reverts_dev);
}

int gsmvi_gsm_factor_local_stage_f64(gsmvi_ctx* ctx, void* stream, int D, int B_local, const double* Z, int ldz,
                                     const double* X, int ldx, const double* G, int ldg, const double* mu0,
                                     const double* F0, int ldf0, double* rec, int ldrec) {
    int st = check_common(ctx, D, B_local, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!Z || !X || !G || !mu0 || !F0 || !rec, "NULL argument");
    BAD_ARG(ldz < D || ldx < D || ldg < D || ldf0 < D, "leading dimension smaller than D");
    BAD_ARG(ldrec < gsmvi_gsm_record_len(D), "ldrec smaller than gsmvi_gsm_record_len(D)");
    if (D > 16384) {
        gsmvi_set_error("%s: %s", __func__, "the factor form supports D <= 16384");
        return GSMVI_ERR_UNSUPPORTED;
    }
    return gsmvi_factor_local_impl(ctx, reinterpret_cast<hipStream_t>(stream), D, B_local, Z, ldz, X, ldx, G, ldg, mu0,
                                   F0, ldf0, rec, ldrec);
}

int gsmvi_gsm_factor_apply_f64(gsmvi_ctx* ctx, void* stream, int D, int B, const double* Z, int ldz,
                               const double* rec, int ldrec, const double* mu0, const double* F0, int ldf0,
                               double* mu, double* F, int ldf, int* info_dev, int* n_reverts_dev) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!Z || !rec || !mu0 || !F0 || !mu || !F || !info_dev, "NULL argument");
    BAD_ARG(ldz < D || ldf0 < D || ldf < D, "leading dimension smaller than D");
    BAD_ARG(ldrec < gsmvi_gsm_record_len(D), "ldrec smaller than gsmvi_gsm_record_len(D)");
    BAD_ARG(F == F0 || mu == mu0, "outputs must not alias inputs");
    if (2 * B > D || 2 * B > GSMVI_FACTOR_NMAX || D > 16384) {
        gsmvi_set_error("%s: %s", __func__, "the factor form needs 2B <= D and 2B <= 256; use gsmvi_gsm_update_f64");
        return GSMVI_ERR_UNSUPPORTED;
    }
    return gsmvi_factor_apply_impl(ctx, reinterpret_cast<hipStream_t>(stream), D, B, Z, ldz, rec, ldrec, mu0, F0, ldf0,
                                   mu, F, ldf, info_dev, n_reverts_dev);
}

int gsmvi_bam_update_f64(gsmvi_ctx* ctx, void* stream, int D, int B, const double* X, int ldx, const double* G,
                         int ldg, const double* mu0, const double* S0, int lds0, double reg, double jitter,
                         double* mu, double* S, int lds, int* info_dev) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!X || !G || !mu0 || !S0 || !mu || !S, "NULL array");
    BAD_ARG(ldx < D || ldg < D || lds0 < D || lds < D, "leading dimension smaller than D");
    BAD_ARG(S == S0 || mu == mu0, "outputs must not alias inputs");
    BAD_ARG(!(reg > 0.0), "reg must be positive");
    return gsmvi_bam_impl(ctx, reinterpret_cast<hipStream_t>(stream), D, B, X, ldx, G, ldg, mu0, S0, lds0, reg,
                          jitter, mu, S, lds, info_dev);
}

int gsmvi_bam_factor_update_f64(gsmvi_ctx* ctx, void* stream, int D, int B, const double* Z, int ldz, const double* X,
                                int ldx, const double* G, int ldg, const double* mu0, const double* F0, int ldf0,
                                double reg, double* mu, double* F, int ldf, int* info_dev, int* n_reverts_dev) {
    int st = check_common(ctx, D, B, __func__);
    if (st != GSMVI_OK) return st;
    BAD_ARG(!Z || !X || !G || !mu0 || !F0 || !mu || !F || !info_dev, "NULL argument");
    BAD_ARG(ldz < D || ldx < D || ldg < D || ldf0 < D || ldf < D, "leading dimension smaller than D");
    BAD_ARG(F == F0 || mu == mu0, "outputs must not alias inputs");
    BAD_ARG(!(reg > 0.0), "reg must be positive");
    if (2 * B > D || 2 * B > GSMVI_FACTOR_NMAX || D > 16384) {
        gsmvi_set_error("%s: %s", __func__, "the factor form needs 2B <= D and 2B <= 256; use gsmvi_bam_update_f64");
        return GSMVI_ERR_UNSUPPORTED;
    }
    return gsmvi_bam_factor_impl(ctx, reinterpret_cast<hipStream_t>(stream), D, B, Z, ldz, X, ldx, G, ldg, mu0, F0, ldf0, reg,
                                 mu, F, ldf, info_dev, n_reverts_dev);
}

}  // extern "C"
