// RCCL-taking C entry points of the batch-sharded updates (SURVEY 8(b) minimum set: "RCCL-sharded variants taking an
// ncclComm_t"; north_star: per-sample contributions exchanged over xGMI before the combined rank-B update).
// Host code only.  The library does NOT link RCCL: the RCCL functions it needs are resolved once (std::call_once) either
// from a library handle the caller supplies (gsmvi_set_rccl_library: the dlopen handle of the RCCL instance that CREATED
// the communicator -- the safe choice when several RCCL copies live in one process, e.g. torch's bundled one), or from the
// RCCL instance already loaded in the calling process, falling back to loading librccl.so.1; a process that never shards
// never loads RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>

#include "../../include/gsmvi_hip.h"
#include "gsmvi_ctx.h"

int gsmvi_bam_small_nmax();

namespace {
typedef ncclResult_t (*all_gather_fn)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
typedef ncclResult_t (*comm_int_fn)(const ncclComm_t, int*);
typedef const char* (*err_str_fn)(ncclResult_t);
struct rccl_api {
    all_gather_fn all_gather = nullptr;
    comm_int_fn count = nullptr, user_rank = nullptr;
    err_str_fn err_str = nullptr;
};
rccl_api g_rccl;
std::mutex g_rccl_mu;
int g_rccl_state = 0;                            // 0 = not tried yet, 1 = resolved, -1 = the last attempt found no RCCL
void* g_rccl_user_handle = nullptr;              // set by gsmvi_set_rccl_library; a new handle re-arms a failed resolve

void resolve_rccl() {
    void* h = g_rccl_user_handle;
    if (!h && dlsym(RTLD_DEFAULT, "ncclAllGather")) h = RTLD_DEFAULT;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (int k = 0; !h && k < 2; ++k) h = dlopen(names[k], RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // already loaded?
    for (int k = 0; !h && k < 2; ++k) h = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    rccl_api a;
    a.all_gather = reinterpret_cast<all_gather_fn>(dlsym(h, "ncclAllGather"));
    a.count = reinterpret_cast<comm_int_fn>(dlsym(h, "ncclCommCount"));
    a.user_rank = reinterpret_cast<comm_int_fn>(dlsym(h, "ncclCommUserRank"));
    a.err_str = reinterpret_cast<err_str_fn>(dlsym(h, "ncclGetErrorString"));
    if (a.all_gather && a.count && a.user_rank) g_rccl = a;
}

bool load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl_state == 0) {
        resolve_rccl();
        g_rccl_state = g_rccl.all_gather != nullptr ? 1 : -1;
    }
    return g_rccl_state == 1;
}

// communicator geometry, validated BEFORE anything is enqueued
int comm_geometry(const char* fn, void* nccl_comm, int* nranks, int* rank) {
    if (!load_rccl()) {
        gsmvi_set_error("%s: %s", fn, "no RCCL library (ncclAllGather) found in this process or on the loader path");
        return GSMVI_ERR_UNSUPPORTED;
    }
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    ncclResult_t r = g_rccl.count(comm, nranks);
    if (r == ncclSuccess) r = g_rccl.user_rank(comm, rank);
    if (r != ncclSuccess || *nranks < 1 || *rank < 0 || *rank >= *nranks) {
        gsmvi_set_error("%s: querying the communicator failed: %s", fn,
                        (r != ncclSuccess && g_rccl.err_str) ? g_rccl.err_str(r) : "bad rank / size");
        return GSMVI_ERR_BAD_ARG;
    }
    return GSMVI_OK;
}

int gather(const char* fn, void* nccl_comm, void* stream, const double* mine, double* all, size_t count, int nranks) {
    if (nranks <= 1) return GSMVI_OK;
    ncclResult_t r = g_rccl.all_gather(mine, all, count, ncclDouble, reinterpret_cast<ncclComm_t>(nccl_comm),
                                       reinterpret_cast<hipStream_t>(stream));
    if (r != ncclSuccess) {
        gsmvi_set_error("%s: ncclAllGather failed: %s", fn, g_rccl.err_str ? g_rccl.err_str(r) : "?");
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}
}  // namespace

extern "C" int gsmvi_set_rccl_library(void* dl_handle) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl_state == 1) {
        gsmvi_set_error("%s: %s", __func__, "RCCL was already resolved; call this before the first sharded entry point");
        return GSMVI_ERR_BAD_ARG;
    }
    // not tried yet, or an earlier attempt found no RCCL (round-3 advice: that attempt used to consume the one-shot resolver,
    // so a handle supplied afterwards was silently never used): the next sharded call resolves again, from this handle
    g_rccl_user_handle = dl_handle;
    g_rccl_state = 0;
    return GSMVI_OK;
}

extern "C" int gsmvi_gsm_update_sharded_f64(gsmvi_ctx* ctx, void* stream, void* nccl_comm, int D, int B_local,
                                            const double* X_local, int ldx, const double* G_local, int ldg,
                                            const double* mu0, const double* S0, int lds0, double* rec_all,
                                            double* mu, double* S, int lds) {
    if (!ctx || !nccl_comm || !X_local || !G_local || !mu0 || !S0 || !rec_all || !mu || !S) {
        gsmvi_set_error("%s: %s", __func__, "NULL argument");
        return GSMVI_ERR_BAD_ARG;
    }
    int nranks = 0, rank = 0;
    int st = comm_geometry(__func__, nccl_comm, &nranks, &rank);
    if (st != GSMVI_OK) return st;
    if (D <= 0 || B_local <= 0 || D > ctx->max_D || (long long)B_local * nranks > ctx->max_B) {
        gsmvi_set_error("%s: %s", __func__, "(D, B_local x ranks) exceeds the context's workspace; create a larger context");
        return GSMVI_ERR_WORKSPACE;
    }
    const int ldrec = gsmvi_gsm_record_len(D);
    const size_t count = (size_t)B_local * (size_t)ldrec;
    double* mine = rec_all + (size_t)rank * count;                      // in-place all-gather: own slot of the result
    st = gsmvi_gsm_local_stage_f64(ctx, stream, D, B_local, X_local, ldx, G_local, ldg, mu0, S0, lds0, mine, ldrec);
    if (st != GSMVI_OK) return st;
    if ((st = gather(__func__, nccl_comm, stream, mine, rec_all, count, nranks)) != GSMVI_OK) return st;
    return gsmvi_gsm_apply_f64(ctx, stream, D, B_local * nranks, rec_all, ldrec, mu0, S0, lds0, mu, S, lds);
}

extern "C" int gsmvi_gsm_factor_update_sharded_f64(gsmvi_ctx* ctx, void* stream, void* nccl_comm, int D, int B_local,
                                                   const double* Z_all, int ldz, const double* X_local, int ldx,
                                                   const double* G_local, int ldg, const double* mu0,
                                                   const double* F0, int ldf0, double* rec_all, double* mu, double* F,
                                                   int ldf, int* info_dev, int* n_reverts_dev) {
    if (!ctx || !nccl_comm || !Z_all || !X_local || !G_local || !mu0 || !F0 || !rec_all || !mu || !F || !info_dev) {
        gsmvi_set_error("%s: %s", __func__, "NULL argument");
        return GSMVI_ERR_BAD_ARG;
    }
    int nranks = 0, rank = 0;
    int st = comm_geometry(__func__, nccl_comm, &nranks, &rank);
    if (st != GSMVI_OK) return st;
    const long long B = (long long)B_local * nranks;
    if (D <= 0 || B_local <= 0 || D > ctx->max_D || B > ctx->max_B) {
        gsmvi_set_error("%s: %s", __func__, "(D, B_local x ranks) exceeds the context's workspace; create a larger context");
        return GSMVI_ERR_WORKSPACE;
    }
    if (2 * B > D || 2 * B > GSMVI_FACTOR_NMAX) {
        gsmvi_set_error("%s: %s", __func__, "the factor form needs 2B <= D and 2B <= its chain's bound for the combined batch");
        return GSMVI_ERR_UNSUPPORTED;
    }
    const int ldrec = gsmvi_gsm_record_len(D);
    const size_t count = (size_t)B_local * (size_t)ldrec;
    double* mine = rec_all + (size_t)rank * count;
    st = gsmvi_gsm_factor_local_stage_f64(ctx, stream, D, B_local, Z_all + (size_t)rank * B_local * ldz, ldz, X_local, ldx,
                                          G_local, ldg, mu0, F0, ldf0, mine, ldrec);
    if (st != GSMVI_OK) return st;
    if ((st = gather(__func__, nccl_comm, stream, mine, rec_all, count, nranks)) != GSMVI_OK) return st;
    return gsmvi_gsm_factor_apply_f64(ctx, stream, D, (int)B, Z_all, ldz, rec_all, ldrec, mu0, F0, ldf0, mu, F, ldf, info_dev,
                                      n_reverts_dev);
}

extern "C" int gsmvi_bam_update_sharded_f64(gsmvi_ctx* ctx, void* stream, void* nccl_comm, int D, int B_local,
                                            const double* X_local, int ldx, const double* G_local, int ldg,
                                            const double* mu0, const double* S0, int lds0, double reg, double jitter,
                                            double* xg_all, double* mu, double* S, int lds, int* info_dev) {
    if (!ctx || !nccl_comm || !X_local || !G_local || !mu0 || !S0 || !xg_all || !mu || !S) {
        gsmvi_set_error("%s: %s", __func__, "NULL argument");
        return GSMVI_ERR_BAD_ARG;
    }
    int nranks = 0, rank = 0;
    int st = comm_geometry(__func__, nccl_comm, &nranks, &rank);
    if (st != GSMVI_OK) return st;
    const long long B = (long long)B_local * nranks;
    if (D <= 0 || B_local <= 0 || ldx < D || ldg < D || lds0 < D || lds < D) {
        gsmvi_set_error("%s: %s", __func__, "non-positive size or a leading dimension smaller than D");
        return GSMVI_ERR_BAD_ARG;
    }
    if (D > ctx->max_D || B > ctx->max_B) {
        gsmvi_set_error("%s: %s", __func__, "(D, B_local x ranks) exceeds the context's workspace; create a larger context");
        return GSMVI_ERR_WORKSPACE;
    }
    if (B > gsmvi_bam_small_nmax()) {            // the combined batch, checked BEFORE the staging copies and the all-gathers
        gsmvi_set_error("%s: %s", __func__, "B_local x ranks exceeds the device chain of the BaM update (B <= 1024)");
        return GSMVI_ERR_UNSUPPORTED;
    }
    // BaM's statistics couple all samples: the ranks exchange their (x_b, g_b) rows -- two all-gathers of B_local x D doubles
    // per rank (256 KiB at D = 1024, B = 128, 8 ranks) -- and every replica runs the identical update
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    double* Xall = xg_all;
    double* Gall = xg_all + (size_t)B * D;
    const size_t count = (size_t)B_local * D;
    hipError_t e = hipMemcpy2DAsync(Xall + rank * count, (size_t)D * sizeof(double), X_local, (size_t)ldx * sizeof(double),
                                    (size_t)D * sizeof(double), (size_t)B_local, hipMemcpyDeviceToDevice, hs);
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(Gall + rank * count, (size_t)D * sizeof(double), G_local, (size_t)ldg * sizeof(double),
                             (size_t)D * sizeof(double), (size_t)B_local, hipMemcpyDeviceToDevice, hs);
    if (e != hipSuccess) {
        gsmvi_set_error("%s: staging copy failed: %s", __func__, hipGetErrorString(e));
        return GSMVI_ERR_HIP;
    }
    if ((st = gather(__func__, nccl_comm, stream, Xall + rank * count, Xall, count, nranks)) != GSMVI_OK) return st;
    if ((st = gather(__func__, nccl_comm, stream, Gall + rank * count, Gall, count, nranks)) != GSMVI_OK) return st;
    return gsmvi_bam_update_f64(ctx, stream, D, (int)B, Xall, D, Gall, D, mu0, S0, lds0, reg, jitter, mu, S, lds, info_dev);
}

extern "C" int gsmvi_bam_factor_update_sharded_f64(gsmvi_ctx* ctx, void* stream, void* nccl_comm, int D, int B_local,
                                                   const double* Z_all, int ldz, const double* X_local, int ldx,
                                                   const double* G_local, int ldg, const double* mu0, const double* F0,
                                                   int ldf0, double reg, double* xg_all, double* mu, double* F, int ldf,
                                                   int* info_dev, int* n_reverts_dev) {
    if (!ctx || !nccl_comm || !Z_all || !X_local || !G_local || !mu0 || !F0 || !xg_all || !mu || !F || !info_dev) {
        gsmvi_set_error("%s: %s", __func__, "NULL argument");
        return GSMVI_ERR_BAD_ARG;
    }
    int nranks = 0, rank = 0;
    int st = comm_geometry(__func__, nccl_comm, &nranks, &rank);
    if (st != GSMVI_OK) return st;
    const long long B = (long long)B_local * nranks;
    if (D <= 0 || B_local <= 0 || ldz < D || ldx < D || ldg < D || ldf0 < D || ldf < D) {
        gsmvi_set_error("%s: %s", __func__, "non-positive size or a leading dimension smaller than D");
        return GSMVI_ERR_BAD_ARG;
    }
    if (D > ctx->max_D || B > ctx->max_B) {
        gsmvi_set_error("%s: %s", __func__, "(D, B_local x ranks) exceeds the context's workspace; create a larger context");
        return GSMVI_ERR_WORKSPACE;
    }
    if (2 * B > D || 2 * B > GSMVI_FACTOR_NMAX) {
        gsmvi_set_error("%s: %s", __func__, "the factor form needs 2B <= D and 2B <= its chain's bound for the combined batch");
        return GSMVI_ERR_UNSUPPORTED;
    }
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    double* Xall = xg_all;
    double* Gall = xg_all + (size_t)B * D;
    const size_t count = (size_t)B_local * D;
    hipError_t e = hipMemcpy2DAsync(Xall + rank * count, (size_t)D * sizeof(double), X_local, (size_t)ldx * sizeof(double),
                                    (size_t)D * sizeof(double), (size_t)B_local, hipMemcpyDeviceToDevice, hs);
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(Gall + rank * count, (size_t)D * sizeof(double), G_local, (size_t)ldg * sizeof(double),
                             (size_t)D * sizeof(double), (size_t)B_local, hipMemcpyDeviceToDevice, hs);
    if (e != hipSuccess) {
        gsmvi_set_error("%s: staging copy failed: %s", __func__, hipGetErrorString(e));
        return GSMVI_ERR_HIP;
    }
    if ((st = gather(__func__, nccl_comm, stream, Xall + rank * count, Xall, count, nranks)) != GSMVI_OK) return st;
    if ((st = gather(__func__, nccl_comm, stream, Gall + rank * count, Gall, count, nranks)) != GSMVI_OK) return st;
    return gsmvi_bam_factor_update_f64(ctx, stream, D, (int)B, Z_all, ldz, Xall, D, Gall, D, mu0, F0, ldf0, reg, mu, F, ldf,
                                       info_dev, n_reverts_dev);
}
