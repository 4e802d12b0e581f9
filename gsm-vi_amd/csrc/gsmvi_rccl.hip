// RCCL-taking C entry point of the batch-sharded dense update (SURVEY 8(b) minimum set: "RCCL-sharded variants
// taking an ncclComm_t"; north_star: per-sample contributions exchanged over xGMI before the combined rank-B update).
// Host code only.  The library does NOT link RCCL: the three RCCL functions it needs are resolved at first use from
// the RCCL instance already loaded in the calling process (the one that created the communicator), falling back to
// loading librccl.so.1, so a process that never shards never loads RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../include/gsmvi_hip.h"
#include "gsmvi_ctx.h"

namespace {
typedef ncclResult_t (*all_gather_fn)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
typedef ncclResult_t (*comm_int_fn)(const ncclComm_t, int*);
typedef const char* (*err_str_fn)(ncclResult_t);
struct rccl_api {
    all_gather_fn all_gather = nullptr;
    comm_int_fn count = nullptr, user_rank = nullptr;
    err_str_fn err_str = nullptr;
    bool tried = false;
};
rccl_api g_rccl;

bool load_rccl() {
    if (g_rccl.tried) return g_rccl.all_gather != nullptr;
    g_rccl.tried = true;
    void* h = nullptr;
    if (dlsym(RTLD_DEFAULT, "ncclAllGather")) h = RTLD_DEFAULT;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (int k = 0; !h && k < 2; ++k) h = dlopen(names[k], RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // already loaded?
    for (int k = 0; !h && k < 2; ++k) h = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    g_rccl.all_gather = reinterpret_cast<all_gather_fn>(dlsym(h, "ncclAllGather"));
    g_rccl.count = reinterpret_cast<comm_int_fn>(dlsym(h, "ncclCommCount"));
    g_rccl.user_rank = reinterpret_cast<comm_int_fn>(dlsym(h, "ncclCommUserRank"));
    g_rccl.err_str = reinterpret_cast<err_str_fn>(dlsym(h, "ncclGetErrorString"));
    if (!g_rccl.all_gather || !g_rccl.count || !g_rccl.user_rank) g_rccl.all_gather = nullptr;
    return g_rccl.all_gather != nullptr;
}
}  // namespace

extern "C" int gsmvi_gsm_update_sharded_f64(gsmvi_ctx* ctx, void* stream, void* nccl_comm, int D, int B_local,
                                            const double* X_local, int ldx, const double* G_local, int ldg,
                                            const double* mu0, const double* S0, int lds0, double* rec_all,
                                            double* mu, double* S, int lds) {
    if (!ctx || !nccl_comm || !X_local || !G_local || !mu0 || !S0 || !rec_all || !mu || !S) {
        gsmvi_set_error("%s: %s", __func__, "NULL argument");
        return GSMVI_ERR_BAD_ARG;
    }
    if (!load_rccl()) {
        gsmvi_set_error("%s: %s", __func__, "no RCCL library (ncclAllGather) found in this process or on the loader path");
        return GSMVI_ERR_UNSUPPORTED;
    }
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    int nranks = 0, rank = 0;
    ncclResult_t r = g_rccl.count(comm, &nranks);
    if (r == ncclSuccess) r = g_rccl.user_rank(comm, &rank);
    if (r != ncclSuccess || nranks < 1 || rank < 0 || rank >= nranks) {
        gsmvi_set_error("%s: querying the communicator failed: %s", __func__,
                        (r != ncclSuccess && g_rccl.err_str) ? g_rccl.err_str(r) : "bad rank / size");
        return GSMVI_ERR_BAD_ARG;
    }
    const int ldrec = gsmvi_gsm_record_len(D);
    const size_t count = (size_t)B_local * (size_t)ldrec;
    double* mine = rec_all + (size_t)rank * count;                      // in-place all-gather: own slot of the result
    int st = gsmvi_gsm_local_stage_f64(ctx, stream, D, B_local, X_local, ldx, G_local, ldg, mu0, S0, lds0, mine, ldrec);
    if (st != GSMVI_OK) return st;
    if (nranks > 1) {
        r = g_rccl.all_gather(mine, rec_all, count, ncclDouble, comm, reinterpret_cast<hipStream_t>(stream));
        if (r != ncclSuccess) {
            gsmvi_set_error("%s: ncclAllGather failed: %s", __func__, g_rccl.err_str ? g_rccl.err_str(r) : "?");
            return GSMVI_ERR_HIP;
        }
    }
    return gsmvi_gsm_apply_f64(ctx, stream, D, B_local * nranks, rec_all, ldrec, mu0, S0, lds0, mu, S, lds);
}
