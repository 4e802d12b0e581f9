/* Plain-C consumer of the RCCL-taking entry point (include/gsmvi_hip.h: gsmvi_gsm_update_sharded_f64): one process
 * per GPU, the caller owns the ncclComm_t.
 *   rccl_sharded <in.bin> <out.bin> <nranks> <rank> <idfile>
 * Rank 0 writes the ncclUniqueId to <idfile>, the other ranks wait for it.  Every rank reads the full problem of
 * in.bin (format of abi_smoke.c), takes rows [rank*B/nranks, (rank+1)*B/nranks) of X and G as its shard and writes
 * mu[D], S[D*D] of the combined update to <out.bin>.<rank>. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include "gsmvi_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_ABI(x) do { int s_ = (x); if (s_ != GSMVI_OK) { fprintf(stderr, "%s -> %s (%s)\n", #x, gsmvi_status_string(s_), gsmvi_last_error()); return 3; } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 4; } } while (0)

int main(int argc, char** argv) {
    if (argc != 6) return 1;
    const int nranks = atoi(argv[3]), rank = atoi(argv[4]);
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    int D, B;
    if (fread(&D, 4, 1, f) != 1 || fread(&B, 4, 1, f) != 1) return 1;
    const size_t nbd = (size_t)B * D, ndd = (size_t)D * D;
    double* h = (double*)malloc(sizeof(double) * (2 * nbd + D + ndd));
    if (fread(h, sizeof(double), 2 * nbd + D + ndd, f) != 2 * nbd + D + ndd) return 1;
    fclose(f);
    if (B % nranks) return 1;
    const int Bl = B / nranks;

    int ndev = 0;
    CHECK_HIP(hipGetDeviceCount(&ndev));
    const int dev = rank % ndev;
    CHECK_HIP(hipSetDevice(dev));
    ncclUniqueId id;
    if (rank == 0) {
        CHECK_NCCL(ncclGetUniqueId(&id));
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", argv[5]);
        f = fopen(tmp, "wb");
        if (!f || fwrite(&id, sizeof id, 1, f) != 1) return 1;
        fclose(f);
        rename(tmp, argv[5]);
    } else {
        int tries = 0;
        while ((f = fopen(argv[5], "rb")) == NULL && tries++ < 600) usleep(100000);
        if (!f || fread(&id, sizeof id, 1, f) != 1) return 1;
        fclose(f);
    }
    ncclComm_t comm;
    CHECK_NCCL(ncclCommInitRank(&comm, nranks, id, rank));

    gsmvi_ctx* ctx = NULL;
    CHECK_ABI(gsmvi_create(&ctx, dev, D, B));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    const int ldrec = gsmvi_gsm_record_len(D);
    double *X, *G, *mu0, *S0, *mu, *S, *rec;
    CHECK_HIP(hipMalloc((void**)&X, sizeof(double) * Bl * D));
    CHECK_HIP(hipMalloc((void**)&G, sizeof(double) * Bl * D));
    CHECK_HIP(hipMalloc((void**)&mu0, sizeof(double) * D));
    CHECK_HIP(hipMalloc((void**)&S0, sizeof(double) * ndd));
    CHECK_HIP(hipMalloc((void**)&mu, sizeof(double) * D));
    CHECK_HIP(hipMalloc((void**)&S, sizeof(double) * ndd));
    CHECK_HIP(hipMalloc((void**)&rec, sizeof(double) * (size_t)B * ldrec));
    CHECK_HIP(hipMemcpy(X, h + (size_t)rank * Bl * D, sizeof(double) * Bl * D, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(G, h + nbd + (size_t)rank * Bl * D, sizeof(double) * Bl * D, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(mu0, h + 2 * nbd, sizeof(double) * D, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(S0, h + 2 * nbd + D, sizeof(double) * ndd, hipMemcpyHostToDevice));

    for (int rep = 0; rep < 3; ++rep)       /* repeated calls: the resolved RCCL entry points are reused */
        CHECK_ABI(gsmvi_gsm_update_sharded_f64(ctx, st, comm, D, Bl, X, D, G, D, mu0, S0, D, rec, mu, S, D));
    if (gsmvi_gsm_update_sharded_f64(ctx, st, NULL, D, Bl, X, D, G, D, mu0, S0, D, rec, mu, S, D) != GSMVI_ERR_BAD_ARG)
        return 5;
    CHECK_HIP(hipStreamSynchronize(st));

    double* out = (double*)malloc(sizeof(double) * (D + ndd));
    CHECK_HIP(hipMemcpy(out, mu, sizeof(double) * D, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + D, S, sizeof(double) * ndd, hipMemcpyDeviceToHost));
    char name[4096];
    snprintf(name, sizeof name, "%s.%d", argv[2], rank);
    f = fopen(name, "wb");
    if (!f) return 1;
    fwrite(out, sizeof(double), D + ndd, f);
    fclose(f);
    CHECK_NCCL(ncclCommDestroy(comm));
    CHECK_ABI(gsmvi_destroy(ctx));
    return 0;
}
