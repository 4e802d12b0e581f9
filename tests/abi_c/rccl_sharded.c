/* Plain-C consumer of the RCCL-taking entry points (include/gsmvi_hip.h: gsmvi_gsm_update_sharded_f64,
 * gsmvi_gsm_factor_update_sharded_f64, gsmvi_bam_update_sharded_f64, gsmvi_bam_factor_update_sharded_f64): one process per GPU, the caller owns the ncclComm_t.
 *   rccl_sharded <in.bin> <out.bin> <nranks> <rank> <idfile>
 * Rank 0 writes the ncclUniqueId to <idfile>, the other ranks wait for it.  Every rank reads the full problem of
 * in.bin (int D, int B, then X[B*D], G[B*D], mu0[D], S0[D*D], Z[B*D], F0[D*D] with X = mu0 + Z F0, S0 = F0^T F0), takes
 * rows [rank*B/nranks, (rank+1)*B/nranks) of X and G as its shard and writes to <out.bin>.<rank>:
 *   mu[D], S[D*D] (dense update) | mu[D], F[D*D], flag (factor-form update) | mu[D], S[D*D] (BaM update, reg 2, jitter 0)
 *   | mu[D], F[D*D], flag (factor-form BaM update, reg 2; only when 2B <= D, zeros otherwise). */
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
    const size_t nin = 3 * nbd + D + 2 * ndd;
    double* h = (double*)malloc(sizeof(double) * nin);
    if (fread(h, sizeof(double), nin, f) != nin) return 1;
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

    const size_t nout = 4 * (D + ndd) + 2;
    double* out = (double*)malloc(sizeof(double) * nout);
    CHECK_HIP(hipMemcpy(out, mu, sizeof(double) * D, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + D, S, sizeof(double) * ndd, hipMemcpyDeviceToHost));

    /* factor-form update through the communicator: Z is replicated, X and G are this rank's rows */
    double *Z, *F0, *xg;
    int* flag;
    CHECK_HIP(hipMalloc((void**)&Z, sizeof(double) * nbd));
    CHECK_HIP(hipMalloc((void**)&F0, sizeof(double) * ndd));
    CHECK_HIP(hipMalloc((void**)&xg, sizeof(double) * 2 * nbd));
    CHECK_HIP(hipMalloc((void**)&flag, 2 * sizeof(int)));
    CHECK_HIP(hipMemset(flag, 0, 2 * sizeof(int)));
    CHECK_HIP(hipMemcpy(Z, h + 2 * nbd + D + ndd, sizeof(double) * nbd, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(F0, h + 3 * nbd + D + ndd, sizeof(double) * ndd, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep)
        CHECK_ABI(gsmvi_gsm_factor_update_sharded_f64(ctx, st, comm, D, Bl, Z, D, X, D, G, D, mu0, F0, D, rec, mu, S, D, flag,
                                                      flag + 1));
    CHECK_HIP(hipStreamSynchronize(st));
    int hflag[2];
    CHECK_HIP(hipMemcpy(hflag, flag, sizeof hflag, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + D + ndd, mu, sizeof(double) * D, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + 2 * D + ndd, S, sizeof(double) * ndd, hipMemcpyDeviceToHost));
    out[2 * (D + ndd)] = (double)(hflag[0] + 1000 * hflag[1]);

    /* BaM update through the communicator: the (x_b, g_b) rows are all-gathered into xg */
    CHECK_ABI(gsmvi_bam_update_sharded_f64(ctx, st, comm, D, Bl, X, D, G, D, mu0, S0, D, 2.0, 0.0, xg, mu, S, D, flag));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(out + 2 * (D + ndd) + 1, mu, sizeof(double) * D, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + 2 * (D + ndd) + 1 + D, S, sizeof(double) * ndd, hipMemcpyDeviceToHost));

    /* factor-form BaM update through the communicator (round 4): Z replicated, the (x_b, g_b) rows all-gathered into xg */
    {
        double* o4 = out + 3 * (D + ndd) + 1;
        memset(o4, 0, sizeof(double) * (D + ndd + 1));
        if (2 * B <= D) {
            CHECK_HIP(hipMemset(flag, 0, 2 * sizeof(int)));
            CHECK_ABI(gsmvi_bam_factor_update_sharded_f64(ctx, st, comm, D, Bl, Z, D, X, D, G, D, mu0, F0, D, 2.0, xg, mu, S, D,
                                                          flag, flag + 1));
            CHECK_HIP(hipStreamSynchronize(st));
            CHECK_HIP(hipMemcpy(hflag, flag, sizeof hflag, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(o4, mu, sizeof(double) * D, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(o4 + D, S, sizeof(double) * ndd, hipMemcpyDeviceToHost));
            o4[D + ndd] = (double)(hflag[0] + 1000 * hflag[1]);
        } else if (gsmvi_bam_factor_update_sharded_f64(ctx, st, comm, D, Bl, Z, D, X, D, G, D, mu0, F0, D, 2.0, xg, mu, S, D,
                                                       flag, flag + 1) != GSMVI_ERR_UNSUPPORTED) {
            return 6;                       /* 2B > D must be refused before anything is enqueued */
        }
    }

    char name[4096];
    snprintf(name, sizeof name, "%s.%d", argv[2], rank);
    f = fopen(name, "wb");
    if (!f) return 1;
    fwrite(out, sizeof(double), nout, f);
    fclose(f);
    CHECK_NCCL(ncclCommDestroy(comm));
    CHECK_ABI(gsmvi_destroy(ctx));
    return 0;
}
