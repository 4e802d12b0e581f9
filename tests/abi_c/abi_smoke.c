/* Plain-C consumer of the C ABI (include/gsmvi_hip.h): no C++, no torch -- raw HIP device pointers and sizes only.
 * Reads D, B and the inputs of one gsm_update (gsmvi/gsm_numpy.py:27-55) from a binary file written by the test
 * harness, runs gsmvi_gsm_update_f64 and gsmvi_potrf_f64 on stream 0, writes mu, S, R and the Cholesky flag back.
 *   abi_smoke <in.bin> <out.bin>
 * in.bin : int32 D, int32 B, then doubles X[B*D], G[B*D], mu0[D], S0[D*D]
 * out.bin: doubles mu[D], S[D*D], R[D*D], then int32 info */
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime_api.h>
#include "gsmvi_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_ABI(x) do { int s_ = (x); if (s_ != GSMVI_OK) { fprintf(stderr, "%s -> %s (%s)\n", #x, gsmvi_status_string(s_), gsmvi_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc != 3) return 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    int D, B;
    if (fread(&D, 4, 1, f) != 1 || fread(&B, 4, 1, f) != 1) return 1;
    const size_t nbd = (size_t)B * D, ndd = (size_t)D * D;
    double* h = (double*)malloc(sizeof(double) * (2 * nbd + D + ndd));
    if (fread(h, sizeof(double), 2 * nbd + D + ndd, f) != 2 * nbd + D + ndd) return 1;
    fclose(f);

    if (gsmvi_abi_version() != GSMVI_ABI_VERSION) return 4;
    gsmvi_ctx* ctx = NULL;
    CHECK_ABI(gsmvi_create(&ctx, 0, D, B));
    double *X, *G, *mu0, *S0, *mu, *S, *R;
    int* info;
    CHECK_HIP(hipMalloc((void**)&X, sizeof(double) * nbd));
    CHECK_HIP(hipMalloc((void**)&G, sizeof(double) * nbd));
    CHECK_HIP(hipMalloc((void**)&mu0, sizeof(double) * D));
    CHECK_HIP(hipMalloc((void**)&S0, sizeof(double) * ndd));
    CHECK_HIP(hipMalloc((void**)&mu, sizeof(double) * D));
    CHECK_HIP(hipMalloc((void**)&S, sizeof(double) * ndd));
    CHECK_HIP(hipMalloc((void**)&R, sizeof(double) * ndd));
    CHECK_HIP(hipMalloc((void**)&info, sizeof(int)));
    CHECK_HIP(hipMemcpy(X, h, sizeof(double) * nbd, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(G, h + nbd, sizeof(double) * nbd, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(mu0, h + 2 * nbd, sizeof(double) * D, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(S0, h + 2 * nbd + D, sizeof(double) * ndd, hipMemcpyHostToDevice));

    CHECK_ABI(gsmvi_gsm_update_f64(ctx, NULL, D, B, X, D, G, D, mu0, S0, D, mu, S, D));
    CHECK_ABI(gsmvi_potrf_f64(ctx, NULL, D, S, D, R, D, info));
    /* argument errors come back as status codes, never as exceptions or aborts */
    if (gsmvi_gsm_update_f64(ctx, NULL, D, B, X, D - 1, G, D, mu0, S0, D, mu, S, D) != GSMVI_ERR_BAD_ARG) return 5;
    if (gsmvi_gsm_update_f64(ctx, NULL, D, B, X, D, G, D, mu0, S0, D, mu, S0, D) != GSMVI_ERR_BAD_ARG) return 6;
    CHECK_HIP(hipDeviceSynchronize());

    double* out = (double*)malloc(sizeof(double) * (D + 2 * ndd));
    int hinfo = -1;
    CHECK_HIP(hipMemcpy(out, mu, sizeof(double) * D, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + D, S, sizeof(double) * ndd, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + D + ndd, R, sizeof(double) * ndd, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost));
    f = fopen(argv[2], "wb");
    if (!f) return 1;
    fwrite(out, sizeof(double), D + 2 * ndd, f);
    fwrite(&hinfo, 4, 1, f);
    fclose(f);
    CHECK_ABI(gsmvi_destroy(ctx));
    return 0;
}
