/* Host-only walk through the C ABI for the sanitizer build (libgsmvi_hip_asan.so: AddressSanitizer + UBSan on the host
 * code of every translation unit).  No GPU needed: every call either is pure host logic or must be rejected by the
 * argument validation BEFORE any HIP call.  Exit code 0 = every status was the expected one (and no sanitizer report,
 * which would abort the process). */
#include <stdio.h>
#include <string.h>
#include "gsmvi_hip.h"

#define EXPECT(call, want) do { int s_ = (call); if (s_ != (want)) { fprintf(stderr, "%s -> %d (%s), wanted %d\n", #call, s_, gsmvi_last_error(), (want)); return 1; } } while (0)

int main(void) {
    if (gsmvi_abi_version() != GSMVI_ABI_VERSION) return 1;
    for (int s = -1; s < 8; ++s)
        if (!gsmvi_status_string(s) || !strlen(gsmvi_status_string(s))) return 1;
    /* workspace arithmetic over a range of sizes, including the largest supported shapes */
    size_t prev = 0;
    for (int D = 1; D <= 16384; D = D * 3 + 1)
        for (int B = 1; B <= 256; B *= 2) {
            const size_t w = gsmvi_workspace_bytes(D, B);
            if (w == 0) return 1;
            prev += w & 1;
        }
    if (gsmvi_workspace_bytes(0, 4) != 0 || gsmvi_workspace_bytes(4, -1) != 0) return 1;
    if (gsmvi_gsm_record_len(5) != 16 || gsmvi_gsm_record_len(1024) != 3072) return 1;
    double buf[64];
    int flag = 0;
    gsmvi_ctx* ctx = NULL;
    EXPECT(gsmvi_create(NULL, 0, 8, 2), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_create(&ctx, 0, 0, 2), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_destroy(NULL), GSMVI_OK);
    EXPECT(gsmvi_set_tuning(NULL, "no_fast", 1), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_gsm_update_f64(NULL, NULL, 8, 2, buf, 8, buf, 8, buf, buf, 8, buf, buf, 8), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_gsm_update_general_f64(NULL, NULL, 8, 2, buf, 8, buf, 8, buf, buf, 8, buf, buf, 8), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_gsm_local_stage_f64(NULL, NULL, 8, 2, buf, 8, buf, 8, buf, buf, 8, buf, 24), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_gsm_apply_f64(NULL, NULL, 8, 2, buf, 24, buf, buf, 8, buf, buf, 8), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_gsm_update_sharded_f64(NULL, NULL, NULL, 8, 2, buf, 8, buf, 8, buf, buf, 8, buf, buf, buf, 8), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_gsm_rows_stage_f64(NULL, NULL, 8, 2, 4, buf, 8, buf, 8, buf, 4), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_gsm_records_f64(NULL, NULL, 8, 2, buf, 8, buf, 8, buf, buf, buf, 24), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_gsm_apply_rows_f64(NULL, NULL, 8, 2, 0, 4, buf, 24, buf, buf, 8, buf, buf, 8), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_gsm_factor_update_f64(NULL, NULL, 8, 2, buf, 8, buf, 8, buf, 8, buf, buf, 8, buf, buf, 8, &flag, NULL), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_gsm_factor_local_stage_f64(NULL, NULL, 8, 2, buf, 8, buf, 8, buf, 8, buf, buf, 8, buf, 24), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_gsm_factor_apply_f64(NULL, NULL, 8, 2, buf, 8, buf, 24, buf, buf, 8, buf, buf, 8, &flag, NULL), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_gaussian_score_f64(NULL, NULL, 8, 2, buf, 8, buf, buf, 8, buf, 8), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_potrf_f64(NULL, NULL, 8, buf, 8, buf, 8, &flag), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_sample_f64(NULL, NULL, 8, 2, buf, 8, buf, buf, 8, buf, 8), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_randn_f64(NULL, NULL, 1, 0, 16, buf, NULL), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_commit_f64(NULL, NULL, 8, &flag, buf, buf, 8, buf, buf, 8, NULL), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_gram_f64(NULL, NULL, 8, buf, 8, buf, 8), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_whiten_rows_f64(NULL, NULL, 8, 2, buf, 8, buf, 8, buf, buf, 8, buf), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_bam_update_f64(NULL, NULL, 8, 2, buf, 8, buf, 8, buf, buf, 8, 1.0, 0.0, buf, buf, 8, &flag), GSMVI_ERR_BAD_ARG);
    EXPECT(gsmvi_set_profiling(NULL, 1), GSMVI_ERR_BAD_ARG);
    if (!strlen(gsmvi_last_error())) return 1;
    /* with no device, create must fail cleanly (status, message), with one it must succeed and be destroyable */
    int n = 0;
    int st = gsmvi_device_count(&n);
    if (st == GSMVI_OK) {
        EXPECT(gsmvi_create(&ctx, n, 8, 2), GSMVI_ERR_BAD_ARG);          /* device index out of range */
        EXPECT(gsmvi_create(&ctx, 0, 64, 8), GSMVI_OK);
        EXPECT(gsmvi_set_tuning(ctx, "no_such_knob", 1), GSMVI_ERR_BAD_ARG);
        EXPECT(gsmvi_gsm_update_f64(ctx, NULL, 128, 8, buf, 128, buf, 128, buf, buf, 128, buf, buf + 1, 128), GSMVI_ERR_WORKSPACE);
        EXPECT(gsmvi_destroy(ctx), GSMVI_OK);
    } else {
        if (st != GSMVI_ERR_NO_DEVICE || n != 0) return 1;
        EXPECT(gsmvi_create(&ctx, 0, 8, 2), GSMVI_ERR_NO_DEVICE);
    }
    printf("abi_hostcheck ok (%zu)\n", prev);
    return 0;
}
