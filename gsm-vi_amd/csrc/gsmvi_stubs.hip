// Entry points whose kernels have not landed yet fail loudly (never a CPU fallback).
#include "../../include/gsmvi_hip.h"
#include "gsmvi_ctx.h"
int gsmvi_bam_impl(gsmvi_ctx*, hipStream_t, int, int, const double*, int, const double*, int, const double*,
                   const double*, int, double, double, double*, double*, int, int*) {
    gsmvi_set_error("%s%s", "gsmvi_bam_update_f64: not implemented yet", "");
    return GSMVI_ERR_UNSUPPORTED;
}
