/*
 * gsmvi_hip_debug.h -- diagnostic entry points (NOT part of the drop-in boundary of gsmvi_hip.h).  They are exported by
 * libgsmvi_hip_debug.so only -- the same objects linked with csrc/exports_debug.map -- never by the product library
 * libgsmvi_hip.so (its export list hides them and --gc-sections drops their host code).  Used by scripts/ (in-kernel
 * timelines, soaks); select the debug build with GSMVI_HIP_DEBUG_LIB=1 before importing gsmvi_amd.
 */
#ifndef GSMVI_HIP_DEBUG_H
#define GSMVI_HIP_DEBUG_H

#include "gsmvi_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* Read back in-kernel s_memrealtime stamps: with the tuning knob "timeline" = 1 the first n words of the stamp buffer
 * ([kernel slot 0..3][512 workgroups][8 words]; workgroups beyond 512 write nothing), otherwise (knob "cov_dbg" bits
 * 16 / 128) the first n words of the panel-partial slab. */
int gsmvi_debug_read_stamps(gsmvi_ctx* ctx, unsigned long long* out, int n);
/* Read back a slice of the context workspace (region 0 panel slabs, 1 finished panels, 2 small matrices). */
int gsmvi_debug_read_workspace(gsmvi_ctx* ctx, int region, size_t offset, double* out, size_t n);
/* Device address of the region's base (0: panel partials, 1: finished panels, 2: small matrices), for in-place views. */
int gsmvi_debug_workspace_ptr(gsmvi_ctx* ctx, int region, double** out);
/* The one-workgroup n x n Cholesky kernels of the factor path's 2B x 2B chain (64 < n <= 128) on caller data: A (n x n, upper
 * triangle read) -> R (upper), and with_inverse != 0 also W = R^-T (lower) with the rank-revealing rule of the Gram matrix.
 * For soak / determinism scripts. */
int gsmvi_debug_chol128(void* stream, int n, int with_inverse, const double* A, double* R, double* W, int* info_dev);

/* Calibration for bench.py: a plain streaming copy of n doubles on `stream` (16 bytes per lane, non-temporal) -- the rate a
 * kernel that only moves bytes reaches on this box, the yardstick beside the 8 TB/s specification. */
int gsmvi_debug_stream_copy_f64(void* stream, double* dst, const double* src, size_t n);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* GSMVI_HIP_DEBUG_H */
