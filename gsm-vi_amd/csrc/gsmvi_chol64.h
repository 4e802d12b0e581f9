// Shared pieces of the in-LDS 64 x 64 Cholesky (gsmvi_chol64b.h: chol64_blk): the lane broadcast and the semi-definite rule.
// (Until round 4 this header also held the barrier-per-pivot factorisation chol64_rows_s and its helper-wave companions, the
// last users of which -- k_bam_chol_out -- moved to chol64_blk; they were deleted with it.)
#pragma once
#include "gsmvi_common.h"

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)(u & 0xffffffffu), lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), lane);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// SEMIDEF = true (chol64_blk, chol128w_body, k_chol128) factors a positive SEMI-definite matrix (a Gram matrix G = Rt Rt^T whose
// rows may be linearly dependent).  The CALLER has lowered every diagonal entry by its rounding floor, G_pp <- G_pp (1 -
// GSMVI_DEP_TOL) (a relative perturbation of 64 eps, scale-free per row, so rows of tiny norm are fine); a pivot <= 0 then means
// "not above the rounding floor of its row": it cannot be told from zero in fp64 -- and a tiny POSITIVE noise pivot is
// as harmful as a negative one (1 / sqrt(1e-34) in the factor: scripts/dbg_dep128.py), hence the shift instead of a
// plain sign test.  (Comparing against a per-pivot floor read from LDS inside the loop put that read on the pivot chain:
// 12.2 -> 14.8 us per 64 x 64 block.)  Such a row is DEPENDENT: its row of the factor and its diagonal come out as
// ZERO, the pivot is skipped in the elimination, no failure is reported (R^T R still equals G to rounding).  allow_dep
// (block-uniform) = false turns the dependent verdict into a FAILURE instead: the callers pass "every diagonal entry is of
// moderate size" (max_p G_pp < 2^32), because dropping a row perturbs the represented matrix by up to sqrt(GSMVI_DEP_TOL) |row|
// -- harmless for whitened draws (|z|^2 ~ D), meaningless for |z| ~ 1e10 (fixture G4).  NaN / inf pivots fail in every mode.
// SEMIDEF = false is the plain positive-definite test (pivot <= 0 fails).
#define GSMVI_DEP_TOL 1.4210854715202004e-14        /* 64 eps */
