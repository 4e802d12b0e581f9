// Whitened draws Z ~ N(0, I) on the device: counter-based Philox4x32-10 + Box-Muller, fp64.
//
// Replaces the z-stream behind np.random.multivariate_normal (gsmvi/gsm_numpy.py:105,116; the JAX twins
// re-seed from a sub-key every iteration, gsmvi/gsm.py:117-119, gsmvi/bam.py:191-193).  The stream is a pure
// function of (seed, call, element index): element pair p = i / 2 of call c is the Philox block with
//   counter = (p lo, p hi, c lo, c hi),  key = (seed lo, seed hi),
// its four 32-bit outputs make two 53-bit uniforms u1, u2 in (0, 1) and the pair is
//   z[2p] = sqrt(-2 ln u1) cos(2 pi u2),  z[2p+1] = sqrt(-2 ln u1) sin(2 pi u2).
// No state lives on the device, so every rank of a sharded fit draws the identical Z from the same key,
// a draw can be replayed, and the call is graph-capturable.  The tests hold a CPU restatement (integer part
// bit-exact, pinned to the Random123 known-answer vectors).
#include "gsmvi_common.h"
#include "gsmvi_ctx.h"
#include "../../include/gsmvi_hip.h"

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 53-bit uniform in (0, 1): ((a >> 5) 2^26 + (b >> 6) + 1/2) 2^-53 -- never 0, never 1
__device__ __forceinline__ double u53(unsigned a, unsigned b) {
    const unsigned long long m = ((unsigned long long)(a >> 5) << 26) | (unsigned long long)(b >> 6);
    return ((double)m + 0.5) * (1.0 / 9007199254740992.0);
}

// One thread per element pair; `raw` (may be NULL) receives the four Philox words of every pair (tests).
// blockIdx.y = c: draw number call + c goes to out + c n (several iterations' draws from one launch).  call_in (may be
// NULL): a device word added to `call` -- with call_out = the OTHER word of a ping-pong pair receiving *call_in + gridDim.y,
// a captured launch advances through the stream on every replay of its graph (nobody in this launch reads call_out).
__global__ __launch_bounds__(256) void k_randn(unsigned long long seed, unsigned long long call, long long n,
                                               double* __restrict__ out, unsigned* __restrict__ raw,
                                               const unsigned long long* __restrict__ call_in,
                                               unsigned long long* __restrict__ call_out) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long base = call_in ? *call_in : 0ull;
    if (call_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *call_out = base + gridDim.y;
    call += base + blockIdx.y;
    out += (size_t)blockIdx.y * n;
    if (2 * p >= n) return;
    unsigned w[4];
    philox4x32_10((unsigned)p, (unsigned)((unsigned long long)p >> 32), (unsigned)call, (unsigned)(call >> 32),
                  (unsigned)seed, (unsigned)(seed >> 32), w);
    if (raw) {
#pragma unroll
        for (int k = 0; k < 4; ++k) raw[4 * p + k] = w[k];
    }
    const double u1 = u53(w[0], w[1]), u2 = u53(w[2], w[3]);
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    out[2 * p] = r * c;
    if (2 * p + 1 < n) out[2 * p + 1] = r * s;
}

static int randn_launch(const char* fn, gsmvi_ctx* ctx, void* stream, uint64_t seed, uint64_t call, int ncalls, int64_t n,
                        double* out, uint32_t* raw, const uint64_t* call_in, uint64_t* call_out) {
    if (!ctx) { gsmvi_set_error("%s: %s", fn, "ctx is NULL"); return GSMVI_ERR_BAD_ARG; }
    if (n < 0 || ncalls < 1 || ncalls > 65535 || (n > 0 && !out) || (call_out && call_out == call_in)) {
        gsmvi_set_error("%s: %s", fn, "bad size, out is NULL, or call_out aliases call_in");
        return GSMVI_ERR_BAD_ARG;
    }
    if (n == 0) return GSMVI_OK;
    const long long pairs = (n + 1) / 2;
    hipLaunchKernelGGL(k_randn, dim3((unsigned)((pairs + 255) / 256), (unsigned)ncalls), dim3(256), 0,
                       static_cast<hipStream_t>(stream), (unsigned long long)seed, (unsigned long long)call, (long long)n, out,
                       raw, reinterpret_cast<const unsigned long long*>(call_in),
                       reinterpret_cast<unsigned long long*>(call_out));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gsmvi_set_error("launch of %s failed: %s", "k_randn", hipGetErrorString(e));
        return GSMVI_ERR_HIP;
    }
    return GSMVI_OK;
}

extern "C" int gsmvi_randn_f64(gsmvi_ctx* ctx, void* stream, uint64_t seed, uint64_t call, int64_t n, double* out,
                               uint32_t* raw) {
    return randn_launch("gsmvi_randn_f64", ctx, stream, seed, call, 1, n, out, raw, nullptr, nullptr);
}

extern "C" int gsmvi_randn_batch_f64(gsmvi_ctx* ctx, void* stream, uint64_t seed, uint64_t call0, int ncalls, int64_t n,
                                     double* out, const uint64_t* call_in_dev, uint64_t* call_out_dev) {
    return randn_launch("gsmvi_randn_batch_f64", ctx, stream, seed, call0, ncalls, n, out, nullptr, call_in_dev,
                        call_out_dev);
}
