// Where do the 4.2 us of one Newton-Schulz launch (gsmvi_bam_small.hip, n = 128: 64 / 128 workgroups, one 16 x 16 block each) go?
// A chain of 15 x (M = Z Y; Y' = c Y T, Z' = c T Z) launches, timed eagerly and replayed from a hipGraph, in variants:
//   0 shipped form (coefficient word read first, early return, then the operand loads)      1 no coefficient read (c2 as argument)
//   2 operand loads issued BEFORE the coefficient read       3 as 0 with 512 threads (K over 8 waves)     4 as 1 with 512 threads
//   5 empty kernels (launch floor)       6 as 1 without the stores (load + MFMA only)       7 as 1, loads only (no MFMA, one store)
//   8 / 9 as 1 with nontemporal / agent-scope (write-through) stores      10 / 11 one block per wave, whole K, no LDS reduction
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/nsbench scripts/nsbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double v4d __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int NW, int VAR>
__device__ __forceinline__ void block(const double* __restrict__ A, const double* __restrict__ Bm, double* __restrict__ Out, int blk,
                                      int nb, int nk, double c2, double scale, int ld, const double* coef, int k) {
    __shared__ double red[NW * 256];
    const int i0 = (blk / nb) * 16, j0 = (blk % nb) * 16;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, cc = l & 15, ks = l >> 4;
    constexpr int NU = (36 + NW - 1) / NW;                    // k-steps per wave at ld = 144
    double av[NU], bv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int st = w + NW * u, kk = 4 * st + ks, kc = kk < ld ? kk : ld - 1;
        av[u] = A[(size_t)(i0 + cc) * ld + kc];
        bv[u] = Bm[(size_t)kc * ld + j0 + cc];
    }
    if (VAR == 2) {                                           // coefficient read behind the operand loads
        const double ks_ = coef[40], fl = coef[42];
        c2 = coef[k];
        scale = MODE == 0 ? 1.0 : sqrt(c2);
        if ((double)k >= ks_ || fl != 0.0) return;
    }
    v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int st = w + NW * u, kk = 4 * st + ks, kc = kk < ld ? kk : ld - 1;
        double a = MODE == 2 ? ((kc == i0 + cc ? 1.5 : 0.0) - 0.5 * c2 * av[u]) : av[u];
        double b = MODE == 1 ? ((kc == j0 + cc ? 1.5 : 0.0) - 0.5 * c2 * bv[u]) : bv[u];
        if (st >= nk) { a = 0.0; b = 0.0; }
        if (VAR == 7) { acc0[0] += a * b; continue; }
        if (u & 1) acc1 = MFMA(a, b, acc1); else acc0 = MFMA(a, b, acc0);
    }
    if (VAR == 6) { if (acc0[0] + acc1[0] == 12345.678) Out[0] = 1.0; return; }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w * 256 + (ks + 4 * r) * 16 + cc] = acc0[r] + acc1[r];
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 256) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < NW; ++q) v += red[q * 256 + t];
        double* o = Out + (size_t)(i0 + (t >> 4)) * ld + j0 + (t & 15);
        if (VAR == 8) __builtin_nontemporal_store(scale * v, o);
        else if (VAR == 9) __hip_atomic_store(o, scale * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *o = scale * v;
    }
}
// one 16 x 16 block per WAVE, whole K in that wave (36 loads per operand in flight), no LDS reduction, no barrier
template <int MODE, int VAR>
__device__ __forceinline__ void block_w(const double* __restrict__ A, const double* __restrict__ Bm, double* __restrict__ Out, int blk,
                                        int nb, int nk, double c2, double scale, int ld) {
    const int i0 = (blk / nb) * 16, j0 = (blk % nb) * 16;
    const int l = threadIdx.x & 63, cc = l & 15, ks = l >> 4;
    double av[36], bv[36];
#pragma unroll
    for (int u = 0; u < 36; ++u) {
        const int kc = 4 * u + ks;
        av[u] = A[(size_t)(i0 + cc) * ld + kc];
        bv[u] = Bm[(size_t)kc * ld + j0 + cc];
    }
    v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 36; ++u) {
        const int kc = 4 * u + ks;
        double a = MODE == 2 ? ((kc == i0 + cc ? 1.5 : 0.0) - 0.5 * c2 * av[u]) : av[u];
        double b = MODE == 1 ? ((kc == j0 + cc ? 1.5 : 0.0) - 0.5 * c2 * bv[u]) : bv[u];
        if (u >= nk) { a = 0.0; b = 0.0; }
        if (u & 1) acc1 = MFMA(a, b, acc1); else acc0 = MFMA(a, b, acc0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double* o = Out + (size_t)(i0 + ks + 4 * r) * ld + j0 + cc;
        if (VAR == 11) __builtin_nontemporal_store(scale * (acc0[r] + acc1[r]), o);
        else *o = scale * (acc0[r] + acc1[r]);
    }
}
template <int VAR>
__global__ __launch_bounds__(256) void k_zy_w(int n, int ld, int k, const double* Ya, const double* Za, const double* Yb,
                                              const double* Zb, double* Mm, const double* coef) {
    if ((double)k >= coef[40] || coef[42] != 0.0) return;
    const int nb = (n + 15) >> 4, blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= nb * nb) return;
    block_w<0, VAR>((k & 1) ? Zb : Za, (k & 1) ? Yb : Ya, Mm, blk, nb, (n + 3) >> 2, 0.0, 1.0, ld);
}
template <int VAR>
__global__ __launch_bounds__(256) void k_step_w(int n, int ld, int k, double* Ya, double* Za, double* Yb, double* Zb,
                                                const double* Mm, const double* coef) {
    if ((double)k >= coef[40] || coef[42] != 0.0) return;
    const double c2 = coef[k], c = sqrt(c2);
    const int nb = (n + 15) >> 4, nk = (n + 3) >> 2, blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= 2 * nb * nb) return;
    if (blk < nb * nb) block_w<1, VAR>((k & 1) ? Yb : Ya, Mm, (k & 1) ? Ya : Yb, blk, nb, nk, c2, c, ld);
    else block_w<2, VAR>(Mm, (k & 1) ? Zb : Za, (k & 1) ? Za : Zb, blk - nb * nb, nb, nk, c2, c, ld);
}
template <int NW, int VAR>
__global__ __launch_bounds__(NW * 64) void k_zy(int n, int ld, int k, const double* Ya, const double* Za, const double* Yb,
                                                const double* Zb, double* Mm, const double* coef, double c2arg) {
    if (VAR == 5) return;
    if (VAR == 0 || VAR == 3) { if ((double)k >= coef[40] || coef[42] != 0.0) return; }
    const double* Y = (k & 1) ? Yb : Ya;
    const double* Z = (k & 1) ? Zb : Za;
    block<0, NW, VAR>(Z, Y, Mm, blockIdx.x, (n + 15) >> 4, (n + 3) >> 2, 0.0, 1.0, ld, coef, k);
}
template <int NW, int VAR>
__global__ __launch_bounds__(NW * 64) void k_step(int n, int ld, int k, double* Ya, double* Za, double* Yb, double* Zb,
                                                  const double* Mm, const double* coef, double c2arg) {
    if (VAR == 5) return;
    double c2 = c2arg;
    if (VAR == 0 || VAR == 3) { if ((double)k >= coef[40] || coef[42] != 0.0) return; c2 = coef[k]; }
    const double c = sqrt(c2);
    const double* Yi = (k & 1) ? Yb : Ya;
    const double* Zi = (k & 1) ? Zb : Za;
    double* Yo = (k & 1) ? Ya : Yb;
    double* Zo = (k & 1) ? Za : Zb;
    const int nb = (n + 15) >> 4, nk = (n + 3) >> 2;
    if ((int)blockIdx.x < nb * nb) block<1, NW, VAR>(Yi, Mm, Yo, blockIdx.x, nb, nk, c2, c, ld, coef, k);
    else block<2, NW, VAR>(Mm, Zi, Zo, blockIdx.x - nb * nb, nb, nk, c2, c, ld, coef, k);
}
template <int NW, int VAR>
static void chain(hipStream_t st, int n, int ld, int steps, double* Ya, double* Za, double* Yb, double* Zb, double* Mm, double* coef) {
    const int nb = (n + 15) / 16;
    if (VAR == 10 || VAR == 11) {
        for (int k = 0; k < steps; ++k) {
            hipLaunchKernelGGL((k_zy_w<VAR>), dim3((nb * nb + 3) / 4), dim3(256), 0, st, n, ld, k, Ya, Za, Yb, Zb, Mm, coef);
            hipLaunchKernelGGL((k_step_w<VAR>), dim3((2 * nb * nb + 3) / 4), dim3(256), 0, st, n, ld, k, Ya, Za, Yb, Zb, Mm, coef);
        }
        return;
    }
    for (int k = 0; k < steps; ++k) {
        hipLaunchKernelGGL((k_zy<NW, VAR>), dim3(nb * nb), dim3(NW * 64), 0, st, n, ld, k, Ya, Za, Yb, Zb, Mm, coef, 1.0);
        hipLaunchKernelGGL((k_step<NW, VAR>), dim3(2 * nb * nb), dim3(NW * 64), 0, st, n, ld, k, Ya, Za, Yb, Zb, Mm, coef, 1.0);
    }
}
template <int NW, int VAR>
static int run(const char* name, int n, int ld, int steps, double* Ya, double* Za, double* Yb, double* Zb, double* Mm, double* coef,
               const std::vector<double>& y0, const std::vector<double>& z0) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto reset = [&]() {
        hipMemcpyAsync(Ya, y0.data(), y0.size() * 8, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(Za, z0.data(), z0.size() * 8, hipMemcpyHostToDevice, st);
    };
    std::vector<float> te, tg;
    for (int rep = 0; rep < 40; ++rep) {
        reset();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        chain<NW, VAR>(st, n, ld, steps, Ya, Za, Yb, Zb, Mm, coef);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 5) te.push_back(ms * 1e3f);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    chain<NW, VAR>(st, n, ld, steps, Ya, Za, Yb, Zb, Mm, coef);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 40; ++rep) {
        reset();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 5) tg.push_back(ms * 1e3f);
    }
    std::sort(te.begin(), te.end()); std::sort(tg.begin(), tg.end());
    std::vector<double> h(ld * ld);
    CK(hipMemcpy(h.data(), (steps & 1) ? Yb : Ya, h.size() * 8, hipMemcpyDeviceToHost));
    printf("%-46s n=%d: eager median %7.1f us (%.2f per launch)   graph median %7.1f us (%.2f per launch)   Y[0][0]=%.6f\n", name, n,
           te[te.size() / 2], te[te.size() / 2] / (2 * steps), tg[tg.size() / 2], tg[tg.size() / 2] / (2 * steps), h[0]);
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
    return 0;
}
int main() {
    for (int n : {128, 64}) {
        const int ld = 144, steps = 15;
        const size_t LL = (size_t)ld * ld;
        double *Ya, *Za, *Yb, *Zb, *Mm, *coef;
        CK(hipMalloc((void**)&Ya, LL * 8)); CK(hipMalloc((void**)&Za, LL * 8)); CK(hipMalloc((void**)&Yb, LL * 8));
        CK(hipMalloc((void**)&Zb, LL * 8)); CK(hipMalloc((void**)&Mm, LL * 8)); CK(hipMalloc((void**)&coef, 64 * 8));
        CK(hipMemset(Yb, 0, LL * 8)); CK(hipMemset(Zb, 0, LL * 8)); CK(hipMemset(Mm, 0, LL * 8));
        std::vector<double> y0(LL, 0.0), z0(LL, 0.0), cf(64, 1.0);
        for (int i = 0; i < n; ++i) { y0[(size_t)i * ld + i] = 0.2 + 0.8 * i / n; z0[(size_t)i * ld + i] = 1.0; }   // diagonal test problem
        cf[40] = 15.0; cf[41] = 1.0; cf[42] = 0.0;
        CK(hipMemcpy(coef, cf.data(), 64 * 8, hipMemcpyHostToDevice));
#define RUN(NW, VAR, NAME) if (run<NW, VAR>(NAME, n, ld, steps, Ya, Za, Yb, Zb, Mm, coef, y0, z0)) return 1
        RUN(4, 0, "0 shipped: coef read, return, loads");
        RUN(4, 1, "1 no coef read (argument)");
        RUN(4, 2, "2 loads issued before the coef read");
        RUN(8, 3, "3 shipped form, 512 threads");
        RUN(8, 4, "4 no coef read, 512 threads");
        RUN(4, 5, "5 empty kernels");
        RUN(4, 6, "6 no coef read, no stores");
        RUN(4, 7, "7 no coef read, loads + one store, no MFMA");
        RUN(4, 8, "8 as 1, nontemporal stores");
        RUN(4, 9, "9 as 1, agent-scope atomic (sc1) stores");
        RUN(4, 10, "10 one block per wave, whole K, no LDS reduce");
        RUN(4, 11, "11 as 10, nontemporal stores");
        hipFree(Ya); hipFree(Za); hipFree(Yb); hipFree(Zb); hipFree(Mm); hipFree(coef);
    }
    return 0;
}
