// Correctness + phase timing of the blocked in-LDS Cholesky chol64_blk (gsm-vi_amd/csrc/gsmvi_chol64b.h) on one workgroup.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o scripts/chol64b_test scripts/chol64b_test.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__device__ unsigned long long g_stamp[16];
__device__ unsigned long long g_cyc[2];
#define CHOLB_STAMP(k) do { if (threadIdx.x == 0) g_stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ unsigned long long g_ps[2][16];
#define CHOLB_PSTAMP(h, i) ((void)0)
#include "../gsm-vi_amd/csrc/gsmvi_chol64b.h"

template <bool SEMIDEF, bool AUG>
__global__ __launch_bounds__(512) void k(int n, const double* A, double* R, double* W, int* fail) {
    constexpr int ES = AUG ? 146 : 82;
    __shared__ __attribute__((aligned(16))) double E[64 * ES];
    __shared__ __attribute__((aligned(16))) double scr[CHOLB_SCRATCH_DOUBLES(AUG)];
    __shared__ int sf;
    for (int e = threadIdx.x; e < 64 * 64; e += 512) {
        const int i = e >> 6, q = e & 63;
        double v = (i < n && q < n && q >= i) ? A[i * n + q] : (i == q ? 1.0 : 0.0);
        if (SEMIDEF && i == q && i < n) v -= GSMVI_DEP_TOL * v;
        E[i * ES + q] = v;
    }
    __syncthreads();
    CHOLB_STAMP(0);
    if (threadIdx.x == 0) g_cyc[0] = __builtin_amdgcn_s_memtime();
    chol64_blk<ES, SEMIDEF, AUG>(E, scr, n, &sf, true);
    CHOLB_STAMP(15);
    if (threadIdx.x == 0) g_cyc[1] = __builtin_amdgcn_s_memtime();
    for (int e = threadIdx.x; e < 64 * 64; e += 512) {
        const int i = e >> 6, q = e & 63;
        R[e] = E[i * ES + q];
        if (AUG) W[e] = E[i * ES + 64 + q];
    }
    if (threadIdx.x == 0) *fail = sf;
}

static void host_chol(int n, const std::vector<double>& A, std::vector<double>& R) {   // upper, R^T R = A
    R.assign(n * n, 0.0);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i <= j; ++i) {
            long double s = A[i * n + j];
            for (int kk = 0; kk < i; ++kk) s -= (long double)R[kk * n + i] * R[kk * n + j];
            R[i * n + j] = (i == j) ? (double)sqrtl(s) : (double)(s / R[i * n + i]);
        }
}

template <bool SEMIDEF, bool AUG>
static int run(const char* name, int n, const std::vector<double>& A, int expect_fail, bool check) {
    double *dA, *dR, *dW; int* df;
    hipMalloc(&dA, n * n * 8); hipMalloc(&dR, 64 * 64 * 8); hipMalloc(&dW, 64 * 64 * 8); hipMalloc(&df, 4);
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<SEMIDEF, AUG>), dim3(1), dim3(512), 0, 0, n, dA, dR, dW, df);
    hipDeviceSynchronize();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("%s: HIP error %s\n", name, hipGetErrorString(e)); return 1; }
    unsigned long long st[16]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof st);
    unsigned long long cy[2]; hipMemcpyFromSymbol(cy, HIP_SYMBOL(g_cyc), sizeof cy);
    std::vector<double> R(64 * 64), W(64 * 64); int fail = -1;
    hipMemcpy(R.data(), dR, 64 * 64 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(W.data(), dW, 64 * 64 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(&fail, df, 4, hipMemcpyDeviceToHost);
    int bad = 0;
    double amax = 0, erec = 0, eref = 0, einv = 0, elow = 0;
    if (fail != expect_fail) { bad = 1; }
    if (check && fail == 0) {
        std::vector<double> Rh;
        if (!SEMIDEF) host_chol(n, A, Rh);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) amax = fmax(amax, fabs(A[i * n + j]));
        for (int i = 0; i < n; ++i)
            for (int j = i; j < n; ++j) {
                double s = 0; for (int kk = 0; kk < n; ++kk) s += R[kk * 64 + i] * R[kk * 64 + j];
                erec = fmax(erec, fabs(s - A[i * n + j]) / amax);
                if (!SEMIDEF) eref = fmax(eref, fabs(R[i * 64 + j] - Rh[i * n + j]) / fmax(1e-300, fabs(Rh[j * n + j])));
            }
        for (int i = 0; i < 64; ++i) for (int j = 0; j < i; ++j) elow = fmax(elow, fabs(R[i * 64 + j]));
        if (AUG) {
            // W Rm^T = I where Rm = R with a unit diagonal on dropped rows
            for (int i = 0; i < 64; ++i)
                for (int j = 0; j < 64; ++j) {
                    double s = 0;
                    for (int kk = 0; kk < 64; ++kk) {
                        double rjk = R[kk * 64 + j];              // Rm^T[kk][j] = Rm[j][kk]... use (W Rm^T)[i][j] = sum_k W[i][k] Rm[j][k]
                        (void)rjk;
                        double rm = R[j * 64 + kk];
                        if (kk == j && rm == 0.0) rm = 1.0;
                        s += W[i * 64 + kk] * rm;
                    }
                    einv = fmax(einv, fabs(s - (i == j ? 1.0 : 0.0)));
                }
        }
        // (round 4: the inverse factor is part of the verdict -- until then |W R^T - I| was printed but never tested, so a
        // build whose replicas held the wrong diagonal block still ended in "ALL OK")
        if (erec > 1e-13 || elow != 0.0 || (AUG && !(einv < 1e-10))) bad = 1;
    }
    printf("%-28s n=%2d fail=%d (expect %d)  recon %.1e  vs host %.1e  |W R^T - I| %.1e  lower %.1e  total %.2f us %s\n", name, n,
           fail, expect_fail, erec, eref, einv, elow, (st[15] - st[0]) / 100.0, bad ? "  <-- BAD" : "");
    if (n == 64 && check) {
        printf("    shader clock %.0f MHz;", (double)(cy[1] - cy[0]) / ((st[15] - st[0]) / 100.0));
        printf(" panel/trailing per block step (us):");
        for (int kk = 0; kk < 4; ++kk)
            printf("  [%.2f %.2f]", (st[2 + 2 * kk] - st[1 + 2 * kk]) / 100.0,
                   kk < 3 ? (st[3 + 2 * kk] - st[2 + 2 * kk]) / 100.0 : 0.0);
        printf("\n");
        unsigned long long ps[2][16]; hipMemcpyFromSymbol(ps, HIP_SYMBOL(g_ps), sizeof ps);
        for (int h = 0; h < 2; ++h) {
            printf("    half %d (cycles since half 0 start): load %lld, apply-end %lld, pivots:", h, (long long)(ps[h][0] - ps[0][0]), (long long)(ps[h][1] - ps[0][0]));
            for (int q = 0; q < 8; ++q) printf(" %lld", (long long)(ps[h][2 + q] - ps[0][0]));
            printf(" | scales %lld wb %lld end %lld\n", (long long)(ps[h][10] - ps[0][0]), (long long)(ps[h][11] - ps[0][0]), (long long)(ps[h][12] - ps[0][0]));
        }
    }
    hipFree(dA); hipFree(dR); hipFree(dW); hipFree(df);
    return bad;
}

static std::vector<double> gram(int n, int m, unsigned seed, double cond_pow = 0.0) {
    std::vector<double> X(n * m), A(n * n);
    unsigned long long s = seed * 2654435761ull + 12345;
    for (auto& x : X) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = ((double)(s >> 11) / 9007199254740992.0) - 0.5; }
    if (cond_pow > 0) for (int i = 0; i < n; ++i) { const double sc = pow(10.0, -cond_pow * i / (n - 1)); for (int kk = 0; kk < m; ++kk) X[i * m + kk] *= sc; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double t = 0; for (int kk = 0; kk < m; ++kk) t += X[i * m + kk] * X[j * m + kk]; A[i * n + j] = t; }
    return A;
}

int main() {
    int bad = 0;
    bad += run<false, true>("spd aug", 64, gram(64, 256, 1), 0, true);
    bad += run<false, false>("spd plain", 64, gram(64, 256, 2), 0, true);
    bad += run<false, true>("spd aug n=34", 34, gram(34, 100, 3), 0, true);
    bad += run<false, false>("spd plain n=16", 16, gram(16, 100, 4), 0, true);
    bad += run<false, true>("spd aug n=48", 48, gram(48, 100, 5), 0, true);
    bad += run<false, true>("spd aug n=2", 2, gram(2, 10, 6), 0, true);
    bad += run<false, true>("graded 1e5 aug", 64, gram(64, 256, 7, 5.0), 0, true);
    {   // dependent rows, semi-definite rule
        const int n = 64, m = 256;
        std::vector<double> X(n * m), A(n * n);
        unsigned long long s = 99;
        for (auto& x : X) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = ((double)(s >> 11) / 9007199254740992.0) - 0.5; }
        for (int kk = 0; kk < m; ++kk) { X[10 * m + kk] = 2 * X[3 * m + kk] - X[5 * m + kk]; X[40 * m + kk] = X[39 * m + kk]; X[63 * m + kk] = X[0 * m + kk] + X[62 * m + kk]; }
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double t = 0; for (int kk = 0; kk < m; ++kk) t += X[i * m + kk] * X[j * m + kk]; A[i * n + j] = t; }
        bad += run<true, true>("dependent rows (semidef)", 64, A, 0, true);
        // the same matrix under the plain rule must fail at the first dependent row (index 11) -- or pass by luck of rounding
        run<false, true>("dependent rows (plain rule)", 64, A, 11, false);
    }
    {   // indefinite: fails at pivot 21
        std::vector<double> A = gram(64, 256, 8);
        A[20 * 64 + 20] = -1.0;
        bad += run<false, false>("negative pivot 21", 64, A, 21, false);
        A = gram(64, 256, 9);
        A[5 * 64 + 7] = NAN; A[7 * 64 + 5] = NAN;
        bad += run<false, true>("NaN entry", 64, A, 8, false);
    }
    printf(bad ? "FAILED\n" : "ALL OK\n");
    return bad;
}
