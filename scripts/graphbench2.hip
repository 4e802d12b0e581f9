// Why does a replayed hipGraph of ALTERNATING kernels cost 3.15 us per node when a run of identical nodes costs 1.58?
// (scripts/graphbench.hip, MI355X, ROCm 7.2).  Variants isolate what has to differ between consecutive nodes.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <functional>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_a() {}
__global__ void k_b() {}
__global__ void k_phase(int phase, double* o) { if (phase == 77) o[0] = 1.0; }
__global__ void k_copy(const double* __restrict__ a, double* __restrict__ o) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    o[i] = a[i] + 1.0;
}
static hipStream_t st;
static double time_graph(const std::function<void(int)>& launch, int n, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch(i);
    (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int r = 0; r < 3; ++r) (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps; ++r) (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
    auto t1 = std::chrono::high_resolution_clock::now();
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * (double)n);
}
static double time_eager(const std::function<void(int)>& launch, int n, int reps) {
    for (int i = 0; i < n; ++i) launch(i);
    (void)hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps; ++r) for (int i = 0; i < n; ++i) launch(i);
    (void)hipStreamSynchronize(st);
    auto t1 = std::chrono::high_resolution_clock::now();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * (double)n);
}
int main() {
    CK(hipStreamCreate(&st));
    double *a, *o, *a2, *o2;
    CK(hipMalloc(&a, 8 << 20)); CK(hipMalloc(&o, 8 << 20)); CK(hipMalloc(&a2, 8 << 20)); CK(hipMalloc(&o2, 8 << 20));
    CK(hipMemset(a, 0, 8 << 20)); CK(hipMemset(a2, 0, 8 << 20));
    const int N = 240, R = 50;
    auto A = [&](int) { hipLaunchKernelGGL(k_a, dim3(256), dim3(256), 0, st); };
    printf("graph A A A A            : %.2f us/node\n", time_graph(A, N, R));
    printf("graph A B A B            : %.2f us/node\n", time_graph([&](int i) { if (i & 1) hipLaunchKernelGGL(k_b, dim3(256), dim3(256), 0, st); else hipLaunchKernelGGL(k_a, dim3(256), dim3(256), 0, st); }, N, R));
    printf("graph A A B B            : %.2f us/node\n", time_graph([&](int i) { if (i & 2) hipLaunchKernelGGL(k_b, dim3(256), dim3(256), 0, st); else hipLaunchKernelGGL(k_a, dim3(256), dim3(256), 0, st); }, N, R));
    printf("graph A A A B            : %.2f us/node\n", time_graph([&](int i) { if ((i & 3) == 3) hipLaunchKernelGGL(k_b, dim3(256), dim3(256), 0, st); else hipLaunchKernelGGL(k_a, dim3(256), dim3(256), 0, st); }, N, R));
    printf("graph phase 0 0 0 0      : %.2f us/node\n", time_graph([&](int) { hipLaunchKernelGGL(k_phase, dim3(256), dim3(256), 0, st, 0, o); }, N, R));
    printf("graph phase 0 1 0 1      : %.2f us/node\n", time_graph([&](int i) { hipLaunchKernelGGL(k_phase, dim3(256), dim3(256), 0, st, i & 1, o); }, N, R));
    printf("graph phase i (all diff) : %.2f us/node\n", time_graph([&](int i) { hipLaunchKernelGGL(k_phase, dim3(256), dim3(256), 0, st, i, o); }, N, R));
    printf("graph grid 256/272 alt   : %.2f us/node\n", time_graph([&](int i) { hipLaunchKernelGGL(k_a, dim3((i & 1) ? 272 : 256), dim3(256), 0, st); }, N, R));
    printf("graph block 256/512 alt  : %.2f us/node\n", time_graph([&](int i) { hipLaunchKernelGGL(k_a, dim3(256), dim3((i & 1) ? 512 : 256), 0, st); }, N, R));
    printf("graph copy same args     : %.2f us/node\n", time_graph([&](int) { hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, st, a, o); }, N, R));
    printf("graph copy a->o, o->a    : %.2f us/node\n", time_graph([&](int i) { if (i & 1) hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, st, o, a); else hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, st, a, o); }, N, R));
    printf("graph copy a->o, a2->o2  : %.2f us/node\n", time_graph([&](int i) { if (i & 1) hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, st, a2, o2); else hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, st, a, o); }, N, R));
    printf("eager A A A A            : %.2f us/node\n", time_eager(A, N, R));
    printf("eager A B A B            : %.2f us/node\n", time_eager([&](int i) { if (i & 1) hipLaunchKernelGGL(k_b, dim3(256), dim3(256), 0, st); else hipLaunchKernelGGL(k_a, dim3(256), dim3(256), 0, st); }, N, R));
    printf("eager phase 0 1 0 1      : %.2f us/node\n", time_eager([&](int i) { hipLaunchKernelGGL(k_phase, dim3(256), dim3(256), 0, st, i & 1, o); }, N, R));
    // explicit graph: chain of kernel nodes added by hand (no capture)
    {
        hipGraph_t g; CK(hipGraphCreate(&g, 0));
        hipGraphNode_t prev = nullptr;
        int ph[2] = {0, 1}; double* op = o;
        for (int i = 0; i < N; ++i) {
            void* args[2] = {&ph[i & 1], &op};
            hipKernelNodeParams p = {};
            p.func = (void*)k_phase; p.gridDim = dim3(256); p.blockDim = dim3(256); p.kernelParams = args;
            hipGraphNode_t nd;
            CK(hipGraphAddKernelNode(&nd, g, prev ? &prev : nullptr, prev ? 1 : 0, &p));
            prev = nd;
        }
        hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        auto t1 = std::chrono::high_resolution_clock::now();
        printf("explicit graph phase 0 1 : %.2f us/node\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / (R * (double)N));
    }
    return 0;
}
