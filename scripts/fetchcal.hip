// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of the GSM kernels (MI355X_MICROARCH.md, HBM:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel reads a known number
// of bytes from a 1 GiB buffer (larger than the 256 MiB Infinity Cache) exactly once:
//   k_read16  : 16 B per lane, 1 KiB contiguous per wave instruction        (the guide's calibrated case)
//   k_read8   :  8 B per lane, 512 B contiguous per wave instruction
//   k_read8seg:  8 B per lane as FOUR 128-byte row segments per wave instruction, rows 8 KiB apart -- the shape of
//                the S0 tile loads of k_gsm_cov_sym and of the M stream of k_panel_fast at D = 1024
// Run:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetchcal     (and once more with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ void k_read16(const v2d* __restrict__ a, double* __restrict__ o, size_t n16) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { v2d v = a[i]; s += v.x + v.y; }
    if (s == 1.2345e300) o[0] = s;
}
__global__ void k_read8(const double* __restrict__ a, double* __restrict__ o, size_t n8) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) s += a[i];
    if (s == 1.2345e300) o[0] = s;
}
// matrix of `rows` x 1024 doubles (8 KiB rows): wave w reads columns [16 c0, 16 c0 + 16) of rows 4 r0 .. 4 r0 + 3
__global__ void k_read8seg(const double* __restrict__ a, double* __restrict__ o, size_t rows) {
    const int lane = threadIdx.x & 63, c = lane & 15, ks = lane >> 4;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
    double s = 0;
    const size_t units = (rows / 4) * 64;                 // (row group, 16-column strip) pairs
    for (size_t u = wave; u < units; u += nwaves) {
        const size_t rg = u / 64, strip = u % 64;
        s += a[(rg * 4 + ks) * 1024 + strip * 16 + c];
    }
    if (s == 1.2345e300) o[0] = s;
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    double *a, *o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) return 1;
    (void)hipMemset(a, 0, bytes);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const v2d*)a, o, bytes / 16);
        hipLaunchKernelGGL(k_read8, dim3(4096), dim3(256), 0, 0, a, o, bytes / 8);
        hipLaunchKernelGGL(k_read8seg, dim3(4096), dim3(256), 0, 0, a, o, bytes / 8192);
    }
    (void)hipDeviceSynchronize();
    printf("each kernel reads %zu bytes (%.1f KiB)\n", bytes, bytes / 1024.0);
    return 0;
}
