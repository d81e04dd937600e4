// Calibration kernel for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section):
// streams a known number of bytes with the same per-lane access widths as the E-step kernel
// (8 B/lane conprb planes, 4 B/lane sid planes) and with 16 B/lane for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void read8(const double* __restrict__ a, size_t n, double* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n; i += st) s += a[i];
    if (s == 12345.678) out[0] = s;
}
__global__ void read4(const int* __restrict__ a, size_t n, int* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    int s = 0;
    for (; i < n; i += st) s += a[i];
    if (s == 123456789) out[0] = s;
}
__global__ void read16(const double2* __restrict__ a, size_t n, double* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n; i += st) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}
__global__ void write8(double* a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) a[i] = 1.0;
}
int main() {
    const size_t bytes = 1ull << 30;  // 1 GiB per kernel, > 256 MiB Infinity Cache
    void* d; double* out;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(d, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(read8, dim3(8192), dim3(256), 0, 0, (const double*)d, bytes / 8, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("read8  : %zu bytes %.3f ms %.1f GB/s\n", bytes, ms, bytes / ms / 1e6);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(read4, dim3(8192), dim3(256), 0, 0, (const int*)d, bytes / 4, (int*)out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("read4  : %zu bytes %.3f ms %.1f GB/s\n", bytes, ms, bytes / ms / 1e6);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(read16, dim3(8192), dim3(256), 0, 0, (const double2*)d, bytes / 16, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("read16 : %zu bytes %.3f ms %.1f GB/s\n", bytes, ms, bytes / ms / 1e6);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(write8, dim3(8192), dim3(256), 0, 0, (double*)d, bytes / 8); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("write8 : %zu bytes %.3f ms %.1f GB/s\n", bytes, ms, bytes / ms / 1e6);
    }
    return 0;
}
