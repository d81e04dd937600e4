// Microbenchmark: what TCC FETCH_SIZE (rocprofv3 --pmc FETCH_SIZE, in KB) reports per byte actually requested, by access width --
// the correction factor the roofline's `traffic` needs.  MI355X_MICROARCH.md's x2 holds for wide streaming reads (a wave reading
// 512 B .. 1 KB of consecutive bytes: 128-byte requests counted as 64); a kernel of 64-byte segments (16 lanes x 4 B, the group
// loads of k_model_group) need not have the same factor.  Every kernel reads exactly 1 GiB once (non-temporal: no re-use).
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib ;  rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename T>
__global__ void k_stream(const T* __restrict__ p, size_t n, unsigned long long* sink) {  // consecutive lanes, consecutive elements
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n; i += stride) { T v = __builtin_nontemporal_load(&p[i]); acc += ((const unsigned char*)&v)[0]; }
    if (acc == 0x123456789ull) *sink = acc;
}
// groups of 16 lanes read 16 x 4 = 64 consecutive bytes at a pseudo-random 64-byte-aligned place (every segment exactly once)
__global__ void k_seg64(const unsigned* __restrict__ p, size_t nseg, unsigned long long* sink) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; (t >> 4) < nseg; t += stride) {
        const size_t s = ((t >> 4) * 2654435761ull) % nseg;  // nseg is a power of two and the multiplier odd: a permutation
        acc += __builtin_nontemporal_load(&p[s * 16 + (t & 15)]);
    }
    if (acc == 0x123456789ull) *sink = acc;
}
// the same with 16 lanes x 8 B = 128-byte segments
__global__ void k_seg128(const unsigned long long* __restrict__ p, size_t nseg, unsigned long long* sink) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; (t >> 4) < nseg; t += stride) {
        const size_t s = ((t >> 4) * 2654435761ull) % nseg;
        acc += __builtin_nontemporal_load(&p[s * 16 + (t & 15)]);
    }
    if (acc == 0x123456789ull) *sink = acc;
}
// one lane in 16 reads 8 bytes at a pseudo-random place (a lone 8-byte gather per 128-byte line)
__global__ void k_gather8(const unsigned long long* __restrict__ p, size_t nline, unsigned long long* sink) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; t < nline; t += stride) {
        const size_t s = (t * 2654435761ull) % nline;
        acc += __builtin_nontemporal_load(&p[s * 16]);
    }
    if (acc == 0x123456789ull) *sink = acc;
}

int main() {
    const size_t bytes = 1ull << 30;
    void* d;
    unsigned long long* sink;
    CK(hipMalloc(&d, bytes));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(d, 1, bytes));
    const int grid = 256 * 8, blk = 256;
    hipLaunchKernelGGL(k_stream<unsigned char>, dim3(grid), dim3(blk), 0, 0, (const unsigned char*)d, bytes, sink);
    hipLaunchKernelGGL(k_stream<unsigned>, dim3(grid), dim3(blk), 0, 0, (const unsigned*)d, bytes / 4, sink);
    hipLaunchKernelGGL(k_stream<unsigned long long>, dim3(grid), dim3(blk), 0, 0, (const unsigned long long*)d, bytes / 8, sink);
    hipLaunchKernelGGL(k_stream<u32x4>, dim3(grid), dim3(blk), 0, 0, (const u32x4*)d, bytes / 16, sink);
    hipLaunchKernelGGL(k_seg64, dim3(grid), dim3(blk), 0, 0, (const unsigned*)d, bytes / 64, sink);
    hipLaunchKernelGGL(k_seg128, dim3(grid), dim3(blk), 0, 0, (const unsigned long long*)d, bytes / 128, sink);
    hipLaunchKernelGGL(k_gather8, dim3(grid), dim3(blk), 0, 0, (const unsigned long long*)d, bytes / 128, sink);
    CK(hipDeviceSynchronize());
    printf("fetch_calib: every kernel requested 1 GiB = 1048576 KB (k_gather8: 8 B of every 128-byte line = 65536 KB requested, 1048576 KB of lines touched)\n");
    return 0;
}
