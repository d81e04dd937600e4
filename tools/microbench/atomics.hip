// Microbenchmark: fp64 atomic throughput on MI355X for the count-accumulation patterns of the E step.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomics.hip -o atomics ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>
__global__ void k_atomic(double* counts, const int* idx, size_t n, int M, int per_thread) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    double* base = counts;
    if (MODE == 2) base = counts + (size_t)xcc_id() * M;   // per-XCD private copy
    for (size_t i = t; i < n; i += stride) {
        int a = idx[i];
        if (MODE == 0) unsafeAtomicAdd(&base[a], 1.0);                                   // agent scope (default)
        if (MODE == 1) __hip_atomic_fetch_add(&base[a], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) __hip_atomic_fetch_add(&base[a], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__global__ void k_lds_atomic(double* counts, const int* idx, size_t n, int W) {
    __shared__ double win[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) win[i] = 0.0;
    __syncthreads();
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = t; i < n; i += stride) unsafeAtomicAdd(&win[idx[i] & (W - 1)], 1.0);
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) if (win[i] != 0.0) unsafeAtomicAdd(&counts[i], win[i]);
}

int main() {
    const int M = 50000;
    const size_t n = 1 << 24;
    std::vector<int> h(n);
    double *d_counts; int* d_idx;
    CK(hipMalloc(&d_counts, sizeof(double) * M * 8));
    CK(hipMalloc(&d_idx, sizeof(int) * n));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* pat_name[] = {"uniform-random over 50k", "64 consecutive threads share one address", "all to 64 hot addresses", "all-distinct within a wave, window of 2048"};
    for (int pat = 0; pat < 4; pat++) {
        srand(1);
        for (size_t i = 0; i < n; i++) {
            if (pat == 0) h[i] = rand() % M;
            else if (pat == 1) h[i] = (int)((i / 64) * 7919 % M);
            else if (pat == 2) h[i] = rand() % 64;
            else h[i] = (int)((i * 31) % 2048);
        }
        CK(hipMemcpy(d_idx, h.data(), sizeof(int) * n, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 4; mode++) {
            CK(hipMemset(d_counts, 0, sizeof(double) * M * 8));
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_atomic<0>, dim3(2048), dim3(256), 0, 0, d_counts, d_idx, n, M, 0);
                if (mode == 1) hipLaunchKernelGGL(k_atomic<1>, dim3(2048), dim3(256), 0, 0, d_counts, d_idx, n, M, 0);
                if (mode == 2) hipLaunchKernelGGL(k_atomic<2>, dim3(2048), dim3(256), 0, 0, d_counts, d_idx, n, M, 0);
                if (mode == 3) hipLaunchKernelGGL(k_lds_atomic, dim3(2048), dim3(256), 0, 0, d_counts, d_idx, n, 2048);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            std::vector<double> c((size_t)M * 8);
            CK(hipMemcpy(c.data(), d_counts, sizeof(double) * M * 8, hipMemcpyDeviceToHost));
            double tot = 0; for (double v : c) tot += v;
            const char* mname[] = {"global agent-scope", "global workgroup-scope (same array!)", "per-XCD copy, workgroup-scope", "LDS window then flush"};
            printf("pattern[%s] mode[%s]: %.3f ms  %.2f G atomics/s  sum=%.0f (expect %.0f)%s\n", pat_name[pat], mname[mode], best,
                   n / best / 1e6, tot, 3.0 * n, (tot == 3.0 * n) ? "" : "  <-- MISMATCH");
        }
    }
    return 0;
}
