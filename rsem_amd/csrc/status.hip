// status.hip -- error strings, device probing (C ABI: include/rsem_hip.h).
#include <cstdarg>

#include "common.hpp"

namespace rsem {
static thread_local char g_last_error[1024] = "";
void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}
}  // namespace rsem

namespace {
__global__ void __launch_bounds__(256) k_probe_read(const double* __restrict__ a, size_t n, double* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (; i + 3 * st < n; i += 4 * st) {
        s0 += a[i];
        s1 += a[i + st];
        s2 += a[i + 2 * st];
        s3 += a[i + 3 * st];
    }
    for (; i < n; i += st) s0 += a[i];
    const double s = (s0 + s1) + (s2 + s3);
    if (s == 12345.678) out[0] = s;  // never true for the zero-filled buffer: keeps the loads alive
}
__global__ void __launch_bounds__(256) k_probe_copy(const double* __restrict__ a, double* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * st < n; i += 4 * st) {
        const double v0 = a[i], v1 = a[i + st], v2 = a[i + 2 * st], v3 = a[i + 3 * st];
        b[i] = v0;
        b[i + st] = v1;
        b[i + 2 * st] = v2;
        b[i + 3 * st] = v3;
    }
    for (; i < n; i += st) b[i] = a[i];
}
}  // namespace

extern "C" {

const char* rsem_hip_strerror(int status) {
    switch (status) {
        case RSEM_OK: return "ok";
        case RSEM_ERR_INVALID: return "invalid argument";
        case RSEM_ERR_HIP: return "HIP runtime error";
        case RSEM_ERR_NOMEM: return "out of memory";
        case RSEM_ERR_NODEVICE: return "no usable gfx950 device";
        case RSEM_ERR_STATE: return "invalid call sequence";
        default: return "unknown rsem_status";
    }
}

const char* rsem_hip_last_error(void) { return rsem::g_last_error; }

int rsem_hip_abi_version(void) { return 3; }

int rsem_hip_device_count(int* n) {
    if (!n) return RSEM_ERR_INVALID;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *n = c;
    return RSEM_OK;
}

int rsem_hip_warmup(int device) {
    RSEM_HIP_TRY(hipSetDevice(device));
    RSEM_HIP_TRY(hipFree(nullptr));
    return RSEM_OK;
}

int rsem_hip_stream_probe(int device, uint64_t bytes, int reps, double* read_GBps, double* copy_GBps) {
    RSEM_REQUIRE(read_GBps && copy_GBps && bytes >= (1u << 20) && reps >= 1, "stream probe: bad arguments");
    RSEM_HIP_TRY(hipSetDevice(device));
    const size_t n = bytes / 16;  // two buffers of n doubles
    double *a = nullptr, *b = nullptr;
    RSEM_HIP_TRY(hipMalloc(&a, n * 8));
    if (hipMalloc(&b, n * 8) != hipSuccess) {
        (void)hipFree(a);
        rsem::set_last_error("stream probe: hipMalloc of %zu bytes failed", n * 8);
        return RSEM_ERR_NOMEM;
    }
    hipEvent_t e0, e1;
    int rc = RSEM_OK;
    auto body = [&]() -> int {
        RSEM_HIP_TRY(hipMemset(a, 0, n * 8));
        RSEM_HIP_TRY(hipMemset(b, 0, n * 8));
        RSEM_HIP_TRY(hipEventCreate(&e0));
        RSEM_HIP_TRY(hipEventCreate(&e1));
        double best_r = 0.0, best_c = 0.0;
        const int grid = 256 * 32;  // 32 workgroups of 4 waves per CU
        for (int r = 0; r < reps + 1; r++) {  // the first repetition is a warm-up
            float ms = 0.f;
            RSEM_HIP_TRY(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(k_probe_read, dim3(grid), dim3(256), 0, nullptr, a, n, b);
            RSEM_HIP_TRY(hipEventRecord(e1, nullptr));
            RSEM_HIP_TRY(hipEventSynchronize(e1));
            RSEM_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms > 0.f && n * 8 / (ms * 1e6) > best_r) best_r = n * 8 / (ms * 1e6);
            RSEM_HIP_TRY(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(k_probe_copy, dim3(grid), dim3(256), 0, nullptr, a, b, n);
            RSEM_HIP_TRY(hipEventRecord(e1, nullptr));
            RSEM_HIP_TRY(hipEventSynchronize(e1));
            RSEM_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms > 0.f && n * 16 / (ms * 1e6) > best_c) best_c = n * 16 / (ms * 1e6);
        }
        *read_GBps = best_r;
        *copy_GBps = best_c;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return RSEM_OK;
    };
    rc = body();
    (void)hipFree(a);
    (void)hipFree(b);
    return rc;
}

}  // extern "C"
