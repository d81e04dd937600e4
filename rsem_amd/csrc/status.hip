// status.hip -- error strings, device probing (C ABI: include/rsem_hip.h).
#include <cstring>
#include <cstdarg>

#include "common.hpp"
#include "upload.hpp"

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
typedef double probe_v2 __attribute__((ext_vector_type(2)));
// 16 bytes per lane, eight loads in flight per lane: what a pure streaming kernel reaches on this device
__global__ void __launch_bounds__(256) k_probe_read(const probe_v2* __restrict__ a, size_t n, double* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (; i + 7 * st < n; i += 8 * st) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const probe_v2 v = __builtin_nontemporal_load(&a[i + u * st]);
            s[u] += v.x + v.y;
        }
    }
    for (; i < n; i += st) { const probe_v2 v = a[i]; s[0] += v.x + v.y; }
    const double t = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    if (t == 12345.678) out[0] = t;  // never true for the zero-filled buffer: keeps the loads alive
}
__global__ void __launch_bounds__(256) k_probe_copy(const probe_v2* __restrict__ a, probe_v2* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * st < n; i += 4 * st) {
        const probe_v2 v0 = __builtin_nontemporal_load(&a[i]), v1 = __builtin_nontemporal_load(&a[i + st]);
        const probe_v2 v2 = __builtin_nontemporal_load(&a[i + 2 * st]), v3 = __builtin_nontemporal_load(&a[i + 3 * st]);
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

int rsem_hip_abi_version(void) { return 4; }

int rsem_hip_device_info(int device, const char* key, int64_t* value) {
    if (!key || !value) return RSEM_ERR_INVALID;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) {
        (void)hipGetLastError();
        return RSEM_ERR_NODEVICE;
    }
    if (!strcmp(key, "compute_units")) *value = p.multiProcessorCount;
    else if (!strcmp(key, "clock_khz")) *value = p.clockRate;
    else if (!strcmp(key, "hbm_bytes")) *value = (int64_t)p.totalGlobalMem;
    // (measurement: the staged host -> device copies of this process so far, upload.hpp)
    else if (!strcmp(key, "staged_bytes")) *value = (int64_t)rsem::staged_stats().bytes.load();
    else if (!strcmp(key, "staged_ns")) *value = (int64_t)rsem::staged_stats().ns.load();
    else if (!strcmp(key, "staged_fill_ns")) *value = (int64_t)rsem::staged_stats().ns_fill.load();
    else if (!strcmp(key, "staged_wait_ns")) *value = (int64_t)rsem::staged_stats().ns_wait.load();
    else { rsem::set_last_error("unknown device info key '%s'", key); return RSEM_ERR_INVALID; }
    return RSEM_OK;
}

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

int rsem_hip_preload(int device, int what) {
    int rc = rsem_hip_warmup(device);
    if (rc != RSEM_OK) return rc;
    if (what & RSEM_PRELOAD_EM) rsem::preload_em();
    if (what & RSEM_PRELOAD_MODEL) rsem::preload_model();
    if (what & RSEM_PRELOAD_GIBBS) rsem::preload_gibbs();
    if (what & RSEM_PRELOAD_CI) rsem::preload_ci();
    RSEM_HIP_TRY(hipDeviceSynchronize());
    return RSEM_OK;
}

int rsem_hip_stream_probe(int device, uint64_t bytes, int reps, double* read_GBps, double* copy_GBps) {
    RSEM_REQUIRE(read_GBps && copy_GBps && bytes >= (1u << 20) && reps >= 1, "stream probe: bad arguments");
    RSEM_HIP_TRY(hipSetDevice(device));
    const size_t n = bytes / 32;  // two buffers of n double2
    probe_v2 *a = nullptr, *b = nullptr;
    RSEM_HIP_TRY(hipMalloc(&a, n * 16));
    if (hipMalloc(&b, n * 16) != hipSuccess) {
        (void)hipFree(a);
        rsem::set_last_error("stream probe: hipMalloc of %zu bytes failed", n * 16);
        return RSEM_ERR_NOMEM;
    }
    hipEvent_t e0, e1;
    int rc = RSEM_OK;
    auto body = [&]() -> int {
        RSEM_HIP_TRY(hipMemset(a, 0, n * 16));
        RSEM_HIP_TRY(hipMemset(b, 0, n * 16));
        RSEM_HIP_TRY(hipEventCreate(&e0));
        RSEM_HIP_TRY(hipEventCreate(&e1));
        double best_r = 0.0, best_c = 0.0;
        const int grid = 256 * 16;  // 16 workgroups of 4 waves per CU
        for (int r = 0; r < reps + 1; r++) {  // the first repetition is a warm-up
            float ms = 0.f;
            RSEM_HIP_TRY(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(k_probe_read, dim3(grid), dim3(256), 0, nullptr, a, n, (double*)b);
            RSEM_HIP_TRY(hipEventRecord(e1, nullptr));
            RSEM_HIP_TRY(hipEventSynchronize(e1));
            RSEM_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms > 0.f && n * 16 / (ms * 1e6) > best_r) best_r = n * 16 / (ms * 1e6);
            RSEM_HIP_TRY(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(k_probe_copy, dim3(grid), dim3(256), 0, nullptr, a, b, n);
            RSEM_HIP_TRY(hipEventRecord(e1, nullptr));
            RSEM_HIP_TRY(hipEventSynchronize(e1));
            RSEM_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms > 0.f && n * 32 / (ms * 1e6) > best_c) best_c = n * 32 / (ms * 1e6);
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
