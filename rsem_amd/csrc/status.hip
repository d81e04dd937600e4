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

int rsem_hip_abi_version(void) { return 2; }

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

}  // extern "C"
