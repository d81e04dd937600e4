// common.hpp -- shared helpers of librsem_hip.so (error plumbing, launch geometry).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/rsem_hip.h"

namespace rsem {

// thread-local detail string behind rsem_hip_last_error()
void set_last_error(const char* fmt, ...);

constexpr double kEpsilon = 1e-300;  // utils.h:19
constexpr int kWave = 64;            // gfx950 wavefront

inline int ceil_div(uint64_t a, uint64_t b) { return (int)((a + b - 1) / b); }

// one empty launch per translation unit = its code object is loaded (rsem_hip_preload, status.hip); defined in em.hip, model.hip,
// gibbs.hip, ci.hip
void preload_em();
void preload_model();
void preload_gibbs();
void preload_ci();

}  // namespace rsem

#define RSEM_HIP_TRY(expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            rsem::set_last_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return (_e == hipErrorOutOfMemory) ? RSEM_ERR_NOMEM : RSEM_ERR_HIP;                 \
        }                                                                                       \
    } while (0)

#define RSEM_REQUIRE(cond, msg)                                               \
    do {                                                                      \
        if (!(cond)) {                                                        \
            rsem::set_last_error("%s:%d: %s", __FILE__, __LINE__, msg);       \
            return RSEM_ERR_INVALID;                                          \
        }                                                                     \
    } while (0)
