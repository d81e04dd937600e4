// rng.hpp -- counter-based random numbers for the sampling kernels (Gibbs PARALLEL mode, credibility intervals).
// Philox4x32-10: every draw is a pure function of (key, counter), so kernels can regenerate a variate instead of
// storing it and results do not depend on the launch geometry.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rsem {

// (host-callable as well: tests/rng_kat_check.cpp runs the known-answer vectors of Random123 through these very functions)
__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) {
#ifdef __HIP_DEVICE_COMPILE__
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

struct Philox {
    uint32_t k0, k1;
    __host__ __device__ inline void round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t a, uint32_t b) const {
        uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ a, n1 = lo1, n2 = hi0 ^ c3 ^ b, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    // Philox4x32-10 (Salmon et al., SC'11)
    __host__ __device__ inline void gen(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* out) const {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            round(c0, c1, c2, c3, a, b);
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};

// Philox2x32-10 (same paper): 64 bits per call for half the multiplications -- one 53-bit uniform per read is all the
// Gibbs sweep needs (gibbs_block.hpp).
__host__ __device__ inline void philox2x32_10(uint32_t key, uint32_t c0, uint32_t c1, uint32_t* out) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const uint32_t hi = mulhi32(0xD256D193u, c0), lo = 0xD256D193u * c0;
        c0 = hi ^ key ^ c1;
        c1 = lo;
        key += 0x9E3779B9u;
    }
    out[0] = c0; out[1] = c1;
}

__host__ __device__ inline double u53(uint32_t hi, uint32_t lo) {  // uniform in [0,1)
    return (double)(((uint64_t)(hi >> 5) << 26) | (lo >> 6)) * (1.0 / 9007199254740992.0);
}

// Gamma(shape a, scale 1), a > 0.  Marsaglia & Tsang (2000); a < 1 via Gamma(a+1) * U^(1/a).
__device__ inline double gamma_draw(const Philox& ph, uint32_t idx, uint32_t sweep, double a) {
    double boost = 1.0;
    uint32_t ctr = 0;
    uint32_t r[4];
    if (a < 1.0) {
        ph.gen(idx, sweep, 0x47414d4du, ctr++, r);
        double u = u53(r[0], r[1]);
        if (u <= 0.0) u = 1.0 / 9007199254740992.0;
        boost = exp(log(u) / a);
        a += 1.0;
    }
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        ph.gen(idx, sweep, 0x47414d4du, ctr++, r);
        double u1 = u53(r[0], r[1]), u2 = u53(r[2], r[3]);
        if (u1 <= 0.0) u1 = 1.0 / 9007199254740992.0;
        double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);  // Box-Muller
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        ph.gen(idx, sweep, 0x47414d4du, ctr++, r);
        double u = u53(r[0], r[1]);
        if (u <= 0.0) u = 1.0 / 9007199254740992.0;
        double x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2) return d * v * boost;
        if (log(u) < 0.5 * x2 + d * (1.0 - v + log(v))) return d * v * boost;
    }
}


// Same distribution, one Philox block per attempt (32-bit uniforms for the Box-Muller normal, 53-bit for the
// acceptance test): the bulk sampler of the credibility-interval kernels, shape a >= 1 expected (a < 1 handled).
__device__ inline double gamma_draw_bulk(const Philox& ph, uint32_t c0, uint32_t c1, uint32_t c2, double a) {
    double boost = 1.0;
    uint32_t ctr = 0;
    uint32_t r[4];
    if (a < 1.0) {
        ph.gen(c0, c1, c2, 0x80000000u, r);
        double u = u53(r[0], r[1]);
        if (u <= 0.0) u = 1.0 / 9007199254740992.0;
        boost = exp(log(u) / a);
        a += 1.0;
    }
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        ph.gen(c0, c1, c2, ctr++, r);
        const double u1 = ((double)r[0] + 0.5) * (1.0 / 4294967296.0), u2 = (double)r[1] * (1.0 / 4294967296.0);
        const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);  // Box-Muller
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        double u = u53(r[2], r[3]);
        if (u <= 0.0) u = 1.0 / 9007199254740992.0;
        const double x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2) return d * v * boost;
        if (log(u) < 0.5 * x2 + d * (1.0 - v + log(v))) return d * v * boost;
    }
}

}  // namespace rsem
