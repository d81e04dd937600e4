// ci_stream.hpp -- the reference's OWN random stream for rsem-calculate-credibility-intervals (--ci-stream reference): the TPM
// samples of calcCI.cpp:93-164 drawn exactly as the reference draws them, so that a run with the same --seed and -p gives the
// reference's numbers and not merely its distribution.
//
// What fixes the reference's draws (sampling.h:19-44, calcCI.cpp:171-187, boost 1.55's random library as shipped with it):
//   * thread t of min(-p, nCV) threads owns imdName.countvectors<t> and an MT19937 of its own, seeded with the t-th DISTINCT
//     output of an MT19937 seeded with --seed (engineFactory::new_engine);
//   * per count vector, nSpC times over: for j = 0 .. M a Gamma(c_j + pseudoC, 1) variate -- drawn only for j = 0 and for
//     transcripts with c_j >= 0, an effective length and a mask weight -- divided by mw[j]; the normalisation to TPM in the
//     reference's mix of float and double (calcCI.cpp:128-148);
//   * a gamma variate is boost::random::gamma_distribution<double>: shape 1 an exponential; shape > 1 the tangent rejection
//     method (two uniforms per attempt); shape < 1 the exponential-power rejection method (a uniform and an exponential per
//     attempt); a uniform is one 32-bit output times 2^-32, an exponential -log(1 - uniform).
// The number of generator outputs a variate takes depends on the values drawn, so a thread's stream cannot be cut into
// independent pieces: this is host code on -p threads, as the reference's is (the device's default sampler is the counter-based
// one of ci.hip); the samples are handed to the device for the interval stage (rsem_ci_calculate_samples).
// The transcendental functions are the host C library's, like the reference's: on the same machine the draws are the same bits.
#pragma once
#include <cmath>
#include <cstdint>
#include <set>
#include <vector>

namespace rsemh {

struct RefMt19937 {  // boost::random::mt19937
    uint32_t mt[624];
    int idx;
    explicit RefMt19937(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int k = 0; k < 624; k++) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double uniform01() { return (double)next() * (1.0 / 4294967296.0); }   // uniform_01<double> on a 32-bit engine: always < 1
    double exponential() { return -1.0 / 1.0 * std::log(1.0 - uniform01()); }  // exponential_distribution<double>(1)
};

// the seeds engineFactory::new_engine hands out, in order (sampling.h:26-38)
inline std::vector<uint32_t> ref_engine_seeds(uint32_t seed, int n) {
    RefMt19937 g(seed);
    std::set<uint32_t> seen;
    std::vector<uint32_t> out;
    while ((int)out.size() < n) {
        const uint32_t s = g.next();
        if (seen.insert(s).second) out.push_back(s);
    }
    return out;
}

// boost::random::gamma_distribution<double>(alpha, 1)(engine)
struct RefGamma {
    double alpha, p;
    explicit RefGamma(double a) : alpha(a), p(std::exp(1.0) / (a + std::exp(1.0))) {}
    double draw(RefMt19937& eng) const {
        if (alpha == 1.0) return eng.exponential() * 1.0;
        if (alpha > 1.0) {
            const double pi = 3.14159265358979323846;
            for (;;) {
                const double y = std::tan(pi * eng.uniform01());
                const double x = std::sqrt(2.0 * alpha - 1.0) * y + alpha - 1.0;
                if (x <= 0.0) continue;
                if (eng.uniform01() > (1.0 + y * y) * std::exp((alpha - 1.0) * std::log(x / (alpha - 1.0)) - std::sqrt(2.0 * alpha - 1.0) * y)) continue;
                return x * 1.0;
            }
        }
        for (;;) {
            const double u = eng.uniform01();
            const double y = eng.exponential();
            double x, q;
            if (u < p) {
                x = std::exp(-y / alpha);
                q = p * std::exp(-x);
            } else {
                x = 1.0 + y;
                q = p + (1.0 - p) * std::pow(x, alpha - 1.0);
            }
            if (u >= q) continue;
            return x * 1.0;
        }
    }
};

// sample_theta_from_c for ONE thread's count vectors (calcCI.cpp:93-164): cv = n_cv x (M + 1) counts; for every count vector
// nSpC samples, written as the reference's Buffer lays them out: tpm[(j - 1) * nS + col] for sample column col0 + (v * nSpC + i),
// l_bar[col].  Returns false where the reference would stop at an assert (a sample without expression).
inline bool ref_sample_thread(RefMt19937& eng, int M, const int32_t* cv, int n_cv, int nSpC, double pseudoC, const double* eel, const double* mw,
                              size_t nS, size_t col0, float* tpm_rows, float* l_bars, double eps = 1e-300) {
    std::vector<double> theta((size_t)M + 1);
    std::vector<float> tpm((size_t)M + 1);
    std::vector<RefGamma> gam;
    gam.reserve((size_t)M + 1);
    for (int v = 0; v < n_cv; v++) {
        const int32_t* c = cv + (size_t)v * ((size_t)M + 1);
        gam.clear();
        for (int j = 0; j <= M; j++) gam.emplace_back(c[j] >= 0 ? (double)c[j] + pseudoC : 1.0);
        for (int i = 0; i < nSpC; i++) {
            double sum = 0.0;
            for (int j = 0; j <= M; j++) {
                theta[j] = (j == 0 || (c[j] >= 0 && eel[j] >= eps && mw[j] >= eps)) ? gam[j].draw(eng) / mw[j] : 0.0;
                sum += theta[j];
            }
            if (!(sum >= eps)) return false;
            for (int j = 0; j <= M; j++) theta[j] /= sum;
            sum = 0.0;
            tpm[0] = 0.0f;
            for (int j = 1; j <= M; j++) {
                if (eel[j] >= eps) {
                    tpm[j] = (float)(theta[j] / eel[j]);  // (a float array in the reference: the quotient is rounded here)
                    sum += tpm[j];
                } else tpm[j] = 0.0f;
            }
            if (!(sum >= eps)) return false;
            float l_bar = 0.0f;
            for (int j = 1; j <= M; j++) {
                tpm[j] /= sum;                 // float /= double
                l_bar += tpm[j] * eel[j];      // float += float * double
                tpm[j] *= 1e6;                 // float *= double
            }
            const size_t col = col0 + (size_t)v * nSpC + i;
            l_bars[col] = l_bar;
            for (int j = 1; j <= M; j++) tpm_rows[(size_t)(j - 1) * nS + col] = tpm[j];
        }
    }
    return true;
}

}  // namespace rsemh
