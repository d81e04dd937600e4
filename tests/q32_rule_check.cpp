// Host-only: applies the Q32 rule of sell_layout.hpp (q32_scale_of / q32_mantissa, the functions the device kernels call)
// to rows of doubles read from a file and writes, per row, [qualifies, e] and the mantissas -- tests/test_q32_cpu.py compares
// them with tools/q32_ref.quantize_q32.   usage: q32_rule_check in.bin out.bin n_rows row_len range_bits
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../rsem_amd/csrc/sell_layout.hpp"
int main(int argc, char** argv) {
    if (argc < 6) return 2;
    const long n = atol(argv[3]), L = atol(argv[4]);
    const int range_bits = atoi(argv[5]);
    std::vector<double> v((size_t)n * L);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(v.data(), sizeof(double), v.size(), f) != v.size()) return 3;
    fclose(f);
    std::vector<long long> out((size_t)n * (2 + L));
    for (long i = 0; i < n; i++) {
        // the same scan as k_row_keys
        double vmx = 0.0, vmn = 1.79e308;
        for (long j = 0; j < L; j++) {
            const double x = v[i * L + j];
            if (!(x >= 0.0)) vmx = 1e308;
            vmx = fmax(vmx, x);
            if (x > 0.0) vmn = fmin(vmn, x);
        }
        Q32Scale q{0};
        const bool ok = q32_scale_of(vmx, vmn, range_bits, q);
        out[i * (2 + L)] = ok;
        out[i * (2 + L) + 1] = q.e;
        for (long j = 0; j < L; j++) out[i * (2 + L) + 2 + j] = ok ? (long long)q32_mantissa(v[i * L + j], q.e) : -1;
    }
    f = fopen(argv[2], "wb");
    if (!f || fwrite(out.data(), sizeof(long long), out.size(), f) != out.size()) return 4;
    fclose(f);
    return 0;
}
