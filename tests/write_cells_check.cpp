// write_cells_check.cpp -- TEST INFRASTRUCTURE: write_cells_line (rsem_amd/csrc/host/files.hpp: the long per-transcript rows of .theta /
// .model / the result files formatted in pieces on the host's threads) must write the bytes a plain fprintf loop writes.
#include <cmath>

#include "../rsem_amd/csrc/host/files.hpp"

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 120001;
    std::vector<double> v(n);
    for (long i = 0; i < n; i++) v[i] = std::exp(-(double)(i % 977) / 50.0) * 1.234567891234 * (i % 13 == 0 ? 0.0 : 1.0) / (1 + i % 7);
    // cells longer than 63 characters ("%.2f" of 1e61 and above; up to 312 for the largest double): whole, as fprintf writes them
    if (n > 5000) { v[17] = 1.5e61; v[4001] = 8.25e299; v[n - 2] = 1.7e302; }
    FILE* a = fopen(argv[2], "w");
    FILE* b = fopen(argv[3], "w");
    rsemh::write_cells_line(a, 0, n - 1, ' ', [&](char* buf, long i) { return snprintf(buf, rsemh::kCellBuf, "%.15g", v[i]); });
    rsemh::write_cells_line(a, 1, n - 1, '\t', [&](char* buf, long i) { return snprintf(buf, rsemh::kCellBuf, "%.2f", v[i] * 1e6); });
    for (long i = 0; i < n - 1; i++) fprintf(b, "%.15g ", v[i]);
    fprintf(b, "%.15g\n", v[n - 1]);
    for (long i = 1; i < n; i++) { fprintf(b, "%.2f", v[i] * 1e6); fputc(i < n - 1 ? '\t' : '\n', b); }
    fclose(a); fclose(b);
    return 0;
}
