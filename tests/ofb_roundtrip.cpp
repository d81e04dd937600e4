// ofb_roundtrip.cpp -- TEST INFRASTRUCTURE (CPU): rsem_amd/csrc/host/ofb.hpp against the text hand-off.
//   ofb_roundtrip <imd.ofg> <imdName of a scratch directory>
// (1) every value of the text file is a fixed point of through_15_digits (it IS a 15-digit decimal), and random doubles
//     go where printf("%.15g") + strtod take them; (2) write_ofb of the parsed items + load_ofb gives the same arrays;
// (3) a text file written after the directory makes ofb_present() say no.
#include <cstdio>
#include <random>

#include "../rsem_amd/csrc/host/ofb.hpp"

using namespace rsemh;

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    OfgData D = load_ofg(argv[1]);
    const uint64_t N1 = D.row_ptr.size() - 1, n = D.sid.size();
    for (uint64_t j = 0; j < n; j++)
        if (through_15_digits(D.conprb[j]) != D.conprb[j]) { printf("value %llu is not a fixed point\n", (unsigned long long)j); return 1; }
    std::mt19937_64 g(5);
    for (int i = 0; i < 200000; i++) {
        const double v = std::ldexp((double)(g() >> 11) / 9007199254740992.0 + 0.5, (int)(g() % 600) - 400);
        char tmp[64];
        snprintf(tmp, sizeof(tmp), "%.15g", v);
        if (through_15_digits(v) != strtod(tmp, nullptr)) { printf("through_15_digits(%.17g) differs from printf/strtod\n", v); return 1; }
    }
    const std::string imd = argv[2];
    const int nt = 3;
    std::vector<OfbPart> parts(nt);
    for (int t = 0; t < nt; t++) {
        const uint64_t lo = N1 * t / nt, hi = N1 * (t + 1) / nt;
        for (uint64_t i = lo; i < hi; i++) {
            parts[t].lens.push_back((uint32_t)(D.row_ptr[i + 1] - D.row_ptr[i]));
            for (uint64_t k = D.row_ptr[i]; k < D.row_ptr[i + 1]; k++) { parts[t].sid.push_back(D.sid[k]); parts[t].val.push_back(D.conprb[k]); }
        }
    }
    write_ofb(imd, D.M, D.N0, parts);
    if (!ofb_present(imd)) { printf("ofb_present: no\n"); return 1; }
    OfgData B = load_ofb(imd);
    if (B.M != D.M || B.N0 != D.N0 || B.row_ptr.size() != D.row_ptr.size() || B.sid.size() != n) { printf("header / sizes differ\n"); return 1; }
    for (uint64_t i = 0; i <= N1; i++) if (B.row_ptr[i] != D.row_ptr[i]) { printf("row_ptr differs\n"); return 1; }
    for (uint64_t j = 0; j < n; j++) if (B.sid[j] != D.sid[j] || B.conprb[j] != D.conprb[j]) { printf("items differ\n"); return 1; }
    // a text file written AFTER the arrays wins
    usleep(20000);
    FILE* f = fopen((imd + ".ofg").c_str(), "w");
    fprintf(f, "%d %llu\n", D.M, (unsigned long long)D.N0);
    fclose(f);
    if (ofb_present(imd)) { printf("a newer .ofg must win\n"); return 1; }
    remove_ofb(imd);
    struct stat sb;
    if (stat(ofb_dir(imd).c_str(), &sb) == 0) { printf("remove_ofb left the directory\n"); return 1; }
    printf("ok %llu reads %llu items\n", (unsigned long long)N1, (unsigned long long)n);
    return 0;
}
