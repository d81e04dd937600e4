// ci_stream_check.cpp -- TEST INFRASTRUCTURE for rsem_amd/csrc/host/ci_stream.hpp (the reference's own random stream of
// rsem-calculate-credibility-intervals, --ci-stream reference) without a GPU: draws the TPM samples of a fixture exactly as
// calc_ci.cpp does and writes them out; tests/test_ci_stream_cpu.py puts the oracle's interval arithmetic (a restatement of
// calcCI.cpp:216-284, pinned bit for bit elsewhere) on top and compares with the rows the REFERENCE BINARY appended.
//   ci_stream_check in.bin out.bin    in:  i32 M, nfiles, nSpC, seed; f64 pseudoC; f64 eel[M+1]; f64 mw[M+1];
//                                          per file: i32 n_cv; i32 cv[n_cv][M+1]
//                                     out: f32 tpm[M][nS]; f32 l_bars[nS]
#include <cstdio>
#include <cstdlib>

#include "../rsem_amd/csrc/host/ci_stream.hpp"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[4];
    double pseudoC;
    if (fread(hdr, 4, 4, f) != 4 || fread(&pseudoC, 8, 1, f) != 1) return 2;
    const int M = hdr[0], nfiles = hdr[1], nSpC = hdr[2];
    std::vector<double> eel(M + 1), mw(M + 1);
    if (fread(eel.data(), 8, M + 1, f) != (size_t)M + 1 || fread(mw.data(), 8, M + 1, f) != (size_t)M + 1) return 2;
    std::vector<std::vector<int32_t>> parts(nfiles);
    size_t ncv = 0;
    for (int k = 0; k < nfiles; k++) {
        int32_t n;
        if (fread(&n, 4, 1, f) != 1) return 2;
        parts[k].resize((size_t)n * (M + 1));
        if (fread(parts[k].data(), 4, parts[k].size(), f) != parts[k].size()) return 2;
        ncv += n;
    }
    fclose(f);
    const size_t nS = ncv * nSpC;
    std::vector<float> samples((size_t)M * nS), lbars(nS);
    const std::vector<uint32_t> seeds = rsemh::ref_engine_seeds((uint32_t)hdr[3], nfiles);
    size_t col0 = 0;
    for (int k = 0; k < nfiles; k++) {
        rsemh::RefMt19937 eng(seeds[k]);
        const int n = (int)(parts[k].size() / ((size_t)M + 1));
        if (!rsemh::ref_sample_thread(eng, M, parts[k].data(), n, nSpC, pseudoC, eel.data(), mw.data(), nS, col0, samples.data(), lbars.data())) return 3;
        col0 += (size_t)n * nSpC;
    }
    FILE* g = fopen(argv[2], "wb");
    fwrite(samples.data(), 4, samples.size(), g);
    fwrite(lbars.data(), 4, lbars.size(), g);
    fclose(g);
    return 0;
}
