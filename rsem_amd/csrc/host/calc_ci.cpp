// rsem-calculate-credibility-intervals on MI355X: same argv, same files as the reference program
// (calcCI.cpp:489-581).
//
//   rsem-calculate-credibility-intervals refName imdName statName confidence nCV nSpC nMB
//                                        [-p #Threads] [--seed seed] [--pseudo-count a] [-q]     + [--device d] [--ci-stream device|reference]
//
// reads   refName.{seq,grp[,ta]}, statName.model (gld + mw), imdName.countvectors<k> for k < min(#Threads, nCV)
// appends six rows (TPM lb / ub / cqv, FPKM lb / ub / cqv, "%.6g") to imdName.iso_res (allele_res when the
//         reference is allele-specific, then also the isoform rows to iso_res) and imdName.gene_res.
//
// nMB (the reference's buffer before it spills the sample matrix to imdName.tmp) is accepted and ignored: the matrix
// stays in HBM (rsem_ci_calculate, rsem_amd/csrc/ci.hip).  -p only tells how many count-vector files rsem-run-gibbs
// wrote.  Without --seed the generator is seeded from the clock, as the reference's is (sampling.h:21-24).
//
// --ci-stream device (the default): the expression samples are drawn on the GPU from a counter-based generator -- the reference's
// distribution, not its draws.  --ci-stream reference: the reference's own per-thread MT19937 + boost gamma draws for this
// --seed and -p (host/ci_stream.hpp, on -p host threads like the reference's), the intervals on the GPU: the reference's numbers.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/rsem_hip.h"
#include "ci_stream.hpp"
#include "files.hpp"
#include "model_host.hpp"
#include "results.hpp"

using namespace rsemh;

namespace {

void append_rows(const std::string& path, const float* ci_t, const float* ci_f, int n) {  // calcCI.cpp:443-457
    FILE* fo = fopen(path.c_str(), "a");
    if (!fo) die("Cannot open %s for appending!", path.c_str());
    std::string line;
    char tmp[48];
    for (int part = 0; part < 2; part++) {
        const float* a = part ? ci_f : ci_t;
        for (int k = 0; k < 3; k++) {
            line.clear();
            for (int i = 0; i < n; i++) {
                const int len = snprintf(tmp, sizeof(tmp), "%.6g", (double)a[(size_t)k * n + i]);
                line.append(tmp, len);
                line.push_back(i < n - 1 ? '\t' : '\n');
            }
            fwrite(line.data(), 1, line.size(), fo);
        }
    }
    fclose(fo);
}

}  // namespace

int main(int argc, char* argv[]) {
    if (argc < 8) {
        printf("Usage: rsem-calculate-credibility-intervals reference_name imdName statName confidence nCV nSpC nMB [-p #Threads] [--seed seed] [--pseudo-count pseudo_count] [-q]\n");
        exit(-1);
    }
    const std::string refName = argv[1], imdName = argv[2], statName = argv[3];
    const double confidence = atof(argv[4]);
    const int nCV = atoi(argv[5]), nSpC = atoi(argv[6]);
    int nThreads = 1, device = 0;
    bool quiet = false, hasSeed = false, refStream = false;
    uint64_t seed = 0;
    double pseudoC = 1.0;
    for (int i = 8; i < argc; i++) {  // calcCI.cpp:507-518
        if (!strcmp(argv[i], "-p") && i + 1 < argc) nThreads = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--seed") && i + 1 < argc) {
            hasSeed = true;
            seed = 0;
            for (const char* c = argv[i + 1]; *c; ++c) seed = seed * 10 + (uint64_t)(*c - '0');
            seed &= 0xffffffffu;  // seedType is uint32 (sampling.h:14)
        }
        if (!strcmp(argv[i], "--pseudo-count") && i + 1 < argc) pseudoC = atof(argv[i + 1]);
        if (!strcmp(argv[i], "-q")) quiet = true;
        if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--ci-stream") && i + 1 < argc) {
            if (!strcmp(argv[i + 1], "reference")) refStream = true;
            else if (strcmp(argv[i + 1], "device")) die("--ci-stream: device or reference");
        }
    }
    const bool verbose = !quiet;
    if (!hasSeed) seed = (uint64_t)std::chrono::system_clock::now().time_since_epoch().count();
    if (nCV <= 0 || nSpC <= 0) die("nCV and nSpC must be positive!");
    if (!(confidence > 0.0 && confidence <= 1.0)) die("confidence must be in (0, 1]!");

    RefInfo refs = load_refs(refName + ".seq", false);
    const int M = refs.M;
    GroupInfo gi, ta;
    if (!gi.load(refName + ".grp")) die("Cannot open %s.grp! It may not exist.", refName.c_str());
    const bool alleleS = is_allele_specific(refName);  // calcCI.cpp:531-532
    if (alleleS && !ta.load(refName + ".ta")) die("Cannot load %s.ta!", refName.c_str());
    Model model;
    model.read(statName + ".model", M);
    if ((int)model.mw.size() != M + 1) die("%s.model does not carry the mask weights of %d transcripts!", statName.c_str(), M);
    const std::vector<double> eel = calc_eel(M, refs, model.gld);  // calcCI.cpp:169

    // count vectors: one file per Gibbs thread (calcCI.cpp:171-184, Gibbs.cpp:257-262)
    const int nfiles = std::min(std::max(nThreads, 1), nCV);
    std::vector<std::vector<int32_t>> parts(nfiles);
    parallel_for(nfiles, [&](int k) {
        const std::string path = imdName + ".countvectors" + std::to_string(k);
        MappedFile f;
        if (!f.open(path)) die("Cannot open %s! It may not exist.", path.c_str());
        const char* p = f.data;
        const char* e = f.data + f.size;
        std::vector<int32_t>& v = parts[k];
        v.reserve((size_t)(nCV / nfiles + 1) * (M + 1));
        for (;;) {
            while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
            if (p >= e) break;
            bool neg = false;
            if (*p == '-') { neg = true; ++p; }
            if (p >= e || *p < '0' || *p > '9') die("%s: not a count vector file", path.c_str());
            long x = 0;
            while (p < e && *p >= '0' && *p <= '9') x = x * 10 + (*p++ - '0');
            v.push_back((int32_t)(neg ? -x : x));
        }
        if (v.size() % ((size_t)M + 1)) die("%s: the number of counts is not a multiple of M + 1 = %d", path.c_str(), M + 1);
    });
    std::vector<int32_t> cvecs;
    cvecs.reserve((size_t)nCV * (M + 1));
    for (auto& v : parts) cvecs.insert(cvecs.end(), v.begin(), v.end());
    if (cvecs.size() != (size_t)nCV * (M + 1))
        die("Found %zu count vectors in %s.countvectors0..%d, expected nCV = %d!", cvecs.size() / ((size_t)M + 1), imdName.c_str(), nfiles - 1, nCV);
    for (int v = 0; v < nCV; v++)
        if (cvecs[(size_t)v * (M + 1)] < 0) die("Count vector %d has a negative noise count!", v);  // calcCI.cpp:112

    int ndev = 0;
    rsem_hip_device_count(&ndev);
    if (ndev < 1) die("rsem-calculate-credibility-intervals: no usable GPU (this program has no CPU path)");
    if (device < 0 || device >= ndev) die("--device %d: only %d device(s) present", device, ndev);

    const int m = gi.m, m_trans = alleleS ? ta.m : 0;
    std::vector<float> tpm(3 * (size_t)M), fpkm(3 * (size_t)M), gtpm(3 * (size_t)m), gfpkm(3 * (size_t)m), itpm(3 * (size_t)m_trans),
        ifpkm(3 * (size_t)m_trans);
    rsem_ci_profile prof;
    int rc;
    if (!refStream) {
        rc = rsem_ci_calculate(device, M, nCV, nSpC, cvecs.data(), eel.data(), model.mw.data(), pseudoC, seed, confidence, m,
                               gi.starts.data(), m_trans, alleleS ? ta.starts.data() : nullptr, tpm.data(), fpkm.data(), gtpm.data(),
                               gfpkm.data(), alleleS ? itpm.data() : nullptr, alleleS ? ifpkm.data() : nullptr, &prof);
    } else {
        // The reference's draws: thread t takes the count vectors of imdName.countvectors<t> and the t-th engine of the factory
        // (calcCI.cpp:171-187).  The columns of the sample matrix are laid thread after thread (the reference's own column order is
        // the order in which its threads happen to reach the buffer: no result depends on it).
        const size_t nS = (size_t)nCV * nSpC;
        std::vector<float> samples((size_t)M * nS), l_bars(nS);
        const std::vector<uint32_t> seeds = ref_engine_seeds((uint32_t)seed, nfiles);
        std::vector<size_t> col0(nfiles + 1, 0);
        for (int k = 0; k < nfiles; k++) col0[k + 1] = col0[k] + parts[k].size() / ((size_t)M + 1) * nSpC;
        std::vector<int> ok(nfiles, 1);
        const auto t0 = std::chrono::steady_clock::now();
        parallel_for(nfiles, [&](int k) {
            RefMt19937 eng(seeds[k]);
            ok[k] = ref_sample_thread(eng, M, parts[k].data(), (int)(parts[k].size() / ((size_t)M + 1)), nSpC, pseudoC, eel.data(), model.mw.data(), nS,
                                      col0[k], samples.data(), l_bars.data());
        });
        for (int k = 0; k < nfiles; k++)
            if (!ok[k]) die("a sampled expression vector sums to less than EPSILON (the reference stops at an assert here, calcCI.cpp:135,143)");
        const double host_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        rc = rsem_ci_calculate_samples(device, M, (int32_t)nS, samples.data(), l_bars.data(), confidence, m, gi.starts.data(), m_trans,
                                       alleleS ? ta.starts.data() : nullptr, tpm.data(), fpkm.data(), gtpm.data(), gfpkm.data(),
                                       alleleS ? itpm.data() : nullptr, alleleS ? ifpkm.data() : nullptr, &prof);
        prof.sample_ms = host_s * 1e3;
        prof.n_draws = (uint64_t)M * nS;
        if (verbose) printf("[host] the reference's stream: %d engines, %.1f ms\n", nfiles, host_s * 1e3);
    }
    if (rc != RSEM_OK) die("rsem-calculate-credibility-intervals: %s: %s", rsem_hip_strerror(rc), rsem_hip_last_error());
    if (verbose) {
        printf("Sampling is finished!\n");
        printf("[device] %llu gamma draws %.1f ms, %llu keys sorted %.1f ms, intervals %.1f ms, total %.1f ms\n",
               (unsigned long long)prof.n_draws, prof.sample_ms, (unsigned long long)prof.n_keys_sorted, prof.sort_ms, prof.interval_ms,
               prof.total_ms);
    }

    append_rows(imdName + (alleleS ? ".allele_res" : ".iso_res"), tpm.data(), fpkm.data(), M);  // calcCI.cpp:443-457
    if (alleleS) append_rows(imdName + ".iso_res", itpm.data(), ifpkm.data(), m_trans);           // calcCI.cpp:459-476
    append_rows(imdName + ".gene_res", gtpm.data(), gfpkm.data(), m);                             // calcCI.cpp:478-494
    if (verbose) printf("All credibility intervals are calculated!\n");
    return 0;
}
