// rsem-run-em on MI355X: same argv, same files as the reference program (EM.cpp:541-675).
//
//   rsem-run-em refName read_type sampleName imdName statName [-p N] [-b samInpF has_fai [fai]] [-q]
//               [--gibbs-out] [--sampling] [--seed u32] [--append-names]          + [--lean-device]
//               + ignored-by-the-reference: [--device d] [--ngpus N] [--devices d0,d1,..] [--value-bits 32 [--value-range-bits D]]
//
// Structure (EM<>() of EM.cpp:313-539): text inputs are parsed ONCE into packed arrays and uploaded;
// rounds 1-11 recompute the alignment probabilities with the current read model on the GPU
// (rsem_model_calc_conprb) and, in rounds 1-10, accumulate the model's sufficient statistics
// (rsem_model_estep_update); the O(table) renormalisation between rounds runs here on the host
// (model_host.hpp); from round 12 the device-resident loop rsem_em_run takes over.
// With --ngpus N the reads are split over N GPUs by the reference's own rule for its threads (EM.cpp:135-157,
// rsem_em_shard_rows): rounds 1-11 sum the shards' counts and model statistics on the host in shard order
// (EM.cpp:385-389,400-404), the device loop sums the counts of every round with one RCCL all-reduce.
#include <charconv>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <functional>
#include <memory>
#include <string>
#include <unistd.h>
#include <thread>
#include <vector>

#include "../../../include/rsem_hip.h"
#include "bam_io.hpp"
#include "files.hpp"
#include "model_host.hpp"
#include "reads.hpp"
#include "results.hpp"
#include "ofb.hpp"
#include "rsb.hpp"

using namespace rsemh;

static int MAX_ROUND = 10000;       // EM.cpp:54 (RSEM_HIP_MAX_ROUND lowers it for counter passes over the model rounds: tools/model_group_pmc.sh)
static const int MIN_ROUND = 20;    // EM.cpp:55

static void hip_check(int rc, const char* what) {
    if (rc != RSEM_OK) die("rsem-run-em: %s failed: %s (%s)", what, rsem_hip_strerror(rc), rsem_hip_last_error());
}

// boost::math::cdf(normal(mean, sd), x)
static double normal_cdf(double mean, double sd, double x) { return 0.5 * erfc(-(x - mean) / (sd * sqrt(2.0))); }

// LenDist::setAsNormal (LenDist.h:113-179)
static void set_as_normal(LenDist& d, double mean, double sd, int minL, int maxL) {
    const int meanL = int(mean + .5);
    if (sd < kEpsilon) {
        if (meanL < minL || meanL > maxL) die("Length distribution's probability mass is not within the possible range! MeanL = %d, MinL = %d, MaxL = %d", meanL, minL, maxL);
        d.span = 1; d.lb = meanL - 1; d.ub = meanL;
        d.pdf.assign(2, 0.0); d.cdf.assign(2, 0.0);
        d.pdf[1] = d.cdf[1] = 1.0;
        return;
    }
    if (maxL - minL + 1 > kRange) {
        if (meanL <= minL) maxL = minL + kRange - 1;
        else if (meanL >= maxL) minL = maxL - kRange + 1;
        else {
            double lg = mean - (minL - 0.5), rg = (maxL + 0.5) - mean, half = kRange / 2.0;
            if (lg < half) maxL = minL + kRange - 1;
            else if (rg < half) minL = maxL - kRange + 1;
            else { minL = int(mean - half + 1.0); maxL = int(mean + half); }
        }
    }
    d.lb = minL - 1; d.ub = maxL; d.span = d.ub - d.lb;
    d.pdf.assign(d.span + 1, 0.0); d.cdf.assign(d.span + 1, 0.0);
    double sum = 0.0, old_val = normal_cdf(mean, sd, minL - 0.5);
    for (int i = 1; i <= d.span; i++) {
        double val = normal_cdf(mean, sd, d.lb + i + 0.5);
        d.pdf[i] = val - old_val;
        sum += d.pdf[i];
        old_val = val;
    }
    for (int i = 1; i <= d.span; i++) { d.pdf[i] /= sum; d.cdf[i] = d.cdf[i - 1] + d.pdf[i]; }
    d.trim();
}

// *Model::estimateFromReads (SingleQModel.h:283-327, PairedEndQModel.h:241-290 and the no-Q twins)
static void estimate_from_reads(Model& model, const ReadSetFiles& rs, const RefInfo& refs, std::vector<uint8_t>& lq_alignable) {
    const bool pe = model.paired(), q = model.hasQ();
    LenDist& ld = (pe || model.has_mld) ? model.mld : model.gld;
    std::fill(ld.pdf.begin(), ld.pdf.end(), 0.0);
    std::fill(ld.cdf.begin(), ld.cdf.end(), 0.0);
    long n_warns = 0;
    for (int tag = 0; tag < 3; tag++) {
        if (!rs.present[tag]) continue;
        const ReadFile& a = rs.mate[tag][0];
        const ReadFile& b = rs.mate[tag][1];
        if (pe && a.n != b.n) die("Mate files of the %d-th read category have different numbers of reads!", tag);
        if (tag == 1) lq_alignable.assign(a.n, 0);
        // all statistics are integer counts: per-thread tables, merged afterwards (exact in any order)
        const int nt = a.n > 200000 ? hardware_threads() : 1;
        struct Local { std::vector<double> len, qi, qt, nc; double npc[5] = {0, 0, 0, 0, 0}; long warns = 0; std::string err; };
        std::vector<Local> loc(nt);
        parallel_for(nt, [&](int t) {
            Local& Lc = loc[t];
            Lc.len.assign(ld.pdf.size(), 0.0);
            if (q) { Lc.qi.assign(kQSize, 0.0); Lc.qt.assign((size_t)kQSize * kQSize, 0.0); Lc.nc.assign((size_t)kQSize * 5, 0.0); }
            const uint64_t lo = a.n * t / nt, hi = a.n * (t + 1) / nt;
            for (uint64_t i = lo; i < hi; i++) {
                bool lq;
                if (!pe) lq = a.lq1[i];
                else if (a.len(i) < model.P.seedLen || b.len(i) < model.P.seedLen) lq = true;  // PairedEndReadQ.h:55-62
                else lq = a.lq1[i] && b.lq1[i];
                if (tag == 1) lq_alignable[i] = lq ? 1 : 0;
                if (lq) {
                    if (a.len(i) < model.P.seedLen || (pe && b.len(i) < model.P.seedLen)) ++Lc.warns;
                    continue;
                }
                for (int m = 0; m < (pe ? 2 : 1); m++) {
                    const ReadFile& f = m ? b : a;
                    const int len = f.len(i);
                    if (!(len > ld.lb && len <= ld.ub)) {
                        char msg[200];
                        snprintf(msg, sizeof(msg), "A read of length %d is outside the length range (%d, %d] given to RSEM!", len, ld.lb, ld.ub);
                        Lc.err = msg;
                        return;
                    }
                    Lc.len[len - ld.lb] += 1.0;
                    const uint8_t* sq = f.seq.data() + f.off[i];
                    if (q) {
                        const uint8_t* ql = f.qual.data() + f.off[i];
                        Lc.qi[ql[0]] += 1.0;  // QualDist::update (QualDist.h:55-65)
                        for (int k = 1; k < len; k++) Lc.qt[ql[k - 1] * kQSize + ql[k]] += 1.0;
                        if (tag == 0)
                            for (int k = 0; k < len; k++) Lc.nc[ql[k] * 5 + sq[k]] += 1.0;  // NoiseQProfile::updateC
                    } else if (tag == 0) {
                        for (int k = 0; k < len; k++) Lc.npc[sq[k]] += 1.0;  // NoiseProfile::updateC
                    }
                }
            }
        });
        for (Local& Lc : loc) {
            if (!Lc.err.empty()) die("%s", Lc.err.c_str());
            n_warns += Lc.warns;
            for (size_t k = 0; k < Lc.len.size(); k++) ld.pdf[k] += Lc.len[k];
            if (q) {
                for (int k = 0; k < kQSize; k++) model.qd_init[k] += Lc.qi[k];
                for (size_t k = 0; k < Lc.qt.size(); k++) model.qd_tran[k] += Lc.qt[k];
                for (size_t k = 0; k < Lc.nc.size(); k++) model.nq_c[k] += Lc.nc[k];
            } else
                for (int k = 0; k < 5; k++) model.np_c[k] += Lc.npc[k];
        }
    }
    if (n_warns > 0) fprintf(stderr, "Warning: There are %ld reads ignored in total.\n", n_warns);
    ld.finish();
    if (!pe && model.P.mean >= kEpsilon)
        set_as_normal(model.gld, model.P.mean, model.P.sd, std::max(model.mld.minL(), model.gld.minL()), model.gld.maxL());
    if (q) {
        double sum = 0.0;  // QualDist::finish (QualDist.h:67-82)
        for (int i = 0; i < kQSize; i++) sum += model.qd_init[i];
        for (int i = 0; i < kQSize; i++) model.qd_init[i] /= sum;
        for (int i = 0; i < kQSize; i++) {
            sum = 0.0;
            for (int j = 0; j < kQSize; j++) sum += model.qd_tran[i * kQSize + j];
            if (sum <= 0.0) continue;
            for (int j = 0; j < kQSize; j++) model.qd_tran[i * kQSize + j] /= sum;
        }
        for (int i = 0; i < kQSize; i++) {  // NoiseQProfile::calcInitParams (NoiseQProfile.h:100-112)
            sum = 0.0;
            for (int j = 0; j < kNCodes; j++) sum += (1.0 + model.nq_c[i * 5 + j]);
            for (int j = 0; j < kNCodes; j++) model.nq_p[i * 5 + j] = (model.nq_c[i * 5 + j] + 1.0) / sum;
        }
    } else {
        double sum = 0.0;  // NoiseProfile::calcInitParams (NoiseProfile.h:84-95)
        for (int i = 0; i < kNCodes; i++) sum += (1.0 + model.np_c[i]);
        for (int i = 0; i < kNCodes; i++) model.np_p[i] = (1.0 + model.np_c[i]) / sum;
    }
    model.calc_mw(refs);
}

// name of the i-th record of a read file (only used on an error path)
static std::string read_name_at(const std::string& path, bool fastq, uint64_t i) {
    MappedFile f;
    if (!f.open(path)) return "#" + std::to_string(i);
    const uint64_t want = i * (fastq ? 4 : 2);
    const char* p = f.data;
    const char* e = f.data + f.size;
    for (uint64_t line = 0; p < e; line++) {
        const char* q = (const char*)memchr(p, '\n', e - p);
        if (!q) q = e;
        if (line == want) {
            std::string n(p + 1, q);
            while (!n.empty() && (n.back() == '\r' || n.back() == ' ')) n.pop_back();
            return n;
        }
        p = q + 1;
    }
    return "#" + std::to_string(i);
}

// The coordinate checks of getConPrb (SingleModel.h:108-114, SingleQModel.h:114-120, PairedEndModel.h:105-111,
// PairedEndQModel.h:109-115): the reference exits with these messages when an aligner reported inconsistent read
// lengths; the device kernels index reference sequences and masks with these coordinates, so they are checked here,
// once, before anything is uploaded.  Low-quality reads are skipped exactly as there (getConPrb returns before the
// assertions).
static void validate_alignments(const DatData& dat, const ReadFile* mates, const std::vector<uint8_t>& lq, const RefInfo& refs, int seedLen,
                                bool pe, const std::string& imdName, int read_type) {
    (void)seedLen;
    const uint64_t N1 = dat.N1;
    const int nt = N1 > 200000 ? hardware_threads() : 1;
    struct Bad { uint64_t read = ~0ull, hit = 0; int kind = 0; };
    std::vector<Bad> bad(nt);
    parallel_for(nt, [&](int t) {
        const uint64_t lo = N1 * t / nt, hi = N1 * (t + 1) / nt;
        for (uint64_t i = lo; i < hi && bad[t].read == ~0ull; i++) {
            if (lq[i]) continue;
            const int len1 = mates[0].len(i), len2 = pe ? mates[1].len(i) : 0;
            for (uint64_t k = dat.row_ptr[i]; k < dat.row_ptr[i + 1]; k++) {
                const int32_t sv = dat.sid_signed[k];
                const int sid = sv < 0 ? -sv : sv, dir = sv < 0 ? 1 : 0;
                const long long totLen = refs.totLen[sid], pos = dat.pos[k];
                const long long span = pe ? dat.insertL[k] : len1;
                const long long fpos = dir == 0 ? pos : totLen - pos - span;
                int kind = 0;
                if (fpos < 0) kind = 1;
                else if (fpos + span > totLen) kind = 2;
                else if (span > totLen) kind = 3;
                else if (pe && (span < len1 || span < len2)) kind = 4;
                if (kind) { bad[t].read = i; bad[t].hit = k; bad[t].kind = kind; break; }
            }
        }
    });
    for (const Bad& B : bad) {
        if (B.read == ~0ull) continue;
        std::string name = read_name_at(read_file_names(imdName, 1, read_type)[0], read_type == 1 || read_type == 3, B.read);
        if (pe && name.size() > 2 && name.compare(name.size() - 2, 2, "/1") == 0) name.resize(name.size() - 2);
        const int32_t sv = dat.sid_signed[B.hit];
        const int sid = sv < 0 ? -sv : sv;
        const long long totLen = refs.totLen[sid], pos = dat.pos[B.hit];
        const long long span = pe ? dat.insertL[B.hit] : mates[0].len(B.read);
        const long long fpos = sv < 0 ? totLen - pos - span : pos;
        const char* what = pe ? "fragment" : "read";
        const char* What = pe ? "Fragment" : "Read";
        static const char* hint = "It is possible that the aligner you use gave different read lengths for a same read in SAM file.";
        switch (B.kind) {
            case 1: die("The alignment of %s %s to transcript %d starts at %lld from the forward direction, which should be a non-negative number! %s",
                        what, name.c_str(), sid, fpos, hint);
            case 2: die("%s %s is hung over the end of transcript %d! %s", What, name.c_str(), sid, hint);
            case 3: die("%s %s has length %lld, but it is aligned to transcript %d, whose length (%lld) is shorter than the %s's length!",
                        What, name.c_str(), span, sid, totLen, what);
            default: die("Fragment %s has length %lld, which is shorter than one of its mates! %s", name.c_str(), span, hint);
        }
    }
}

static rsem_model_tables tables_of(const Model& m) {
    rsem_model_tables t;
    memset(&t, 0, sizeof(t));
    t.probF = m.probF;
    t.seedLen = m.P.seedLen;
    t.estRSPD = m.rspd.est ? 1 : 0;
    t.B = m.rspd.B;
    t.rspd_pdf = m.rspd.pdf.data();
    t.rspd_cdf = m.rspd.cdf.data();
    t.gld_lb = m.gld.lb; t.gld_ub = m.gld.ub; t.gld_pdf = m.gld.pdf.data(); t.gld_cdf = m.gld.cdf.data();
    t.has_mld = m.has_mld ? 1 : 0;
    if (m.has_mld) { t.mld_lb = m.mld.lb; t.mld_ub = m.mld.ub; t.mld_pdf = m.mld.pdf.data(); t.mld_cdf = m.mld.cdf.data(); }
    if (m.hasQ()) { t.prof_rows = kQSize; t.prof = m.qpro.data(); t.noise = m.nq_p.data(); }
    else { t.prof_rows = m.proLen; t.prof = m.pro.data(); t.noise = m.np_p; }
    t.mw = m.mw.data();
    return t;
}

int main(int argc, char* argv[]) {
    if (argc < 6) {
        printf("Usage : rsem-run-em refName read_type sampleName imdName statName [-p #Threads] [-b samInpF has_fai? [fai_file]] [-q] "
               "[--gibbs-out] [--sampling] [--seed seed] [--append-names] [--device d]\n\n");
        printf("// model parameters should be in imdName.mparams.\n");
        exit(-1);
    }
    const auto t_start = std::chrono::steady_clock::now();
    // (finer marks inside a phase, RSEM_HIP_TIMING=2: they do not reset the phase's clock)
    const auto t_mark0 = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        static const bool on = getenv("RSEM_HIP_TIMING") && atoi(getenv("RSEM_HIP_TIMING")) >= 2;
        if (on) printf("[timing]     at %7.3f s: %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_mark0).count(), what);
    };
    auto lap = [&](const char* what) {  // phase timing (stdout is not parsed by the pipeline driver)
        static auto last = t_start;  // (the first lap counts from the program's start)
        auto now = std::chrono::steady_clock::now();
        if (getenv("RSEM_HIP_TIMING")) printf("[timing] %-28s %8.3f s\n", what, std::chrono::duration<double>(now - last).count());
        last = now;
    };
    const std::string refName = argv[1];
    const int read_type = atoi(argv[2]);
    const std::string outName = argv[3], imdName = argv[4], statName = argv[5];
    // Gibbs hand-off (host/ofb.hpp): 0 = imdName.ofg (the reference's text), 1 = imdName.ofb/ (arrays), 2 = both; also
    // RSEM_HIP_BINARY=1 / =both in the environment, the switch the unmodified Perl driver cannot put on the command line
    int ofbMode = 0;
    if (const char* e = getenv("RSEM_HIP_MAX_ROUND")) MAX_ROUND = std::max(MIN_ROUND, atoi(e));
    if (const char* e = getenv("RSEM_HIP_BINARY")) ofbMode = !strcmp(e, "both") ? 2 : ((*e && strcmp(e, "0")) ? 1 : 0);
    bool verbose = true, genBamF = false, genGibbsOut = false, appendNames = false, bamSampling = false, hasSeed = false, leanDevice = false;
    uint32_t seed = 0;
    std::string inpSamF, devices_s;
    int device = 0, ngpus = 1, value_bits = 64, value_range_bits = -1, nThreads = 1;
    for (int i = 6; i < argc; i++) {  // EM.cpp:578-595; -p: the host threads of the -b pass (as the reference's hts_set_threads); the EM's parallelism is the GPU
        if (!strcmp(argv[i], "-p") && i + 1 < argc) nThreads = std::max(1, atoi(argv[i + 1]));
        if (!strcmp(argv[i], "-b") && i + 1 < argc) { genBamF = true; inpSamF = argv[i + 1]; }
        if (!strcmp(argv[i], "--sampling")) bamSampling = true;
        if (!strcmp(argv[i], "--seed") && i + 1 < argc) {
            hasSeed = true;
            seed = 0;
            for (const char* c = argv[i + 1]; *c; ++c) seed = seed * 10 + (uint32_t)(*c - '0');
        }
        if (!strcmp(argv[i], "-q")) verbose = false;
        if (!strcmp(argv[i], "--gibbs-out")) genGibbsOut = true;
        if (!strcmp(argv[i], "--gibbs-out-binary")) { genGibbsOut = true; ofbMode = 1; }  // not in the reference: imdName.ofb/ only
        if (!strcmp(argv[i], "--append-names")) appendNames = true;
        if (!strcmp(argv[i], "--lean-device")) leanDevice = true;  // the theta-only rounds hold the sliced layout alone in HBM
        if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--ngpus") && i + 1 < argc) ngpus = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--devices") && i + 1 < argc) devices_s = argv[i + 1];
        if (!strcmp(argv[i], "--value-bits") && i + 1 < argc) value_bits = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--value-range-bits") && i + 1 < argc) value_range_bits = atoi(argv[i + 1]);
    }
    if (value_bits != 64 && value_bits != 32) die("--value-bits must be 64 or 32");
    std::vector<int> devs;  // one entry per shard; a device named twice shares it between two shards (single-GPU testing)
    if (!devices_s.empty()) {
        for (size_t p = 0; p < devices_s.size();) {
            size_t q = devices_s.find(',', p);
            if (q == std::string::npos) q = devices_s.size();
            devs.push_back(atoi(devices_s.substr(p, q - p).c_str()));
            p = q + 1;
        }
    } else {
        for (int d = 0; d < std::max(1, ngpus); d++) devs.push_back(ngpus > 1 ? d : device);
    }
    device = devs[0];
    if (read_type < 0 || read_type > 3) die("Unknown Read Type!");
    // HIP runtime + device context come up (0.5 s) and the kernels' code objects are loaded while the text inputs are parsed
    std::thread warm([device]() { rsem_hip_preload(device, RSEM_PRELOAD_EM | RSEM_PRELOAD_MODEL); });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } warm_joiner{warm};

    uint64_t N0, N1, N2, N_tot;
    load_cnt(statName + ".cnt", N0, N1, N2, N_tot);
    // imd.dat depends on nothing but itself and is the longest of the text inputs: its parse starts before the reference is read (one
    // thread walks the .seq file, 0.3 s at 200 k transcripts, during which the host's other cores had nothing to do until round 6)
    const bool binary_in = rsb_present(imdName);
    DatData dat;
    int n_text_files = 1;
    for (int tag = 0; tag < 3; tag++)
        if ((tag == 0 ? N0 : tag == 1 ? N1 : N2) != 0) n_text_files += (int)read_file_names(imdName, tag, read_type).size();
    const int share = file_parse_threads(n_text_files);
    std::thread dat_parser;
    Joiner dat_joiner{dat_parser};
    if (!binary_in && N1 != 0) dat_parser = std::thread([&]() { dat = load_dat(imdName + ".dat", read_type, share); });
    RefInfo refs = load_refs(refName + ".seq", true);
    const int M = refs.M;
    if (verbose) printf("Refs.loadRefs finished!\n");
    Transcripts T = load_transcripts(refName + ".ti");
    if (T.M != M) die("%s.ti and %s.seq disagree on the number of transcripts!", refName.c_str(), refName.c_str());

    if (N1 == 0) {  // EM.cpp:615-638
        printf("Warning: There are no alignable reads!\n");
        fclose(fopen((statName + ".theta").c_str(), "w"));
        fclose(fopen((statName + ".model").c_str(), "w"));
        std::vector<double> theta(M + 1, 0.0), eel(M + 1, 0.0), countv(M + 1, 0.0);
        for (int i = 1; i <= M; ++i) eel[i] = T.t[i].length;
        write_results_em(M, refName, imdName, T, theta, eel, countv.data(), appendNames);
        if (genBamF) {  // EM.cpp:630-636: the input is copied
            std::string command = "cp " + inpSamF + " " + outName + ".transcript.bam";
            printf("%s\n", command.c_str());
            if (system(command.c_str()) != 0) die("Fail to copy %s!", inpSamF.c_str());
        }
        return 0;
    }

    lap("refs + transcripts");
    // packed references (base ids, mask words): they depend on nothing but the reference, so a thread of its own packs them while
    // the read and alignment files are parsed (0.1 s at GENCODE scale that used to stand between the parsers and the model context)
    std::vector<uint64_t> ref_off(M + 2, 0), mask_off(M + 2, 0);
    std::vector<uint8_t> ref_seq;
    std::vector<uint32_t> mask_words;
    std::thread ref_packer([&]() {
        for (int i = 1; i <= M; i++) {
            ref_off[i + 1] = ref_off[i] + refs.seq[i].size();
            mask_off[i + 1] = mask_off[i] + refs.masks[i].size();
            if ((int)refs.seq[i].size() != refs.totLen[i]) die("%s.seq: sequence %d has length %zu, header says %d", refName.c_str(), i, refs.seq[i].size(), refs.totLen[i]);
        }
        ref_seq.resize(ref_off[M + 1]);
        mask_words.resize(mask_off[M + 1]);
        const int8_t* tbl = base_table();
        const int nt = M > 2000 ? std::max(1, hardware_threads() / 4) : 1;  // (0.6 GB of letters at GENCODE scale: not a job for one thread; the parsers run beside it)
        parallel_for(nt, [&](int t) {
            for (int i = 1 + t; i <= M; i += nt) {
                for (size_t k = 0; k < refs.seq[i].size(); k++) {
                    int8_t id = tbl[(unsigned char)refs.seq[i][k]];
                    if (id < 0) die("Found unknown sequence letter %c at function get_base_id!", refs.seq[i][k]);
                    ref_seq[ref_off[i] + k] = (uint8_t)id;
                }
                std::copy(refs.masks[i].begin(), refs.masks[i].end(), mask_words.begin() + mask_off[i]);
            }
        });
    });
    Joiner ref_joiner{ref_packer};
    ModelParams P = load_mparams(imdName + ".mparams");
    const bool pe = read_type >= 2, hasQ = (read_type == 1 || read_type == 3);

    // ---- inputs, parsed once --------------------------------------------------------------------------
    // imdName.rsb/ (rsem-parse-alignments --binary, host/rsb.hpp): the arrays themselves, mapped; otherwise the
    // reference's text files
    ReadSetFiles rs;
    const uint64_t Ncat[3] = {N0, N1, N2};
    // Text inputs: imd.dat and the read files of every category are independent, so they are parsed at the same time, each
    // by its share of the host threads (at BASELINE configs[2] that is 30 GB of text: 7.7 GB of .dat, 2 x 11 GB of
    // alignable mates, 2 x 0.6 GB of unalignable ones).
    std::thread reads_parser;
    Joiner reads_joiner{reads_parser};
    if (binary_in) load_rsb(imdName, read_type, refs.has_polyA, P.seedLen, dat, rs);
    else {
        struct Job { int tag, m; std::string path; };
        std::vector<Job> jobs;
        for (int tag = 0; tag < 3; tag++) {
            if (Ncat[tag] == 0) continue;
            std::vector<std::string> names = read_file_names(imdName, tag, read_type);
            for (size_t m = 0; m < names.size(); m++) jobs.push_back(Job{tag, (int)m, names[m]});
            rs.present[tag] = true;
        }
        reads_parser = std::thread([&, jobs, share]() {
            std::vector<std::thread> th;
            for (const Job& J : jobs)
                th.emplace_back([&, J]() { rs.mate[J.tag][J.m] = parse_read_file(J.path, hasQ, refs.has_polyA, P.seedLen, share); });
            for (auto& t : th) t.join();
        });
        dat_parser.join();
    }
    if (dat.N1 != N1) die("Number of alignable reads does not match!");
    lap(binary_in ? "map .rsb" : "parse .dat");
    // the EM context (CSR upload, device-side sort into the sliced layout) only needs the hits: build it while the
    // read files are parsed
    int ndev = 0;
    const uint64_t nnz = dat.sid_signed.size();
    std::vector<int32_t> sid_abs(nnz);
    // shards: contiguous read ranges by the reference's thread rule (EM.cpp:135-157); never an empty one
    int S = (int)std::min<uint64_t>(devs.size(), N1);
    std::vector<uint64_t> bounds;
    for (;; --S) {
        bounds.assign(S + 1, 0);
        rsem_em_shard_rows(N1, dat.row_ptr.data(), S, bounds.data());
        bool empty = false;
        for (int k = 0; k < S; k++) empty = empty || bounds[k + 1] == bounds[k];
        if (!empty || S == 1) break;
    }
    devs.resize(S);
    struct Shard {
        int device = 0;
        uint64_t lo = 0, hi = 0, a = 0, b = 0;  // reads [lo, hi), alignments [a, b)
        std::vector<uint64_t> row_ptr;          // rebased copy (only when there is more than one shard)
        const uint64_t* rp = nullptr;
        rsem_em_ctx* em = nullptr;
        rsem_model_ctx* mc = nullptr;
        rsem_comm* comm = nullptr;
        int rc = RSEM_OK;
        std::string err;
    };
    std::vector<Shard> sh(S);
    for (int k = 0; k < S; k++) {
        Shard& X = sh[k];
        X.device = devs[k];
        X.lo = bounds[k]; X.hi = bounds[k + 1];
        X.a = dat.row_ptr[X.lo]; X.b = dat.row_ptr[X.hi];
        if (S == 1) X.rp = dat.row_ptr.data();
        else {
            X.row_ptr.resize(X.hi - X.lo + 1);
            for (uint64_t i = X.lo; i <= X.hi; i++) X.row_ptr[i - X.lo] = dat.row_ptr[i] - X.a;
            X.rp = X.row_ptr.data();
        }
    }
    auto each_shard = [&](const std::function<void(Shard&, int)>& fn) {  // one host thread per shard (= per GPU)
        if (S == 1) { fn(sh[0], 0); return; }
        std::vector<std::thread> th;
        for (int k = 0; k < S; k++) th.emplace_back([&, k]() { fn(sh[k], k); });
        for (auto& t : th) t.join();
    };
    auto check_shards = [&](const char* what) {
        for (int k = 0; k < S; k++)
            if (sh[k].rc != RSEM_OK)
                die("rsem-run-em: %s failed on shard %d (GPU %d): %s (%s)", what, k, sh[k].device, rsem_hip_strerror(sh[k].rc), sh[k].err.c_str());
    };
    std::thread em_builder([&]() {
        if (warm.joinable()) warm.join();
        rsem_hip_device_count(&ndev);
        if (ndev < 1) return;
        {   // |sid| of every alignment, range-checked (560 M of them at configs[2]: not a job for one thread)
            const int nt = nnz > (1u << 22) ? std::max(1, std::min(16, hardware_threads() / 4)) : 1;
            std::vector<int32_t> bad(nt, 0);
            parallel_for(nt, [&](int t) {
                const uint64_t lo = nnz * (uint64_t)t / (uint64_t)nt, hi = nnz * (uint64_t)(t + 1) / (uint64_t)nt;
                for (uint64_t j = lo; j < hi; j++) {
                    const int32_t v = dat.sid_signed[j];
                    sid_abs[j] = v < 0 ? -v : v;
                    if ((sid_abs[j] < 1 || sid_abs[j] > M) && !bad[t]) bad[t] = v ? v : INT32_MIN;
                }
            });
            for (int t = 0; t < nt; t++)
                if (bad[t]) { sh[0].rc = RSEM_ERR_INVALID; sh[0].err = "transcript id " + std::to_string(bad[t] == INT32_MIN ? 0 : bad[t]) + " out of range"; return; }
        }
        for (int k = 0; k < S; k++)
            if (sh[k].device < 0 || sh[k].device >= ndev) { sh[k].rc = RSEM_ERR_NODEVICE; sh[k].err = "no such GPU"; return; }
        each_shard([&](Shard& X, int) {
            X.rc = rsem_em_create(&X.em, X.device, M, X.hi - X.lo, X.b - X.a, X.rp, sid_abs.data() + X.a, nullptr, nullptr);
            if (X.rc != RSEM_OK) X.err = rsem_hip_last_error();
        });
    });
    Joiner em_joiner{em_builder};
    if (reads_parser.joinable()) reads_parser.join();
    for (int tag = 0; tag < 3; tag++) {
        if (Ncat[tag] == 0) continue;
        if (rs.mate[tag][0].n != Ncat[tag]) {
            if (binary_in) die("%s.rsb holds %llu reads of category %d, %s.cnt says %llu!", imdName.c_str(), (unsigned long long)rs.mate[tag][0].n, tag,
                               statName.c_str(), (unsigned long long)Ncat[tag]);
            die("%s holds %llu reads, %s.cnt says %llu!", read_file_names(imdName, tag, read_type)[0].c_str(), (unsigned long long)rs.mate[tag][0].n,
                statName.c_str(), (unsigned long long)Ncat[tag]);
        }
        if (verbose) printf("estimateFromReads, N%d finished.\n", tag);
    }
    lap("parse read files");
    Model model;
    model.init_master(read_type, M, P);
    std::vector<uint8_t> lq;
    estimate_from_reads(model, rs, refs, lq);
    lap("estimateFromReads");
    rs.mate[0][0] = ReadFile(); rs.mate[0][1] = ReadFile(); rs.mate[2][0] = ReadFile(); rs.mate[2][1] = ReadFile();

    // ---- device contexts -----------------------------------------------------------------------------
    validate_alignments(dat, rs.mate[1], lq, refs, model.P.seedLen, pe, imdName, read_type);
    mark("alignments validated");
    em_builder.join();
    mark("EM contexts built (upload + layout, started behind .dat)");
    if (ndev < 1) die("rsem-run-em: no usable GPU (this program has no CPU path)");
    check_shards("rsem_em_create");
    if (ref_packer.joinable()) ref_packer.join();
    mark("references packed");
    each_shard([&](Shard& X, int) {
        rsem_model_data md;
        memset(&md, 0, sizeof(md));
        md.model_type = read_type; md.M = M; md.N1 = X.hi - X.lo; md.nnz = X.b - X.a;
        md.row_ptr = X.rp; md.sid_signed = dat.sid_signed.data() + X.a; md.pos = dat.pos.data() + X.a;
        md.insertL = pe ? dat.insertL.data() + X.a : nullptr;
        for (int m = 0; m < (pe ? 2 : 1); m++) {  // offsets stay absolute: only differences and base + offset are used
            md.read_off[m] = rs.mate[1][m].off.data() + X.lo;
            md.read_seq[m] = rs.mate[1][m].seq.data();
            md.read_qual[m] = hasQ ? rs.mate[1][m].qual.data() : nullptr;
        }
        md.low_quality = lq.data() + X.lo;
        md.ref_off = ref_off.data(); md.ref_seq = ref_seq.data(); md.fullLen = refs.fullLen.data(); md.totLen = refs.totLen.data();
        md.mask_off = mask_off.data(); md.mask_words = mask_words.data();
        X.rc = rsem_model_create(&X.mc, X.em, &md);
        if (X.rc != RSEM_OK) X.err = rsem_hip_last_error();
    });
    check_shards("rsem_model_create");
    mark("model contexts built (reads, references, alignment fields uploaded)");
    if (getenv("RSEM_HIP_TIMING") && atoi(getenv("RSEM_HIP_TIMING")) >= 2) {
        int64_t b = 0, ns = 0, nf = 0, nw = 0;
        rsem_hip_device_info(device, "staged_bytes", &b); rsem_hip_device_info(device, "staged_ns", &ns);
        rsem_hip_device_info(device, "staged_fill_ns", &nf); rsem_hip_device_info(device, "staged_wait_ns", &nw);
        printf("[timing]     staged uploads so far: %.2f GB in %.3f s of calls (%.1f GB/s; filling the pinned buffers %.3f s, waiting for the DMA %.3f s)\n", b / 1e9, ns / 1e9,
               ns ? b / (double)ns : 0.0, nf / 1e9, nw / 1e9);
    }
    // the communicator of the device loop: RCCL, one rank per GPU; shards that share a GPU exchange inside the process
    if (S > 1) {
        bool shared_device = false;
        for (int x = 0; x < S; x++)
            for (int y = x + 1; y < S; y++) shared_device = shared_device || devs[x] == devs[y];
        if (shared_device) {
            std::vector<rsem_comm*> cs(S, nullptr);
            hip_check(rsem_comm_create_local(cs.data(), S, devs.data()), "rsem_comm_create_local");
            for (int k = 0; k < S; k++) sh[k].comm = cs[k];
        } else {
            char id[RSEM_COMM_ID_BYTES];
            hip_check(rsem_comm_unique_id(id), "rsem_comm_unique_id");
            each_shard([&](Shard& X, int k) {
                X.rc = rsem_comm_create(&X.comm, X.device, k, S, id);
                if (X.rc != RSEM_OK) X.err = rsem_hip_last_error();
            });
            check_shards("rsem_comm_create");
        }
        for (int k = 0; k < S; k++) hip_check(rsem_em_set_comm(sh[k].em, sh[k].comm), "rsem_em_set_comm");
        if (verbose)
            for (int k = 0; k < S; k++)  // the reference's "Thread i : N = .., NHit = .." (EM.cpp:155)
                printf("GPU %d : N = %llu, NHit = %llu\n", sh[k].device, (unsigned long long)(sh[k].hi - sh[k].lo), (unsigned long long)(sh[k].b - sh[k].a));
    }
    lap("device contexts + upload");
    if (verbose) printf("EM_init finished!\n");

    // ---- EM (EM.cpp:343-416) ---------------------------------------------------------------------------
    std::vector<double> theta(M + 1, 0.0), theta_new(M + 1, 0.0), counts(M + 1, 0.0);
    theta[0] = std::max(N0 * 1.0 / (N_tot - N2), 1e-8);
    for (int i = 1; i <= M; i++) theta[i] = (1.0 - theta[0]) / M;
    Model::Accum acc;
    acc.prof.assign(hasQ ? (size_t)kQSize * 25 : (size_t)model.proLen * 25, 0.0);
    acc.noise.assign(hasQ ? (size_t)kQSize * 5 : 5, 0.0);
    acc.rspd.assign((size_t)model.rspd.B + 2, 0.0);
    acc.gld.assign((size_t)(P.maxL - (P.minL - 1)) + 1, 0.0);
    int ROUND = 0, totNum = 0;
    double sum = 0.0, bChange = 0.0;
    // per-shard outputs of a round (more than one shard: summed on the host in shard order, EM.cpp:385-389,400-404)
    std::vector<std::vector<double>> s_counts(S > 1 ? S : 0, std::vector<double>(M + 1, 0.0));
    std::vector<Model::Accum> s_acc(S > 1 ? S : 0, acc);
    const bool round_times = getenv("RSEM_HIP_TIMING") != nullptr;
    auto t_round = std::chrono::steady_clock::now();
    do {
        ++ROUND;
        const bool updateModel = ROUND <= 10;  // doesUpdateModel (EM.cpp:307-310)
        const bool calc = model.needCalcConPrb;
        rsem_model_tables t = tables_of(model);
        each_shard([&](Shard& X, int k) {
            X.rc = RSEM_OK;
            if (calc) X.rc = rsem_model_set_tables(X.mc, &t);
            if (X.rc != RSEM_OK) { X.err = rsem_hip_last_error(); return; }
            // one shard: the device M step is the round's M step; several: raw counts (N0 = 0), reduced below
            double* cts = S == 1 ? counts.data() : s_counts[k].data();
            double s1 = 0.0, b1 = 0.0;
            int32_t t1 = 0;
            rsem_model_accum a;
            if (updateModel) {
                Model::Accum& A = S == 1 ? acc : s_acc[k];
                a.prof = A.prof.data(); a.noise = A.noise.data(); a.rspd = A.rspd.data(); a.gld = A.gld.data();
                a.gld0_lb = P.minL - 1; a.gld0_ub = P.maxL;
            }
            const double n0 = S == 1 ? (double)N0 : 0.0;
            double* thn = S == 1 ? theta_new.data() : nullptr;
            if (calc)  // the whole round in one pass over the reads: probabilities, weights, statistics (rsem_model_round)
                X.rc = rsem_model_round(X.mc, theta.data(), n0, cts, thn, &s1, &b1, &t1, updateModel ? &a : nullptr);
            else if (updateModel) X.rc = rsem_model_estep_update(X.mc, theta.data(), n0, cts, thn, &s1, &b1, &t1, &a);
            else X.rc = rsem_em_step(X.em, theta.data(), n0, cts, thn, &s1, &b1, &t1);
            if (X.rc != RSEM_OK) { X.err = rsem_hip_last_error(); return; }
            if (S == 1) { sum = s1; bChange = b1; totNum = t1; }
        });
        check_shards(calc ? "rsem_model_round" : (updateModel ? "rsem_model_estep_update" : "rsem_em_step"));
        model.needCalcConPrb = false;  // EM.cpp:383
        if (S > 1) {
            for (int j = 0; j <= M; j++) {
                double c = s_counts[0][j];
                for (int k = 1; k < S; k++) c += s_counts[k][j];
                counts[j] = c;
            }
            counts[0] += N0;  // EM.cpp:392
            sum = 0.0;
            for (int j = 0; j <= M; j++) sum += counts[j];
            if (!(sum >= kEpsilon)) die("rsem-run-em: the fractional counts sum to %g", sum);
            bChange = 0.0; totNum = 0;
            for (int j = 0; j <= M; j++) {
                theta_new[j] = counts[j] / sum;
                if (theta[j] >= 1e-7) {  // EM.cpp:406-413
                    const double change = fabs(theta_new[j] - theta[j]) / theta[j];
                    if (change >= 0.001) ++totNum;
                    if (bChange < change) bChange = change;
                }
            }
            if (updateModel) {
                for (size_t i = 0; i < acc.prof.size(); i++) { double v = s_acc[0].prof[i]; for (int k = 1; k < S; k++) v += s_acc[k].prof[i]; acc.prof[i] = v; }
                for (size_t i = 0; i < acc.noise.size(); i++) { double v = s_acc[0].noise[i]; for (int k = 1; k < S; k++) v += s_acc[k].noise[i]; acc.noise[i] = v; }
                for (size_t i = 0; i < acc.rspd.size(); i++) { double v = s_acc[0].rspd[i]; for (int k = 1; k < S; k++) v += s_acc[k].rspd[i]; acc.rspd[i] = v; }
                for (size_t i = 0; i < acc.gld.size(); i++) { double v = s_acc[0].gld[i]; for (int k = 1; k < S; k++) v += s_acc[k].gld[i]; acc.gld[i] = v; }
            }
        }
        if (updateModel) model.finish_round(acc, refs);  // model.init(); collect; finish  (EM.cpp:400-404)
        theta.swap(theta_new);
        if (verbose) printf("ROUND = %d, SUM = %.15g, bChange = %g, totNum = %d\n", ROUND, sum, bChange, totNum);
        if (round_times) {
            const auto now = std::chrono::steady_clock::now();
            printf("[timing]   model round %-2d            %8.3f s\n", ROUND, std::chrono::duration<double>(now - t_round).count());
            t_round = now;
        }
        if (ROUND >= 11 && !model.needCalcConPrb) break;  // the CSR values are frozen from here on
    } while (ROUND < MIN_ROUND || (totNum > 0 && ROUND < MAX_ROUND));
    lap("rounds 1-11 (model rounds)");
    // Everything the device needs has long been in HBM.  The parsed reads, alignment coordinates and packed references (tens
    // of GB at BASELINE sizes) go back to the system on a helper thread while the device loop runs -- the host only polls a
    // pinned mirror then -- instead of at exit, where unmapping them was 3 s of wall clock with nothing else left to do.
    // (Not during the model rounds: unmapping takes the address space's lock, and every call of those rounds that touches
    // memory waited for it -- about 1.5 s over 11 rounds at configs[2], profiles/r04b_call.log.  During the device loop it costs
    // 0.3-0.6 s and saves 1.3 s at exit: profiles/r04i_call.log.)  Kept: row_ptr and the transcript ids (.ofg / BAM output).
    std::thread releaser([&]() {
        if (getenv("RSEM_HIP_NO_RELEASE")) return;  // (measurement: everything stays until exit)
        {   // the pages first, on a few threads and without the address space's exclusive lock (files.hpp, give_back_pages)
            std::vector<std::pair<char*, size_t>> big;
            for (int tag = 0; tag < 3; tag++)
                for (int m = 0; m < 2; m++) { big.push_back(rs.mate[tag][m].off.owned_bytes()); big.push_back(rs.mate[tag][m].seq.owned_bytes()); big.push_back(rs.mate[tag][m].qual.owned_bytes()); }
            big.push_back(dat.pos.owned_bytes()); big.push_back(dat.insertL.owned_bytes()); big.push_back(dat.sid_signed.owned_bytes());
            give_back_pages(big, getenv("RSEM_HIP_RELEASE_THREADS") ? atoi(getenv("RSEM_HIP_RELEASE_THREADS")) : 8);
        }
        for (int tag = 0; tag < 3; tag++)
            for (int m = 0; m < 2; m++) rs.mate[tag][m] = ReadFile();
        dat.pos.release(); dat.insertL.release(); dat.sid_signed.release();
        std::vector<uint8_t>().swap(ref_seq);
        std::vector<uint32_t>().swap(mask_words);
        std::vector<uint8_t>().swap(lq);
    });
    Joiner release_joiner{releaser};
    if (leanDevice) {
        // Nothing but the sliced layout is read from here to the last round: the model contexts (the packed reads, their
        // alignments' coordinates, the reference strands) and the caller-order ids and values of the EM contexts go back to the
        // device's allocator; the values are read back from the planes by whatever asks for them afterwards (weights, .ofg).
        each_shard([&](Shard& X, int) {
            rsem_model_destroy(X.mc);
            X.mc = nullptr;
            if (rsem_em_set_option(X.em, "release_csr", 1) != RSEM_OK && verbose)
                printf("--lean-device: the caller-order arrays stay on the device (%s)\n", rsem_hip_last_error());
        });
        lap("lean device");
    }
    if (ROUND < MIN_ROUND || (totNum > 0 && ROUND < MAX_ROUND)) {
        const int round0 = ROUND;
        std::vector<std::vector<double>> s_theta(S, theta);
        std::vector<int> s_rounds(S, ROUND);
        std::vector<int32_t> s_tn(S, 0);
        std::vector<double> s_bc(S, 0.0);
        if (verbose)  // the reference's line after every round (EM.cpp:415), from the device's own record of the round
            hip_check(rsem_em_set_progress(sh[0].em, [](int r, double sm, double bc, int tn, void*) {
                printf("ROUND = %d, SUM = %.15g, bChange = %g, totNum = %d\n", r, sm, bc, tn);
            }, nullptr), "rsem_em_set_progress");
        if (value_bits == 32) {  // the values are frozen now: the theta-only rounds may stream them as Q32 planes (rsem_hip.h)
            each_shard([&](Shard& X, int) {
                X.rc = value_range_bits >= 0 ? rsem_em_set_option(X.em, "value_range_bits", value_range_bits) : RSEM_OK;
                if (X.rc == RSEM_OK) X.rc = rsem_em_set_option(X.em, "value_bits", 32);
                if (X.rc != RSEM_OK) X.err = rsem_hip_last_error();
            });
            check_shards("rsem_em_set_option(value_bits)");
            lap("Q32 value planes");
        }
        each_shard([&](Shard& X, int k) {
            X.rc = rsem_em_run(X.em, s_theta[k].data(), (double)N0, round0, MIN_ROUND, MAX_ROUND, &s_rounds[k], k == 0 ? counts.data() : nullptr,
                               &s_bc[k], &s_tn[k], nullptr);
            if (X.rc != RSEM_OK) X.err = rsem_hip_last_error();
        });
        check_shards("rsem_em_run");
        for (int k = 1; k < S; k++)
            if (s_rounds[k] != s_rounds[0]) die("rsem-run-em: shard %d stopped at round %d, shard 0 at round %d", k, s_rounds[k], s_rounds[0]);
        theta = s_theta[0];
        ROUND = s_rounds[0];
        totNum = s_tn[0];
        bChange = s_bc[0];
    }
    lap("rounds >= 12 (device loop)");
    if (totNum > 0) fprintf(stderr, "Warning: RSEM reaches %d iterations before meeting the convergence criteria.\n", MAX_ROUND);

    // ---- imd.ofg for the Gibbs sampler (EM.cpp:421-458) ---------------------------------------------------
    if (genGibbsOut) {
        std::unique_ptr<double[]> cp(new double[nnz ? nnz : 1]), ncp(new double[N1 ? N1 : 1]);  // filled by the copies below
        OfbHeader ofb_hdr;
        each_shard([&](Shard& X, int) {
            X.rc = X.mc ? rsem_model_get_values(X.mc, cp.get() + X.a, ncp.get() + X.lo) : rsem_em_get_values(X.em, cp.get() + X.a, ncp.get() + X.lo);
            if (X.rc != RSEM_OK) X.err = rsem_hip_last_error();
        });
        check_shards("rsem_model_get_values");
        if (ofbMode >= 1) {
            // the items as arrays; every value rounded through the 15-digit text form, so that .ofb == what .ofg parses to
            const int ntb = N1 > 100000 ? hardware_threads() : 1;
            std::vector<OfbPart> parts(ntb);
            parallel_for(ntb, [&](int t) {
                const uint64_t lo = N1 * t / ntb, hi = N1 * (t + 1) / ntb;
                OfbPart& P = parts[t];
                P.lens.reserve((size_t)(hi - lo));
                P.sid.reserve((size_t)(dat.row_ptr[hi] - dat.row_ptr[lo] + (hi - lo)));
                P.val.reserve((size_t)(dat.row_ptr[hi] - dat.row_ptr[lo] + (hi - lo)));
                for (uint64_t i = lo; i < hi; i++) {
                    uint32_t n = 0;
                    if (ncp[i] >= kEpsilon) { ++n; P.sid.push_back(0); P.val.push_back(through_15_digits(ncp[i])); }
                    for (uint64_t k = dat.row_ptr[i]; k < dat.row_ptr[i + 1]; k++)
                        if (cp[k] >= kEpsilon) { ++n; P.sid.push_back(sid_abs[k]); P.val.push_back(through_15_digits(cp[k])); }
                    if (n > 0) P.lens.push_back(n);  // (a read without items has no line in .ofg either)
                }
            });
            ofb_hdr = write_ofb_arrays(imdName, M, N0, parts);
            if (ofbMode == 1) write_ofb_header(imdName, ofb_hdr);  // (with a text file beside it: after that file is closed, below)
        } else {
            remove_ofb(imdName);  // never leave an older binary hand-off beside a fresh text one
        }
        const std::string ofg_path = imdName + ".ofg";
        if (ofbMode == 1) ::unlink(ofg_path.c_str());
        char head[64];
        const int head_n = snprintf(head, sizeof(head), "%d %llu\n", M, (unsigned long long)N0);
        // rows are formatted by all host threads into per-chunk buffers; every thread then writes its chunk at its own
        // offset of the file (the text is tens of GB at BASELINE sizes: one writer is page-cache bound)
        const int nt = ofbMode == 1 ? 0 : (N1 > 100000 ? hardware_threads() : 1);
        std::vector<std::string> bufs(nt);
        parallel_for(nt, [&](int t) {
            const uint64_t lo = N1 * t / nt, hi = N1 * (t + 1) / nt;
            std::string& b = bufs[t];
            b.reserve((size_t)((dat.row_ptr[hi] - dat.row_ptr[lo]) * 30 + (hi - lo) * 28));
            char tmp[64];
            auto put = [&](int sidv, double v) {  // "<sid> <%.15g> "  (ostream << setprecision(15), EM.cpp:445-452)
                auto r1 = std::to_chars(tmp, tmp + 16, sidv);
                *r1.ptr++ = ' ';
                auto r2 = std::to_chars(r1.ptr, tmp + 60, v, std::chars_format::general, 15);
                *r2.ptr++ = ' ';
                b.append(tmp, r2.ptr - tmp);
            };
            for (uint64_t i = lo; i < hi; i++) {
                int n = 0;
                if (ncp[i] >= kEpsilon) { ++n; put(0, ncp[i]); }
                for (uint64_t k = dat.row_ptr[i]; k < dat.row_ptr[i + 1]; k++)
                    if (cp[k] >= kEpsilon) { ++n; put(sid_abs[k], cp[k]); }
                if (n > 0) b.push_back('\n');
            }
        });
        std::vector<uint64_t> at(nt + 1, (uint64_t)head_n);
        for (int t = 0; t < nt; t++) at[t + 1] = at[t] + bufs[t].size();
        if (ofbMode != 1) {
        const int fd = ::open(ofg_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) die("Cannot open %s.ofg for writing!", imdName.c_str());
        bool wr_ok = ::pwrite(fd, head, head_n, 0) == head_n;
        std::vector<char> okv(nt, 1);
        parallel_for(nt, [&](int t) {
            const char* q = bufs[t].data();
            uint64_t left = bufs[t].size(), off = at[t];
            while (left > 0) {
                const ssize_t k = ::pwrite(fd, q, (size_t)std::min<uint64_t>(left, (uint64_t)1 << 30), (off_t)off);
                if (k <= 0) { okv[t] = 0; return; }
                q += k; off += (uint64_t)k; left -= (uint64_t)k;
            }
            bufs[t] = std::string();
        });
        for (char o : okv) wr_ok = wr_ok && o;
        if (::close(fd) != 0 || !wr_ok) die("Cannot write %s.ofg!", imdName.c_str());
        if (ofbMode == 2) write_ofb_header(imdName, ofb_hdr);  // not older than the text: rsem-run-gibbs takes the arrays (ofb_present)
        }
    }
    lap(ofbMode == 1 ? "write .ofb" : (ofbMode == 2 ? "write .ofg + .ofb" : "write .ofg"));
    // ---- expected counts with the learned theta (EM.cpp:460-478) -------------------------------------------
    std::vector<double> w, w_noise;
    if (genBamF) { w.resize(nnz); w_noise.resize(N1); }
    if (S == 1) {
        hip_check(rsem_em_expected_weights(sh[0].em, theta.data(), (double)N0, counts.data(), genBamF ? w.data() : nullptr,
                                           genBamF ? w_noise.data() : nullptr), "rsem_em_expected_weights");
    } else {
        each_shard([&](Shard& X, int k) {
            X.rc = rsem_em_expected_weights(X.em, theta.data(), 0.0, s_counts[k].data(), genBamF ? w.data() + X.a : nullptr,
                                            genBamF ? w_noise.data() + X.lo : nullptr);
            if (X.rc != RSEM_OK) X.err = rsem_hip_last_error();
        });
        check_shards("rsem_em_expected_weights");
        for (int j = 0; j <= M; j++) {
            double c = s_counts[0][j];
            for (int k = 1; k < S; k++) c += s_counts[k][j];
            counts[j] = c;
        }
        counts[0] += N0;
    }

    // ---- stat.theta (EM.cpp:484-500) ------------------------------------------------------------------------
    FILE* fo = fopen((statName + ".theta").c_str(), "w");
    if (!fo) die("Cannot open %s.theta for writing!", statName.c_str());
    fprintf(fo, "%d\n", M + 1);
    write_cells_line(fo, 0, M, ' ', [&](char* b, long i) { return snprintf(b, rsemh::kCellBuf, "%.15g", theta[i]); });
    std::vector<double> eel = calc_eel(M, refs, model.gld);
    polish_theta(M, theta, eel, model.mw.data());
    write_cells_line(fo, 0, M, ' ', [&](char* b, long i) { return snprintf(b, rsemh::kCellBuf, "%.15g", theta[i]); });
    fclose(fo);

    model.write(statName + ".model");
    write_results_em(M, refName, imdName, T, theta, eel, counts.data(), appendNames);
    if (verbose) printf("Expression Results are written!\n");

    lap("expected counts + results");
    if (genBamF) {  // EM.cpp:504-536
        if (bamSampling) {  // one alignment per read, drawn from its posterior (EM.cpp:507-531), MT19937 as sampling.h
            struct Mt {
                uint32_t mt[624]; int idx;
                explicit Mt(uint32_t s) { mt[0] = s; for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i; idx = 624; }
                uint32_t next() {
                    if (idx >= 624) {
                        for (int k = 0; k < 624; k++) {
                            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                        }
                        idx = 0;
                    }
                    uint32_t y = mt[idx++];
                    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
                    return y;
                }
            } engine(hasSeed ? seed : (uint32_t)time(NULL));
            if (verbose) printf("Begin to sample reads from their posteriors.\n");
            std::vector<double> arr;
            for (uint64_t i = 0; i < N1; i++) {
                const uint64_t fr = dat.row_ptr[i], to = dat.row_ptr[i + 1];
                const int len = (int)(to - fr + 1);
                arr.assign(len, 0.0);
                arr[0] = w_noise[i];
                for (uint64_t k = fr; k < to; k++) arr[k - fr + 1] = arr[k - fr] + w[k];
                long id = -1;
                if (!(arr[len - 1] < kEpsilon)) {  // sample() of sampling.h:50-65
                    const double prb = ((double)engine.next() * (1.0 / 4294967296.0)) * arr[len - 1];
                    int l = 0, r = len - 1;
                    while (l <= r) { int mid = (l + r) / 2; if (arr[mid] <= prb) l = mid + 1; else r = mid - 1; }
                    id = l;
                }
                for (uint64_t k = fr; k < to; k++) w[k] = ((long)(k - fr + 1) == id) ? 1.0 : 0.0;
            }
            if (verbose) printf("Sampling is finished.\n");
        }
        write_transcript_bam(inpSamF, outName + ".transcript.bam", pe, sid_abs.data(), w.data(), nnz, T, nThreads);  // (BamWriter.h:72: hts_set_threads(out, nThreads))
        if (verbose) printf("Bam output file is generated!\n");
        lap("transcript.bam");
    }
    for (Shard& X : sh) {
        rsem_model_destroy(X.mc);
        rsem_em_destroy(X.em);
        rsem_comm_destroy(X.comm);
    }
    lap("device teardown");
    const auto secs = std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t_start).count();
    printf("Time Used for EM.cpp : %d h %02d m %02d s\n", (int)(secs / 3600), (int)(secs % 3600 / 60), (int)(secs % 60));
    if (getenv("RSEM_HIP_TIMING"))
        printf("[timing] %-28s %8.3f s\n", "main() total", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    // every output is closed: skip the destructors of the multi-GB host buffers and the runtime's atexit teardown
    fflush(stdout);
    fflush(stderr);
    if (getenv("RSEM_HIP_NORMAL_EXIT")) return 0;  // profilers write their output from exit handlers
    _exit(0);
}
