// rsem-run-em on MI355X: same argv, same files as the reference program (EM.cpp:541-675).
//
//   rsem-run-em refName read_type sampleName imdName statName [-p N] [-b samInpF has_fai [fai]] [-q]
//               [--gibbs-out] [--sampling] [--seed u32] [--append-names]     + ignored-by-the-reference: [--device d]
//
// Structure (EM<>() of EM.cpp:313-539): text inputs are parsed ONCE into packed arrays and uploaded;
// rounds 1-11 recompute the alignment probabilities with the current read model on the GPU
// (rsem_model_calc_conprb) and, in rounds 1-10, accumulate the model's sufficient statistics
// (rsem_model_estep_update); the O(table) renormalisation between rounds runs here on the host
// (model_host.hpp); from round 12 the device-resident loop rsem_em_run takes over.
#include <charconv>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
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

using namespace rsemh;

static const int MAX_ROUND = 10000, MIN_ROUND = 20;  // EM.cpp:53-55

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

struct ReadSetFiles {  // the three categories of reads (utils.h:129-149): un, alignable, max
    ReadFile mate[3][2];
    bool present[3] = {false, false, false};
};

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
    auto lap = [&](const char* what) {  // phase timing (stdout is not parsed by the pipeline driver)
        static auto last = std::chrono::steady_clock::now();
        auto now = std::chrono::steady_clock::now();
        if (getenv("RSEM_HIP_TIMING")) printf("[timing] %-28s %8.3f s\n", what, std::chrono::duration<double>(now - last).count());
        last = now;
    };
    const std::string refName = argv[1];
    const int read_type = atoi(argv[2]);
    const std::string outName = argv[3], imdName = argv[4], statName = argv[5];
    bool verbose = true, genBamF = false, genGibbsOut = false, appendNames = false, bamSampling = false, hasSeed = false;
    uint32_t seed = 0;
    std::string inpSamF;
    int device = 0;
    for (int i = 6; i < argc; i++) {  // EM.cpp:578-595; -p is accepted and irrelevant (the GPU is the parallelism)
        if (!strcmp(argv[i], "-b") && i + 1 < argc) { genBamF = true; inpSamF = argv[i + 1]; }
        if (!strcmp(argv[i], "--sampling")) bamSampling = true;
        if (!strcmp(argv[i], "--seed") && i + 1 < argc) {
            hasSeed = true;
            seed = 0;
            for (const char* c = argv[i + 1]; *c; ++c) seed = seed * 10 + (uint32_t)(*c - '0');
        }
        if (!strcmp(argv[i], "-q")) verbose = false;
        if (!strcmp(argv[i], "--gibbs-out")) genGibbsOut = true;
        if (!strcmp(argv[i], "--append-names")) appendNames = true;
        if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[i + 1]);
    }
    if (read_type < 0 || read_type > 3) die("Unknown Read Type!");
    // HIP runtime + device context come up (0.5 s) while the text inputs are parsed
    std::thread warm([device]() { rsem_hip_warmup(device); });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } warm_joiner{warm};

    RefInfo refs = load_refs(refName + ".seq", true);
    const int M = refs.M;
    if (verbose) printf("Refs.loadRefs finished!\n");
    Transcripts T = load_transcripts(refName + ".ti");
    if (T.M != M) die("%s.ti and %s.seq disagree on the number of transcripts!", refName.c_str(), refName.c_str());
    uint64_t N0, N1, N2, N_tot;
    load_cnt(statName + ".cnt", N0, N1, N2, N_tot);

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
    ModelParams P = load_mparams(imdName + ".mparams");
    const bool pe = read_type >= 2, hasQ = (read_type == 1 || read_type == 3);

    // ---- inputs, parsed once --------------------------------------------------------------------------
    DatData dat = load_dat(imdName + ".dat", read_type);
    if (dat.N1 != N1) die("Number of alignable reads does not match!");
    lap("parse .dat");
    // the EM context (CSR upload, device-side sort into the sliced layout) only needs the hits: build it while the
    // read files are parsed
    int ndev = 0;
    const uint64_t nnz = dat.sid_signed.size();
    std::vector<int32_t> sid_abs(nnz);
    rsem_em_ctx* em = nullptr;
    int em_rc = RSEM_OK;
    std::string em_err;
    std::thread em_builder([&]() {
        if (warm.joinable()) warm.join();
        rsem_hip_device_count(&ndev);
        if (ndev < 1) return;
        for (uint64_t j = 0; j < nnz; j++) {
            int32_t s = dat.sid_signed[j];
            sid_abs[j] = s < 0 ? -s : s;
            if (sid_abs[j] < 1 || sid_abs[j] > M) { em_rc = RSEM_ERR_INVALID; em_err = "transcript id " + std::to_string(s) + " out of range"; return; }
        }
        em_rc = rsem_em_create(&em, device, M, N1, nnz, dat.row_ptr.data(), sid_abs.data(), nullptr, nullptr);
        if (em_rc != RSEM_OK) em_err = rsem_hip_last_error();
    });
    Joiner em_joiner{em_builder};
    ReadSetFiles rs;
    const uint64_t Ncat[3] = {N0, N1, N2};
    for (int tag = 0; tag < 3; tag++) {
        if (Ncat[tag] == 0) continue;
        std::vector<std::string> names = read_file_names(imdName, tag, read_type);
        for (size_t m = 0; m < names.size(); m++) rs.mate[tag][m] = parse_read_file(names[m], hasQ, refs.has_polyA, P.seedLen);
        rs.present[tag] = true;
        if (rs.mate[tag][0].n != Ncat[tag]) die("%s holds %llu reads, %s.cnt says %llu!", names[0].c_str(),
                                                (unsigned long long)rs.mate[tag][0].n, statName.c_str(), (unsigned long long)Ncat[tag]);
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
    em_builder.join();
    if (ndev < 1) die("rsem-run-em: no usable GPU (this program has no CPU path)");
    if (em_rc != RSEM_OK) die("rsem-run-em: rsem_em_create: %s: %s (%s.dat)", rsem_hip_strerror(em_rc), em_err.c_str(), imdName.c_str());
    // packed references
    std::vector<uint64_t> ref_off(M + 2, 0), mask_off(M + 2, 0);
    for (int i = 1; i <= M; i++) {
        ref_off[i + 1] = ref_off[i] + refs.seq[i].size();
        mask_off[i + 1] = mask_off[i] + refs.masks[i].size();
        if ((int)refs.seq[i].size() != refs.totLen[i]) die("%s.seq: sequence %d has length %zu, header says %d", refName.c_str(), i, refs.seq[i].size(), refs.totLen[i]);
    }
    std::vector<uint8_t> ref_seq(ref_off[M + 1]);
    std::vector<uint32_t> mask_words(mask_off[M + 1]);
    const int8_t* tbl = base_table();
    for (int i = 1; i <= M; i++) {
        for (size_t k = 0; k < refs.seq[i].size(); k++) {
            int8_t id = tbl[(unsigned char)refs.seq[i][k]];
            if (id < 0) die("Found unknown sequence letter %c at function get_base_id!", refs.seq[i][k]);
            ref_seq[ref_off[i] + k] = (uint8_t)id;
        }
        std::copy(refs.masks[i].begin(), refs.masks[i].end(), mask_words.begin() + mask_off[i]);
    }
    rsem_model_data md;
    memset(&md, 0, sizeof(md));
    md.model_type = read_type; md.M = M; md.N1 = N1; md.nnz = nnz;
    md.row_ptr = dat.row_ptr.data(); md.sid_signed = dat.sid_signed.data(); md.pos = dat.pos.data();
    md.insertL = pe ? dat.insertL.data() : nullptr;
    for (int m = 0; m < (pe ? 2 : 1); m++) {
        md.read_off[m] = rs.mate[1][m].off.data();
        md.read_seq[m] = rs.mate[1][m].seq.data();
        md.read_qual[m] = hasQ ? rs.mate[1][m].qual.data() : nullptr;
    }
    md.low_quality = lq.data();
    md.ref_off = ref_off.data(); md.ref_seq = ref_seq.data(); md.fullLen = refs.fullLen.data(); md.totLen = refs.totLen.data();
    md.mask_off = mask_off.data(); md.mask_words = mask_words.data();
    rsem_model_ctx* mc = nullptr;
    hip_check(rsem_model_create(&mc, em, &md), "rsem_model_create");
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
    do {
        ++ROUND;
        const bool updateModel = ROUND <= 10;  // doesUpdateModel (EM.cpp:307-310)
        if (model.needCalcConPrb) {
            rsem_model_tables t = tables_of(model);
            hip_check(rsem_model_set_tables(mc, &t), "rsem_model_set_tables");
            hip_check(rsem_model_calc_conprb(mc), "rsem_model_calc_conprb");
            model.needCalcConPrb = false;  // EM.cpp:383
        }
        if (updateModel) {
            rsem_model_accum a;
            a.prof = acc.prof.data(); a.noise = acc.noise.data(); a.rspd = acc.rspd.data(); a.gld = acc.gld.data();
            a.gld0_lb = P.minL - 1; a.gld0_ub = P.maxL;
            hip_check(rsem_model_estep_update(mc, theta.data(), (double)N0, counts.data(), theta_new.data(), &sum, &bChange, &totNum, &a),
                      "rsem_model_estep_update");
            model.finish_round(acc, refs);  // model.init(); collect; finish  (EM.cpp:400-404)
        } else {
            hip_check(rsem_em_step(em, theta.data(), (double)N0, counts.data(), theta_new.data(), &sum, &bChange, &totNum), "rsem_em_step");
        }
        theta.swap(theta_new);
        if (verbose) printf("ROUND = %d, SUM = %.15g, bChange = %g, totNum = %d\n", ROUND, sum, bChange, totNum);
        if (ROUND >= 11 && !model.needCalcConPrb) break;  // the CSR values are frozen from here on
    } while (ROUND < MIN_ROUND || (totNum > 0 && ROUND < MAX_ROUND));
    lap("rounds 1-11 (model rounds)");
    if (ROUND < MIN_ROUND || (totNum > 0 && ROUND < MAX_ROUND)) {
        int rounds = ROUND;
        int32_t tn = 0;
        hip_check(rsem_em_run(em, theta.data(), (double)N0, ROUND, MIN_ROUND, MAX_ROUND, &rounds, counts.data(), &bChange, &tn, nullptr),
                  "rsem_em_run");
        ROUND = rounds;
        totNum = tn;
        if (verbose) printf("ROUND = %d, bChange = %g, totNum = %d\n", ROUND, bChange, totNum);
    }
    lap("rounds >= 12 (device loop)");
    if (totNum > 0) fprintf(stderr, "Warning: RSEM reaches %d iterations before meeting the convergence criteria.\n", MAX_ROUND);

    // ---- imd.ofg for the Gibbs sampler (EM.cpp:421-458) ---------------------------------------------------
    if (genGibbsOut) {
        std::vector<double> cp(nnz), ncp(N1);
        hip_check(rsem_model_get_values(mc, cp.data(), ncp.data()), "rsem_model_get_values");
        FILE* fo = fopen((imdName + ".ofg").c_str(), "w");
        if (!fo) die("Cannot open %s.ofg for writing!", imdName.c_str());
        fprintf(fo, "%d %llu\n", M, (unsigned long long)N0);
        // rows are formatted by all host threads into per-chunk buffers, then written in order
        const int nt = N1 > 100000 ? hardware_threads() : 1;
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
        for (auto& b : bufs) fwrite(b.data(), 1, b.size(), fo);
        fclose(fo);
    }
    lap("write .ofg");
    // ---- expected counts with the learned theta (EM.cpp:460-478) -------------------------------------------
    std::vector<double> w, w_noise;
    if (genBamF) { w.resize(nnz); w_noise.resize(N1); }
    hip_check(rsem_em_expected_weights(em, theta.data(), (double)N0, counts.data(), genBamF ? w.data() : nullptr,
                                       genBamF ? w_noise.data() : nullptr), "rsem_em_expected_weights");

    // ---- stat.theta (EM.cpp:484-500) ------------------------------------------------------------------------
    FILE* fo = fopen((statName + ".theta").c_str(), "w");
    if (!fo) die("Cannot open %s.theta for writing!", statName.c_str());
    fprintf(fo, "%d\n", M + 1);
    for (int i = 0; i < M; i++) fprintf(fo, "%.15g ", theta[i]);
    fprintf(fo, "%.15g\n", theta[M]);
    std::vector<double> eel = calc_eel(M, refs, model.gld);
    polish_theta(M, theta, eel, model.mw.data());
    for (int i = 0; i < M; i++) fprintf(fo, "%.15g ", theta[i]);
    fprintf(fo, "%.15g\n", theta[M]);
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
        write_transcript_bam(inpSamF, outName + ".transcript.bam", pe, sid_abs.data(), w.data(), nnz, T);
        if (verbose) printf("Bam output file is generated!\n");
        lap("transcript.bam");
    }
    rsem_model_destroy(mc);
    rsem_em_destroy(em);
    lap("device teardown");
    const auto secs = std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t_start).count();
    printf("Time Used for EM.cpp : %d h %02d m %02d s\n", (int)(secs / 3600), (int)(secs % 3600 / 60), (int)(secs % 60));
    if (getenv("RSEM_HIP_TIMING"))
        printf("[timing] %-28s %8.3f s\n", "main() total", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    // every output is closed: skip the destructors of the multi-GB host buffers and the runtime's atexit teardown
    fflush(stdout);
    fflush(stderr);
    _exit(0);
}
