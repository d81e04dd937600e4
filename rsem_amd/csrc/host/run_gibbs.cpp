// rsem-run-gibbs on MI355X: same argv, same files as the reference program (Gibbs.cpp:425-530).
//
//   rsem-run-gibbs refName imdName statName BURNIN NSAMPLES GAP [-p N] [--seed s] [--pseudo-count a]
//                  [--prior file] [-q]   + ignored-by-the-reference extras:
//                  [--gibbs-mode auto|exact|parallel] [--gibbs-thin k] [--device d | --devices d0,d1,..]
//
// -p N keeps its meaning "N independent chains, N count-vector files" (Gibbs.cpp:211-226, calcCI opens one
// file per thread).  The chains are dealt to the available GPUs; a GPU advances all of its chains together (one wave
// per chain) and the per-chain accumulators meet in one RCCL reduce (include/rsem_hip.h: rsem_gibbs_run_chains,
// rsem_comm_*).
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rsem_hip.h"
#include "files.hpp"
#include "ofb.hpp"
#include "model_host.hpp"
#include "posterior_moments.hpp"
#include "results.hpp"

using namespace rsemh;

int main(int argc, char* argv[]) {
    if (argc < 7) {
        printf("Usage: rsem-run-gibbs reference_name imdName statName BURNIN NSAMPLES GAP [-p #Threads] [--seed seed] "
               "[--pseudo-count pseudo_count] [--prior file] [-q] [--gibbs-mode auto|exact|parallel] [--gibbs-thin k] [--device d]\n");
        exit(-1);
    }
    const std::string refName = argv[1], imdName = argv[2], statName = argv[3];
    const int BURNIN = atoi(argv[4]), NSAMPLES = atoi(argv[5]), GAP = atoi(argv[6]);
    int nThreads = 1, thin = 0, device = -1;
    bool hasSeed = false, quiet = false, has_prior = false, dry_run = false;
    uint32_t seed = 0;
    double pseudoC = 1.0;
    std::string fprior, mode_s = "auto", devices_s;
    for (int i = 7; i < argc; i++) {  // order-insensitive strcmp scan, unknown tokens ignored (Gibbs.cpp:456-473)
        if (!strcmp(argv[i], "-p") && i + 1 < argc) nThreads = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--seed") && i + 1 < argc) {
            hasSeed = true;
            seed = 0;
            for (const char* c = argv[i + 1]; *c; ++c) seed = seed * 10 + (uint32_t)(*c - '0');
        }
        if (!strcmp(argv[i], "--pseudo-count") && i + 1 < argc) pseudoC = atof(argv[i + 1]);
        if (!strcmp(argv[i], "-q")) quiet = true;
        if (!strcmp(argv[i], "--prior") && i + 1 < argc) { has_prior = true; fprior = argv[i + 1]; }
        if (!strcmp(argv[i], "--gibbs-mode") && i + 1 < argc) mode_s = argv[i + 1];
        if (!strcmp(argv[i], "--gibbs-thin") && i + 1 < argc) thin = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--devices") && i + 1 < argc) devices_s = argv[i + 1];
        if (!strcmp(argv[i], "--dry-run")) dry_run = true;  // load the inputs, print the sampler that would run, exit (no GPU work)
    }
    const bool verbose = !quiet;
    if (NSAMPLES <= 1) die("NSAMPLES must be larger than 1, otherwise the posterior variance cannot be calculated!");
    if (nThreads > NSAMPLES) {
        nThreads = NSAMPLES;
        fprintf(stderr, "Warning: Number of samples is less than number of threads! Change the number of threads to %d!\n", nThreads);
    }
    if (nThreads < 1) nThreads = 1;

    // the HIP runtime, the device context and the sampler's code object come up while the items are read (not for --dry-run: no GPU work)
    std::thread warm;
    struct WarmJoin { std::thread& t; ~WarmJoin() { if (t.joinable()) t.join(); } } warm_join{warm};
    if (!dry_run) {
        int dev0 = device < 0 ? 0 : device;
        if (!devices_s.empty()) dev0 = atoi(devices_s.c_str());
        warm = std::thread([dev0]() { (void)rsem_hip_preload(dev0, RSEM_PRELOAD_GIBBS); });
    }
    // load_data (Gibbs.cpp:101-137)
    RefInfo refs = load_refs(refName + ".seq", false);
    const int M = refs.M;
    // imdName.ofb/ (rsem-run-em --gibbs-out with the binary hand-off, host/ofb.hpp): the same items, mapped; otherwise the
    // reference's text file
    OfgData ofg = ofb_present(imdName) ? load_ofb(imdName) : load_ofg(imdName + ".ofg");
    if (ofg.M != M) die("M in %s.ofg is not consistent with %s.seq!", imdName.c_str(), refName.c_str());
    const uint64_t N0 = ofg.N0, N1 = ofg.row_ptr.size() - 1;
    if (verbose) printf("Loading data is finished!\n");
    GroupInfo gi;
    if (!gi.load(refName + ".grp")) die("Cannot open %s.grp! It may not exist.", refName.c_str());
    const bool alleleS = is_allele_specific(refName);  // Gibbs.cpp:145-146
    GroupInfo gt, ta;
    if (alleleS && (!gt.load(refName + ".gt") || !ta.load(refName + ".ta"))) die("Cannot load %s.gt / %s.ta!", refName.c_str(), refName.c_str());
    const int m_trans = alleleS ? ta.m : 0;
    if (verbose) printf("Loading group information is finished!\n");
    // load_omit_info (Gibbs.cpp:152-167)
    std::vector<int32_t> init_counts(M + 1, 0);
    double totc = M + 1;
    {
        FILE* fi = fopen((imdName + ".omit").c_str(), "r");
        if (!fi) die("Cannot open %s.omit!", imdName.c_str());
        int tid;
        while (fscanf(fi, "%d", &tid) == 1) {
            if (tid < 0 || tid > M) die("%s.omit: transcript id %d out of range", imdName.c_str(), tid);
            init_counts[tid] = -1;
            --totc;
        }
        fclose(fi);
    }
    totc = totc * pseudoC + N0 + N1;
    // load_prior_info (Gibbs.cpp:171-194)
    std::vector<double> pseudo_counts;
    if (has_prior) {
        pseudo_counts.assign(M + 1, 0.0);
        FILE* fi = fopen(fprior.c_str(), "r");
        if (!fi) die("Cannot open %s!", fprior.c_str());
        char* line = nullptr;
        size_t cap = 0;
        for (int i = 1; i <= M; ++i) {
            double prior = 0.0;
            if (getline(&line, &cap, fi) >= 0) sscanf(line, "%lf", &prior);
            if (init_counts[i] == 0) pseudo_counts[i] = prior;
        }
        free(line);
        fclose(fi);
        totc = 1;
        for (int i = 1; i <= M; ++i)
            if (init_counts[i] == 0) totc += pseudo_counts[i];
        totc += N0 + N1;
    }
    // init_model_related (Gibbs.cpp:197-204): expected effective lengths and mask weights from stat.model
    Model model;
    model.read(statName + ".model", M);
    if ((int)model.mw.size() != M + 1) die("%s.model does not carry the mask weights of %d transcripts!", statName.c_str(), M);
    std::vector<double> eel = calc_eel(M, refs, model.gld);

    int ndev = 0;
    rsem_hip_device_count(&ndev);
    if (ndev < 1 && !dry_run) die("rsem-run-gibbs: no usable GPU (this program has no CPU path)");
    if (dry_run) ndev = std::max(ndev, 1);
    // GPUs that take part: --devices a,b,.. (a device may be named twice: its chains then form two groups, which is how
    // the multi-GPU code path is exercised on a one-GPU machine), --device d, or the first min(#GPUs, #chains) devices
    std::vector<int> devs;
    if (!devices_s.empty()) {
        for (size_t p = 0; p < devices_s.size();) {
            size_t q = devices_s.find(',', p);
            if (q == std::string::npos) q = devices_s.size();
            const int d = atoi(devices_s.substr(p, q - p).c_str());
            if (d < 0 || (d >= ndev && !dry_run)) die("rsem-run-gibbs: --devices names GPU %d, but there are %d", d, ndev);
            devs.push_back(d);
            p = q + 1;
        }
    } else if (device >= 0) {
        if (device >= ndev && !dry_run) die("rsem-run-gibbs: --device %d, but there are %d GPUs", device, ndev);
        devs.push_back(device);
    } else {
        for (int d = 0; d < std::min(ndev, nThreads); d++) devs.push_back(d);
    }
    if ((int)devs.size() > nThreads) devs.resize(nThreads);
    const int nworkers = (int)devs.size();
    // Sampler.  exact = the reference's collapsed chain (bit-identical count vectors), one wave per chain, all the chains
    // of a GPU advancing together; parallel = the data-augmentation sampler for the same posterior (every sweep fills
    // the GPU; the chains of a GPU run one after the other).
    int mode;
    // auto (the default): the reference's own chains (exact) unless they would take more than kAutoExactLimitS here.  A chain is
    // advanced by a team of W = min(64, compute units / chains of the GPU) workgroups (gibbs_exact_team.hpp); a read visit costs
    // about kTeamCyclesBase + kTeamCyclesPerW / W shader cycles per chain (measured at BASELINE configs[2]'s
    // shape on a 2.4 GHz MI355X: 0.0146 us with W = 32, 0.019 with 16, 0.028 with 8, 0.047 with 4) and kExactCyclesPerVisit with
    // one workgroup per chain (more chains than half the compute units); clock and compute units are the device's own
    // (rsem_hip_device_info).  The choice is printed and recorded in <statName>.gibbs_sampler; --gibbs-mode exact / parallel
    // overrule it.  Above the limit the DEFAULT outputs are not the reference's chains (same posterior, a different Markov
    // chain): INTEGRATION.md says so, and so does the program, on stdout and stderr.
    constexpr double kTeamCyclesBase = 27.0, kTeamCyclesPerW = 315.0;
    constexpr double kExactCyclesPerVisit = 195.0, kAutoExactLimitS = 900.0;
    int64_t dev_cus = 256, dev_khz = 2400000;
    if (!dry_run) {
        (void)rsem_hip_device_info(devs[0], "compute_units", &dev_cus);
        (void)rsem_hip_device_info(devs[0], "clock_khz", &dev_khz);
        if (dev_cus < 1) dev_cus = 256;
        if (dev_khz < 100000) dev_khz = 2400000;
    }
    const int rounds_per_chain = BURNIN + 1 + ((NSAMPLES + nThreads - 1) / nThreads - 1) * GAP;
    const int chains_per_gpu = (nThreads + nworkers - 1) / nworkers;
    int team_w = (int)std::min<int64_t>(64, dev_cus / std::max(1, chains_per_gpu));
    if (team_w < 8) team_w = 1;  // (gibbs.hip: smaller teams do not pay)
    const double cycles_per_visit = team_w > 1 ? kTeamCyclesBase + kTeamCyclesPerW / (double)team_w : kExactCyclesPerVisit;
    const double us_per_visit = cycles_per_visit / ((double)dev_khz * 1e-3);
    const double est_exact_s = (double)rounds_per_chain * (double)N1 * us_per_visit * 1e-6 * (double)((chains_per_gpu + dev_cus - 1) / dev_cus);
    if (mode_s == "exact") mode = RSEM_GIBBS_EXACT;
    else if (mode_s == "parallel") mode = RSEM_GIBBS_PARALLEL;
    else if (mode_s == "auto") {
        mode = est_exact_s <= kAutoExactLimitS ? RSEM_GIBBS_EXACT : RSEM_GIBBS_PARALLEL;
        if (mode == RSEM_GIBBS_PARALLEL) {
            fprintf(stderr, "rsem-run-gibbs: the reference's chains would take about %.0f s here (%d rounds x %llu reads); using the data-augmentation "
                            "sampler for the same posterior instead (--gibbs-mode exact forces the reference's chains)\n",
                    est_exact_s, rounds_per_chain, (unsigned long long)N1);
            if (verbose)
                printf("Gibbs sampler: data-augmentation (auto: the reference's own chains were estimated at %.0f s, limit %.0f s; "
                       "--gibbs-mode exact forces them)\n", est_exact_s, kAutoExactLimitS);
        }
    } else die("rsem-run-gibbs: unknown --gibbs-mode '%s' (exact, parallel or auto)", mode_s.c_str());
    if (!dry_run) {
        if (FILE* fs = fopen((statName + ".gibbs_sampler").c_str(), "w")) {
            fprintf(fs, "sampler %s\nrequested %s\nestimated_exact_seconds %.1f\nrounds_per_chain %d\nchains %d\ngpu_groups %d\n",
                    mode == RSEM_GIBBS_EXACT ? "exact" : "parallel", mode_s.c_str(), est_exact_s, rounds_per_chain, nThreads, nworkers);
            fclose(fs);
        }
    }
    if (thin <= 0) thin = (mode == RSEM_GIBBS_PARALLEL) ? 8 : 1;
    if (mode == RSEM_GIBBS_PARALLEL)  // never silently: these draws are a different Markov chain than the reference's
        fprintf(stderr, "rsem-run-gibbs: data-augmentation sampler (--gibbs-mode parallel), %d sweeps per round\n", thin);
    if (dry_run) {
        printf("dry run: %s sampler, %d chain(s) on %d GPU group(s), %d sweep(s) per round, N1 = %llu, M = %d\n",
               mode == RSEM_GIBBS_EXACT ? "exact" : "parallel", nThreads, nworkers, mode == RSEM_GIBBS_PARALLEL ? thin : 1,
               (unsigned long long)N1, M);
        return 0;
    }
    if (verbose) printf("Gibbs started! (%s sampler, %d chain(s), %d GPU(s))\n", mode == RSEM_GIBBS_EXACT ? "exact" : "parallel", nThreads, nworkers);

    // chain seeds: engineFactory (sampling.h:19-44); without --seed the reference seeds from time(NULL)
    std::vector<uint32_t> seeds(nThreads);
    rsem_gibbs_chain_seeds(hasSeed ? seed : (uint32_t)time(NULL), nThreads, seeds.data());

    const int quotient = NSAMPLES / nThreads, left = NSAMPLES % nThreads;  // Gibbs.cpp:215-223
    // Chains are dealt round-robin to the GPU groups; a group runs its chains in ONE rsem_gibbs_run_chains call and the
    // groups' accumulator sums meet in a single reduce to group 0 (RCCL over xGMI; release(), Gibbs.cpp:372-388).
    bool shared_device = false;
    for (int a = 0; a < nworkers; a++)
        for (int b2 = a + 1; b2 < nworkers; b2++) shared_device = shared_device || devs[a] == devs[b2];
    std::vector<rsem_comm*> comms(nworkers, nullptr);
    char comm_id[RSEM_COMM_ID_BYTES];
    const bool want_comm = nworkers > 1 || getenv("RSEM_HIP_FORCE_COMM");
    if (want_comm) {
        if (shared_device) {
            int rc = rsem_comm_create_local(comms.data(), nworkers, devs.data());
            if (rc != RSEM_OK) die("rsem-run-gibbs: rsem_comm_create_local: %s (%s)", rsem_hip_strerror(rc), rsem_hip_last_error());
        } else {
            int rc = rsem_comm_unique_id(comm_id);
            if (rc != RSEM_OK) die("rsem-run-gibbs: rsem_comm_unique_id: %s (%s)", rsem_hip_strerror(rc), rsem_hip_last_error());
        }
    }
    std::vector<double> pme_c(M + 1, 0.0), pve_c(M + 1, 0.0), pme_tpm(M + 1, 0.0), pme_fpkm(M + 1, 0.0), pve_c_genes(gi.m, 0.0);
    std::vector<double> pve_c_trans(m_trans, 0.0);
    std::vector<std::string> errors(nworkers);
    std::vector<std::thread> workers;
    for (int w = 0; w < nworkers; w++) {
        workers.emplace_back([&, w]() {
            const int dev = devs[w];
            auto fail = [&](const char* what, int rc) { errors[w] = std::string(what) + ": " + rsem_hip_strerror(rc) + ": " + rsem_hip_last_error(); };
            int rc = RSEM_OK;
            if (want_comm && !shared_device) {  // collective: every group calls it
                rc = rsem_comm_create(&comms[w], dev, w, nworkers, comm_id);
                if (rc != RSEM_OK) { fail("rsem_comm_create", rc); return; }
            }
            rsem_gibbs_ctx* g = nullptr;
            rc = rsem_gibbs_create(&g, dev, M, N1, ofg.sid.size(), ofg.row_ptr.data(), ofg.sid.data(), ofg.conprb.data(),
                                   init_counts.data(), has_prior ? pseudo_counts.data() : nullptr, pseudoC, totc, N0,
                                   eel.data(), model.mw.data(), gi.m, gi.starts.data());
            if (rc == RSEM_OK && alleleS) rc = rsem_gibbs_set_allele_groups(g, m_trans, ta.starts.data());
            if (rc == RSEM_OK && comms[w]) rc = rsem_gibbs_set_comm(g, comms[w]);
            if (rc != RSEM_OK) { fail("rsem_gibbs_create", rc); return; }
            std::vector<int> mine;
            for (int k = w; k < nThreads; k += nworkers) mine.push_back(k);
            std::vector<uint32_t> my_seeds;
            std::vector<int32_t> my_ns;
            std::vector<std::vector<int32_t>> cv(mine.size());
            std::vector<int32_t*> cv_ptr;
            for (size_t j = 0; j < mine.size(); j++) {
                const int k = mine[j];
                my_seeds.push_back(seeds[k]);
                my_ns.push_back(quotient + (k < left ? 1 : 0));
                cv[j].resize((size_t)my_ns.back() * (M + 1));
                cv_ptr.push_back(cv[j].data());
            }
            // group 0 receives the totals; the other groups' outputs are scratch
            std::vector<double> s0, s1, s2, s3, s4, s5;
            const bool root = (w == 0);
            if (!root) { s0.resize(M + 1); s1.resize(M + 1); s2.resize(M + 1); s3.resize(M + 1); s4.resize(gi.m); s5.resize(m_trans); }
            rc = rsem_gibbs_run_chains(g, mode, (int)mine.size(), my_seeds.data(), BURNIN, my_ns.data(), GAP, thin, cv_ptr.data(),
                                       root ? pme_c.data() : s0.data(), root ? pve_c.data() : s1.data(), root ? pme_tpm.data() : s2.data(),
                                       root ? pme_fpkm.data() : s3.data(), root ? pve_c_genes.data() : s4.data(),
                                       alleleS ? (root ? pve_c_trans.data() : s5.data()) : nullptr, nullptr);
            if (rc != RSEM_OK) { fail("rsem_gibbs_run_chains", rc); rsem_gibbs_destroy(g); return; }
            // writeCountVector (Gibbs.cpp:257-262): one file per chain, written by the group that ran it
            for (size_t j = 0; j < mine.size(); j++) {
                const int k = mine[j];
                FILE* fo = fopen((imdName + ".countvectors" + std::to_string(k)).c_str(), "w");
                if (!fo) { errors[w] = "cannot write count vectors"; break; }
                std::string line;
                line.reserve((size_t)(M + 1) * 8);
                char tmp[16];
                for (int s = 0; s < my_ns[j]; s++) {
                    const int32_t* c = cv[j].data() + (size_t)s * (M + 1);
                    line.clear();
                    for (int i = 0; i <= M; i++) {
                        auto r = std::to_chars(tmp, tmp + sizeof(tmp), c[i]);
                        line.append(tmp, r.ptr - tmp);
                        line.push_back(i < M ? ' ' : '\n');
                    }
                    fwrite(line.data(), 1, line.size(), fo);
                }
                fclose(fo);
                if (verbose) printf("Chain %d is finished!\n", k);
            }
            rsem_gibbs_destroy(g);
        });
    }
    for (auto& t : workers) t.join();
    for (auto* cm : comms) rsem_comm_destroy(cm);
    for (auto& e : errors)
        if (!e.empty()) die("rsem-run-gibbs: %s", e.c_str());

    // the accumulators hold sums over all kept samples: means and variances (posterior_moments.hpp; Gibbs.cpp:389-423)
    rsem_host::finish_per_transcript(NSAMPLES, pme_c, pve_c);
    rsem_host::finish_means(NSAMPLES, pme_tpm);
    rsem_host::finish_means(NSAMPLES, pme_fpkm);
    rsem_host::finish_per_group(NSAMPLES, pme_c, gi.starts, pve_c_genes);
    if (alleleS) rsem_host::finish_per_group(NSAMPLES, pme_c, ta.starts, pve_c_trans);
    if (verbose) printf("Gibbs finished!\n");
    write_results_gibbs(M, gi, imdName, pme_c, pme_fpkm, pme_tpm, pve_c, pve_c_genes, alleleS, &gt, &ta, &pve_c_trans);
    if (verbose) printf("Gibbs based expression values are written!\n");
    return 0;
}
