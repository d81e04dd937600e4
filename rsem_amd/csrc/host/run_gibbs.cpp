// rsem-run-gibbs on MI355X: same argv, same files as the reference program (Gibbs.cpp:425-530).
//
//   rsem-run-gibbs refName imdName statName BURNIN NSAMPLES GAP [-p N] [--seed s] [--pseudo-count a]
//                  [--prior file] [-q]   + ignored-by-the-reference extras:
//                  [--gibbs-mode exact|parallel] [--gibbs-thin k] [--device d]
//
// -p N keeps its meaning "N independent chains, N count-vector files" (Gibbs.cpp:211-226, calcCI opens one
// file per thread); the chains run on the available GPUs through librsem_hip (include/rsem_hip.h).
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
#include "model_host.hpp"
#include "results.hpp"

using namespace rsemh;

int main(int argc, char* argv[]) {
    if (argc < 7) {
        printf("Usage: rsem-run-gibbs reference_name imdName statName BURNIN NSAMPLES GAP [-p #Threads] [--seed seed] "
               "[--pseudo-count pseudo_count] [--prior file] [-q] [--gibbs-mode exact|parallel] [--gibbs-thin k] [--device d]\n");
        exit(-1);
    }
    const std::string refName = argv[1], imdName = argv[2], statName = argv[3];
    const int BURNIN = atoi(argv[4]), NSAMPLES = atoi(argv[5]), GAP = atoi(argv[6]);
    int nThreads = 1, thin = 0, device = -1;
    bool hasSeed = false, quiet = false, has_prior = false, dry_run = false;
    uint32_t seed = 0;
    double pseudoC = 1.0;
    std::string fprior, mode_s = "auto";
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
        if (!strcmp(argv[i], "--dry-run")) dry_run = true;  // load the inputs, print the sampler that would run, exit (no GPU work)
    }
    const bool verbose = !quiet;
    if (NSAMPLES <= 1) die("NSAMPLES must be larger than 1, otherwise the posterior variance cannot be calculated!");
    if (nThreads > NSAMPLES) {
        nThreads = NSAMPLES;
        fprintf(stderr, "Warning: Number of samples is less than number of threads! Change the number of threads to %d!\n", nThreads);
    }
    if (nThreads < 1) nThreads = 1;

    // load_data (Gibbs.cpp:101-137)
    RefInfo refs = load_refs(refName + ".seq", false);
    const int M = refs.M;
    OfgData ofg = load_ofg(imdName + ".ofg");
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
    // The reference's own chain (EXACT: bit-identical count vectors) is one wave walking the reads in order, ~0.4 us per
    // read and round, and the chains of one GPU run one after the other -- so its cost is read-rounds per GPU.
    const double chain_rows = (double)N1 * (BURNIN + 1.0 + std::ceil(NSAMPLES / (double)nThreads) * GAP);
    const int gpus = device >= 0 ? 1 : std::max(1, std::min(ndev, nThreads));
    const double rows_per_gpu = chain_rows * std::ceil(nThreads / (double)gpus);
    const double kExactBudget = 2.5e7;  // ~10 s
    int mode;
    if (mode_s == "exact") {
        mode = RSEM_GIBBS_EXACT;
        if (rows_per_gpu > kExactBudget)  // honoured, but never silently: this can be hours
            fprintf(stderr, "Warning: --gibbs-mode exact on %.3g read-rounds per GPU will take about %.0f s (one wave per chain, chains in "
                            "sequence); --gibbs-mode parallel samples the same posterior in a fraction of that.\n",
                    rows_per_gpu, rows_per_gpu * 0.4e-6);
    } else if (mode_s == "parallel") mode = RSEM_GIBBS_PARALLEL;
    else mode = rows_per_gpu <= kExactBudget ? RSEM_GIBBS_EXACT : RSEM_GIBBS_PARALLEL;  // auto: exact only when it is cheap
    if (thin <= 0) thin = (mode == RSEM_GIBBS_PARALLEL) ? 8 : 1;
    if (dry_run) {
        printf("dry run: %s sampler, %d chain(s), %d sweep(s) per round, N1 = %llu, M = %d\n", mode == RSEM_GIBBS_EXACT ? "exact" : "parallel",
               nThreads, mode == RSEM_GIBBS_PARALLEL ? thin : 1, (unsigned long long)N1, M);
        return 0;
    }
    if (verbose) printf("Gibbs started! (%s sampler, %d chain(s), %d GPU(s))\n", mode == RSEM_GIBBS_EXACT ? "exact" : "parallel", nThreads, ndev);

    // chain seeds: engineFactory (sampling.h:19-44); without --seed the reference seeds from time(NULL)
    std::vector<uint32_t> seeds(nThreads);
    rsem_gibbs_chain_seeds(hasSeed ? seed : (uint32_t)time(NULL), nThreads, seeds.data());

    const int quotient = NSAMPLES / nThreads, left = NSAMPLES % nThreads;  // Gibbs.cpp:215-223
    std::vector<std::vector<double>> acc(nThreads * 4, std::vector<double>(M + 1, 0.0));
    std::vector<std::vector<double>> acc_g(nThreads, std::vector<double>(gi.m, 0.0));
    std::vector<std::vector<double>> acc_t(nThreads, std::vector<double>(m_trans, 0.0));
    std::vector<std::string> errors(nThreads);
    const int nworkers = device >= 0 ? 1 : std::min(ndev, nThreads);
    std::vector<std::thread> workers;
    for (int w = 0; w < nworkers; w++) {
        workers.emplace_back([&, w]() {
            const int dev = device >= 0 ? device : w;
            rsem_gibbs_ctx* g = nullptr;
            int rc = rsem_gibbs_create(&g, dev, M, N1, ofg.sid.size(), ofg.row_ptr.data(), ofg.sid.data(), ofg.conprb.data(),
                                       init_counts.data(), has_prior ? pseudo_counts.data() : nullptr, pseudoC, totc, N0,
                                       eel.data(), model.mw.data(), gi.m, gi.starts.data());
            if (rc == RSEM_OK && alleleS) rc = rsem_gibbs_set_allele_groups(g, m_trans, ta.starts.data());
            if (rc != RSEM_OK) { errors[w] = std::string(rsem_hip_strerror(rc)) + ": " + rsem_hip_last_error(); return; }
            for (int k = w; k < nThreads; k += nworkers) {
                const int ns = quotient + (k < left ? 1 : 0);
                std::vector<int32_t> cv((size_t)ns * (M + 1));
                rc = rsem_gibbs_run(g, mode, seeds[k], BURNIN, ns, GAP, thin, cv.data(), acc[k * 4 + 0].data(), acc[k * 4 + 1].data(),
                                    acc[k * 4 + 2].data(), acc[k * 4 + 3].data(), acc_g[k].data(), nullptr);
                if (rc == RSEM_OK && alleleS) rc = rsem_gibbs_get_pve_c_trans(g, acc_t[k].data());
                if (rc != RSEM_OK) { errors[w] = std::string(rsem_hip_strerror(rc)) + ": " + rsem_hip_last_error(); break; }
                // writeCountVector (Gibbs.cpp:257-262): one file per chain
                FILE* fo = fopen((imdName + ".countvectors" + std::to_string(k)).c_str(), "w");
                if (!fo) { errors[w] = "cannot write count vectors"; break; }
                std::string line;
                line.reserve((size_t)(M + 1) * 8);
                char tmp[16];
                for (int s = 0; s < ns; s++) {
                    const int32_t* c = cv.data() + (size_t)s * (M + 1);
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
    for (auto& e : errors)
        if (!e.empty()) die("rsem-run-gibbs: %s", e.c_str());

    // release() (Gibbs.cpp:355-423)
    std::vector<double> pme_c(M + 1, 0.0), pve_c(M + 1, 0.0), pme_tpm(M + 1, 0.0), pme_fpkm(M + 1, 0.0), pve_c_genes(gi.m, 0.0);
    for (int k = 0; k < nThreads; k++) {
        for (int j = 0; j <= M; j++) {
            pme_c[j] += acc[k * 4 + 0][j];
            pve_c[j] += acc[k * 4 + 1][j];
            pme_tpm[j] += acc[k * 4 + 2][j];
            pme_fpkm[j] += acc[k * 4 + 3][j];
        }
        for (int j = 0; j < gi.m; j++) pve_c_genes[j] += acc_g[k][j];
    }
    for (int i = 0; i <= M; i++) {
        pme_c[i] /= NSAMPLES;
        pve_c[i] = (pve_c[i] - double(NSAMPLES) * pme_c[i] * pme_c[i]) / double(NSAMPLES - 1);
        if (pve_c[i] < 0.0) pve_c[i] = 0.0;
        pme_tpm[i] /= NSAMPLES;
        pme_fpkm[i] /= NSAMPLES;
    }
    for (int i = 0; i < gi.m; i++) {
        double pme_c_gene = 0.0;
        for (int j = gi.starts[i]; j < gi.starts[i + 1]; j++) pme_c_gene += pme_c[j];
        pve_c_genes[i] = (pve_c_genes[i] - double(NSAMPLES) * pme_c_gene * pme_c_gene) / double(NSAMPLES - 1);
        if (pve_c_genes[i] < 0.0) pve_c_genes[i] = 0.0;
    }
    std::vector<double> pve_c_trans(m_trans, 0.0);
    if (alleleS) {
        for (int k = 0; k < nThreads; k++)
            for (int j = 0; j < m_trans; j++) pve_c_trans[j] += acc_t[k][j];
        for (int i = 0; i < m_trans; i++) {
            double pme_c_tran = 0.0;
            for (int j = ta.starts[i]; j < ta.starts[i + 1]; j++) pme_c_tran += pme_c[j];
            pve_c_trans[i] = (pve_c_trans[i] - double(NSAMPLES) * pme_c_tran * pme_c_tran) / double(NSAMPLES - 1);
            if (pve_c_trans[i] < 0.0) pve_c_trans[i] = 0.0;
        }
    }
    if (verbose) printf("Gibbs finished!\n");
    write_results_gibbs(M, gi, imdName, pme_c, pme_fpkm, pme_tpm, pve_c, pve_c_genes, alleleS, &gt, &ta, &pve_c_trans);
    if (verbose) printf("Gibbs based expression values are written!\n");
    return 0;
}
