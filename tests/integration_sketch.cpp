// Compile-only check of INTEGRATION.md section B: the binding a maintainer of the reference would add -- the per-thread
// HitContainer<HitType> of EM.cpp flattened into the CSR that rsem_em_create takes, the frozen-probability rounds
// handed to rsem_em_run, the final pass to rsem_em_expected_weights -- against the reference's OWN headers
// (g++ -fsyntax-only -I<reference> -I<repo>/include; tests/test_capi_cpu.py, where /root/reference exists).
// Uses the reference's public member functions only; nothing of the reference is copied here.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "HitContainer.h"
#include "PairedEndHit.h"
#include "SingleHit.h"
#include "utils.h"

#include "rsem_hip.h"

template <class HitType>
struct FlatShards {
    std::vector<uint64_t> row_ptr;
    std::vector<int32_t> sid;
    std::vector<double> conprb, ncp;
    // after init<>() has filled hitvs[t] / ncpvs[t] (EM.cpp:97-174)
    void flatten(int nThreads, HitContainer<HitType>** hitvs, double** ncpvs) {
        row_ptr.assign(1, 0);
        for (int t = 0; t < nThreads; t++)
            for (READ_INT_TYPE i = 0; i < hitvs[t]->getN(); i++) {
                for (HIT_INT_TYPE j = hitvs[t]->getSAt(i); j < hitvs[t]->getSAt(i + 1); j++) {
                    sid.push_back(hitvs[t]->getHitAt(j).getSid());  // strand already stripped (SingleHit.h:26)
                    conprb.push_back(hitvs[t]->getHitAt(j).getConPrb());
                }
                row_ptr.push_back(sid.size());
                ncp.push_back(ncpvs[t][i]);
            }
    }
};

template <class HitType>
int frozen_rounds_on_the_gpu(int M, READ_INT_TYPE N0, int nThreads, HitContainer<HitType>** hitvs, double** ncpvs, std::vector<double>& theta,
                             double* counts, int ROUND, int MIN_ROUND, int MAX_ROUND, std::vector<double>& w, std::vector<double>& w_noise) {
    FlatShards<HitType> F;
    F.flatten(nThreads, hitvs, ncpvs);
    rsem_em_ctx* ctx = NULL;
    if (rsem_em_create(&ctx, /*device*/ 0, M, F.ncp.size(), F.sid.size(), F.row_ptr.data(), F.sid.data(), NULL, NULL) != RSEM_OK) {
        fprintf(stderr, "%s\n", rsem_hip_last_error());
        exit(-1);
    }
    rsem_em_set_values(ctx, F.conprb.data(), F.ncp.data());  // after round 11 has recomputed conprb (EM.cpp:383)
    int rounds = ROUND, totNum = 0;
    double bChange = 0.0;
    if (rsem_em_run(ctx, &theta[0], (double)N0, ROUND, MIN_ROUND, MAX_ROUND, &rounds, counts, &bChange, &totNum, NULL) != RSEM_OK) exit(-1);
    w.resize(F.sid.size());
    w_noise.resize(F.ncp.size());
    rsem_em_expected_weights(ctx, &theta[0], (double)N0, counts, w.data(), w_noise.data());  // EM.cpp:460-478
    rsem_em_destroy(ctx);
    return rounds;
}

template int frozen_rounds_on_the_gpu<SingleHit>(int, READ_INT_TYPE, int, HitContainer<SingleHit>**, double**, std::vector<double>&, double*, int, int, int,
                                                 std::vector<double>&, std::vector<double>&);
template int frozen_rounds_on_the_gpu<PairedEndHit>(int, READ_INT_TYPE, int, HitContainer<PairedEndHit>**, double**, std::vector<double>&, double*, int,
                                                    int, int, std::vector<double>&, std::vector<double>&);
