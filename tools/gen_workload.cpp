// gen_workload.cpp -- BENCH / TEST INFRASTRUCTURE: the synthetic EM workload of tools/synth_data.py as threaded C++ (shared
// library, ctypes), for the sizes numpy cannot build in a bench slot: BASELINE configs[4] (100 M reads x 500 k transcripts x
// ~40 alignments = 4 G alignments, 50 GB) and the cross-gene variant of configs[2].  Same model as make_em_workload():
// genes own contiguous transcript ids; every gene has 2k "segments" (exon-combination classes), each compatible with a fixed
// subset of the gene's isoforms; a read picks a transcript by expression, then one of the segments containing it, and
// aligns to exactly that segment's isoform set; true theta ~ lognormal(0, 2) with 30 % zeros; conprb = 10^U(-60,-3) per
// read with a per-hit jitter 10^N(0,0.5); ncp = 10^U(-130,-40).  `cross_frac` of the reads ALSO hit 1..3 transcripts of a
// different, random gene (paralogs / cross-gene multi-mappers: the tuples that leave a unit's LDS window).
// Everything is a pure function of (parameters, seed, shard): per-read counter-based hashing, independent of the number of
// threads.  Not the numpy generator's numbers (another random stream), the same distribution.
//   g++ -O2 -std=c++17 -shared -fPIC -pthread -o tools/bin/libgenwl.so tools/gen_workload.cpp
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline uint64_t key(uint64_t seed, uint64_t a, uint64_t b, uint64_t c) { return mix64(mix64(mix64(seed ^ (a * 0xD1B54A32D192ED03ull)) + b) + c * 0x8CB92BA72F3D8DD7ull); }
inline double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

struct Plan {
    int64_t N1;
    int32_t M;
    double cross_frac;
    uint64_t seed;
    uint32_t shard;
    int threads;
    std::vector<int32_t> gstart;        // first 0-based transcript of every gene (+ M)
    std::vector<int32_t> t_gene;        // gene of a transcript
    std::vector<uint64_t> seg_mask;     // isoform subset of a segment (bit j = isoform j of its gene)
    std::vector<int32_t> seg_gene;
    std::vector<int64_t> t_ptr;         // transcript -> segments containing it
    std::vector<int32_t> t_segs;
    std::vector<double> cdf;
    std::vector<uint32_t> read_seg;     // pass 1
    std::vector<uint8_t> read_extra;    // pass 1: 0..3 cross-gene hits
    std::vector<uint64_t> row_ptr;
};

enum : uint64_t { S_GENES = 1, S_SEG = 2, S_THETA = 3, S_READ = 4, S_HIT = 5, S_CROSS = 6 };

template <typename F>
void parallel_for(int64_t n, int threads, F f) {
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, (n + 65535) / 65536));
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back([=]() { f(n * t / threads, n * (t + 1) / threads); });
    for (auto& x : th) x.join();
}

}  // namespace

extern "C" {

void* wl_plan(int64_t N1, int32_t M, double mean_hits, int kmin, int kmax, double cross_frac, uint64_t seed, uint32_t shard, int threads,
              uint64_t* nnz_out) {
    if (N1 < 1 || M < 1 || kmin < 1 || kmax > 64 || kmin > kmax) return nullptr;
    Plan* P = new Plan();
    P->N1 = N1; P->M = M; P->cross_frac = cross_frac; P->seed = seed; P->shard = shard;
    P->threads = threads > 0 ? threads : (int)std::max(1u, std::thread::hardware_concurrency());
    // genes
    {
        int32_t tot = 0;
        uint64_t g = 0;
        while (tot < M) {
            int k = kmin + (int)(u01(key(seed, S_GENES, g, 0)) * (kmax - kmin + 1));
            k = std::min(k, M - tot);
            P->gstart.push_back(tot);
            tot += k;
            ++g;
        }
        P->gstart.push_back(M);
    }
    const int n_genes = (int)P->gstart.size() - 1;
    P->t_gene.resize(M);
    for (int g = 0; g < n_genes; g++)
        for (int t = P->gstart[g]; t < P->gstart[g + 1]; t++) P->t_gene[t] = g;
    const double mean_k = (double)M / n_genes;
    const double p_inc = std::min(0.95, std::max(0.05, (mean_hits - 1.0) / std::max(mean_k - 1.0, 1e-9)));
    // segments: 2k per gene, each a non-empty subset of the gene's isoforms
    for (int g = 0; g < n_genes; g++) {
        const int k = P->gstart[g + 1] - P->gstart[g];
        for (int s = 0; s < 2 * k; s++) {
            uint64_t m = 0;
            for (int j = 0; j < k; j++)
                if (u01(key(seed, S_SEG, (uint64_t)g << 8 | (uint64_t)s, j)) < p_inc) m |= 1ull << j;
            m |= 1ull << (int)(u01(key(seed, S_SEG, (uint64_t)g << 8 | (uint64_t)s, 1000)) * k);
            P->seg_mask.push_back(m);
            P->seg_gene.push_back(g);
        }
    }
    const int64_t n_seg = (int64_t)P->seg_mask.size();
    // transcript -> segments containing it
    P->t_ptr.assign((size_t)M + 1, 0);
    for (int64_t s = 0; s < n_seg; s++)
        for (uint64_t m = P->seg_mask[s]; m; m &= m - 1) P->t_ptr[P->gstart[P->seg_gene[s]] + __builtin_ctzll(m) + 1]++;
    for (int t = 0; t < M; t++) P->t_ptr[t + 1] += P->t_ptr[t];
    P->t_segs.resize((size_t)P->t_ptr[M]);
    {
        std::vector<int64_t> fill(P->t_ptr.begin(), P->t_ptr.end() - 1);
        for (int64_t s = 0; s < n_seg; s++)
            for (uint64_t m = P->seg_mask[s]; m; m &= m - 1) P->t_segs[fill[P->gstart[P->seg_gene[s]] + __builtin_ctzll(m)]++] = (int32_t)s;
    }
    // expression
    P->cdf.resize(M);
    double acc = 0.0;
    for (int t = 0; t < M; t++) {
        const double u1 = std::max(u01(key(seed, S_THETA, t, 0)), 1e-300), u2 = u01(key(seed, S_THETA, t, 1));
        double th = std::exp(2.0 * std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2));
        if (u01(key(seed, S_THETA, t, 2)) < 0.3 || P->t_ptr[t + 1] == P->t_ptr[t]) th = 0.0;
        acc += th;
        P->cdf[t] = acc;
    }
    for (int t = 0; t < M; t++) P->cdf[t] /= acc;
    // pass 1: the segment (and the number of cross-gene hits) of every read -> row lengths
    P->read_seg.resize((size_t)N1);
    P->read_extra.assign((size_t)N1, 0);
    P->row_ptr.assign((size_t)N1 + 1, 0);
    const uint64_t rs = seed ^ ((uint64_t)shard << 40);
    parallel_for(N1, P->threads, [P, rs, M](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; i++) {
            const double u = u01(key(rs, S_READ, (uint64_t)i, 0));
            int t = (int)(std::upper_bound(P->cdf.begin(), P->cdf.end(), u) - P->cdf.begin());
            t = std::min(t, M - 1);
            while (P->t_ptr[t + 1] == P->t_ptr[t]) t = (t + 1) % M;  // (cdf ties: a transcript nobody expresses)
            const int64_t ns = P->t_ptr[t + 1] - P->t_ptr[t];
            const int32_t s = P->t_segs[P->t_ptr[t] + (int64_t)(u01(key(rs, S_READ, (uint64_t)i, 1)) * ns)];
            P->read_seg[i] = (uint32_t)s;
            int extra = 0;
            if (P->cross_frac > 0.0 && u01(key(rs, S_CROSS, (uint64_t)i, 0)) < P->cross_frac) extra = 1 + (int)(u01(key(rs, S_CROSS, (uint64_t)i, 1)) * 3.0);
            P->read_extra[i] = (uint8_t)extra;
            P->row_ptr[i + 1] = (uint64_t)__builtin_popcountll(P->seg_mask[s]) + (uint64_t)extra;
        }
    });
    for (int64_t i = 0; i < N1; i++) P->row_ptr[i + 1] += P->row_ptr[i];
    if (nnz_out) *nnz_out = P->row_ptr[N1];
    return P;
}

void wl_fill(void* h, uint64_t* row_ptr, int32_t* sid, double* conprb, double* ncp) {
    Plan* P = (Plan*)h;
    const int64_t N1 = P->N1;
    memcpy(row_ptr, P->row_ptr.data(), sizeof(uint64_t) * ((size_t)N1 + 1));
    const uint64_t rs = P->seed ^ ((uint64_t)P->shard << 40);
    const int n_genes = (int)P->gstart.size() - 1;
    parallel_for(N1, P->threads, [=](int64_t a, int64_t b) {
        const double ln10 = 2.302585092994046;
        for (int64_t i = a; i < b; i++) {
            const uint32_t s = P->read_seg[i];
            const int g = P->seg_gene[s];
            uint64_t o = P->row_ptr[i];
            const double e_row = -60.0 + 57.0 * u01(key(rs, S_READ, (uint64_t)i, 2));
            auto value = [&](uint64_t j) {
                const double u1 = std::max(u01(key(rs, S_HIT, (uint64_t)i, 2 * j)), 1e-300), u2 = u01(key(rs, S_HIT, (uint64_t)i, 2 * j + 1));
                const double n = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
                return std::exp(ln10 * (e_row + 0.5 * n));
            };
            uint64_t j = 0;
            for (uint64_t m = P->seg_mask[s]; m; m &= m - 1, j++) {
                sid[o + j] = P->gstart[g] + __builtin_ctzll(m) + 1;  // 1-based, ascending within the segment
                conprb[o + j] = value(j);
            }
            const int extra = P->read_extra[i];
            if (extra) {
                int g2 = (int)(u01(key(rs, S_CROSS, (uint64_t)i, 2)) * n_genes);
                if (g2 == g) g2 = (g2 + 1) % n_genes;
                const int k2 = P->gstart[g2 + 1] - P->gstart[g2];
                const int o2 = (int)(u01(key(rs, S_CROSS, (uint64_t)i, 3)) * k2);
                for (int x = 0; x < extra; x++, j++) {
                    sid[o + j] = P->gstart[g2] + (o2 + x) % k2 + 1;  // (a gene with fewer than `extra` isoforms repeats one: allowed)
                    conprb[o + j] = value(j);
                }
            }
            ncp[i] = std::exp(ln10 * (-130.0 + 90.0 * u01(key(rs, S_READ, (uint64_t)i, 3))));
        }
    });
}

void wl_free(void* h) { delete (Plan*)h; }

}  // extern "C"
