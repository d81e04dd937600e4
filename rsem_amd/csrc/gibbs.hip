// gibbs.hip -- rsem-run-gibbs's sampler on MI355X (gfx950).  C ABI: include/rsem_hip.h.
//
// The reference chain (Gibbs.cpp:265-353) is a collapsed Gibbs sampler: reads are visited strictly
// in order, each visit removes the read from `counts`, draws its transcript with probability
// proportional to (counts[sid] + alpha) * conprb, and adds it back.  `counts` is loop-carried from
// read to read (and nearly every read carries the noise transcript 0), so a chain has no
// read-level parallelism.  Two device samplers are provided:
//
//   RSEM_GIBBS_EXACT     the reference chain itself: same visiting order, same left-to-right
//                        cumulative sums, MT19937 + u = mt()*2^-32 (sampling.h:50-65), hence the
//                        same integer count vectors bit for bit.  One wave; 63 lanes stage the
//                        next tile of reads into LDS while lane 0 walks the chain.  Latency
//                        bound by construction -- the verification mode.
//   RSEM_GIBBS_PARALLEL  the data-augmentation (uncollapsed) Gibbs sampler for the same posterior:
//                        theta | z ~ Dirichlet(counts + alpha) (one Gamma draw per transcript,
//                        Marsaglia-Tsang, Philox4x32-10 counter RNG), then all z_i | theta drawn
//                        independently in parallel over the sliced layout of sell_layout.hpp --
//                        an E-step-shaped, HBM-bound sweep (8 B per alignment inside runs of
//                        identical reads).  It is a different Markov chain with the same
//                        stationary distribution; it mixes more slowly per sweep on weakly
//                        identified isoform pairs, which `thin` (extra sweeps per counted round)
//                        compensates.  Results are deterministic for a fixed seed.
//
// Per kept sample (Gibbs.cpp:313-346): theta = (counts + alpha) / totc, polishTheta,
// calcExpressionValues (WriteResults.h:55-104) and the running sums -- on the device.
#include <cmath>

#include "rng.hpp"
#include "sell_layout.hpp"

namespace {

using rsem::kEpsilon;

using rsem::Philox;
using rsem::u53;
using rsem::gamma_draw;

// g[i] = Gamma(counts[i] + alpha_i) (unnormalised Dirichlet draw); omitted transcripts (counts < 0) get 0
// ... and the counts are re-armed for the z pass that follows (init_counts: 0, or -1 for omitted transcripts; the noise
// bin starts at N0), which saves the separate reset launch of every sweep
__global__ void k_sample_theta(int32_t M, int32_t* __restrict__ counts, const double* __restrict__ alpha,
                               double pseudoC, Philox ph, uint32_t sweep, double* g, const int32_t* __restrict__ init_counts,
                               int32_t n0) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > M) return;
    int c = counts[i];
    g[i] = (c < 0) ? 0.0 : gamma_draw(ph, (uint32_t)i, sweep, (double)c + (alpha ? alpha[i] : pseudoC));
    counts[i] = init_counts[i] + (i == 0 ? n0 : 0);
}

__global__ void k_fill_double(int32_t n, double v, double* g) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] = v;
}

__global__ void k_reset_counts(int32_t M, const int32_t* __restrict__ init_counts, int32_t n0, int32_t* counts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= M) counts[i] = init_counts[i] + (i == 0 ? n0 : 0);
}

constexpr int kGWindow = 2048;  // sids per workgroup window (g values: 16 KB, int counts: 8 KB of LDS)

// g[base, base+span) -> LDS, count window zeroed; every wave of the workgroup calls this exactly once
__device__ inline void stage_gwindows(int base, int span, int M, const double* __restrict__ g, double* g_win, int* cnt_win) {
    for (int i = threadIdx.x; i < span; i += blockDim.x) {
        const int sidv = base + i;
        g_win[i] = (sidv >= 0 && sidv <= M) ? g[sidv] : 0.0;
        cnt_win[i] = 0;
    }
    __syncthreads();
}

template <int K>
struct SliceRegs {
    int id[K];
    double c[K];
    double nc;
};

// z_i | g for the reads of one block (T slices, one wave), lane-major runs as in the E step: a lane
// keeps the g values and integer pick counters of its current sid tuple in registers and spills
// them to the workgroup's LDS window when the tuple changes.  Weight order inside a read: noise,
// then the G lanes of the read in order, each lane's K planes in order.
template <int K>
__device__ inline void gibbs_block(const Shape& S, uint32_t T, uint32_t s_begin, uint32_t s_end, int lane, int base, int span,
                                   const double* __restrict__ g, double g0, double* g_win, int* cnt_win,
                                   const double* __restrict__ scp, const int32_t* __restrict__ ssid,
                                   const double* __restrict__ sncp, const unsigned long long* __restrict__ masks,
                                   const Philox& ph, uint32_t sweep, int32_t* counts, int& noise, int M) {
    const int lg = S.lg, G = 1 << lg;
    const int gl = lane & (G - 1);
    const bool g0lane = (gl == 0);
    const uint32_t R = 64u >> lg;
    const int gbase = lane & ~(G - 1);
    uint32_t m_base = s_begin;
    unsigned long long mv = (s_begin + lane < s_end) ? masks[s_begin + lane] : ~0ull;
    auto mask_of = [&](uint32_t t) -> unsigned long long {
        if (t - m_base >= 64u) {
            m_base = t;
            mv = (t + lane < s_end) ? masks[t + lane] : ~0ull;
        }
        const int src = (int)(t - m_base);
        const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)mv, src);
        const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(mv >> 32), src);
        return ((unsigned long long)hi << 32) | lo;
    };
    auto issue = [&](uint32_t t, unsigned long long m, SliceRegs<K>& b) {
        const uint32_t sl = t - S.slice_base;
        const uint64_t pl = (S.plane_base + (uint64_t)sl * K) * 64 + lane;
        const bool want = (m >> lane) & 1ull;
#pragma unroll
        for (int k = 0; k < K; k++) b.id[k] = ssid[want ? pl + (uint64_t)k * 64 : 0];
#pragma unroll
        for (int k = 0; k < K; k++) b.c[k] = scp[pl + (uint64_t)k * 64];
        b.nc = g0lane ? sncp[S.slot_base + sl * R + (lane >> lg)] : 0.0;
    };
    int rsid[K], acc[K];
    double rg[K];
#pragma unroll
    for (int k = 0; k < K; k++) { rsid[k] = 0; acc[k] = 0; rg[k] = 0.0; }
    auto spill = [&]() {
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (acc[k] != 0) {
                const unsigned off = (unsigned)(rsid[k] - base);
                if (off < (unsigned)span) atomicAdd(&cnt_win[off], acc[k]);
                else atomicAdd(&counts[rsid[k]], acc[k]);
            }
            acc[k] = 0;
        }
    };
    auto sample = [&](const SliceRegs<K>& cur, unsigned long long cur_m, uint32_t s) {
        if (cur_m != 0ull) {
            if ((cur_m >> lane) & 1ull) {
                spill();
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const int sidv = cur.id[k];
                    rsid[k] = sidv;
                    const unsigned off = (unsigned)(sidv - base);
                    rg[k] = (off < (unsigned)span) ? g_win[off] : g[sidv];
                }
            }
        }
        const double f0 = g0 * cur.nc;
        double f[K];
        double part = f0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            f[k] = rg[k] * cur.c[k];
            part += f[k];
        }
        double incl = part;  // inclusive scan over the G lanes of the read
        for (int d = 1; d < G; d <<= 1) {
            double o = __shfl_up(incl, d);
            if (gl >= d) incl += o;
        }
        double excl = __shfl_up(incl, 1);
        if (gl == 0) excl = 0.0;
        const double total = __shfl(incl, gbase + G - 1);
        // one uniform per read, keyed by the read's position in the sorted order (layout independent)
        const uint32_t sl = s - S.slice_base;
        const uint32_t b = sl / T, t = sl % T, r = (uint32_t)lane >> lg;
        const uint32_t left = S.n_rows - b * R * T;
        const uint32_t nb = left < R * T ? left : R * T;
        const uint32_t Tb = (nb + R - 1) / R;
        const uint32_t p = S.row_base + b * R * T + r * Tb + t;
        uint32_t rnd[4] = {0, 0, 0, 0};
        if (g0lane) ph.gen(p, sweep, 0x5a5a5a5au, 0u, rnd);
        const double u = __shfl(u53(rnd[0], rnd[1]), gbase);
        double target = u * total;
        if (target >= total) target = total * (1.0 - 1.1102230246251565e-16);
        int pick = -2;  // -2: not mine, -1: noise, k >= 0: my plane k
        if (total > 0.0 && target >= excl && target < incl) {
            double run = excl;
            if (g0lane) { run += f0; if (target < run) pick = -1; }
            if (pick == -2) {
                int last = -2;
#pragma unroll
                for (int k = 0; k < K; k++)
                    if (pick == -2) {
                        run += f[k];
                        if (f[k] > 0.0) last = k;
                        if (target < run) pick = k;
                    }
                if (pick == -2) pick = (last >= 0) ? last : ((g0lane && f0 > 0.0) ? -1 : -2);
            }
        }
        noise += (pick == -1);
#pragma unroll
        for (int k = 0; k < K; k++) acc[k] += (pick == k);
    };
    SliceRegs<K> A, B;
    unsigned long long mA = ~0ull, mB = 0;
    issue(s_begin, mA, A);
    stage_gwindows(base, span, M, g, g_win, cnt_win);  // the first slice's loads fly while the windows are staged
    for (uint32_t s = s_begin; s < s_end; s += 2) {
        if (s + 1 < s_end) { mB = mask_of(s + 1); issue(s + 1, mB, B); }
        sample(A, mA, s);
        if (s + 1 >= s_end) break;
        if (s + 2 < s_end) { mA = mask_of(s + 2); issue(s + 2, mA, A); }
        sample(B, mB, s + 1);
    }
    spill();
}

__global__ __launch_bounds__(kBlock) void k_sample_z_lane(
    const Shape* __restrict__ shapes, const Unit* __restrict__ units, uint32_t T, int M,
    const double* __restrict__ g, const double* __restrict__ scp, const int32_t* __restrict__ ssid,
    const double* __restrict__ sncp, const unsigned long long* __restrict__ masks, Philox ph, uint32_t sweep,
    int32_t* counts) {
    __shared__ double g_win[kGWindow];
    __shared__ int cnt_win[kGWindow];
    __shared__ int s_noise;
    const Unit U = units[blockIdx.x];
    if (threadIdx.x == 0) s_noise = 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int noise = 0;
    {
        const Shape S = U.S;
        const uint32_t u_end = S.slice_base + U.slice_begin + U.n_slices;
        const uint32_t s_begin = S.slice_base + U.slice_begin + (uint32_t)w * U.per_wave;
        const uint32_t s_end = min(u_end, s_begin + U.per_wave);
        const double g0 = g[0];
        if (s_begin < u_end) switch (S.K) {
            case 1: gibbs_block<1>(S, T, s_begin, s_end, lane, U.base, U.span, g, g0, g_win, cnt_win, scp, ssid, sncp, masks, ph, sweep, counts, noise, M); break;
            case 2: gibbs_block<2>(S, T, s_begin, s_end, lane, U.base, U.span, g, g0, g_win, cnt_win, scp, ssid, sncp, masks, ph, sweep, counts, noise, M); break;
            case 3: gibbs_block<3>(S, T, s_begin, s_end, lane, U.base, U.span, g, g0, g_win, cnt_win, scp, ssid, sncp, masks, ph, sweep, counts, noise, M); break;
            default: gibbs_block<4>(S, T, s_begin, s_end, lane, U.base, U.span, g, g0, g_win, cnt_win, scp, ssid, sncp, masks, ph, sweep, counts, noise, M); break;
        } else stage_gwindows(U.base, U.span, M, g, g_win, cnt_win);
    }
    for (int d = 32; d >= 1; d >>= 1) noise += __shfl_xor(noise, d);
    if (lane == 0 && noise) atomicAdd(&s_noise, noise);
    __syncthreads();
    for (int i = threadIdx.x; i < U.span; i += blockDim.x) {
        const int v = cnt_win[i];
        if (v != 0) atomicAdd(&counts[U.base + i], v);
    }
    if (threadIdx.x == 0 && s_noise) atomicAdd(&counts[0], s_noise);
}

// reads with > 256 alignments: thread per read over the caller's CSR
__global__ void k_sample_z_long(uint32_t n_rows, const uint32_t* __restrict__ row_list, uint32_t row_id_base,
                                const uint64_t* __restrict__ row_ptr, const int32_t* __restrict__ sid,
                                const double* __restrict__ cp, const double* __restrict__ ncp,
                                const double* __restrict__ g, Philox ph, uint32_t sweep, int32_t* counts) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows) return;
    uint32_t i = row_list[t];
    uint64_t fr = row_ptr[i], to = row_ptr[i + 1];
    double f0 = g[0] * ncp[i], total = f0;
    for (uint64_t j = fr; j < to; j++) total += g[sid[j]] * cp[j];
    if (!(total > 0.0)) return;
    uint32_t r[4];
    ph.gen(row_id_base + t, sweep, 0x5a5a5a5au, 0u, r);
    double target = u53(r[0], r[1]) * total;
    if (target >= total) target = total * (1.0 - 1.1102230246251565e-16);
    double run = f0;
    int pick = 0;
    if (!(target < run)) {
        int last = 0;
        bool found = false;
        for (uint64_t j = fr; j < to && !found; j++) {
            double f = g[sid[j]] * cp[j];
            run += f;
            if (f > 0.0) last = sid[j];
            if (target < run) { pick = sid[j]; found = true; }
        }
        if (!found) pick = last;
    }
    atomicAdd(&counts[pick], 1);
}

// ---- EXACT mode: the reference chain on one wave ------------------------------------------------

constexpr int kTileRows = 64;
constexpr int kTileItems = 3072;

struct MtState { uint32_t mt[624]; int idx; };

__device__ inline uint32_t mt_next(uint32_t* mt, int& idx) {  // lane 0 only; mt in LDS
    if (idx >= 624) {
        for (int k = 0; k < 624; k++) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// One sweep over all reads in file order (Gibbs.cpp:297-311), or the initial assignment
// (Gibbs.cpp:281-291) when kInit.  blockDim = 64.
template <bool kInit>
__global__ __launch_bounds__(64) void k_gibbs_exact(uint64_t N1, const uint64_t* __restrict__ row_ptr,
                                                     const int32_t* __restrict__ sid,
                                                     const double* __restrict__ cp, int32_t* counts, int32_t* z,
                                                     const double* __restrict__ alpha, double pseudoC,
                                                     MtState* mt_state) {
    __shared__ uint32_t mt[624];
    __shared__ uint64_t t_rp[kTileRows + 1];
    __shared__ int32_t t_sid[kTileItems];
    __shared__ double t_cp[kTileItems];
    __shared__ double arr[kTileItems];
    __shared__ int32_t t_z[kTileRows];
    const int lane = threadIdx.x;
    for (int i = lane; i < 624; i += 64) mt[i] = mt_state->mt[i];
    int idx = mt_state->idx;
    __syncthreads();

    for (uint64_t i0 = 0; i0 < N1; i0 += kTileRows) {
        const int nr = (int)min((uint64_t)kTileRows, N1 - i0);
        for (int r = lane; r <= nr; r += 64) t_rp[r] = row_ptr[i0 + r];
        if (lane < nr) t_z[lane] = kInit ? 0 : z[i0 + lane];
        __syncthreads();
        const uint64_t base = t_rp[0];
        const uint64_t n_items = t_rp[nr] - base;
        const bool staged = n_items <= (uint64_t)kTileItems;
        if (staged) {
            for (uint64_t j = lane; j < n_items; j += 64) { t_sid[j] = sid[base + j]; t_cp[j] = cp[base + j]; }
        }
        __syncthreads();
        if (lane == 0) {
            for (int r = 0; r < nr; r++) {
                const uint64_t fr = t_rp[r] - base, to = t_rp[r + 1] - base;
                const int len = (int)(to - fr);
                if (!kInit) --counts[t_z[r]];
                int l = 0;
                if (staged || len <= kTileItems) {
                    double cum = 0.0;
                    for (int j = 0; j < len; j++) {
                        int s = staged ? t_sid[fr + j] : sid[base + fr + j];
                        double p = staged ? t_cp[fr + j] : cp[base + fr + j];
                        double a = kInit ? p : ((double)counts[s] + (alpha ? alpha[s] : pseudoC)) * p;
                        cum = (j == 0) ? a : cum + a;  // arr[j] = a; arr[j] += arr[j-1]
                        arr[j] = cum;
                    }
                    double prb = ((double)mt_next(mt, idx) * (1.0 / 4294967296.0)) * arr[len - 1];
                    int lo = 0, hi = len - 1;
                    while (lo <= hi) {  // sampling.h:55-60
                        int mid = (lo + hi) / 2;
                        if (arr[mid] <= prb) lo = mid + 1; else hi = mid - 1;
                    }
                    l = lo < len ? lo : len - 1;
                } else {
                    // a single read with more alignments than the LDS tile: two passes over global memory
                    double tot = 0.0;
                    for (int j = 0; j < len; j++) {
                        int s = sid[base + fr + j];
                        double p = cp[base + fr + j];
                        double a = kInit ? p : ((double)counts[s] + (alpha ? alpha[s] : pseudoC)) * p;
                        tot = (j == 0) ? a : tot + a;
                    }
                    double prb = ((double)mt_next(mt, idx) * (1.0 / 4294967296.0)) * tot;
                    double cum = 0.0;
                    l = len - 1;
                    for (int j = 0; j < len; j++) {
                        int s = sid[base + fr + j];
                        double p = cp[base + fr + j];
                        double a = kInit ? p : ((double)counts[s] + (alpha ? alpha[s] : pseudoC)) * p;
                        cum = (j == 0) ? a : cum + a;
                        if (cum > prb) { l = j; break; }
                    }
                }
                int zn = staged ? t_sid[fr + l] : sid[base + fr + l];
                ++counts[zn];
                t_z[r] = zn;
            }
        }
        __syncthreads();
        if (lane < nr) z[i0 + lane] = t_z[lane];
        __syncthreads();
    }
    for (int i = lane; i < 624; i += 64) mt_state->mt[i] = mt[i];
    if (lane == 0) mt_state->idx = idx;
}

// ---- per-sample statistics (Gibbs.cpp:313-346, WriteResults.h:55-104) ---------------------------

constexpr int kStatBlock = 1024;

__device__ inline double block_sum_1024(double v) {
    __shared__ double red[kStatBlock / 64];
    __syncthreads();
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < kStatBlock / 64; i++) t += red[i];
    return t;
}

__global__ __launch_bounds__(kStatBlock) void k_gibbs_stats(int32_t M, const int32_t* __restrict__ counts,
                                                             const double* __restrict__ alpha, double pseudoC,
                                                             double totc, const double* __restrict__ eel,
                                                             const double* __restrict__ mw, double* tmp,
                                                             double* pme_c, double* pve_c, double* pme_tpm,
                                                             double* pme_fpkm) {
    // theta = (counts + alpha) / totc, then polishTheta
    double s = 0.0;
    for (int i = threadIdx.x; i <= M; i += blockDim.x) {
        int c = counts[i];
        double th = (c < 0) ? 0.0 : ((double)c + (alpha ? alpha[i] : pseudoC)) / totc;
        if (i > 0 && (mw[i] < kEpsilon || eel[i] < kEpsilon)) th = 0.0;
        else th = th / mw[i];
        tmp[i] = th;
        s += th;
    }
    const double sum = block_sum_1024(s);
    // calcExpressionValues: frac over eel >= EPS (i >= 1)
    double d1 = 0.0;
    for (int i = threadIdx.x; i <= M; i += blockDim.x) {
        double th = tmp[i] / sum;
        double fr = (i >= 1 && eel[i] >= kEpsilon) ? th : 0.0;
        tmp[i] = fr;
        d1 += fr;
    }
    double denom = block_sum_1024(d1);
    if (denom < kEpsilon) denom = 1.0;
    double d2 = 0.0;
    for (int i = threadIdx.x; i <= M; i += blockDim.x) {
        double fp = 0.0;
        if (i >= 1 && eel[i] >= kEpsilon) fp = (tmp[i] / denom) * 1e9 / eel[i];
        tmp[i] = fp;
        d2 += fp;
    }
    double denom2 = block_sum_1024(d2);
    if (denom2 < kEpsilon) denom2 = 1.0;
    for (int i = threadIdx.x; i <= M; i += blockDim.x) {
        double c = (double)counts[i];
        double fp = tmp[i];
        pme_c[i] += c;
        pve_c[i] += c * c;
        pme_fpkm[i] += fp;
        pme_tpm[i] += (i >= 1) ? fp / denom2 * 1e6 : 0.0;
    }
}

__global__ void k_gibbs_gene_stats(int32_t m, const int32_t* __restrict__ grp, const int32_t* __restrict__ counts,
                                   double* pve_c_genes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double c = 0.0;
    for (int j = grp[i]; j < grp[i + 1]; j++) c += (double)counts[j];
    pve_c_genes[i] += c * c;
}

void host_mt_seed(MtState& g, uint32_t seed) {  // boost::random::mt19937 seeding
    g.mt[0] = seed;
    for (int i = 1; i < 624; i++) g.mt[i] = 1812433253u * (g.mt[i - 1] ^ (g.mt[i - 1] >> 30)) + (uint32_t)i;
    g.idx = 624;
}

uint32_t host_mt_next(MtState& g) {
    if (g.idx >= 624) {
        for (int k = 0; k < 624; k++) {
            uint32_t y = (g.mt[k] & 0x80000000u) | (g.mt[(k + 1) % 624] & 0x7fffffffu);
            g.mt[k] = g.mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g.idx = 0;
    }
    uint32_t y = g.mt[g.idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

}  // namespace

struct rsem_gibbs_ctx {
    int device = 0;
    int32_t M = 0, m = 0;
    uint64_t N1 = 0, nitems = 0, nhits = 0, N0 = 0;
    double pseudoC = 1.0, totc = 0.0;
    hipStream_t stream = nullptr;
    int n_cus = 256;
    // items exactly as given (noise column inline): EXACT mode
    uint64_t* d_irp = nullptr;
    int32_t* d_isid = nullptr;
    double* d_icp = nullptr;
    // noise split out: PARALLEL mode
    uint64_t* d_row_ptr = nullptr;
    int32_t* d_sid = nullptr;
    double* d_cp = nullptr;
    double* d_ncp = nullptr;
    SellLayout L;
    double* d_scp = nullptr;
    double* d_sncp = nullptr;
    Unit* d_units = nullptr;
    uint32_t n_units = 0;
    // state
    int32_t* d_init_counts = nullptr;
    int32_t* d_counts = nullptr;
    int32_t* d_z = nullptr;
    double* d_g = nullptr;
    double* d_alpha = nullptr;
    double* d_eel = nullptr;
    double* d_mw = nullptr;
    int32_t* d_grp = nullptr;
    double* d_tmp = nullptr;
    double* d_acc[4] = {nullptr, nullptr, nullptr, nullptr};
    double* d_acc_genes = nullptr;
    MtState* d_mt = nullptr;
    // allele-specific: transcript groups over alleles
    int32_t m_trans = 0;
    int32_t* d_ta = nullptr;
    double* d_acc_trans = nullptr;
};

extern "C" {

int rsem_gibbs_chain_seeds(uint32_t seed, int nchains, uint32_t* out) {
    RSEM_REQUIRE(out && nchains >= 0, "bad argument");
    MtState g;
    host_mt_seed(g, seed);
    int n = 0;
    while (n < nchains) {  // sampling.h:26-38: skip seeds already handed out
        uint32_t s = host_mt_next(g);
        bool dup = false;
        for (int i = 0; i < n; i++) if (out[i] == s) { dup = true; break; }
        if (!dup) out[n++] = s;
    }
    return RSEM_OK;
}

int rsem_gibbs_destroy(rsem_gibbs_ctx* c) {
    if (!c) return RSEM_OK;
    (void)hipSetDevice(c->device);
    hipFree(c->d_irp); hipFree(c->d_isid); hipFree(c->d_icp); hipFree(c->d_row_ptr); hipFree(c->d_sid);
    hipFree(c->d_cp); hipFree(c->d_ncp); sell_free(c->L); hipFree(c->d_scp); hipFree(c->d_sncp);
    hipFree(c->d_init_counts); hipFree(c->d_counts); hipFree(c->d_z); hipFree(c->d_g); hipFree(c->d_alpha);
    hipFree(c->d_eel); hipFree(c->d_mw); hipFree(c->d_grp); hipFree(c->d_tmp);
    for (int i = 0; i < 4; i++) hipFree(c->d_acc[i]);
    hipFree(c->d_acc_genes); hipFree(c->d_mt); hipFree(c->d_units); hipFree(c->d_ta); hipFree(c->d_acc_trans);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RSEM_OK;
}

int rsem_gibbs_set_allele_groups(rsem_gibbs_ctx* c, int32_t m_trans, const int32_t* ta) {
    RSEM_REQUIRE(c && ta && m_trans >= 1, "bad argument");
    RSEM_HIP_TRY(hipSetDevice(c->device));
    hipFree(c->d_ta); hipFree(c->d_acc_trans);
    c->d_ta = nullptr; c->d_acc_trans = nullptr;
    RSEM_HIP_TRY(dmalloc(&c->d_ta, (size_t)m_trans + 1));
    RSEM_HIP_TRY(dmalloc(&c->d_acc_trans, (size_t)m_trans));
    RSEM_HIP_TRY(hipMemcpy(c->d_ta, ta, sizeof(int32_t) * ((size_t)m_trans + 1), hipMemcpyHostToDevice));
    RSEM_HIP_TRY(hipMemset(c->d_acc_trans, 0, sizeof(double) * m_trans));
    c->m_trans = m_trans;
    return RSEM_OK;
}

int rsem_gibbs_get_pve_c_trans(rsem_gibbs_ctx* c, double* out) {
    RSEM_REQUIRE(c && out, "NULL argument");
    if (!c->m_trans) { rsem::set_last_error("allele groups were never set"); return RSEM_ERR_STATE; }
    RSEM_HIP_TRY(hipSetDevice(c->device));
    RSEM_HIP_TRY(hipMemcpy(out, c->d_acc_trans, sizeof(double) * c->m_trans, hipMemcpyDeviceToHost));
    return RSEM_OK;
}

int rsem_gibbs_create(rsem_gibbs_ctx** out, int device, int32_t M, uint64_t N1, uint64_t nitems, const uint64_t* row_ptr,
                      const int32_t* sid, const double* conprb, const int32_t* init_counts, const double* alpha,
                      double pseudoC, double totc, uint64_t N0, const double* eel, const double* mw, int32_t m,
                      const int32_t* grp) {
    RSEM_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    RSEM_REQUIRE(M >= 1 && m >= 1, "M and m must be >= 1");
    RSEM_REQUIRE(row_ptr && sid && conprb && init_counts && eel && mw && grp, "NULL argument");
    RSEM_REQUIRE(N1 < 0xfffffff0ull, "N1 too large for one chain context");
    RSEM_REQUIRE(row_ptr[0] == 0 && row_ptr[N1] == nitems, "row_ptr[0] != 0 or row_ptr[N1] != nitems");
    RSEM_REQUIRE(N0 < 0x7fffffffull, "N0 does not fit the reference's int counts");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        (void)hipGetLastError();
        rsem::set_last_error("no HIP device %d (have %d)", device, ndev);
        return RSEM_ERR_NODEVICE;
    }
    // split the noise column out (host, once): hits CSR + per-read noise conprb
    std::vector<uint64_t> rp(N1 + 1, 0);
    std::vector<int32_t> hs;
    std::vector<double> hc, nc(N1, 0.0);
    hs.reserve(nitems);
    hc.reserve(nitems);
    for (uint64_t i = 0; i < N1; i++) {
        RSEM_REQUIRE(row_ptr[i + 1] >= row_ptr[i], "row_ptr is not monotone");
        RSEM_REQUIRE(row_ptr[i + 1] > row_ptr[i], "a read without any item cannot be sampled");
        for (uint64_t j = row_ptr[i]; j < row_ptr[i + 1]; j++) {
            RSEM_REQUIRE(sid[j] >= 0 && sid[j] <= M, "sid outside 0..M");
            if (sid[j] == 0) nc[i] += conprb[j];
            else { hs.push_back(sid[j]); hc.push_back(conprb[j]); }
        }
        rp[i + 1] = hs.size();
    }
    RSEM_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    RSEM_HIP_TRY(hipGetDeviceProperties(&prop, device));
    rsem_gibbs_ctx* c = new (std::nothrow) rsem_gibbs_ctx();
    if (!c) return RSEM_ERR_NOMEM;
    c->device = device; c->M = M; c->m = m; c->N1 = N1; c->nitems = nitems; c->nhits = hs.size(); c->N0 = N0;
    c->pseudoC = pseudoC; c->totc = totc;
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
#define G_TRY(expr)                                                                                     \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            rsem::set_last_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));  \
            rsem_gibbs_destroy(c);                                                                      \
            return _e == hipErrorOutOfMemory ? RSEM_ERR_NOMEM : RSEM_ERR_HIP;                           \
        }                                                                                               \
    } while (0)
    G_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    hipStream_t st = c->stream;
    const size_t nM = (size_t)M + 1;
    G_TRY(dmalloc(&c->d_irp, N1 + 1)); G_TRY(dmalloc(&c->d_isid, nitems)); G_TRY(dmalloc(&c->d_icp, nitems));
    G_TRY(dmalloc(&c->d_row_ptr, N1 + 1)); G_TRY(dmalloc(&c->d_sid, hs.size())); G_TRY(dmalloc(&c->d_cp, hs.size()));
    G_TRY(dmalloc(&c->d_ncp, N1));
    G_TRY(dmalloc(&c->d_init_counts, nM)); G_TRY(dmalloc(&c->d_counts, nM)); G_TRY(dmalloc(&c->d_z, N1));
    G_TRY(dmalloc(&c->d_g, nM)); G_TRY(dmalloc(&c->d_eel, nM)); G_TRY(dmalloc(&c->d_mw, nM));
    G_TRY(dmalloc(&c->d_grp, (size_t)m + 1)); G_TRY(dmalloc(&c->d_tmp, nM)); G_TRY(dmalloc(&c->d_mt, 1));
    for (int i = 0; i < 4; i++) G_TRY(dmalloc(&c->d_acc[i], nM));
    G_TRY(dmalloc(&c->d_acc_genes, (size_t)m));
    G_TRY(hipMemcpyAsync(c->d_irp, row_ptr, sizeof(uint64_t) * (N1 + 1), hipMemcpyHostToDevice, st));
    G_TRY(hipMemcpyAsync(c->d_isid, sid, sizeof(int32_t) * nitems, hipMemcpyHostToDevice, st));
    G_TRY(hipMemcpyAsync(c->d_icp, conprb, sizeof(double) * nitems, hipMemcpyHostToDevice, st));
    G_TRY(hipMemcpyAsync(c->d_row_ptr, rp.data(), sizeof(uint64_t) * (N1 + 1), hipMemcpyHostToDevice, st));
    if (!hs.empty()) {
        G_TRY(hipMemcpyAsync(c->d_sid, hs.data(), sizeof(int32_t) * hs.size(), hipMemcpyHostToDevice, st));
        G_TRY(hipMemcpyAsync(c->d_cp, hc.data(), sizeof(double) * hc.size(), hipMemcpyHostToDevice, st));
    }
    if (N1) G_TRY(hipMemcpyAsync(c->d_ncp, nc.data(), sizeof(double) * N1, hipMemcpyHostToDevice, st));
    G_TRY(hipMemcpyAsync(c->d_init_counts, init_counts, sizeof(int32_t) * nM, hipMemcpyHostToDevice, st));
    G_TRY(hipMemcpyAsync(c->d_eel, eel, sizeof(double) * nM, hipMemcpyHostToDevice, st));
    G_TRY(hipMemcpyAsync(c->d_mw, mw, sizeof(double) * nM, hipMemcpyHostToDevice, st));
    G_TRY(hipMemcpyAsync(c->d_grp, grp, sizeof(int32_t) * ((size_t)m + 1), hipMemcpyHostToDevice, st));
    if (alpha) {
        G_TRY(dmalloc(&c->d_alpha, nM));
        G_TRY(hipMemcpyAsync(c->d_alpha, alpha, sizeof(double) * nM, hipMemcpyHostToDevice, st));
    }
    G_TRY(hipStreamSynchronize(st));
#undef G_TRY
    // a fixed block length keeps the layout (and with it nothing but performance) device independent
    int rc = sell_build(c->L, st, N1, M, c->d_row_ptr, c->d_sid, (uint32_t)c->n_cus * 4 * 6 * 5 / 2);
    if (rc == RSEM_OK) {
        hipError_t e1 = dmalloc(&c->d_scp, c->L.n_planes * 64), e2 = dmalloc(&c->d_sncp, (size_t)c->L.n_slots);
        if (e1 != hipSuccess || e2 != hipSuccess) rc = RSEM_ERR_NOMEM;
    }
    if (rc == RSEM_OK) rc = sell_fill_values(c->L, st, c->d_row_ptr, c->d_cp, c->d_ncp, c->d_scp, c->d_sncp);
    if (rc == RSEM_OK && hipStreamSynchronize(st) != hipSuccess) rc = RSEM_ERR_HIP;
    std::vector<Unit> units;
    if (rc == RSEM_OK) rc = sell_build_units(c->L, units, kGWindow);
    if (rc == RSEM_OK) {
        c->n_units = (uint32_t)units.size();
        if (dmalloc(&c->d_units, units.size()) != hipSuccess) rc = RSEM_ERR_NOMEM;
        else if (!units.empty() &&
                 hipMemcpy(c->d_units, units.data(), sizeof(Unit) * units.size(), hipMemcpyHostToDevice) != hipSuccess)
            rc = RSEM_ERR_HIP;
    }
    if (rc != RSEM_OK) { rsem_gibbs_destroy(c); return rc; }
    *out = c;
    return RSEM_OK;
}

int rsem_gibbs_run(rsem_gibbs_ctx* c, int mode, uint32_t seed, int burnin, int nsamples, int gap, int thin,
                   int32_t* count_vectors, double* pme_c, double* pve_c, double* pme_tpm, double* pme_fpkm,
                   double* pve_c_genes, double* sweep_ms) {
    RSEM_REQUIRE(c && pme_c && pve_c && pme_tpm && pme_fpkm && pve_c_genes, "NULL argument");
    RSEM_REQUIRE(mode == RSEM_GIBBS_EXACT || mode == RSEM_GIBBS_PARALLEL, "unknown mode");
    RSEM_REQUIRE(burnin >= 0 && nsamples >= 1 && gap >= 1, "bad chain parameters");
    if (thin < 1) thin = 1;
    RSEM_HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t nM = (size_t)c->M + 1;
    const int gM = rsem::ceil_div(nM, kBlock);
    for (int i = 0; i < 4; i++) RSEM_HIP_TRY(hipMemsetAsync(c->d_acc[i], 0, sizeof(double) * nM, st));
    RSEM_HIP_TRY(hipMemsetAsync(c->d_acc_genes, 0, sizeof(double) * c->m, st));
    if (c->m_trans) RSEM_HIP_TRY(hipMemsetAsync(c->d_acc_trans, 0, sizeof(double) * c->m_trans, st));
    int32_t* d_cv = nullptr;
    if (count_vectors) RSEM_HIP_TRY(dmalloc(&d_cv, (size_t)nsamples * nM));
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    RSEM_HIP_TRY(hipEventCreate(&ev0));
    RSEM_HIP_TRY(hipEventCreate(&ev1));
    Philox ph{seed, 0x52534547u};  // 'RSEG'
    uint32_t sweep_no = 0;

    auto parallel_z = [&](uint32_t sw, bool reset) -> int {
        if (reset)
            hipLaunchKernelGGL(k_reset_counts, dim3(gM), dim3(kBlock), 0, st, c->M, c->d_init_counts, (int32_t)c->N0,
                               c->d_counts);
        if (c->n_units)
            hipLaunchKernelGGL(k_sample_z_lane, dim3(c->n_units), dim3(kBlock), 0, st, c->L.d_shapes, c->d_units, c->L.T, c->M,
                               c->d_g, c->d_scp, c->L.d_ssid, c->d_sncp, c->L.d_masks, ph, sw, c->d_counts);
        if (c->L.n_long_rows)
            hipLaunchKernelGGL(k_sample_z_long, dim3(rsem::ceil_div(c->L.n_long_rows, kBlock)), dim3(kBlock), 0, st,
                               c->L.n_long_rows, c->L.d_order + c->L.n_sell_rows, c->L.n_sell_rows, c->d_row_ptr, c->d_sid,
                               c->d_cp, c->d_ncp, c->d_g, ph, sw, c->d_counts);
        RSEM_HIP_TRY(hipGetLastError());
        return RSEM_OK;
    };

    int rc = RSEM_OK;
    // initial state: z_i ~ conprb (Gibbs.cpp:281-291)
    if (mode == RSEM_GIBBS_EXACT) {
        MtState h;
        host_mt_seed(h, seed);
        RSEM_HIP_TRY(hipMemcpyAsync(c->d_mt, &h, sizeof(MtState), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_reset_counts, dim3(gM), dim3(kBlock), 0, st, c->M, c->d_init_counts, (int32_t)c->N0, c->d_counts);
        hipLaunchKernelGGL(k_gibbs_exact<true>, dim3(1), dim3(64), 0, st, c->N1, c->d_irp, c->d_isid, c->d_icp, c->d_counts,
                           c->d_z, c->d_alpha, c->pseudoC, c->d_mt);
        RSEM_HIP_TRY(hipGetLastError());
        RSEM_HIP_TRY(hipStreamSynchronize(st));  // h must outlive the copy
    } else {
        hipLaunchKernelGGL(k_fill_double, dim3(gM), dim3(kBlock), 0, st, (int32_t)nM, 1.0, c->d_g);
        rc = parallel_z(sweep_no++, true);
        if (rc != RSEM_OK) return rc;
    }
    const int chainlen = 1 + (nsamples - 1) * gap;
    int kept = 0;
    RSEM_HIP_TRY(hipEventRecord(ev0, st));
    for (int round = 1; round <= burnin + chainlen; round++) {
        if (mode == RSEM_GIBBS_EXACT) {
            hipLaunchKernelGGL(k_gibbs_exact<false>, dim3(1), dim3(64), 0, st, c->N1, c->d_irp, c->d_isid, c->d_icp,
                               c->d_counts, c->d_z, c->d_alpha, c->pseudoC, c->d_mt);
            RSEM_HIP_TRY(hipGetLastError());
        } else {
            for (int t = 0; t < thin; t++) {
                hipLaunchKernelGGL(k_sample_theta, dim3(gM), dim3(kBlock), 0, st, c->M, c->d_counts, c->d_alpha, c->pseudoC,
                                   ph, sweep_no, c->d_g, c->d_init_counts, (int32_t)c->N0);
                rc = parallel_z(sweep_no++, false);
                if (rc != RSEM_OK) return rc;
            }
        }
        if (round > burnin && (round - burnin - 1) % gap == 0) {
            if (d_cv) RSEM_HIP_TRY(hipMemcpyAsync(d_cv + (size_t)kept * nM, c->d_counts, sizeof(int32_t) * nM,
                                                  hipMemcpyDeviceToDevice, st));
            ++kept;
            hipLaunchKernelGGL(k_gibbs_stats, dim3(1), dim3(kStatBlock), 0, st, c->M, c->d_counts, c->d_alpha, c->pseudoC,
                               c->totc, c->d_eel, c->d_mw, c->d_tmp, c->d_acc[0], c->d_acc[1], c->d_acc[2], c->d_acc[3]);
            hipLaunchKernelGGL(k_gibbs_gene_stats, dim3(rsem::ceil_div(c->m, kBlock)), dim3(kBlock), 0, st, c->m, c->d_grp,
                               c->d_counts, c->d_acc_genes);
            if (c->m_trans)
                hipLaunchKernelGGL(k_gibbs_gene_stats, dim3(rsem::ceil_div(c->m_trans, kBlock)), dim3(kBlock), 0, st, c->m_trans,
                                   c->d_ta, c->d_counts, c->d_acc_trans);
            RSEM_HIP_TRY(hipGetLastError());
        }
    }
    RSEM_HIP_TRY(hipEventRecord(ev1, st));
    RSEM_HIP_TRY(hipMemcpyAsync(pme_c, c->d_acc[0], sizeof(double) * nM, hipMemcpyDeviceToHost, st));
    RSEM_HIP_TRY(hipMemcpyAsync(pve_c, c->d_acc[1], sizeof(double) * nM, hipMemcpyDeviceToHost, st));
    RSEM_HIP_TRY(hipMemcpyAsync(pme_tpm, c->d_acc[2], sizeof(double) * nM, hipMemcpyDeviceToHost, st));
    RSEM_HIP_TRY(hipMemcpyAsync(pme_fpkm, c->d_acc[3], sizeof(double) * nM, hipMemcpyDeviceToHost, st));
    RSEM_HIP_TRY(hipMemcpyAsync(pve_c_genes, c->d_acc_genes, sizeof(double) * c->m, hipMemcpyDeviceToHost, st));
    if (d_cv) RSEM_HIP_TRY(hipMemcpyAsync(count_vectors, d_cv, sizeof(int32_t) * (size_t)nsamples * nM, hipMemcpyDeviceToHost, st));
    RSEM_HIP_TRY(hipStreamSynchronize(st));
    if (sweep_ms) {
        float ms = 0.f;
        RSEM_HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
        int sweeps = (burnin + chainlen) * (mode == RSEM_GIBBS_PARALLEL ? thin : 1);
        *sweep_ms = sweeps ? ms / sweeps : 0.0;
    }
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    hipFree(d_cv);
    return RSEM_OK;
}

}  // extern "C"
