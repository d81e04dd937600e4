// estep_emu.cpp -- TEST INFRASTRUCTURE: runs rsem_amd/csrc/estep_block.hpp (the per-wave body of the LANE E-step kernel) on
// the CPU.  One OS thread per lane, 256 per workgroup; the cross-lane intrinsics are exchanges through memory with a
// barrier on either side; LDS is an array per workgroup; atomics are CAS loops.  The layout (sort, shapes, planes, masks,
// units) is rebuilt here on the host with sell_layout.hpp's own index helpers (row_to_slot, shape_G / shape_R, the policy
// table, the Q32 rule); what is being tested is the kernel body: lane -> (read, position) mapping, prefetch rings, Q32
// arithmetic, reductions over a read's lanes, spills, the compile-time variants.  Never part of the product.
//
//   estep_emu in.bin out.bin        in:  i32 M, N1, T, policy, q32 (0/1), range_bits, from_counts (0/1), pad; f64 N0
//                                        u64 row_ptr[N1+1]; i32 sid[nnz]; f64 cp[nnz]; f64 ncp[N1]; f64 theta[M+1]
//                                   out: f64 counts[M+1] (without N0), f64 noise total, f64 reads with a non-zero normaliser
// Build (tests/test_estep_emu_cpu.py): hipcc -DRSEM_EMU [-DRSEM_GENERAL_G=1 -DRSEM_FAST_RCP=1 ...] tests/estep_emu.cpp -lpthread
#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../rsem_amd/csrc/sell_layout.hpp"

// ---- the machine ---------------------------------------------------------------------------------------------------
namespace emu {
struct Wave {
    pthread_barrier_t bar;
    unsigned long long slot[64];
};
struct Block {
    pthread_barrier_t bar;
    Wave w[4];
};
thread_local int t_tid = 0;
thread_local Block* t_blk = nullptr;
inline Wave& wave() { return t_blk->w[t_tid >> 6]; }
inline int lane() { return t_tid & 63; }
template <typename T>
inline T exchange(T v, int src) {  // every lane of the wave calls this; returns lane src's v (own when src is out of range)
    static_assert(sizeof(T) <= 8, "8-byte slots");
    Wave& w = wave();
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    w.slot[lane()] = raw;
    pthread_barrier_wait(&w.bar);
    if (src >= 0 && src < 64) raw = w.slot[src];
    pthread_barrier_wait(&w.bar);
    T r;
    memcpy(&r, &raw, sizeof(T));
    return r;
}
inline unsigned long long ballot(bool p) {
    Wave& w = wave();
    w.slot[lane()] = p ? 1ull : 0ull;
    pthread_barrier_wait(&w.bar);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) m |= w.slot[i] << i;
    pthread_barrier_wait(&w.bar);
    return m;
}
inline int dpp_src(int ctrl) {
    const int l = lane();
    static const int p1[4] = {1, 0, 3, 2}, p2[4] = {2, 3, 0, 1};
    switch (ctrl) {
        case 0xB1: return (l & ~3) | p1[l & 3];
        case 0x4E: return (l & ~3) | p2[l & 3];
        case 0x141: return (l & ~7) | (7 - (l & 7));
        case 0x140: return (l & ~15) | (15 - (l & 15));
    }
    fprintf(stderr, "estep_emu: DPP control %#x not modelled\n", ctrl);
    abort();
}
inline void atomic_add(double* p, double v) {
    auto* a = reinterpret_cast<std::atomic<unsigned long long>*>(p);
    unsigned long long old = a->load(std::memory_order_relaxed), neu;
    do {
        double d;
        memcpy(&d, &old, 8);
        d += v;
        memcpy(&neu, &d, 8);
    } while (!a->compare_exchange_weak(old, neu, std::memory_order_relaxed));
}
inline double ll_as_double(long long x) { double d; memcpy(&d, &x, 8); return d; }
inline long long double_as_ll(double d) { long long x; memcpy(&x, &d, 8); return x; }
}  // namespace emu

#define RSEM_DEVFN inline
#define RSEM_TIDX (emu::t_tid)
#define RSEM_BDIM 256
#define RSEM_SYNC() pthread_barrier_wait(&emu::t_blk->bar)
#define RSEM_SHFL_XOR(v, d) emu::exchange(v, emu::lane() ^ (d))
#define RSEM_SHFL_DOWN(v, d) emu::exchange(v, emu::lane() + (d))
#define RSEM_SHFL(v, src) emu::exchange(v, (src) & 63)
#define RSEM_BALLOT(p) emu::ballot(p)
#define RSEM_READLANE(v, src) emu::exchange(v, src)
#define RSEM_ATOMIC_ADD(p, v) emu::atomic_add(p, v)
#define RSEM_LDS_ADD(p, v) emu::atomic_add(p, v)
#define RSEM_RCP(x) (1.0 / (x) * (1.0 + 3e-8))  /* v_rcp_f64 is not exact either: the Newton steps must repair this */
#define RSEM_DPP_MOV(v, ctrl) emu::exchange(v, emu::dpp_src(ctrl))
#define RSEM_LL_AS_DOUBLE(x) emu::ll_as_double(x)
#define RSEM_DOUBLE_AS_LL(x) emu::double_as_ll(x)

namespace {
using rsem::kEpsilon;
constexpr int kTotSlots = 64;
constexpr int kWindow = 2048;
#include "../rsem_amd/csrc/estep_block.hpp"
}  // namespace

// ---- the layout, on the host -----------------------------------------------------------------------------------------
struct HostLayout {
    uint32_t T = 4;
    std::vector<Shape> shapes;
    std::vector<uint32_t> order;
    std::vector<int32_t> ssid;
    std::vector<unsigned char> sval;
    std::vector<double> sncp;
    std::vector<int16_t> sexp;
    std::vector<unsigned long long> masks;
    uint32_t n_slices = 0;
};

static uint32_t host_mix32(uint32_t h, uint32_t v) { return h ^ (v + 0x9e3779b9u + (h << 6) + (h >> 2)); }

static int shape_of_len(uint64_t L, int policy) {
#if RSEM_GENERAL_G
    static uint16_t tab[2][257];
    static bool init = false;
    if (!init) { shape_policy_table(0, tab[0]); shape_policy_table(1, tab[1]); init = true; }
    return L <= 256 ? tab[policy][L] : kLongShape;
#else
    (void)policy;
    return shape_id_of(L);
#endif
}

static void build_layout(HostLayout& H, int M, uint64_t N1, const uint64_t* rp, const int32_t* sid, const double* cp, const double* ncp,
                         int policy, bool q32, int range_bits) {
    // keys (k_row_keys), sorted rows, shapes (the host loop of sell_build)
    std::vector<std::pair<uint64_t, uint32_t>> keyed(N1);
    for (uint64_t i = 0; i < N1; i++) {
        uint32_t h = 0x811c9dc5u, mn = 0xffffffffu;
        double vmx = 0.0, vmn = 1.79e308;
        for (uint64_t j = rp[i]; j < rp[i + 1]; j++) {
            h = host_mix32(h, (uint32_t)sid[j]);
            mn = std::min(mn, (uint32_t)sid[j]);
            if (!(cp[j] >= 0.0)) vmx = 1e308;
            vmx = fmax(vmx, cp[j]);
            if (cp[j] > 0.0) vmn = fmin(vmn, cp[j]);
        }
        if (mn > kKeyMinSidCap) mn = kKeyMinSidCap;
        int shape = shape_of_len(rp[i + 1] - rp[i], policy);
        if (shape == kLongShape) { fprintf(stderr, "estep_emu: rows with more than 256 alignments are not modelled\n"); exit(2); }
        Q32Scale q;
        if (q32 && q32_scale_of(vmx, vmn, range_bits, q)) shape += kShapesPerFmt;
        keyed[i] = {((uint64_t)shape << (64 - kShapeBits)) | ((uint64_t)mn << 32) | h, (uint32_t)i};
    }
    std::stable_sort(keyed.begin(), keyed.end());
    H.order.resize(N1);
    for (uint64_t p = 0; p < N1; p++) H.order[p] = keyed[p].second;
    uint64_t n_planes = 0, val_bytes = 0;
    uint32_t n_slots = 0;
    for (uint64_t p = 0; p < N1;) {
        const int id = (int)(keyed[p].first >> (64 - kShapeBits));
        uint64_t e = p;
        while (e < N1 && (int)(keyed[e].first >> (64 - kShapeBits)) == id) ++e;
        Shape S{};
        S.fmt = id / kShapesPerFmt;
#if RSEM_GENERAL_G
        {
            const int G = (id % kShapesPerFmt) / 4 + 1;
            int lg = 0;
            while ((1 << lg) < G) ++lg;
            S.lg = ((1 << lg) == G) ? lg : -G;
        }
#else
        S.lg = (id % kShapesPerFmt) / 4;
#endif
        S.K = id % 4 + 1;
        S.row_base = (uint32_t)p;
        S.n_rows = (uint32_t)(e - p);
        const uint32_t rps = shape_R(S);
        S.n_slices = (S.n_rows + rps - 1) / rps;
        S.slice_base = H.n_slices;
        S.plane_base = n_planes;
        S.slot_base = n_slots;
        S.val_base = val_bytes;
        H.n_slices += S.n_slices;
        n_planes += (uint64_t)S.n_slices * S.K;
        n_slots += S.n_slices * rps;
        val_bytes += (uint64_t)S.n_slices * S.K * plane_bytes(S.fmt);
        H.shapes.push_back(S);
        p = e;
    }
    H.ssid.assign(n_planes * 64, 0);
    H.sval.assign(val_bytes + 8, 0);
    H.sncp.assign(n_slots + 1, 0.0);
    H.sexp.assign(n_slots + 1, 0);
    // planes (k_fill_sell)
    for (const Shape& S : H.shapes) {
        const int G = shape_G(S);
        for (uint32_t q = 0; q < S.n_rows; q++) {
            uint32_t sl, r;
            row_to_slot(S, H.T, q, sl, r);
            const uint32_t orig = H.order[S.row_base + q];
            const uint64_t fr = rp[orig];
            const int L = (int)(rp[orig + 1] - fr);
            const uint64_t pl_local = (uint64_t)sl * S.K * 64, pl0 = S.plane_base * 64 + pl_local;
            const uint32_t slot = S.slot_base + sl * shape_R(S) + r;
            Q32Scale qs{0};
            if (S.fmt == kFmtQ32) {
                double vmx = 0.0;
                for (int c = 0; c < L; c++) vmx = fmax(vmx, cp[fr + c]);
                if (!q32_scale_of(vmx, vmx, 0, qs)) { fprintf(stderr, "estep_emu: inconsistent Q32 decision\n"); exit(2); }
                H.sexp[slot] = (int16_t)qs.e;
            }
            for (int c = 0; c < L; c++) {
                const uint64_t off = (uint64_t)(c / G) * 64 + r * G + (c % G);
                H.ssid[pl0 + off] = sid[fr + c];
                if (S.fmt == kFmtQ32) ((uint32_t*)(H.sval.data() + S.val_base))[pl_local + off] = q32_mantissa(cp[fr + c], qs.e);
                else ((double*)(H.sval.data() + S.val_base))[pl_local + off] = cp[fr + c];
            }
            H.sncp[slot] = ncp[orig];
        }
    }
    // masks (k_slice_masks)
    H.masks.assign(H.n_slices, 0);
    for (const Shape& S : H.shapes) {
        const int G = shape_G(S);
        for (uint32_t sl = 0; sl < S.n_slices; sl++) {
            const uint64_t pl0 = (S.plane_base + (uint64_t)sl * S.K) * 64;
            unsigned long long m = 0;
            for (int l = 0; l < 64; l++) {
                bool changed = (sl % H.T == 0);
                for (int k = 0; k < S.K && !changed; k++)
                    changed = H.ssid[pl0 + (uint64_t)k * 64 + l] != H.ssid[pl0 - (uint64_t)S.K * 64 + (uint64_t)k * 64 + l];
                if (changed) m |= 1ull << l;
            }
            unsigned long long full = 0;
            for (int l = 0; l < 64; l++) {
                const int gb = (l / G) * G;
                const unsigned long long grp = (G == 64) ? ~0ull : (((1ull << G) - 1) << gb);
                if (m & grp) full |= 1ull << l;
            }
            H.masks[S.slice_base + sl] = full;
        }
    }
    (void)M;
}

// ---- one workgroup = four blocks of one shape (the wrapper of k_estep_lane, plain theta or theta from counts) -----------
struct Job {
    const HostLayout* H;
    Shape S;
    uint32_t slice_begin, n_slices;
    int base, span, M;
    bool from_counts;
    const double* theta;  // plain: theta[M+1]; from_counts: counts[M+1] followed by 2 * kTotSlots totals
    double N0;
    double* counts;
    double* tot_noise;
    double* tot_neff;
    double th_win[kWindow], cnt_win[kWindow];
    emu::Block blk;
};

template <bool kFC>
static void lane_body(Job* J, int tid) {
    emu::t_tid = tid;
    emu::t_blk = &J->blk;
    const int lane = tid & 63, w = tid >> 6;
    const HostLayout& H = *J->H;
    const Shape& S = J->S;
    const uint32_t per_wave = H.T;
    const uint32_t u_end = S.slice_base + J->slice_begin + J->n_slices;
    const uint32_t s_begin = S.slice_base + J->slice_begin + (uint32_t)w * per_wave;
    const uint32_t s_end = std::min(u_end, s_begin + per_wave);
    const double* tsrc = J->theta + J->M + 1;
    double noise = 0.0, neff = 0.0;
#define EMU_BLOCK(KK, QQ)                                                                                                          \
    estep_block<KK, kFC, QQ, (QQ ? kQ32Depth[KK - 1] : kF64Depth[KK - 1])>(S, s_begin, s_end, lane, J->base, J->span, J->theta, tsrc, J->N0, \
        J->th_win, J->cnt_win, H.sval.data(), H.sexp.data(), H.ssid.data(), H.sncp.data(), H.masks.data(), J->counts, noise, neff, J->M)
    if (s_begin < u_end) switch (S.K + 4 * S.fmt) {
        case 1: EMU_BLOCK(1, false); break;
        case 2: EMU_BLOCK(2, false); break;
        case 3: EMU_BLOCK(3, false); break;
        case 4: EMU_BLOCK(4, false); break;
        case 5: EMU_BLOCK(1, true); break;
        case 6: EMU_BLOCK(2, true); break;
        case 7: EMU_BLOCK(3, true); break;
        default: EMU_BLOCK(4, true); break;
    } else {
        const ThetaSrc th = theta_src<kFC>(J->theta, tsrc, J->N0, lane);
        stage_windows<kFC>(J->base, J->span, J->M, th, J->th_win, J->cnt_win);
    }
#undef EMU_BLOCK
    RSEM_SYNC();
    for (int i = tid; i < J->span; i += 256)
        if (J->cnt_win[i] != 0.0) emu::atomic_add(&J->counts[J->base + i], J->cnt_win[i]);
    emu::atomic_add(J->tot_noise, noise);
    emu::atomic_add(J->tot_neff, neff);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    int32_t hdr[8];
    double N0;
    if (fread(hdr, 4, 8, f) != 8 || fread(&N0, 8, 1, f) != 1) return 3;
    const int M = hdr[0];
    const uint64_t N1 = (uint64_t)hdr[1];
    std::vector<uint64_t> rp(N1 + 1);
    if (fread(rp.data(), 8, N1 + 1, f) != N1 + 1) return 3;
    const uint64_t nnz = rp[N1];
    std::vector<int32_t> sid(nnz);
    std::vector<double> cp(nnz), ncp(N1);
    const bool from_counts = hdr[6] != 0;
    std::vector<double> theta((size_t)M + 1 + (from_counts ? 2 * kTotSlots : 0));
    if (fread(sid.data(), 4, nnz, f) != nnz || fread(cp.data(), 8, nnz, f) != nnz || fread(ncp.data(), 8, N1, f) != N1 ||
        fread(theta.data(), 8, theta.size(), f) != theta.size()) return 3;
    fclose(f);
    HostLayout H;
    H.T = (uint32_t)hdr[2];
    build_layout(H, M, N1, rp.data(), sid.data(), cp.data(), ncp.data(), hdr[3], hdr[4] != 0, hdr[5]);
    std::vector<double> counts((size_t)M + 1, 0.0);
    double tot_noise = 0.0, tot_neff = 0.0;
    Job* J = new Job();
    pthread_barrier_init(&J->blk.bar, nullptr, 256);
    for (int w = 0; w < 4; w++) pthread_barrier_init(&J->blk.w[w].bar, nullptr, 64);
    for (const Shape& S : H.shapes) {
        for (uint32_t b0 = 0; b0 < S.n_slices; b0 += 4 * H.T) {  // a unit = 4 blocks (sell_build_units' full-size unit)
            J->H = &H;
            J->S = S;
            J->slice_begin = b0;
            J->n_slices = std::min<uint32_t>(4 * H.T, S.n_slices - b0);
            int lo = 0x7fffffff, hi = 0;
            for (uint64_t p = (S.plane_base + (uint64_t)b0 * S.K) * 64; p < (S.plane_base + (uint64_t)(b0 + J->n_slices) * S.K) * 64; p++)
                if (H.ssid[p] > 0) { lo = std::min(lo, (int)H.ssid[p]); hi = std::max(hi, (int)H.ssid[p]); }
            if (lo > hi) { lo = 1; hi = 1; }
            J->base = lo;
            J->span = std::min(hi - lo + 1, hdr[7] > 0 ? hdr[7] : kWindow);  // hdr[7]: a smaller window, to force the out-of-window path
            J->M = M;
            J->from_counts = from_counts;
            J->theta = theta.data();
            J->N0 = N0;
            J->counts = counts.data();
            J->tot_noise = &tot_noise;
            J->tot_neff = &tot_neff;
            std::vector<std::thread> th;
            for (int t = 0; t < 256; t++) th.emplace_back(from_counts ? lane_body<true> : lane_body<false>, J, t);
            for (auto& t : th) t.join();
        }
    }
    f = fopen(argv[2], "wb");
    if (!f) return 4;
    fwrite(counts.data(), 8, counts.size(), f);
    fwrite(&tot_noise, 8, 1, f);
    fwrite(&tot_neff, 8, 1, f);
    fclose(f);
    return 0;
}
