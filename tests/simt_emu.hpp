// simt_emu.hpp -- TEST INFRASTRUCTURE shared by tests/estep_emu.cpp and tests/gibbs_emu.cpp: a CPU stand-in for one workgroup
// of four 64-lane waves (one OS thread per lane; the cross-lane intrinsics of rsem_amd/csrc/simt_macros.hpp as exchanges
// through memory with a barrier on either side; atomics as CAS loops) and the sliced layout rebuilt on the host with
// sell_layout.hpp's own index helpers.  Define RSEM_EMU before including.  Never part of the product.
#pragma once
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
    if (ctrl >= 0 && ctrl < 0x100) return (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);  // quad_perm
    if (ctrl > 0x110 && ctrl < 0x120) return ((l & 15) >= (ctrl & 15)) ? l - (ctrl & 15) : -1;  // row_shr (no source: own value here, 0 on the GPU; callers mask it)
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
inline void store_same(double* p, double v) {
    unsigned long long u;
    memcpy(&u, &v, 8);
    reinterpret_cast<std::atomic<unsigned long long>*>(p)->store(u, std::memory_order_relaxed);
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
#define RSEM_SHFL_UP(v, d) emu::exchange(v, emu::lane() - (d))
#define RSEM_SHFL(v, src) emu::exchange(v, (src) & 63)
#define RSEM_BALLOT(p) emu::ballot(p)
#define RSEM_READLANE(v, src) emu::exchange(v, src)
#define RSEM_ATOMIC_ADD(p, v) emu::atomic_add(p, v)
#define RSEM_ATOMIC_ADD_I32(p, v) __atomic_fetch_add(p, v, __ATOMIC_RELAXED)
#define RSEM_LDS_ADD(p, v) emu::atomic_add(p, v)
#define RSEM_LDS_ADD_I32(p, v) __atomic_fetch_add(p, v, __ATOMIC_RELAXED)
#define RSEM_LDS_FETCH_ADD_I32(p, v) __atomic_fetch_add(p, v, __ATOMIC_RELAXED)
#define RSEM_WAVE_SYNC() pthread_barrier_wait(&emu::wave().bar)
#define RSEM_WAIT_VM0() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define RSEM_READFIRSTLANE(v) emu::exchange(v, 0)
#define RSEM_PIN(x) (void)(x)
#define RSEM_SCHED_FENCE() (void)0
#define RSEM_RCP(x) (1.0 / (x) * (1.0 + 3e-8))  /* v_rcp_f64 is not exact either: the Newton steps must repair this */
#define RSEM_DPP_MOV(v, ctrl) emu::exchange(v, emu::dpp_src(ctrl))
#define RSEM_LL_AS_DOUBLE(x) emu::ll_as_double(x)
#define RSEM_DOUBLE_AS_LL(x) emu::double_as_ll(x)
#define RSEM_NT_LOAD(p) (*(p))
#define RSEM_STORE_SAME(p, v) emu::store_same(p, v)

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
    // split rows (policy 2): far entries in file order, each with the row slot of its read; extra / inv per slot from x_slot_base
    struct Far { int32_t sid; double cp; uint32_t slot; };
    std::vector<Far> far;
    uint32_t x_slot_base = 0, n_slots = 0;
    bool has_x = false;
};

static void build_layout(HostLayout& H, int M, uint64_t N1, const uint64_t* rp, const int32_t* sid, const double* cp, const double* ncp,
                         int policy, bool q32, int range_bits, int apart = kLayoutWindow) {
    // keys: sell_layout.hpp's row_key_of (the body of k_row_keys); sorted rows; shapes (the host loop of sell_build)
    std::vector<std::pair<uint64_t, uint32_t>> keyed(N1);
    (void)policy;
    for (uint64_t i = 0; i < N1; i++) {
        int err = 0;
        const uint64_t key = row_key_of(i, M, rp, sid, q32 ? cp : nullptr, range_bits, apart, &err, policy == 2 ? 1 : (policy == 3 ? 2 : 0));  // policy 3: every read that reaches beyond its window splits
        if (err) { fprintf(stderr, "simt_emu: bad CSR (%d)\n", err); exit(2); }
        if ((int)(key >> (64 - kShapeBits)) == kLongShape) { fprintf(stderr, "simt_emu: rows with more than 256 alignments are not modelled\n"); exit(2); }
        keyed[i] = {key, (uint32_t)i};
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
        S.lg = (id % kShapesPerFmt) / 4;
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
        if (S.fmt == kFmtF64X && !H.has_x) { H.x_slot_base = n_slots; H.has_x = true; }
        n_slots += S.n_slices * rps;
        val_bytes += (uint64_t)S.n_slices * S.K * plane_bytes(S.fmt);
        H.shapes.push_back(S);
        p = e;
    }
    H.ssid.assign(n_planes * 64, 0);
    H.sval.assign(val_bytes + 8, 0);
    H.sncp.assign(n_slots + 1, 0.0);
    H.sexp.assign(n_slots + 1, 0);
    H.n_slots = n_slots;
    // planes: sell_fill_row (the body of k_fill_sell)
    for (const Shape& S : H.shapes)
        for (uint32_t q = 0; q < S.n_rows; q++) {
            {   // slot_to_row (what k_mark_stray_reads finds a read by) inverts row_to_slot
                uint32_t sl, r, back = ~0u;
                row_to_slot(S, H.T, q, sl, r);
                if (!slot_to_row(S, H.T, sl, r, back) || back != q) { fprintf(stderr, "simt_emu: slot_to_row(row_to_slot(%u)) = %u\n", q, back); exit(2); }
            }
            int err = 0;
            const uint32_t anchor = (uint32_t)((keyed[S.row_base + q].first >> 32) & kKeyMinSidCap);
            sell_fill_row<true>(S, H.T, S.row_base + q, H.order.data(), rp, sid, cp, ncp, H.ssid.data(), H.sval.data(), H.sncp.data(), H.sexp.data(), &err,
                                anchor, apart);
            if (err) { fprintf(stderr, "simt_emu: inconsistent Q32 decision\n"); exit(2); }
            if (S.fmt == kFmtF64X) {  // the far entries of a split row (k_x_far of sell_layout.hpp)
                uint32_t sl, r;
                row_to_slot(S, H.T, q, sl, r);
                const uint32_t slot = S.slot_base + sl * shape_R(S) + r;
                const uint32_t orig = H.order[S.row_base + q];
                for (uint64_t j = rp[orig]; j < rp[orig + 1]; j++)
                    if (!in_split_window(sid[j], anchor, apart)) H.far.push_back({sid[j], cp ? cp[j] : 0.0, slot});
            }
        }
    // masks: slice_lane_changed / read_lanes_of (the body of k_slice_masks; its two ballots are the loops over l)
    H.masks.assign(H.n_slices, 0);
    for (const Shape& S : H.shapes)
        for (uint32_t sl = 0; sl < S.n_slices; sl++) {
            unsigned long long m = 0, full = 0;
            for (int l = 0; l < 64; l++)
                if (slice_lane_changed(S, H.T, sl, l, H.ssid.data())) m |= 1ull << l;
            for (int l = 0; l < 64; l++)
                if (m & read_lanes_of(S, l)) full |= 1ull << l;
            H.masks[S.slice_base + sl] = full;
        }
    (void)M;
}

