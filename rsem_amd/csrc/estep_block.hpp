// estep_block.hpp -- the per-wave body of the LANE E-step kernel (k_estep_lane of em.hip).
//
// Included by em.hip INSIDE its anonymous namespace (after kEpsilon / kTotSlots / sell_layout.hpp are in scope), and by
// tests/estep_emu.cpp, which runs this very code on the CPU -- one OS thread per lane, the cross-lane intrinsics replaced by
// exchanges through memory -- to check the lane mapping, the prefetch rings, the Q32 arithmetic and the prepared variants
// against the oracle without a GPU.  The product build uses the GPU intrinsics directly: everything that differs between
// the two goes through the macros below, which expand to the intrinsic in the product (same tokens, same code).
#pragma once
#include <type_traits>

#include "simt_macros.hpp"

RSEM_DEVFN double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += RSEM_SHFL_XOR(v, d);
    return v;
}


// Variant LANE (default).  One workgroup = one Unit (sell_layout.hpp): each of its 4 waves walks one
// block of T slices.  theta[base .. base+kWindow) is staged in LDS once, counts for the same sid
// window accumulate in LDS (ds_add_f64) and leave the workgroup as ONE device atomic per touched
// sid.  Inside a block every lane follows consecutive sorted reads: while the sid tuple does not
// change (slice mask bit clear) the lane multiplies cached theta values with the streamed conprb
// and adds the normalised fractions into registers; on a change it spills its registers to the
// LDS window and takes sid / theta of the new tuple (the sid planes of a slice are loaded only when its mask is not zero,
// then by all lanes).  The loads of the next slice are issued before slice s is reduced (software pipeline of 2 register
// sets; static instruction stream per K, format and kFar).
// Where theta comes from.  Plain: the array the M-step kernel wrote.  kFC ("from counts", the fused loop of rsem_em_run):
// the PREVIOUS round's raw counts and its two totals -- theta_i = (counts_i + (i == 0 ? noise + N0 : 0)) / (N0 + reads with
// a non-zero normaliser), the very expression the M step evaluates (EM.cpp:392-398), so the E step does not wait for an
// M-step kernel at all; convergence statistics run beside it on a second stream (k_mstep_fast<true>).
struct ThetaSrc {
    const double* v;
    double extra0, sum;
};
template <bool kFC>
RSEM_DEVFN double theta_at(const ThetaSrc& t, int i) {
    if (!kFC) return t.v[i];
    return (t.v[i] + (i == 0 ? t.extra0 : 0.0)) / t.sum;
}
// the two totals of the source round: every wave sums the slots itself (fixed order: identical in all waves)
template <bool kFC>
RSEM_DEVFN ThetaSrc theta_src(const double* __restrict__ v, const double* __restrict__ tsrc, double N0, int lane);

// theta[base, base+span) -> LDS, count window zeroed; every wave of the workgroup calls this exactly once
template <bool kFC>
RSEM_DEVFN void stage_windows(int base, int span, int M, const ThetaSrc& th, double* th_win, double* cnt_win) {
    for (int i = RSEM_TIDX; i < span; i += RSEM_BDIM) {
        const int sidv = base + i;
        th_win[i] = (sidv >= 0 && sidv <= M) ? theta_at<kFC>(th, sidv) : 0.0;
        cnt_win[i] = 0.0;
    }
    RSEM_SYNC();
}

// Split rows (F64X shapes, sell_layout.hpp): the row holds a read's in-window alignments only.  extra[slot - slot_base] =
// the sum of theta * conprb over the read's other alignments (written before this kernel by k_far_rowsum), added to the
// read's normaliser here; inv[slot - slot_base] = the reciprocal of the normaliser (0 when the read carries no mass), handed
// on to k_far_colsum, which adds the far alignments' fractions to the counts in transcript order.
struct XArgs {
    const double* extra = nullptr;
    double* inv = nullptr;
    uint32_t slot_base = 0;
};

// The far queue of a wave (kFQ: the launch over the units with ids outside their window, em.hip).  A lane that spills a partial count
// for an id outside its unit's window takes a place here -- (id, value) in LDS -- instead of issuing a global atomic: with an atomic
// possibly in flight (they do not return in order with loads) every wait the compiler places in the loop is a wait for EVERYTHING, and
// the loads of the next slice, issued a moment ago, are waited for before this slice is reduced: the units of far-reaching reads ran
// without any overlap of loads and arithmetic (rounds 3-5; profiles/r06d_xrows_probe_units_fit_window.log: 10 % of the reads, a
// fifth of the launch).  The queue is emptied -- its atomics issued AND waited for, RSEM_WAIT_VM0, so that the compiler sees none in
// flight behind that point -- where a new tuple starts and fewer than a slice's worth of places are left, and at the block's end.
#ifndef RSEM_FARQ_CAP
#define RSEM_FARQ_CAP 384
#endif
constexpr int kFarQCap = RSEM_FARQ_CAP;  // places per wave: > 64 * 4 (what one slice can append) + room to make emptying rare (tests build it smaller)
static_assert(kFarQCap > 256, "a slice of K = 4 planes can append 256 entries");
struct FarQueue {
    int* sid = nullptr;     // [kFarQCap]
    double* val = nullptr;  // [kFarQCap]
    int* n = nullptr;       // places taken
};

// one slice's loads: sids (only where a tuple starts), values (doubles, or Q32 mantissas + the read's exponent), noise
template <int K, bool kQ>
struct SliceRegs {
    int id[K];
    typename std::conditional<kQ, uint32_t, double>::type c[K];
    double nc;
    int e;
    double x;        // split rows: the far part of the normaliser
    uint32_t slot0;  // first row slot of the slice (split rows: where the reciprocal goes)
    double traw[K];  // kFQ: the theta source's words for the ids outside the window, requested a slice ahead (pregather)
};

// 2^e for the exponents q32_scale_of admits (always a normal double)
RSEM_DEVFN double pow2_of(int e) { return RSEM_LL_AS_DOUBLE((long long)(1023 + e) << 52); }

// How many slices a wave keeps in flight (register sets), per K = 1..4.  Two: the next slice's loads fly while this one is
// reduced.  Until round 3 that was true on paper only -- the waits the compiler placed were waits for everything (flat
// loads, loads under lane predicates and scalar branches, a loop that could leave in the middle: see take_tuples, issue
// and the loops below) -- and deeper rings of the smaller Q32 sets (8 / 6 / 4 / 3) bought back some of the lost overlap.
// With exact waits they buy nothing (profiles/r03p_pipelined_waits.log: Q32 at C3 0.713 ms with 2 sets, 0.729 with 4 / 4 /
// 3 / 2; F64 3 / 3 / 2 / 2 0.970 against 0.975) and cost registers; the ring loop stays for depths > 2 (tests build it).
#ifndef RSEM_F64_DEPTHS
#define RSEM_F64_DEPTHS 2, 2, 2, 2
#endif
#ifndef RSEM_Q32_DEPTHS
#define RSEM_Q32_DEPTHS 2, 2, 2, 2
#endif
// Measured in round 3 (profiles/r03a_variants_and_steps.log, C3 / C2, F64 / Q32 launch) and adopted: the per-read normaliser's
// butterfly over 2..16 lanes with DPP moves instead of ds_bpermute (same additions in the same order: bit-identical; Q32
// -4.6 %), the count spill spelled as an LDS atomic (ds_add_f64) or a global one instead of ONE flat atomic on a selected
// address (Q32 -4 %), and the non-temporal hint on the value planes (read once per launch: they no longer push theta /
// counts / sid planes out of L2 and the Infinity Cache; F64 -4 % at C3, -12 % at C2).  Measured and dropped (within +-1 %):
// reciprocal by Newton steps, clamp fast path, fused multiply-add accumulation, mantissa conversion by an exponent word,
// ballot-counted normalisers, lane groups that are not a power of two (+5 %: the cheaper planes lost to the dearer
// lane arithmetic); the non-temporal hint on the sid planes, noise / exponent slots and slice masks as well (profiles/
// r03b_variants.log: F64 at C3 -0.5 %, everything else +3..+8 %: those lines are re-used across neighbouring workgroups).
template <typename T>
RSEM_DEVFN T stream_load(const T* p) { return RSEM_NT_LOAD(p); }
template <int kCtrl>
RSEM_DEVFN double dpp_take(double v) {  // the value of the lane the DPP control selects (all lanes active here)
    const long long b = RSEM_DOUBLE_AS_LL(v);
    const int lo = RSEM_DPP_MOV((int)(unsigned)b, kCtrl);
    const int hi = RSEM_DPP_MOV((int)(unsigned)(b >> 32), kCtrl);
    return RSEM_LL_AS_DOUBLE(((long long)hi << 32) | (unsigned)lo);
}
// sum over the 2^lg lanes of a read, every lane ends with the total.  Steps 1, 2: quad permutes; 4: row_half_mirror (the
// other quad of the 8-group: all its lanes hold the same quad sum); 8: row_mirror; 16, 32: ds_bpermute as before.
RSEM_DEVFN double read_sum_dpp(double part, int lg) {
    if (lg >= 1) part += dpp_take<0xB1>(part);   // quad_perm [1,0,3,2]
    if (lg >= 2) part += dpp_take<0x4E>(part);   // quad_perm [2,3,0,1]
    if (lg >= 3) part += dpp_take<0x141>(part);  // row_half_mirror
    if (lg >= 4) part += dpp_take<0x140>(part);  // row_mirror
    for (int d = 16; d < (1 << lg); d <<= 1) part += RSEM_SHFL_XOR(part, d);
    return part;
}
constexpr int kF64Depth[4] = {RSEM_F64_DEPTHS};
constexpr int kQ32Depth[4] = {RSEM_Q32_DEPTHS};

// kFar: the unit has ids outside its window (Unit::pad[0], found at layout time: sell_flag_far_units).  A unit WITHOUT
// runs a loop that touches global memory only through its streaming loads: theta and counts in LDS, no gather, no global
// atomic -- with a global atomic possibly in flight (they do not return in order with loads) every wait the compiler
// places in the loop is a wait for everything, and a gather in the middle of a slice drains the prefetch.
template <int K, bool kFC, bool kQ, int NBUF, bool kFar, bool kX = false, bool kFQ = false>
RSEM_DEVFN void estep_block(const Shape& S, uint32_t s_begin, uint32_t s_end, int lane, int base, int span,
                                   const double* __restrict__ theta, const double* __restrict__ tsrc, double N0, double* th_win, double* cnt_win,
                                   const unsigned char* __restrict__ sval, const int16_t* __restrict__ sexp, const int32_t* __restrict__ ssid,
                                   const double* __restrict__ sncp, const unsigned long long* __restrict__ masks,
                                   double* counts, double& noise, double& neff, int M, const XArgs& X = XArgs(), const FarQueue& FQ = FarQueue()) {
    static_assert(!kFQ || kFar, "the far queue belongs to the loop of the units with ids outside their window");
    static_assert(!kFQ || NBUF >= 3, "the far-queue loop requests theta a slice ahead: three register sets at least (the ring loop)");
    using ValT = typename std::conditional<kQ, uint32_t, double>::type;
    const ValT* __restrict__ scp = (const ValT*)(sval + S.val_base);  // this shape's value planes
    const int lg = S.lg;
    const int g = lane & ((1 << lg) - 1);
    const bool g0 = (g == 0);
    const uint32_t R = 64u >> lg;
    // 64 slices' masks at a time, one per lane
    uint32_t m_base = s_begin;
    unsigned long long mv = (s_begin + lane < s_end) ? masks[s_begin + lane] : ~0ull;
    auto mask_of = [&](uint32_t t) -> unsigned long long {
        if (t - m_base >= 64u) {
            m_base = t;
            mv = (t + lane < s_end) ? masks[t + lane] : ~0ull;
            RSEM_PIN(mv);  // (the wait for this load stays in this rare branch: at the join it would be a wait for everything, every slice)
        }
        const int src = (int)(t - m_base);
        const uint32_t lo = RSEM_READLANE((int)(uint32_t)mv, src);
        const uint32_t hi = RSEM_READLANE((int)(uint32_t)(mv >> 32), src);
        return ((unsigned long long)hi << 32) | lo;
    };
    // Everything about a slice except the lane is uniform over the wave (s_begin / s_end come from the wave index, which
    // the caller hands over through readfirstlane): the slice's base addresses are scalar, a lane adds its own constant
    // offset -- no per-lane 64-bit address arithmetic.  The sids of a slice are loaded by all lanes or (mask 0, a scalar
    // branch) by none: lanes that do not start a tuple ignore theirs.
    const unsigned ulane = (unsigned)lane, uslot = ulane >> lg;
    auto issue = [&](uint32_t t, unsigned long long m, SliceRegs<K, kQ>& b) {
        const uint32_t sl = t - S.slice_base;
        const uint64_t v0 = (uint64_t)sl * (K * 64);              // first entry of the slice within the shape's planes
        const ValT* __restrict__ vp = scp + v0;
        // The sid planes of a slice are read only where a tuple starts in it (m != 0), then by all lanes: one scalar branch.
        // (Behind a branch the compiler counts the loads in flight as on the path with the fewest, so its wait for this
        // slice's planes also waits for the next slice's sid planes where that slice has them.  Issuing the same K loads
        // always -- to the shape's first slice where m == 0: L1 / L2 hits, exact waits on every path -- was measured and
        // is slower: 0.971 against 0.936 ms at configs[2], Q32 0.713 against 0.700, profiles/r03t; tried again in round 6 for the
        // units of split rows alone, where a tuple starts in most slices: no difference either way, profiles/r06b_xrows_probe.log,
        // r06c_xrows_probe.log.)
        if (kFQ || m != 0ull) {  // (the far-queue loop reads the ids of every slice: a tuple starts in nearly all of them, and pregather wants no branch)
            const int32_t* __restrict__ ip = ssid + (S.plane_base * 64 + v0);
#pragma unroll
            for (int k = 0; k < K; k++) b.id[k] = ip[k * 64 + ulane];
        }
#pragma unroll
        for (int k = 0; k < K; k++) b.c[k] = stream_load(&vp[k * 64 + ulane]);
        const uint32_t slot0 = S.slot_base + sl * R;
        // (all lanes of a read load its noise probability and exponent, not only the first: a load under a lane predicate
        // is a branch to the compiler, and with a conditional load in flight its waits for the OTHER loads turn into
        // waits for everything)
        b.nc = (sncp + slot0)[uslot];
        b.e = kQ ? (int)(sexp + slot0)[uslot] : 0;
        b.x = kX ? (X.extra + (slot0 - X.slot_base))[uslot] : 0.0;  // (unconditional within the instantiation, like the noise value)
        b.slot0 = slot0;
    };
    auto spill = [&](const int* rsid, double* acc) {  // lane-private partial counts -> LDS window
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (acc[k] != 0.0) {
                const unsigned off = (unsigned)(rsid[k] - base);
                if (!kFar) RSEM_LDS_ADD(&cnt_win[off < (unsigned)span ? off : 0u], acc[k]);  // (the clamp never acts: no id of the unit is outside)
                else if (off < (unsigned)span) RSEM_LDS_ADD(&cnt_win[off], acc[k]);
                else if (kFQ) {
                    const int at = RSEM_LDS_FETCH_ADD_I32(FQ.n, 1);  // (room for a whole slice was made before this slice: far_queue_room)
                    FQ.sid[at] = rsid[k];
                    FQ.val[at] = acc[k];
                } else RSEM_ATOMIC_ADD(&counts[rsid[k]], acc[k]);
            }
            acc[k] = 0.0;
        }
    };
    // kFQ, all lanes: empty the wave's queue if a slice's worth of appends (64 K) might not fit any more, or if `all`
    auto far_queue_room = [&](bool all) {
        if (!kFQ) return;
        RSEM_WAVE_SYNC();
        const int n = RSEM_READFIRSTLANE(*FQ.n);
        if (n > (all ? 0 : kFarQCap - 64 * K)) {  // (uniform over the wave)
            for (int i = lane; i < n; i += 64) RSEM_ATOMIC_ADD(&counts[FQ.sid[i]], FQ.val[i]);
            RSEM_WAIT_VM0();
            RSEM_WAVE_SYNC();
            if (lane == 0) *FQ.n = 0;
            RSEM_WAVE_SYNC();
        }
    };

    int rsid[K];
    double rth[K], acc[K];
#pragma unroll
    for (int k = 0; k < K; k++) { rsid[k] = 0; rth[k] = 0.0; acc[k] = 0.0; }
    ThetaSrc th{theta, 0.0, 1.0};
    double th0 = 0.0;
    // kFQ: theta of the ids outside the window is requested ONE SLICE AHEAD of its use -- until round 6 a lane that started a tuple
    // with such an id asked for it inside reduce() and the wave waited a whole round trip to memory, every slice of a unit of
    // far-reaching reads.  Every lane issues its K loads whatever it needs (word 0 where it needs none: a cache hit), so that no
    // path through the loop has fewer loads in flight than another and the compiler's waits stay exact.
    auto pregather = [&](SliceRegs<K, kQ>& b, unsigned long long m) {
        if (!kFQ) return;
        const bool starts = ((m >> lane) & 1ull) != 0ull;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int sidv = b.id[k];
            const bool want = starts && !((unsigned)(sidv - base) < (unsigned)span);
            b.traw[k] = th.v[want ? sidv : 0];
        }
    };
    auto reduce = [&](const SliceRegs<K, kQ>& cur, unsigned long long cur_m) {
        if (cur_m != 0ull) {                 // wave-uniform
            far_queue_room(false);
            if ((cur_m >> lane) & 1ull) {    // lanes whose read starts a new sid tuple
                spill(rsid, acc);
                // theta of the new tuple: the window (LDS) for every id -- a clamped offset where the id is outside --, then,
                // in a branch of its own, global memory for the ids outside.  NOT `in ? th_win[off] : theta[sid]`: the
                // compiler turns that into one FLAT load of a selected address, and after a flat load every wait is
                // vmcnt(0) lgkmcnt(0) -- at the join below, i.e. in every slice, with or without a new tuple: the next
                // slice's loads, issued a moment ago, were waited for before this slice was reduced (no prefetch at all;
                // profiles/r03p).  The branch consumes its loads itself (RSEM_PIN), so the common path waits only for
                // what it needs.
                bool far = false;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const int sidv = cur.id[k];
                    rsid[k] = sidv;
                    const unsigned off = (unsigned)(sidv - base);
                    const bool in = off < (unsigned)span;
                    rth[k] = th_win[in ? off : 0u];
                    far = far || !in;
                }
                if (kFQ) {  // (requested a slice ago: pregather)
#pragma unroll
                    for (int k = 0; k < K; k++)
                        if (!((unsigned)(rsid[k] - base) < (unsigned)span))
                            rth[k] = kFC ? (cur.traw[k] + (rsid[k] == 0 ? th.extra0 : 0.0)) / th.sum : cur.traw[k];
                } else if (kFar && far) {
                    double t[K];
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        const bool in = (unsigned)(rsid[k] - base) < (unsigned)span;
                        t[k] = theta_at<kFC>(th, in ? 0 : rsid[k]);
                    }
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        RSEM_PIN(t[k]);
                        if (!((unsigned)(rsid[k] - base) < (unsigned)span)) rth[k] = t[k];
                    }
                }
            }
        }
        double f0 = g0 ? th0 * cur.nc : 0.0;
        if (f0 < kEpsilon) f0 = 0.0;
        double f[K];
        double part = f0 + ((kX && g0) ? cur.x : 0.0);
        const double scale = kQ ? pow2_of(cur.e) : 1.0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            // Q32: mantissa * 2^e is exact, so this is the F64 expression on the rounded value
            const double cv = kQ ? (double)cur.c[k] * scale : (double)cur.c[k];
            double v = rth[k] * cv;
            if (v < kEpsilon) v = 0.0;
            f[k] = v;
            part += v;
        }
        part = read_sum_dpp(part, lg);
        const double inv = (part >= kEpsilon) ? 1.0 / part : 0.0;
        // (every lane of the read stores the same value rather than its first lane alone: no lane predicate, no branch; the
        // store costs the split shapes' loop 90 us of 1.2 ms at configs[2] with 10 % cross-gene reads either way,
        // profiles/r04d_call.log)
        if (kX) RSEM_STORE_SAME(&(X.inv + (cur.slot0 - X.slot_base))[uslot], inv);
        noise += f0 * inv;
        // reads whose fractions sum to one: sum(counts) without a reduction
        neff += (g0 && part >= kEpsilon) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < K; k++) acc[k] = acc[k] + f[k] * inv;
    };
    if constexpr (NBUF == 2) {
        // ping-pong register sets A / B: the loads of the next slice are in flight while this one is reduced
        SliceRegs<K, kQ> A, B;
        unsigned long long mA = ~0ull, mB = 0;  // a block always starts fresh
        issue(s_begin, mA, A);
        th = theta_src<kFC>(theta, tsrc, N0, lane);  // (after the first slice's loads were issued: they fly meanwhile)
        th0 = theta_at<kFC>(th, 0);
        stage_windows<kFC>(base, span, M, th, th_win, cnt_win);  // ... and while the windows are staged
        // The steady loop issues UNCONDITIONALLY (the tail is peeled off): where a path that issued nothing joins one that
        // did, the compiler's wait for an older load is the one that is right for the path with the fewest loads behind it --
        // on the other path a wait for the loads just issued, i.e. no prefetch.
        uint32_t s = s_begin;
        for (; s + 2 < s_end; s += 2) {
            mB = mask_of(s + 1);
            issue(s + 1, mB, B);
            reduce(A, mA);
            mA = mask_of(s + 2);
            issue(s + 2, mA, A);
            reduce(B, mB);
        }
        if (s + 1 < s_end) {
            mB = mask_of(s + 1);
            issue(s + 1, mB, B);
            reduce(A, mA);
            reduce(B, mB);
        } else {
            reduce(A, mA);
        }
    } else {
        // ring of NBUF register sets (fully unrolled: every index is static): NBUF - 1 slices in flight while one is reduced
        SliceRegs<K, kQ> buf[NBUF];
        unsigned long long mk[NBUF];
        mk[0] = ~0ull;  // a block always starts fresh
        issue(s_begin, mk[0], buf[0]);
        th = theta_src<kFC>(theta, tsrc, N0, lane);
        th0 = theta_at<kFC>(th, 0);
        stage_windows<kFC>(base, span, M, th, th_win, cnt_win);
        // Every issue is UNCONDITIONAL (see the two-set loop above): past the block's end the last slice is simply loaded
        // again (NBUF - 1 redundant slices per block of ~160, never reduced), and the steady loop runs over whole groups
        // of NBUF slices; what is left (< NBUF slices, all issued by then) is reduced after it.
        auto within = [&](uint32_t t) -> uint32_t { return t < s_end ? t : s_end - 1u; };
#pragma unroll
        for (int j = 1; j < NBUF - 1; j++) {
            const uint32_t tt = within(s_begin + j);
            mk[j] = mask_of(tt);
            issue(tt, mk[j], buf[j]);
        }
        pregather(buf[0], mk[0]);
        uint32_t s = s_begin;
        for (; s + NBUF <= s_end; s += NBUF) {
#pragma unroll
            for (int j = 0; j < NBUF; j++) {
                constexpr int ahead = NBUF - 1;
                const int nj = (j + ahead) % NBUF;  // static after unrolling
                const uint32_t tt = within(s + j + ahead);
                mk[nj] = mask_of(tt);
                issue(tt, mk[nj], buf[nj]);
                pregather(buf[(j + 1) % NBUF], mk[(j + 1) % NBUF]);  // (the slice behind this one: its ids were requested a step ago)
                reduce(buf[j], mk[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < NBUF - 1; j++)
            if (s + j < s_end) {  // (uniform over the wave)
                if (j > 0) pregather(buf[j], mk[j]);
                reduce(buf[j], mk[j]);
            }
    }
    far_queue_room(false);
    spill(rsid, acc);
    far_queue_room(true);
}

template <bool kFC>
RSEM_DEVFN ThetaSrc theta_src(const double* __restrict__ v, const double* __restrict__ tsrc, double N0, int lane) {
    ThetaSrc t{v, 0.0, 1.0};
    if (kFC) {
        const double a = wave_sum(tsrc[lane]);
        const double b = wave_sum(tsrc[kTotSlots + lane]);
        t.extra0 = a + N0;  // counts[0] += noise + N0 (EM.cpp:392)
        t.sum = b + N0;     // = sum(counts) (EM.cpp:395): every read with a non-zero normaliser carries mass one
    }
    return t;
}

