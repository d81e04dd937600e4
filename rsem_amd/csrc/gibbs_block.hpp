// gibbs_block.hpp -- the per-wave body of the Gibbs PARALLEL sweep kernel (k_sample_z_lane of gibbs.hip).
//
// Included by gibbs.hip INSIDE its anonymous namespace (after kEpsilon / Philox / sell_layout.hpp are in scope) and by
// tests/gibbs_emu.cpp, which runs this very code on the CPU (tests/simt_emu.hpp).  Intrinsics go through simt_macros.hpp.
#pragma once
#include "simt_macros.hpp"

// Measured in round 3 and adopted (profiles/r03a_variants_and_steps.log, r03b_variants.log, r03c_exact_and_sweep.log:
// 1.538 -> 1.35 ms per sweep at C3): scalar slice addressing with a per-slice table of sorted positions (no divisions by T
// and R in every slice), the read's uniform from Philox2x32-10 (64 bits per call) instead of Philox4x32-10, the non-temporal
// hint on the value planes.  Measured and dropped (within the +-3 % run-to-run noise of this kernel, identical picks): the
// scan over a read's lanes with DPP moves.  The read's lanes sharing the work of its next G uniforms was within the noise
// too while every wait of the loop was a wait for everything; with exact waits (estep_block.hpp) the kernel is bound by
// its instructions and the sharing is worth 4-6 % (profiles/r03z): adopted, positions by arithmetic instead of a table.
// g[base, base+span) -> LDS, count window zeroed; every wave of the workgroup calls this exactly once
RSEM_DEVFN void stage_gwindows(int base, int span, int M, const double* __restrict__ g, double* g_win, int* cnt_win) {
    for (int i = RSEM_TIDX; i < span; i += RSEM_BDIM) {
        const int sidv = base + i;
        g_win[i] = (sidv >= 0 && sidv <= M) ? g[sidv] : 0.0;
        cnt_win[i] = 0;
    }
    RSEM_SYNC();
}

template <int K>
struct SliceRegs {
    int id[K];
    double c[K];
    double nc;
};

// of a slice: the sorted position of the read in row slot 0 and the slot stride of its block, so that the position of the
// read in slot r (the key of its random number) is x + r * y; within a block x grows by one per slice
struct PtabEntry { uint32_t x, y; };
__host__ RSEM_DEVFN PtabEntry slice_ptab_entry(const Shape& S, uint32_t T, uint32_t sl) {
    const uint32_t R = shape_R(S);
    const uint32_t b = sl / T, t = sl % T;
    const uint32_t left = S.n_rows - b * R * T;
    const uint32_t nb = left < R * T ? left : R * T;
    return PtabEntry{S.row_base + b * R * T + t, (nb + R - 1) / R};
}

// z_i | g for the reads of one block (T slices, one wave), lane-major runs as in the E step: a lane
// keeps the g values and integer pick counters of its current sid tuple in registers and spills
// them to the workgroup's LDS window when the tuple changes.  Weight order inside a read: noise,
// then the G lanes of the read in order, each lane's K planes in order.
// kFar: the unit has ids outside its window (estep_block.hpp); units without never leave LDS inside the loop
template <int K, bool kFar>
RSEM_DEVFN void gibbs_block(const Shape& S, uint32_t T, uint32_t s_begin, uint32_t s_end, int lane, int base, int span,
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
            RSEM_PIN(mv);  // (the wait for this load stays in this rare branch: at the join it would be a wait for everything, every slice)
        }
        const int src = (int)(t - m_base);
        const uint32_t lo = RSEM_READLANE((int)(uint32_t)mv, src);
        const uint32_t hi = RSEM_READLANE((int)(uint32_t)(mv >> 32), src);
        return ((unsigned long long)hi << 32) | lo;
    };
    auto issue = [&](uint32_t t, unsigned long long m, SliceRegs<K>& b) {
        const uint32_t sl = t - S.slice_base;
        // as in the E step (em.hip, estep_block): scalar slice bases + a constant lane offset; the sids of a slice are
        // loaded by all lanes or (mask 0) by none
        const uint64_t p0 = (S.plane_base + (uint64_t)sl * K) * 64;
        const unsigned ulane = (unsigned)lane;
        if (m != 0ull) {  // (a scalar branch; loading the K planes always, for exact waits on every path, is slower: estep_block.hpp, profiles/r03u)
            const int32_t* __restrict__ ip = ssid + p0;
#pragma unroll
            for (int k = 0; k < K; k++) b.id[k] = ip[k * 64 + ulane];
        }
        const double* __restrict__ vp = scp + p0;
#pragma unroll
        for (int k = 0; k < K; k++) b.c[k] = RSEM_NT_LOAD(&vp[k * 64 + ulane]);
        b.nc = (sncp + (S.slot_base + sl * R))[ulane >> lg];  // (all lanes of the read: no load under a lane predicate)
    };
    // the sorted position of the read in row slot r of slice t (the key of its random number), by arithmetic: a wave's
    // slices lie in at most two blocks, whose first positions and slot strides are taken once (slice_ptab_entry)
    const uint32_t sl0 = s_begin - S.slice_base;
    const PtabEntry eA = slice_ptab_entry(S, T, sl0);
    const uint32_t startB = S.slice_base + (sl0 / T + 1) * T;  // first slice of the next block (possibly >= s_end)
    const PtabEntry eB = startB < s_end ? slice_ptab_entry(S, T, startB - S.slice_base) : eA;
    auto pos_of = [&](uint32_t t, uint32_t r) -> uint32_t {
        return t < startB ? eA.x + (t - s_begin) + r * eA.y : eB.x + (t - startB) + r * eB.y;
    };
    double u_bank = 0.0;
    int rsid[K], acc[K];
    double rg[K];
#pragma unroll
    for (int k = 0; k < K; k++) { rsid[k] = 0; acc[k] = 0; rg[k] = 0.0; }
    auto spill = [&]() {
        // (LDS atomics spelled out: one atomic on a selected address would be a FLAT one)
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (acc[k] != 0) {
                const unsigned off = (unsigned)(rsid[k] - base);
                if (!kFar) RSEM_LDS_ADD_I32(&cnt_win[off < (unsigned)span ? off : 0u], acc[k]);  // (the clamp never acts)
                else if (off < (unsigned)span) RSEM_LDS_ADD_I32(&cnt_win[off], acc[k]);
                else RSEM_ATOMIC_ADD_I32(&counts[rsid[k]], acc[k]);
            }
            acc[k] = 0;
        }
    };
    auto sample = [&](const SliceRegs<K>& cur, unsigned long long cur_m, uint32_t s) {
        if (cur_m != 0ull) {
            if ((cur_m >> lane) & 1ull) {
                spill();
                // as in the E step (estep_block.hpp): LDS for every id, a branch of its own for the ids outside the window --
                // never one flat load of a selected address, after which every wait is for everything in flight
                bool far = false;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const int sidv = cur.id[k];
                    rsid[k] = sidv;
                    const unsigned off = (unsigned)(sidv - base);
                    const bool in = off < (unsigned)span;
                    rg[k] = g_win[in ? off : 0u];
                    far = far || !in;
                }
                if (kFar && far) {
                    double t[K];
#pragma unroll
                    for (int k = 0; k < K; k++) t[k] = g[(unsigned)(rsid[k] - base) < (unsigned)span ? 0 : rsid[k]];
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        RSEM_PIN(t[k]);
                        if (!((unsigned)(rsid[k] - base) < (unsigned)span)) rg[k] = t[k];
                    }
                }
            }
        }
        const double f0 = g0lane ? g0 * cur.nc : 0.0;
        double f[K];
        double part = f0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            f[k] = rg[k] * cur.c[k];
            part += f[k];
        }
        double incl = part;  // inclusive scan over the G lanes of the read
        for (int d = 1; d < G; d <<= 1) {
            double o = RSEM_SHFL_UP(incl, d);
            if (gl >= d) incl += o;
        }
        // (one lane per read: nothing to exchange -- a uniform branch, lg comes from the unit descriptor)
        double excl = 0.0, total = incl;
        if (lg > 0) {
            excl = RSEM_SHFL_UP(incl, 1);
            if (gl == 0) excl = 0.0;
            total = RSEM_SHFL(incl, gbase + G - 1);
        }
        // one uniform per read, keyed by the read's position in the sorted order (layout independent)
        const uint32_t key = ph.k0 ^ ((ph.k1 << 13) | (ph.k1 >> 19)) ^ 0x5a5a5a5au;
        double u;
        // The G lanes of a read take turns: every G slices lane i of the read draws the read slot's uniform for slice s + i
        // (all lanes busy in one Philox pass instead of one lane in G, G times), and each slice fetches its uniform from the
        // lane that drew it.  Same uniform for the same (sorted position, sweep) as before: the picks do not change.
        if (lg == 0) {
            uint32_t rnd[2];
            rsem::philox2x32_10(key, pos_of(s, (uint32_t)lane), sweep, rnd);
            u = u53(rnd[0], rnd[1]);
        } else {
            const uint32_t phase = (s - s_begin) & (uint32_t)(G - 1);  // (uniform over the wave)
            if (phase == 0u) {
                const uint32_t t = s + (uint32_t)gl;
                uint32_t rnd[2] = {0, 0};
                if (t < s_end) rsem::philox2x32_10(key, pos_of(t, (uint32_t)lane >> lg), sweep, rnd);
                u_bank = u53(rnd[0], rnd[1]);
            }
            u = RSEM_SHFL(u_bank, gbase + (int)phase);
        }
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
    // (the steady loop issues unconditionally, the tail is peeled off: estep_block.hpp says why)
    uint32_t s = s_begin;
    for (; s + 2 < s_end; s += 2) {
        mB = mask_of(s + 1);
        issue(s + 1, mB, B);
        sample(A, mA, s);
        mA = mask_of(s + 2);
        issue(s + 2, mA, A);
        sample(B, mB, s + 1);
    }
    if (s + 1 < s_end) {
        mB = mask_of(s + 1);
        issue(s + 1, mB, B);
        sample(A, mA, s);
        sample(B, mB, s + 1);
    } else {
        sample(A, mA, s);
    }
    spill();
}

