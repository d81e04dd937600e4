// gibbs_block.hpp -- the per-wave body of the Gibbs PARALLEL sweep kernel (k_sample_z_lane of gibbs.hip).
//
// Included by gibbs.hip INSIDE its anonymous namespace (after kEpsilon / Philox / sell_layout.hpp are in scope) and by
// tests/gibbs_emu.cpp, which runs this very code on the CPU (tests/simt_emu.hpp).  Intrinsics go through simt_macros.hpp.
#pragma once
#include "simt_macros.hpp"

// Measured in round 3 and adopted (profiles/r03a_variants_and_steps.log: 1.538 -> 1.396 ms per sweep at C3): scalar slice
// addressing with a per-slice table of sorted positions (no divisions by T and R in every slice), the read's uniform from
// Philox2x32-10 (64 bits per call) instead of Philox4x32-10.  Candidates for the next measurement:
//   RSEM_GIBBS_NT          the value planes (read once per sweep) with the non-temporal hint, as in the E step
//   RSEM_GIBBS_RNG_SPREAD  a read that occupies G lanes computes ONE uniform per slice in its first lane while the other
//                          G - 1 lanes wait: instead lane j of the read computes the uniform of slice s + j once every G
//                          slices and each slice fetches its own with one cross-lane move (same keys, same numbers)
#ifndef RSEM_GIBBS_NT
#define RSEM_GIBBS_NT 0
#endif
#ifndef RSEM_GIBBS_RNG_SPREAD
#define RSEM_GIBBS_RNG_SPREAD 0
#endif
//   RSEM_GIBBS_DPP         the scan over a read's lanes and the two broadcasts with DPP moves (row_shr / quad_perm) instead of
//                          ds_bpermute wherever a read occupies <= 16 (scan) / <= 4 (broadcasts) lanes: the same values move,
//                          the picks are bit-identical
#ifndef RSEM_GIBBS_DPP
#define RSEM_GIBBS_DPP 0
#endif
template <int kCtrl>
RSEM_DEVFN double gdpp(double v) {  // the value of the lane the DPP control selects (0 where it selects none)
    const long long b = RSEM_DOUBLE_AS_LL(v);
    const int lo = RSEM_DPP_MOV((int)(unsigned)b, kCtrl);
    const int hi = RSEM_DPP_MOV((int)(unsigned)(b >> 32), kCtrl);
    return RSEM_LL_AS_DOUBLE(((long long)hi << 32) | (unsigned)lo);
}

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

// per slice: the sorted position of the read in row slot 0 and the slot stride of its block, so that the position of the
// read in slot r (the key of its random number) is x + r * y without the divisions by T and R in every slice
// (RSEM_GIBBS_SCALAR_ADDR; k_slice_ptab in gibbs.hip tabulates it)
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
template <int K>
RSEM_DEVFN void gibbs_block(const Shape& S, uint32_t T, uint32_t s_begin, uint32_t s_end, int lane, int base, int span,
                                   const double* __restrict__ g, double g0, double* g_win, int* cnt_win,
                                   const double* __restrict__ scp, const int32_t* __restrict__ ssid,
                                   const double* __restrict__ sncp, const unsigned long long* __restrict__ masks,
                                   const PtabEntry* __restrict__ ptab,
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
        if (m != 0ull) {
            const int32_t* __restrict__ ip = ssid + p0;
#pragma unroll
            for (int k = 0; k < K; k++) b.id[k] = ip[k * 64 + ulane];
        }
        const double* __restrict__ vp = scp + p0;
#pragma unroll
        for (int k = 0; k < K; k++) b.c[k] = RSEM_GIBBS_NT ? RSEM_NT_LOAD(&vp[k * 64 + ulane]) : vp[k * 64 + ulane];
        b.nc = g0lane ? (sncp + (S.slot_base + sl * R))[ulane >> lg] : 0.0;
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
                if (off < (unsigned)span) RSEM_ATOMIC_ADD_I32(&cnt_win[off], acc[k]);
                else RSEM_ATOMIC_ADD_I32(&counts[rsid[k]], acc[k]);
            }
            acc[k] = 0;
        }
    };
    double u_batch = 0.0;  // RSEM_GIBBS_RNG_SPREAD: this lane's share of the read's next G uniforms
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
        const bool dpp_scan = RSEM_GIBBS_DPP && lg >= 1 && lg <= 4;  // (uniform; row_shr moves within rows of 16 lanes: a read's lanes never straddle one)
        if (dpp_scan) {
            double o = gdpp<0x111>(incl);
            if (gl >= 1) incl += o;
            if (lg >= 2) { o = gdpp<0x112>(incl); if (gl >= 2) incl += o; }
            if (lg >= 3) { o = gdpp<0x114>(incl); if (gl >= 4) incl += o; }
            if (lg >= 4) { o = gdpp<0x118>(incl); if (gl >= 8) incl += o; }
        } else {
            for (int d = 1; d < G; d <<= 1) {
                double o = RSEM_SHFL_UP(incl, d);
                if (gl >= d) incl += o;
            }
        }
        // (one lane per read: nothing to exchange -- a uniform branch, lg comes from the unit descriptor)
        double excl = 0.0, total = incl;
        if (lg > 0) {
            excl = dpp_scan ? gdpp<0x111>(incl) : RSEM_SHFL_UP(incl, 1);
            if (gl == 0) excl = 0.0;
            if (RSEM_GIBBS_DPP && lg == 1) total = gdpp<0xF5>(incl);       // quad_perm [1,1,3,3]
            else if (RSEM_GIBBS_DPP && lg == 2) total = gdpp<0xFF>(incl);  // quad_perm [3,3,3,3]
            else total = RSEM_SHFL(incl, gbase + G - 1);
        }
        // one uniform per read, keyed by the read's position in the sorted order (layout independent)
        const uint32_t key = ph.k0 ^ ((ph.k1 << 13) | (ph.k1 >> 19)) ^ 0x5a5a5a5au;
        double u;
        if (RSEM_GIBBS_RNG_SPREAD && lg > 0) {
            const uint32_t off = (s - s_begin) & (uint32_t)(G - 1);  // (uniform over the wave)
            if (off == 0) {
                const uint32_t sj = s + (uint32_t)gl;  // lane gl of the read: the uniform of slice s + gl
                double ub = 0.0;
                if (sj < s_end) {
                    const PtabEntry pj = ptab[sj];
                    uint32_t r2[2];
                    rsem::philox2x32_10(key, pj.x + ((uint32_t)lane >> lg) * pj.y, sweep, r2);
                    ub = u53(r2[0], r2[1]);
                }
                u_batch = ub;
            }
            u = RSEM_SHFL(u_batch, gbase + (int)off);
        } else {
            const PtabEntry pt = ptab[s];  // (s is uniform over the wave: a scalar load)
            const uint32_t p = pt.x + ((uint32_t)lane >> lg) * pt.y;
            uint32_t rnd[2] = {0, 0};
            if (g0lane) rsem::philox2x32_10(key, p, sweep, rnd);
            u = u53(rnd[0], rnd[1]);
            if (RSEM_GIBBS_DPP && lg == 1) u = gdpp<0xA0>(u);       // quad_perm [0,0,2,2]
            else if (RSEM_GIBBS_DPP && lg == 2) u = gdpp<0x00>(u);  // quad_perm [0,0,0,0]
            else if (lg > 0) u = RSEM_SHFL(u, gbase);
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
    for (uint32_t s = s_begin; s < s_end; s += 2) {
        if (s + 1 < s_end) { mB = mask_of(s + 1); issue(s + 1, mB, B); }
        sample(A, mA, s);
        if (s + 1 >= s_end) break;
        if (s + 2 < s_end) { mA = mask_of(s + 2); issue(s + 2, mA, A); }
        sample(B, mB, s + 1);
    }
    spill();
}

