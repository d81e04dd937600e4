// gibbs_exact_wg.hpp -- the per-wave body of k_gibbs_exact_wg (gibbs.hip): the reference's Gibbs chain (Gibbs.cpp:265-311,
// sampling.h:50-65) with ONE WORKGROUP of kXW waves per chain.
//
// Included by gibbs.hip inside its anonymous namespace and by tests/gibbs_exact_emu.cpp, which runs this very code on the
// CPU (one OS thread per lane, kXW waves) against the oracle's chain.  Everything that differs between the two goes through
// the GX_* macros below, which expand to the GPU intrinsic in the product.
//
// The chain is sequential from read to read only through `counts`.  Reads are cut into TILES of consecutive reads, one read
// per lane (gx_build_tiles: a table built once per context, the cut depends on the row pointers only).  A tile lives in LDS
// TRANSPOSED: item k of the read in lane r sits at [k * S + r] (S = 64, 32, 16 or 8 lanes by the tile's longest read), so a
// lane walking its own read touches consecutive banks -- no bank conflicts, and no lane ever touches another lane's items
// (the first version kept the items in file order: 36 k cycles per tile inside the token, LDS-bound; profiles/r03c).
// Wave w of the workgroup owns tiles w, w + kXW, ...: at any time it stages its tile (every lane loads its own read; HBM
// latency hidden behind the other waves' turns) and prepares what does not depend on the counts -- per hashed id the lanes
// whose read carries it (`hold`), per lane the EARLIER lanes that share an id with it (`pred`) -- and then waits for the
// TOKEN (`next_tile` in LDS).  Holding the token it
//   1. takes the tile's MT19937 outputs (read r of the tile takes the r-th next output: the sequential order),
//   2. gathers counts[sid] for its read's items -- exact: every earlier tile has been committed with device atomics,
//   3. evaluates all reads of the tile at once, one read per lane, and resolves the dependencies INSIDE the tile by
//      fixed-point rounds: a lane's draw depends on the moves (z_old -> z_new) of EARLIER lanes that touch one of its
//      transcripts; every round the lanes with a moved predecessor (one AND of the moved-lanes ballot with `pred`) recompute
//      the deltas those moves apply to their items (`hold` finds the candidates, zo[] / zn[] decide exactly) and redraw with
//      the SAME random number if a delta changed; a round in which no draw changes leaves every lane consistent with all
//      earlier lanes, which by induction over the lane index is the sequential chain's state (lane 0 depends on nobody),
//   4. commits the moves (counts[z_old]--, counts[z_new]++, z[]), waits for them and passes the token on.
// Same visiting order, same left-to-right cumulative sums (one lane sums one read), same MT19937 stream as the reference:
// the integer count vectors are the reference's, bit for bit.  Uniform pseudo count only: with --prior (per-transcript
// pseudo counts, Gibbs.cpp:171-194) a tile would need 8 more bytes of LDS per item; those runs use the one-wave kernel
// k_gibbs_exact_coop.
#pragma once
#include <type_traits>
#include <vector>

#ifndef GX_EMU
#define GX_DEVFN __device__ inline
#define GX_HOSTDEVFN __host__ __device__ inline
// LDS operations of one wave execute in order, so lanes of a wave that exchange data through LDS only need the COMPILER
// to keep the order: wavefront-scope fences, no instruction
#define GX_WAVE_SYNC()                                        \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#define GX_BALLOT(p) __ballot(p)
#define GX_LDS_OR64(p, v) (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
// candidate for a measurement: a chain's counts are touched by ONE workgroup per launch, so workgroup scope is enough
#ifndef RSEM_GX_SCOPE
#define RSEM_GX_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
#define GX_CNT_LOAD(p) __hip_atomic_load(p, __ATOMIC_RELAXED, RSEM_GX_SCOPE)
#define GX_CNT_ADD(p, v) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, RSEM_GX_SCOPE)
#define GX_TOKEN_LOAD(p) __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)
#define GX_TOKEN_STORE(p, v) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP)
#define GX_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define GX_SLEEP() __builtin_amdgcn_s_sleep(1)
#endif

#ifndef RSEM_GX_W
#define RSEM_GX_W 4
#endif
constexpr int kXW = RSEM_GX_W;  // waves per chain (the token section sets the pace: 4, 6 and 8 waves measured the same)
constexpr int kXCap = 2048;     // LDS entries per wave: S lanes x up to kXCap / S items per read
constexpr int kXMaxLen = kXCap / 8;  // 256: a longer read is a tile of its own, walked over global memory
constexpr int kXSlots = 256;    // hashed transcript ids (the noise transcript, id 0, which every read carries, is kept apart)
constexpr int kXChunk = 16;     // items of a read handled per step with independent (pipelined) LDS reads

struct XWaveLds {  // one per wave: 37.4 KB, kXW of them + XShared = 152 KB of the CU's 163.8 KB
    unsigned long long rp[65];
    unsigned long long hold[kXSlots];  // per hashed id (not the noise id 0): the lanes whose read carries such an item
    double p[kXCap];
    int32_t sid[kXCap];
    int32_t c[kXCap];          // counts[sid] after every earlier tile, minus 1 where the read itself sits
    int32_t zo[64], zn[64];
    signed char dl[kXCap];     // what the moves of EARLIER reads of the tile add to this item's count
};
struct XShared {
    uint32_t mt[624];
    int idx;
    unsigned next_tile;
};

// lanes per tile for a longest read of m items (m <= kXMaxLen)
GX_HOSTDEVFN int gx_lanes_for(int m) { return m <= kXCap / 64 ? 64 : m <= kXCap / 32 ? 32 : m <= kXCap / 16 ? 16 : 8; }

// Tiles: greedy cut into runs of consecutive reads such that the run has at most gx_lanes_for(its longest read) reads; a read
// with more than kXMaxLen items is a tile of its own.  Depends on the row pointers only.  (Host; also tests/gibbs_exact_emu.cpp.)
inline void gx_build_tiles(uint64_t N1, const uint64_t* row_ptr, std::vector<uint32_t>& tiles) {
    tiles.clear();
    tiles.reserve(N1 / 48 + 2);
    uint64_t i = 0;
    while (i < N1) {
        tiles.push_back((uint32_t)i);
        uint64_t m = row_ptr[i + 1] - row_ptr[i];
        uint64_t e = i + 1;
        if (m <= (uint64_t)kXMaxLen) {
            while (e < N1) {
                const uint64_t l = row_ptr[e + 1] - row_ptr[e];
                const uint64_t m2 = l > m ? l : m;
                if (m2 > (uint64_t)kXMaxLen || e + 1 - i > (uint64_t)gx_lanes_for((int)m2)) break;
                m = m2;
                ++e;
            }
        }
        i = e;
    }
    tiles.push_back((uint32_t)N1);
}

// Phase profile (variant builds only, -DRSEM_GX_PROFILE=1; the product's kernel reads no timer): shader-clock cycles per
// wave summed into prof[0..6] = stage + prepare | wait for the token | random numbers | gather | first draw | resolve
// rounds | commit + wait + pass the token; prof[7] = tiles, prof[8] = resolve rounds
#ifndef RSEM_GX_PROFILE
#define RSEM_GX_PROFILE 0
#endif
#if RSEM_GX_PROFILE && !defined(GX_EMU)
#define GX_CLOCK() ((unsigned long long)clock64())
#else
#define GX_CLOCK() 0ull
#endif

GX_DEVFN uint32_t gx_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// the in-place MT19937 twist by one wave, 64 words per pass in increasing order (see mt_regen_wave of gibbs.hip for why the
// plain pass order reproduces the sequential loop); the token holder is the only wave that touches mt[]
GX_DEVFN void gx_mt_regen(uint32_t* mt, int lane) {
    for (int k0 = 0; k0 < 624; k0 += 64) {
        const int k = k0 + lane;
        uint32_t v = 0;
        if (k < 624) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            v = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        GX_WAVE_SYNC();
        if (k < 624) mt[k] = v;
        GX_WAVE_SYNC();
    }
}

GX_DEVFN int gx_slot(int s) { return s & (kXSlots - 1); }

// One sweep over all reads in file order (Gibbs.cpp:297-311), or the initial assignment (Gibbs.cpp:281-291) when kInit.
// Called by every lane of every wave of the chain's workgroup; sh->mt / sh->idx / sh->next_tile (= 0) are set up before.
template <bool kInit>
GX_DEVFN void gibbs_exact_wg_body(int lane, int w, XShared* sh, XWaveLds* my, uint32_t n_tiles, const uint32_t* __restrict__ tile_start,
                                  const uint64_t* __restrict__ row_ptr, const int32_t* __restrict__ sid, const double* __restrict__ cp,
                                  int32_t* counts, int32_t* z, double pseudoC, unsigned long long* prof) {
    unsigned long long pa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t t = (uint32_t)w; t < n_tiles; t += kXW) {
        unsigned long long tk = GX_CLOCK();
        auto lap = [&](int i) {
            if (RSEM_GX_PROFILE) {
                const unsigned long long n = GX_CLOCK();
                pa[i] += n - tk;
                tk = n;
            }
        };
        const uint64_t r0 = tile_start[t];
        const int nr = (int)(tile_start[t + 1] - tile_start[t]);  // 1 .. 64
        // ---- before the token: everything that does not depend on the counts ------------------------------------------------
        my->rp[lane] = row_ptr[r0 + (uint64_t)(lane < nr ? lane : nr)];
        if (lane == 0) my->rp[64] = row_ptr[r0 + (uint64_t)nr];
        const bool mine = lane < nr;
        int z_old = 0;
        if (!kInit && mine) z_old = z[r0 + lane];
        if (!kInit) {
#pragma unroll
            for (int u = 0; u < kXSlots / 64; u++) my->hold[u * 64 + lane] = 0ull;
        }
        GX_WAVE_SYNC();
        const uint64_t base = my->rp[lane];  // this lane's read
        const uint64_t len64 = mine ? my->rp[lane + 1] - base : 0ull;
        const bool long_tile = GX_BALLOT(len64 > (uint64_t)kXMaxLen) != 0ull;  // (uniform; then nr == 1)
        const int len = long_tile ? 0 : (int)len64;
        int maxlen = 0;  // (uniform) the tile's longest read, bit by bit from the top
#pragma unroll
        for (int b = 8; b >= 0; b--)
            if (GX_BALLOT(len >= (maxlen | (1 << b))) != 0ull) maxlen |= 1 << b;
        const int S = gx_lanes_for(maxlen);  // the tile table guarantees nr <= S
        auto at = [&](int k) -> int { return k * S + lane; };
        for (int k0 = 0; k0 < maxlen; k0 += 8) {  // every lane loads its own read: eight (sid, conprb) pairs in flight
            int s8[8];
            double p8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const bool in = k0 + u < len;
                s8[u] = in ? sid[base + (uint64_t)(k0 + u)] : 0;
                p8[u] = in ? cp[base + (uint64_t)(k0 + u)] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (k0 + u < len) {
                    my->sid[at(k0 + u)] = s8[u];
                    my->p[at(k0 + u)] = p8[u];
                    if (!kInit) my->dl[at(k0 + u)] = 0;
                }
        }
        unsigned long long pred = 0ull;  // the earlier lanes of the tile whose read shares a (hashed) id with this one
        bool has0 = false;               // the read carries the noise transcript (every read of an .ofg file does)
        if (!kInit) {
            for (int k0 = 0; k0 < len; k0 += kXChunk) {
                int s[kXChunk];
#pragma unroll
                for (int j = 0; j < kXChunk; j++) s[j] = (k0 + j < len) ? my->sid[at(k0 + j)] : -1;
#pragma unroll
                for (int j = 0; j < kXChunk; j++)
                    if (k0 + j < len) {
                        if (s[j] != 0) GX_LDS_OR64(&my->hold[gx_slot(s[j])], 1ull << lane);
                        else has0 = true;
                    }
            }
            GX_WAVE_SYNC();
            for (int k0 = 0; k0 < len; k0 += kXChunk) {
                int s[kXChunk];
#pragma unroll
                for (int j = 0; j < kXChunk; j++) s[j] = (k0 + j < len) ? my->sid[at(k0 + j)] : -1;
#pragma unroll
                for (int j = 0; j < kXChunk; j++)
                    if (k0 + j < len && s[j] != 0) pred |= my->hold[gx_slot(s[j])];
            }
            pred &= below;
        }
        lap(0);
        // ---- the token: tiles commit in file order ------------------------------------------------------------------------------
        while (GX_TOKEN_LOAD(&sh->next_tile) != t) GX_SLEEP();
        lap(1);
        int idx = sh->idx;
        GX_WAVE_SYNC();
        uint32_t* mt = sh->mt;
        if (long_tile) {
            // lane 0 walks the read over global memory, two passes (as k_gibbs_exact_coop does)
            if (idx >= 624) { gx_mt_regen(mt, lane); idx = 0; }
            const uint32_t rnd = gx_temper(mt[idx]);
            idx += 1;
            if (lane == 0) {
                const uint64_t fr64 = base, n = len64;
                if (!kInit) {
                    GX_CNT_ADD(&counts[z_old], -1);
                    GX_WAIT_VM();
                }
                auto wt = [&](uint64_t j) -> double {
                    const int s = sid[j];
                    const double p = cp[j];
                    if (kInit) return p;
                    return ((double)GX_CNT_LOAD(&counts[s]) + pseudoC) * p;
                };
                double tot = 0.0;
                for (uint64_t j = 0; j < n; j++) { const double a = wt(fr64 + j); tot = (j == 0) ? a : tot + a; }
                const double prb = ((double)rnd * (1.0 / 4294967296.0)) * tot;
                double cum = 0.0;
                uint64_t l = n - 1;
                for (uint64_t j = 0; j < n; j++) {
                    const double a = wt(fr64 + j);
                    cum = (j == 0) ? a : cum + a;
                    if (cum > prb) { l = j; break; }
                }
                const int zn = sid[fr64 + l];
                GX_CNT_ADD(&counts[zn], 1);
                z[r0] = zn;
            }
        } else {
            // the next nr MT19937 outputs
            uint32_t rnd = 0;
            {
                if (idx >= 624) { gx_mt_regen(mt, lane); idx = 0; }
                const int avail = 624 - idx;
                if (lane < avail && mine) rnd = gx_temper(mt[idx + lane]);
                if (nr > avail) {
                    GX_WAVE_SYNC();
                    gx_mt_regen(mt, lane);
                    if (lane >= avail && mine) rnd = gx_temper(mt[lane - avail]);
                    idx = nr - avail;
                } else {
                    idx += nr;
                }
            }
            lap(2);
            if (!kInit) {
                // counts of this read's items as they are after every earlier tile; the read itself leaves its current
                // transcript before it is weighed (Gibbs.cpp:298)
                for (int k0 = 0; k0 < len; k0 += 8) {
                    int s8[8], c8[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) s8[u] = (k0 + u < len) ? my->sid[at(k0 + u)] : 0;
#pragma unroll
                    for (int u = 0; u < 8; u++) c8[u] = (k0 + u < len) ? GX_CNT_LOAD(&counts[s8[u]]) : 0;
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (k0 + u < len) my->c[at(k0 + u)] = c8[u] - (s8[u] == z_old ? 1 : 0);
                }
            }
            lap(3);
            // sample() of sampling.h:50-65 on arr[k] = arr[k-1] + weight_k: the index of the first partial sum > prb, which for
            // a non-decreasing array is the number of partial sums <= prb (what the binary search there finds), capped at
            // len-1.  0.0 + a == a and x + 0.0 == x exactly, so the padded positions leave the left-to-right sums bit-identical.
            // kDelta: the items' counts carry the deltas of earlier reads' moves (a redraw).
            auto draw = [&](auto with_delta) -> int {
                constexpr bool kDelta = decltype(with_delta)::value;
                auto load = [&](int k0, double* a) {  // the weights of items k0 .. k0 + kXChunk - 1 (0.0 past the read's end)
                    int cc[kXChunk];
                    double pp[kXChunk];
#pragma unroll
                    for (int j = 0; j < kXChunk; j++) {
                        const bool in = k0 + j < len;
                        pp[j] = in ? my->p[at(k0 + j)] : 0.0;
                        cc[j] = (in && !kInit) ? my->c[at(k0 + j)] : 0;
                        if (kDelta && in) cc[j] += (int)my->dl[at(k0 + j)];
                    }
#pragma unroll
                    for (int j = 0; j < kXChunk; j++) a[j] = kInit ? pp[j] : ((double)cc[j] + pseudoC) * pp[j];
                };
                double part[kXChunk], a[kXChunk];
                double run = 0.0;
                load(0, a);
#pragma unroll
                for (int j = 0; j < kXChunk; j++) {
                    run += (j < len) ? a[j] : 0.0;
                    part[j] = run;
                }
                for (int k0 = kXChunk; k0 < len; k0 += kXChunk) {
                    load(k0, a);
#pragma unroll
                    for (int j = 0; j < kXChunk; j++) run += (k0 + j < len) ? a[j] : 0.0;
                }
                const double prb = ((double)rnd * (1.0 / 4294967296.0)) * run;
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < kXChunk; j++) cnt += (j < len && part[j] <= prb) ? 1 : 0;
                double r2 = part[kXChunk - 1];
                for (int k0 = kXChunk; k0 < len; k0 += kXChunk) {
                    load(k0, a);
#pragma unroll
                    for (int j = 0; j < kXChunk; j++) {
                        r2 += (k0 + j < len) ? a[j] : 0.0;
                        cnt += (k0 + j < len && r2 <= prb) ? 1 : 0;
                    }
                }
                const int l = cnt < len ? cnt : len - 1;
                return my->sid[at(l)];
            };
            int z_new = mine ? draw(std::false_type{}) : z_old;
            lap(4);
            if (!kInit) {
                unsigned long long mm = GX_BALLOT(mine && z_new != z_old);  // the lanes whose read moves
                bool hasd = false;                                          // some dl of this lane's items is not zero
                while (mm != 0ull) {
                    // only a lane with a moved predecessor (or with deltas left from a predecessor that moved back) has work;
                    // the noise transcript is everybody's: moves to or from it concern every later read that carries it
                    const unsigned long long nmov = GX_BALLOT(mine && z_new != z_old && (z_old == 0 || z_new == 0));
                    const bool affected = mine && (((mm & pred) != 0ull) || (has0 && (nmov & below) != 0ull) || hasd);
                    if (GX_BALLOT(affected) == 0ull) break;
                    if (RSEM_GX_PROFILE) pa[8] += 1;
                    my->zo[lane] = z_old;
                    my->zn[lane] = z_new;
                    GX_WAVE_SYNC();
                    bool dirty = false;
                    if (affected) {
                        bool nz = false;
                        for (int k0 = 0; k0 < len; k0 += kXChunk) {
                            int s[kXChunk], od[kXChunk];
                            unsigned long long cand[kXChunk];
#pragma unroll
                            for (int j = 0; j < kXChunk; j++) {
                                const bool in = k0 + j < len;
                                s[j] = in ? my->sid[at(k0 + j)] : -1;
                                od[j] = in ? (int)my->dl[at(k0 + j)] : 0;
                            }
#pragma unroll
                            for (int j = 0; j < kXChunk; j++)
                                cand[j] = (k0 + j >= len) ? 0ull : (s[j] == 0 ? (nmov & below) : (my->hold[gx_slot(s[j])] & mm & below));
#pragma unroll
                            for (int j = 0; j < kXChunk; j++) {
                                int dd = 0;
                                unsigned long long m = cand[j];
                                while (m) {  // (rare: an earlier lane that moved AND carries this hashed id)
                                    const int r1 = __builtin_ctzll(m);
                                    m &= m - 1ull;
                                    dd += (my->zn[r1] == s[j] ? 1 : 0) - (my->zo[r1] == s[j] ? 1 : 0);
                                }
                                if (dd != od[j]) {
                                    my->dl[at(k0 + j)] = (signed char)dd;
                                    dirty = true;
                                }
                                nz = nz || dd != 0;
                            }
                        }
                        hasd = nz;
                    }
                    int z2 = z_new;
                    if (dirty) z2 = draw(std::true_type{});
                    const bool changed = mine && z2 != z_new;
                    z_new = z2;
                    if (GX_BALLOT(changed) == 0ull) break;  // every lane is consistent with all earlier lanes
                    mm = GX_BALLOT(mine && z_new != z_old);
                    GX_WAVE_SYNC();  // (zo / zn are rewritten)
                }
                lap(5);
                if (mine && z_new != z_old) {
                    GX_CNT_ADD(&counts[z_old], -1);
                    GX_CNT_ADD(&counts[z_new], 1);
                    z[r0 + lane] = z_new;
                }
            } else if (mine) {
                GX_CNT_ADD(&counts[z_new], 1);
                z[r0 + lane] = z_new;
            }
        }
        GX_WAVE_SYNC();
        if (lane == 0) sh->idx = idx;
        GX_WAIT_VM();  // this tile's count updates are performed before the next tile gathers
        GX_WAVE_SYNC();
        if (lane == 0) GX_TOKEN_STORE(&sh->next_tile, t + 1);
        lap(6);
        if (RSEM_GX_PROFILE) pa[7] += 1;
    }
#if RSEM_GX_PROFILE && !defined(GX_EMU)
    if (prof && lane == 0)
        for (int i = 0; i < 9; i++) (void)__hip_atomic_fetch_add(&prof[i], pa[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    (void)prof;
    (void)pa;
#endif
}
