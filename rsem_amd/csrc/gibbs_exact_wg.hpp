// gibbs_exact_wg.hpp -- the per-wave body of k_gibbs_exact_wg (gibbs.hip): the reference's Gibbs chain (Gibbs.cpp:265-311,
// sampling.h:50-65) with ONE WORKGROUP of kXW waves per chain.
//
// Included by gibbs.hip inside its anonymous namespace and by tests/gibbs_exact_emu.cpp, which runs this very code on the
// CPU (one OS thread per lane, kXW waves) against the oracle's chain.  Everything that differs between the two goes through
// the GX_* macros below, which expand to the GPU intrinsic in the product.
//
// The chain is sequential from read to read only through `counts`.  Reads are cut into TILES of <= 64 consecutive reads
// and <= kXItems items (a table built once per context: the cut depends on the row pointers only).  Wave w of the workgroup
// owns tiles w, w + kXW, ...: it stages its tile's items into its own LDS region at any time (coalesced loads, HBM latency
// hidden behind the other waves' turns) and then waits for the TOKEN (`next_tile` in LDS).  Holding the token it
//   1. takes the tile's MT19937 outputs (read r of the tile takes the r-th next output: the sequential order),
//   2. gathers counts[sid] for the tile's items -- exact: every earlier tile has been committed with device atomics,
//   3. evaluates all reads of the tile at once, one read per lane, and resolves the dependencies INSIDE the tile by
//      fixed-point rounds: a lane's draw depends on the moves (z_old -> z_new) of EARLIER lanes that touch one of its
//      transcripts; every round each lane recomputes the deltas those moves apply to its items (a 64-bit lane mask per
//      hashed transcript id finds the candidates, zo[] / zn[] decide exactly) and redraws with the SAME random number if a
//      delta changed; a round in which no draw changes leaves every lane consistent with all earlier lanes, which by
//      induction over the lane index is the sequential chain's state (lane 0 never depends on anybody),
//   4. commits the moves (counts[z_old]--, counts[z_new]++, z[]), waits for them and passes the token on.
// Same visiting order, same left-to-right cumulative sums (one lane sums one read), same MT19937 stream as the reference:
// the integer count vectors are the reference's, bit for bit.  Uniform pseudo count only: with --prior (per-transcript
// pseudo counts, Gibbs.cpp:171-194) a tile would need 8 more bytes of LDS per item; those runs use the one-wave kernel
// k_gibbs_exact_coop.
#pragma once

#ifndef GX_EMU
#define GX_DEVFN __device__ inline
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
#define RSEM_GX_W 8
#endif
constexpr int kXW = RSEM_GX_W;  // waves per chain
constexpr int kXItems = 896;   // items per tile: 64 reads of 12.4 items (BASELINE configs[2]) = 794 on average
constexpr int kXSlots = 256;   // hashed transcript ids; slot kXSlots = the noise transcript (id 0: every read carries it)

struct XWaveLds {  // one per wave: 18.4 KB, kXW of them + XShared = 150 KB of the CU's 160 KB
    unsigned long long rp[65];
    unsigned long long mask[kXSlots + 64];
    double p[kXItems];
    int32_t sid[kXItems];
    int32_t c[kXItems];
    int32_t zo[64], zn[64];
    signed char d[kXItems];
};
struct XShared {
    uint32_t mt[624];
    int idx;
    unsigned next_tile;
};

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

GX_DEVFN int gx_slot(int s) { return s == 0 ? kXSlots : (s & (kXSlots - 1)); }

// One sweep over all reads in file order (Gibbs.cpp:297-311), or the initial assignment (Gibbs.cpp:281-291) when kInit.
// Called by every lane of every wave of the chain's workgroup; sh->mt / sh->idx / sh->next_tile (= 0) are set up before.
template <bool kInit>
GX_DEVFN void gibbs_exact_wg_body(int lane, int w, XShared* sh, XWaveLds* my, uint32_t n_tiles, const uint32_t* __restrict__ tile_start,
                                  const uint64_t* __restrict__ row_ptr, const int32_t* __restrict__ sid, const double* __restrict__ cp,
                                  int32_t* counts, int32_t* z, double pseudoC) {
    for (uint32_t t = (uint32_t)w; t < n_tiles; t += kXW) {
        const uint64_t r0 = tile_start[t];
        const int nr = (int)(tile_start[t + 1] - tile_start[t]);  // 1 .. 64
        // ---- before the token: everything that does not depend on the counts
        my->rp[lane] = row_ptr[r0 + (uint64_t)(lane < nr ? lane : nr)];
        if (lane == 0) my->rp[64] = row_ptr[r0 + (uint64_t)nr];
        const bool mine = lane < nr;
        int z_old = 0;
        if (!kInit && mine) z_old = z[r0 + lane];
        GX_WAVE_SYNC();
        const uint64_t base = my->rp[0];
        const uint64_t T64 = my->rp[nr] - base;
        const bool long_tile = T64 > (uint64_t)kXItems;  // one read with more items than a tile holds (then nr == 1)
        const uint32_t T = long_tile ? 0u : (uint32_t)T64;
        for (uint32_t j0 = 0; j0 < T; j0 += 64 * 8) {  // coalesced, eight loads in flight per lane
            int s8[8];
            double p8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t j = j0 + u * 64 + lane;
                s8[u] = j < T ? sid[base + j] : 0;
                p8[u] = j < T ? cp[base + j] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t j = j0 + u * 64 + lane;
                if (j < T) {
                    my->sid[j] = s8[u];
                    my->p[j] = p8[u];
                }
            }
        }
        const uint32_t fr = mine ? (uint32_t)(my->rp[lane] - base) : 0;
        const int len = (mine && !long_tile) ? (int)(my->rp[lane + 1] - my->rp[lane]) : 0;
        GX_WAVE_SYNC();
        // ---- the token: tiles commit in file order
        while (GX_TOKEN_LOAD(&sh->next_tile) != t) GX_SLEEP();
        int idx = sh->idx;
        GX_WAVE_SYNC();
        uint32_t* mt = sh->mt;
        if (long_tile) {
            // lane 0 walks the read over global memory, two passes (as k_gibbs_exact_coop does)
            if (idx >= 624) { gx_mt_regen(mt, lane); idx = 0; }
            const uint32_t rnd = gx_temper(mt[idx]);
            idx += 1;
            if (lane == 0) {
                const uint64_t fr64 = base, n = T64;
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
            if (!kInit) {
                // counts of the tile's items as they are after every earlier tile
                for (uint32_t j0 = 0; j0 < T; j0 += 64 * 8) {
                    int c8[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t j = j0 + u * 64 + lane;
                        c8[u] = j < T ? GX_CNT_LOAD(&counts[my->sid[j]]) : 0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t j = j0 + u * 64 + lane;
                        if (j < T) my->c[j] = c8[u];
                    }
                }
                // the read leaves its current transcript (Gibbs.cpp:298): delta -1 on that item (own items: no other lane
                // reads or writes them)
                for (int k = 0; k < len; k++) my->d[fr + k] = (signed char)(my->sid[fr + k] == z_old ? -1 : 0);
            }
            GX_WAVE_SYNC();
            auto weight = [&](uint32_t j) -> double {
                if (kInit) return my->p[j];
                return ((double)(my->c[j] + (int)my->d[j]) + pseudoC) * my->p[j];
            };
            // sample() of sampling.h:50-65 on arr[k] = arr[k-1] + weight_k: the index of the first partial sum > prb, which for
            // a non-decreasing array is the number of partial sums <= prb (what the binary search there finds), capped at
            // len-1.  0.0 + a == a and x + 0.0 == x exactly, so the padded positions leave the left-to-right sums bit-identical.
            constexpr int kChunk = 16;
            auto draw = [&]() -> int {
                double part[kChunk];
                double run = 0.0;
#pragma unroll
                for (int j = 0; j < kChunk; j++) {
                    const double a = (j < len) ? weight(fr + j) : 0.0;
                    run += a;
                    part[j] = run;
                }
                for (int k = kChunk; k < len; k++) run += weight(fr + k);
                const double prb = ((double)rnd * (1.0 / 4294967296.0)) * run;
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < kChunk; j++) cnt += (j < len && part[j] <= prb) ? 1 : 0;
                if (len > kChunk) {
                    double r2 = part[kChunk - 1];
                    for (int k = kChunk; k < len; k++) {
                        r2 += weight(fr + k);
                        cnt += (r2 <= prb) ? 1 : 0;
                    }
                }
                const int l = cnt < len ? cnt : len - 1;
                return my->sid[fr + l];
            };
            int z_new = mine ? draw() : z_old;
            if (!kInit) {
                const unsigned long long below = (1ull << lane) - 1ull;
                for (;;) {
                    const bool moved = mine && z_new != z_old;
                    if (GX_BALLOT(moved) == 0ull) break;  // nobody moves: nothing to resolve, nothing to commit
                    my->zo[lane] = z_old;
                    my->zn[lane] = z_new;  // (lanes that did not move: zn == zo, they contribute nothing below)
#pragma unroll
                    for (int u = 0; u < (kXSlots + 64) / 64; u++) my->mask[u * 64 + lane] = 0ull;
                    GX_WAVE_SYNC();
                    if (moved) {
                        GX_LDS_OR64(&my->mask[gx_slot(z_old)], 1ull << lane);
                        GX_LDS_OR64(&my->mask[gx_slot(z_new)], 1ull << lane);
                    }
                    GX_WAVE_SYNC();
                    bool dirty = false;
                    for (int k = 0; k < len; k++) {
                        const int s = my->sid[fr + k];
                        unsigned long long m = my->mask[gx_slot(s)] & below;
                        int dd = (s == z_old) ? -1 : 0;
                        while (m) {
                            const int r1 = __builtin_ctzll(m);
                            m &= m - 1ull;
                            dd += (my->zn[r1] == s ? 1 : 0) - (my->zo[r1] == s ? 1 : 0);
                        }
                        if (dd != (int)my->d[fr + k]) {
                            my->d[fr + k] = (signed char)dd;
                            dirty = true;
                        }
                    }
                    int z2 = z_new;
                    if (dirty) z2 = draw();
                    const bool changed = mine && z2 != z_new;
                    z_new = z2;
                    if (GX_BALLOT(changed) == 0ull) break;  // every lane is consistent with all earlier lanes
                    GX_WAVE_SYNC();  // (zo / zn / mask are rewritten)
                }
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
    }
}
