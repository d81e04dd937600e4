// gibbs_exact_wg.hpp -- the body of k_gibbs_exact_wg (gibbs.hip): the reference's Gibbs chain (Gibbs.cpp:265-311,
// sampling.h:50-65) with ONE WORKGROUP of 256 threads per chain.
//
// Included by gibbs.hip inside its anonymous namespace and by tests/gibbs_exact_emu.cpp, which runs this very code on the
// CPU (one OS thread per lane) against the oracle's chain.  Everything that differs between the two goes through the GX_*
// macros below, which expand to the GPU intrinsic in the product.
//
// The chain is sequential from read to read only through `counts`, and one visit changes at most two of its entries.  Reads
// are cut into TILES of up to 256 consecutive reads and kXCap items (gx_build_tiles: a table built once per context, the cut
// depends on the row pointers only).  The workgroup takes the tiles in file order, one read per thread, in phases separated
// by workgroup barriers, so that every phase's latency is paid once per 256 reads:
//   1. stage the tile (coalesced loads into LDS) and mark which item is the read's current transcript;
//   2. take the tile's MT19937 outputs (read r of the tile takes the r-th next output: the sequential order) and gather
//      counts[sid] for the tile's items -- exact: every earlier tile has been committed;
//   3. evaluate all reads at once, one read per thread;
//   4. resolve the dependencies INSIDE the tile by fixed-point rounds: a thread's draw depends on the moves (z_old -> z_new)
//      of EARLIER threads that touch one of its transcripts.  Every round the moving threads enter their two endpoints in an
//      exact-keyed LDS hash table (per id: the threads that move TO it and the threads that move FROM it, as 256-bit masks);
//      the tile's items are then looked up item-major (thread g: items g, g + 256, ...) and each gets its delta as two
//      popcounts over the bits of the threads EARLIER than its read -- no loop over predecessors, whatever the number of
//      reads of one hot gene in the tile; a thread whose read has a changed delta redraws with the SAME random number; a round in which no draw changes leaves every thread consistent with all earlier
//      threads, which by induction over the thread index is the sequential chain's state (thread 0 depends on nobody);
//   5. commit the moves (counts[z_old]--, counts[z_new]++, z[]).
// Same visiting order, same left-to-right cumulative sums (one thread sums one read), same MT19937 stream as the reference:
// the integer count vectors are the reference's, bit for bit.  Uniform pseudo count only: with --prior (per-transcript
// pseudo counts, Gibbs.cpp:171-194) the tile would need 8 more bytes of LDS per item; those runs use the one-wave kernel
// k_gibbs_exact_coop.
// History (profiles/r03b..r03d): a first design gave every wave its own 64-read tile and passed a token from tile to tile
// (staging hidden behind the other waves' turns).  Its token section was one wave executing ~2500 dependent instructions
// at ~16 cycles each -- 36-41 k cycles per 64 reads whatever the number of waves, the LDS layout (file order or
// transposed) or the memory scope of the count updates.  Here the same instructions run in four waves side by side.  The
// first block-synchronous version found an item's delta by walking the earlier moved threads that share its hashed id:
// quadratic in the reads of a hot gene per tile, 72 k of 106 k cycles per 256-read tile at configs[2] (profiles/r03e, r03f).
#pragma once
#include <type_traits>
#include <vector>

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
#define GX_BLOCK_SYNC() __syncthreads()
#define GX_BALLOT(p) __ballot(p)
#define GX_LDS_OR64(p, v) (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define GX_LDS_CAS32(p, expected, desired) atomicCAS(p, expected, desired) /* returns the old value */
#define GX_POPC64(x) __popcll(x)
#define GX_CNT_LOAD(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GX_CNT_ADD(p, v) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GX_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

constexpr int kXW = 4;           // waves of READS per tile
constexpr int kXT = 64 * kXW;    // reads per tile = the threads that own a read
// Threads of the workgroup.  256 (the product): every thread owns a read and 16 of the tile's items.  -DRSEM_GX_THREADS=512: the
// first four waves own a read each thread, the other four own none and share the item-major phases (staging, gather, filter
// scan, look-ups) -- two waves per SIMD, so that one wave's dependent instructions (~16 cycles each with a SIMD to itself)
// overlap the other's: 0.065 us per read visit against 0.079 with ONE workgroup per chain (profiles/r04u_call.log,
// r05b_team_profile.log), but 119 against 105 ms per round for the teams of gibbs_exact_team.hpp (more waves at every barrier),
// which is what runs whenever the device has compute units to spare.  The emulator runs both (tests/test_gibbs_exact_emu_cpu.py).
#ifndef RSEM_GX_THREADS
#define RSEM_GX_THREADS 256
#endif
constexpr int kXThr = RSEM_GX_THREADS;
static_assert(kXThr == kXT || kXThr == 2 * kXT, "256 or 512 threads per chain");
constexpr int kXCap = 4096;      // items per tile: 256 reads of 12.4 items (BASELINE configs[2]) = 3175 on average
constexpr int kXKeys = 1024;     // entries of the move-endpoint table (at most 2 * kXT endpoints: load <= 0.5)
constexpr int kXChunk = 16;      // items of a read handled per step with independent (pipelined) LDS reads
constexpr int kXTail = 4;        // ... per step behind the last whole step of kXChunk
constexpr int kXPlanes = kXCap / kXThr;  // items per thread in the item-major phases

constexpr int kXBits = 8192;
struct XTile {  // the workgroup's LDS: 150 KB of the CU's 160 KB
    unsigned long long rp[kXT + 1];
    unsigned long long ends[kXKeys][2][kXW];  // per entry: the threads that move TO the id / FROM the id (all zero between rounds)
    int32_t key[kXKeys];                      // id + 1 of the entry, 0 = free (all zero between rounds)
    double p[kXCap];
    int32_t sid[kXCap];
    int32_t c[kXCap];          // counts[sid] after every earlier tile, minus 1 where the read itself sits
    unsigned char ownr[kXCap]; // the thread (read of the tile) the item belongs to
    int16_t dl[kXCap];         // what the moves of EARLIER reads of the tile add to this item's count (-255 .. 255)
    int32_t zold[kXT];         // the reads' current transcripts
    unsigned long long mm[kXW], chg[kXW], dirty[kXW];  // per wave: threads that move / whose draw changed / with a changed delta
    unsigned long long bits[kXBits / 64];     // one bit per hashed endpoint id (all zero between rounds): the scan's filter
    uint32_t mt[624];
    int idx;
    // the team of workgroups of gibbs_exact_team.hpp: which waves published a change; what the team barrier returned
    unsigned long long pub[kXW];
    unsigned long long team_epoch;
    int team_res;
};

// Tiles: greedy cut into runs of <= kXT consecutive reads holding <= kXCap items; a read with more items than that is a tile
// of its own (walked over global memory).  Depends on the row pointers only.  (Host; also tests/gibbs_exact_emu.cpp.)
inline void gx_build_tiles(uint64_t N1, const uint64_t* row_ptr, std::vector<uint32_t>& tiles) {
    tiles.clear();
    tiles.reserve(N1 / 200 + 2);
    uint64_t i = 0;
    while (i < N1) {
        tiles.push_back((uint32_t)i);
        const uint64_t b = row_ptr[i];
        uint64_t e = i + 1;  // the first read always belongs to the tile
        while (e < N1 && e - i < (uint64_t)kXT && row_ptr[e + 1] - b <= (uint64_t)kXCap) ++e;
        i = e;
    }
    tiles.push_back((uint32_t)N1);
}

// Phase profile (variant builds only, -DRSEM_GX_PROFILE=1; the product's kernel reads no timer): shader-clock cycles of
// thread 0's wave summed into prof[0..5] = stage | own flags | random numbers + gather | first draw | resolve rounds |
// commit; prof[7] = tiles, prof[8] = resolve rounds; inside the rounds prof[9..14] = enter endpoints | barrier | scan | item
// walk | redraw | ballot + clean-up + barriers (prof[4] then holds only the rest), prof[15] = items thread 0 walked
#ifndef RSEM_GX_PROFILE
#define RSEM_GX_PROFILE 0
#endif
// -DRSEM_GX_FENCES=1 (a variant build, not measured yet): a scheduling fence at every phase boundary and nothing else.  The
// build that reads the phase clocks runs a tile in 186 ms-per-round terms where the product's runs 206 (profiles/r04s_call.log,
// r04v_call.log): the clock reads keep the compiler from mixing the phases' instructions, and this asks for the same without them.
#ifndef RSEM_GX_FENCES
#define RSEM_GX_FENCES 0
#endif
// -DRSEM_GX_STATIC_WALK=1 (a variant build, not measured yet): see the look-ups of a round.
#ifndef RSEM_GX_STATIC_WALK
#define RSEM_GX_STATIC_WALK 0
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

// the in-place MT19937 twist by ONE wave, 64 words per pass in increasing order (see mt_regen_wave of gibbs.hip for why the
// plain pass order reproduces the sequential loop)
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

GX_DEVFN unsigned gx_hash(int s) { return ((unsigned)s * 2654435761u) >> 22; }  // 10 bits: kXKeys entries
GX_DEVFN unsigned gx_bit(int s) { return ((unsigned)s * 2246822519u) >> 19; }   // 13 bits: kXBits
static_assert(kXBits == 8192, "gx_bit returns 13 bits");
static_assert(kXKeys == 1024, "gx_hash returns 10 bits");

// One sweep over all reads in file order (Gibbs.cpp:297-311), or the initial assignment (Gibbs.cpp:281-291) when kInit.
// Called by every thread of the chain's workgroup (g = 0 .. kXThr-1; the threads g >= kXT own no read); L->mt / L->idx hold the chain's generator.
template <bool kInit>
GX_DEVFN void gibbs_exact_wg_body(int g, XTile* L, uint32_t n_tiles, const uint32_t* __restrict__ tile_start, const uint64_t* __restrict__ tile_items,
                                  const uint64_t* __restrict__ row_ptr, const int32_t* __restrict__ sid, const double* __restrict__ cp,
                                  int32_t* counts, int32_t* z, double pseudoC, unsigned long long* prof) {
    const int lane = g & 63, w = g >> 6;
    const bool rd = kXThr == kXT || g < kXT;  // a thread that owns one of the tile's read slots (all of them in the 256-thread build)
    unsigned long long pa[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // What the staging of a tile needs is loaded one tile AHEAD, into registers (the kernel may use 512 of them): issued when
    // the tile before it starts its first draw, after that tile's gather of counts -- loads return in order, so anything
    // issued earlier would sit in front of the gather --, used a resolve round later.  Staging is then LDS writes only (it
    // was three global round trips: row pointers, two batches of items; 6.5 k of a tile's 51 k cycles, profiles/r03o).
    // tile_items[t] = row_ptr[tile_start[t]] (host table, so that no load of the look-ahead depends on another one); the
    // four numbers a fetch needs are themselves read one step earlier (`la`).
    struct Ahead {
        uint32_t r0;
        int nr;
        uint64_t base, T64;
        unsigned long long rp;
        int z;
        int s[kXPlanes];
        double p[kXPlanes];
    } A;
    struct { uint32_t r0, r1; uint64_t b0, b1; } la;
    auto look = [&](uint32_t tn) {  // the numbers of tile tn (or of the empty tile behind the last one)
        const uint32_t a = tn < n_tiles ? tn : n_tiles, b = tn < n_tiles ? tn + 1 : n_tiles;
        la.r0 = tile_start[a];
        la.r1 = tile_start[b];
        la.b0 = tile_items[a];
        la.b1 = tile_items[b];
    };
    auto fetch = [&]() {  // the tile `la` describes -> A
        A.r0 = la.r0;
        A.nr = (int)(la.r1 - la.r0);
        A.base = la.b0;
        A.T64 = la.b1 - la.b0;
        const uint32_t Tn = A.T64 > (uint64_t)kXCap ? 0u : (uint32_t)A.T64;
        A.rp = A.nr > 0 ? row_ptr[(uint64_t)A.r0 + (uint64_t)(g < A.nr ? g : A.nr)] : la.b0;
        A.z = (!kInit && g < A.nr) ? z[(uint64_t)A.r0 + g] : 0;
#pragma unroll
        for (int u = 0; u < kXPlanes; u++) {
            const uint32_t j = (uint32_t)u * kXThr + g;
            A.s[u] = j < Tn ? sid[A.base + j] : 0;
            A.p[u] = j < Tn ? cp[A.base + j] : 0.0;
        }
    };
    look(0);
    fetch();
    look(1);
    for (uint32_t t = 0; t < n_tiles; t++) {
        unsigned long long tk = GX_CLOCK();
        auto lap = [&](int i) {
            if (RSEM_GX_PROFILE) {
                const unsigned long long n = GX_CLOCK();
                pa[i] += n - tk;
                tk = n;
            }
#if RSEM_GX_FENCES && !defined(GX_EMU)
            __builtin_amdgcn_sched_barrier(0);
#endif
        };
        const uint64_t r0 = A.r0;
        const int nr = A.nr;  // 1 .. kXT
        // ---- stage (from the registers loaded ahead) -------------------------------------------------------------------------
        const uint64_t base = A.base;
        const uint64_t T64 = A.T64;
        const bool long_tile = T64 > (uint64_t)kXCap;  // one read with more items than a tile holds (then nr == 1)
        const uint32_t T = long_tile ? 0u : (uint32_t)T64;
        if (rd) L->rp[g] = A.rp;
        if (g == 0) L->rp[kXT] = base + T64;
        const bool mine = g < nr;
        const int z_old = A.z;
        int sj[kXPlanes];  // the tile's ids item-major (item u * kXThr + g): kept in registers for the gather and the rounds
#pragma unroll
        for (int u = 0; u < kXPlanes; u++) {
            const uint32_t j = (uint32_t)u * kXThr + g;
            sj[u] = A.s[u];
            if (j < T) {
                L->sid[j] = A.s[u];
                L->p[j] = A.p[u];
                if (!kInit) L->dl[j] = 0;
            }
        }
        GX_WAIT_VM();  // the previous tile's count updates (this thread's) are performed: after the barrier, everybody's
        GX_BLOCK_SYNC();
        const uint32_t fr = mine ? (uint32_t)(L->rp[g] - base) : 0;
        const int len = (mine && !long_tile) ? (int)(L->rp[g + 1] - L->rp[g]) : 0;
        int idx = L->idx;
        lap(0);
        // ---- whose item is it (the gather and the rounds walk the items item-major: other threads' items) ---------------------------
        if (!kInit) {
            for (int k = 0; k < len; k++) L->ownr[fr + k] = (unsigned char)g;
            if (rd) L->zold[g] = z_old;
            if (rd && lane == 0) L->dirty[w] = 0ull;
            GX_BLOCK_SYNC();
        }
        lap(1);
        uint32_t* mt = L->mt;
        if (long_tile) {
            // thread 0 walks the read over global memory, two passes (as k_gibbs_exact_coop does)
            if (idx >= 624) {  // (uniform)
                if (w == 0) gx_mt_regen(mt, lane);
                idx = 0;
                GX_BLOCK_SYNC();
            }
            if (g == 0) {
                const uint32_t rnd = gx_temper(mt[idx]);
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
            idx += 1;
            fetch();  // (the next tile's staging data, as below)
            look(t + 2);
        } else {
            // the next nr MT19937 outputs; read r of the tile takes the r-th
            uint32_t rnd = 0;
            {
                if (idx >= 624) {  // (uniform)
                    if (w == 0) gx_mt_regen(mt, lane);
                    idx = 0;
                    GX_BLOCK_SYNC();
                }
                const int avail = 624 - idx;
                if (g < avail && mine) rnd = gx_temper(mt[idx + g]);
                if (nr > avail) {
                    GX_BLOCK_SYNC();
                    if (w == 0) gx_mt_regen(mt, lane);
                    GX_BLOCK_SYNC();
                    if (g >= avail && mine) rnd = gx_temper(mt[g - avail]);
                    idx = nr - avail;
                } else {
                    idx += nr;
                }
            }
            if (!kInit) {
                // counts of the tile's items as they are after every earlier tile (item-major: neighbouring threads fetch
                // neighbouring ids), the read's own unit taken off where it sits (Gibbs.cpp:298: the read leaves its transcript
                // before it is weighed)
                int cj[kXPlanes], zo[kXPlanes];
                // (unconditional loads, see draw(): an item past the tile's end has id 0 in sj[] and reads item 0's owner.  Issuing the
                // count loads a phase earlier, behind the staging barrier, bought nothing: 4.2 k -> 2.9 k here, 2.1 k -> 3.5 k there,
                // profiles/r04r4_call.log.)
#pragma unroll
                for (int u = 0; u < kXPlanes; u++) cj[u] = GX_CNT_LOAD(&counts[sj[u]]);
                int ow_[kXPlanes];
#pragma unroll
                for (int u = 0; u < kXPlanes; u++) {
                    const uint32_t j = (uint32_t)u * kXThr + g;
                    ow_[u] = (int)L->ownr[j < T ? j : 0u];
                }
#pragma unroll
                for (int u = 0; u < kXPlanes; u++) zo[u] = L->zold[ow_[u]];
#pragma unroll
                for (int u = 0; u < kXPlanes; u++) {
                    const uint32_t j = (uint32_t)u * kXThr + g;
                    if (j < T) L->c[j] = cj[u] - (sj[u] == zo[u] ? 1 : 0);  // (j >= T: nothing stored, whatever was read)
                }
            }
            GX_BLOCK_SYNC();
            lap(2);
            fetch();      // the next tile's staging data (tile t + 1, described by `la`), in flight from here on
            look(t + 2);  // ... and the numbers of the one after it
            // sample() of sampling.h:50-65 on arr[k] = arr[k-1] + weight_k: the index of the first partial sum > prb, which for
            // a non-decreasing array is the number of partial sums <= prb (what the binary search there finds), capped at
            // len-1.  0.0 + a == a and x + 0.0 == x exactly, so the padded positions leave the left-to-right sums bit-identical.
            // kDelta: the items' counts carry the deltas of earlier reads' moves (a redraw).
            auto draw = [&](auto with_delta) -> int {
                constexpr bool kDelta = decltype(with_delta)::value;
                // (No load sits under a condition: `in ? L->p[..] : 0.0` compiles to a branch around the load with its own wait,
                // sixteen LDS round trips one after the other -- 10.2 k of a tile's 50 k cycles in the first draw alone,
                // profiles/r04r_call.log.  Positions past the read's end read its last item again and are masked afterwards.)
                const int last = len > 0 ? len - 1 : 0;
                auto load = [&](auto width, int k0, double* a) {  // the weights of items k0 .. k0 + W - 1 (0.0 past the read's end)
                    constexpr int W = decltype(width)::value;
                    int cc[W];
                    double pp[W];
#pragma unroll
                    for (int j = 0; j < W; j++) {
                        const uint32_t at = fr + (uint32_t)(k0 + j < len ? k0 + j : last);
                        pp[j] = L->p[at];
                        cc[j] = kInit ? 0 : L->c[at];
                        if (kDelta) cc[j] += (int)L->dl[at];
                    }
#pragma unroll
                    for (int j = 0; j < W; j++) {
                        const double wgt = kInit ? pp[j] : ((double)cc[j] + pseudoC) * pp[j];
                        a[j] = (k0 + j < len) ? wgt : 0.0;
                    }
                };
                // Items 0 .. 15 in one step (their partial sums are kept for the second pass); behind them steps of 16 while the
                // read has 16 more, then steps of 4: a wave runs a step if ANY of its reads needs it, and with one read of 17 items
                // among 64 a second step of 16 cost as much as the first (3 of them per draw, 10 k cycles per tile at
                // configs[2], profiles/r04r_call.log).  Every read still adds its own items strictly left to right.
                using Wide = std::integral_constant<int, kXChunk>;
                using Narrow = std::integral_constant<int, kXTail>;
                double part[kXChunk], a[kXChunk];
                double run = 0.0;
                load(Wide{}, 0, a);
#pragma unroll
                for (int j = 0; j < kXChunk; j++) {
                    run += (j < len) ? a[j] : 0.0;
                    part[j] = run;
                }
                {
                    int k0 = kXChunk;
                    for (; k0 + kXChunk <= len; k0 += kXChunk) {
                        load(Wide{}, k0, a);
#pragma unroll
                        for (int j = 0; j < kXChunk; j++) run += a[j];
                    }
                    for (; k0 < len; k0 += kXTail) {
                        load(Narrow{}, k0, a);
#pragma unroll
                        for (int j = 0; j < kXTail; j++) run += (k0 + j < len) ? a[j] : 0.0;
                    }
                }
                const double prb = ((double)rnd * (1.0 / 4294967296.0)) * run;
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < kXChunk; j++) cnt += (j < len && part[j] <= prb) ? 1 : 0;
                double r2 = part[kXChunk - 1];
                {
                    int k0 = kXChunk;
                    for (; k0 + kXChunk <= len; k0 += kXChunk) {
                        load(Wide{}, k0, a);
#pragma unroll
                        for (int j = 0; j < kXChunk; j++) {
                            r2 += a[j];
                            cnt += (r2 <= prb) ? 1 : 0;
                        }
                    }
                    for (; k0 < len; k0 += kXTail) {
                        load(Narrow{}, k0, a);
#pragma unroll
                        for (int j = 0; j < kXTail; j++) {
                            r2 += (k0 + j < len) ? a[j] : 0.0;
                            cnt += (k0 + j < len && r2 <= prb) ? 1 : 0;
                        }
                    }
                }
                const int l = cnt < len ? cnt : len - 1;
                return L->sid[fr + l];
            };
            int z_new = mine ? draw(std::false_type{}) : z_old;
            lap(3);
            if (!kInit) {
                // Rounds.  The table, the filter and the dirty words are all zero here (every round cleans up after itself).
                for (;;) {
                    const bool mv = mine && z_new != z_old;
                    const int z_ent = z_new;  // (the endpoint entered below: z_new may change in this round)
                    const unsigned long long bm = GX_BALLOT(mv);
                    if (rd && lane == 0) L->mm[w] = bm;
                    unsigned h_fr = 0, h_to = 0;  // this thread's entries
                    if (mv) {
                        // enter the two endpoints: claim a free entry or find the id's entry (linear probing; key = id + 1)
                        auto enter = [&](int id, int dir) -> unsigned {
                            unsigned h = gx_hash(id);
                            for (;;) {
                                int old = L->key[h];  // (a hot id has many movers: all but the first find it with a plain read)
                                if (old == 0) old = GX_LDS_CAS32(&L->key[h], 0, id + 1);
                                if (old == 0 || old == id + 1) break;
                                h = (h + 1) & (kXKeys - 1);
                            }
                            GX_LDS_OR64(&L->ends[h][dir][w], 1ull << lane);
                            const unsigned b = gx_bit(id);
                            GX_LDS_OR64(&L->bits[b >> 6], 1ull << (b & 63));
                            return h;
                        };
                        h_fr = enter(z_old, 1);
                        h_to = enter(z_new, 0);
                    }
                    lap(9);
                    GX_BLOCK_SYNC();
                    lap(10);
                    bool any_moved = false;
#pragma unroll
                    for (int q = 0; q < kXW; q++) any_moved = any_moved || L->mm[q] != 0ull;
                    if (!any_moved) break;  // (uniform) nobody moves: the table is untouched, nothing to resolve or commit
                    if (RSEM_GX_PROFILE) pa[8] += 1;
                    // Every item's delta: moves of EARLIER threads (earlier than the item's read) to its id minus moves from its
                    // id.  Item-major -- thread g takes items g, g + 256, ... whoever they belong to: their ids are still in its
                    // registers and every thread has the same number of them.  Only an item whose id MAY have an entry (its bit
                    // of the 8192-bit filter the movers set: 512 endpoints at most, so few false hits) or that still carries a
                    // delta from an earlier round is looked up: most items are neither.  A changed delta marks the item's read.
                    // (Until r03n every thread walked its own read's items: the slowest lane of a wave set the pace, 21.8 k
                    // cycles per tile at configs[2] with the table's first probe as the filter, 8.7 k with this one.)
                    unsigned need = 0;
                    int dv[kXPlanes];  // the items' deltas as the round found them
                    {
                        unsigned long long bw[kXPlanes];
#pragma unroll
                        for (int u = 0; u < kXPlanes; u++) {
                            const uint32_t j = (uint32_t)u * kXThr + g;
                            const unsigned b = gx_bit(sj[u]);
                            bw[u] = L->bits[b >> 6];           // (unconditional loads, see draw(); masked below)
                            dv[u] = (int)L->dl[j < T ? j : 0u];
                        }
#pragma unroll
                        for (int u = 0; u < kXPlanes; u++) {
                            const uint32_t j = (uint32_t)u * kXThr + g;
                            if (j >= T) { bw[u] = 0ull; dv[u] = 0; }
                        }
#pragma unroll
                        for (int u = 0; u < kXPlanes; u++) {
                            const unsigned b = gx_bit(sj[u]);
                            if (((bw[u] >> (b & 63)) & 1ull) != 0ull || dv[u] != 0) need |= 1u << u;
                        }
                    }
                    lap(11);
#if RSEM_GX_STATIC_WALK
                    // Variant (not measured yet): the look-ups plane by plane instead of item by item -- every plane's first probe,
                    // owner and masks are loaded whether the thread needs them or not, so no load waits for a decision and the
                    // ids stay in registers; only an item whose first probe hits ANOTHER id's entry (rare: the table is at most
                    // half full) is left to the loop below.
                    {
                        unsigned again = 0;
#pragma unroll
                        for (int u = 0; u < kXPlanes; u++) {
                            const uint32_t j = (uint32_t)u * kXThr + g, jc = j < T ? j : 0u;
                            const unsigned h = gx_hash(sj[u]);
                            const int kv = L->key[h];
                            const int o = (int)L->ownr[jc], ow = o >> 6;
                            unsigned long long to[kXW], from[kXW];
#pragma unroll
                            for (int q = 0; q < kXW; q++) { to[q] = L->ends[h][0][q]; from[q] = L->ends[h][1][q]; }
                            const unsigned long long part = (1ull << (o & 63)) - 1ull;
                            int dd = 0;
#pragma unroll
                            for (int q = 0; q < kXW; q++) {
                                const unsigned long long bef = q < ow ? ~0ull : (q == ow ? part : 0ull);
                                dd += GX_POPC64(to[q] & bef) - GX_POPC64(from[q] & bef);
                            }
                            const bool wanted = ((need >> u) & 1u) != 0u;
                            const bool other = kv != 0 && kv != sj[u] + 1;  // another id's entry: probe on, below
                            if (kv == 0) dd = 0;
                            if (wanted && other) again |= 1u << u;
                            if (wanted && !other && dd != dv[u]) {
                                L->dl[j] = (int16_t)dd;
                                GX_LDS_OR64(&L->dirty[ow], 1ull << (o & 63));
                            }
                        }
                        need = again;
                    }
#endif
                    // (Two items per step -- two independent chains of LDS round trips sharing their waits -- was slower: 15.6 k
                    // cycles per tile instead of 11.7 k, profiles/r04r4_call.log.)
                    for (; need != 0u; need &= need - 1u) {
                        const int u = __builtin_ctz(need);
                        const uint32_t j = (uint32_t)u * kXThr + g;
                        const int sv = L->sid[j];  // (= sj[u]; a register array cannot be indexed by a variable)
                        const int o = (int)L->ownr[j], ow = o >> 6;
                        unsigned h = gx_hash(sv);
                        int kv = L->key[h];
                        while (kv != 0 && kv != sv + 1) {
                            h = (h + 1) & (kXKeys - 1);
                            kv = L->key[h];
                        }
                        int dd = 0;
                        if (kv != 0) {
                            const unsigned long long part = (1ull << (o & 63)) - 1ull;  // the owner's wave: the lanes before it
#pragma unroll
                            for (int q = 0; q < kXW; q++) {
                                const unsigned long long bef = q < ow ? ~0ull : (q == ow ? part : 0ull);
                                dd += GX_POPC64(L->ends[h][0][q] & bef) - GX_POPC64(L->ends[h][1][q] & bef);
                            }
                        }
                        if (dd != (int)L->dl[j]) {
                            L->dl[j] = (int16_t)dd;
                            GX_LDS_OR64(&L->dirty[ow], 1ull << (o & 63));
                        }
                        if (RSEM_GX_PROFILE) pa[15] += 1;
                    }
                    lap(12);
                    GX_BLOCK_SYNC();  // every delta of this round is in place
                    // (Measured and dropped: the redraw of a marked read by its whole wave -- lane k weighs item k, every lane adds
                    // the weights up in order, one ballot counts the partial sums -- costs ~1 k cycles per read, and a round marks
                    // about twenty reads per wave, most of them in the tile's last wave: 21 k + 61 k cycles per tile instead of
                    // 4 k + 6 k, profiles/r04r3_call.log.)
                    const bool dirty = rd && ((L->dirty[w] >> lane) & 1ull) != 0ull;
                    int z2 = z_new;
                    if (dirty) z2 = draw(std::true_type{});
                    lap(13);
                    const unsigned long long ch = GX_BALLOT(mine && z2 != z_new);
                    z_new = z2;
                    if (rd && lane == 0) L->chg[w] = ch;
                    GX_BLOCK_SYNC();  // every thread has read its dirty bit (and the look-ups were finished a barrier ago)
                    if (mv) {          // leave the table as it was found: all zero
#pragma unroll
                        for (int q = 0; q < kXW; q++) {
                            L->ends[h_fr][0][q] = 0ull; L->ends[h_fr][1][q] = 0ull;
                            L->ends[h_to][0][q] = 0ull; L->ends[h_to][1][q] = 0ull;
                        }
                        L->key[h_fr] = 0;
                        L->key[h_to] = 0;
                        L->bits[gx_bit(z_old) >> 6] = 0ull;
                        L->bits[gx_bit(z_ent) >> 6] = 0ull;
                    }
                    if (rd && lane == 0) L->dirty[w] = 0ull;
                    bool any_changed = false;
#pragma unroll
                    for (int q = 0; q < kXW; q++) any_changed = any_changed || L->chg[q] != 0ull;
                    GX_BLOCK_SYNC();  // the table is clean again (and chg / mm may be rewritten)
                    lap(14);
                    if (!any_changed) break;  // (uniform) every thread is consistent with all earlier threads
                }
                lap(4);
                if (mine && z_new != z_old) {
                    GX_CNT_ADD(&counts[z_old], -1);
                    GX_CNT_ADD(&counts[z_new], 1);
                    z[r0 + g] = z_new;
                }
            } else if (mine) {
                GX_CNT_ADD(&counts[z_new], 1);
                z[r0 + g] = z_new;
            }
        }
        // (this tile's count updates must be performed before the next tile gathers: every thread waits for its own at the next
        // tile's first barrier, behind that tile's staging -- LDS writes from registers --, not here)
        if (g == 0) L->idx = idx;
        GX_BLOCK_SYNC();
        lap(5);
        if (RSEM_GX_PROFILE) pa[7] += 1;
    }
#if RSEM_GX_PROFILE && !defined(GX_EMU)
    if (prof && g == 0)
        for (int i = 0; i < 16; i++) (void)__hip_atomic_fetch_add(&prof[i], pa[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    (void)prof;
    (void)pa;
#endif
}
