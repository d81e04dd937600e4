// gibbs_exact_team.hpp -- the body of k_gibbs_exact_team (gibbs.hip): the reference's Gibbs chain (Gibbs.cpp:265-311,
// sampling.h:50-65) advanced by a TEAM of W workgroups per chain -- the same chain, bit for bit, as the one workgroup of
// gibbs_exact_wg.hpp, whose tile machinery (staging, gather, draws, the resolve rounds inside a tile) this file re-uses.
//
// Included by gibbs.hip inside its anonymous namespace and by tests/gibbs_exact_team_emu.cpp, which runs this very code on the
// CPU (one OS thread per lane, W workgroups side by side, real barriers) against the oracle's chain.
//
// Why a team.  The chain is sequential from read to read only through `counts`, and one visit changes at most two entries.  One
// workgroup resolves that inside a 256-read tile by fixed-point rounds (gibbs_exact_wg.hpp); between tiles it is still one
// instruction stream per chain, and `-p 8` keeps 8 of 256 compute units busy.  Here W consecutive tiles form a WINDOW and are
// taken by W workgroups AT ONCE, tile w of the window by workgroup w:
//   1. every workgroup resolves its tile as if it were alone, on the counts as they are after the previous window;
//   2. it PUBLISHES its reads' moves (z_old -> z_new) in the chain's net table: per transcript id one cell per workgroup (and one
//      per group of 16 workgroups) holding moves-to minus moves-from of that workgroup's tile, and a reference count per id that
//      says whether any cell of the id is in use;
//   3. team barrier; a workgroup then takes, for every item of its tile whose id is in use, the sum of the cells of the EARLIER
//      workgroups of the window -- what the earlier tiles' moves add to that id's count, X -- and where X differs from the
//      value it used, corrects the item's count, redraws the item's read with the SAME random number and resolves the tile
//      again; if draws changed it publishes the difference;
//   4. team barrier; while any workgroup published a change, step 3 is repeated.  Tile 0 depends on nobody and is final after
//      step 1; once tiles 0 .. w-1 are final, tile w reads their final cells and is final one iteration later: at most W
//      iterations, in practice two or three, and a pass in which nobody publishes leaves every tile consistent with all
//      earlier ones -- the sequential chain's state (induction over the tile index, as inside a tile over the thread index);
//   5. commit: counts and z; team barrier; next window.  The workgroup's cells and reference counts are taken back BEHIND that
//      barrier, beside the next window's first phase: the tables exist twice and a window uses the copy of its parity, so nobody
//      reads what is being taken back, and the take-backs are performed before the workgroup arrives at the next team barrier --
//      a window before the copy is used again (until round 6 they stood in front of the barrier: 6 of a moved read's 8 commit
//      atomics, all 256 workgroups at once).  Between sweeps both copies are all-bias / all-zero.
// Values read while another workgroup is still publishing are intermediate guesses like any other: only the pass in which
// nothing is published decides, and in that pass nothing is written.  The random stream is position-addressed: read r of a sweep
// takes the r-th output of the sweep whoever visits it, every workgroup carries its own copy of the generator and twists it
// forward to the block its tile starts in.
// A read with more items than a tile holds is a window of its own (workgroup 0 walks it over global memory).
// W = 1 is one workgroup per chain: no table, no barrier, the phases of a window are the resolve rounds of its one tile.
// Two passes like gibbs_exact_wg.hpp: the uniform pseudo count, and --prior inside namespace gx_prior.
#include "gibbs_exact_wg.hpp"
#if (!defined(RSEM_GX_PRIOR) && !defined(GX_TEAM_UNIFORM_PASS)) || (defined(RSEM_GX_PRIOR) && !defined(GX_TEAM_PRIOR_PASS))
#ifdef RSEM_GX_PRIOR
#define GX_TEAM_PRIOR_PASS
#else
#define GX_TEAM_UNIFORM_PASS
#endif

#ifndef GX_EMU
#define GX_G_LOAD32(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GX_G_LOAD64(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GX_G_STORE32(p, v) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GX_G_STORE64(p, v) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GX_G_ADD32(p, v) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GX_G_ADD64(p, v) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// Everything workgroups of a team exchange goes through relaxed agent-scope atomics (performed at the device's coherence point,
// never served from a cache that another workgroup does not see); what orders them around the barrier's counter is program order
// plus a wait for this wave's outstanding memory operations.  An agent-scope release / acquire FENCE would in addition write the
// whole L2 back and invalidate it (buffer_wbl2 / buffer_inv) for data nobody shares: -DRSEM_GX_TEAM_FENCES=1 builds that variant.
#ifndef RSEM_GX_TEAM_FENCES
#define RSEM_GX_TEAM_FENCES 0
#endif
#if RSEM_GX_TEAM_FENCES
#define GX_TEAM_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define GX_TEAM_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define GX_TEAM_RELEASE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define GX_TEAM_ACQUIRE() asm volatile("" ::: "memory")
#endif
// pause between two looks at the barrier's counter, in units of 64 clocks: 256 workgroups polling without a pause slow everybody's
// memory traffic down (0 / 1 / 4 / 8 / 16 / 32: 101 / 98-100 / 93-95 / 95 / 94 / 96.5 ms per round at a fifth of configs[2], profiles/r05w_*)
#ifndef RSEM_GX_SLEEP
#define RSEM_GX_SLEEP 8
#endif
#define GX_SPIN_PAUSE() __builtin_amdgcn_s_sleep(RSEM_GX_SLEEP)
#define GX_WALL() ((unsigned long long)wall_clock64()) /* 100 MHz, constant */
#endif

constexpr int kXTeamMax = 64;          // workgroups per chain at most (one 8-byte word of group cells: 4 groups of 16)
#ifdef RSEM_GX_STEP
constexpr int kXStep = RSEM_GX_STEP;
#else
constexpr int kXStep = kXPlanes % 4 == 0 ? 4 : (kXPlanes % 3 == 0 ? 3 : 2);
#endif  // items per thread whose cells are loaded together in the cross look-ups
static_assert(kXPlanes % kXStep == 0, "whole steps");
constexpr unsigned kXBias = 0x8080u;   // a cell holds net + bias (hipMemset with 0x80 makes an all-bias table)
constexpr unsigned long long kXSpinLimit = 3000000000ull;  // wall-clock ticks (30 s) a workgroup waits for its team at most

// One tile of a window as the workgroup that takes it needs it (host table, gx_build_windows): slot [win * W + w].
struct XSlot {
    uint64_t base;   // first item of the tile
    uint64_t T64;    // items of the tile
    uint32_t r0;     // first read
    uint32_t nr;     // reads (0: this workgroup has no tile in this window)
    uint32_t flags;  // bit 0: the window is a single long read (workgroup 0 walks it; no publishing, one barrier)
    uint32_t pad;
};

// Per chain, in global memory; zeroed before the first launch of a run.
struct XTeamCtl {
    unsigned long long cnt[4];  // barrier k of a run counts in cnt[k & 3]: arrivals in the low half, "my workgroup changed a draw" in the high
    unsigned long long epoch;   // barriers completed by the launches so far (workgroup 0 adds its count when it leaves)
    unsigned int abort;         // a workgroup waited longer than kXSpinLimit: everybody leaves, the host reports it
    unsigned int pad;
};

struct XTeam {
    int W, tw;             // workgroups of the team, this one's index
    XTeamCtl* ctl;
    // each table twice: copy (window & 1) behind the other, (M + 2) rows apart
    uint32_t* net;         // [2][(M + 2)][nw] words of two cells: cell w of id s = what workgroup w's tile adds to counts[s] (+ bias)
    uint32_t* gnet;        // [2][(M + 2)][2] words: cell g = the sum over workgroups 16 g .. 16 g + 15 (+ bias)
    int32_t* ref;          // [2][(M + 2)] moves of the window that hold a reference on the id (0: every cell of the id is at bias)
    uint32_t nw;           // words per row of net: roundup(W, 16) / 2
    const XSlot* slots;
    uint32_t n_win;
    uint64_t N1;
    int32_t M;             // row M + 1 of the tables is never written (the address of loads that are not needed)
    unsigned long long spin_limit;  // wall-clock ticks a workgroup waits at a team barrier at most (kXSpinLimit; tests lower it)
};

// Windows: up to W consecutive tiles; a long tile is a window of its own.  (Host; also the emulator.)
inline void gx_build_windows(int W, const std::vector<uint32_t>& tiles, const std::vector<uint64_t>& tile_items, std::vector<XSlot>& slots) {
    slots.clear();
    const size_t n_tiles = tiles.size() - 1;
    size_t t = 0;
    while (t < n_tiles) {
        const bool lng = tile_items[t + 1] - tile_items[t] > (uint64_t)kXCap;
        size_t e = t + 1;
        if (!lng)
            while (e < n_tiles && e - t < (size_t)W && tile_items[e + 1] - tile_items[e] <= (uint64_t)kXCap) ++e;
        for (int w = 0; w < W; w++) {
            XSlot s{};
            const size_t q = t + (size_t)w;
            if (q < e) {
                s.base = tile_items[q];
                s.T64 = tile_items[q + 1] - tile_items[q];
                s.r0 = tiles[q];
                s.nr = tiles[q + 1] - tiles[q];
            }
            s.flags = lng ? 1u : 0u;
            slots.push_back(s);
        }
        t = e;
    }
}

// The MT19937 twist by the whole workgroup, in place: three steps of up to 227 independent words (0 .. 226 need old words only,
// 227 .. 453 the new words of the first step, 454 .. 623 those of the second, word 623 the new word 0), reads and writes of a
// step separated by barriers.  A workgroup of a team twists through every block of the sweep (13 per window of 32 tiles) to reach
// the one its tile starts in: with one wave doing it in ten passes (gx_mt_regen) that was 20 k cycles per window.  All threads
// call; returns behind a barrier.
GX_DEVFN void gx_mt_regen_wg(uint32_t* mt, int g) {
    auto tw = [](uint32_t a, uint32_t b) -> uint32_t {
        const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    };
    static_assert(kXThr >= 227, "a step has up to 227 words");
    uint32_t v = 0;
    if (g < 227) v = mt[g + 397] ^ tw(mt[g], mt[g + 1]);
    GX_BLOCK_SYNC();
    if (g < 227) mt[g] = v;
    GX_BLOCK_SYNC();
    if (g < 227) v = mt[g] ^ tw(mt[g + 227], mt[g + 228]);
    GX_BLOCK_SYNC();
    if (g < 227) mt[g + 227] = v;
    GX_BLOCK_SYNC();
    if (g < 170) v = mt[g + 227] ^ tw(mt[g + 454], mt[(g + 455) % 624]);
    GX_BLOCK_SYNC();
    if (g < 170) mt[g + 454] = v;
    GX_BLOCK_SYNC();
}

// All threads of the workgroup call this.  `flag`: this workgroup's contribution to the OR over the team (uniform over the
// workgroup).  Returns the OR, or -1 when the team gave up (then everybody returns from the body at once).
GX_DEVFN int gx_team_barrier(int g, XTile* L, const XTeam& tm, unsigned long long epoch, unsigned& nbar, bool flag) {
    GX_WAIT_VM();      // this thread's global writes are performed ...
    GX_BLOCK_SYNC();   // ... and so are the workgroup's
    if (g == 0) {
        XTeamCtl* c = tm.ctl;
        const unsigned slot = (unsigned)((epoch + (unsigned long long)nbar) & 3ull);
        // one atomic per workgroup: arrival and flag together, so that whoever sees W arrivals sees every flag
        GX_TEAM_RELEASE();
        GX_G_ADD64(&c->cnt[slot], 1ull + (flag ? (1ull << 32) : 0ull));
        int res = 0;
        unsigned long long t0 = 0, v = 0;
        unsigned polls = 0;
        for (;;) {
            v = GX_G_LOAD64(&c->cnt[slot]);
            if ((unsigned)(v & 0xFFFFFFFFull) >= (unsigned)tm.W) break;
            if ((++polls & 1023u) == 0u) {
                if (GX_G_LOAD32(&c->abort) != 0u) { res = -1; break; }
                const unsigned long long now = GX_WALL();
                if (t0 == 0) t0 = now;
                else if (now - t0 > tm.spin_limit) { GX_G_STORE32(&c->abort, 1u); res = -1; break; }
            }
            GX_SPIN_PAUSE();
        }
        GX_TEAM_ACQUIRE();
        if (res == 0) {
            res = (v >> 32) != 0ull ? 1 : 0;
            // the slot of barrier k + 2 (= of barrier k - 2: everybody left its loop before arriving at barrier k - 1, and nobody
            // arrives at barrier k + 2 before this workgroup has arrived at k + 1, behind this store)
            if (tm.tw == 0) GX_G_STORE64(&c->cnt[(slot + 2u) & 3u], 0ull);
        }
        L->team_res = res;
    }
    GX_BLOCK_SYNC();
    nbar += 1u;
    return L->team_res;
}

// One sweep over all reads (Gibbs.cpp:297-311), or the initial assignment (Gibbs.cpp:281-291) when kInit, by the team.  Called
// by every thread of every workgroup of the chain's team; L->mt / L->idx hold the chain's generator as it is BEFORE the sweep
// (every workgroup has the same copy) and, in workgroup 0, as it is after the sweep when the function returns true.
// Returns false when the team gave up (XTeamCtl::abort).
template <bool kInit>
GX_DEVFN bool gibbs_exact_team_body(int g, XTile* L, const XTeam& tm, const uint64_t* __restrict__ row_ptr, const int32_t* __restrict__ sid,
                                    const double* __restrict__ cp, int32_t* counts, int32_t* z, double pseudoC,
                                    const double* __restrict__ alpha /* --prior pass: pseudo_counts[M + 1]; else unused */, unsigned long long* prof) {
    const int lane = g & 63, w = g >> 6;
    const bool rd = kXThr == kXT || g < kXT;
    const int W = tm.W, tw = tm.tw;
    unsigned long long pa[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // ---- what does not change over the sweep ---------------------------------------------------------------------------------
    const unsigned long long idx0 = (unsigned long long)L->idx;  // raw position of the sweep's first output in block 0 (0 .. 624)
    unsigned long long blk_cur = 0;                              // the block L->mt holds
    GX_BLOCK_SYNC();
    unsigned long long epoch = 0;
    if (!kInit && W > 1) {
        if (g == 0) L->team_epoch = GX_G_LOAD64(&tm.ctl->epoch);
        GX_BLOCK_SYNC();
        epoch = L->team_epoch;
    }
    unsigned nbar = 0;
    // the cells of EARLIER workgroups: of my group of 16 in net (4 words of 4 cells), of the groups before mine in gnet (1 word)
    const int grp = tw >> 4;
    unsigned long long m_net[4], m_g = 0ull;
    int n_cells = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        m_net[q] = 0ull;
        for (int i = 0; i < 4; i++)
            if (grp * 16 + q * 4 + i < tw) { m_net[q] |= 0xFFFFull << (16 * i); ++n_cells; }
    }
    for (int i = 0; i < 4; i++)
        if (i < grp) { m_g |= 0xFFFFull << (16 * i); ++n_cells; }
    const int bias_total = n_cells * (int)kXBias;
    const int dummy_id = tm.M + 1;
    auto swar = [](unsigned long long v) -> int {  // the sum of the four 16-bit fields
        const unsigned long long t = (v & 0x0000FFFF0000FFFFull) + ((v >> 16) & 0x0000FFFF0000FFFFull);
        return (int)(unsigned)((t + (t >> 32)) & 0xFFFFFFFFull);
    };
    // the tables' rows of id s in the copy window `wn` uses
    auto trow = [&](uint32_t wn, int s) -> size_t { return (size_t)(wn & 1u) * ((size_t)tm.M + 2) + (size_t)s; };
    auto cell_add_at = [&](size_t row, int delta) {  // this workgroup's cell of a row, and its group's
        GX_G_ADD32(&tm.net[row * tm.nw + (size_t)(tw >> 1)], (uint32_t)delta << (16 * (tw & 1)));
        GX_G_ADD32(&tm.gnet[row * 2 + (size_t)(grp >> 1)], (uint32_t)delta << (16 * (grp & 1)));
    };
    // The move this thread's read published in the previous window, taken back (see the head of the file): that window's copy of
    // the tables is not read again before the window after this one, and the atomics are performed before this workgroup's next
    // arrival at a team barrier.  Issued behind the first draw, where nothing waits for memory until the phase's barrier.  What is to
    // be taken back waits in LDS (L->tb_old / tb_pub; equal: nothing), not in registers: the kernel has none to spare.
    bool tb_any = false;  // (uniform) the previous window published
    auto take_back = [&](uint32_t prev_win) {
        if (tb_any) {  // (uniform)
            const int o = rd ? L->tb_old[g] : 0, p = rd ? L->tb_pub[g] : 0;
            if (o != p) {
                const size_t ro = trow(prev_win, o), rp_ = trow(prev_win, p);
                cell_add_at(ro, 1);
                cell_add_at(rp_, -1);
                GX_G_ADD32(&tm.ref[ro], -1);
                GX_G_ADD32(&tm.ref[rp_], -1);
            }
            tb_any = false;
        }
    };

    struct Ahead {
        uint32_t r0;
        int nr;
        uint64_t base, T64;
        uint32_t flags;
        unsigned long long rp;
        int z;
        int s[kXPlanes];
        double p[kXPlanes];
    } A;
    XSlot la;
    auto look = [&](uint32_t win) {  // the slot of window `win` (or an empty one behind the last)
        if (win < tm.n_win) la = tm.slots[(size_t)win * (size_t)W + (size_t)tw];
        else { la.base = 0; la.T64 = 0; la.r0 = 0; la.nr = 0; la.flags = 0; la.pad = 0; }
    };
    auto fetch = [&]() {  // the tile `la` describes -> A
        A.r0 = la.r0;
        A.nr = (int)la.nr;
        A.base = la.base;
        A.T64 = la.T64;
        A.flags = la.flags;
        const uint32_t Tn = A.T64 > (uint64_t)kXCap ? 0u : (uint32_t)A.T64;
        A.rp = A.nr > 0 ? row_ptr[(uint64_t)A.r0 + (uint64_t)(g < A.nr ? g : A.nr)] : la.base;
        A.z = (!kInit && g < A.nr) ? z[(uint64_t)A.r0 + g] : 0;
#pragma unroll
        for (int u = 0; u < kXPlanes; u++) {
            const uint32_t j = (uint32_t)u * kXThr + g;
            A.s[u] = j < Tn ? sid[A.base + j] : 0;
            A.p[u] = j < Tn ? cp[A.base + j] : 0.0;
        }
    };
    uint32_t* mt = L->mt;
    auto seek = [&](unsigned long long b) {  // (uniform) twist the generator forward to block b; every thread may read mt afterwards
        for (; blk_cur < b; blk_cur++) gx_mt_regen_wg(mt, g);
    };
    look(0);
    fetch();
    look(1);
    for (uint32_t win = 0; win < tm.n_win; win++) {
        unsigned long long tk = GX_CLOCK();
        auto lap = [&](int i) {
            if (RSEM_GX_PROFILE) {
                const unsigned long long n = GX_CLOCK();
                pa[i] += n - tk;
                tk = n;
            }
        };
        const uint64_t r0 = A.r0;
        const int nr = A.nr;  // 0 .. kXT (0: no tile for this workgroup in this window)
        const bool win_long = (A.flags & 1u) != 0u;
        // ---- stage (from the registers loaded ahead) -------------------------------------------------------------------------
        const uint64_t base = A.base;
        const uint64_t T64 = A.T64;
        const bool long_tile = T64 > (uint64_t)kXCap;
        const uint32_t T = long_tile ? 0u : (uint32_t)T64;
        if (rd) L->rp[g] = A.rp;
        if (g == 0) L->rp[kXT] = base + T64;
        const bool mine = g < nr;
        const int z_old = A.z;
        int sj[kXPlanes];
        int xc[kXPlanes];  // what the earlier tiles of the window add to the item's count, as this tile currently has it
#pragma unroll
        for (int u = 0; u < kXPlanes; u++) {
            const uint32_t j = (uint32_t)u * kXThr + g;
            sj[u] = A.s[u];
            xc[u] = 0;
            if (j < T) {
                L->sid[j] = A.s[u];
                L->p[j] = A.p[u];
                if (!kInit) L->dl[j] = 0;
            }
        }
        GX_WAIT_VM();
        GX_BLOCK_SYNC();
        const uint32_t fr = mine ? (uint32_t)(L->rp[g] - base) : 0;
        const int len = (mine && !long_tile) ? (int)(L->rp[g + 1] - L->rp[g]) : 0;
        lap(0);
        if (!kInit) {
            for (int k = 0; k < len; k++) L->ownr[fr + k] = (unsigned char)g;
            if (rd) L->zold[g] = z_old;
            if (rd && lane == 0) L->dirty[w] = 0ull;
            GX_BLOCK_SYNC();
        }
        lap(1);
        // ---- the tile's random numbers: read r of the sweep takes output r ------------------------------------------------------
        uint32_t rnd = 0;
        if (nr > 0) {  // (uniform)
            const unsigned long long first = idx0 + (unsigned long long)r0;
            const unsigned long long b0 = first / 624ull;
            const int off = (int)(first - b0 * 624ull);
            seek(b0);
            const int avail = 624 - off;
            if (g < avail && mine) rnd = gx_temper(mt[off + g]);
            if (nr > avail) {
                GX_BLOCK_SYNC();
                seek(b0 + 1ull);
                if (g >= avail && mine) rnd = gx_temper(mt[g - avail]);
            }
        }
        int z_new = z_old, z_pub = z_old;
        // sample() of sampling.h:50-65 on the staged tile, exactly as in gibbs_exact_wg.hpp (see there for why it looks like this)
        auto draw = [&](auto with_delta) -> int {
            constexpr bool kDelta = decltype(with_delta)::value;
            const int last = len > 0 ? len - 1 : 0;
            auto load = [&](auto width, int k0, double* a) {
                constexpr int Wd = decltype(width)::value;
                int cc[Wd];
                double pp[Wd], aa[Wd];
#pragma unroll
                for (int j = 0; j < Wd; j++) {
                    const uint32_t at = fr + (uint32_t)(k0 + j < len ? k0 + j : last);
                    aa[j] = 0.0;
                    pp[j] = L->p[at];
                    cc[j] = kInit ? 0 : L->c[at];
                    if (kDelta) cc[j] += (int)L->dl[at];
                    if (kXPrior && !kInit) aa[j] = L->al[at];
                }
#pragma unroll
                for (int j = 0; j < Wd; j++) {
                    const double wgt = kInit ? pp[j] : ((double)cc[j] + (kXPrior ? aa[j] : pseudoC)) * pp[j];  // Gibbs.cpp:300-303
                    a[j] = (k0 + j < len) ? wgt : 0.0;
                }
            };
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
        if (long_tile) {
            // thread 0 of workgroup 0 walks the read over global memory, two passes; nobody else has a tile in this window
            if (g == 0) {
                const uint64_t fr64 = base, n = T64;
                if (!kInit) {
                    GX_CNT_ADD(&counts[z_old], -1);
                    GX_WAIT_VM();
                }
                auto wt = [&](uint64_t j) -> double {
                    const int s = sid[j];
                    const double p = cp[j];
                    if (kInit) return p;
                    return ((double)GX_CNT_LOAD(&counts[s]) + (kXPrior ? alpha[s] : pseudoC)) * p;
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
            fetch();
            look(win + 2);
        } else {
            if (!kInit && nr > 0) {
                // counts of the tile's items as they are after the previous window, the read's own unit taken off where it sits
                int cj[kXPlanes], zo[kXPlanes];
#pragma unroll
                for (int u = 0; u < kXPlanes; u++) cj[u] = GX_CNT_LOAD(&counts[sj[u]]);
                if (kXPrior) {  // (the items' pseudo counts ride along with their counts: one round trip for both)
                    double aj[kXPlanes];
#pragma unroll
                    for (int u = 0; u < kXPlanes; u++) aj[u] = alpha[sj[u]];
#pragma unroll
                    for (int u = 0; u < kXPlanes; u++) {
                        const uint32_t j = (uint32_t)u * kXThr + g;
                        if (j < T) L->al[j] = aj[u];
                    }
                }
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
                    if (j < T) L->c[j] = cj[u] - (sj[u] == zo[u] ? 1 : 0);
                }
            }
            GX_BLOCK_SYNC();
            lap(2);
            fetch();        // the tile this workgroup takes in the next window, in flight from here on
            look(win + 2);  // ... and the slot of the one after it
            if (mine) z_new = draw(std::false_type{});
            lap(3);
        }
        take_back(win - 1u);
        // ---- the phases of a window ---------------------------------------------------------------------------------------------------
        // A phase = take what the EARLIER tiles of the window published (not in the first phase), one resolve round inside the tile
        // (gibbs_exact_wg.hpp, step 4: the movers enter their endpoints, every item gets the delta of the earlier reads of the
        // tile), ONE redraw of every read whose counts changed either way, publish what changed, team barrier.  The window is done
        // after a phase in which no draw of any tile changed: every read is then consistent with the moves of all earlier reads of
        // the tile and of all earlier tiles.  (The rounds inside a tile are not run to their own fixed point first: a tile that
        // needs a second round would keep 31 others waiting at the barrier, and the window needs a second phase anyway.)
        if (!kInit && !win_long) {
            bool dl_live = false;      // (uniform) deltas of an earlier round may be in place
            bool z_moved = true;       // (uniform) some draw of the tile changed since its last round (the first draw counts)
            for (int phase = 0;; phase++) {
                // -- X of every item: the cells of the earlier workgroups (see the head of the file) --------------------------------------
                if (phase > 0 && tw > 0 && nr > 0) {  // (uniform over the workgroup)
                    int rf[kXPlanes];
#pragma unroll
                    for (int u = 0; u < kXPlanes; u++) rf[u] = GX_G_LOAD32(&tm.ref[trow(win, sj[u])]);
                    // All loads of a step are issued whether needed or not (an item that needs none reads row M + 1, which nobody
                    // writes): no load waits for a decision.  Words that hold no earlier workgroup's cell are not loaded at all.
                    auto take = [&](auto n_words, auto with_groups) {
                        constexpr int NQ = decltype(n_words)::value;
                        constexpr bool kG = decltype(with_groups)::value;
#pragma unroll
                        for (int u0 = 0; u0 < kXPlanes; u0 += kXStep) {
                            unsigned long long vn[kXStep][NQ > 0 ? NQ : 1], vg[kXStep];
#pragma unroll
                            for (int i = 0; i < kXStep; i++) {
                                const int u = u0 + i;
                                const uint32_t j = (uint32_t)u * kXThr + g;
                                const size_t row = trow(win, (j < T && rf[u] != 0) ? sj[u] : dummy_id);
                                const unsigned long long* pn = (const unsigned long long*)(tm.net + row * tm.nw + (size_t)grp * 8);
                                const unsigned long long* pg = (const unsigned long long*)(tm.gnet + row * 2);
#pragma unroll
                                for (int q = 0; q < NQ; q++) vn[i][q] = GX_G_LOAD64(pn + q);
                                vg[i] = kG ? GX_G_LOAD64(pg) : 0ull;
                            }
#pragma unroll
                            for (int i = 0; i < kXStep; i++) {
                                const int u = u0 + i;
                                const uint32_t j = (uint32_t)u * kXThr + g;
                                int x = (kG ? swar(vg[i] & m_g) : 0) - bias_total;
#pragma unroll
                                for (int q = 0; q < NQ; q++) x += swar(vn[i][q] & m_net[q]);
                                if (j < T && x != xc[u]) {
                                    L->c[j] += x - xc[u];
                                    xc[u] = x;
                                    const int o = (int)L->ownr[j];
                                    GX_LDS_OR64(&L->dirty[o >> 6], 1ull << (o & 63));
                                }
                            }
                        }
                    };
                    using Y = std::true_type;
                    using N = std::false_type;
                    const int nq = (tw - grp * 16 + 3) >> 2;  // words of my group of 16 that hold an earlier workgroup's cell (0 .. 4)
                    switch (nq * 2 + (grp > 0 ? 1 : 0)) {      // (uniform)
                        case 1: take(std::integral_constant<int, 0>{}, Y{}); break;
                        case 2: take(std::integral_constant<int, 1>{}, N{}); break;
                        case 3: take(std::integral_constant<int, 1>{}, Y{}); break;
                        case 4: take(std::integral_constant<int, 2>{}, N{}); break;
                        case 5: take(std::integral_constant<int, 2>{}, Y{}); break;
                        case 6: take(std::integral_constant<int, 3>{}, N{}); break;
                        case 7: take(std::integral_constant<int, 3>{}, Y{}); break;
                        case 8: take(std::integral_constant<int, 4>{}, N{}); break;
                        default: take(std::integral_constant<int, 4>{}, Y{}); break;
                    }
                    lap(10);
                }
                // -- one resolve round inside the tile ----------------------------------------------------------------------------------
                bool any_changed = false;
                if (nr > 0 && !z_moved) {  // (uniform) no draw of the tile changed since its last round: every delta inside the tile stands
                    GX_BLOCK_SYNC();       // the corrected counts and the marks of the look-ups above are in place
                    const bool dirty = rd && ((L->dirty[w] >> lane) & 1ull) != 0ull;
                    int z2 = z_new;
                    if (dirty) z2 = draw(std::true_type{});
                    const unsigned long long ch = GX_BALLOT(mine && z2 != z_new);
                    z_new = z2;
                    if (rd && lane == 0) L->chg[w] = ch;
                    GX_BLOCK_SYNC();
                    if (rd && lane == 0) L->dirty[w] = 0ull;
#pragma unroll
                    for (int q = 0; q < kXW; q++) any_changed = any_changed || L->chg[q] != 0ull;
                    z_moved = any_changed;
                    lap(11);
                } else if (nr > 0) {  // (uniform)
                    const bool mv = mine && z_new != z_old;
                    const int z_ent = z_new;
                    const unsigned long long bm = GX_BALLOT(mv);
                    if (rd && lane == 0) L->mm[w] = bm;
                    unsigned h_fr = 0, h_to = 0;
                    if (mv) {
                        auto enter = [&](int id, int dir) -> unsigned {
                            unsigned h = gx_hash(id);
                            for (;;) {
                                int old = GX_LDS_PEEK32(&L->key[h]);
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
                    GX_BLOCK_SYNC();  // the endpoints are entered (and the corrected counts and marks of the look-ups above are in place)
                    bool any_moved = false;
#pragma unroll
                    for (int q = 0; q < kXW; q++) any_moved = any_moved || L->mm[q] != 0ull;
                    if (any_moved || dl_live) {  // (uniform) otherwise the table is empty and every delta is zero
                        if (RSEM_GX_PROFILE) pa[8] += 1;
                        unsigned need = 0;
                        int dv[kXPlanes];
                        {
                            unsigned long long bw[kXPlanes];
#pragma unroll
                            for (int u = 0; u < kXPlanes; u++) {
                                const uint32_t j = (uint32_t)u * kXThr + g;
                                const unsigned b = gx_bit(sj[u]);
                                bw[u] = L->bits[b >> 6];
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
                        for (; need != 0u; need &= need - 1u) {
                            const int u = __builtin_ctz(need);
                            const uint32_t j = (uint32_t)u * kXThr + g;
                            const int sv = L->sid[j];
                            const int o = (int)L->ownr[j], ow = o >> 6;
                            unsigned h = gx_hash(sv);
                            int kv = L->key[h];
                            while (kv != 0 && kv != sv + 1) {
                                h = (h + 1) & (kXKeys - 1);
                                kv = L->key[h];
                            }
                            int dd = 0;
                            if (kv != 0) {
                                const unsigned long long part = (1ull << (o & 63)) - 1ull;
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
                        }
                        GX_BLOCK_SYNC();  // every delta of this round is in place
                    }
                    dl_live = any_moved;
                    lap(4);
                    const bool dirty = rd && ((L->dirty[w] >> lane) & 1ull) != 0ull;
                    int z2 = z_new;
                    if (dirty) z2 = draw(std::true_type{});
                    const unsigned long long ch = GX_BALLOT(mine && z2 != z_new);
                    z_new = z2;
                    if (rd && lane == 0) L->chg[w] = ch;
                    GX_BLOCK_SYNC();  // every thread has read its mark (and the look-ups were finished a barrier ago)
                    if (mv) {          // leave the table as it was found: all zero
#pragma unroll
                        for (int q = 0; q < kXW; q++) {
                            GX_LDS_STORE_SAME(&L->ends[h_fr][0][q], 0ull); GX_LDS_STORE_SAME(&L->ends[h_fr][1][q], 0ull);
                            GX_LDS_STORE_SAME(&L->ends[h_to][0][q], 0ull); GX_LDS_STORE_SAME(&L->ends[h_to][1][q], 0ull);
                        }
                        GX_LDS_STORE_SAME(&L->key[h_fr], 0);
                        GX_LDS_STORE_SAME(&L->key[h_to], 0);
                        GX_LDS_STORE_SAME(&L->bits[gx_bit(z_old) >> 6], 0ull);
                        GX_LDS_STORE_SAME(&L->bits[gx_bit(z_ent) >> 6], 0ull);
                    }
                    if (rd && lane == 0) L->dirty[w] = 0ull;
#pragma unroll
                    for (int q = 0; q < kXW; q++) any_changed = any_changed || L->chg[q] != 0ull;
                    z_moved = any_changed;
                    lap(11);
                }
                // -- publish what changed since the last time (z_pub: what this read has in the tables; z_old: nothing) -------------------
                bool flag = any_changed;
                if (W > 1) {
                    const bool ch = mine && z_new != z_pub;
                    if (ch) {
                        const size_t r_pub = trow(win, z_pub), r_new = trow(win, z_new), r_old = trow(win, z_old);
                        cell_add_at(r_pub, -1);
                        cell_add_at(r_new, 1);
                        // a published move holds one reference on either endpoint
                        if (z_pub == z_old) GX_G_ADD32(&tm.ref[r_old], 1);
                        else GX_G_ADD32(&tm.ref[r_pub], -1);
                        if (z_new == z_old) GX_G_ADD32(&tm.ref[r_old], -1);
                        else GX_G_ADD32(&tm.ref[r_new], 1);
                        z_pub = z_new;
                    }
                    const unsigned long long cb = GX_BALLOT(ch);
                    if (rd && lane == 0) L->pub[w] = cb;
                    GX_BLOCK_SYNC();  // (also: the table is clean again, chg / mm may be rewritten)
#pragma unroll
                    for (int q = 0; q < kXW; q++) flag = flag || L->pub[q] != 0ull;
                    lap(5);
                    const int any = gx_team_barrier(g, L, tm, epoch, nbar, flag);
                    lap(6);
                    if (any < 0) return false;
                    if (RSEM_GX_PROFILE) pa[9] += 1;
                    if (any == 0) break;  // (uniform over the team) no draw of any tile changed in this phase
                } else {
                    GX_BLOCK_SYNC();  // the table is clean again (and chg / mm may be rewritten)
                    if (!flag) break;
                }
            }
        }
        // ---- commit --------------------------------------------------------------------------------------------------------------
        if (!kInit) {
            if (!long_tile && mine && z_new != z_old) {
                GX_CNT_ADD(&counts[z_old], -1);
                GX_CNT_ADD(&counts[z_new], 1);
                z[r0 + g] = z_new;
            }
            if (W > 1) {
                // every workgroup's updates of counts are performed before anybody gathers for the next window
                if (gx_team_barrier(g, L, tm, epoch, nbar, false) < 0) return false;
                // the published move is taken back beside the next window's first phase (take_back above; every thread writes
                // and reads its own words)
                if (!win_long) {  // (uniform)
                    if (rd) { L->tb_old[g] = mine ? z_old : 0; L->tb_pub[g] = mine ? z_pub : 0; }
                    tb_any = true;
                }
            } else {
                GX_BLOCK_SYNC();  // (the next tile's staging overwrites what a draw may still be reading)
            }
        } else {
            if (mine && !long_tile) {
                GX_CNT_ADD(&counts[z_new], 1);
                z[r0 + g] = z_new;
            }
            GX_BLOCK_SYNC();  // (as above: the initial assignment has no other barrier behind its draw)
        }
        lap(13);
        if (RSEM_GX_PROFILE && nr > 0) pa[7] += 1;
    }
    take_back(tm.n_win - 1u);
    // ---- the generator after the sweep (workgroup 0 hands it on) ----------------------------------------------------------------------
    if (tw == 0) {
        if (tm.N1 > 0) {
            const unsigned long long last = idx0 + (unsigned long long)tm.N1 - 1ull;
            const unsigned long long b = last / 624ull;
            GX_BLOCK_SYNC();
            seek(b);
            if (g == 0) L->idx = (int)(last - b * 624ull) + 1;
        }
        if (!kInit && W > 1 && g == 0) GX_G_STORE64(&tm.ctl->epoch, epoch + (unsigned long long)nbar);
    }
    GX_BLOCK_SYNC();
#if RSEM_GX_PROFILE && !defined(GX_EMU)
    if (prof && g == 0)
        for (int i = 0; i < 16; i++) (void)__hip_atomic_fetch_add(&prof[i], pa[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    (void)prof;
    (void)pa;
#endif
    return true;
}

#ifndef GX_EMU
struct TeamArgs {  // what every workgroup of every team needs (one argument: cooperative launches take an array of pointers)
    int W, nchains;
    XTeamCtl* ctl;       // [nchains]
    uint32_t* net;       // [nchains][2][(M + 2) * nw]
    uint32_t* gnet;      // [nchains][2][(M + 2) * 2]
    int32_t* ref;        // [nchains][2][M + 2]
    uint32_t nw;
    const XSlot* slots;  // [n_win][W]
    uint32_t n_win;
    uint64_t N1;
    int32_t M;
    unsigned long long spin_limit;  // 0: kXSpinLimit
};

// Workgroup b belongs to chain b % nchains: workgroups are dealt to the 8 XCDs round-robin, so with 8 chains (or a divisor or
// multiple of 8) a chain's team shares one XCD and its L2.  The generator is read from mt_in and handed on in mt_out (two
// buffers: a launch without team barriers -- the initial assignment, W = 1 -- has workgroups that start after workgroup 0 has left).
template <bool kInit>
__global__ __launch_bounds__(kXThr) void k_gibbs_exact_team(TeamArgs ta, const uint64_t* __restrict__ row_ptr, const int32_t* __restrict__ sid,
                                                          const double* __restrict__ cp, int32_t* counts_base, int32_t* z_base, double pseudoC,
                                                          const double* __restrict__ alpha, const GxMtState* __restrict__ mt_in, GxMtState* mt_out,
                                                          const int32_t* __restrict__ last_round, int round, uint64_t stride_c, uint64_t stride_z,
                                                          unsigned long long* prof) {
    __shared__ XTile tile;
    // (a team spread over all XCDs instead -- chain = b / W -- is slower: 104.6 against 98.3 ms per round, profiles/r05w_*)
    const int chain = (int)(blockIdx.x % (unsigned)ta.nchains), tw = (int)(blockIdx.x / (unsigned)ta.nchains);
    if (round > last_round[chain]) return;  // (uniform over the team)
    if (ta.ctl && ta.ctl[chain].abort != 0u) return;  // an earlier launch of the run gave up: the host starts the run over (uniform)
    const GxMtState* src = mt_in + chain;
    for (int i = threadIdx.x; i < 624; i += blockDim.x) tile.mt[i] = src->mt[i];
    if (threadIdx.x == 0) tile.idx = src->idx;
    // the move-endpoint table starts out all zero (every resolve round leaves it that way)
    for (int i = threadIdx.x; i < kXKeys * 2 * kXW; i += blockDim.x) (&tile.ends[0][0][0])[i] = 0ull;
    for (int i = threadIdx.x; i < kXKeys; i += blockDim.x) tile.key[i] = 0;
    for (int i = threadIdx.x; i < kXBits / 64; i += blockDim.x) tile.bits[i] = 0ull;
    __syncthreads();
    XTeam tm;
    tm.W = ta.W;
    tm.tw = tw;
    tm.ctl = ta.ctl ? ta.ctl + chain : nullptr;
    tm.net = ta.net ? ta.net + (size_t)chain * 2 * ((size_t)ta.M + 2) * ta.nw : nullptr;
    tm.gnet = ta.gnet ? ta.gnet + (size_t)chain * 2 * ((size_t)ta.M + 2) * 2 : nullptr;
    tm.ref = ta.ref ? ta.ref + (size_t)chain * 2 * ((size_t)ta.M + 2) : nullptr;
    tm.nw = ta.nw;
    tm.slots = ta.slots;
    tm.n_win = ta.n_win;
    tm.N1 = ta.N1;
    tm.M = ta.M;
    tm.spin_limit = ta.spin_limit ? ta.spin_limit : kXSpinLimit;
    const bool ok = gibbs_exact_team_body<kInit>((int)threadIdx.x, &tile, tm, row_ptr, sid, cp, counts_base + (uint64_t)chain * stride_c,
                                                 z_base + (uint64_t)chain * stride_z, pseudoC, alpha, prof);
    __syncthreads();
    if (ok && tw == 0) {
        GxMtState* dst = mt_out + chain;
        for (int i = threadIdx.x; i < 624; i += blockDim.x) dst->mt[i] = tile.mt[i];
        if (threadIdx.x == 0) dst->idx = tile.idx;
    }
}
#endif  // !GX_EMU

#endif  // (pass guard)
