// gibbs_exact_wg.hpp -- the TILE machinery of the exact Gibbs chain (the reference's chain, Gibbs.cpp:265-311, sampling.h:50-65,
// bit for bit): constants, the workgroup's LDS record, the tile table, MT19937 helpers.  The sweep itself is
// gibbs_exact_team.hpp (k_gibbs_exact_team of gibbs.hip): a team of W workgroups per chain, W = 1 being the one workgroup per
// chain that rounds 3-4 shipped from this file.
//
// The chain is sequential from read to read only through `counts`, and one visit changes at most two of its entries.  Reads
// are cut into TILES of up to 256 consecutive reads and kXCap items (gx_build_tiles: a table built once per context, the cut
// depends on the row pointers only).  A workgroup takes a tile, one read per thread, in phases separated by workgroup
// barriers, so that every phase's latency is paid once per 256 reads:
//   1. stage the tile (coalesced loads into LDS) and mark which item is the read's current transcript;
//   2. take the tile's MT19937 outputs (read r of the sweep takes output r: the sequential order) and gather counts[sid] for
//      the tile's items;
//   3. evaluate all reads at once, one read per thread;
//   4. resolve the dependencies INSIDE the tile by fixed-point rounds: a thread's draw depends on the moves (z_old -> z_new)
//      of EARLIER threads that touch one of its transcripts.  Every round the moving threads enter their two endpoints in an
//      exact-keyed LDS hash table (per id: the threads that move TO it and the threads that move FROM it, as 256-bit masks);
//      the tile's items are then looked up item-major (thread g: items g, g + 256, ...) and each gets its delta as two
//      popcounts over the bits of the threads EARLIER than its read -- no loop over predecessors, whatever the number of
//      reads of one hot gene in the tile; a thread whose read has a changed delta redraws with the SAME random number; a round
//      in which no draw changes leaves every thread consistent with all earlier threads, which by induction over the thread
//      index is the sequential chain's state (thread 0 depends on nobody);
//   5. commit the moves (counts[z_old]--, counts[z_new]++, z[]).
// Same visiting order, same left-to-right cumulative sums (one thread sums one read), same MT19937 stream as the reference:
// the integer count vectors are the reference's, bit for bit.
//
// TWO PASSES.  gibbs.hip and the emulators include this file (and gibbs_exact_team.hpp) twice: as it is for the uniform pseudo
// count, and inside `namespace gx_prior` with RSEM_GX_PRIOR defined for --prior (per-transcript pseudo counts, Gibbs.cpp:171-194,
// 300-303: the weight of an item is (count + pseudo_counts[sid]) * conprb): the tile then carries 8 more bytes of LDS per item
// and holds 3072 items instead of 4096.  Everything below exists once per pass.
// History (profiles/r03b..r03f, r04r..r04v): one wave per chain with a token passed from tile to tile; a block-synchronous
// version that walked the earlier moved threads per item (quadratic in the reads of a hot gene); the move-endpoint table with
// every thread walking its own read's items; the 8192-bit filter and item-major look-ups; no LDS load under a condition.
#include <type_traits>
#include <vector>
#if (!defined(RSEM_GX_PRIOR) && !defined(GX_WG_UNIFORM_PASS)) || (defined(RSEM_GX_PRIOR) && !defined(GX_WG_PRIOR_PASS))
#ifdef RSEM_GX_PRIOR
#define GX_WG_PRIOR_PASS
constexpr bool kXPrior = true;
#else
#define GX_WG_UNIFORM_PASS
constexpr bool kXPrior = false;
#endif

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
// a look at a word other lanes may be claiming with GX_LDS_CAS32, and a store of a value every storing lane agrees on: plain LDS
// accesses on the GPU (a word is read and written whole); the emulator makes them relaxed atomics, so that its ThreadSanitizer
// build (tests/test_gibbs_exact_team_emu_cpu.py) reports exactly the accesses that are NOT meant to overlap
#define GX_LDS_PEEK32(p) (*(p))
#define GX_LDS_STORE_SAME(p, v) (*(p) = (v))
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
constexpr int kXCap = kXPrior ? 3072 : 4096;  // items per tile: 256 reads of 12.4 items (BASELINE configs[2]) = 3175 on average
constexpr int kXKeys = 1024;     // entries of the move-endpoint table (at most 2 * kXT endpoints: load <= 0.5)
constexpr int kXChunk = 16;      // items of a read handled per step with independent (pipelined) LDS reads
constexpr int kXTail = 4;        // ... per step behind the last whole step of kXChunk
constexpr int kXPlanes = kXCap / kXThr;  // items per thread in the item-major phases

struct GxMtState { uint32_t mt[624]; int idx; };  // a chain's MT19937 between launches (layout of gibbs.hip's MtState)

constexpr int kXBits = 8192;
struct XTile {  // the workgroup's LDS: 153 KB (--prior: 158 KB) of the CU's 160 KB
    unsigned long long rp[kXT + 1];
    unsigned long long ends[kXKeys][2][kXW];  // per entry: the threads that move TO the id / FROM the id (all zero between rounds)
    int32_t key[kXKeys];                      // id + 1 of the entry, 0 = free (all zero between rounds)
    double p[kXCap];
    double al[kXPrior ? kXCap : 1];  // --prior: pseudo_counts[sid] of the item
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
    int32_t tb_old[kXT], tb_pub[kXT];  // the move a read published in the previous window, to be taken back (equal: none)
    unsigned long long team_epoch;
    int team_res;
};

// Tiles: greedy cut into runs of <= kXT consecutive reads holding <= kXCap items; a read with more items than that is a tile
// of its own (walked over global memory).  Depends on the row pointers only.  (Host; also the emulators.)
// `soft_cap` (0: none): a tile is closed at that many items although LDS would hold more (a measurement knob: tiles that differ less
// in their item-major phases; 3300 / 3500 / 3700 items against 4096 at configs[2]'s shape: 105 / 102 / 103 against 101 ms per round,
// profiles/r06i_*).  The chain does not depend on the cut.
inline void gx_build_tiles(uint64_t N1, const uint64_t* row_ptr, std::vector<uint32_t>& tiles, uint64_t soft_cap = 0) {
    const uint64_t cap = soft_cap > 0 && soft_cap < (uint64_t)kXCap ? soft_cap : (uint64_t)kXCap;
    tiles.clear();
    tiles.reserve(N1 / 200 + 2);
    uint64_t i = 0;
    while (i < N1) {
        tiles.push_back((uint32_t)i);
        const uint64_t b = row_ptr[i];
        uint64_t e = i + 1;  // the first read always belongs to the tile
        while (e < N1 && e - i < (uint64_t)kXT && row_ptr[e + 1] - b <= cap) ++e;
        i = e;
    }
    tiles.push_back((uint32_t)N1);
}

// Phase profile (variant builds only, -DRSEM_GX_PROFILE=1; the product's kernel reads no timer): shader-clock cycles of thread 0's
// wave summed into prof[] -- see the laps of gibbs_exact_team.hpp and the line gibbs.hip prints from them.
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

GX_DEVFN unsigned gx_hash(int s) { return ((unsigned)s * 2654435761u) >> 22; }  // 10 bits: kXKeys entries
GX_DEVFN unsigned gx_bit(int s) { return ((unsigned)s * 2246822519u) >> 19; }   // 13 bits: kXBits
static_assert(kXBits == 8192, "gx_bit returns 13 bits");
static_assert(kXKeys == 1024, "gx_hash returns 10 bits");

#endif  // (pass guard)
