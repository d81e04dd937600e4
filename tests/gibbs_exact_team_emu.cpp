// gibbs_exact_team_emu.cpp -- TEST INFRASTRUCTURE: runs rsem_amd/csrc/gibbs_exact_team.hpp (the body of k_gibbs_exact_team: the
// reference's Gibbs chain advanced by a TEAM of workgroups, W tiles of a window at once) on the CPU: one OS thread per lane,
// W workgroups of kXThr threads side by side, workgroup barriers and wave synchronisation points as real barriers, the team
// barrier as the very spin loop the GPU runs, LDS / global atomics as CPU atomics.  Never part of the product.
//
//   gibbs_exact_team_emu in.bin out.bin W     in:  i32 M, N1, rounds, seed, N0, has_alpha, pad, pad; f64 pseudoC
//                                                  u64 row_ptr[N1+1]; i32 sid[n]; f64 cp[n]; i32 init_counts[M+1]; [f64 alpha[M+1]]
//   Built with -DRSEM_GX_PRIOR=1 it is the --prior pass of the two headers (per-transcript pseudo counts: has_alpha must be 1).
//                                             out: i32 counts[rounds][M+1]  (after every sweep; the initial assignment is not dumped)
//   stderr: windows, team barriers and cross iterations per sweep (how the tests know the team path was taken)
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace emu {
struct Wave {
    pthread_barrier_t bar;
    unsigned long long slot[64];
};
thread_local int t_lane = 0;
thread_local Wave* t_wave = nullptr;
thread_local pthread_barrier_t* t_block = nullptr;
inline void wave_sync() { pthread_barrier_wait(&t_wave->bar); }
inline unsigned long long ballot(bool p) {
    t_wave->slot[t_lane] = p ? 1ull : 0ull;
    pthread_barrier_wait(&t_wave->bar);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) m |= t_wave->slot[i] << i;
    pthread_barrier_wait(&t_wave->bar);
    return m;
}
}  // namespace emu

#define GX_EMU 1
#define GX_DEVFN inline
#define GX_HOSTDEVFN inline
#define GX_WAVE_SYNC() emu::wave_sync()
#define GX_BLOCK_SYNC() pthread_barrier_wait(emu::t_block)
#define GX_BALLOT(p) emu::ballot(p)
#define GX_LDS_OR64(p, v) (void)__atomic_fetch_or(p, v, __ATOMIC_RELAXED)
#define GX_LDS_CAS32(p, expected, desired) __sync_val_compare_and_swap(p, expected, desired)
#define GX_LDS_PEEK32(p) __atomic_load_n(p, __ATOMIC_RELAXED)
#define GX_LDS_STORE_SAME(p, v) __atomic_store_n(p, v, __ATOMIC_RELAXED)
#define GX_POPC64(x) __builtin_popcountll(x)
#define GX_CNT_LOAD(p) __atomic_load_n(p, __ATOMIC_RELAXED)
#define GX_CNT_ADD(p, v) (void)__atomic_fetch_add(p, v, __ATOMIC_RELAXED)
#define GX_WAIT_VM() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define GX_G_LOAD32(p) __atomic_load_n(p, __ATOMIC_RELAXED)
#define GX_G_LOAD64(p) __atomic_load_n(p, __ATOMIC_RELAXED)
#define GX_G_STORE32(p, v) __atomic_store_n(p, v, __ATOMIC_RELAXED)
#define GX_G_STORE64(p, v) __atomic_store_n(p, v, __ATOMIC_RELAXED)
#define GX_G_ADD32(p, v) (void)__atomic_fetch_add(p, v, __ATOMIC_RELAXED)
#define GX_G_ADD64(p, v) (void)__atomic_fetch_add(p, v, __ATOMIC_RELAXED)
#define GX_TEAM_RELEASE() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define GX_TEAM_ACQUIRE() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define GX_SPIN_PAUSE() sched_yield()
#define GX_WALL() 0ull
#define __restrict__

#include "../rsem_amd/csrc/gibbs_exact_team.hpp"

struct Machine {
    XTile tile;
    emu::Wave wave[kXThr / 64];
    pthread_barrier_t block_bar;
};

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: gibbs_exact_team_emu in.bin out.bin W\n"); return 2; }
    const int W = atoi(argv[3]);
    if (W < 1 || W > kXTeamMax) { fprintf(stderr, "W out of range\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t hdr[8];
    double pseudoC;
    if (fread(hdr, 4, 8, f) != 8 || fread(&pseudoC, 8, 1, f) != 1) return 2;
    const int M = hdr[0], rounds = hdr[2];
    const uint64_t N1 = (uint64_t)hdr[1];
    const uint32_t seed = (uint32_t)hdr[3];
    const int N0 = hdr[4];
    std::vector<uint64_t> rp(N1 + 1);
    if (fread(rp.data(), 8, N1 + 1, f) != N1 + 1) return 2;
    const uint64_t n = rp[N1];
    std::vector<int32_t> sid(n), init(M + 1);
    std::vector<double> cp(n);
    if (fread(sid.data(), 4, n, f) != n || fread(cp.data(), 8, n, f) != n || fread(init.data(), 4, M + 1, f) != (size_t)M + 1) return 2;
    std::vector<double> alpha;
    if (hdr[5]) {
        alpha.resize(M + 1);
        if (fread(alpha.data(), 8, M + 1, f) != (size_t)M + 1) return 2;
    }
    if (kXPrior != (hdr[5] != 0)) { fprintf(stderr, "this build is the %s pass\n", kXPrior ? "--prior" : "uniform pseudo count"); return 2; }
    fclose(f);

    std::vector<uint32_t> tiles;
    gx_build_tiles(N1, rp.data(), tiles);  // the product's own rules
    std::vector<uint64_t> tile_items(tiles.size());
    for (size_t i = 0; i < tiles.size(); i++) tile_items[i] = rp[tiles[i]];
    std::vector<XSlot> slots;
    gx_build_windows(W, tiles, tile_items, slots);
    const uint32_t n_win = (uint32_t)(slots.size() / (size_t)W);
    std::vector<int32_t> counts(init), z(N1 ? N1 : 1, 0);
    counts[0] += N0;
    std::vector<int32_t> out((size_t)rounds * (M + 1));

    // the team's tables: all bias / all zero between windows (checked after every sweep)
    const uint32_t nw = (uint32_t)(((W + 15) / 16) * 16 / 2);
    std::vector<uint32_t> net((size_t)2 * (M + 2) * nw, kXBias | (kXBias << 16)), gnet((size_t)2 * (M + 2) * 2, kXBias | (kXBias << 16));
    std::vector<int32_t> ref((size_t)2 * (M + 2), 0);  // (two copies of each: a window uses the one of its parity)
    XTeamCtl ctl;
    memset(&ctl, 0, sizeof(ctl));

    std::vector<Machine>* mcs = new std::vector<Machine>(W);
    uint32_t mt0[624];
    mt0[0] = seed;  // boost::random::mt19937 seeding (host_mt_seed of gibbs.hip)
    for (int i = 1; i < 624; i++) mt0[i] = 1812433253u * (mt0[i - 1] ^ (mt0[i - 1] >> 30)) + (uint32_t)i;
    int idx0 = 624;
    for (int k = 0; k < W; k++) {
        Machine& mc = (*mcs)[k];
        memset(&mc.tile, 0, sizeof(mc.tile));
        for (int w = 0; w < kXThr / 64; w++) pthread_barrier_init(&mc.wave[w].bar, nullptr, 64);
        pthread_barrier_init(&mc.block_bar, nullptr, kXThr);
    }
    pthread_barrier_t all;
    pthread_barrier_init(&all, nullptr, (unsigned)(W * kXThr));
    std::atomic<int> failed{0};
    unsigned long long bars_before = 0;

    auto thread_main = [&](int k, int tid) {
        Machine& mc = (*mcs)[k];
        emu::t_lane = tid & 63;
        emu::t_wave = &mc.wave[tid >> 6];
        emu::t_block = &mc.block_bar;
        XTeam tm;
        tm.W = W; tm.tw = k; tm.ctl = &ctl; tm.net = net.data(); tm.gnet = gnet.data(); tm.ref = ref.data(); tm.nw = nw;
        tm.slots = slots.data(); tm.n_win = n_win; tm.N1 = N1; tm.M = M; tm.spin_limit = kXSpinLimit;
        for (int round = 0; round <= rounds; round++) {
            // what the kernel wrapper does: every workgroup loads the chain's generator as it is before the sweep
            if (tid == 0) { memcpy(mc.tile.mt, mt0, sizeof(mt0)); mc.tile.idx = idx0; }
            pthread_barrier_wait(&all);
            bool ok;
            if (round == 0)
                ok = gibbs_exact_team_body<true>(tid, &mc.tile, tm, rp.data(), sid.data(), cp.data(), counts.data(), z.data(), pseudoC, kXPrior ? alpha.data() : nullptr, nullptr);
            else
                ok = gibbs_exact_team_body<false>(tid, &mc.tile, tm, rp.data(), sid.data(), cp.data(), counts.data(), z.data(), pseudoC, kXPrior ? alpha.data() : nullptr, nullptr);
            if (!ok) failed = 1;
            pthread_barrier_wait(&all);
            if (k == 0 && tid == 0) {
                memcpy(mt0, mc.tile.mt, sizeof(mt0));  // workgroup 0 hands the generator on
                idx0 = mc.tile.idx;
                if (round >= 1) memcpy(&out[(size_t)(round - 1) * (M + 1)], counts.data(), sizeof(int32_t) * (M + 1));
                for (size_t i = 0; i < net.size(); i++) if (net[i] != (kXBias | (kXBias << 16))) failed = 2;
                for (size_t i = 0; i < gnet.size(); i++) if (gnet[i] != (kXBias | (kXBias << 16))) failed = 2;
                for (size_t i = 0; i < ref.size(); i++) if (ref[i] != 0) failed = 2;
                if (round >= 1)
                    fprintf(stderr, "sweep %d: %u windows, %llu team barriers\n", round, n_win, ctl.epoch - bars_before);
                bars_before = ctl.epoch;
            }
            pthread_barrier_wait(&all);
        }
    };
    std::vector<std::thread> th;
    for (int k = 0; k < W; k++)
        for (int t = 0; t < kXThr; t++) th.emplace_back(thread_main, k, t);
    for (auto& t : th) t.join();
    if (failed) { fprintf(stderr, "gibbs_exact_team_emu: failure %d (1: the team gave up, 2: tables not clean after a sweep)\n", failed.load()); return 3; }
    FILE* g = fopen(argv[2], "wb");
    if (!g) { perror(argv[2]); return 2; }
    fwrite(out.data(), 4, out.size(), g);
    fclose(g);
    return 0;
}
