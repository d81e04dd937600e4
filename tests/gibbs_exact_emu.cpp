// gibbs_exact_emu.cpp -- TEST INFRASTRUCTURE: runs rsem_amd/csrc/gibbs_exact_wg.hpp (the per-wave body of k_gibbs_exact_wg,
// the workgroup-per-chain exact Gibbs sampler) on the CPU: one OS thread per lane, kXW waves per chain, the workgroup
// barriers and the wave synchronisation points as real barriers, LDS / count atomics as CPU atomics.  Never part of the
// product.
//
//   gibbs_exact_emu in.bin out.bin     in:  i32 M, N1, rounds, seed, N0, pad, pad, pad; f64 pseudoC
//                                           u64 row_ptr[N1+1]; i32 sid[n]; f64 cp[n]; i32 init_counts[M+1]
//                                      out: i32 counts[rounds][M+1]  (after every sweep; the initial assignment is not dumped)
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
#define GX_POPC64(x) __builtin_popcountll(x)
#define GX_CNT_LOAD(p) __atomic_load_n(p, __ATOMIC_RELAXED)
#define GX_CNT_ADD(p, v) (void)__atomic_fetch_add(p, v, __ATOMIC_RELAXED)
#define GX_WAIT_VM() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __restrict__

#include "../rsem_amd/csrc/gibbs_exact_wg.hpp"

struct Machine {
    XTile tile;
    emu::Wave wave[kXThr / 64];
    pthread_barrier_t block_bar;
};

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: gibbs_exact_emu in.bin out.bin\n"); return 2; }
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
    fclose(f);

    std::vector<uint32_t> tiles;
    gx_build_tiles(N1, rp.data(), tiles);  // the product's own rule
    std::vector<uint64_t> tile_items(tiles.size());
    for (size_t i = 0; i < tiles.size(); i++) tile_items[i] = rp[tiles[i]];
    const uint32_t n_tiles = (uint32_t)tiles.size() - 1;
    std::vector<int32_t> counts(init), z(N1 ? N1 : 1, 0);
    counts[0] += N0;
    std::vector<int32_t> out((size_t)rounds * (M + 1));

    static Machine mc;
    // boost::random::mt19937 seeding (host_mt_seed of gibbs.hip)
    mc.tile.mt[0] = seed;
    for (int i = 1; i < 624; i++) mc.tile.mt[i] = 1812433253u * (mc.tile.mt[i - 1] ^ (mc.tile.mt[i - 1] >> 30)) + (uint32_t)i;
    mc.tile.idx = 624;
    memset(mc.tile.ends, 0, sizeof(mc.tile.ends));  // (the kernel wrapper clears the move-endpoint table once per launch)
    memset(mc.tile.key, 0, sizeof(mc.tile.key));
    memset(mc.tile.bits, 0, sizeof(mc.tile.bits));
    for (int w = 0; w < kXThr / 64; w++) pthread_barrier_init(&mc.wave[w].bar, nullptr, 64);
    pthread_barrier_init(&mc.block_bar, nullptr, kXThr);

    auto thread_main = [&](int tid) {
        const int lane = tid & 63, w = tid >> 6;
        emu::t_lane = lane;
        emu::t_wave = &mc.wave[w];
        emu::t_block = &mc.block_bar;
        (void)lane;
        for (int round = 0; round <= rounds; round++) {
            pthread_barrier_wait(&mc.block_bar);
            if (round == 0)
                gibbs_exact_wg_body<true>(tid, &mc.tile, n_tiles, tiles.data(), tile_items.data(), rp.data(), sid.data(), cp.data(), counts.data(), z.data(), pseudoC, nullptr);
            else
                gibbs_exact_wg_body<false>(tid, &mc.tile, n_tiles, tiles.data(), tile_items.data(), rp.data(), sid.data(), cp.data(), counts.data(), z.data(), pseudoC, nullptr);
            pthread_barrier_wait(&mc.block_bar);
            if (tid == 0 && round >= 1) memcpy(&out[(size_t)(round - 1) * (M + 1)], counts.data(), sizeof(int32_t) * (M + 1));
            pthread_barrier_wait(&mc.block_bar);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < kXThr; t++) th.emplace_back(thread_main, t);
    for (auto& t : th) t.join();
    FILE* g = fopen(argv[2], "wb");
    if (!g) { perror(argv[2]); return 2; }
    fwrite(out.data(), 4, out.size(), g);
    fclose(g);
    return 0;
}
