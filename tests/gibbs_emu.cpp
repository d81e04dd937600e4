// gibbs_emu.cpp -- TEST INFRASTRUCTURE: runs rsem_amd/csrc/gibbs_block.hpp (the per-wave body of the Gibbs PARALLEL sweep
// kernel, k_sample_z_lane) on the CPU, on tests/simt_emu.hpp's thread-per-lane machine.  Never part of the product.
//
//   gibbs_emu in.bin out.bin     in:  i32 M, N1, T, n_sweeps, seed, window, pad, pad
//                                     u64 row_ptr[N1+1]; i32 sid[nnz]; f64 cp[nnz]; f64 ncp[N1]; f64 g[M+1]
//                                out: i32 counts[n_sweeps][M+1]  (picks per transcript of every sweep; index 0 = noise)
#include "simt_emu.hpp"

#include "../rsem_amd/csrc/rng.hpp"

namespace {
using rsem::kEpsilon;
using rsem::Philox;
using rsem::u53;
constexpr int kGWindow = 2048;
#include "../rsem_amd/csrc/gibbs_block.hpp"
}  // namespace

struct Job {
    const HostLayout* H;
    Shape S;
    uint32_t slice_begin, n_slices, per_wave;
    int base, span, M;
    bool far;  // the rule of sell_flag_far_units (sell_layout.hpp)
    const double* g;
    Philox ph;
    uint32_t sweep;
    int32_t* counts;
    double g_win[kGWindow];
    int cnt_win[kGWindow];
    int s_noise;
    emu::Block blk;
};

static void lane_body(Job* J, int tid) {
    emu::t_tid = tid;
    emu::t_blk = &J->blk;
    const int lane = tid & 63, w = tid >> 6;
    const HostLayout& H = *J->H;
    const Shape& S = J->S;
    const uint32_t T = H.T;
    const uint32_t u_end = S.slice_base + J->slice_begin + J->n_slices;
    const uint32_t s_begin = S.slice_base + J->slice_begin + (uint32_t)w * J->per_wave;
    const uint32_t s_end = std::min(u_end, s_begin + J->per_wave);
    const double* scp = (const double*)H.sval.data();  // F64 layout: plane p at scp + p * 64
    int noise = 0;
    const double g0 = J->g[0];
#define EMU_BLOCK(KK, FF) gibbs_block<KK, FF>(S, T, s_begin, s_end, lane, J->base, J->span, J->g, g0, J->g_win, J->cnt_win, scp, H.ssid.data(), H.sncp.data(), H.masks.data(), J->ph, J->sweep, J->counts, noise, J->M)
    if (s_begin < u_end) switch (S.K + (J->far ? 4 : 0)) {
        case 1: EMU_BLOCK(1, false); break;
        case 2: EMU_BLOCK(2, false); break;
        case 3: EMU_BLOCK(3, false); break;
        case 4: EMU_BLOCK(4, false); break;
        case 5: EMU_BLOCK(1, true); break;
        case 6: EMU_BLOCK(2, true); break;
        case 7: EMU_BLOCK(3, true); break;
        default: EMU_BLOCK(4, true); break;
    } else stage_gwindows(J->base, J->span, J->M, J->g, J->g_win, J->cnt_win);
#undef EMU_BLOCK
    if (noise) __atomic_fetch_add(&J->s_noise, noise, __ATOMIC_RELAXED);
    RSEM_SYNC();
    for (int i = tid; i < J->span; i += 256)
        if (J->cnt_win[i] != 0) __atomic_fetch_add(&J->counts[J->base + i], J->cnt_win[i], __ATOMIC_RELAXED);
    if (tid == 0 && J->s_noise) __atomic_fetch_add(&J->counts[0], J->s_noise, __ATOMIC_RELAXED);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    int32_t hdr[8];
    if (fread(hdr, 4, 8, f) != 8) return 3;
    const int M = hdr[0];
    const uint64_t N1 = (uint64_t)hdr[1];
    const int n_sweeps = hdr[3];
    std::vector<uint64_t> rp(N1 + 1);
    if (fread(rp.data(), 8, N1 + 1, f) != N1 + 1) return 3;
    const uint64_t nnz = rp[N1];
    std::vector<int32_t> sid(nnz);
    std::vector<double> cp(nnz), ncp(N1), g((size_t)M + 1);
    if (fread(sid.data(), 4, nnz, f) != nnz || fread(cp.data(), 8, nnz, f) != nnz || fread(ncp.data(), 8, N1, f) != N1 ||
        fread(g.data(), 8, g.size(), f) != g.size()) return 3;
    fclose(f);
    HostLayout H;
    H.T = (uint32_t)hdr[2];
    build_layout(H, M, N1, rp.data(), sid.data(), cp.data(), ncp.data(), 0, false, 0);
    std::vector<int32_t> counts((size_t)n_sweeps * (M + 1), 0);
    Job* J = new Job();
    pthread_barrier_init(&J->blk.bar, nullptr, 256);
    for (int w = 0; w < 4; w++) pthread_barrier_init(&J->blk.w[w].bar, nullptr, 64);
    for (int sw = 0; sw < n_sweeps; sw++)
        for (const Shape& S : H.shapes)
            // a unit = 4 blocks, a wave each (sell_build_units' full-size unit); hdr[6] = 1: 2 blocks, half a block per wave
            // (its half-size unit: with an odd T the second wave starts in one block and ends in the next)
            for (uint32_t b0 = 0; b0 < S.n_slices; b0 += (hdr[6] == 1 ? 2 : 4) * H.T) {
                J->H = &H;
                J->S = S;
                J->slice_begin = b0;
                J->n_slices = std::min<uint32_t>((hdr[6] == 1 ? 2 : 4) * H.T, S.n_slices - b0);
                J->per_wave = hdr[6] == 1 ? (H.T + 1) / 2 : H.T;
                int lo = 0x7fffffff, hi = 0;
                for (uint64_t p = (S.plane_base + (uint64_t)b0 * S.K) * 64; p < (S.plane_base + (uint64_t)(b0 + J->n_slices) * S.K) * 64; p++)
                    if (H.ssid[p] > 0) { lo = std::min(lo, (int)H.ssid[p]); hi = std::max(hi, (int)H.ssid[p]); }
                if (lo > hi) { lo = 1; hi = 1; }
                J->base = lo;
                J->span = std::min(hi - lo + 1, hdr[5] > 0 ? hdr[5] : kGWindow);
                J->far = false;
                {
                    Unit U{};
                    U.base = J->base;
                    U.span = J->span;
                    for (uint64_t p = (S.plane_base + (uint64_t)b0 * S.K) * 64; p < (S.plane_base + (uint64_t)(b0 + J->n_slices) * S.K) * 64; p++)
                        J->far = J->far || unit_entry_is_far(U, H.ssid[p]);
                }
                J->M = M;
                J->g = g.data();
                J->ph = Philox{(uint32_t)hdr[4], 0x52534547u};
                J->sweep = (uint32_t)sw;
                J->counts = counts.data() + (size_t)sw * (M + 1);
                J->s_noise = 0;
                std::vector<std::thread> th;
                for (int t = 0; t < 256; t++) th.emplace_back(lane_body, J, t);
                for (auto& t : th) t.join();
            }
    f = fopen(argv[2], "wb");
    if (!f) return 4;
    fwrite(counts.data(), 4, counts.size(), f);
    fclose(f);
    return 0;
}
