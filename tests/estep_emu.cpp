// estep_emu.cpp -- TEST INFRASTRUCTURE: runs rsem_amd/csrc/estep_block.hpp (the per-wave body of the LANE E-step kernel) on
// the CPU.  One OS thread per lane, 256 per workgroup; the cross-lane intrinsics are exchanges through memory with a
// barrier on either side; LDS is an array per workgroup; atomics are CAS loops.  The layout (sort, shapes, planes, masks,
// units) is rebuilt here on the host with sell_layout.hpp's own index helpers (row_to_slot, shape_G / shape_R, the policy
// table, the Q32 rule); what is being tested is the kernel body: lane -> (read, position) mapping, prefetch rings, Q32
// arithmetic, reductions over a read's lanes, spills, the compile-time variants.  Never part of the product.
//
//   estep_emu in.bin out.bin        in:  i32 M, N1, T, policy (1: sort key without the apart bit), q32 (0/1), range_bits, from_counts (0/1), pad; f64 N0
//                                        u64 row_ptr[N1+1]; i32 sid[nnz]; f64 cp[nnz]; f64 ncp[N1]; f64 theta[M+1]
//                                   out: f64 counts[M+1] (without N0), f64 noise total, f64 reads with a non-zero normaliser
// Build (tests/test_estep_emu_cpu.py): hipcc -DRSEM_EMU [-DRSEM_F64_DEPTHS=... -DRSEM_Q32_DEPTHS=...] tests/estep_emu.cpp -lpthread
#include "simt_emu.hpp"

namespace {
using rsem::kEpsilon;
constexpr int kTotSlots = 64;
constexpr int kWindow = 2048;
#include "../rsem_amd/csrc/estep_block.hpp"
}  // namespace

// ---- one workgroup = four blocks of one shape (the wrapper of k_estep_lane, plain theta or theta from counts) -----------
struct Job {
    const HostLayout* H;
    Shape S;
    uint32_t slice_begin, n_slices;
    int base, span, M;
    bool far;  // the rule of sell_flag_far_units (sell_layout.hpp): some id of the unit lies outside its window
    bool far_queue;  // ... and such units take the far-queue instantiation (environment ESTEP_EMU_FAR_QUEUE=0: the loop with global atomics)
    bool from_counts;
    const double* theta;  // plain: theta[M+1]; from_counts: counts[M+1] followed by 2 * kTotSlots totals
    double N0;
    double* counts;
    double* tot_noise;
    double* tot_neff;
    double th_win[kWindow], cnt_win[kWindow];
    XArgs xa;  // split rows (policy 2)
    // the far queues of the four waves (k_estep_lane<.., kFQ = true>: the launch over the units with ids outside their window)
    int fq_sid[4][kFarQCap];
    double fq_val[4][kFarQCap];
    int fq_n[4];
    emu::Block blk;
};

template <bool kFC>
static void lane_body(Job* J, int tid) {
    emu::t_tid = tid;
    emu::t_blk = &J->blk;
    const int lane = tid & 63, w = tid >> 6;
    const HostLayout& H = *J->H;
    const Shape& S = J->S;
    const uint32_t per_wave = H.T;
    const uint32_t u_end = S.slice_base + J->slice_begin + J->n_slices;
    const uint32_t s_begin = S.slice_base + J->slice_begin + (uint32_t)w * per_wave;
    const uint32_t s_end = std::min(u_end, s_begin + per_wave);
    const double* tsrc = J->theta + J->M + 1;
    double noise = 0.0, neff = 0.0;
#define EMU_BLOCK(KK, QQ, FF, XX)                                                                                                      \
    estep_block<KK, kFC, QQ, (QQ ? kQ32Depth[KK - 1] : kF64Depth[KK - 1]), FF, XX>(S, s_begin, s_end, lane, J->base, J->span, J->theta, tsrc, J->N0, \
        J->th_win, J->cnt_win, H.sval.data(), H.sexp.data(), H.ssid.data(), H.sncp.data(), H.masks.data(), J->counts, noise, neff, J->M, J->xa)
    FarQueue fq;
    fq.sid = J->fq_sid[w]; fq.val = J->fq_val[w]; fq.n = &J->fq_n[w];
    if (lane == 0) J->fq_n[w] = 0;
#define EMU_BLOCK_FQ(KK, QQ)                                                                                                          \
    estep_block<KK, kFC, QQ, 3, true, false, true>(S, s_begin, s_end, lane, J->base, J->span, J->theta, tsrc, J->N0, \
        J->th_win, J->cnt_win, H.sval.data(), H.sexp.data(), H.ssid.data(), H.sncp.data(), H.masks.data(), J->counts, noise, neff, J->M, J->xa, fq)
    // the dispatch of k_estep_lane (em.hip); the units with ids outside their window (not those of split rows) go to the launch with the
    // far queue, as launch_estep deals them
    const int code = (S.K - 1) | ((S.fmt == kFmtQ32 ? 1 : 0) << 2) | ((J->far ? 1 : 0) << 3) | (((!kFC && S.fmt == kFmtF64X) ? 1 : 0) << 4);
    if (s_begin < u_end) switch (code) {
        case 0: EMU_BLOCK(1, false, false, false); break;
        case 1: EMU_BLOCK(2, false, false, false); break;
        case 2: EMU_BLOCK(3, false, false, false); break;
        case 3: EMU_BLOCK(4, false, false, false); break;
        case 4: EMU_BLOCK(1, true, false, false); break;
        case 5: EMU_BLOCK(2, true, false, false); break;
        case 6: EMU_BLOCK(3, true, false, false); break;
        case 7: EMU_BLOCK(4, true, false, false); break;
        case 8: if (J->far_queue) EMU_BLOCK_FQ(1, false); else EMU_BLOCK(1, false, true, false); break;
        case 9: if (J->far_queue) EMU_BLOCK_FQ(2, false); else EMU_BLOCK(2, false, true, false); break;
        case 10: if (J->far_queue) EMU_BLOCK_FQ(3, false); else EMU_BLOCK(3, false, true, false); break;
        case 11: if (J->far_queue) EMU_BLOCK_FQ(4, false); else EMU_BLOCK(4, false, true, false); break;
        case 12: if (J->far_queue) EMU_BLOCK_FQ(1, true); else EMU_BLOCK(1, true, true, false); break;
        case 13: if (J->far_queue) EMU_BLOCK_FQ(2, true); else EMU_BLOCK(2, true, true, false); break;
        case 14: if (J->far_queue) EMU_BLOCK_FQ(3, true); else EMU_BLOCK(3, true, true, false); break;
        case 15: if (J->far_queue) EMU_BLOCK_FQ(4, true); else EMU_BLOCK(4, true, true, false); break;
        default:
            if constexpr (!kFC) switch (code & 11) {
                case 0: EMU_BLOCK(1, false, false, true); break;
                case 1: EMU_BLOCK(2, false, false, true); break;
                case 2: EMU_BLOCK(3, false, false, true); break;
                case 3: EMU_BLOCK(4, false, false, true); break;
                case 8: EMU_BLOCK(1, false, true, true); break;
                case 9: EMU_BLOCK(2, false, true, true); break;
                case 10: EMU_BLOCK(3, false, true, true); break;
                default: EMU_BLOCK(4, false, true, true); break;
            }
            break;
    } else {
        const ThetaSrc th = theta_src<kFC>(J->theta, tsrc, J->N0, lane);
        stage_windows<kFC>(J->base, J->span, J->M, th, J->th_win, J->cnt_win);
    }
#undef EMU_BLOCK
#undef EMU_BLOCK_FQ
    RSEM_SYNC();
    for (int i = tid; i < J->span; i += 256)
        if (J->cnt_win[i] != 0.0) emu::atomic_add(&J->counts[J->base + i], J->cnt_win[i]);
    emu::atomic_add(J->tot_noise, noise);
    emu::atomic_add(J->tot_neff, neff);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    int32_t hdr[8];
    double N0;
    if (fread(hdr, 4, 8, f) != 8 || fread(&N0, 8, 1, f) != 1) return 3;
    const int M = hdr[0];
    const uint64_t N1 = (uint64_t)hdr[1];
    std::vector<uint64_t> rp(N1 + 1);
    if (fread(rp.data(), 8, N1 + 1, f) != N1 + 1) return 3;
    const uint64_t nnz = rp[N1];
    std::vector<int32_t> sid(nnz);
    std::vector<double> cp(nnz), ncp(N1);
    const bool from_counts = hdr[6] != 0;
    std::vector<double> theta((size_t)M + 1 + (from_counts ? 2 * kTotSlots : 0));
    if (fread(sid.data(), 4, nnz, f) != nnz || fread(cp.data(), 8, nnz, f) != nnz || fread(ncp.data(), 8, N1, f) != N1 ||
        fread(theta.data(), 8, theta.size(), f) != theta.size()) return 3;
    fclose(f);
    HostLayout H;
    H.T = (uint32_t)hdr[2];
    build_layout(H, M, N1, rp.data(), sid.data(), cp.data(), ncp.data(), hdr[3], hdr[4] != 0, hdr[5], hdr[3] == 1 ? 0 : (hdr[7] > 0 ? hdr[7] : kLayoutWindow));  // hdr[3] = 1: the key without its apart bit; a smaller window is also the bit's reach
    std::vector<double> counts((size_t)M + 1, 0.0);
    double tot_noise = 0.0, tot_neff = 0.0;
    // split rows (policy 2, plain theta only): k_far_rowsum of em.hip before the blocks ...
    std::vector<double> xextra(H.n_slots - H.x_slot_base + 1, 0.0), xinv(H.n_slots - H.x_slot_base + 1, 0.0);
    if (!H.far.empty() && from_counts) { fprintf(stderr, "estep_emu: split rows need a plain theta\n"); return 5; }
    for (const HostLayout::Far& e : H.far) {
        double fv = theta[e.sid] * e.cp;
        if (fv < kEpsilon) fv = 0.0;
        xextra[e.slot - H.x_slot_base] += fv;
    }
    Job* J = new Job();
    J->far_queue = !(getenv("ESTEP_EMU_FAR_QUEUE") && atoi(getenv("ESTEP_EMU_FAR_QUEUE")) == 0);
    J->xa.extra = xextra.data();
    J->xa.inv = xinv.data();
    J->xa.slot_base = H.x_slot_base;
    pthread_barrier_init(&J->blk.bar, nullptr, 256);
    for (int w = 0; w < 4; w++) pthread_barrier_init(&J->blk.w[w].bar, nullptr, 64);
    for (const Shape& S : H.shapes) {
        for (uint32_t b0 = 0; b0 < S.n_slices; b0 += 4 * H.T) {  // a unit = 4 blocks (sell_build_units' full-size unit)
            J->H = &H;
            J->S = S;
            J->slice_begin = b0;
            J->n_slices = std::min<uint32_t>(4 * H.T, S.n_slices - b0);
            int lo = 0x7fffffff, hi = 0;
            for (uint64_t p = (S.plane_base + (uint64_t)b0 * S.K) * 64; p < (S.plane_base + (uint64_t)(b0 + J->n_slices) * S.K) * 64; p++)
                if (H.ssid[p] > 0) { lo = std::min(lo, (int)H.ssid[p]); hi = std::max(hi, (int)H.ssid[p]); }
            if (lo > hi) { lo = 1; hi = 1; }
            J->base = lo;
            J->span = std::min(hi - lo + 1, hdr[7] > 0 ? hdr[7] : kWindow);  // hdr[7]: a smaller window, to force the out-of-window path
            J->far = false;
            {
                Unit U{};
                U.base = J->base;
                U.span = J->span;
                for (uint64_t p = (S.plane_base + (uint64_t)b0 * S.K) * 64; p < (S.plane_base + (uint64_t)(b0 + J->n_slices) * S.K) * 64; p++)
                    J->far = J->far || unit_entry_is_far(U, H.ssid[p]);
            }
            J->M = M;
            J->from_counts = from_counts;
            J->theta = theta.data();
            J->N0 = N0;
            J->counts = counts.data();
            J->tot_noise = &tot_noise;
            J->tot_neff = &tot_neff;
            std::vector<std::thread> th;
            for (int t = 0; t < 256; t++) th.emplace_back(from_counts ? lane_body<true> : lane_body<false>, J, t);
            for (auto& t : th) t.join();
        }
    }
    // ... and k_far_colsum after them: the far alignments' fractions with the reciprocal normalisers the blocks left
    for (const HostLayout::Far& e : H.far) {
        double fv = theta[e.sid] * e.cp;
        if (fv < kEpsilon) fv = 0.0;
        counts[e.sid] += fv * xinv[e.slot - H.x_slot_base];
    }
    fprintf(stderr, "estep_emu: %zu far entries of split rows\n", H.far.size());
    f = fopen(argv[2], "wb");
    if (!f) return 4;
    fwrite(counts.data(), 8, counts.size(), f);
    fwrite(&tot_noise, 8, 1, f);
    fwrite(&tot_neff, 8, 1, f);
    fclose(f);
    return 0;
}
