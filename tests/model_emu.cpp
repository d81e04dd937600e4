// model_emu.cpp -- TEST INFRASTRUCTURE: runs rsem_amd/csrc/model_block.hpp (the group-per-read body of the model rounds'
// kernel, k_model_group) on the CPU -- one OS thread per lane, 256 per workgroup, tests/simt_emu.hpp -- on seeded synthetic
// reads / transcripts / tables, and compares conprb, noise conprb and the four count tables with a plain thread-per-alignment
// restatement of getConPrb / getNoiseConPrb / the E step's weights / update (the loops of SingleQModel.h:101-221,
// PairedEndQModel.h:94-188 and the no-quality twins, EM.cpp:199-244) written here with the same scalar helpers.  What is
// being tested is the lane mapping: runs of identical windows (also across the 16-alignment chunks of a long read), the
// positional split of the products and count updates (reads longer than 128 bases take a second pass), the DPP
// reductions, reads of different lengths / alignment counts side by side in one wave.  Never part of the product.
//
//   model_emu <model_type 0..3> <seed> <estRSPD 0|1> <has_mld 0|1 (single-end only)>      exit 0 = all tables agree
// Build (tests/test_model_emu_cpu.py): hipcc -DRSEM_EMU -O1 -std=c++17 tests/model_emu.cpp -lpthread
#include <random>

#include "simt_emu.hpp"

namespace {
using rsem::kEpsilon;
#include "../rsem_amd/csrc/model_block.hpp"
}  // namespace

struct Job {
    DevData D;
    DevTables T;
    const double* theta;
    double *cp, *ncp;
    AccumPtrs A;
    PlaneOut PO;
    bool q, pe, update;
    int n_blocks, block;
    // "LDS"
    double s_prob[kQProbLds], s_nprob[kQNoiseProbLds], s_prof[kProfLds], s_noise[kNoiseLds], s_rspd[kRspdLds], s_gld[kGldLds];
    emu::Block blk;
};

template <bool kQ, bool kPE, bool kUpdate>
static void lane_body(Job* J, int tid) {
    emu::t_tid = tid;
    emu::t_blk = &J->blk;
    // the wrapper of k_model_group (model.hip), block size 256 here
    if (kQ) for (int i = tid; i < kQProbLds; i += 256) J->s_prob[i] = i < 2500 ? J->T.prof[i] : 1.0;  // (as k_model_group: the pad code's entries are 1)
    for (int i = tid; i < (kQ ? kQNoiseProbLds : 5); i += 256) J->s_nprob[i] = i < (kQ ? 500 : 5) ? J->T.noise[i] : 1.0;
    if (kUpdate) {
        for (int i = tid; i < kProfLds; i += 256) J->s_prof[i] = 0.0;
        for (int i = tid; i < kNoiseLds; i += 256) J->s_noise[i] = 0.0;
        for (int i = tid; i < kRspdLds; i += 256) J->s_rspd[i] = 0.0;
        for (int i = tid; i < kGldLds; i += 256) J->s_gld[i] = 0.0;
    }
    RSEM_SYNC();
    const int lane = tid & 63;
    // the mapping of k_model_group (model.hip): chunks of rows dealt to the workgroups round-robin, a workgroup's waves taking 4 rows
    // each per step inside a chunk (chunks of 48 rows here -- three steps of this 4-wave workgroup -- so that the data sets of the
    // tests hold several chunks per workgroup and a last, shorter one)
    model_group_rows<kQ, kPE, kUpdate>(J->D, J->T, J->theta, J->cp, J->ncp, J->A, kQ ? J->s_prob : J->T.prof, J->s_nprob, J->s_prof, J->s_noise,
                                       J->s_rspd, J->s_gld, (uint64_t)(tid >> 6) * 4, 16, lane, J->PO, 48, (uint64_t)J->block, (uint64_t)J->n_blocks);
    if (!kUpdate) return;
    RSEM_SYNC();
    const int nprof = std::min(kQ ? 2500 : kProfLds, J->T.prof_rows * 25);
    for (int i = tid; i < nprof; i += 256)
        if (J->s_prof[i] != 0.0) emu::atomic_add(&J->A.prof[i], J->s_prof[i]);
    for (int i = tid; i < (kQ ? 500 : 5); i += 256)
        if (J->s_noise[i] != 0.0) emu::atomic_add(&J->A.noise[i], J->s_noise[i]);
    if (J->A.rspd)
        for (int i = tid; i < std::min(kRspdLds, J->T.B + 2); i += 256)
            if (J->s_rspd[i] != 0.0) emu::atomic_add(&J->A.rspd[i], J->s_rspd[i]);
    if (J->A.gld)
        for (int i = tid; i < std::min(kGldLds, J->A.gld0_ub - J->A.gld0_lb + 1); i += 256)
            if (J->s_gld[i] != 0.0) emu::atomic_add(&J->A.gld[i], J->s_gld[i]);
}

static void run_block(Job* J) {
    void (*fn)(Job*, int) = nullptr;
#define PICK(QQ, PP, UU) if (J->q == QQ && J->pe == PP && J->update == UU) fn = lane_body<QQ, PP, UU>;
    PICK(false, false, false) PICK(false, false, true) PICK(true, false, false) PICK(true, false, true)
    PICK(false, true, false) PICK(false, true, true) PICK(true, true, false) PICK(true, true, true)
#undef PICK
    std::vector<std::thread> th;
    for (int t = 0; t < 256; t++) th.emplace_back(fn, J, t);
    for (auto& t : th) t.join();
}

// ---- the restatement: one alignment at a time, factors in read order ----------------------------------------------------
static double seq_profile_prob(bool q, const double* prof, const uint8_t* rs, const uint8_t* rq, int len, const uint8_t* ref) {
    double p = 1.0;
    for (int i = 0; i < len; i++) p *= prof[((q ? rq[i] : i) * 5 + ref[i]) * 5 + rs[i]];
    return p;
}

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const int type = atoi(argv[1]);
    const unsigned seed = (unsigned)atoi(argv[2]);
    const int estRSPD = atoi(argv[3]), has_mld = atoi(argv[4]);
    const bool q = type == 1 || type == 3, pe = type >= 2;
    std::mt19937_64 rng(seed);
    auto irand = [&](int a, int b) { return (int)(rng() % (uint64_t)(b - a + 1)) + a; };
    auto urand = [&]() { return (double)(rng() >> 11) * (1.0 / 9007199254740992.0); };

    // transcripts: families of near-copies, so that a read's alignments mostly (not always) see identical windows
    const int M = 60;
    std::vector<int32_t> fullLen(M + 1, 0), totLen(M + 1, 0);
    std::vector<std::vector<uint8_t>> tseq(M + 1);
    for (int t = 1; t <= M; t++) {
        if ((t - 1) % 6 == 0) {
            const int len = irand(400, 900);
            tseq[t].resize(len);
            for (auto& b : tseq[t]) b = (uint8_t)(rng() % 100 == 0 ? 4 : rng() & 3);
        } else {
            tseq[t] = tseq[t - 1];
            for (int k = 0; k < 6; k++) tseq[t][(size_t)irand(0, (int)tseq[t].size() - 1)] = (uint8_t)(rng() & 3);
        }
        totLen[t] = (int32_t)tseq[t].size();
        fullLen[t] = totLen[t] - (t % 7 == 0 ? 30 : 0);  // some transcripts carry a poly(A) tail
    }
    std::vector<uint64_t> soff(2 * (size_t)(M + 1), 0), mask_off(M + 2, 0);
    uint64_t tot = 0;
    for (int t = 1; t <= M; t++)
        for (int d = 0; d < 2; d++) { soff[2 * t + d] = tot; tot += ((uint64_t)totLen[t] + 7) / 8 * 8; }
    std::vector<uint8_t> strands(tot + 32, 0);
    for (int t = 1; t <= M; t++)
        for (int p = 0; p < totLen[t]; p++) {
            strands[soff[2 * t] + p] = tseq[t][p];
            const uint8_t b = tseq[t][totLen[t] - p - 1];
            strands[soff[2 * t + 1] + p] = b == 4 ? 4 : 3 - b;
        }
    std::vector<uint32_t> mask_words;
    for (int t = 1; t <= M; t++) {
        mask_off[t] = mask_words.size();
        for (int w = 0; w < (totLen[t] + 31) / 32; w++) mask_words.push_back(rng() % 10 == 0 ? (uint32_t)rng() & (uint32_t)rng() & (uint32_t)rng() : 0u);
    }
    mask_off[M + 1] = mask_words.size();

    // reads and alignments
    const uint64_t N1 = 203;  // (not a multiple of 4: the last wave step has empty groups)
    const int minLen = 30, maxLen = 150;
    std::vector<uint64_t> row_ptr{0};
    std::vector<int32_t> sid_signed, pos, insertL;
    std::vector<uint8_t> lq(N1, 0);
    std::vector<std::vector<uint8_t>> rseq[2], rqual[2];
    for (int m = 0; m < 2; m++) { rseq[m].resize(N1); rqual[m].resize(N1); }
    for (uint64_t i = 0; i < N1; i++) {
        const int fam = irand(0, M / 6 - 1) * 6 + 1;
        const int len1 = irand(minLen, maxLen), len2 = irand(minLen, maxLen);
        const int insert = pe ? irand(std::max(len1, len2), std::min(std::max(len1, len2) + 150, 380)) : len1;
        const int dir = (int)(rng() & 1);
        const int p0 = irand(0, 400 - insert - 3);  // all family members are >= 400 long; alignments are shifted by up to 2
        const int nal = (i % 17 == 0) ? irand(17, 40) : irand(1, 12);  // some reads take several 16-alignment chunks
        lq[i] = (i % 23 == 5) ? 1 : 0;
        for (int k = 0; k < nal; k++) {
            const int t = fam + (k % 6);
            const int p = p0 + ((k / 6) % 3);  // later rounds of the family: shifted windows (different bases)
            sid_signed.push_back(dir ? -t : t);
            pos.push_back(p);
            insertL.push_back(insert);
        }
        row_ptr.push_back(sid_signed.size());
        // the read's bases: the first alignment's window with errors
        const int t0 = fam;
        for (int m = 0; m < (pe ? 2 : 1); m++) {
            const int len = m ? len2 : len1;
            const int d = m ? !dir : dir;
            const int wp = m ? totLen[t0] - p0 - insert : p0;
            rseq[m][i].resize(len);
            rqual[m][i].resize(len);
            for (int k = 0; k < len; k++) {
                uint8_t b = strands[soff[2 * t0 + d] + (uint64_t)std::min(std::max(wp + k, 0), totLen[t0] - 1)];
                if (rng() % 20 == 0) b = (uint8_t)(rng() % 5);
                rseq[m][i][k] = b;
                rqual[m][i][k] = (uint8_t)irand(2, 93);
            }
        }
    }
    const uint64_t nnz = sid_signed.size();
    // packed reads: 8 codes per word, every read on a word boundary
    std::vector<uint64_t> roff8[2], seqw[2], qualw[2];
    std::vector<int32_t> rlen[2];
    for (int m = 0; m < (pe ? 2 : 1); m++) {
        roff8[m].assign(N1 + 1, 0);
        rlen[m].resize(N1);
        for (uint64_t i = 0; i < N1; i++) { rlen[m][i] = (int32_t)rseq[m][i].size(); roff8[m][i + 1] = roff8[m][i] + (rseq[m][i].size() + 7) / 8; }
        seqw[m].assign(roff8[m][N1] + 2, 0);
        qualw[m].assign(roff8[m][N1] + 2, 0);
        for (uint64_t i = 0; i < N1; i++) {
            if (!q) {
                for (size_t k = 0; k < rseq[m][i].size(); k++) seqw[m][roff8[m][i] + k / 8] |= (uint64_t)rseq[m][i][k] << (8 * (k % 8));
                continue;
            }
            // quality models: one 16-bit code per position, positions 8w .. 8w + 3 in seqw[w], 8w + 4 .. 8w + 7 in qualw[w], the pad code
            // past the read's end (DevData of model_block.hpp; k_code_reads of model.hip)
            const size_t l = rseq[m][i].size();
            for (size_t k = 0; k < (l + 7) / 8 * 8; k++) {
                const uint64_t c = k < l ? read_code8(rqual[m][i][k], rseq[m][i][k]) : kPadCode8;
                ((k % 8) < 4 ? seqw : qualw)[m][roff8[m][i] + k / 8] |= c << (16 * (k % 4));
            }
        }
    }
    // same_prev flags (k_window_flags of model.hip): byte-wise comparison of the windows
    auto window = [&](uint64_t j, int m, int len) -> const uint8_t* {
        const int s = sid_signed[j], t = s < 0 ? -s : s, d = s < 0 ? 1 : 0;
        (void)len;
        return m == 0 ? &strands[soff[2 * t + d] + (uint64_t)pos[j]] : &strands[soff[2 * t + !d] + (uint64_t)(totLen[t] - pos[j] - insertL[j])];
    };
    std::vector<uint8_t> same_prev(nnz, 0);
    for (uint64_t i = 0; i < N1; i++)
        for (uint64_t j = row_ptr[i] + 1; j < row_ptr[i + 1]; j++) {
            if (lq[i]) continue;
            for (int m = 0; m < (pe ? 2 : 1); m++)
                if (!memcmp(window(j, m, 0), window(j - 1, m, 0), (size_t)rlen[m][i])) same_prev[j] |= (uint8_t)(1 << m);
        }

    // tables
    const int B = 20;
    std::vector<double> rspd_pdf(B + 2, 0.0), rspd_cdf(B + 2, 0.0);
    for (int i = 1; i <= B; i++) rspd_pdf[i] = 0.2 + urand();
    { double s = 0; for (int i = 1; i <= B; i++) s += rspd_pdf[i]; for (int i = 1; i <= B; i++) { rspd_pdf[i] /= s; rspd_cdf[i] = rspd_cdf[i - 1] + rspd_pdf[i]; } }
    auto make_ld = [&](int lb, int ub, std::vector<double>& pdf, std::vector<double>& cdf) {
        pdf.assign(ub - lb + 1, 0.0); cdf.assign(ub - lb + 1, 0.0);
        double s = 0;
        for (int i = 1; i <= ub - lb; i++) { pdf[i] = 0.1 + urand(); s += pdf[i]; }
        for (int i = 1; i <= ub - lb; i++) { pdf[i] /= s; cdf[i] = cdf[i - 1] + pdf[i]; }
    };
    std::vector<double> gld_pdf, gld_cdf, mld_pdf, mld_cdf;
    const int gld_lb = pe || has_mld ? 20 : minLen - 1, gld_ub = pe || has_mld ? 420 : maxLen;
    make_ld(gld_lb, gld_ub, gld_pdf, gld_cdf);
    make_ld(minLen - 1, maxLen, mld_pdf, mld_cdf);
    const int prof_rows = q ? 100 : maxLen;
    std::vector<double> prof((size_t)prof_rows * 25), noise(q ? 500 : 5), mw(M + 1, 1.0);
    for (auto& v : prof) v = 0.05 + urand();
    for (auto& v : noise) v = 0.05 + 0.5 * urand();
    for (int t = 0; t <= M; t++) mw[t] = t % 11 == 3 ? 0.0 : 0.5 + 0.5 * urand();
    mw[0] = 0.9;
    std::vector<double> theta(M + 1);
    { double s = 0; for (auto& v : theta) { v = urand() < 0.2 ? 0.0 : urand(); s += v; } for (auto& v : theta) v /= s; }
    // (theta[0] x tiny noise probabilities could all clamp to zero: keep a visible noise share)
    DevTables T{};
    T.probF = 0.3; T.seedLen = 25; T.estRSPD = estRSPD; T.B = B; T.rspd_pdf = rspd_pdf.data(); T.rspd_cdf = rspd_cdf.data();
    T.gld_lb = gld_lb; T.gld_ub = gld_ub; T.gld_pdf = gld_pdf.data(); T.gld_cdf = gld_cdf.data();
    T.has_mld = pe ? 1 : has_mld; T.mld_lb = minLen - 1; T.mld_ub = maxLen; T.mld_pdf = mld_pdf.data(); T.mld_cdf = mld_cdf.data();
    T.prof_rows = prof_rows; T.prof = prof.data(); T.noise = noise.data(); T.mw = mw.data();
    DevData D{};
    D.model_type = type; D.M = M; D.N1 = N1; D.nnz = nnz; D.row_ptr = row_ptr.data(); D.hit_row = nullptr;
    D.sid_signed = sid_signed.data(); D.pos = pos.data(); D.insertL = insertL.data();
    for (int m = 0; m < (pe ? 2 : 1); m++) { D.roff8[m] = roff8[m].data(); D.rlen[m] = rlen[m].data(); D.rseq_w[m] = seqw[m].data(); D.rqual_w[m] = qualw[m].data(); }
    D.lq = lq.data(); D.soff = soff.data(); D.refw = (const uint64_t*)strands.data(); D.fullLen = fullLen.data(); D.totLen = totLen.data();
    D.mask_off = mask_off.data(); D.mask_words = mask_words.data(); D.same_prev = same_prev.data();
    // DevData::aw0 .. atot and bit 2 of the flags, with the product's own function (k_alignment_fields of model.hip)
    std::vector<uint32_t> aw0(nnz, 0), aw1(nnz, 0), afull(nnz, 1), atot(nnz, 1);
    for (uint64_t i = 0; i < N1; i++)
        for (uint64_t j = row_ptr[i]; j < row_ptr[i + 1]; j++) {
            if (lq[i]) continue;
            const AlnFields F = pe ? alignment_fields<true>(D, T.seedLen, j) : alignment_fields<false>(D, T.seedLen, j);
            aw0[j] = F.a0; aw1[j] = F.a1; afull[j] = F.full; atot[j] = F.tot;
            if (F.masked) same_prev[j] |= 4;
        }
    D.aw0 = aw0.data(); D.aw1 = aw1.data(); D.afull = afull.data(); D.atot = atot.data();

    // ---- restatement ------------------------------------------------------------------------------------------------------
    std::vector<double> rcp(nnz, 0.0), rncp(N1, 0.0);
    const int gld0_lb = 0, gld0_ub = 500;
    std::vector<double> rprof(prof.size(), 0.0), rnoise(noise.size(), 0.0), rrspd(B + 2, 0.0), rgld(gld0_ub - gld0_lb + 1, 0.0);
    for (uint64_t i = 0; i < N1; i++) {
        if (lq[i]) continue;
        const int len1 = rlen[0][i], len2 = pe ? rlen[1][i] : 0;
        for (uint64_t j = row_ptr[i]; j < row_ptr[i + 1]; j++) {
            ChunkRegs R{};
            const int s = sid_signed[j];
            R.has = true; R.sid = s < 0 ? -s : s; R.dir = s < 0 ? 1 : 0; R.pos = pos[j]; R.insertL = insertL[j];
            R.fullLen = fullLen[R.sid]; R.totLen = totLen[R.sid];
            {   // RefSeq::getMask at the seed / fragment start (SingleModel.h:104-106, PairedEndModel.h:100-102), stated on its own
                const int p = pe ? (R.dir == 0 ? R.pos : R.totLen - R.pos - R.insertL) : (R.dir == 0 ? R.pos : R.totLen - R.pos - T.seedLen);
                R.flags = (p >= 0 && p < R.fullLen && ((mask_words[mask_off[R.sid] + (p >> 5)] >> (p & 31)) & 1u)) ? 4u : 0u;
            }
            const double p1 = seq_profile_prob(q, prof.data(), rseq[0][i].data(), rqual[0][i].data(), len1, window(j, 0, len1));
            const double p2 = pe ? seq_profile_prob(q, prof.data(), rseq[1][i].data(), rqual[1][i].data(), len2, window(j, 1, len2)) : 1.0;
            rcp[j] = pe ? alignment_prob<true>(D, T, R, len1, len2, p1, p2) : alignment_prob<false>(D, T, R, len1, len2, p1, p2);
        }
        const double* lpdf = (pe || T.has_mld) ? T.mld_pdf : T.gld_pdf;
        const int llb = (pe || T.has_mld) ? T.mld_lb : T.gld_lb;
        double pr = lpdf[len1 - llb];
        for (int k = 0; k < len1; k++) pr *= q ? noise[rqual[0][i][k] * 5 + rseq[0][i][k]] : noise[rseq[0][i][k]];
        if (pe) {
            double p2 = lpdf[len2 - llb];
            for (int k = 0; k < len2; k++) p2 *= q ? noise[rqual[1][i][k] * 5 + rseq[1][i][k]] : noise[rseq[1][i][k]];
            pr *= p2;
        }
        if (pr < kEpsilon) pr = 0.0;
        rncp[i] = mw[0] < kEpsilon ? 0.0 : pr / mw[0];
        // E step weights + update
        double f0 = theta[0] * rncp[i];
        if (f0 < kEpsilon) f0 = 0.0;
        double sum = f0;
        for (uint64_t j = row_ptr[i]; j < row_ptr[i + 1]; j++) {
            double f = theta[std::abs(sid_signed[j])] * rcp[j];
            if (f < kEpsilon) f = 0.0;
            sum += f;
        }
        if (!(sum >= kEpsilon)) continue;
        for (uint64_t j = row_ptr[i]; j < row_ptr[i + 1]; j++) {
            const int s = sid_signed[j], t = s < 0 ? -s : s, d = s < 0 ? 1 : 0;
            double f = theta[t] * rcp[j];
            if (f < kEpsilon) f = 0.0;
            const double w = f / sum;
            if (w < kEpsilon) continue;
            auto radd = [&](int fpos, int fl) {  // RSPD::update with plain adds
                if (fpos >= fl) return;
                int k;
                double a = fpos * 1.0 / fl, b;
                for (k = (int)(((long long)fpos) * B / fl + 1); k < (int)((((long long)fpos + 1) * B - 1) / fl + 1); k++) { b = k * 1.0 / B; rrspd[k] += (b - a) * fl * w; a = b; }
                b = (fpos + 1.0) / fl;
                rrspd[k] += (b - a) * fl * w;
            };
            if (!pe) {
                if (estRSPD) {
                    if (T.probF >= 0.1 && d == 0) radd(pos[j], fullLen[t]);
                    if (T.probF < 0.1 && d == 1) radd(totLen[t] - pos[j] - len1, fullLen[t]);
                }
            } else {
                rgld[insertL[j] - gld0_lb] += w;
                if (estRSPD) radd(d == 0 ? pos[j] : totLen[t] - pos[j] - insertL[j], fullLen[t]);
            }
            for (int m = 0; m < (pe ? 2 : 1); m++) {
                const uint8_t* ref = window(j, m, 0);
                for (int k = 0; k < rlen[m][i]; k++) rprof[((q ? rqual[m][i][k] : k) * 5 + ref[k]) * 5 + rseq[m][i][k]] += w;
            }
        }
        const double wn = f0 / sum;
        if (wn >= kEpsilon)
            for (int m = 0; m < (pe ? 2 : 1); m++)
                for (int k = 0; k < rlen[m][i]; k++) rnoise[q ? rqual[m][i][k] * 5 + rseq[m][i][k] : rseq[m][i][k]] += wn;
    }

    // ---- the kernel body, emulated: once without and once with the update ---------------------------------------------------
    int bad = 0;
    auto cmp = [&](const char* what, const std::vector<double>& a, const std::vector<double>& b, double tol) {
        double worst = 0.0;
        size_t at = 0;
        for (size_t i = 0; i < a.size(); i++) {
            const double d = fabs(a[i] - b[i]) / std::max(fabs(b[i]), 1e-290);
            if (a[i] != b[i] && d > worst) { worst = d; at = i; }
        }
        printf("%-22s n=%zu max rel diff %.3g%s\n", what, a.size(), worst, worst > tol ? "   <-- MISMATCH" : "");
        if (worst > tol) { printf("   at %zu: %.17g vs %.17g\n", at, a[at], b[at]); ++bad; }
    };
    // the EM context's sliced layout of these reads (simt_emu.hpp builds it with sell_layout.hpp's own helpers): the kernel
    // writes the values in place, and must leave exactly what the scatter pass (sell_fill_row) makes of its CSR output
    std::vector<int32_t> sid_abs(nnz);
    for (uint64_t j = 0; j < nnz; j++) sid_abs[j] = std::abs(sid_signed[j]);
    HostLayout H;
    H.T = 3;
    build_layout(H, M, N1, row_ptr.data(), sid_abs.data(), nullptr, nullptr, 0, false, 0);
    std::vector<uint32_t> rank(N1);
    for (uint64_t p = 0; p < N1; p++) rank[H.order[p]] = (uint32_t)p;
    for (int update = 0; update < 2; update++) {
        std::vector<unsigned char> sval(H.sval.size(), 0);
        std::vector<double> sncp(H.sncp.size(), 0.0);
        std::vector<double> cp(nnz, -1.0), ncp(N1, -1.0), aprof(prof.size(), 0.0), anoise(noise.size(), 0.0), arspd(B + 2, 0.0), agld(rgld.size(), 0.0);
        Job* J = new Job();
        pthread_barrier_init(&J->blk.bar, nullptr, 256);
        for (int w = 0; w < 4; w++) pthread_barrier_init(&J->blk.w[w].bar, nullptr, 64);
        J->D = D; J->T = T; J->theta = theta.data(); J->cp = cp.data(); J->ncp = ncp.data();
        J->A = AccumPtrs{aprof.data(), anoise.data(), estRSPD ? arspd.data() : nullptr, pe ? agld.data() : nullptr, gld0_lb, gld0_ub};
        J->PO = PlaneOut{rank.data(), H.shapes.data(), (int)H.shapes.size(), H.T, (uint32_t)N1, sval.data(), sncp.data()};
        J->q = q; J->pe = pe; J->update = update != 0;
        J->n_blocks = 3;
        for (J->block = 0; J->block < J->n_blocks; J->block++) run_block(J);
        delete J;
        {   // planes written in place == the scatter pass over the kernel's own CSR output, bit for bit
            std::vector<unsigned char> want(H.sval.size(), 0);
            std::vector<double> wncp(H.sncp.size(), 0.0);
            for (const Shape& S : H.shapes)
                for (uint32_t qq = 0; qq < S.n_rows; qq++)
                    sell_fill_row<false>(S, H.T, S.row_base + qq, H.order.data(), row_ptr.data(), nullptr, cp.data(), ncp.data(), nullptr, want.data(), wncp.data(), nullptr, nullptr);
            const bool same = !memcmp(want.data(), sval.data(), want.size()) && !memcmp(wncp.data(), sncp.data(), wncp.size() * 8);
            printf("%-22s %zu bytes of value planes, %zu row slots: %s\n", "planes in place", want.size(), wncp.size(), same ? "identical" : "DIFFERENT   <-- MISMATCH");
            if (!same) ++bad;
        }
        cmp(update ? "conprb (update pass)" : "conprb", cp, rcp, 1e-12);
        cmp(update ? "noise conprb (update)" : "noise conprb", ncp, rncp, 1e-12);
        if (update) {
            cmp("profile counts", aprof, rprof, 1e-10);
            cmp("noise profile counts", anoise, rnoise, 1e-10);
            if (estRSPD) cmp("rspd counts", arspd, rrspd, 1e-10);
            if (pe) cmp("fragment length counts", agld, rgld, 1e-10);
            double tot = 0;
            for (double v : aprof) tot += v;
            if (!(tot > 0)) { printf("profile counts are all zero: the test data does not exercise the update\n"); ++bad; }
        }
    }
    // how much of the special cases the data holds
    int long_rows = 0, shared = 0, contd = 0;
    for (uint64_t i = 0; i < N1; i++) {
        if (row_ptr[i + 1] - row_ptr[i] > 16) ++long_rows;
        for (uint64_t j = row_ptr[i]; j < row_ptr[i + 1]; j++) {
            if (same_prev[j] & 3) ++shared;
            if ((same_prev[j] & 3) && (j - row_ptr[i]) % 16 == 0) ++contd;
        }
    }
    printf("reads %llu alignments %llu, reads with > 16 alignments %d, alignments sharing a window with their predecessor %d, of them first of a chunk %d\n",
           (unsigned long long)N1, (unsigned long long)nnz, long_rows, shared, contd);
    if (!long_rows || !shared || !contd) { printf("test data lacks a special case\n"); ++bad; }
    return bad ? 1 : 0;
}
