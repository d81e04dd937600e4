// model_block.hpp -- the per-wave body of the model rounds' kernel (k_model_group of model.hip): alignment probabilities
// (getConPrb / getNoiseConPrb), posterior weights and the model's sufficient statistics of ONE round in one pass over the
// reads, laid out so that every global load of a wave is a contiguous run of words.
//
// Included by model.hip INSIDE its anonymous namespace (after kEpsilon is in scope) and by tests/model_emu.cpp, which runs
// this very code on the CPU (one OS thread per lane, tests/simt_emu.hpp) against a thread-per-alignment restatement.
//
// Mapping.  A read belongs to a GROUP of 16 lanes (4 reads per wave).  Two kinds of work alternate inside a group:
//   * lane = alignment (16 at a time; a read with more takes several chunks): the per-alignment fields (transcript, position,
//     insert length, window flags) are 16 consecutive words -- one coalesced request per field and group -- and everything of
//     getConPrb that is per alignment (orientation, fragment / mate length, RSPD, mask, mw) is computed by its lane;
//   * lane = 8 consecutive positions of the read (16 lanes x 8 = 128 positions per pass): the product over the read of
//     profile entries -- the expensive part of getConPrb -- is computed ONCE per run of alignments whose reference window
//     holds the same bases (DevData::same_prev), every lane multiplying its own 8 factors, the 16 partial products
//     combined by a DPP butterfly; the profile counts of the update are added the same way, one LDS atomic per position
//     and run with the run's summed posterior weight.
// The thread-per-read kernels this replaces (k_weights_csr, k_update_read, k_noise, k_profile_heads + k_conprb) walked
// their read serially with dependent, uncoalesced loads: bound by memory latency at 3-5 waves per SIMD
// (profiles/r04a_model_rounds_full_size.*).
//
// Reference semantics (file:line under /root/reference): getConPrb SingleModel.h:95-146, SingleQModel.h:101-151,
// PairedEndModel.h:90-134, PairedEndQModel.h:94-138; getNoiseConPrb SingleQModel.h:153-162, PairedEndQModel.h:140-155;
// update SingleQModel.h:168-215, PairedEndQModel.h:161-180; updateNoise :217-221 / :182-188; the E step's weights
// EM.cpp:199-244; parts LenDist.h:56-77, RSPD.h:43-75, QProfile.h:88-120, Profile.h:91-120, NoiseQProfile.h:74-114,
// NoiseProfile.h:64-101, RefSeq.h:84-92.
// Arithmetic differences from the reference: the factors of a profile product are multiplied in a different order
// (8 in read order per lane, then a tree over the lanes), posterior weights of a run are summed before they are added to
// the profile counts.  All <= a few ulp; the tests hold the tables to 1e-6.
#pragma once
#include "simt_macros.hpp"

constexpr int kProfLds = 5120;  // doubles of the profile COUNT table kept in LDS (Q: 2500; no-Q: 204 positions)
constexpr int kGldLds = 1024;
constexpr int kRspdLds = 128;
constexpr int kNoiseLds = 512;
// Quality models: the code of a read position (DevData) and the tables' padded sizes -- entries 2500 + 5 r of the profile table and
// entry 500 of the noise table are 1.0 (what the pad code of the positions past a read's end selects).
constexpr unsigned kPadCode8 = 8u * 2500u;
constexpr int kQProbLds = 2528, kQNoiseProbLds = 512;
RSEM_DEVFN unsigned read_code8(unsigned quality, unsigned base) { return 8u * (25u * quality + base); }
// byte offset of the noise table's entry [quality][base] from a code: quality = code / 200 (exact for codes up to the pad's: a
// multiplication and a shift), 8 * (5 quality + base) = code - 160 quality
RSEM_DEVFN unsigned noise_off8(unsigned code8) { return code8 - 160u * ((code8 * 1311u) >> 18); }

struct DevTables {  // device copies of rsem_model_tables
    double probF;
    int seedLen, estRSPD, B;
    const double *rspd_pdf, *rspd_cdf;
    int gld_lb, gld_ub;
    const double *gld_pdf, *gld_cdf;
    int has_mld, mld_lb, mld_ub;
    const double *mld_pdf, *mld_cdf;
    int prof_rows;
    const double* prof;
    const double* noise;
    const double* mw;
};

struct DevData {
    int model_type, M;
    uint64_t N1, nnz;
    const uint64_t* row_ptr;
    const uint32_t* hit_row;
    const int32_t* sid_signed;
    const int32_t* pos;
    const int32_t* insertL;
    // reads, 8 positions per pair of 64-bit words, every read starting on a word boundary.  Models without qualities: rseq_w[w] = the
    // base ids of positions 8w .. 8w + 7, one byte each.  Quality models (round 6): a position is ONE 16-bit code, read_code8(quality,
    // base) = 8 * (25 * quality + base) -- the byte offset of entry [quality][0][base] of the 100 x 5 x 5 profile table, so that the
    // entry for reference base r is at code + 40 r: one multiply-add per position where two byte extracts, two multiply-adds and the
    // selects for positions past the read's end stood (the kernel is bound by instruction issue).  rseq_w[w] = the codes
    // of positions 8w .. 8w + 3, rqual_w[w] = those of 8w + 4 .. 8w + 7; positions past the end hold kPadCode8, whose table entries are 1.
    const uint64_t* roff8[2];   // [N1+1] first word of read i
    const int32_t* rlen[2];     // [N1]
    const uint64_t* rseq_w[2];
    const uint64_t* rqual_w[2];
    const uint8_t* lq;
    // both strands of every transcript as base ids (strand 1 = reverse complement), word-aligned starts
    const uint64_t* soff;       // [2*(M+1)] byte offset of strand dir of transcript sid: soff[2*sid + dir]
    const uint64_t* refw;
    const int32_t* fullLen;
    const int32_t* totLen;
    const uint64_t* mask_off;
    const uint32_t* mask_words;
    // per alignment: bit 0 / bit 1 = the reference window of mate 1 / mate 2 holds the same bases as the window of the
    // read's PREVIOUS alignment (computed once at create: the windows never change); 0 for a read's first alignment;
    // bit 2 = the alignment's start position is masked (RefSeq::getMask at the seed / fragment position: alignment_fields)
    const uint8_t* same_prev;
    // per alignment, computed once (alignment_fields): what getConPrb needs of the alignment's TRANSCRIPT -- lengths, the byte
    // addresses of the mates' windows in the strand array.  Until round 6 the round kernel looked these up per alignment and round:
    // six gathers into per-transcript tables (fullLen, totLen, soff x 2, mask_off -> mask_words) behind the alignment's own fields,
    // a dependent round trip and -- a read's ~11 transcripts lying in ~11 different lines of each table -- most of the kernel's
    // fetched bytes (counters: profiles/r06b_model_group_pmc.json).  Now 16 bytes per alignment streamed beside sid / pos.
    const uint32_t* aw0;     // [nnz] window of mate 1 (byte address into refw)
    const uint32_t* aw1;     // [nnz] window of mate 2 (paired-end only)
    const uint32_t* afull;   // [nnz] fullLen of the transcript
    const uint32_t* atot;    // [nnz] totLen of the transcript
};

// Where the sliced layout of the EM context (sell_layout.hpp) keeps the values of a read: with this the round kernel writes
// every alignment probability straight into its value plane (and the noise probability into its row slot) instead of
// leaving that to a scatter pass over the CSR (k_fill_sell: thread per read, 17.5 ms per round at configs[2]).  F64 planes
// only (the model rounds run before the planes are switched to Q32); reads with > 256 alignments live in the CSR alone.
struct PlaneOut {
    const uint32_t* rank;   // caller row -> sorted row (inverse of SellLayout::d_order); nullptr: no plane output
    const Shape* shapes;    // the layout's shape table (the kernel keeps a copy in LDS: every read looks its shape up)
    int n_shapes;
    uint32_t T, n_sell_rows;
    unsigned char* sval;
    double* sncp;
};
constexpr int kPlaneShapesMax = 96;  // >= kMaxShapes of sell_layout.hpp

// the shape a sorted row belongs to: the last one whose first row is <= ps (binary search over the LDS copy)
RSEM_DEVFN int plane_shape_of(const Shape* shapes, int n_shapes, uint32_t ps) {
    int lo = 0, hi = n_shapes;  // shapes[lo].row_base <= ps < shapes[hi].row_base (hi == n: past the end)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (shapes[mid].row_base <= ps) lo = mid; else hi = mid;
    }
    return lo;
}

struct AccumPtrs {
    double* prof;   // [prof_rows*25]
    double* noise;  // [100*5] or [5]
    double* rspd;   // [B+2]
    double* gld;    // [span0+1]
    int gld0_lb, gld0_ub;
};

// LenDist::getAdjustedProb (LenDist.h:63-68)
RSEM_DEVFN double ld_adj(const double* pdf, const double* cdf, int lb, int ub, int len, int refL) {
    if (len <= lb || len > ub || refL <= lb) return 0.0;
    return pdf[len - lb] / cdf[(ub < refL ? ub : refL) - lb];
}
// RSPD::evalCDF / getAdjustedProb (RSPD.h:63-75)
RSEM_DEVFN double rspd_cdf_at(const DevTables& T, int fpos, int fullLen) {
    int i = (int)(((long long)fpos) * T.B / fullLen);
    double val = fpos * 1.0 / fullLen * T.B;
    return T.rspd_cdf[i] + (val - i) * T.rspd_pdf[i + 1];
}
RSEM_DEVFN double rspd_adj(const DevTables& T, int fpos, int effL, int fullLen) {
    if (!T.estRSPD) return 1.0 / effL;
    double denom = rspd_cdf_at(T, effL, fullLen);
    return denom >= kEpsilon ? (rspd_cdf_at(T, fpos + 1, fullLen) - rspd_cdf_at(T, fpos, fullLen)) / denom : 0.0;
}
// RefSeq::get_id (RefSeq.h:84-87) is a table look-up here: both strands are stored, so the base ids of strand
// positions p..p+7 are eight consecutive bytes of the word-aligned array.
RSEM_DEVFN uint64_t funnel8(uint64_t w0, uint64_t w1, int sh) { return sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0; }
RSEM_DEVFN bool ref_mask(const DevData& D, int sid, int p) {  // RefSeq.h:89-92
    return (D.mask_words[D.mask_off[sid] + (p >> 5)] >> (p & 31)) & 1u;
}
// The per-alignment fields above, from the per-transcript tables (alignment j of a read that is not low-quality; seedLen: the
// model's, SingleQModel.h:108-110).  `masked` = what getConPrb asks RefSeq::getMask for this alignment (SingleModel.h:104-106,
// PairedEndModel.h:100-102); the position is range-checked here, the kernel's own test `pos >= fullLen` comes first there too.
struct AlnFields { uint32_t a0, a1, full, tot; bool masked; };
template <bool kPE>
RSEM_DEVFN AlnFields alignment_fields(const DevData& D, int seedLen, uint64_t j) {
    AlnFields F;
    const int s = D.sid_signed[j];
    const int sid = s < 0 ? -s : s, dir = s < 0 ? 1 : 0;
    const int pos = D.pos[j], ins = kPE ? D.insertL[j] : 0;
    const int full = D.fullLen[sid], tot = D.totLen[sid];
    F.full = (uint32_t)full;
    F.tot = (uint32_t)tot;
    F.a0 = (uint32_t)(D.soff[2 * sid + dir] + (uint64_t)pos);
    F.a1 = kPE ? (uint32_t)(D.soff[2 * sid + (dir ^ 1)] + (uint64_t)(tot - pos - ins)) : 0u;
    const int p = kPE ? (dir == 0 ? pos : tot - pos - ins) : (dir == 0 ? pos : tot - pos - seedLen);
    F.masked = (p >= 0 && p < full) ? ref_mask(D, sid, p) : false;
    return F;
}
RSEM_DEVFN void add_tbl(double* lds, int cap, double* glob, int idx, double v) {
    if (idx < cap) RSEM_LDS_ADD(&lds[idx], v);
    else RSEM_ATOMIC_ADD(&glob[idx], v);
}
// RSPD::update (RSPD.h:43-59)
RSEM_DEVFN void rspd_update(double* lds, double* glob, int B, int fpos, int fullLen, double frac) {
    if (fpos >= fullLen) return;
    int i;
    double a = fpos * 1.0 / fullLen, b;
    for (i = (int)(((long long)fpos) * B / fullLen + 1); i < (int)((((long long)fpos + 1) * B - 1) / fullLen + 1); i++) {
        b = i * 1.0 / B;
        add_tbl(lds, kRspdLds, glob, i, (b - a) * fullLen * frac);
        a = b;
    }
    b = (fpos + 1.0) / fullLen;
    add_tbl(lds, kRspdLds, glob, i, (b - a) * fullLen * frac);
}

// ---- the 16 lanes of a group ------------------------------------------------------------------------------------------
constexpr int kGrp = 16;  // lanes per read

template <int kCtrl>
RSEM_DEVFN double mdl_dpp_take(double v) {  // the value of the lane the DPP control selects (all lanes active here)
    const long long b = RSEM_DOUBLE_AS_LL(v);
    const int lo = RSEM_DPP_MOV((int)(unsigned)b, kCtrl);
    const int hi = RSEM_DPP_MOV((int)(unsigned)(b >> 32), kCtrl);
    return RSEM_LL_AS_DOUBLE(((long long)hi << 32) | (unsigned)lo);
}
// sum / product over the 16 lanes of a group (a DPP row), every lane ends with the result
RSEM_DEVFN double grp_sum(double v) {
    v += mdl_dpp_take<0xB1>(v);   // quad_perm [1,0,3,2]
    v += mdl_dpp_take<0x4E>(v);   // quad_perm [2,3,0,1]
    v += mdl_dpp_take<0x141>(v);  // row_half_mirror
    v += mdl_dpp_take<0x140>(v);  // row_mirror
    return v;
}
RSEM_DEVFN double grp_prod(double v) {
    v *= mdl_dpp_take<0xB1>(v);
    v *= mdl_dpp_take<0x4E>(v);
    v *= mdl_dpp_take<0x141>(v);
    v *= mdl_dpp_take<0x140>(v);
    return v;
}
RSEM_DEVFN int grp_ctz(unsigned m) {  // index of the lowest set bit of a 16-bit mask (m != 0)
    int i = 0;
    while (!((m >> i) & 1u)) ++i;
    return i;
}

// One mate of a group's read: where its packed words are and its length.  (The words of lane g are loaded where they are used,
// every time: L1 hits, against 8 registers per lane that would cost the kernel's fourth wave per SIMD.)
struct MateWords {
    const uint64_t* seq;   // base of the mate's word arrays (uniform: scalar registers) ...
    const uint64_t* qual;
    uint64_t r8;           // ... and the read's first word in them
    int len;
};

// (Q)Profile::getProb over the lane's share of the read (QProfile.h:111-120, Profile.h:114-120): positions 8wi..8wi+7 for
// wi = g, g + 16, ...; reference window starting at byte address a of the strand array.  Returns the lane's partial product.
template <bool kQ>
RSEM_DEVFN double lane_profile_product(const double* prof, const MateWords& W, const uint64_t* __restrict__ refw, uint64_t a, int g, bool on) {
    double p = 1.0;
    if (!on) return p;
    const uint64_t* rw = refw + (a >> 3);
    const int sh = (int)(a & 7) * 8;
    for (int wi = g; wi * 8 < W.len; wi += kGrp) {
        const uint64_t rf = funnel8(rw[wi], rw[wi + 1], sh);
        const uint64_t sb = W.seq[W.r8 + wi];
        const uint64_t qb = kQ ? W.qual[W.r8 + wi] : 0;
        if (kQ) {
            // a position = a 16-bit code (DevData); past the read's end the pad code, whose entries are 1: no index select, no value select
            // (the eight table reads issued together, then the eight multiplications in read order: left to itself the compiler waits
            // for every read before it issues the next)
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const unsigned code = (unsigned)(((u < 4 ? sb : qb) >> (16 * (u & 3))) & 0xffffu);
                const unsigned r = (unsigned)((rf >> (8 * u)) & 0xffu);
                t[u] = *(const double*)((const char*)prof + (code + 40u * r));
            }
            RSEM_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < 8; u++) p *= t[u];
            continue;
        }
        const int n = W.len - wi * 8;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int row = wi * 8 + u;
            const int idx = (row * 5 + (int)((rf >> (8 * u)) & 0xff)) * 5 + (int)((sb >> (8 * u)) & 0xff);
            // (no look-up under a condition: `(u < n) ? prof[idx] : 1.0` compiles to a branch around the load with a wait of
            // its own, eight round trips one after the other; past the read's end entry 0 is read and not used)
            const double t = prof[(u < n) ? idx : 0];
            p *= (u < n) ? t : 1.0;
        }
    }
    return p;
}

// (Q)Profile::update over the lane's share of the read (QProfile.h:88-93, Profile.h:91-96)
template <bool kQ>
RSEM_DEVFN void lane_profile_update(double* s_prof, double* g_prof, const MateWords& W, const uint64_t* __restrict__ refw, uint64_t a, int g,
                                    double frac) {
    const uint64_t* rw = refw + (a >> 3);
    const int sh = (int)(a & 7) * 8;
    for (int wi = g; wi * 8 < W.len; wi += kGrp) {
        const uint64_t rf = funnel8(rw[wi], rw[wi + 1], sh);
        const uint64_t sb = W.seq[W.r8 + wi];
        const uint64_t qb = kQ ? W.qual[W.r8 + wi] : 0;
        const int n = W.len - wi * 8 < 8 ? W.len - wi * 8 : 8;
        for (int u = 0; u < n; u++) {
            if (kQ) {  // (quality < 100: every entry is inside the 2500-entry table in LDS)
                const unsigned code = (unsigned)(((u < 4 ? sb : qb) >> (16 * (u & 3))) & 0xffffu);
                RSEM_LDS_ADD((double*)((char*)s_prof + (code + 40u * (unsigned)((rf >> (8 * u)) & 0xffu))), frac);
                continue;
            }
            // (positions beyond the LDS table go to global memory)
            add_tbl(s_prof, kProfLds, g_prof, ((wi * 8 + u) * 5 + (int)((rf >> (8 * u)) & 0xff)) * 5 + (int)((sb >> (8 * u)) & 0xff), frac);
        }
    }
}

// Noise(Q)Profile::getProb / update over the lane's share of the read (NoiseQProfile.h:74-98, NoiseProfile.h:64-82)
template <bool kQ>
RSEM_DEVFN double lane_noise_product(const double* nprob, const MateWords& W, int g, bool on) {
    double p = 1.0;
    if (!on) return p;
    for (int wi = g; wi * 8 < W.len; wi += kGrp) {
        const uint64_t sb = W.seq[W.r8 + wi];
        const uint64_t qb = kQ ? W.qual[W.r8 + wi] : 0;
        if (kQ) {  // (the pad code selects entry 500 = 1)
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const unsigned code = (unsigned)(((u < 4 ? sb : qb) >> (16 * (u & 3))) & 0xffffu);
                t[u] = *(const double*)((const char*)nprob + noise_off8(code));
            }
            RSEM_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < 8; u++) p *= t[u];
            continue;
        }
        const int n = W.len - wi * 8;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int b = (int)((sb >> (8 * u)) & 0xff);
            const double t = nprob[(u < n) ? b : 0];  // (see lane_profile_product)
            p *= (u < n) ? t : 1.0;
        }
    }
    return p;
}
template <bool kQ>
RSEM_DEVFN void lane_noise_update(double* s_noise, const MateWords& W, int g, double frac) {
    for (int wi = g; wi * 8 < W.len; wi += kGrp) {
        const uint64_t sb = W.seq[W.r8 + wi];
        const uint64_t qb = kQ ? W.qual[W.r8 + wi] : 0;
        const int n = W.len - wi * 8 < 8 ? W.len - wi * 8 : 8;
        for (int u = 0; u < n; u++) {
            if (kQ) {
                const unsigned code = (unsigned)(((u < 4 ? sb : qb) >> (16 * (u & 3))) & 0xffffu);
                RSEM_LDS_ADD((double*)((char*)s_noise + noise_off8(code)), frac);
            } else {
                RSEM_LDS_ADD(&s_noise[(int)((sb >> (8 * u)) & 0xff)], frac);
            }
        }
    }
}

// One chunk of a read: lane g holds alignment c * 16 + g.
struct ChunkRegs {
    bool has;        // this lane holds an alignment of an active (not low-quality) read
    int sid, dir, pos, insertL, fullLen, totLen;
    unsigned flags;  // same_prev bits
    uint64_t a[2];   // window addresses of the two mates (byte addresses into the strand array)
    double cp;       // P(read, alignment | transcript), as written to the CSR
};

// Everything of getConPrb that is per alignment, given the profile products of the mates (pp1, pp2).
template <bool kPE>
RSEM_DEVFN double alignment_prob(const DevData& D, const DevTables& T, const ChunkRegs& R, int len1, int len2, double pp1, double pp2) {
    double prob = 0.0;
    const int sid = R.sid, dir = R.dir, pos = R.pos, fullLen = R.fullLen, totLen = R.totLen;
    if (!kPE) {
        const int fpos = dir == 0 ? pos : totLen - pos - len1;
        const int seedPos = dir == 0 ? pos : totLen - pos - T.seedLen;
        if (!(seedPos >= fullLen || (R.flags & 4u))) {
            double value;
            if (T.has_mld) {  // SingleQModel.h:127-136
                const int minL = len1 > T.gld_lb + 1 ? len1 : T.gld_lb + 1, maxL = totLen - pos < T.gld_ub ? totLen - pos : T.gld_ub;
                value = 0.0;
                for (int fragLen = minL; fragLen <= maxL; fragLen++) {
                    const int pfpos = dir == 0 ? pos : totLen - pos - fragLen;
                    const int effL = fullLen < totLen - fragLen + 1 ? fullLen : totLen - fragLen + 1;
                    value += ld_adj(T.gld_pdf, T.gld_cdf, T.gld_lb, T.gld_ub, fragLen, totLen) * rspd_adj(T, pfpos, effL, fullLen) *
                             ld_adj(T.mld_pdf, T.mld_cdf, T.mld_lb, T.mld_ub, len1, fragLen);
                }
            } else {
                const int effL = fullLen < totLen - len1 + 1 ? fullLen : totLen - len1 + 1;
                value = ld_adj(T.gld_pdf, T.gld_cdf, T.gld_lb, T.gld_ub, len1, totLen) * rspd_adj(T, fpos, effL, fullLen);
            }
            const double ori = dir == 0 ? T.probF : 1.0 - T.probF;
            prob = ori * value * pp1;
            if (prob < kEpsilon) prob = 0.0;
            prob = (T.mw[sid] < kEpsilon) ? 0.0 : prob / T.mw[sid];
        }
    } else {  // PairedEndQModel.h:94-138
        const int insertLen = R.insertL;
        const int fpos = dir == 0 ? pos : totLen - pos - insertLen;
        const int effL = fullLen < totLen - insertLen + 1 ? fullLen : totLen - insertLen + 1;
        if (!(fpos >= fullLen || (R.flags & 4u))) {
            const double ori = dir == 0 ? T.probF : 1.0 - T.probF;
            prob = ori * ld_adj(T.gld_pdf, T.gld_cdf, T.gld_lb, T.gld_ub, insertLen, totLen) * rspd_adj(T, fpos, effL, fullLen);
            prob *= ld_adj(T.mld_pdf, T.mld_cdf, T.mld_lb, T.mld_ub, len1, insertLen) * pp1;
            prob *= ld_adj(T.mld_pdf, T.mld_cdf, T.mld_lb, T.mld_ub, len2, insertLen) * pp2;
            if (prob < kEpsilon) prob = 0.0;
            prob = (T.mw[sid] < kEpsilon) ? 0.0 : prob / T.mw[sid];
        }
    }
    return prob;
}

// The rows [row0, N1) in steps of row_stride, this wave taking rows row0 + (lane >> 4) of every step (4 reads per wave).
// With chunk_rows: the rows are cut into chunks of that many; this WORKGROUP takes chunks chunk0, chunk0 + chunk_stride, ... and walks
// each one with row0 / row_stride counted inside the chunk -- its waves' successive steps then touch neighbouring rows, so the halves
// of a 128-byte line that one step leaves unused are used by the next (same L1, same L2) instead of by another workgroup on another
// XCD (the grid-wide stride of rounds 4-5 fetched 4.2 x the bytes the kernel uses: profiles/r06b_model_group_pmc.json).
// prob / nprob: the profile / noise probability tables (LDS copies where they fit); s_*: the LDS count tables of the update.
// theta (kUpdate): the round's theta, for the posterior weights.  cp / ncp: the CSR values, written for every read.
template <bool kQ, bool kPE, bool kUpdate>
RSEM_DEVFN void model_group_rows(const DevData& D, const DevTables& T, const double* __restrict__ theta, double* __restrict__ cp,
                                 double* __restrict__ ncp, const AccumPtrs& A, const double* prob, const double* nprob, double* s_prof,
                                 double* s_noise, double* s_rspd, double* s_gld, uint64_t row0, uint64_t row_stride, int lane,
                                 const PlaneOut& PO, uint64_t chunk_rows = 0, uint64_t chunk0 = 0, uint64_t chunk_stride = 1) {
    const int g = lane & (kGrp - 1);
    const int g0 = lane & ~(kGrp - 1);  // first lane of my group
    constexpr int kMates = kPE ? 2 : 1;
    // What the kernel needs to know about a read before it can ask for anything else (its row header), loaded for a row that
    // exists (the last one stands in for the rows past the end) and masked afterwards: a load under a condition is a branch with
    // a wait of its own.  (Requesting the NEXT step's header at the top of a step and using it a step later -- 10 more registers per
    // lane -- bought nothing: 13.15 against 13.17 ms per round at a fifth of configs[2], and 20.8 ms with three waves per SIMD to
    // make room for it, profiles/r05q_model_rounds_header_ahead.log.  The step's first round trip is not what the kernel waits for.)
    struct RowHdr {
        uint8_t lq;
        uint64_t fr, to;
        uint64_t r8[2];
        int rl[2];
        uint32_t rank;
    };
    auto load_hdr = [&](uint64_t row) -> RowHdr {
        RowHdr H;
        const uint64_t rowc = row < D.N1 ? row : D.N1 - 1;
        H.lq = D.lq[rowc];
        H.fr = D.row_ptr[rowc];
        H.to = D.row_ptr[rowc + 1];
        H.r8[0] = H.r8[1] = 0;
        H.rl[0] = H.rl[1] = 0;
#pragma unroll
        for (int m = 0; m < (kPE ? 2 : 1); m++) { H.r8[m] = D.roff8[m][rowc]; H.rl[m] = D.rlen[m][rowc]; }
        H.rank = PO.rank ? PO.rank[rowc] : 0u;
        return H;
    };
    const uint64_t n_chunks = chunk_rows ? (D.N1 + chunk_rows - 1) / chunk_rows : 1;
    for (uint64_t ch = chunk_rows ? chunk0 : 0; ch < n_chunks; ch += chunk_stride)  // (workgroup-uniform)
    for (uint64_t rbase = ch * chunk_rows + row0, rend = chunk_rows ? (ch + 1) * chunk_rows : D.N1; rbase < rend && rbase < D.N1; rbase += row_stride) {  // (wave-uniform)
        const uint64_t row = rbase + (uint64_t)(lane >> 4);
        const bool valid = row < D.N1;
        const RowHdr H = load_hdr(row);
        const uint8_t lq_v = H.lq;
        const uint64_t fr_v = H.fr, to_v = H.to;
        uint64_t r8_v[2] = {H.r8[0], H.r8[1]};
        int rl_v[2] = {H.rl[0], H.rl[1]};
        const uint32_t rank_v = H.rank;
        const bool active = valid && !lq_v;
        const uint64_t fr = valid ? fr_v : 0, to = valid ? to_v : 0;
        const int L = (int)(to - fr);
        int maxL = L;
        { int o = RSEM_SHFL_XOR(maxL, 16); maxL = o > maxL ? o : maxL; o = RSEM_SHFL_XOR(maxL, 32); maxL = o > maxL ? o : maxL; }
        MateWords W[kMates];
#pragma unroll
        for (int m = 0; m < kMates; m++) {
            W[m].r8 = active ? r8_v[m] : 0;
            W[m].seq = D.rseq_w[m];
            W[m].qual = kQ ? D.rqual_w[m] : nullptr;
            W[m].len = active ? rl_v[m] : 0;
        }
        const int len1 = W[0].len, len2 = kPE ? W[kMates - 1].len : 0;
        // the read's place in the sliced layout: alignment c goes to plane c >> lg of its slice, lane r * G + (c & (G - 1))
        double* plane = nullptr;
        double* nslot = nullptr;  // the read's noise value in the layout
        int p_lg = 0;
        uint32_t p_r = 0;
        if (PO.rank && valid) {
            const uint32_t ps = rank_v;
            if (ps < PO.n_sell_rows) {
                const Shape& S = PO.shapes[plane_shape_of(PO.shapes, PO.n_shapes, ps)];
                uint32_t slice_local;
                row_to_slot(S, PO.T, ps - S.row_base, slice_local, p_r);
                plane = (double*)(PO.sval + S.val_base) + (uint64_t)slice_local * S.K * 64;
                nslot = PO.sncp + (S.slot_base + slice_local * shape_R(S) + p_r);
                p_lg = S.lg;
                p_r = (p_r << p_lg);  // first lane of the read within a plane row
            }
        }
        auto plane_put = [&](int c_idx, double v) {
            if (plane) plane[(uint64_t)(c_idx >> p_lg) * 64 + p_r + (uint32_t)(c_idx & ((1 << p_lg) - 1))] = v;
        };

        // per-chunk state carried from one chunk of a long read to the next: the window and product of the run that was open
        // at the end of the chunk
        uint64_t carry_a[kMates];
        double carry_p[kMates];
#pragma unroll
        for (int m = 0; m < kMates; m++) { carry_a[m] = 0; carry_p[m] = 0.0; }

        auto load_chunk = [&](int c, ChunkRegs& R) {
            const int idx = c * kGrp + g;
            const bool in = valid && idx < L;
            R.has = active && idx < L;
            // (ONE round trip: the alignment's own fields, among them what getConPrb needs of its transcript, DevData::aw0 .. atot --
            // every load issued for an alignment that exists, the read's first or the file's first for the lanes without one, and
            // masked afterwards; what still hangs on the transcript id -- mw, theta -- is two 1.6 MB tables)
            const uint64_t j = fr + (uint64_t)(in ? idx : 0);
            const int s_v = D.sid_signed[j], pos_v = D.pos[j], ins_v = kPE ? D.insertL[j] : 0;
            const unsigned fl_v = D.same_prev[j];
            const uint32_t a0_v = D.aw0[j], a1_v = kPE ? D.aw1[j] : 0u, full_v = D.afull[j], tot_v = D.atot[j];
            const int s = R.has ? s_v : 1;
            R.sid = s < 0 ? -s : s;
            R.dir = s < 0 ? 1 : 0;
            R.pos = R.has ? pos_v : 0;
            R.insertL = R.has ? ins_v : 0;
            R.flags = R.has ? fl_v : 0u;
            R.fullLen = R.has ? (int)full_v : 1;
            R.totLen = R.has ? (int)tot_v : 1;
            R.a[0] = R.has ? (uint64_t)a0_v : 0;
            if (kPE) R.a[kMates - 1] = R.has ? (uint64_t)a1_v : 0;
            R.cp = 0.0;
            if (in && !R.has) { cp[j] = 0.0; plane_put(idx, 0.0); }  // low-quality read: every alignment gets probability 0 (SingleQModel.h:102)
        };
        // Runs of one mate within a chunk.  heads: bit i = lane i of the group starts a run here (its window differs from its
        // predecessor's, or it is lane 0 of a later chunk, where the run open at the end of the previous chunk continues:
        // then the run's window is carry_a and, for the products, its value is carry_p -- nothing is recomputed).
        // For every run: fn(window address, first lane, one-past-last lane, continues_previous_chunk)
        auto runs_of = [&](const ChunkRegs& R, int m, int c, auto&& fn) {
            const unsigned bit = 1u << m;
            const bool cont0 = (g == 0) && c > 0 && R.has && (R.flags & bit);
            const bool head = R.has && (!(R.flags & bit) || g == 0);
            const unsigned long long hb = RSEM_BALLOT(head);
            unsigned hg = (unsigned)((hb >> g0) & 0xffffull);
            const uint64_t ea = cont0 ? carry_a[m] : R.a[m];
            const int c0 = RSEM_SHFL((int)cont0, g0);  // does the group's first run continue the previous chunk's?
            bool first = true;
            uint64_t last_a = carry_a[m];
            while (RSEM_BALLOT(hg != 0u) != 0ull) {  // (wave-uniform: the 4 groups step through their runs together)
                const bool on = hg != 0u;
                const int h = on ? grp_ctz(hg) : 0;
                const unsigned rest = hg & (hg - 1u);
                const int nh = (on && rest) ? grp_ctz(rest) : kGrp;
                const uint64_t ah = RSEM_SHFL(ea, g0 + h);
                fn(on, ah, h, nh, on && first && c0 != 0);
                if (on) last_a = ah;
                first = false;
                hg = rest;
            }
            carry_a[m] = last_a;
        };

        double rowsum = 0.0;  // sum over the read's alignments of theta * conprb (each clamped), for the weights
        const int nchunks = (maxL + kGrp - 1) / kGrp;
        for (int c = 0; c < nchunks; c++) {
            ChunkRegs R;
            load_chunk(c, R);
            double pp[kMates];
#pragma unroll
            for (int m = 0; m < kMates; m++) {
                pp[m] = carry_p[m];
                double lastp = carry_p[m];
                runs_of(R, m, c, [&](bool on, uint64_t ah, int h, int nh, bool continues) {
                    double p;
                    if (RSEM_BALLOT(on && !continues) != 0ull) {  // (a step in which every group only continues computes nothing)
                        p = grp_prod(lane_profile_product<kQ>(prob, W[m], D.refw, ah, g, on && !continues));
                        if (continues) p = carry_p[m];
                    } else p = carry_p[m];
                    if (on && g >= h && g < nh) pp[m] = p;
                    if (on) lastp = p;
                });
                carry_p[m] = lastp;
            }
            if (R.has) {
                R.cp = alignment_prob<kPE>(D, T, R, len1, len2, pp[0], kPE ? pp[kMates - 1] : 1.0);
                cp[fr + (uint64_t)(c * kGrp + g)] = R.cp;
                plane_put(c * kGrp + g, R.cp);
            }
            if (kUpdate) {
                const double th_v = theta[R.sid];  // (sid = 1 for a lane without an alignment)
                double f = R.has ? th_v * R.cp : 0.0;
                if (f < kEpsilon) f = 0.0;
                rowsum += grp_sum(f);
            }
        }
        // noise (getNoiseConPrb): SingleQModel.h:153-162, PairedEndQModel.h:140-155
        double nval = 0.0;
        {
            const double* lpdf = (kPE || T.has_mld) ? T.mld_pdf : T.gld_pdf;
            const int llb = (kPE || T.has_mld) ? T.mld_lb : T.gld_lb;
            double p = grp_prod(lane_noise_product<kQ>(nprob, W[0], g, active));
            double pr = active ? lpdf[len1 - llb] * p : 0.0;
            if (kPE) {
                const double p2 = grp_prod(lane_noise_product<kQ>(nprob, W[kMates - 1], g, active));
                if (active) pr *= lpdf[len2 - llb] * p2;
            }
            if (pr < kEpsilon) pr = 0.0;
            nval = (T.mw[0] < kEpsilon) ? 0.0 : pr / T.mw[0];
            if (!active) nval = 0.0;
            if (valid && g == 0) {
                ncp[row] = nval;
                if (nslot) *nslot = nval;
            }
        }
        if (!kUpdate) continue;

        // ---- posterior weights of this round (EM.cpp:199-244) and the model's statistics -------------------------------
        double f0 = theta[0] * nval;
        if (f0 < kEpsilon) f0 = 0.0;
        const double sum = rowsum + f0;
        const bool ok = active && sum >= kEpsilon;
        const double wn = ok ? f0 / sum : 0.0;
#pragma unroll
        for (int m = 0; m < kMates; m++) carry_a[m] = 0;
        for (int c = 0; c < nchunks; c++) {
            // (the chunk is loaded again rather than kept: L1 / L2 hits, against 16 registers that cost the kernel its fourth
            // wave per SIMD or 14 GB of scratch traffic per launch at a fifth of configs[2], profiles/r04c_model_group_pmc_fifth_size.json)
            ChunkRegs R;
            load_chunk(c, R);
            const double cp_v = cp[fr + (uint64_t)((valid && c * kGrp + g < L) ? c * kGrp + g : 0)], th_v = theta[R.sid];
            R.cp = R.has ? cp_v : 0.0;
            double f = R.has ? th_v * R.cp : 0.0;
            if (f < kEpsilon) f = 0.0;
            double w = ok ? f / sum : 0.0;
            if (w < kEpsilon) w = 0.0;  // `if (frac < kEpsilon) continue;` of the update loops
            if (w > 0.0) {
                if (!kPE) {
                    if (T.estRSPD) {  // only one strand estimates the RSPD; helper models have no mld (SingleQModel.h:176-213)
                        if (T.probF >= 0.1 && R.dir == 0) rspd_update(s_rspd, A.rspd, T.B, R.pos, R.fullLen, w);
                        if (T.probF < 0.1 && R.dir == 1) rspd_update(s_rspd, A.rspd, T.B, R.totLen - R.pos - len1, R.fullLen, w);
                    }
                } else {
                    add_tbl(s_gld, kGldLds, A.gld, R.insertL - A.gld0_lb, w);  // LenDist::update (LenDist.h:46-49)
                    if (T.estRSPD) {
                        const int fpos = R.dir == 0 ? R.pos : R.totLen - R.pos - R.insertL;
                        rspd_update(s_rspd, A.rspd, T.B, fpos, R.fullLen, w);
                    }
                }
            }
            // profile counts: the weights of a run are summed, then every lane adds the sum at its 8 positions
#pragma unroll
            for (int m = 0; m < kMates; m++)
                runs_of(R, m, c, [&](bool on, uint64_t ah, int h, int nh, bool) {
                    const double gw = grp_sum((on && g >= h && g < nh) ? w : 0.0);
                    if (on && gw > 0.0) lane_profile_update<kQ>(s_prof, A.prof, W[m], D.refw, ah, g, gw);
                });
        }
        if (wn >= kEpsilon) {  // updateNoise (SingleQModel.h:217-221, PairedEndQModel.h:182-188)
#pragma unroll
            for (int m = 0; m < kMates; m++) lane_noise_update<kQ>(s_noise, W[m], g, wn);
        }
    }
}
