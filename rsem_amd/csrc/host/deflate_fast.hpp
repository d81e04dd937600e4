// deflate_fast.hpp -- a DEFLATE encoder (RFC 1951) for the blocks of the -b pass's transcript.bam (host/bam_io.hpp): one block of
// up to 65 280 input bytes -> one raw deflate stream (a single dynamic-Huffman block, or a stored one where that is shorter).
//
// Why not zlib's.  The reference's samtools writes BAM with zlib at level 6 (BamWriter.h:39-146 over bgzf); on 64 threads that is
// what the whole pass waits for (30 MB/s per thread; 49 with shorter chains, bam_io.hpp).  zlib pays per BYTE: a hash insertion for
// every position (also inside a 250-byte match), a chain walk for every position that is not, bits sent a code at a time.  A
// transcript BAM is a stream of records in which every alignment of a read repeats the read's name, bases and qualities -- the
// previous record, one record length back -- and whose first occurrence (packed bases, qualities) matches nothing.  So here:
//   * one hash probe per position (4 bytes -> the last position with that hash) plus the LAST MATCH'S DISTANCE tried first -- the
//     next field of a record repeats at the same distance as the field before it;
//   * a match is extended 8 bytes per compare, and positions inside a long match are not hashed one by one (the next record finds
//     the copy through the positions behind the match's head);
//   * one step of lazy evaluation for short matches only;
//   * tokens are buffered, the two Huffman codes are built from their counts (lengths limited to 15 bits), and the block is
//     written through a 64-bit bit buffer, a token's code and extra bits in one go.
// The output is plain DEFLATE: any inflate reads it (tests/test_deflate_fast_cpu.py holds it to zlib's inflate on records, text,
// runs, noise, every length around the limits, and a few hundred thousand random blocks).  Not a general-purpose compressor: the
// window is the block (positions are 16 bits), and nothing is tuned for inputs that look unlike the above.
#pragma once
#include <cstdint>
#include <cstring>

namespace rsemh {

class FastDeflate {
   public:
    static constexpr size_t kMaxIn = 0xff00;            // input bytes per call at most
    static constexpr size_t kMaxOut = kMaxIn + 5 + 16;  // what a call writes at most (a stored block; slack for the bit buffer's tail)

    // p[0 .. n) -> out; returns the bytes written (<= kMaxOut; out must hold kMaxOut + 8)
    size_t compress(const uint8_t* p, size_t n, uint8_t* out) {
        if (n == 0) { out[0] = 0x03; out[1] = 0x00; return 2; }  // an empty final block with the fixed codes
        parse(p, n);
        build_codes();
        const size_t bits = block_bits();
        if ((bits + 7) / 8 >= n + 5) return stored(p, n, out);
        return emit(out);
    }

   private:
    static constexpr int kHashBits = 15;
    static constexpr int kMinMatch = 4, kMaxMatch = 258;
    static constexpr int kLazyBelow = 24;  // a match shorter than this is weighed against the one a byte later
    static constexpr int kReps = 1;         // distances remembered besides the last one (6: 0.6 % smaller, half the rate)
    static constexpr size_t kCareful = 48;  // bytes behind a long match that are searched with care (parse)
    static constexpr size_t kFar3 = 4096;   // a match of three bytes farther back than this costs more than its literals
    static constexpr int kNumLit = 286, kNumDist = 30, kNumCl = 19;

    // position tables: (call number << 16) | position, so that a new call need not clear them
    uint32_t head_[1 << kHashBits], head2_[1 << kHashBits], head3_[1 << kHashBits];
    uint32_t epoch_ = 0;
    // tokens: bits 0..8 length - 3 + 1 (0: a literal), bits 9..16 the literal; bits 16..31 distance - 1 of a match
    uint32_t tok_[kMaxIn + 8];
    size_t ntok_ = 0;
    uint32_t f_lit_[kNumLit], f_dist_[kNumDist];
    uint8_t l_lit_[kNumLit], l_dist_[kNumDist], l_cl_[kNumCl];
    uint16_t c_lit_[kNumLit], c_dist_[kNumDist], c_cl_[kNumCl];
    int hlit_ = 257, hdist_ = 1, hclen_ = 4;
    uint8_t clseq_[kNumLit + kNumDist];   // the run-length coded sequence of code lengths: symbols 0..18 ...
    uint8_t clext_[kNumLit + kNumDist];   // ... and the value of their extra bits
    int ncl_ = 0;
    uint32_t f_cl_[kNumCl];

    static inline uint32_t load32(const uint8_t* q) { uint32_t v; memcpy(&v, q, 4); return v; }
    static inline uint64_t load64(const uint8_t* q) { uint64_t v; memcpy(&v, q, 8); return v; }
    static inline uint32_t hash4(uint32_t v) { return (v * 2654435761u) >> (32 - kHashBits); }

    // length symbol (257..285) and extra bits of a match length 3..258; distance symbol (0..29) and extra bits of a distance 1..32768
    struct Table256 {
        uint8_t sym[256];
        Table256() {
            static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
            int s = 0;
            for (int len = 3; len <= 258; len++) {
                while (s + 1 < 29 && base[s + 1] <= len) ++s;
                sym[len - 3] = (uint8_t)s;
            }
        }
    };

    static inline int len_sym(int len, int& ebits, int& eval) {
        static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t ext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const Table256 T;
        const int s = T.sym[len - 3];
        ebits = ext[s];
        eval = len - base[s];
        return 257 + s;
    }
    static inline int dist_sym(int dist, int& ebits, int& eval) {
        static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        const unsigned d = (unsigned)dist - 1;
        int s;
        if (d < 4) s = (int)d;
        else {
            const int hb = 31 - __builtin_clz(d);  // d in [2^hb, 2^(hb+1))
            s = 2 * hb + (int)((d >> (hb - 1)) & 1u);
        }
        ebits = s < 4 ? 0 : (s >> 1) - 1;
        eval = dist - base[s];
        return s;
    }
    static inline int match_len(const uint8_t* a, const uint8_t* b, int max) {  // bytes a and b (b behind a) have in common, <= max
        int l = 0;
        while (l + 8 <= max) {
            const uint64_t x = load64(a + l) ^ load64(b + l);
            if (x) return l + (__builtin_ctzll(x) >> 3);
            l += 8;
        }
        while (l < max && a[l] == b[l]) ++l;
        return l;
    }

    void parse(const uint8_t* p, size_t n) {
        if ((++epoch_ & 0xffffu) == 0u || epoch_ == 1u) {  // (65 536 calls later a stale entry could pass for one of this call)
            memset(head_, 0, sizeof(head_));
            memset(head2_, 0, sizeof(head2_));
            memset(head3_, 0, sizeof(head3_));
            if ((epoch_ & 0xffffu) == 0u) ++epoch_;
        }
        const uint32_t tag = epoch_ << 16;
        // an entry of this call -> its position; anything else -> 0xffff (no position: positions are < 0xff00)
        auto pos_of = [tag](uint32_t e) -> unsigned { return (e & 0xffff0000u) == tag ? (e & 0xffffu) : 0xffffu; };
        memset(f_lit_, 0, sizeof(f_lit_));
        memset(f_dist_, 0, sizeof(f_dist_));
        ntok_ = 0;
        size_t i = 0;
        int rep = 0;             // the last match's distance (0: none)
        int reps[kReps];         // the distinct distances before it, most recent first
        for (int k = 0; k < kReps; k++) reps[k] = 0;
        // "Careful" positions: the kCareful bytes behind a long match -- in a stream of BAM records that is the next record's fixed
        // fields (ids, positions, flags, the weight: 36 + a few bytes that repeat field by field from DIFFERENT earlier records),
        // where zlib's chains of candidates find twice the matches one probe does; 15 % of the bytes, a third of the positions that
        // are looked at at all.  There: the last kReps distances, the two last positions with the hash, and matches of three bytes.
        // Everywhere else (a new read's bases and qualities: nothing to find) one probe.
        size_t careful_until = 0;
        auto find = [&](size_t at, int& dist) -> int {
            const int max = (int)(n - at < (size_t)kMaxMatch ? n - at : (size_t)kMaxMatch);
            int best = 0;
            dist = 0;
            if (max < kMinMatch) return 0;  // (the last three bytes go as literals)
            const bool careful = at < careful_until;
            const uint32_t v = load32(p + at);
            if (rep && (size_t)rep <= at && load32(p + at - rep) == v) {  // (rep <= 32768: it was a match's distance)
                best = match_len(p + at, p + at - rep, max);
                dist = rep;
            }
            if (careful && best < 16)
                for (int k = 0; k < kReps; k++) {
                    const int r = reps[k];
                    if (r && (size_t)r <= at && load32(p + at - r) == v) {
                        const int l = match_len(p + at, p + at - r, max);
                        if (l > best) { best = l; dist = r; }
                    }
                }
            const uint32_t h = hash4(v);
            const uint32_t e1 = head_[h];
            const unsigned c = pos_of(e1);
            head_[h] = tag | (uint32_t)at;
            if (c != 0xffffu && best < max && at - c <= 32768u && (int)(at - c) != dist && load32(p + c) == v) {  // (DEFLATE's window)
                const int l = match_len(p + at, p + c, max);
                if (l > best) { best = l; dist = (int)(at - c); }
            }
            if (careful) {
                const unsigned c2 = pos_of(head2_[h]);
                head2_[h] = e1;
                if (c2 != 0xffffu && best < max && at - c2 <= 32768u && (int)(at - c2) != dist && load32(p + c2) == v) {
                    const int l = match_len(p + at, p + c2, max);
                    if (l > best) { best = l; dist = (int)(at - c2); }
                }
                const uint32_t h3 = hash4(v << 8);
                const unsigned c3 = pos_of(head3_[h3]);
                head3_[h3] = tag | (uint32_t)at;
                if (best < 3 && c3 != 0xffffu && at - c3 <= kFar3 && ((load32(p + c3) ^ v) & 0xffffffu) == 0) {
                    best = match_len(p + at, p + c3, max);
                    dist = (int)(at - c3);
                }
                return best >= 3 ? best : 0;
            }
            return best >= kMinMatch ? best : 0;
        };
        auto put_lit = [&](uint8_t b) { tok_[ntok_++] = (uint32_t)b << 9; f_lit_[b]++; };
        auto put_match = [&](int len, int dist) {
            tok_[ntok_++] = (uint32_t)(len - 2) | ((uint32_t)(dist - 1) << 16);
            int eb, ev;
            f_lit_[len_sym(len, eb, ev)]++;
            f_dist_[dist_sym(dist, eb, ev)]++;
        };
        const size_t hash_end = n >= 4 ? n - 3 : 0;  // positions that have four bytes to hash
        while (i < n) {
            int dist = 0, len = i < hash_end ? find(i, dist) : 0;
            if (!len) { put_lit(p[i]); ++i; continue; }
            if (len < kLazyBelow && i + 1 < hash_end) {  // a longer match a byte later wins (this byte goes as a literal)
                int d2 = 0;
                const int l2 = find(i + 1, d2);
                if (l2 > len + 1) {
                    put_lit(p[i]);
                    ++i;
                    len = l2;
                    dist = d2;
                }
            }
            // the match may begin earlier than where it was found (a lazy step, a probe that missed): it takes back the literals it
            // covers (2 % fewer bytes on record streams)
            while (len < kMaxMatch && ntok_ > 0 && (tok_[ntok_ - 1] & 0x1ffu) == 0u && i > (size_t)dist && p[i - 1] == p[i - 1 - (size_t)dist]) {
                --ntok_;
                f_lit_[p[i - 1]]--;
                --i;
                ++len;
            }
            put_match(len, dist);
            if (dist != rep) {  // rep moves to the front of reps, dist leaves them
                int k = 0;
                while (k < kReps - 1 && reps[k] != dist) ++k;
                for (; k > 0; k--) reps[k] = reps[k - 1];
                reps[0] = rep;
                rep = dist;
            }
            // the positions the match covers: hashed where a later record will look for them -- all of a short match, the first and
            // last few of a long one (its middle is found through the neighbours: a match starting there would have started earlier)
            const size_t e = i + (size_t)len;
            const bool careful = i < careful_until;
            auto enter = [&](size_t k) {
                const uint32_t v = load32(p + k), h = hash4(v);
                if (careful) { head2_[h] = head_[h]; head3_[hash4(v << 8)] = tag | (uint32_t)k; }
                head_[h] = tag | (uint32_t)k;
            };
            if (len <= 16) {
                for (size_t k = i + 1; k < e && k < hash_end; k++) enter(k);
            } else {
                for (size_t k = i + 1; k < i + 4 && k < hash_end; k++) enter(k);
                for (size_t k = e - 6; k < e && k < hash_end; k++) enter(k);
            }
            if (len >= 64) careful_until = e + kCareful;
            i = e;
        }
        f_lit_[256] = 1;  // end of block
    }

    // Code lengths (<= max_bits) for the symbols with a non-zero count: Huffman's algorithm on the sorted counts (two queues), the
    // lengths' histogram bent to the limit, lengths handed out by rank.  One used symbol gets length 1.
    static void code_lengths(const uint32_t* freq, int nsym, int max_bits, uint8_t* len) {
        struct Node { uint32_t f; int16_t sym; };
        Node leaf[kNumLit];
        int m = 0;
        for (int s = 0; s < nsym; s++) {
            len[s] = 0;
            if (freq[s]) { leaf[m].f = freq[s]; leaf[m].sym = (int16_t)s; ++m; }
        }
        if (m == 0) return;
        if (m == 1) { len[leaf[0].sym] = 1; return; }
        // sort the leaves by count (insertion sort on <= 286 items is not what this file's time goes into ... but 286^2 / 4 per block
        // is 20 k steps: a shell sort)
        for (int gap = m / 2; gap > 0; gap = gap == 2 ? 1 : (int)(gap / 2.2)) {
            for (int a = gap; a < m; a++) {
                const Node t = leaf[a];
                int b = a;
                for (; b >= gap && (leaf[b - gap].f > t.f || (leaf[b - gap].f == t.f && leaf[b - gap].sym > t.sym)); b -= gap) leaf[b] = leaf[b - gap];
                leaf[b] = t;
            }
        }
        // two-queue Huffman: internal nodes come out in non-decreasing order of weight
        uint32_t wint[kNumLit];
        int16_t parent_leaf[kNumLit], parent_int[kNumLit];
        int li = 0, ii = 0, ni = 0;
        auto take = [&](uint32_t& w, bool& is_leaf) -> int {
            if (li < m && (ii >= ni || leaf[li].f <= wint[ii])) { w = leaf[li].f; is_leaf = true; return li++; }
            w = wint[ii];
            is_leaf = false;
            return ii++;
        };
        for (int k = 0; k < m - 1; k++) {
            uint32_t w1, w2;
            bool l1, l2;
            const int a = take(w1, l1), b = take(w2, l2);
            wint[ni] = w1 + w2;
            if (l1) parent_leaf[a] = (int16_t)ni; else parent_int[a] = (int16_t)ni;
            if (l2) parent_leaf[b] = (int16_t)ni; else parent_int[b] = (int16_t)ni;
            ++ni;
        }
        // depths: the root is the last internal node
        int16_t dint[kNumLit];
        dint[ni - 1] = 0;
        for (int k = ni - 2; k >= 0; k--) dint[k] = (int16_t)(dint[parent_int[k]] + 1);
        int count[64];
        for (int b = 0; b < 64; b++) count[b] = 0;
        for (int k = 0; k < m; k++) {
            int d = dint[parent_leaf[k]] + 1;
            if (d > 63) d = 63;
            count[d]++;
        }
        // bend to max_bits: everything deeper moves up to the limit, then the Kraft sum is brought back to one by lengthening the
        // deepest codes that are still shorter than the limit
        for (int b = max_bits + 1; b < 64; b++) { count[max_bits] += count[b]; count[b] = 0; }
        uint64_t total = 0;
        for (int b = max_bits; b > 0; b--) total += (uint64_t)count[b] << (max_bits - b);
        while (total > ((uint64_t)1 << max_bits)) {
            count[max_bits]--;
            for (int b = max_bits - 1; b > 0; b--)
                if (count[b]) { count[b]--; count[b + 1] += 2; break; }
            total--;
        }
        // rank order: the rarest symbols take the longest codes
        int k = 0;
        for (int b = max_bits; b > 0; b--)
            for (int c = count[b]; c > 0; c--) len[leaf[k++].sym] = (uint8_t)b;
    }

    static void canonical(const uint8_t* len, int nsym, uint16_t* code) {  // RFC 1951 3.2.2, the codes bit-reversed (they are sent MSB first)
        int bl_count[16] = {0};
        for (int s = 0; s < nsym; s++) bl_count[len[s]]++;
        bl_count[0] = 0;
        uint16_t next[16];
        uint16_t c = 0;
        for (int b = 1; b <= 15; b++) { c = (uint16_t)((c + bl_count[b - 1]) << 1); next[b] = c; }
        for (int s = 0; s < nsym; s++) {
            const int l = len[s];
            if (!l) { code[s] = 0; continue; }
            uint16_t v = next[l]++, r = 0;
            for (int b = 0; b < l; b++) { r = (uint16_t)((r << 1) | (v & 1)); v >>= 1; }
            code[s] = r;
        }
    }

    void build_codes() {
        code_lengths(f_lit_, kNumLit, 15, l_lit_);
        code_lengths(f_dist_, kNumDist, 15, l_dist_);
        hlit_ = kNumLit;
        while (hlit_ > 257 && l_lit_[hlit_ - 1] == 0) --hlit_;
        hdist_ = kNumDist;
        while (hdist_ > 1 && l_dist_[hdist_ - 1] == 0) --hdist_;
        // (a block without a match still declares one distance code; inflate accepts a single code of length 0 ... zlib's inflate
        // does, others are stricter: give it length 1)
        bool any_dist = false;
        for (int s = 0; s < hdist_; s++) any_dist = any_dist || l_dist_[s];
        if (!any_dist) l_dist_[0] = 1;
        canonical(l_lit_, kNumLit, c_lit_);
        canonical(l_dist_, kNumDist, c_dist_);
        // the two length sequences as one, run-length coded (RFC 1951 3.2.7)
        uint8_t seq[kNumLit + kNumDist];
        int ns = 0;
        for (int s = 0; s < hlit_; s++) seq[ns++] = l_lit_[s];
        for (int s = 0; s < hdist_; s++) seq[ns++] = l_dist_[s];
        memset(f_cl_, 0, sizeof(f_cl_));
        ncl_ = 0;
        for (int a = 0; a < ns;) {
            const int v = seq[a];
            int run = 1;
            while (a + run < ns && seq[a + run] == v) ++run;
            int left = run;
            if (v == 0) {
                while (left >= 11) { const int r = left > 138 ? 138 : left; clseq_[ncl_] = 18; clext_[ncl_++] = (uint8_t)(r - 11); f_cl_[18]++; left -= r; }
                if (left >= 3) { clseq_[ncl_] = 17; clext_[ncl_++] = (uint8_t)(left - 3); f_cl_[17]++; left = 0; }
                while (left > 0) { clseq_[ncl_] = 0; clext_[ncl_++] = 0; f_cl_[0]++; --left; }
            } else {
                clseq_[ncl_] = (uint8_t)v; clext_[ncl_++] = 0; f_cl_[v]++; --left;
                while (left >= 3) { const int r = left > 6 ? 6 : left; clseq_[ncl_] = 16; clext_[ncl_++] = (uint8_t)(r - 3); f_cl_[16]++; left -= r; }
                while (left > 0) { clseq_[ncl_] = (uint8_t)v; clext_[ncl_++] = 0; f_cl_[v]++; --left; }
            }
            a += run;
        }
        code_lengths(f_cl_, kNumCl, 7, l_cl_);
        canonical(l_cl_, kNumCl, c_cl_);
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        hclen_ = 19;
        while (hclen_ > 4 && l_cl_[order[hclen_ - 1]] == 0) --hclen_;
    }

    size_t block_bits() const {
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        size_t bits = 3 + 5 + 5 + 4 + 3 * (size_t)hclen_;
        for (int k = 0; k < ncl_; k++) bits += l_cl_[clseq_[k]] + (clseq_[k] == 16 ? 2 : clseq_[k] == 17 ? 3 : clseq_[k] == 18 ? 7 : 0);
        for (int s = 0; s < kNumLit; s++) bits += (size_t)f_lit_[s] * (l_lit_[s] + (s >= 257 ? lext[s - 257] : 0));
        for (int s = 0; s < kNumDist; s++) bits += (size_t)f_dist_[s] * (l_dist_[s] + (s < 4 ? 0 : (s >> 1) - 1));
        return bits;
    }

    static size_t stored(const uint8_t* p, size_t n, uint8_t* out) {
        out[0] = 0x01;  // BFINAL = 1, BTYPE = 00, the rest of the byte is padding
        out[1] = (uint8_t)(n & 0xff); out[2] = (uint8_t)(n >> 8);
        out[3] = (uint8_t)(~n & 0xff); out[4] = (uint8_t)((~n >> 8) & 0xff);
        memcpy(out + 5, p, n);
        return n + 5;
    }

    size_t emit(uint8_t* out) {
        uint64_t acc = 0;
        int nb = 0;
        uint8_t* o = out;
        auto put = [&](uint32_t v, int bits) {  // bits <= 32, nb < 32 on entry
            acc |= (uint64_t)v << nb;
            nb += bits;
            if (nb >= 32) { memcpy(o, &acc, 4); o += 4; acc >>= 32; nb -= 32; }
        };
        put(1, 1);  // BFINAL
        put(2, 2);  // dynamic Huffman codes
        put((uint32_t)(hlit_ - 257), 5);
        put((uint32_t)(hdist_ - 1), 5);
        put((uint32_t)(hclen_ - 4), 4);
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int k = 0; k < hclen_; k++) put(l_cl_[order[k]], 3);
        for (int k = 0; k < ncl_; k++) {
            const int s = clseq_[k];
            put(c_cl_[s], l_cl_[s]);
            if (s == 16) put(clext_[k], 2);
            else if (s == 17) put(clext_[k], 3);
            else if (s == 18) put(clext_[k], 7);
        }
        for (size_t t = 0; t < ntok_; t++) {
            const uint32_t tk = tok_[t];
            const int lm = (int)(tk & 0x1ff);
            if (!lm) {
                const int b = (int)((tk >> 9) & 0xff);
                put(c_lit_[b], l_lit_[b]);
                continue;
            }
            int eb, ev;
            const int ls = len_sym(lm + 2, eb, ev);
            // a length code (<= 15 bits) and its extra bits (<= 5) in one go
            put((uint32_t)c_lit_[ls] | ((uint32_t)ev << l_lit_[ls]), l_lit_[ls] + eb);
            const int dist = (int)(tk >> 16) + 1;
            const int ds = dist_sym(dist, eb, ev);
            put((uint32_t)c_dist_[ds] | ((uint32_t)ev << l_dist_[ds]), l_dist_[ds] + eb);  // <= 15 + 13 bits
        }
        put(c_lit_[256], l_lit_[256]);
        while (nb > 0) { *o++ = (uint8_t)(acc & 0xff); acc >>= 8; nb -= 8; }
        return (size_t)(o - out);
    }
};

}  // namespace rsemh
