// inflate_fast.hpp -- a DEFLATE decoder (RFC 1951) for the BGZF blocks of BAM INPUT in the -b pass (host/bam_io.hpp): one block's raw
// deflate stream -> exactly `isize` bytes, or `false`.
//
// BAM input is inflated before anything else can happen to it, and with the pass's own encoder (deflate_fast.hpp) zlib's inflate had
// become its largest stage (31 of 110 core-seconds at 10 % of configs[2], profiles/r06r_*).  zlib decodes a symbol per table look-up with
// a bit buffer refilled a byte at a time and copies matches byte by byte when they overlap; here: a 64-bit bit buffer refilled 8
// bytes at a time, one look-up in an 11-bit table for almost every literal / length code (longer codes through a second level), a
// 9-bit table for distances, matches copied 8 bytes at a time (16 where the distance allows), and a fast loop that checks its bounds
// once per token with a margin on either side; the last bytes of a block take a careful loop.
//
// The caller does not trust it: bam_io.hpp compares the CRC-32 of what comes out with the block's own (crc32_fold.hpp makes that
// cheap) and hands the block to zlib when this decoder says `false` or the checksum differs -- a corrupt or hostile input ends in
// zlib's error path as before.  Every read and write below is bounds-checked against the buffers the caller names (the input may be
// anything); tests/test_deflate_fast_cpu.py runs it on zlib's output at every level and strategy, on deflate_fast.hpp's, on stored and
// fixed-code blocks, on truncated and bit-flipped streams (under ASan + UBSan too).
#pragma once
#include <cstdint>
#include <cstring>

namespace rsemh {

class FastInflate {
   public:
    // in[0 .. n_in): a complete raw deflate stream; out[0 .. n_out): where its n_out bytes go -- not a byte beyond is written (the
    // blocks of a super-chunk are inflated side by side into one buffer).  true = the stream ended with its final block exactly at
    // n_out bytes.
    bool inflate(const uint8_t* in, size_t n_in, uint8_t* out, size_t n_out) {
        in_ = in; in_end_ = in + n_in; ip_ = in;
        bb_ = 0; bn_ = 0;
        size_t op = 0;
        for (;;) {
            refill();
            if (bn_ < 3) return false;
            const unsigned fin = (unsigned)(bb_ & 1u), type = (unsigned)((bb_ >> 1) & 3u);
            drop(3);
            if (type == 0) {
                // stored: the rest of the byte is skipped, LEN / NLEN, the bytes
                drop(bn_ & 7);
                refill();
                if (bn_ < 32) return false;
                const unsigned len = (unsigned)(bb_ & 0xffffu), nlen = (unsigned)((bb_ >> 16) & 0xffffu);
                drop(32);
                if ((len ^ 0xffffu) != nlen) return false;
                // the bit buffer holds whole bytes now: give them back
                ip_ -= bn_ >> 3;
                bb_ = 0; bn_ = 0;
                if ((size_t)(in_end_ - ip_) < len || n_out - op < len) return false;
                memcpy(out + op, ip_, len);
                ip_ += len;
                op += len;
            } else if (type == 1 || type == 2) {
                if (type == 1) { if (!fixed_tables()) return false; }
                else if (!dynamic_tables()) return false;
                if (!codes(out, n_out, op)) return false;
            } else return false;
            if (fin) break;
        }
        return op == n_out;
    }

   private:
    // Decode tables.  An entry: bits 0..3 the code's length in THIS table's index bits (or, for a link, the primary bits), bits 4..7
    // kind and extra-bit count, bits 8.. the value:
    //   literal:        kLit | len,                      value = the byte
    //   length / dist:  kBase | (extra << 4 ... ) etc.   value = the base
    //   end of block:   kEnd
    //   link:           kLink,  value = offset of the sub-table, low bits = its index width
    // Laid out for one 32-bit load per symbol.
    static constexpr int kLitBits = 11, kDistBits = 9;
    static constexpr uint32_t kKindLit = 0u << 28, kKindBase = 1u << 28, kKindEnd = 2u << 28, kKindLink = 3u << 28, kKindBad = 4u << 28, kKindMask = 7u << 28;
    // entry = kind | (extra bits << 20) | (value << 4) | code length (1..15; 0 in an unused slot -> kKindBad)
    static uint32_t make(uint32_t kind, unsigned extra, unsigned value, unsigned len) { return kind | ((uint32_t)extra << 20) | ((uint32_t)value << 4) | len; }

    uint32_t lit_[(1 << kLitBits) + 288 * 16];   // primary + the sub-tables (codes of <= 15 bits: <= 4 more bits; a table that does not fit fails the block)
    uint32_t dist_[(1 << kDistBits) + 32 * 64];  // (<= 6 more bits)
    size_t lit_size_ = 0, dist_size_ = 0;

    const uint8_t *in_ = nullptr, *in_end_ = nullptr, *ip_ = nullptr;
    uint64_t bb_ = 0;
    unsigned bn_ = 0;

    inline void refill() {  // at least 56 bits if the input has them
        if (in_end_ - ip_ >= 8) {
            uint64_t v;
            memcpy(&v, ip_, 8);
            bb_ |= v << bn_;
            const unsigned take = (63u - bn_) >> 3;
            ip_ += take;
            bn_ += take * 8;
        } else {
            while (bn_ <= 56 && ip_ < in_end_) { bb_ |= (uint64_t)(*ip_++) << bn_; bn_ += 8; }
        }
    }
    inline void drop(unsigned n) { bb_ >>= n; bn_ -= n; }

    // canonical Huffman codes of `lens[0 .. n)` -> a two-level table with `pbits` primary bits; sym_entry(s, len) gives the entry of
    // symbol s without its length.  false: over-subscribed or (with more than one code) incomplete.
    template <typename F>
    bool build(const uint8_t* lens, int n, int pbits, uint32_t* tab, size_t cap, size_t& used, F sym_entry) {
        int count[16] = {0};
        for (int s = 0; s < n; s++) count[lens[s]]++;
        count[0] = 0;
        int nz = 0;
        for (int b = 1; b <= 15; b++) nz += count[b];
        const size_t psize = (size_t)1 << pbits;
        for (size_t i = 0; i < psize; i++) tab[i] = kKindBad;
        used = psize;
        if (nz == 0) return true;  // (a block that never uses this alphabet: every look-up is an error)
        // Kraft: complete, or a single code of length 1 (RFC 1951 allows one distance code; zlib accepts the same for literals)
        long left = 1;
        for (int b = 1; b <= 15; b++) { left <<= 1; left -= count[b]; if (left < 0) return false; }
        if (left > 0 && !(nz == 1 && count[1] == 1)) return false;
        unsigned next[16];
        { unsigned c = 0; for (int b = 1; b <= 15; b++) { c = (c + (unsigned)count[b - 1]) << 1; next[b] = c; } }
        auto rev = [](unsigned v, int bits) { unsigned r = 0; for (int i = 0; i < bits; i++) { r = (r << 1) | (v & 1u); v >>= 1; } return r; };
        // sub-table width per primary prefix: the longest code below it
        // first pass: codes that fit the primary table; second: the longer ones, grouped by their primary-bits prefix
        unsigned nx[16];
        for (int b = 0; b < 16; b++) nx[b] = next[b];
        // longest code length per prefix (for codes longer than pbits)
        // (prefix = the first pbits bits sent = the low pbits of the reversed code)
        static thread_local uint8_t maxlen[1 << kLitBits];
        bool any_long = false;
        for (int b = pbits + 1; b <= 15; b++) any_long = any_long || count[b];
        if (any_long) memset(maxlen, 0, psize);
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (!l) continue;
            const unsigned code = nx[l]++;
            const unsigned r = rev(code, l);
            if (l <= pbits) {
                const uint32_t e = sym_entry(s) | (uint32_t)l;
                for (size_t i = r; i < psize; i += (size_t)1 << l) tab[i] = e;
            } else {
                const unsigned pre = r & (unsigned)(psize - 1);
                if ((unsigned)l > maxlen[pre]) maxlen[pre] = (uint8_t)l;
            }
        }
        if (!any_long) return true;
        // allocate the sub-tables
        for (size_t pre = 0; pre < psize; pre++) {
            if (!maxlen[pre]) continue;
            const int sb = maxlen[pre] - pbits;
            if (used + ((size_t)1 << sb) > cap) return false;
            tab[pre] = kKindLink | ((uint32_t)used << 4) | (uint32_t)sb;
            for (size_t i = 0; i < ((size_t)1 << sb); i++) tab[used + i] = kKindBad;
            used += (size_t)1 << sb;
        }
        for (int b = 0; b < 16; b++) nx[b] = next[b];
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (!l) continue;
            const unsigned code = nx[l]++;
            if (l <= pbits) continue;
            const unsigned r = rev(code, l);
            const unsigned pre = r & (unsigned)(psize - 1);
            const uint32_t link = tab[pre];
            const size_t base = (link >> 4) & 0xffffffu;
            const int sb = (int)(link & 15u);
            const uint32_t e = sym_entry(s) | (uint32_t)(l - pbits);
            for (size_t i = r >> pbits; i < ((size_t)1 << sb); i += (size_t)1 << (l - pbits)) tab[base + i] = e;
        }
        return true;
    }

    static uint32_t lit_entry(int s) {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        if (s < 256) return make(kKindLit, 0, (unsigned)s, 0);
        if (s == 256) return make(kKindEnd, 0, 0, 0);
        if (s <= 285) return make(kKindBase, lext[s - 257], lbase[s - 257], 0);
        return kKindBad;
    }
    static uint32_t dist_entry(int s) {
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        if (s >= 30) return kKindBad;
        return make(kKindBase, s < 4 ? 0 : (unsigned)(s >> 1) - 1, dbase[s], 0);
    }

    bool fixed_tables() {
        uint8_t l[288 + 32];
        for (int s = 0; s < 144; s++) l[s] = 8;
        for (int s = 144; s < 256; s++) l[s] = 9;
        for (int s = 256; s < 280; s++) l[s] = 7;
        for (int s = 280; s < 288; s++) l[s] = 8;
        for (int s = 0; s < 32; s++) l[288 + s] = 5;
        return build(l, 288, kLitBits, lit_, sizeof(lit_) / 4, lit_size_, lit_entry) && build(l + 288, 32, kDistBits, dist_, sizeof(dist_) / 4, dist_size_, dist_entry);
    }

    bool dynamic_tables() {
        refill();
        if (bn_ < 14) return false;
        const int hlit = (int)(bb_ & 31u) + 257, hdist = (int)((bb_ >> 5) & 31u) + 1, hclen = (int)((bb_ >> 10) & 15u) + 4;
        drop(14);
        if (hlit > 286 || hdist > 30) return false;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int k = 0; k < hclen; k++) {
            if (bn_ < 3) { refill(); if (bn_ < 3) return false; }
            cl[order[k]] = (uint8_t)(bb_ & 7u);
            drop(3);
        }
        uint32_t ctab[1 << 7];
        size_t cused;
        if (!build(cl, 19, 7, ctab, 1 << 7, cused, [](int s) { return make(kKindLit, 0, (unsigned)s, 0); })) return false;
        uint8_t lens[286 + 30 + 16];
        int n = 0;
        while (n < hlit + hdist) {
            refill();
            const uint32_t e = ctab[bb_ & 127u];
            const unsigned l = e & 15u;
            if ((e & kKindMask) != kKindLit || l == 0 || l > bn_) return false;
            drop(l);
            const unsigned s = (e >> 4) & 0xffffu;
            if (s < 16) { lens[n++] = (uint8_t)s; continue; }
            unsigned rep, v = 0;
            if (s == 16) {
                if (n == 0 || bn_ < 2) return false;
                v = lens[n - 1];
                rep = 3 + (unsigned)(bb_ & 3u);
                drop(2);
            } else if (s == 17) {
                if (bn_ < 3) return false;
                rep = 3 + (unsigned)(bb_ & 7u);
                drop(3);
            } else {
                if (bn_ < 7) return false;
                rep = 11 + (unsigned)(bb_ & 127u);
                drop(7);
            }
            if (n + (int)rep > hlit + hdist) return false;
            while (rep--) lens[n++] = (uint8_t)v;
        }
        if (lens[256] == 0) return false;  // no end-of-block code
        return build(lens, hlit, kLitBits, lit_, sizeof(lit_) / 4, lit_size_, lit_entry) &&
               build(lens + hlit, hdist, kDistBits, dist_, sizeof(dist_) / 4, dist_size_, dist_entry);
    }

    // the symbols of one block, up to its end-of-block code.  Two loops over the same tables: the fast one runs while the input has
    // 16 bytes and the output 258 + 16 bytes to spare -- a refill then always gives 56 bits, more than the longest token takes (48), and
    // a copy may write whole words -- so it checks neither; the careful one finishes the block.
    bool codes(uint8_t* out, size_t n_out, size_t& op_io) {
        size_t op = op_io;
        const uint32_t lmask = (1u << kLitBits) - 1u, dmask = (1u << kDistBits) - 1u;
        while (in_end_ - ip_ >= 16 && n_out - op >= 258 + 16) {
            refill();
            uint32_t e = lit_[bb_ & lmask];
            if (__builtin_expect((e & kKindMask) == kKindLit && (e & 15u) != 0u, 1)) {  // literals: up to three from one refill (3 x 15 bits)
                drop(e & 15u);
                out[op++] = (uint8_t)(e >> 4);
                e = lit_[bb_ & lmask];
                if ((e & kKindMask) == kKindLit && (e & 15u) != 0u) {
                    drop(e & 15u);
                    out[op++] = (uint8_t)(e >> 4);
                    e = lit_[bb_ & lmask];
                    if ((e & kKindMask) == kKindLit && (e & 15u) != 0u) { drop(e & 15u); out[op++] = (uint8_t)(e >> 4); }
                }
                continue;
            }
            if ((e & kKindMask) == kKindLink) {
                const unsigned sb = e & 15u;
                drop(kLitBits);
                e = lit_[((e >> 4) & 0xffffffu) + (bb_ & ((1u << sb) - 1u))];
            }
            const unsigned l = e & 15u;
            const uint32_t kind = e & kKindMask;
            if (l == 0 || kind > kKindEnd) return false;
            drop(l);
            if (kind == kKindLit) { out[op++] = (uint8_t)(e >> 4); continue; }
            if (kind == kKindEnd) { op_io = op; return true; }
            const unsigned lx = (e >> 20) & 31u;
            const unsigned len = ((e >> 4) & 0xffffu) + (unsigned)(bb_ & ((1u << lx) - 1u));
            drop(lx);
            uint32_t d = dist_[bb_ & dmask];
            if ((d & kKindMask) == kKindLink) {
                const unsigned sb = d & 15u;
                drop(kDistBits);
                d = dist_[((d >> 4) & 0xffffffu) + (bb_ & ((1u << sb) - 1u))];
            }
            const unsigned dl = d & 15u;
            if ((d & kKindMask) != kKindBase || dl == 0) return false;
            drop(dl);
            const unsigned dx = (d >> 20) & 31u;
            const size_t dist = ((d >> 4) & 0xffffu) + (size_t)(bb_ & ((1u << dx) - 1u));
            drop(dx);
            if (dist > op) return false;
            uint8_t* dst = out + op;
            const uint8_t* src = dst - dist;
            if (dist >= 16) {
                for (unsigned k = 0; k < len; k += 16) { uint64_t a, b; memcpy(&a, src + k, 8); memcpy(&b, src + k + 8, 8); memcpy(dst + k, &a, 8); memcpy(dst + k + 8, &b, 8); }
            } else if (dist >= 8) {
                for (unsigned k = 0; k < len; k += 8) { uint64_t a; memcpy(&a, src + k, 8); memcpy(dst + k, &a, 8); }
            } else if (dist == 1) {
                memset(dst, src[0], len);
            } else {
                for (unsigned k = 0; k < len; k++) dst[k] = src[k];
            }
            op += len;
        }
        for (;;) {  // the careful loop: every bit and byte asked for first
            refill();
            uint32_t e = lit_[bb_ & lmask];
            if ((e & kKindMask) == kKindLink) {
                const unsigned sb = e & 15u;
                const size_t base = (e >> 4) & 0xffffffu;
                if (bn_ < (unsigned)kLitBits) return false;
                e = lit_[base + ((bb_ >> kLitBits) & ((1u << sb) - 1u))];
                if ((e & 15u) + (unsigned)kLitBits > bn_ || (e & kKindMask) > kKindEnd) return false;
                drop(kLitBits);
            }
            const unsigned l = e & 15u;
            if (l == 0 || l > bn_) return false;
            drop(l);
            const uint32_t kind = e & kKindMask;
            if (kind == kKindLit) {
                if (op >= n_out) return false;
                out[op++] = (uint8_t)(e >> 4);
                continue;
            }
            if (kind == kKindEnd) break;
            if (kind != kKindBase) return false;
            const unsigned lx = (e >> 20) & 31u;
            if (lx > bn_) return false;
            const unsigned len = ((e >> 4) & 0xffffu) + (unsigned)(bb_ & ((1u << lx) - 1u));
            drop(lx);
            uint32_t d = dist_[bb_ & dmask];
            if ((d & kKindMask) == kKindLink) {
                const unsigned sb = d & 15u;
                const size_t base = (d >> 4) & 0xffffffu;
                if (bn_ < (unsigned)kDistBits) return false;
                d = dist_[base + ((bb_ >> kDistBits) & ((1u << sb) - 1u))];
                if ((d & 15u) + (unsigned)kDistBits > bn_) return false;
                drop(kDistBits);
            }
            const unsigned dl = d & 15u;
            if ((d & kKindMask) != kKindBase || dl == 0 || dl > bn_) return false;
            drop(dl);
            const unsigned dx = (d >> 20) & 31u;
            if (dx > bn_) return false;
            const size_t dist = ((d >> 4) & 0xffffu) + (size_t)(bb_ & ((1u << dx) - 1u));
            drop(dx);
            if (dist > op || len > n_out - op) return false;
            uint8_t* dst = out + op;
            const uint8_t* src = dst - dist;
            for (unsigned k = 0; k < len; k++) dst[k] = src[k];
            op += len;
        }
        op_io = op;
        return true;
    }
};

}  // namespace rsemh
