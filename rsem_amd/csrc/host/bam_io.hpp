// bam_io.hpp -- the `-b` option of rsem-run-em (BamWriter.h:39-146, sam_utils.h:72-76, SamHeader.hpp): copy the
// input alignments to <sample>.transcript.bam with MAPQ and a ZW:f tag set from each alignment's posterior
// weight.  The reference links htslib for this; here the two formats involved are implemented directly on
// zlib: a BGZF reader/writer, a BAM record walker, and a SAM-text -> BAM record encoder (SAM spec v1).
//
// The reference hands the output to htslib with nThreads compression threads (BamWriter.h:72, hts_set_threads) and reads,
// re-weights and writes the records on one thread.  Here the whole pass is cut into pieces that N threads take side by side
// (write_transcript_bam below): BGZF blocks are independent deflate streams of at most 64 KB, so a piece of the input becomes a
// run of blocks of its own, and the pieces' runs are written in order; which alignment weight a piece starts with is settled by
// a counting pass before it.  -b is on by default in rsem-calculate-expression (:61,626-632): at BASELINE configs[2] that is
// 560 M records through zlib.
#pragma once
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>

#include "crc32_fold.hpp"
#include "deflate_fast.hpp"
#include "inflate_fast.hpp"
#include "files.hpp"

namespace rsemh {

// ---- BGZF (blocked gzip, SAM spec 4.1) ----------------------------------------------------------------

// One encoder per thread, re-used from block to block.  The blocks' DEFLATE streams come from this repository's own encoder
// (deflate_fast.hpp: made for streams of BAM records -- 4.6 x zlib's rate a thread at 1.03 x its bytes on the bench's input); with
// RSEM_HIP_DEFLATE in the environment from zlib.
struct BgzfDeflater {
    z_stream zs;
    bool live = false, use_zlib = false, asked = false;
    int tune[4] = {0, 0, 0, 0}, nt = 0;
    std::unique_ptr<FastDeflate> fast;
    static constexpr size_t kBlock = 0xff00;  // input bytes per block, as htslib cuts them
    static_assert(kBlock <= FastDeflate::kMaxIn && FastDeflate::kMaxOut + 8 <= 0x10000 - 18 - 8, "a block fits its BGZF frame");
    ~BgzfDeflater() { if (live) deflateEnd(&zs); }
    size_t zlib_block(const uint8_t* p, size_t n, uint8_t* dst, size_t cap) {
        if (!live) {
            // zlib's level 6 (what the reference's samtools writes BAM with) walks hash chains of up to 128 candidates and stops at a
            // match of 128 bytes; a transcript BAM repeats a read's sequence and qualities in every one of its alignments' records
            // (the longest match sits at the head of the chain) and its qualities match nothing anywhere.  Chains of 16 and no early
            // stop (matches up to 258) deflate 1.6 x as fast AND 0.9 % smaller on the bench's input (30 -> 49 MB/s per thread, 524.8 ->
            // 520.1 MB at 2 % of configs[2]; profiles/r06m_*, r06n_*: level 4 is as fast and 0.3 % larger, level 1 2.5 x and 6 % larger).
            // RSEM_HIP_DEFLATE="level[,memLevel[,good,lazy,nice,chain]]"; "zlib" or "" = level 6 tuned as above; "6,8" = zlib's own level 6.
            int lv = Z_DEFAULT_COMPRESSION, ml = 8;
            nt = 6; tune[0] = 8; tune[1] = 16; tune[2] = 258; tune[3] = 16;
            const char* e = getenv("RSEM_HIP_DEFLATE");
            if (e && *e >= '0' && *e <= '9') nt = sscanf(e, "%d,%d,%d,%d,%d,%d", &lv, &ml, &tune[0], &tune[1], &tune[2], &tune[3]);
            if (deflateInit2(&zs, lv, Z_DEFLATED, -15, nt >= 2 ? ml : 8, Z_DEFAULT_STRATEGY) != Z_OK) die("zlib deflateInit2 failed");
            live = true;
        } else if (deflateReset(&zs) != Z_OK) die("zlib deflateReset failed");
        if (nt == 6) (void)deflateTune(&zs, tune[0], tune[1], tune[2], tune[3]);  // (a reset restores the level's own)
        zs.next_in = (Bytef*)p; zs.avail_in = (uInt)n;
        zs.next_out = dst; zs.avail_out = (uInt)cap;
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) die("BGZF block does not fit");
        return (size_t)zs.total_out;
    }
    // append the BGZF block of p[0..n), n <= kBlock, to out
    void block(const uint8_t* p, size_t n, std::vector<uint8_t>& out) {
        if (!asked) {
            asked = true;
            use_zlib = getenv("RSEM_HIP_DEFLATE") != nullptr;
            if (!use_zlib) fast.reset(new FastDeflate());
        }
        const size_t at = out.size();
        out.resize(at + 0x10000);
        uint8_t* o = out.data() + at;
        const uint32_t clen = (uint32_t)(use_zlib ? zlib_block(p, n, o + 18, 0x10000 - 18 - 8) : fast->compress(p, n, o + 18));
        static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        memcpy(o, hdr, 16);
        const uint16_t bsize = (uint16_t)(clen + 25);  // total block size - 1
        o[16] = bsize & 0xff; o[17] = bsize >> 8;
        const uint32_t crc = crc32_fast(0u, p, n);  // (crc32_fold.hpp: 18 GB/s against 0.9 for zlib's)
        const uint32_t isize = (uint32_t)n;
        memcpy(o + 18 + clen, &crc, 4);
        memcpy(o + 22 + clen, &isize, 4);
        out.resize(at + clen + 26);
    }
    // a run of blocks for p[0..n)
    void stream(const uint8_t* p, size_t n, std::vector<uint8_t>& out) {
        for (size_t i = 0; i < n; i += kBlock) block(p + i, std::min(kBlock, n - i), out);
    }
};
inline void bgzf_write_eof(FILE* f) {
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    fwrite(eof, 1, 28, f);
}

// N threads that take the items 0 .. n-1 of a stage as they become free; run() returns when the stage is done.
class StagePool {
   public:
    explicit StagePool(int n) : n_(std::max(1, n)) {
        for (int i = 1; i < n_; i++) th_.emplace_back([this, i] { loop(i); });
    }
    ~StagePool() {
        { std::lock_guard<std::mutex> g(m_); quit_ = true; ++gen_; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int threads() const { return n_; }
    void run(size_t n_items, const std::function<void(size_t, int)>& fn) {
        if (n_ == 1 || n_items <= 1) { for (size_t i = 0; i < n_items; i++) fn(i, 0); return; }
        { std::lock_guard<std::mutex> g(m_); fn_ = &fn; items_ = n_items; next_.store(0); busy_ = n_ - 1; ++gen_; }
        cv_.notify_all();
        work(0);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return busy_ == 0; });
        fn_ = nullptr;
    }

   private:
    int n_;
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t, int)>* fn_ = nullptr;
    size_t items_ = 0;
    std::atomic<size_t> next_{0};
    int busy_ = 0;
    unsigned long gen_ = 0;
    bool quit_ = false;
    void work(int me) { for (size_t i; (i = next_.fetch_add(1)) < items_;) (*fn_)(i, me); }
    void loop(int me) {
        unsigned long seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> g(m_); cv_.wait(g, [&] { return gen_ != seen; }); seen = gen_; if (quit_) return; }
            work(me);
            { std::lock_guard<std::mutex> g(m_); if (--busy_ == 0) done_.notify_one(); }
        }
    }
};

class BgzfReader {
   public:
    bool open(const std::string& path) { f_ = fopen(path.c_str(), "rb"); return f_ != nullptr; }
    ~BgzfReader() { if (f_) fclose(f_); }
    bool read(void* dst, size_t n) {  // false at (clean or unclean) end of data
        uint8_t* d = (uint8_t*)dst;
        while (n) {
            if (pos_ == blk_.size() && !fill()) return false;
            size_t k = std::min(n, blk_.size() - pos_);
            memcpy(d, blk_.data() + pos_, k);
            d += k; pos_ += k; n -= k; consumed_ += k;
        }
        return true;
    }
    uint64_t consumed() const { return consumed_; }  // bytes of the uncompressed stream handed out so far

   private:
    FILE* f_ = nullptr;
    std::vector<uint8_t> blk_;
    size_t pos_ = 0;
    uint64_t consumed_ = 0;
    bool fill() {
        for (;;) {
            uint8_t h[18];
            if (fread(h, 1, 18, f_) != 18) return false;
            if (h[0] != 0x1f || h[1] != 0x8b || !(h[3] & 4)) die("input BAM: not a BGZF block");
            const int xlen = h[10] | (h[11] << 8);
            std::vector<uint8_t> extra(xlen);
            memcpy(extra.data(), h + 12, std::min(6, xlen));
            if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, f_) != (size_t)(xlen - 6)) return false;
            int bsize = -1;
            for (int i = 0; i + 4 <= xlen;) {
                int slen = extra[i + 2] | (extra[i + 3] << 8);
                if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2) bsize = extra[i + 4] | (extra[i + 5] << 8);
                i += 4 + slen;
            }
            if (bsize < 0) die("input BAM: BGZF block without BC field");
            const int clen = bsize + 1 - 12 - xlen - 8;
            std::vector<uint8_t> comp(clen + 8);
            if (fread(comp.data(), 1, clen + 8, f_) != (size_t)(clen + 8)) return false;
            uint32_t isize;
            memcpy(&isize, comp.data() + clen + 4, 4);
            blk_.resize(isize);
            pos_ = 0;
            if (isize == 0) continue;  // empty (EOF marker) block
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) die("zlib inflateInit2 failed");
            zs.next_in = comp.data(); zs.avail_in = clen;
            zs.next_out = blk_.data(); zs.avail_out = isize;
            int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END) die("input BAM: corrupt BGZF block");
            return true;
        }
    }
};

// ---- alignment records ---------------------------------------------------------------------------------

struct AlnHeader {
    std::string text;
    std::vector<std::string> names;
    std::vector<int32_t> lens;
};

struct AlnRecord {  // BAM record without the leading block_size
    std::vector<uint8_t> d;
    int32_t refID() const { int32_t v; memcpy(&v, d.data(), 4); return v; }
    uint16_t flag() const { return (uint16_t)(d[14] | (d[15] << 8)); }
    bool mapped() const { return !(flag() & 4); }
    bool read1() const { return flag() & 64; }
};

// the header's @SQ lines -> reference dictionary (what sam_hdr_parse derives)
inline void parse_sq(AlnHeader& H) {
    H.names.clear(); H.lens.clear();
    size_t p = 0;
    while (p < H.text.size()) {
        size_t e = H.text.find('\n', p);
        if (e == std::string::npos) e = H.text.size();
        if (e - p > 3 && H.text.compare(p, 3, "@SQ") == 0) {
            std::string name; long len = 0;
            size_t q = p + 3;
            while (q < e) {
                size_t t = H.text.find('\t', q + 1);
                if (t == std::string::npos || t > e) t = e;
                if (t - q > 4 && H.text.compare(q + 1, 3, "SN:") == 0) name = H.text.substr(q + 4, t - q - 4);
                if (t - q > 4 && H.text.compare(q + 1, 3, "LN:") == 0) len = atol(H.text.c_str() + q + 4);
                q = t;
            }
            H.names.push_back(name);
            H.lens.push_back((int32_t)len);
        }
        p = e + 1;
    }
}

// SamHeader (SamHeader.cpp:60-108, SamHeader.hpp:37-62): regroup the lines and add @PG ID:RSEM once
inline std::string header_with_pg(const std::string& text) {
    std::string HD, SQ, RG, PG, CO, other;
    bool has_rsem = false;
    size_t p = 0;
    while (p < text.size()) {
        size_t e = text.find('\n', p);
        if (e == std::string::npos) e = text.size();
        std::string line = text.substr(p, e - p);
        p = e + 1;
        if (line.empty() || line[0] != '@') continue;
        std::string tag = line.substr(1, 2);
        if (tag == "HD") HD = line + "\n";
        else if (tag == "SQ") SQ += line + "\n";
        else if (tag == "RG") RG += line + "\n";
        else if (tag == "PG") {
            size_t q = 3;
            while (q < line.size()) {
                size_t t = line.find('\t', q + 1);
                if (t == std::string::npos) t = line.size();
                if (line.compare(q + 1, 3, "ID:") == 0 && line.substr(q + 4, t - q - 4) == "RSEM") has_rsem = true;
                q = t;
            }
            PG += line + "\n";
        } else if (tag == "CO") CO += line + "\n";
        else other += line;  // (sic: the reference drops the newline of unknown records, SamHeader.cpp:104)
    }
    if (!has_rsem) PG += "@PG\tID:RSEM\n";
    return HD + SQ + RG + PG + CO + other;
}

class AlignmentReader {
   public:
    AlnHeader header;
    void open(const std::string& path) {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) die("Cannot open %s! It may not exist.", path.c_str());
        int c0 = fgetc(f), c1 = fgetc(f);
        fclose(f);
        is_bam_ = (c0 == 0x1f && c1 == 0x8b);
        if (is_bam_) {
            if (!bgzf_.open(path)) die("Cannot open %s!", path.c_str());
            char magic[4];
            int32_t l_text, n_ref;
            if (!bgzf_.read(magic, 4) || memcmp(magic, "BAM\1", 4) != 0) die("%s is gzip-compressed but not a BAM file", path.c_str());
            bgzf_.read(&l_text, 4);
            header.text.resize(l_text);
            bgzf_.read(&header.text[0], l_text);
            while (!header.text.empty() && header.text.back() == '\0') header.text.pop_back();
            bgzf_.read(&n_ref, 4);
            for (int i = 0; i < n_ref; i++) {
                int32_t l_name, l_ref;
                bgzf_.read(&l_name, 4);
                std::string nm(l_name, '\0');
                bgzf_.read(&nm[0], l_name);
                nm.resize(strlen(nm.c_str()));
                bgzf_.read(&l_ref, 4);
                header.names.push_back(nm);
                header.lens.push_back(l_ref);
            }
            records_at_ = bgzf_.consumed();
        } else {
            if (!map_.open(path)) die("Cannot open %s!", path.c_str());
            p_ = map_.data;
            end_ = map_.data + map_.size;
            while (p_ < end_ && *p_ == '@') {
                const char* nl = (const char*)memchr(p_, '\n', end_ - p_);
                const char* e = nl ? nl : end_;
                header.text.append(p_, e - p_);
                header.text.push_back('\n');
                p_ = nl ? nl + 1 : end_;
            }
            parse_sq(header);
            for (size_t i = 0; i < header.names.size(); i++) name2id_.push_back({header.names[i], (int)i});
            std::sort(name2id_.begin(), name2id_.end());
        }
    }
    bool is_bam() const { return is_bam_; }
    // SAM input: the alignment lines (the mapped file behind the header)
    const char* body_begin() const { return p_; }
    const char* body_end() const { return end_; }
    // BAM input: where the first record starts in the uncompressed stream
    uint64_t records_at() const { return records_at_; }

   private:
    bool is_bam_ = false;
    BgzfReader bgzf_;
    MappedFile map_;
    const char *p_ = nullptr, *end_ = nullptr;
    uint64_t records_at_ = 0;
    std::vector<std::pair<std::string, int>> name2id_;

    int ref_id(const std::string& n) const {
        if (n == "*") return -1;
        auto it = std::lower_bound(name2id_.begin(), name2id_.end(), std::make_pair(n, -1));
        if (it == name2id_.end() || it->first != n) die("SAM record refers to unknown reference %s", n.c_str());
        return it->second;
    }
    int ref_id_of(const char* n, size_t len) const {  // (the same look-up on a view of the line)
        if (len == 1 && *n == '*') return -1;
        size_t lo = 0, hi = name2id_.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            const std::string& k = name2id_[mid].first;
            const size_t m = std::min(k.size(), len);
            int c = memcmp(k.data(), n, m);
            if (c == 0) c = k.size() < len ? -1 : (k.size() > len ? 1 : 0);
            if (c < 0) lo = mid + 1; else hi = mid;
        }
        if (lo == name2id_.size() || name2id_[lo].first.size() != len || memcmp(name2id_[lo].first.data(), n, len) != 0)
            die("SAM record refers to unknown reference %s", std::string(n, len).c_str());
        return name2id_[lo].second;
    }
    struct Nt16Table {  // base letter -> BAM's 4-bit code ("=ACMGRSVTWYHKDBN", either case; anything else N)
        uint8_t code[256];
        Nt16Table() {
            static const char* nt16 = "=ACMGRSVTWYHKDBN";
            for (int c = 0; c < 256; c++) code[c] = 15;
            for (int k = 0; k < 16; k++) { code[(unsigned char)nt16[k]] = (uint8_t)k; code[(unsigned char)tolower((unsigned char)nt16[k])] = (uint8_t)k; }
        }
    };
    static int reg2bin(int64_t beg, int64_t end) {  // hts_reg2bin(beg, end, 14, 5)
        int l, s = 14, t = ((1 << 15) - 1) / 7;
        for (--end, l = 5; l > 0; --l, s += 3, t -= 1 << (l * 3))
            if (beg >> s == end >> s) return t + (int)(beg >> s);
        return 0;
    }
    template <typename T>
    static void put(std::vector<uint8_t>& d, T v) { const uint8_t* p = (const uint8_t*)&v; d.insert(d.end(), p, p + sizeof(T)); }

   public:
    // one SAM text line -> BAM record bytes (SAM spec 4.2; integer tags take the smallest fitting type as htslib does); any thread
    void encode_sam_line(const char* b, const char* e, AlnRecord& r) const {
        // (no heap traffic for the eleven fixed fields -- at 10 % of configs[2] a SAM input is 112 M lines, and strings, vectors and a
        // strchr per base were most of the pass's CPU time there: the fields stay views into the line, numbers are read in place, the
        // bases go through a table)
        static thread_local std::vector<std::pair<const char*, const char*>> f;
        f.clear();
        for (const char* q = b;;) {
            const char* t = (const char*)memchr(q, '\t', e - q);
            f.push_back({q, t ? t : e});
            if (!t) break;
            q = t + 1;
        }
        if (f.size() < 11) die("SAM line with fewer than 11 fields");
        auto len_of = [&](int i) -> size_t { return (size_t)(f[i].second - f[i].first); };
        auto num = [&](int i) -> long long {  // (as atol: optional sign, digits, stops at the first other character)
            const char* q = f[i].first;
            const char* qe = f[i].second;
            bool neg = false;
            if (q < qe && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
            long long v = 0;
            while (q < qe && *q >= '0' && *q <= '9') v = v * 10 + (*q++ - '0');
            return neg ? -v : v;
        };
        auto is_str = [&](int i, const char* lit) -> bool { const size_t n = strlen(lit); return len_of(i) == n && !memcmp(f[i].first, lit, n); };
        const int flag = (int)num(1), mapq = (int)num(4);
        const int32_t pos = (int32_t)num(3) - 1, pnext = (int32_t)num(7) - 1, tlen = (int32_t)num(8);
        const int32_t tid = ref_id_of(f[2].first, len_of(2));
        const int32_t mtid = is_str(6, "=") ? tid : ref_id_of(f[6].first, len_of(6));
        static thread_local std::vector<uint32_t> cig;
        cig.clear();
        int64_t rlen = 0;
        if (!is_str(5, "*")) {
            static const char* ops = "MIDNSHP=X";
            for (const char *c = f[5].first, *ce = f[5].second; c < ce;) {
                uint32_t n = 0;
                while (c < ce && *c >= '0' && *c <= '9') n = n * 10 + (uint32_t)(*c++ - '0');
                const char* o = c < ce ? strchr(ops, *c++) : nullptr;
                if (!o || !*o) die("SAM record with unknown CIGAR operation");
                const int op = (int)(o - ops);
                cig.push_back(n << 4 | (uint32_t)op);
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += n;
            }
        }
        const int64_t endp = (cig.empty() || (flag & 4)) ? (int64_t)pos + 1 : (int64_t)pos + rlen;
        const char* seq = f[9].first;
        const char* qual = f[10].first;
        const int l_seq = is_str(9, "*") ? 0 : (int)len_of(9);
        const size_t l_name = len_of(0);
        std::vector<uint8_t>& d = r.d;
        d.resize(32 + l_name + 1 + 4 * cig.size() + (size_t)(l_seq + 1) / 2 + (size_t)l_seq);
        uint8_t* o = d.data();
        auto w32 = [&](int32_t v) { memcpy(o, &v, 4); o += 4; };
        auto w16 = [&](uint16_t v) { memcpy(o, &v, 2); o += 2; };
        w32(tid); w32(pos);
        *o++ = (uint8_t)(l_name + 1); *o++ = (uint8_t)mapq;
        w16((uint16_t)reg2bin(pos, endp));
        w16((uint16_t)cig.size()); w16((uint16_t)flag);
        w32(l_seq); w32(mtid); w32(pnext); w32(tlen);
        memcpy(o, f[0].first, l_name); o += l_name; *o++ = 0;
        if (!cig.empty()) { memcpy(o, cig.data(), 4 * cig.size()); o += 4 * cig.size(); }
        static const Nt16Table nt16;
        for (int i = 0; i + 1 < l_seq; i += 2) *o++ = (uint8_t)(nt16.code[(unsigned char)seq[i]] << 4 | nt16.code[(unsigned char)seq[i + 1]]);
        if (l_seq & 1) *o++ = (uint8_t)(nt16.code[(unsigned char)seq[l_seq - 1]] << 4);
        if (is_str(10, "*")) { memset(o, 0xff, (size_t)l_seq); o += l_seq; }
        else for (int i = 0; i < l_seq; i++) *o++ = (uint8_t)(qual[i] - 33);
        for (size_t k = 11; k < f.size(); k++) {
            const char* t = f[k].first;
            const size_t n = f[k].second - t;
            if (n < 5 || t[2] != ':' || t[4] != ':') continue;
            d.push_back(t[0]); d.push_back(t[1]);
            const char type = t[3];
            const std::string v(t + 5, n - 5);
            if (type == 'A') { d.push_back('A'); d.push_back(v.empty() ? 0 : v[0]); }
            else if (type == 'i') {
                long long x = atoll(v.c_str());
                if (x < 0) {
                    if (x >= -128) { d.push_back('c'); put<int8_t>(d, (int8_t)x); }
                    else if (x >= -32768) { d.push_back('s'); put<int16_t>(d, (int16_t)x); }
                    else { d.push_back('i'); put<int32_t>(d, (int32_t)x); }
                } else {
                    if (x <= 255) { d.push_back('C'); put<uint8_t>(d, (uint8_t)x); }
                    else if (x <= 65535) { d.push_back('S'); put<uint16_t>(d, (uint16_t)x); }
                    else { d.push_back('I'); put<uint32_t>(d, (uint32_t)x); }
                }
            } else if (type == 'f') { d.push_back('f'); put<float>(d, (float)atof(v.c_str())); }
            else if (type == 'Z' || type == 'H') { d.push_back(type); d.insert(d.end(), v.begin(), v.end()); d.push_back(0); }
            else if (type == 'B') {
                d.push_back('B');
                const char sub = v.empty() ? 'c' : v[0];
                d.push_back(sub);
                std::vector<std::string> vals;
                for (size_t i = 1; i < v.size();) {
                    size_t c = v.find(',', i + 1);
                    if (c == std::string::npos) c = v.size();
                    if (v[i] == ',') vals.push_back(v.substr(i + 1, c - i - 1));
                    i = c;
                }
                put<int32_t>(d, (int32_t)vals.size());
                for (auto& s : vals) {
                    switch (sub) {
                        case 'c': put<int8_t>(d, (int8_t)atoi(s.c_str())); break;
                        case 'C': put<uint8_t>(d, (uint8_t)atoi(s.c_str())); break;
                        case 's': put<int16_t>(d, (int16_t)atoi(s.c_str())); break;
                        case 'S': put<uint16_t>(d, (uint16_t)atoi(s.c_str())); break;
                        case 'i': put<int32_t>(d, (int32_t)atoll(s.c_str())); break;
                        case 'I': put<uint32_t>(d, (uint32_t)atoll(s.c_str())); break;
                        default: put<float>(d, (float)atof(s.c_str())); break;
                    }
                }
            } else d.resize(d.size() - 2);  // unknown type: drop the tag
        }
    }
};

// bam_prb_to_mapq (sam_utils.h:72-76)
inline uint8_t prb_to_mapq(double val) {
    double err = 1.0 - val;
    if (err <= 1e-10) return 100;
    return (uint8_t)(-10 * log10(err) + .5);
}

// BamWriter::set_alignment_weight (BamWriter.h:39-48): MAPQ + ZW:f
inline void set_alignment_weight(AlnRecord& r, double prb) {
    std::vector<uint8_t>& d = r.d;
    d[9] = prb_to_mapq(prb);
    const float val = (float)prb;
    uint16_t n_cig;
    int32_t l_seq;
    memcpy(&n_cig, d.data() + 12, 2);
    memcpy(&l_seq, d.data() + 16, 4);
    size_t p = 32 + d[8] + 4 * (size_t)n_cig + (size_t)(l_seq + 1) / 2 + (size_t)l_seq;
    while (p + 3 <= d.size()) {  // walk the auxiliary fields
        const bool zw = d[p] == 'Z' && d[p + 1] == 'W';
        const char type = (char)d[p + 2];
        if (zw) { memcpy(d.data() + p + 3, &val, 4); return; }
        p += 3;
        switch (type) {
            case 'A': case 'c': case 'C': p += 1; break;
            case 's': case 'S': p += 2; break;
            case 'i': case 'I': case 'f': p += 4; break;
            case 'd': p += 8; break;
            case 'Z': case 'H': while (p < d.size() && d[p]) ++p; ++p; break;
            case 'B': {
                const char sub = (char)d[p];
                int32_t n;
                memcpy(&n, d.data() + p + 1, 4);
                const int sz = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                p += 5 + (size_t)n * sz;
                break;
            }
            default: die("input alignment record carries an auxiliary field of unknown type '%c'", type);
        }
    }
    d.push_back('Z'); d.push_back('W'); d.push_back('f');
    const uint8_t* v = (const uint8_t*)&val;
    d.insert(d.end(), v, v + 4);
}

// BamWriter::work (BamWriter.h:82-146).  weights: one per alignment (per mate pair for paired-end data), in .dat order.
//
// The input is taken in super-chunks of ~256 MB (RSEM_HIP_BAM_CHUNK bytes; the tests use small ones), each in three stages on
// `nthreads` threads (StagePool), a barrier between stages:
//   SAM input: the super-chunk's lines are cut into pieces at line boundaries; (A) count the records of every piece -- piece
//     boundaries then move by one line where a mate pair would be split; (B) count the alignments (pairs) of every piece that
//     carry a weight, whence the index of every piece's first weight; (C) encode the piece's lines as BAM records, set MAPQ and
//     ZW:f, deflate the piece's records as a run of BGZF blocks of its own.
//   BAM input: (A) inflate the super-chunk's BGZF blocks side by side into one buffer (a record cut by the end of the buffer is
//     carried over to the next super-chunk); the records are framed by one thread (four bytes per record) and cut into pieces;
//     (B), (C) as above, on copies of the records.
// The pieces' block runs are written in order.  Output: the same records in the same order as the reference writes; the BGZF
// block boundaries differ from htslib's (a run of blocks per piece instead of one per 0xff00 bytes of the whole stream).
inline void write_transcript_bam(const std::string& inpF, const std::string& outF, bool paired, const int32_t* hit_sid,
                                 const double* weights, uint64_t n_hits, const Transcripts& T, int nthreads = 1) {
    AlignmentReader in;
    in.open(inpF);
    // external (header order) -> internal sid, Transcripts::buildMappings (Transcripts.h:96-144)
    std::vector<std::pair<std::string, int>> dict;
    for (int i = 1; i <= T.M; i++) dict.push_back({T.type == 2 ? T.t[i].seqname : T.t[i].transcript_id, i});
    std::sort(dict.begin(), dict.end());
    std::vector<int> e2i(in.header.names.size(), 0);
    for (size_t k = 0; k < in.header.names.size(); k++) {
        auto it = std::lower_bound(dict.begin(), dict.end(), std::make_pair(in.header.names[k], -1));
        if (it == dict.end() || it->first != in.header.names[k]) die("RSEM can not recognize reference sequence name %s!", in.header.names[k].c_str());
        e2i[k] = it->second;
    }
    nthreads = std::max(1, std::min(nthreads, (int)std::max(1u, std::thread::hardware_concurrency())));
    if (const int granted = cgroup_cpu_cores()) nthreads = std::max(1, std::min(nthreads, granted));  // (no more threads than the cgroup grants cores)
    StagePool pool(nthreads);
    std::vector<BgzfDeflater> defl(nthreads);
    FILE* fo = fopen(outF.c_str(), "wb");
    if (!fo) die("Cannot open %s for writing!", outF.c_str());
    {  // header
        AlnHeader out_h;
        out_h.text = header_with_pg(in.header.text);
        parse_sq(out_h);
        std::vector<uint8_t> raw, comp;
        auto put = [&](const void* q, size_t n) { raw.insert(raw.end(), (const uint8_t*)q, (const uint8_t*)q + n); };
        put("BAM\1", 4);
        const int32_t l_text = (int32_t)out_h.text.size();
        put(&l_text, 4);
        put(out_h.text.data(), l_text);
        const int32_t n_ref = (int32_t)out_h.names.size();
        put(&n_ref, 4);
        for (int i = 0; i < n_ref; i++) {
            const int32_t l_name = (int32_t)out_h.names[i].size() + 1;
            put(&l_name, 4);
            put(out_h.names[i].c_str(), l_name);
            put(&out_h.lens[i], 4);
        }
        defl[0].stream(raw.data(), raw.size(), comp);
        fwrite(comp.data(), 1, comp.size(), fo);
    }
    size_t super_bytes = 256u << 20;
    if (const char* e = getenv("RSEM_HIP_BAM_CHUNK")) super_bytes = std::max<size_t>(64, (size_t)atoll(e));
    const size_t n_pieces_max = (size_t)nthreads * 4;

    // what stages B and C do with a record (SE) or a pair of records (PE), shared by both kinds of input
    auto emit = [](std::vector<uint8_t>& raw, const AlnRecord& r) {
        const int32_t bs = (int32_t)r.d.size();
        raw.insert(raw.end(), (const uint8_t*)&bs, (const uint8_t*)&bs + 4);
        raw.insert(raw.end(), r.d.begin(), r.d.end());
    };
    auto finish_unit = [&](std::vector<uint8_t>& raw, AlnRecord& a, AlnRecord* b, uint64_t& h) {  // BamWriter.h:101-141
        if (!paired) {
            if (a.mapped()) {
                if (h >= n_hits) die("The alignment file holds more alignments than the .dat file!");
                if (e2i[a.refID()] != hit_sid[h]) die("The alignment file and the .dat file are out of step!");
                set_alignment_weight(a, weights[h++]);
            }
            emit(raw, a);
        } else {
            AlnRecord* r1 = &a;
            AlnRecord* r2 = b;
            if (!r1->read1()) std::swap(r1, r2);
            if (r1->mapped() && r2->mapped()) {
                if (h >= n_hits) die("The alignment file holds more alignments than the .dat file!");
                if (e2i[r1->refID()] != hit_sid[h] || e2i[r2->refID()] != hit_sid[h]) die("The alignment file and the .dat file are out of step!");
                set_alignment_weight(*r1, weights[h]);
                set_alignment_weight(*r2, weights[h]);
                ++h;
            }
            emit(raw, *r1);
            emit(raw, *r2);
        }
    };
    struct Piece {
        size_t b = 0, e = 0;          // SAM: byte range of the lines; BAM: range of record indices
        uint64_t records = 0, units = 0, h0 = 0;
        std::vector<uint8_t> out;
    };
    std::vector<Piece> pieces;
    uint64_t h_total = 0;
    // Where the pass's time goes (RSEM_HIP_TIMING: one line at the end).  Wall clock of the stages as the main thread sees them, and
    // -- summed over the threads -- the seconds and bytes of the three things a thread does: inflate, encode / copy + weigh, deflate.
    struct PassClock {
        double wall[5] = {0, 0, 0, 0, 0};  // A (count / inflate), frame, B, C, waiting for the writer
        std::atomic<uint64_t> ns_inflate{0}, ns_encode{0}, ns_deflate{0}, b_inflated{0}, b_encoded{0}, b_deflated{0};
        double write_s = 0, frame_thread_s = 0;
        std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
        void lap(int i) { const auto n = std::chrono::steady_clock::now(); wall[i] += std::chrono::duration<double>(n - t).count(); t = n; }
    } clk;
    auto ns_since = [](std::chrono::steady_clock::time_point t0) {
        return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    };
    // The pieces' block runs are written in order by a thread of their own while the next super-chunk is counted, encoded and
    // deflated (until round 6 the main thread wrote them between two super-chunks with every worker idle).
    std::thread writer;
    std::vector<Piece> writing;
    auto write_pieces = [&]() {
        if (writer.joinable()) writer.join();
        clk.lap(4);
        writing.swap(pieces);
        writer = std::thread([&]() {
            const auto t0 = std::chrono::steady_clock::now();
            for (Piece& P : writing) {
                if (!P.out.empty() && fwrite(P.out.data(), 1, P.out.size(), fo) != P.out.size()) die("Cannot write %s!", outF.c_str());
                std::vector<uint8_t>().swap(P.out);
            }
            clk.write_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
    };
    const int per_unit = paired ? 2 : 1;

    if (!in.is_bam()) {
        const char* cur = in.body_begin();
        const char* const end = in.body_end();
        auto next_line = [](const char*& q, const char* lim, const char*& b, const char*& e) -> bool {  // the next non-empty line of [q, lim)
            while (q < lim) {
                const char* nl = (const char*)memchr(q, '\n', lim - q);
                b = q;
                e = nl ? nl : lim;
                q = nl ? nl + 1 : lim;
                if (e > b && e[-1] == '\r') --e;
                if (e > b) return true;
            }
            return false;
        };
        auto line_end_after = [&](const char* q) -> const char* {  // first line start at or behind q
            if (q >= end) return end;
            if (q == in.body_begin() || q[-1] == '\n') return q;
            const char* nl = (const char*)memchr(q, '\n', end - q);
            return nl ? nl + 1 : end;
        };
        while (cur < end) {
            const char* sc_end = line_end_after(std::min(end, cur + super_bytes));
            // pieces at line boundaries
            const size_t total = (size_t)(sc_end - cur);
            const size_t np = std::max<size_t>(1, std::min(n_pieces_max, total / 4096 + 1));
            pieces.assign(np, Piece());
            for (size_t i = 0; i < np; i++) {
                const char* b = i == 0 ? cur : line_end_after(cur + total * i / np);
                pieces[i].b = (size_t)(std::min(b, sc_end) - in.body_begin());
            }
            for (size_t i = 0; i < np; i++) pieces[i].e = i + 1 < np ? pieces[i + 1].b : (size_t)(sc_end - in.body_begin());
            // (A) records per piece
            pool.run(np, [&](size_t i, int) {
                const char *q = in.body_begin() + pieces[i].b, *lim = in.body_begin() + pieces[i].e, *b, *e;
                uint64_t n = 0;
                while (next_line(q, lim, b, e)) ++n;
                pieces[i].records = n;
            });
            clk.lap(0);
            if (paired) {  // no pair may straddle two pieces: a piece that would start with a second mate gives that line to its predecessor
                const char* const base = in.body_begin();
                uint64_t before = 0;
                long last = -1;  // the last piece so far that holds records
                for (size_t i = 0; i < np; i++) {
                    if (pieces[i].records == 0) continue;
                    if (before & 1) {  // (then last >= 0)
                        const char *q = base + pieces[i].b, *lim = base + pieces[i].e, *lb, *le;
                        next_line(q, lim, lb, le);
                        const size_t nb = (size_t)(q - base);
                        pieces[last].e = nb;
                        pieces[last].records += 1;
                        for (size_t z = (size_t)last + 1; z < i; z++) pieces[z].b = pieces[z].e = nb;
                        pieces[i].b = nb;
                        pieces[i].records -= 1;
                        before += 1;
                        if (pieces[i].records == 0) continue;
                    }
                    before += pieces[i].records;
                    last = (long)i;
                }
                if (before & 1) {  // the super-chunk ends inside a pair: take the second mate's line in, or drop a last record without a mate
                    const char *q = sc_end, *lb, *le;
                    if (next_line(q, end, lb, le)) {
                        sc_end = q;
                        const size_t nb = (size_t)(q - base);
                        pieces[last].e = nb;
                        pieces[last].records += 1;
                        for (size_t z = (size_t)last + 1; z < np; z++) pieces[z].b = pieces[z].e = nb;
                    } else {  // (`while (in.next(a) && in.next(b))` of the reference: it is not written)
                        Piece& P = pieces[last];
                        const char *q2 = base + P.b, *lim = base + P.e, *last_start = q2;
                        while (next_line(q2, lim, lb, le)) last_start = lb;
                        P.e = (size_t)(last_start - base);
                        P.records -= 1;
                        sc_end = end;
                    }
                }
            }
            // (B) weighted units per piece
            pool.run(np, [&](size_t i, int) {
                const char *q = in.body_begin() + pieces[i].b, *lim = in.body_begin() + pieces[i].e, *b, *e;
                auto flag_of = [](const char* lb, const char* le) -> int {
                    const char* t = (const char*)memchr(lb, '\t', le - lb);
                    return t ? atoi(t + 1) : 4;
                };
                uint64_t u = 0;
                while (next_line(q, lim, b, e)) {
                    const int f1 = flag_of(b, e);
                    if (!paired) { u += !(f1 & 4); continue; }
                    const char *b2, *e2;
                    if (!next_line(q, lim, b2, e2)) break;
                    u += !(f1 & 4) && !(flag_of(b2, e2) & 4);
                }
                pieces[i].units = u;
            });
            for (size_t i = 0; i < np; i++) { pieces[i].h0 = h_total; h_total += pieces[i].units; }
            clk.lap(2);
            // (C) encode, weigh, deflate
            pool.run(np, [&](size_t i, int me) {
                Piece& P = pieces[i];
                const char *q = in.body_begin() + P.b, *lim = in.body_begin() + P.e, *b, *e;
                std::vector<uint8_t> raw;
                raw.reserve((P.e - P.b) / 2 + 64);
                AlnRecord ra, rb;
                uint64_t h = P.h0;
                const auto t_enc = std::chrono::steady_clock::now();
                while (next_line(q, lim, b, e)) {
                    in.encode_sam_line(b, e, ra);
                    if (paired) {
                        const char *b2, *e2;
                        if (!next_line(q, lim, b2, e2)) break;
                        in.encode_sam_line(b2, e2, rb);
                    }
                    finish_unit(raw, ra, &rb, h);
                }
                if (h != P.h0 + P.units) die("internal error: a piece of the alignment file weighed %llu alignments, counted %llu", (unsigned long long)(h - P.h0), (unsigned long long)P.units);
                clk.ns_encode += ns_since(t_enc);
                clk.b_encoded += raw.size();
                const auto t_def = std::chrono::steady_clock::now();
                defl[me].stream(raw.data(), raw.size(), P.out);
                clk.ns_deflate += ns_since(t_def);
                clk.b_deflated += P.out.size();
            });
            clk.lap(3);
            write_pieces();
            cur = sc_end;
        }
    } else {
        // the BGZF blocks of the file
        MappedFile mf;
        if (!mf.open(inpF)) die("Cannot open %s!", inpF.c_str());
        struct Blk { size_t off, clen; uint32_t isize, crc; };
        std::vector<Blk> blks;
        for (size_t o = 0; o + 18 <= mf.size;) {
            const uint8_t* h = (const uint8_t*)mf.data + o;
            if (h[0] != 0x1f || h[1] != 0x8b || !(h[3] & 4)) die("input BAM: not a BGZF block");
            const int xlen = h[10] | (h[11] << 8);
            int bsize = -1;
            for (int i = 0; i + 4 <= xlen;) {
                const uint8_t* x = h + 12 + i;
                const int slen = x[2] | (x[3] << 8);
                if (x[0] == 'B' && x[1] == 'C' && slen == 2) bsize = x[4] | (x[5] << 8);
                i += 4 + slen;
            }
            if (bsize < 0 || o + (size_t)bsize + 1 > mf.size) die("input BAM: truncated or corrupt BGZF block");
            uint32_t isize, crc;
            memcpy(&isize, h + bsize + 1 - 4, 4);
            memcpy(&crc, h + bsize + 1 - 8, 4);
            blks.push_back({o + 12 + (size_t)xlen, (size_t)bsize + 1 - 12 - xlen - 8, isize, crc});
            o += (size_t)bsize + 1;
        }
        // A super-chunk goes through: (A) inflate -- on the pool; frame -- one thread walks the records' length words, four bytes per
        // ~200: a chain of cache misses, 22 s of a 59-second pass at 10 % of configs[2] when it stood between (A) and (B)
        // (profiles/r06f_e2e_bam.log); (B), (C) -- on the pool; write -- the writer thread.  Since round 6 the NEXT super-chunk is
        // inflated before, and framed WHILE, this one is weighed and deflated: two buffers, the framer on a thread of its own.
        struct RawBuf {  // (no value-initialisation: a std::vector's resize zero-filled 23 GB over the pass)
            uint8_t* p = nullptr;
            size_t cap = 0, n = 0;
            ~RawBuf() { free(p); }
            void size_to(size_t want, size_t keep) {
                if (want > cap) {
                    const size_t nc = want + want / 8 + 4096;
                    uint8_t* q = (uint8_t*)malloc(nc);
                    if (!q) die("out of memory (%zu bytes for a super-chunk of the alignment file)", nc);
                    if (keep) memcpy(q, p, keep);
                    free(p);
                    p = q;
                    cap = nc;
                }
                n = want;
            }
        };
        struct Chunk {
            RawBuf buf;
            std::vector<uint64_t> rec;  // offsets of the records' block_size words in buf
            size_t usable = 0;
            bool last = false;
        } ch[2];
        std::vector<uint8_t> carry;
        uint64_t skip = in.records_at();  // header bytes of the uncompressed stream still to pass
        size_t bi = 0;
        std::atomic<uint64_t> fast_blocks{0}, slow_blocks{0};
        const bool no_fast_inflate = getenv("RSEM_HIP_INFLATE_ZLIB") != nullptr;  // (measurement / tests: zlib for every block)
        // (A): the carried bytes of the chunk before, then blocks bi .. as far as a super-chunk goes
        auto inflate_chunk = [&](Chunk& C) {
            size_t be = bi, bytes = 0;
            while (be < blks.size() && (be == bi || bytes + blks[be].isize <= super_bytes)) bytes += blks[be++].isize;
            std::vector<size_t> at(be - bi + 1, carry.size());
            for (size_t k = bi; k < be; k++) at[k - bi + 1] = at[k - bi] + blks[k].isize;
            C.buf.size_to(at.back(), 0);
            if (!carry.empty()) memcpy(C.buf.p, carry.data(), carry.size());
            const size_t b0 = bi;
            pool.run(be - bi, [&](size_t k, int) {
                const Blk& B = blks[b0 + k];
                if (!B.isize) return;
                const auto t_inf = std::chrono::steady_clock::now();
                // this repository's decoder first (inflate_fast.hpp: 1.6 x zlib's rate) -- believed only if the block's own CRC-32
                // agrees with what came out; anything else (its `false`, a checksum that differs) goes to zlib, whose verdict stands
                static thread_local std::unique_ptr<FastInflate> fi;
                if (!fi) fi.reset(new FastInflate());
                uint8_t* dst = C.buf.p + at[k];
                bool done = !no_fast_inflate && fi->inflate((const uint8_t*)mf.data + B.off, B.clen, dst, B.isize) && crc32_fast(0u, dst, B.isize) == B.crc;
                if (done) fast_blocks.fetch_add(1, std::memory_order_relaxed);
                if (!done) {
                    z_stream zs;
                    memset(&zs, 0, sizeof(zs));
                    if (inflateInit2(&zs, -15) != Z_OK) die("zlib inflateInit2 failed");
                    zs.next_in = (Bytef*)mf.data + B.off; zs.avail_in = (uInt)B.clen;
                    zs.next_out = dst; zs.avail_out = B.isize;
                    const int rc = inflate(&zs, Z_FINISH);
                    inflateEnd(&zs);
                    if (rc != Z_STREAM_END) die("input BAM: corrupt BGZF block");
                    if (crc32_fast(0u, dst, B.isize) != B.crc) die("input BAM: a BGZF block's checksum does not match its data");  // (as htslib's reader)
                    slow_blocks.fetch_add(1, std::memory_order_relaxed);
                }
                clk.ns_inflate += ns_since(t_inf);
                clk.b_inflated += B.isize;
            });
            bi = be;
            C.last = bi >= blks.size();
        };
        // frame: the records of the chunk, what is left over for the next one.  One walk over the length words is a chain of cache
        // misses (2 GB/s: 11.3 s beside the stages at 10 % of configs[2], which the pass cannot get below once deflate is faster,
        // profiles/r06g_e2e_bam_10pct_bam_input.log), so the chunk is cut into segments that are walked at the same time: segment 0 from
        // the known start, every other one from a GUESSED record start -- the first offset in the segment at which a chain of
        // plausible record headers begins (lengths that fit each other, a reference id the header knows, a printable NUL-terminated
        // name).  A guess is never trusted: the walks are stitched in order, and a segment's records are taken only if the walk
        // before it (itself verified) arrives exactly at the segment's first record; otherwise that stretch is walked again from
        // the verified position (never seen on real files; the tests force it).
        double frame_s = 0.0;
        const int32_t n_ref_in = (int32_t)in.header.names.size();
        int frame_threads = std::max(1, std::min(16, nthreads / 4));
        if (const char* e = getenv("RSEM_HIP_BAM_FRAME_THREADS")) frame_threads = std::max(1, atoi(e));  // (tests)
        uint64_t seg_taken = 0, seg_again = 0;  // guessed segments whose records were taken / stretches walked again
        size_t frame_seg_bytes = (size_t)4 << 20;  // a segment is worth a thread from here on (tests: RSEM_HIP_BAM_FRAME_SEG)
        if (const char* e = getenv("RSEM_HIP_BAM_FRAME_SEG")) frame_seg_bytes = std::max<size_t>(1, (size_t)atoll(e));
        const bool frame_force_bad = getenv("RSEM_HIP_BAM_BAD_GUESS") != nullptr;  // tests: every guess a record late, or none
        auto plausible = [n_ref_in](const uint8_t* buf, size_t n, size_t p, size_t& next) -> bool {
            if (p + 36 > n) return false;
            int32_t bs, refid, pos, lseq;
            memcpy(&bs, buf + p, 4);
            if (bs < 32 || bs > (1 << 28)) return false;
            memcpy(&refid, buf + p + 4, 4);
            memcpy(&pos, buf + p + 8, 4);
            memcpy(&lseq, buf + p + 20, 4);
            const unsigned lname = buf[p + 12], ncig = (unsigned)buf[p + 16] | ((unsigned)buf[p + 17] << 8);
            if (refid < -1 || refid >= n_ref_in || pos < -1 || lname < 1 || lseq < 0) return false;
            if (32ull + lname + 4ull * ncig + ((uint64_t)lseq + 1) / 2 + (uint64_t)lseq > (uint64_t)bs) return false;
            if (p + 36 + lname > n) return false;  // (only what can be checked whole is plausible)
            if (buf[p + 36 + lname - 1] != 0) return false;
            for (unsigned i = 0; i + 1 < lname; i++) { const uint8_t ch = buf[p + 36 + i]; if (ch < 33 || ch > 126) return false; }
            next = p + 4 + (size_t)bs;
            return true;
        };
        // the records from `pos` on whose length words lie before `lim` (and that are complete in the chunk); returns where it stopped
        auto walk = [](const uint8_t* buf, size_t n, size_t pos, size_t lim, std::vector<uint64_t>& rec, bool& cut) -> size_t {
            cut = false;
            while (pos < lim && pos + 4 <= n) {
                int32_t bs;
                memcpy(&bs, buf + pos, 4);
                if (bs < 32) die("input BAM: corrupt alignment record");
                if (pos + 4 + (size_t)bs > n) { cut = true; break; }
                __builtin_prefetch(buf + pos + 4 + (size_t)bs + 1024);  // (a record's length word lies ~3 lines behind the last)
                rec.push_back(pos);
                pos += 4 + (size_t)bs;
            }
            return pos;
        };
        auto frame_chunk = [&](Chunk& C) {
            const auto t0 = std::chrono::steady_clock::now();
            const uint8_t* buf = C.buf.p;
            const size_t n = C.buf.n;
            size_t pos = 0;
            if (skip) { const size_t k = (size_t)std::min<uint64_t>(skip, n); pos = k; skip -= k; }
            C.rec.clear();
            const size_t span = n > pos ? n - pos : 0;
            const int K = (int)std::max<size_t>(1, std::min<size_t>((size_t)frame_threads, span / frame_seg_bytes + 1));
            if (K == 1 && !frame_force_bad) {
                bool cut;
                pos = walk(buf, n, pos, n, C.rec, cut);
            } else {
                const int KK = std::max(K, frame_force_bad ? 3 : 1);
                struct Seg { size_t lo = 0, first = 0, stop = 0; bool found = false, cut = false; std::vector<uint64_t> rec; };
                std::vector<Seg> seg((size_t)KK);
                for (int k = 0; k < KK; k++) seg[k].lo = pos + span * (size_t)k / (size_t)KK;
                auto run_seg = [&](int k) {
                    Seg& S = seg[k];
                    const size_t lim = k + 1 < KK ? seg[k + 1].lo : n;
                    if (k == 0) { S.first = S.lo; S.found = true; }
                    else {
                        for (size_t q = S.lo; q < lim && !S.found; q++) {
                            size_t a = q, nx;
                            int ok = 0;
                            while (ok < 6 && plausible(buf, n, a, nx)) { ++ok; a = nx; if (a + 36 > n) break; }
                            if (ok >= 6 || (ok >= 1 && a + 36 > n)) { S.first = q; S.found = true; }
                        }
                        if (S.found && frame_force_bad) {  // (tests) a wrong guess: the record behind the first one / no guess at all
                            int32_t bs;
                            memcpy(&bs, buf + S.first, 4);
                            S.first += 4 + (size_t)bs;
                            if ((k & 1) == 0 || S.first >= lim) S.found = false;
                        }
                    }
                    if (S.found) S.stop = walk(buf, n, S.first, lim, S.rec, S.cut);
                };
                {
                    std::vector<std::thread> th;
                    for (int k = 1; k < KK; k++) th.emplace_back(run_seg, k);
                    run_seg(0);
                    for (auto& t : th) t.join();
                }
                // stitch: `pos` = the verified position; a segment is taken if its first record is where the verified walk arrived
                bool cut = false;
                for (int k = 0; k < KK && !cut; k++) {
                    Seg& S = seg[k];
                    const size_t lim = k + 1 < KK ? seg[k + 1].lo : n;
                    if (S.found && S.first == pos && !S.rec.empty()) {
                        C.rec.insert(C.rec.end(), S.rec.begin(), S.rec.end());
                        pos = S.stop;
                        cut = S.cut;
                        seg_taken += k > 0;
                    } else if (pos < lim) {  // no guess, a wrong one, or the walk before ran past it: this stretch again, from what is known
                        pos = walk(buf, n, pos, lim, C.rec, cut);
                        seg_again += 1;
                    }
                }
            }
            size_t usable = C.rec.size();
            size_t carry_from = pos;  // the first byte behind the complete records
            if (paired && (usable & 1)) { --usable; carry_from = C.rec[usable]; }  // keep pairs together: the odd record waits for its mate
            if (C.last) {
                if (pos != n) die("input BAM: the last record is cut short");
                carry.clear();  // (a last record without a mate is not written: `while (in.next(a) && in.next(b))` of the reference)
            } else {
                carry.assign(buf + carry_from, buf + n);
            }
            C.usable = usable;
            frame_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        };
        int cur = 0;
        if (!blks.empty()) {
            inflate_chunk(ch[0]);
            clk.lap(0);
            frame_chunk(ch[0]);
            clk.lap(1);
        }
        for (bool more = !blks.empty(); more;) {
            Chunk& C = ch[cur];
            Chunk& N = ch[cur ^ 1];
            const bool have_next = !C.last;
            std::thread framer;
            if (have_next) {
                inflate_chunk(N);  // (its carried bytes are those frame_chunk(C) left)
                clk.lap(0);
                framer = std::thread([&]() { frame_chunk(N); });
            }
            const uint8_t* buf = C.buf.p;
            const std::vector<uint64_t>& rec = C.rec;
            const size_t units_all = C.usable / per_unit;
            const size_t np = std::max<size_t>(1, std::min(n_pieces_max, units_all / 64 + 1));
            pieces.assign(np, Piece());
            for (size_t i = 0; i < np; i++) { pieces[i].b = units_all * i / np * per_unit; pieces[i].e = units_all * (i + 1) / np * per_unit; }
            auto load = [&](size_t r, AlnRecord& out) {
                int32_t bs;
                memcpy(&bs, buf + rec[r], 4);
                out.d.assign(buf + rec[r] + 4, buf + rec[r] + 4 + bs);
            };
            auto flag_at = [&](size_t r) -> int { const uint8_t* d = buf + rec[r] + 4; return d[14] | (d[15] << 8); };
            // (B)
            pool.run(np, [&](size_t i, int) {
                uint64_t u = 0;
                for (size_t r = pieces[i].b; r < pieces[i].e; r += per_unit)
                    u += paired ? (!(flag_at(r) & 4) && !(flag_at(r + 1) & 4)) : !(flag_at(r) & 4);
                pieces[i].units = u;
            });
            for (size_t i = 0; i < np; i++) { pieces[i].h0 = h_total; h_total += pieces[i].units; }
            clk.lap(2);
            // (C)
            pool.run(np, [&](size_t i, int me) {
                Piece& P = pieces[i];
                std::vector<uint8_t> raw;
                AlnRecord ra, rb;
                uint64_t h = P.h0;
                const auto t_enc = std::chrono::steady_clock::now();
                for (size_t r = P.b; r < P.e; r += per_unit) {
                    load(r, ra);
                    if (paired) load(r + 1, rb);
                    finish_unit(raw, ra, &rb, h);
                }
                clk.ns_encode += ns_since(t_enc);
                clk.b_encoded += raw.size();
                const auto t_def = std::chrono::steady_clock::now();
                defl[me].stream(raw.data(), raw.size(), P.out);
                clk.ns_deflate += ns_since(t_def);
                clk.b_deflated += P.out.size();
            });
            clk.lap(3);
            write_pieces();
            if (framer.joinable()) framer.join();
            clk.lap(1);  // (what the framer took beyond (B) + (C))
            more = have_next;
            cur ^= 1;
        }
        clk.frame_thread_s = frame_s;
        if (getenv("RSEM_HIP_TIMING")) printf("[timing]   framing: %d walks at once, %llu guessed segments taken, %llu stretches walked again; blocks inflated by inflate_fast.hpp %llu, by zlib %llu\n", frame_threads, (unsigned long long)seg_taken, (unsigned long long)seg_again, (unsigned long long)fast_blocks.load(), (unsigned long long)slow_blocks.load());
    }
    if (writer.joinable()) writer.join();
    clk.lap(4);
    if (h_total != n_hits) die("The alignment file holds %s alignments (%llu) than the .dat file (%llu)!", h_total < n_hits ? "fewer" : "more", (unsigned long long)h_total, (unsigned long long)n_hits);
    if (getenv("RSEM_HIP_TIMING")) {
        auto rate = [](uint64_t bytes, uint64_t ns) { return ns ? (double)bytes / 1e6 / ((double)ns * 1e-9) : 0.0; };
        printf("[timing]   transcript.bam pass, %d threads: stages (wall) %s %.2f s | waiting for the framer %.2f (framing itself %.2f s beside the stages) | count weights %.2f | %s + deflate %.2f | waiting for the writer %.2f"
               " (writing itself %.2f s beside them); per thread: inflate %.0f MB/s (%.1f GB), %s %.0f MB/s (%.1f GB of records), deflate %.0f MB/s in -> %.1f GB out\n",
               nthreads, in.is_bam() ? "inflate" : "count lines", clk.wall[0], clk.wall[1], clk.frame_thread_s, clk.wall[2], in.is_bam() ? "copy + weigh" : "encode + weigh", clk.wall[3],
               clk.wall[4], clk.write_s, rate(clk.b_inflated, clk.ns_inflate), (double)clk.b_inflated / 1e9, in.is_bam() ? "copy + weigh" : "encode + weigh",
               rate(clk.b_encoded, clk.ns_encode), (double)clk.b_encoded / 1e9, rate(clk.b_encoded, clk.ns_deflate), (double)clk.b_deflated / 1e9);
    }
    bgzf_write_eof(fo);
    if (fclose(fo) != 0) die("Cannot write %s!", outF.c_str());
}

}  // namespace rsemh
