// bam_io.hpp -- the `-b` option of rsem-run-em (BamWriter.h:39-146, sam_utils.h:72-76, SamHeader.hpp): copy the
// input alignments to <sample>.transcript.bam with MAPQ and a ZW:f tag set from each alignment's posterior
// weight.  The reference links htslib for this; here the two formats involved are implemented directly on
// zlib: a BGZF reader/writer, a BAM record walker, and a SAM-text -> BAM record encoder (SAM spec v1).
#pragma once
#include <zlib.h>

#include "files.hpp"

namespace rsemh {

// ---- BGZF (blocked gzip, SAM spec 4.1) ----------------------------------------------------------------

class BgzfWriter {
   public:
    bool open(const std::string& path) { f_ = fopen(path.c_str(), "wb"); buf_.reserve(kBlock); return f_ != nullptr; }
    void write(const void* p, size_t n) {
        const uint8_t* s = (const uint8_t*)p;
        while (n) {
            size_t k = std::min(n, (size_t)kBlock - buf_.size());
            buf_.insert(buf_.end(), s, s + k);
            s += k; n -= k;
            if (buf_.size() == kBlock) flush();
        }
    }
    void close() {
        flush();
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        fwrite(eof, 1, 28, f_);
        fclose(f_);
        f_ = nullptr;
    }

   private:
    static constexpr int kBlock = 0xff00;
    FILE* f_ = nullptr;
    std::vector<uint8_t> buf_;
    void flush() {
        if (buf_.empty()) return;
        uint8_t out[0x10000];
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("zlib deflateInit2 failed");
        zs.next_in = buf_.data(); zs.avail_in = (uInt)buf_.size();
        zs.next_out = out + 18; zs.avail_out = sizeof(out) - 18 - 8;
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) die("BGZF block does not fit");
        const uint32_t clen = (uint32_t)zs.total_out;
        deflateEnd(&zs);
        const uint8_t hdr[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
        memcpy(out, hdr, 12);
        out[12] = 'B'; out[13] = 'C'; out[14] = 2; out[15] = 0;
        const uint16_t bsize = (uint16_t)(clen + 25);  // total block size - 1
        out[16] = bsize & 0xff; out[17] = bsize >> 8;
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf_.data(), (uInt)buf_.size());
        const uint32_t isize = (uint32_t)buf_.size();
        memcpy(out + 18 + clen, &crc, 4);
        memcpy(out + 22 + clen, &isize, 4);
        fwrite(out, 1, clen + 26, f_);
        buf_.clear();
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
            d += k; pos_ += k; n -= k;
        }
        return true;
    }

   private:
    FILE* f_ = nullptr;
    std::vector<uint8_t> blk_;
    size_t pos_ = 0;
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
    bool next(AlnRecord& r) {
        if (is_bam_) {
            int32_t bs;
            if (!bgzf_.read(&bs, 4)) return false;
            r.d.resize(bs);
            return bgzf_.read(r.d.data(), bs);
        }
        while (p_ < end_) {
            const char* nl = (const char*)memchr(p_, '\n', end_ - p_);
            const char* e = nl ? nl : end_;
            const char* b = p_;
            p_ = nl ? nl + 1 : end_;
            if (e > b && e[-1] == '\r') --e;
            if (e == b) continue;
            encode_sam_line(b, e, r);
            return true;
        }
        return false;
    }

   private:
    bool is_bam_ = false;
    BgzfReader bgzf_;
    MappedFile map_;
    const char *p_ = nullptr, *end_ = nullptr;
    std::vector<std::pair<std::string, int>> name2id_;

    int ref_id(const std::string& n) const {
        if (n == "*") return -1;
        auto it = std::lower_bound(name2id_.begin(), name2id_.end(), std::make_pair(n, -1));
        if (it == name2id_.end() || it->first != n) die("SAM record refers to unknown reference %s", n.c_str());
        return it->second;
    }
    static int reg2bin(int64_t beg, int64_t end) {  // hts_reg2bin(beg, end, 14, 5)
        int l, s = 14, t = ((1 << 15) - 1) / 7;
        for (--end, l = 5; l > 0; --l, s += 3, t -= 1 << (l * 3))
            if (beg >> s == end >> s) return t + (int)(beg >> s);
        return 0;
    }
    template <typename T>
    static void put(std::vector<uint8_t>& d, T v) { const uint8_t* p = (const uint8_t*)&v; d.insert(d.end(), p, p + sizeof(T)); }

    // one SAM text line -> BAM record bytes (SAM spec 4.2; integer tags take the smallest fitting type as htslib does)
    void encode_sam_line(const char* b, const char* e, AlnRecord& r) {
        std::vector<std::pair<const char*, const char*>> f;
        for (const char* q = b;;) {
            const char* t = (const char*)memchr(q, '\t', e - q);
            f.push_back({q, t ? t : e});
            if (!t) break;
            q = t + 1;
        }
        if (f.size() < 11) die("SAM line with fewer than 11 fields");
        auto str = [&](int i) { return std::string(f[i].first, f[i].second - f[i].first); };
        const std::string qname = str(0), rname = str(2), cigar = str(5), rnext = str(6), seq = str(9), qual = str(10);
        const int flag = atoi(str(1).c_str()), mapq = atoi(str(4).c_str());
        const int32_t pos = (int32_t)atol(str(3).c_str()) - 1, pnext = (int32_t)atol(str(7).c_str()) - 1, tlen = (int32_t)atol(str(8).c_str());
        const int32_t tid = ref_id(rname);
        const int32_t mtid = rnext == "=" ? tid : ref_id(rnext);
        std::vector<uint32_t> cig;
        int64_t rlen = 0;
        if (cigar != "*") {
            static const char* ops = "MIDNSHP=X";
            for (size_t i = 0; i < cigar.size();) {
                uint32_t n = 0;
                while (i < cigar.size() && isdigit((unsigned char)cigar[i])) n = n * 10 + (cigar[i++] - '0');
                const char* o = strchr(ops, cigar[i++]);
                if (!o) die("SAM record with unknown CIGAR operation");
                const int op = (int)(o - ops);
                cig.push_back(n << 4 | op);
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += n;
            }
        }
        const int64_t endp = (cig.empty() || (flag & 4)) ? (int64_t)pos + 1 : (int64_t)pos + rlen;
        const int l_seq = seq == "*" ? 0 : (int)seq.size();
        std::vector<uint8_t>& d = r.d;
        d.clear();
        put<int32_t>(d, tid); put<int32_t>(d, pos);
        d.push_back((uint8_t)(qname.size() + 1)); d.push_back((uint8_t)mapq);
        put<uint16_t>(d, (uint16_t)reg2bin(pos, endp));
        put<uint16_t>(d, (uint16_t)cig.size()); put<uint16_t>(d, (uint16_t)flag);
        put<int32_t>(d, l_seq); put<int32_t>(d, mtid); put<int32_t>(d, pnext); put<int32_t>(d, tlen);
        d.insert(d.end(), qname.begin(), qname.end()); d.push_back(0);
        for (uint32_t c : cig) put<uint32_t>(d, c);
        static const char* nt16 = "=ACMGRSVTWYHKDBN";
        for (int i = 0; i < l_seq; i += 2) {
            auto code = [&](char c) { const char* o = strchr(nt16, toupper((unsigned char)c)); return o && c ? (int)(o - nt16) : 15; };
            d.push_back((uint8_t)(code(seq[i]) << 4 | (i + 1 < l_seq ? code(seq[i + 1]) : 0)));
        }
        if (qual == "*") d.insert(d.end(), l_seq, 0xff);
        else for (int i = 0; i < l_seq; i++) d.push_back((uint8_t)(qual[i] - 33));
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
inline void write_transcript_bam(const std::string& inpF, const std::string& outF, bool paired, const int32_t* hit_sid,
                                 const double* weights, uint64_t n_hits, const Transcripts& T) {
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
    AlnHeader out_h;
    out_h.text = header_with_pg(in.header.text);
    parse_sq(out_h);
    BgzfWriter out;
    if (!out.open(outF)) die("Cannot open %s for writing!", outF.c_str());
    out.write("BAM\1", 4);
    const int32_t l_text = (int32_t)out_h.text.size();
    out.write(&l_text, 4);
    out.write(out_h.text.data(), l_text);
    const int32_t n_ref = (int32_t)out_h.names.size();
    out.write(&n_ref, 4);
    for (int i = 0; i < n_ref; i++) {
        const int32_t l_name = (int32_t)out_h.names[i].size() + 1;
        out.write(&l_name, 4);
        out.write(out_h.names[i].c_str(), l_name);
        out.write(&out_h.lens[i], 4);
    }
    auto emit = [&](const AlnRecord& r) {
        const int32_t bs = (int32_t)r.d.size();
        out.write(&bs, 4);
        out.write(r.d.data(), r.d.size());
    };
    uint64_t h = 0;
    AlnRecord a, b;
    if (!paired) {
        while (in.next(a)) {
            if (a.mapped()) {
                if (h >= n_hits) die("The alignment file holds more alignments than the .dat file!");
                if (e2i[a.refID()] != hit_sid[h]) die("The alignment file and the .dat file are out of step!");
                set_alignment_weight(a, weights[h++]);
            }
            emit(a);
        }
    } else {
        while (in.next(a) && in.next(b)) {
            AlnRecord* r1 = &a;
            AlnRecord* r2 = &b;
            if (!r1->read1()) std::swap(r1, r2);
            if (r1->mapped() && r2->mapped()) {
                if (h >= n_hits) die("The alignment file holds more alignments than the .dat file!");
                if (e2i[r1->refID()] != hit_sid[h] || e2i[r2->refID()] != hit_sid[h]) die("The alignment file and the .dat file are out of step!");
                set_alignment_weight(*r1, weights[h]);
                set_alignment_weight(*r2, weights[h]);
                ++h;
            }
            emit(*r1);
            emit(*r2);
        }
    }
    if (h != n_hits) die("The alignment file holds fewer alignments (%llu) than the .dat file (%llu)!", (unsigned long long)h, (unsigned long long)n_hits);
    out.close();
}

}  // namespace rsemh
