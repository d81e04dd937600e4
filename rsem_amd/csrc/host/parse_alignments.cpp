// rsem-parse-alignments -- drop-in for the reference's alignment parser (parseIt.cpp:1-260, SamParser.h:28-330):
// SAM/BAM in, the EM stage's input files out, byte-identical to the reference's:
//
//   rsem-parse-alignments refName imdName statName alignF read_type [-t fai_file] [-tag tagName] [-q] [-p threads]
//                         [--binary [--text]]        (not in the reference; also RSEM_HIP_BINARY=1 / =both in the environment;
//                                                     =1 there keeps the alignable read files as text for the driver's
//                                                     rsem-build-read-index step, --binary alone writes no text at all)
//
//   imdName.dat                        N1 nHits read_type / one line of hits per alignable read
//   imdName_{un,alignable,max}[_1|_2].{fa,fq}   reads by category (empty categories are removed)
//   imdName.omit                       transcripts the alignment header does not declare
//   statName.cnt                       N0 N1 N2 / nUnique nMulti nIsoMulti / nHits read_type / hits-per-read histogram
//
// --binary writes imdName.rsb/ INSTEAD of imdName.dat and the read files: the same alignments and reads as the arrays
// rsem-run-em uploads (host/rsb.hpp), so that the EM stage maps them instead of parsing decimal text (the reference's
// text round trip: parseIt.cpp:139-147 -> HitContainer.h:63-79, SingleReadQ.h:38-60); --text keeps the text files as
// well.  .cnt and .omit are always written.
//
// Host-only stage (byte shuffling, no GPU work); it feeds the EM hot path.  The reference
// walks the file on one thread through htslib.  Here the decoded stream is cut into waves; inside a wave the records
// are indexed, cut into chunks at read boundaries (a record starts a new read when it is unaligned or its name
// differs from the previous record's, SamParser.h:128-131) and the chunks run the reference's state machine
// independently; their outputs are appended in file order.  BAM: the BGZF blocks of a wave are inflated in
// parallel (zlib only, no htslib).  `-t` is accepted and ignored (it only matters for CRAM).
#include <zlib.h>

#include <atomic>
#include <charconv>
#include <map>
#include <memory>
#include <unordered_map>

#include "files.hpp"
#include "reads.hpp"
#include "rsb.hpp"

using namespace rsemh;

namespace {

// ---- one alignment record, decoded from SAM text or BAM bytes to the few fields the parser needs -------------------

struct Rec {
    const char* name = nullptr;  // canonical name: first whitespace-delimited word of QNAME (sam_utils.h:54-61)
    uint32_t name_len = 0;
    int flag = 0, tid = -1, pos = 0, l_seq = 0;
    bool cigar_ok = false;       // one M/=/X operation covering the whole read (sam_utils.h:64-70)
    bool is_bam = false;
    const uint8_t* seq = nullptr;   // SAM: characters; BAM: packed 4-bit codes
    const uint8_t* qual = nullptr;  // SAM: phred+33 characters or nullptr for "*"; BAM: raw phred
    long long tag = 0;              // value of the -tag field when it is an integer field, else 0 (bam_aux2i)
    bool paired() const { return flag & 1; }
    bool mapped() const { return !(flag & 4); }
    bool rev() const { return flag & 16; }
    bool read1() const { return flag & 64; }
    bool read2() const { return flag & 128; }
    bool same_name(const Rec& o) const { return name_len == o.name_len && memcmp(name, o.name, name_len) == 0; }
    std::string name_str() const { return std::string(name, name_len); }
};

struct Config {
    int read_type = 0;
    bool paired = false, has_q = false;
    std::string rt_tag;  // "" = no filtered-read tag
    std::unordered_map<std::string, int> tid_of;  // SAM RNAME -> header index
    std::vector<int32_t> e2i, target_len, gid_of;
    int n_targets = 0;
    bool want_text = true, want_bin = false;
    // RSEM_HIP_BINARY=1 in the ENVIRONMENT (the unmodified Perl driver built the command line): binary hand-off, but the
    // alignable read files are still written as text -- the driver's next command is the reference's rsem-build-read-index
    // on exactly those files (rsem-calculate-expression:597-604), which would end the run without them
    bool keep_alignable = false;
    bool reads_text(int category) const { return want_text || (keep_alignable && category == 1); }
};

struct ParseError { std::string msg; };
[[noreturn]] void fail(const std::string& m) { throw ParseError{m}; }

uint8_t g_nt16[256];       // character -> 4-bit code, as htslib's seq_nt16_table
char g_fwd[16], g_rc[16];  // code -> base as sequenced (sam_utils.h:78-112); 0 = not allowed
void init_tables() {
    memset(g_nt16, 15, sizeof(g_nt16));
    const char* nt = "=ACMGRSVTWYHKDBN";
    for (int i = 0; i < 16; i++) { g_nt16[(uint8_t)nt[i]] = (uint8_t)i; g_nt16[(uint8_t)tolower(nt[i])] = (uint8_t)i; }
    g_nt16['0'] = 1; g_nt16['1'] = 2; g_nt16['2'] = 4; g_nt16['3'] = 8;
    memset(g_fwd, 0, 16); memset(g_rc, 0, 16);
    g_fwd[1] = 'A'; g_fwd[2] = 'C'; g_fwd[4] = 'G'; g_fwd[8] = 'T'; g_fwd[15] = 'N';
    g_rc[1] = 'T'; g_rc[2] = 'G'; g_rc[4] = 'C'; g_rc[8] = 'A'; g_rc[15] = 'N';
}

void set_name(Rec& r, const char* q, size_t n) {
    size_t k = 0;
    while (k < n && !(q[k] == ' ' || (q[k] >= '\t' && q[k] <= '\r'))) ++k;  // " \t\n\r\f\v"
    r.name = q; r.name_len = (uint32_t)k;
}

// SAM text line [b, e) without the newline (SAM spec 1.4)
void decode_sam(const char* b, const char* e, const Config& cfg, Rec& r) {
    if (e > b && e[-1] == '\r') --e;
    const char* f[12];
    int nf = 0;
    f[nf++] = b;
    for (const char* q = b; nf < 12;) {
        const char* t = (const char*)memchr(q, '\t', e - q);
        if (!t) break;
        f[nf++] = q = t + 1;
    }
    if (nf < 11) fail("SAM line with fewer than 11 fields: " + std::string(b, std::min<size_t>(e - b, 60)));
    auto fend = [&](int i) { return i + 1 < nf ? f[i + 1] - 1 : e; };
    r = Rec();
    set_name(r, f[0], fend(0) - f[0]);
    r.flag = (int)strtol(f[1], nullptr, 0);
    const size_t rn = fend(2) - f[2];
    if (rn == 1 && f[2][0] == '*') r.tid = -1;
    else {
        auto it = cfg.tid_of.find(std::string(f[2], rn));
        r.tid = it == cfg.tid_of.end() ? -1 : it->second;
    }
    r.pos = (int)strtol(f[3], nullptr, 10) - 1;
    const char* sq = f[9];
    const size_t ls = fend(9) - sq;
    r.l_seq = (ls == 1 && sq[0] == '*') ? 0 : (int)ls;
    r.seq = (const uint8_t*)sq;
    const char* ql = f[10];
    const size_t lq = fend(10) - ql;
    r.qual = (lq == 1 && ql[0] == '*' && r.l_seq != 1) ? nullptr : (const uint8_t*)ql;
    {   // CIGAR: ^[0-9]+[M=X]$ with the read's length
        const char* c = f[5];
        const char* ce = fend(5);
        long n = 0;
        const char* q = c;
        while (q < ce && *q >= '0' && *q <= '9') n = n * 10 + (*q++ - '0');
        r.cigar_ok = q > c && q + 1 == ce && (*q == 'M' || *q == '=' || *q == 'X') && n == r.l_seq;
    }
    if (!cfg.rt_tag.empty() && nf == 12) {
        for (const char* q = f[11]; q < e;) {
            const char* t = (const char*)memchr(q, '\t', e - q);
            const char* qe = t ? t : e;
            if (qe - q >= 5 && q[0] == cfg.rt_tag[0] && q[1] == cfg.rt_tag[1] && q[2] == ':') {
                if (q[3] == 'i') r.tag = strtoll(q + 5, nullptr, 10);
                break;
            }
            if (!t) break;
            q = t + 1;
        }
    }
}

// BAM record body [d, d+n) (after block_size; SAM spec 4.2)
void decode_bam(const uint8_t* d, size_t n, const Config& cfg, Rec& r) {
    if (n < 32) fail("truncated BAM record");
    r = Rec();
    r.is_bam = true;
    int32_t v;
    memcpy(&v, d, 4); r.tid = v;
    memcpy(&v, d + 4, 4); r.pos = v;
    const int l_name = d[8];
    const int n_cigar = d[12] | (d[13] << 8);
    r.flag = d[14] | (d[15] << 8);
    memcpy(&v, d + 16, 4); r.l_seq = v;
    const uint8_t* p = d + 32;
    if (r.l_seq < 0 || (size_t)32 + l_name + 4 * (size_t)n_cigar + ((size_t)r.l_seq + 1) / 2 + (size_t)r.l_seq > n) fail("corrupt BAM record");
    set_name(r, (const char*)p, l_name ? strnlen((const char*)p, l_name) : 0);
    p += l_name;
    if (n_cigar >= 1) {
        uint32_t c;
        memcpy(&c, p, 4);
        const int op = c & 15;
        r.cigar_ok = n_cigar == 1 && (op == 0 || op == 7 || op == 8) && (int32_t)(c >> 4) == r.l_seq;
    }
    p += 4 * (size_t)n_cigar;
    r.seq = p;
    p += (r.l_seq + 1) / 2;
    r.qual = p;
    p += r.l_seq;
    if (!cfg.rt_tag.empty()) {
        const uint8_t* e = d + n;
        while (p + 3 <= e) {
            const bool hit = p[0] == (uint8_t)cfg.rt_tag[0] && p[1] == (uint8_t)cfg.rt_tag[1];
            const char type = (char)p[2];
            p += 3;
            size_t sz = 0;
            long long val = 0;
            switch (type) {
                case 'A': sz = 1; break;
                case 'c': val = (int8_t)p[0]; sz = 1; break;
                case 'C': val = p[0]; sz = 1; break;
                case 's': { int16_t x; memcpy(&x, p, 2); val = x; sz = 2; break; }
                case 'S': { uint16_t x; memcpy(&x, p, 2); val = x; sz = 2; break; }
                case 'i': { int32_t x; memcpy(&x, p, 4); val = x; sz = 4; break; }
                case 'I': { uint32_t x; memcpy(&x, p, 4); val = x; sz = 4; break; }
                case 'f': sz = 4; break;
                case 'd': sz = 8; break;
                case 'Z': case 'H': sz = strnlen((const char*)p, e - p) + 1; break;
                case 'B': {
                    if (p + 5 > e) { sz = e - p; break; }
                    const char sub = (char)p[0];
                    int32_t cnt; memcpy(&cnt, p + 1, 4);
                    const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                    sz = 5 + w * (size_t)cnt;
                    break;
                }
                default: sz = e - p; break;  // malformed: stop
            }
            if (hit) { r.tag = val; break; }
            p += sz;
        }
    }
}

void append_seq(std::string& out, const Rec& r) {  // sam_utils.h:78-112
    const int L = r.l_seq;
    const size_t o = out.size();
    out.resize(o + L);
    char* w = &out[o];
    const bool rev = r.rev();
    const char* tab = rev ? g_rc : g_fwd;
    for (int i = 0; i < L; i++) {
        const int k = rev ? L - 1 - i : i;
        const int code = r.is_bam ? (r.seq[k >> 1] >> ((~k & 1) << 2)) & 15 : g_nt16[r.seq[k]];
        const char b = tab[code];
        if (!b) fail("Read " + r.name_str() + ": its sequence has a base other than A, C, G, T, N!");
        w[i] = b;
    }
}

void append_qual(std::string& out, const Rec& r) {  // sam_utils.h:114-132
    const int L = r.l_seq;
    const size_t o = out.size();
    out.resize(o + L);
    char* w = &out[o];
    if (!r.qual) { memset(w, (char)(0xff + 33), L); return; }  // "*": htslib stores 0xff
    const int add = r.is_bam ? 33 : 0;
    if (r.rev()) for (int i = 0; i < L; i++) w[i] = (char)(r.qual[L - 1 - i] + add);
    else for (int i = 0; i < L; i++) w[i] = (char)(r.qual[i] + add);
}

inline void append_int(std::string& s, long long v) {
    char tmp[24];
    auto res = std::to_chars(tmp, tmp + sizeof(tmp), v);
    s.append(tmp, res.ptr - tmp);
}

// ---- what a chunk of units produces -----------------------------------------------------------------------------------

struct ChunkOut {
    std::string dat, reads[3][2];
    // --binary: the same content as arrays (host/rsb.hpp)
    std::vector<uint32_t> b_rowlen, b_len[3][2];
    std::vector<int32_t> b_sid, b_pos, b_ins;
    std::vector<uint8_t> b_seq[3][2], b_qual[3][2];
    long long N[3] = {0, 0, 0}, nHits = 0, nMulti = 0, nIsoMulti = 0, units = 0;
    std::map<long long, long long> counter;
    std::vector<std::string> warns;
    long long n_warns = 0;
    std::string error;
};

// A wave: decoded bytes + the offsets of its records
struct Wave {
    const char* base = nullptr;
    bool is_bam = false;
    std::vector<size_t> beg, end;  // record i = [beg[i], end[i])  (SAM: line without '\n'; BAM: record body)
    void decode(size_t i, const Config& cfg, Rec& r) const {
        if (is_bam) decode_bam((const uint8_t*)base + beg[i], end[i] - beg[i], cfg, r);
        else decode_sam(base + beg[i], base + end[i], cfg, r);
    }
};

// unit = one record (single-end) or two adjacent records (paired-end, mate 1 first after the swap of SamParser.h:195)
struct Unit { Rec a, b; int rt; };

void load_unit(const Wave& w, size_t u, const Config& cfg, Unit& U) {
    if (!cfg.paired) {
        w.decode(u, cfg, U.a);
        U.rt = U.a.mapped() ? 1 : (!cfg.rt_tag.empty() && U.a.tag > 0 ? 2 : 0);  // SamParser.h:61-67
    } else {
        w.decode(2 * u, cfg, U.a);
        w.decode(2 * u + 1, cfg, U.b);
        if (!U.a.read1()) std::swap(U.a, U.b);
        if (U.a.mapped() && U.b.mapped()) U.rt = 1;  // SamParser.h:70-82
        else U.rt = (!cfg.rt_tag.empty() && (U.a.tag > 0 || U.b.tag > 0)) ? 2 : 0;
    }
}

// the reference's loop (parseIt.cpp:75-152 + SamParser::parseNext) over units [u0, u1), which begin at a read boundary
void run_chunk(const Wave& w, size_t u0, size_t u1, const Config& cfg, ChunkOut& out) {
    struct Mate { std::string name, seq, qual; int len = 0; } cur[2];
    int cur_val = -2;
    std::string hits;
    std::vector<int32_t> gids, h_sid, h_pos, h_ins;
    const int n_os = cfg.paired ? 2 : 1;

    auto flush = [&]() {  // parseIt.cpp:92-118
        if (cur_val >= 0) {
            for (int j = 0; j < n_os; j++) {  // SingleRead.h:52-55, SingleReadQ.h:57-60
                if (cfg.reads_text(cur_val)) {
                    std::string& o = out.reads[cur_val][j];
                    o.push_back(cfg.has_q ? '@' : '>');
                    o += cur[j].name; o.push_back('\n');
                    o += cur[j].seq; o.push_back('\n');
                    if (cfg.has_q) { o += "+\n"; o += cur[j].qual; o.push_back('\n'); }
                }
                if (cfg.want_bin) {  // what rsem-run-em would decode from the text: get_base_id (utils.h:36-50), c2q (QProfile.h:44)
                    const int8_t* tbl = base_table();
                    const std::string& sq = cur[j].seq;
                    std::vector<uint8_t>& bs = out.b_seq[cur_val][j];
                    const size_t o0 = bs.size();
                    bs.resize(o0 + sq.size());
                    for (size_t i = 0; i < sq.size(); i++) { const int8_t id = tbl[(unsigned char)sq[i]]; bs[o0 + i] = id < 0 ? 255 : (uint8_t)id; }
                    if (cfg.has_q) {
                        const std::string& ql = cur[j].qual;
                        std::vector<uint8_t>& bq = out.b_qual[cur_val][j];
                        bq.resize(o0 + sq.size());
                        for (size_t i = 0; i < sq.size(); i++) bq[o0 + i] = i < ql.size() ? (uint8_t)((unsigned char)ql[i] - 33) : 255;
                    }
                    out.b_len[cur_val][j].push_back((uint32_t)sq.size());
                }
            }
            ++out.N[cur_val];
        }
        if (cur_val != 1 && !gids.empty()) fail("Read " + cur[0].name + " is both unalignable and alignable according to the input file!");
        if (cur_val == 1) {
            const long long k = (long long)gids.size();
            out.nHits += k;
            std::sort(gids.begin(), gids.end());
            if (std::unique(gids.begin(), gids.end()) - gids.begin() > 1) ++out.nMulti;  // HitContainer.h:94-108
            if (k > 1) ++out.nIsoMulti;
            if (k > 0 && cfg.want_text) { append_int(out.dat, k); out.dat += hits; out.dat.push_back('\n'); }
            if (k > 0 && cfg.want_bin) {
                out.b_rowlen.push_back((uint32_t)k);
                out.b_sid.insert(out.b_sid.end(), h_sid.begin(), h_sid.end());
                out.b_pos.insert(out.b_pos.end(), h_pos.begin(), h_pos.end());
                if (cfg.paired) out.b_ins.insert(out.b_ins.end(), h_ins.begin(), h_ins.end());
            }
            ++out.counter[k];
        }
        hits.clear();
        gids.clear();
        h_sid.clear(); h_pos.clear(); h_ins.clear();
    };

    Unit U;
    try {
        for (size_t u = u0; u < u1; u++) {
            load_unit(w, u, cfg, U);
            const Rec& a = U.a;
            const Rec& b = U.b;
            std::string name = a.name_str(), name2;
            if (!cfg.paired) {
                if (a.paired()) fail("Read " + name + ": Find a paired end read in the file!");
            } else {
                if (!(a.paired() && b.paired())) fail("Read " + name + ": One of the mate is not paired-end! (RSEM assumes the two mates of a paired-end read should be adjacent)");
                if (!(a.read1() && b.read2())) fail("Read " + name + ": The adjacent two lines do not represent the two mates of a paired-end read! (RSEM assumes the two mates of a paired-end read should be adjacent)");
                if (a.mapped() != b.mapped()) fail("Read " + name + ": RSEM currently does not support partial alignments!");
                name2 = b.name_str();
                if (name != name2 && ++out.n_warns <= 50)
                    out.warns.push_back("Warning: Detected a read pair whose two mates have different names--" + name + " and " + name2 + "!");
            }
            const int rt = U.rt;
            if (rt != 1 || cur[0].name != name) {  // a new read (SamParser.h:128-131)
                flush();
                cur_val = rt;
                cur[0].name.swap(name);
                cur[0].len = a.l_seq;
                cur[0].seq.clear(); append_seq(cur[0].seq, a);
                if (cfg.has_q) { cur[0].qual.clear(); append_qual(cur[0].qual, a); }
                if (cfg.paired) {
                    cur[1].name.swap(name2);
                    cur[1].len = b.l_seq;
                    cur[1].seq.clear(); append_seq(cur[1].seq, b);
                    if (cfg.has_q) { cur[1].qual.clear(); append_qual(cur[1].qual, b); }
                }
            } else if (!cfg.paired) {
                if (cur[0].len != a.l_seq) fail("Read " + cur[0].name + " has alignments with inconsistent read lengths!");
            } else if (cur[0].len != a.l_seq || cur[1].len != b.l_seq)
                fail("Paired-end read " + cur[0].name + " has alignments with inconsistent mate lengths!");
            if (rt == 1) {
                const std::string& nm = cur[0].name;
                if (!a.cigar_ok || (cfg.paired && !b.cigar_ok)) fail("Read " + nm + ": RSEM currently does not support gapped alignments, sorry!\n");
                if (cfg.paired && a.tid != b.tid) fail("Read " + nm + ": The two mates do not align to a same transcript! RSEM does not support discordant alignments.");
                if (a.tid < 0 || a.tid >= cfg.n_targets) fail("Read " + nm + ": aligned to a reference sequence the header does not declare!");
                const int sid = cfg.e2i[a.tid];
                const int tlen = cfg.target_len[a.tid];
                // SamParser.h:136-141, 218-223
                const int h_s = a.rev() ? -sid : sid;
                const int h_p = a.rev() ? tlen - a.pos - a.l_seq : a.pos;
                const int h_i = !cfg.paired ? 0 : (a.rev() ? a.pos + a.l_seq - b.pos : b.pos + b.l_seq - a.pos);
                if (cfg.want_text) {
                    hits.push_back(' ');
                    append_int(hits, h_s); hits.push_back(' '); append_int(hits, h_p);
                    if (cfg.paired) { hits.push_back(' '); append_int(hits, h_i); }
                }
                if (cfg.want_bin) { h_sid.push_back(h_s); h_pos.push_back(h_p); if (cfg.paired) h_ins.push_back(h_i); }
                gids.push_back(cfg.gid_of[sid]);
            }
            ++out.units;
        }
        flush();
    } catch (const ParseError& e) {
        out.error = e.msg;
    }
}

// ---- decoded-stream sources -----------------------------------------------------------------------------------------------

struct Source {
    virtual ~Source() {}
    // bytes of the stream from the last commit point on, at least `want` of them unless the stream ends first
    virtual void wave(size_t want, const char*& base, size_t& n, bool& final) = 0;
    virtual void commit(size_t consumed) = 0;  // the first `consumed` bytes of the last wave are done
};

struct SamSource : Source {
    MappedFile f;
    size_t pos = 0;
    void wave(size_t want, const char*& base, size_t& n, bool& final) override {
        base = f.data + pos;
        n = std::min(want, f.size - pos);
        final = pos + n == f.size;
    }
    void commit(size_t consumed) override { pos += consumed; }
};

struct BamSource : Source {
    MappedFile f;
    size_t cpos = 0;            // compressed offset of the first block not yet in `tail`
    std::vector<char> tail;     // decoded bytes after the last commit point that are already inflated
    std::vector<char> buf;
    size_t cpos_buf = 0;        // compressed offset after the blocks in `buf`
    int threads = 1;
    void wave(size_t want, const char*& base, size_t& n, bool& final) override {
        struct Blk { size_t off, clen, isize, out; };
        std::vector<Blk> blks;
        size_t total = tail.size(), o = cpos;
        const uint8_t* d = (const uint8_t*)f.data;
        while (total < want && o < f.size) {
            if (o + 18 > f.size) die("Truncated BGZF block in the alignment file!");
            if (!(d[o] == 31 && d[o + 1] == 139 && d[o + 2] == 8 && (d[o + 3] & 4))) die("The alignment file is not BGZF-compressed!");
            const int xlen = d[o + 10] | (d[o + 11] << 8);
            int bsize = -1;
            for (int i = 0; i + 4 <= xlen;) {
                const uint8_t* x = d + o + 12 + i;
                const int slen = x[2] | (x[3] << 8);
                if (x[0] == 'B' && x[1] == 'C' && slen == 2) bsize = x[4] | (x[5] << 8);
                i += 4 + slen;
            }
            if (bsize < 0 || o + bsize + 1 > f.size) die("Corrupt BGZF block in the alignment file!");
            const size_t blen = (size_t)bsize + 1;
            uint32_t isize;
            memcpy(&isize, d + o + blen - 4, 4);
            blks.push_back(Blk{o + 12 + xlen, blen - 12 - xlen - 8, isize, total});
            total += isize;
            o += blen;
        }
        buf.resize(total);
        if (!tail.empty()) memcpy(buf.data(), tail.data(), tail.size());
        std::atomic<size_t> next{0};
        std::atomic<bool> bad{false};
        parallel_for(std::min<int>(threads, (int)std::max<size_t>(blks.size(), 1)), [&](int) {
            z_stream zs;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= blks.size()) break;
                const Blk& b = blks[i];
                if (b.isize == 0) continue;
                memset(&zs, 0, sizeof(zs));
                if (inflateInit2(&zs, -15) != Z_OK) { bad = true; break; }
                zs.next_in = (Bytef*)(d + b.off); zs.avail_in = (uInt)b.clen;
                zs.next_out = (Bytef*)buf.data() + b.out; zs.avail_out = (uInt)b.isize;
                const int rc = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (rc != Z_STREAM_END || zs.avail_out != 0) { bad = true; break; }
            }
        });
        if (bad) die("Corrupt BGZF block in the alignment file (inflate failed)!");
        cpos_buf = o;
        base = buf.data();
        n = total;
        final = o >= f.size;
    }
    void commit(size_t consumed) override {
        std::vector<char> t(buf.begin() + consumed, buf.end());
        tail.swap(t);
        cpos = cpos_buf;
    }
};

}  // namespace

int main(int argc, char* argv[]) {
    if (argc < 6) {
        printf("Usage : rsem-parse-alignments refName imdName statName alignF read_type [-t fai_file] [-tag tagName] [-q]\n");
        exit(-1);
    }
    const std::string refName = argv[1], imdName = argv[2], statName = argv[3], alignF = argv[4];
    Config cfg;
    cfg.read_type = atoi(argv[5]);
    if (cfg.read_type < 0 || cfg.read_type > 3) die("read_type must be 0, 1, 2 or 3!");
    cfg.paired = cfg.read_type >= 2;
    cfg.has_q = cfg.read_type & 1;
    bool verbose = true;
    int threads = hardware_threads();
    size_t wave_bytes = 0;
    for (int i = 6; i < argc; i++) {
        if (!strcmp(argv[i], "-tag") && i + 1 < argc) cfg.rt_tag = argv[i + 1];
        if (!strcmp(argv[i], "-q")) verbose = false;
        if (!strcmp(argv[i], "-p") && i + 1 < argc) threads = std::max(1, atoi(argv[i + 1]));
        if (!strcmp(argv[i], "--wave-bytes") && i + 1 < argc) wave_bytes = (size_t)atoll(argv[i + 1]);  // testing knob
    }
    {   // binary hand-off to rsem-run-em (host/rsb.hpp): a flag, or the environment when the Perl driver builds the command line
        bool bin = false, text = false;
        for (int i = 6; i < argc; i++) { bin = bin || !strcmp(argv[i], "--binary"); text = text || !strcmp(argv[i], "--text"); }
        bool from_env = false;
        if (const char* e = getenv("RSEM_HIP_BINARY")) {
            if (!strcmp(e, "both")) { bin = true; text = true; }
            else if (*e && strcmp(e, "0")) { from_env = !bin; bin = true; }
        }
        cfg.want_bin = bin;
        cfg.want_text = !bin || text;
        cfg.keep_alignable = from_env && !cfg.want_text;
    }
    if (!cfg.rt_tag.empty() && cfg.rt_tag.size() != 2) die("-tag expects a two-character SAM tag!");
    if (!wave_bytes) wave_bytes = std::max<size_t>((size_t)threads << 23, (size_t)64 << 20);
    init_tables();

    GroupInfo gi;
    if (!gi.load(refName + ".grp")) die("Cannot open %s.grp! It may not exist.", refName.c_str());
    cfg.gid_of.assign(gi.starts.empty() ? 1 : gi.starts.back() + 1, 0);
    for (int g = 0; g < gi.m; g++)
        for (int s = gi.starts[g]; s < gi.starts[g + 1]; s++) cfg.gid_of[s] = g;
    const Transcripts T = load_transcripts(refName + ".ti");
    const int M = T.M;

    // ---- open the alignment file, read its header ------------------------------------------------------------------------
    MappedFile probe;
    if (!probe.open(alignF)) die("Cannot open %s! It may not exist.", alignF.c_str());
    const bool is_bam = probe.size >= 2 && (uint8_t)probe.data[0] == 31 && (uint8_t)probe.data[1] == 139;
    std::unique_ptr<Source> src;
    std::vector<std::string> target_names;
    size_t header_bytes = 0;
    if (!is_bam) {
        SamSource* s = new SamSource();
        s->f = probe;
        probe = MappedFile();  // ownership moved
        src.reset(s);
        size_t p = 0;
        while (p < s->f.size && s->f.data[p] == '@') {
            const char* nl = (const char*)memchr(s->f.data + p, '\n', s->f.size - p);
            const size_t e = nl ? (size_t)(nl - s->f.data) : s->f.size;
            if (e - p > 3 && memcmp(s->f.data + p, "@SQ", 3) == 0) {
                std::string name; long len = 0;
                size_t q = p + 3;
                while (q < e) {
                    const char* t = (const char*)memchr(s->f.data + q + 1, '\t', e - q - 1);
                    const size_t tt = t ? (size_t)(t - s->f.data) : e;
                    if (tt - q > 4 && memcmp(s->f.data + q + 1, "SN:", 3) == 0) name.assign(s->f.data + q + 4, tt - q - 4);
                    if (tt - q > 4 && memcmp(s->f.data + q + 1, "LN:", 3) == 0) len = atol(s->f.data + q + 4);
                    q = tt;
                }
                if (!name.empty() && name.back() == '\r') name.pop_back();
                target_names.push_back(name);
                cfg.target_len.push_back((int32_t)len);
            }
            p = nl ? e + 1 : e;
        }
        s->pos = p;
    } else {
        BamSource* s = new BamSource();
        s->f = probe;
        probe = MappedFile();
        s->threads = threads;
        src.reset(s);
        size_t want = (size_t)16 << 20;
        for (;;) {  // the header may span many BGZF blocks: grow until it fits
            const char* b; size_t n; bool fin;
            s->wave(want, b, n, fin);
            auto need = [&](size_t k) { return k <= n; };
            bool ok = false;
            do {
                if (!need(12)) break;
                if (memcmp(b, "BAM\1", 4) != 0) die("%s is gzip-compressed but not a BAM file!", alignF.c_str());
                int32_t l_text; memcpy(&l_text, b + 4, 4);
                size_t o = 8 + (size_t)l_text;
                if (!need(o + 4)) break;
                int32_t n_ref; memcpy(&n_ref, b + o, 4);
                o += 4;
                target_names.clear(); cfg.target_len.clear();
                bool short_read = false;
                for (int i = 0; i < n_ref; i++) {
                    if (!need(o + 4)) { short_read = true; break; }
                    int32_t l_name; memcpy(&l_name, b + o, 4);
                    if (!need(o + 4 + l_name + 4)) { short_read = true; break; }
                    target_names.emplace_back(b + o + 4, l_name > 0 ? l_name - 1 : 0);
                    int32_t l_ref; memcpy(&l_ref, b + o + 4 + l_name, 4);
                    cfg.target_len.push_back(l_ref);
                    o += 8 + (size_t)l_name;
                }
                if (short_read) break;
                header_bytes = o;
                ok = true;
            } while (false);
            if (ok) { s->commit(header_bytes); break; }
            if (fin) die("Fail to parse sam header!");
            want *= 4;
        }
    }

    // external (header order) -> internal transcript ids, and the .omit list (Transcripts.h:105-143)
    cfg.n_targets = (int)target_names.size();
    if (cfg.n_targets <= 0) die("The SAM/BAM file declares less than one reference sequence!");
    if (cfg.n_targets > M) die("The SAM/BAM file declares more reference sequences (%d) than RSEM knows (%d)!", cfg.n_targets, M);
    if (cfg.n_targets < M)
        fprintf(stderr, "Warning: The SAM/BAM file declares less reference sequences (%d) than RSEM knows (%d)! Please make sure that you aligned your reads against transcript sequences instead of genome.\n", cfg.n_targets, M);
    cfg.e2i.assign(cfg.n_targets, 0);
    {
        std::unordered_map<std::string, int> dict;
        dict.reserve((size_t)M * 2);
        for (int i = 1; i <= M; i++) {
            const std::string& tid = T.type == 2 ? T.t[i].seqname : T.t[i].transcript_id;
            if (!dict.emplace(tid, i).second) die("RSEM's indices might be corrupted, %s appears more than once!", tid.c_str());
        }
        std::vector<char> appeared(M + 1, 0);
        cfg.tid_of.reserve((size_t)cfg.n_targets * 2);
        for (int i = 0; i < cfg.n_targets; i++) {
            auto it = dict.find(target_names[i]);
            if (it == dict.end()) die("RSEM can not recognize reference sequence name %s!", target_names[i].c_str());
            if (it->second <= 0) die("Reference sequence name %s appears more than once in the SAM/BAM file!", target_names[i].c_str());
            cfg.e2i[i] = it->second;
            appeared[it->second] = 1;
            it->second = -1;
            cfg.tid_of.emplace(target_names[i], i);
        }
        FILE* fo = fopen((imdName + ".omit").c_str(), "w");
        if (!fo) die("Cannot open %s.omit for writing!", imdName.c_str());
        for (int i = 1; i <= M; i++)
            if (!appeared[i]) fprintf(fo, "%d\n", i);
        fclose(fo);
    }

    // category files: 0 unalignable, 1 alignable, 2 filtered ("max") (utils.h:129-149)
    const int n_os = cfg.paired ? 2 : 1;
    FILE* cat[3][2] = {{nullptr}};
    std::string cat_path[3][2];
    for (int c = 0; c < 3; c++) {
        if (!cfg.reads_text(c)) continue;
        const std::vector<std::string> names = read_file_names(imdName, c, cfg.read_type);
        for (int j = 0; j < n_os; j++) {
            cat_path[c][j] = names[j];
            cat[c][j] = fopen(names[j].c_str(), "w");
            if (!cat[c][j]) die("Cannot open %s for writing!", names[j].c_str());
        }
    }
    FILE* fdat = nullptr;
    if (cfg.want_text) {
        fdat = fopen((imdName + ".dat").c_str(), "w");
        if (!fdat) die("Cannot open %s.dat for writing!", imdName.c_str());
        fprintf(fdat, "%-99s\n", "");  // patched once the totals are known (parseIt.cpp:195-204)
    }
    std::unique_ptr<RsbWriter> rsb;
    if (cfg.want_bin) rsb.reset(new RsbWriter(imdName, cfg.read_type));
    else remove_rsb(imdName);  // a text-only run must not leave an OLDER binary hand-off behind: rsem-run-em would prefer it

    long long N[3] = {0, 0, 0}, nHits = 0, nMulti = 0, nIsoMulti = 0, cnt = 0, n_warns = 0, next_report = 1000000;
    std::map<long long, long long> counter;
    const size_t per_unit = cfg.paired ? 2 : 1;

    for (;;) {
        const char* base; size_t n; bool final;
        src->wave(wave_bytes, base, n, final);
        Wave w;
        w.base = base;
        w.is_bam = is_bam;
        size_t used = 0;  // bytes covered by complete records
        if (!is_bam) {
            // line starts, found in parallel
            size_t body = n;
            if (!final) {
                const char* nl = (const char*)memrchr(base, '\n', n);
                body = nl ? (size_t)(nl - base) + 1 : 0;
            }
            const int np = (int)std::min<size_t>((size_t)threads, body / (1 << 20) + 1);
            std::vector<std::vector<size_t>> starts(np);
            parallel_for(np, [&](int t) {
                const size_t lo = body / np * t, hi = t + 1 == np ? body : body / np * (t + 1);
                std::vector<size_t>& v = starts[t];
                if (t == 0 && body > 0) v.push_back(0);
                for (const char* q = base + lo; q < base + hi;) {
                    const char* nl = (const char*)memchr(q, '\n', base + hi - q);
                    if (!nl) break;
                    const size_t s = (size_t)(nl - base) + 1;
                    if (s < body) v.push_back(s);
                    q = nl + 1;
                }
            });
            for (auto& v : starts) w.beg.insert(w.beg.end(), v.begin(), v.end());
            w.end.resize(w.beg.size());
            for (size_t i = 0; i < w.beg.size(); i++) {
                size_t e = i + 1 < w.beg.size() ? w.beg[i + 1] - 1 : body;
                if (i + 1 == w.beg.size() && e > w.beg[i] && base[e - 1] == '\n') --e;
                w.end[i] = e;
            }
            // drop empty lines (a blank trailing line is common)
            size_t k = 0;
            for (size_t i = 0; i < w.beg.size(); i++)
                if (w.end[i] > w.beg[i] && !(w.end[i] == w.beg[i] + 1 && base[w.beg[i]] == '\r')) { w.beg[k] = w.beg[i]; w.end[k] = w.end[i]; ++k; }
            w.beg.resize(k); w.end.resize(k);
            used = body;
        } else {
            size_t o = 0;
            while (o + 4 <= n) {
                uint32_t bs;
                memcpy(&bs, base + o, 4);
                if (o + 4 + bs > n) break;
                w.beg.push_back(o + 4);
                w.end.push_back(o + 4 + bs);
                o += 4 + (size_t)bs;
            }
            used = o;
            if (final && used != n && verbose) fprintf(stderr, "Warning: the BAM file ends with a truncated record.\n");
        }
        size_t n_units = w.beg.size() / per_unit;  // an unpaired last record waits for its mate (or is dropped at EOF)
        size_t keep = n_units;                     // units processed in this wave
        if (!final) {
            // the last read may continue in the next wave: carry it over from its first unit
            Unit U, P;
            bool too_small = n_units == 0;
            try {
                while (!too_small) {
                    load_unit(w, keep - 1, cfg, U);
                    if (U.rt != 1) { --keep; break; }
                    if (keep == 1) { too_small = true; break; }
                    load_unit(w, keep - 2, cfg, P);
                    --keep;
                    if (!U.a.same_name(P.a)) break;
                }
            } catch (const ParseError& e) { die("%s", e.msg.c_str()); }
            if (too_small || keep == 0) { wave_bytes *= 2; continue; }  // one read larger than the wave: widen it
            used = w.beg[keep * per_unit] - (is_bam ? 4 : 0);
        }
        // chunk boundaries at read starts
        const size_t want_chunks = std::max<size_t>(1, std::min<size_t>((size_t)threads * 4, keep / 256 + 1));
        std::vector<size_t> cut{0};
        {
            Unit U, P;
            try {
                for (size_t c = 1; c < want_chunks; c++) {
                    size_t u = keep / want_chunks * c;
                    if (u <= cut.back()) continue;
                    while (u < keep) {  // advance to a unit that starts a read whatever came before
                        load_unit(w, u, cfg, U);
                        if (U.rt != 1) break;
                        load_unit(w, u - 1, cfg, P);
                        if (!U.a.same_name(P.a)) break;
                        ++u;
                    }
                    if (u < keep && u > cut.back()) cut.push_back(u);
                }
            } catch (const ParseError& e) { die("%s", e.msg.c_str()); }
        }
        cut.push_back(keep);
        const int nc = (int)cut.size() - 1;
        std::vector<ChunkOut> outs(nc);
        std::atomic<int> next{0};
        parallel_for(std::min(threads, nc), [&](int) {
            for (;;) {
                const int c = next.fetch_add(1);
                if (c >= nc) break;
                run_chunk(w, cut[c], cut[c + 1], cfg, outs[c]);
            }
        });
        for (int c = 0; c < nc; c++) {
            ChunkOut& o = outs[c];
            for (auto& m : o.warns)
                if (++n_warns <= 50) fprintf(stderr, "%s\n", m.c_str());
            n_warns += o.n_warns - (long long)o.warns.size();
            if (cfg.want_text) fwrite(o.dat.data(), 1, o.dat.size(), fdat);
            for (int k = 0; k < 3; k++)
                for (int j = 0; j < n_os && cfg.reads_text(k); j++) fwrite(o.reads[k][j].data(), 1, o.reads[k][j].size(), cat[k][j]);
            if (rsb && o.error.empty()) {
                rsb->append_hits(o.b_rowlen.data(), o.b_rowlen.size(), o.b_sid.data(), o.b_pos.data(), o.b_ins.data());
                for (int k = 0; k < 3; k++)
                    for (int j = 0; j < n_os; j++)
                        rsb->append_reads(k, j, o.b_len[k][j].data(), o.b_len[k][j].size(), o.b_seq[k][j].data(), o.b_qual[k][j].data());
            }
            for (int k = 0; k < 3; k++) N[k] += o.N[k];
            nHits += o.nHits; nMulti += o.nMulti; nIsoMulti += o.nIsoMulti;
            for (auto& kv : o.counter) counter[kv.first] += kv.second;
            cnt += o.units;
            if (!o.error.empty()) die("%s", o.error.c_str());
        }
        while (verbose && cnt >= next_report) { printf("Parsed %lld entries\n", next_report); next_report += 1000000; }
        if (final) break;
        src->commit(used);
    }
    if (n_warns > 0) fprintf(stderr, "Warning: Detected %lld lines containing read pairs whose two mates have different names.\n", n_warns);
    const long long nUnique = N[1] - nMulti;

    if (fdat) {
        fflush(fdat);
        fseek(fdat, 0, SEEK_SET);
        fprintf(fdat, "%lld %lld %d", N[1], nHits, cfg.read_type);
        fclose(fdat);
    }
    if (rsb) {
        rsb->finish();
        const RsbHeader& h = rsb->header();
        if ((long long)h.N[0] != N[0] || (long long)h.N[1] != N[1] || (long long)h.N[2] != N[2] || (long long)h.nHits != nHits)
            die("rsem-parse-alignments: the binary hand-off does not add up (%llu %llu %llu reads, %llu alignments)!", (unsigned long long)h.N[0],
                (unsigned long long)h.N[1], (unsigned long long)h.N[2], (unsigned long long)h.nHits);
    }

    FILE* fc = fopen((statName + ".cnt").c_str(), "w");
    if (!fc) die("Cannot open %s.cnt for writing!", statName.c_str());
    fprintf(fc, "%lld %lld %lld %lld\n", N[0], N[1], N[2], N[0] + N[1] + N[2]);
    fprintf(fc, "%lld %lld %lld\n", nUnique, nMulti, nIsoMulti);
    fprintf(fc, "%lld %d\n", nHits, cfg.read_type);
    fprintf(fc, "0\t%lld\n", N[0]);
    for (auto& kv : counter) fprintf(fc, "%lld\t%lld\n", kv.first, kv.second);
    fprintf(fc, "Inf\t%lld\n", N[2]);
    fclose(fc);

    for (int c = 0; c < 3; c++)
        for (int j = 0; j < n_os && cfg.reads_text(c); j++) {
            fclose(cat[c][j]);
            if (N[c] == 0) remove(cat_path[c][j].c_str());
        }
    if (verbose) printf("Done!\n");
    return 0;
}
