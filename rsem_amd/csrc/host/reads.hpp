// reads.hpp -- imd.dat and imd_{un,alignable,max}[_1|_2].f[aq] into packed arrays, once.
// The reference streams these text files again in every one of rounds 1-11 (EM.cpp:195-202,
// ReadReader.h); here they are parsed a single time (multi-threaded, mmap) into the byte arrays the
// device kernels use.  Formats: SURVEY.md Appendix A.
#pragma once
#include "files.hpp"

namespace rsemh {

// get_base_id (utils.h:36-50): A C G T N (either case) -> 0..4
// (Built once, by whichever thread comes first, behind the language's own guard for function-local statics: the read files of the
// two mates and the reference strands are converted by threads that start at the same time, and a hand-made "init" flag let a
// second thread wipe the table while the first was already reading it -- a rare "unknown sequence letter T".)
inline const int8_t* base_table() {
    struct Table {
        int8_t t[256];
        Table() {
            memset(t, -1, sizeof(t));
            t['a'] = t['A'] = 0; t['c'] = t['C'] = 1; t['g'] = t['G'] = 2; t['t'] = t['T'] = 3; t['n'] = t['N'] = 4;
        }
    };
    static const Table tbl;
    return tbl.t;
}

struct ReadFile {
    uint64_t n = 0;
    Arr<uint64_t> off;         // [n+1]
    Arr<uint8_t> seq;          // base ids
    Arr<uint8_t> qual;         // quality - 33 (FASTQ only)
    std::vector<uint8_t> lq1;  // Single*Read::calc_lq of this mate alone
    int len(uint64_t i) const { return (int)(off[i + 1] - off[i]); }
};

// SingleRead::calc_lq / SingleReadQ::calc_lq (SingleRead.h:57-88, SingleReadQ.h:63-95), on the raw characters
inline bool calc_lq_single(const char* s, int len, bool hasPolyA, int seedLen) {
    if (len < seedLen) return true;
    if (!hasPolyA) return false;
    int numA = 0, numT = 0, numAO = 0, numTO = 0;
    const int threshold_1 = int(0.9 * len - 1.5 * sqrt(len * 1.0) + 0.5);
    const int threshold_2 = (kOLen - 1) / 2 + 1;
    for (int i = 0; i < len; i++) {
        if (s[i] == 'A') { ++numA; if (i < kOLen) ++numAO; }
        if (s[i] == 'T') { ++numT; if (i >= len - kOLen) ++numTO; }
    }
    if (numA >= threshold_1) return numAO >= threshold_2;
    if (numT >= threshold_1) return numTO >= threshold_2;
    return false;
}

// the same on base ids (A = 0, T = 3): reads that arrive already decoded (host/rsb.hpp)
inline bool calc_lq_single_ids(const uint8_t* s, int len, bool hasPolyA, int seedLen) {
    if (len < seedLen) return true;
    if (!hasPolyA) return false;
    int numA = 0, numT = 0, numAO = 0, numTO = 0;
    const int threshold_1 = int(0.9 * len - 1.5 * sqrt(len * 1.0) + 0.5);
    const int threshold_2 = (kOLen - 1) / 2 + 1;
    for (int i = 0; i < len; i++) {
        if (s[i] == 0) { ++numA; if (i < kOLen) ++numAO; }
        if (s[i] == 3) { ++numT; if (i >= len - kOLen) ++numTO; }
    }
    if (numA >= threshold_1) return numAO >= threshold_2;
    if (numT >= threshold_1) return numTO >= threshold_2;
    return false;
}

// FASTA: 2 lines per read; FASTQ: 4 lines per read (SingleRead.h:38-50, SingleReadQ.h:38-56).
// Three parallel scans of the mapped text, nothing in between: (0) newlines per chunk, which gives every chunk's phase
// within the 2 / 4-line records, and the chunk boundaries are moved to record starts; (1) records and bases per chunk,
// which gives every chunk its place in the final arrays; (2) the bases, qualities, lengths and low-quality flags are
// written straight to those places.  (Until round 4 every chunk collected its reads in vectors of its own that were
// then copied into place: at BASELINE configs[2] 22 GB of intermediate 4 KB pages, first-touched by 64 threads of one
// address space -- the page faults, not the parsing, set the time.)
// files above this size are parsed by several threads, each a run of whole records (RSEM_HIP_PARSE_SPLIT_BYTES: tests set it
// to a few hundred bytes so that the chunk-boundary logic is exercised on small files)
inline size_t parse_split_bytes() {
    static const size_t v = []() { const char* e = getenv("RSEM_HIP_PARSE_SPLIT_BYTES"); return e ? (size_t)std::max(0ll, atoll(e)) : ((size_t)32 << 20); }();
    return v;
}

inline void prefault_text(const char* p, size_t n) {
#ifdef MADV_POPULATE_READ
    static const bool on = !getenv("RSEM_HIP_NO_PREFAULT");
    if (on && n >= ((size_t)1 << 20)) {
        const uintptr_t a = (uintptr_t)p & ~(uintptr_t)4095;
        (void)madvise((void*)a, (size_t)((uintptr_t)p + n - a), MADV_POPULATE_READ);  // page tables of the chunk in one call
    }
#else
    (void)p; (void)n;
#endif
}

inline ReadFile parse_read_file(const std::string& path, bool fastq, bool hasPolyA, int seedLen, int threads = 0) {
    MappedFile f;
    if (!f.open(path)) die("Cannot open %s! It may not exist.", path.c_str());
    const int L = fastq ? 4 : 2;
    const int nt = f.size > parse_split_bytes() ? (threads > 0 ? threads : hardware_threads()) : 1;
    std::vector<size_t> cut = line_chunks(f.data, 0, f.size, nt);
    int nc = (int)cut.size() - 1;
    const char* fend = f.data + f.size;
    // scan 0: lines per chunk
    std::vector<uint64_t> nl(nc, 0);
    parallel_for(nc, [&](int c) {
        uint64_t k = 0;
        const char* p = f.data + cut[c];
        const char* e = f.data + cut[c + 1];
        prefault_text(p, (size_t)(e - p));
        while (p < e) {
            const char* q = (const char*)memchr(p, '\n', e - p);
            if (!q) { ++k; break; }  // last line without a newline
            ++k;
            p = q + 1;
        }
        nl[c] = k;
    });
    // chunk boundaries -> record starts (a chunk that starts in the middle of a record gives its head to the chunk before)
    {
        uint64_t line = 0;
        for (int c = 0; c < nc; c++) {
            const uint64_t first = line;
            line += nl[c];
            if (c == 0) continue;
            size_t q = cut[c];
            for (uint64_t k = first; k % L != 0 && q < f.size; k++) {
                const char* e2 = (const char*)memchr(f.data + q, '\n', f.size - q);
                q = e2 ? (size_t)(e2 - f.data) + 1 : f.size;
            }
            cut[c] = std::min(q, f.size);
        }
        for (int c = 1; c <= nc; c++) cut[c] = std::max(cut[c], cut[c - 1]);  // (a chunk shorter than a record becomes empty)
    }
    auto next_line = [&](const char*& p, const char*& b, const char*& le) -> bool {
        if (p >= fend) return false;
        b = p;
        const char* q = (const char*)memchr(p, '\n', fend - p);
        le = q ? q : fend;
        p = q ? q + 1 : fend;
        if (le > b && le[-1] == '\r') --le;
        return true;
    };
    // scan 1: records and bases per chunk
    std::vector<uint64_t> nrec(nc, 0), nbase(nc, 0);
    parallel_for(nc, [&](int c) {
        const char* p = f.data + cut[c];
        const char* e = f.data + cut[c + 1];
        const char *b, *le;
        uint64_t r = 0, nb = 0;
        while (p < e) {
            if (!next_line(p, b, le)) break;
            if (b == le) continue;  // stray empty line at the end
            if (*b != (fastq ? '@' : '>')) die("Read file %s does not look like a %s file!", path.c_str(), fastq ? "FASTQ" : "FASTA");
            if (!next_line(p, b, le)) die("%s: truncated record", path.c_str());
            nb += (uint64_t)(le - b);
            ++r;
            if (fastq) {
                if (!next_line(p, b, le) || b == le || *b != '+') die("Read file %s does not look like a FASTQ file!", path.c_str());
                if (!next_line(p, b, le)) die("%s: truncated record", path.c_str());
            }
        }
        nrec[c] = r; nbase[c] = nb;
    });
    ReadFile R;
    std::vector<uint64_t> r0(nc + 1, 0), b0(nc + 1, 0);
    for (int c = 0; c < nc; c++) { r0[c + 1] = r0[c] + nrec[c]; b0[c + 1] = b0[c] + nbase[c]; }
    R.n = r0[nc];
    uint64_t* off = R.off.alloc(R.n + 1);
    uint8_t* seq = R.seq.alloc(b0[nc]);
    uint8_t* qual = fastq ? R.qual.alloc(b0[nc]) : nullptr;
    R.lq1.resize(R.n);
    off[R.n] = b0[nc];
    const int8_t* tbl = base_table();
    // scan 2: straight into place
    parallel_for(nc, [&](int c) {
        const char* p = f.data + cut[c];
        const char* e = f.data + cut[c + 1];
        const char *b, *le;
        uint64_t r = r0[c], o = b0[c];
        while (p < e) {
            if (!next_line(p, b, le)) break;
            if (b == le) continue;
            if (!next_line(p, b, le)) break;
            const int len = (int)(le - b);
            off[r] = o;
            uint8_t* sq = seq + o;
            for (int i = 0; i < len; i++) {
                int8_t id = tbl[(unsigned char)b[i]];
                if (id < 0) die("Found unknown sequence letter %c at function get_base_id!", b[i]);
                sq[i] = (uint8_t)id;
            }
            R.lq1[r] = calc_lq_single(b, len, hasPolyA, seedLen) ? 1 : 0;
            if (fastq) {
                next_line(p, b, le);
                if (!next_line(p, b, le)) break;
                if ((int)(le - b) != len) die("%s: quality string and sequence differ in length", path.c_str());
                uint8_t* ql = qual + o;
                for (int i = 0; i < len; i++) {
                    int qv = (unsigned char)b[i] - 33;  // c2q (QProfile.h:44)
                    if (qv < 0 || qv > 93) die("%s: quality character out of range", path.c_str());
                    ql[i] = (uint8_t)qv;
                }
            }
            o += (uint64_t)len;
            ++r;
        }
    });
    return R;
}

// genReadFileNames (utils.h:129-149)
inline std::vector<std::string> read_file_names(const std::string& imdName, int tagType, int read_type) {
    static const char* tags[3] = {"un", "alignable", "max"};
    const char* suf = (read_type == 0 || read_type == 2) ? "fa" : "fq";
    std::vector<std::string> v;
    if (read_type < 2) v.push_back(imdName + "_" + tags[tagType] + "." + suf);
    else {
        v.push_back(imdName + "_" + tags[tagType] + "_1." + suf);
        v.push_back(imdName + "_" + tags[tagType] + "_2." + suf);
    }
    return v;
}

struct DatData {  // imd.dat (HitContainer.h:63-91, SingleHit.h:44-51, PairedEndHit.h:27-34)
    uint64_t N1 = 0, nHits = 0;
    int read_type = 0;
    Arr<uint64_t> row_ptr;
    Arr<int32_t> sid_signed, pos, insertL;
};

inline DatData load_dat(const std::string& path, int expect_read_type, int threads = 0) {
    MappedFile f;
    if (!f.open(path)) die("Cannot open %s! It may not exist.", path.c_str());
    DatData D;
    const char* p = f.data;
    const char* end = f.data + f.size;
    long long a, b, c;
    if (!parse_long(p, end, a) || !parse_long(p, end, b) || !parse_long(p, end, c)) die("%s: bad header", path.c_str());
    D.N1 = (uint64_t)a; D.nHits = (uint64_t)b; D.read_type = (int)c;
    if (D.read_type != expect_read_type) die("Data file (.dat) does not have the right read type!");
    const bool pe = D.read_type >= 2;
    const char* nl = (const char*)memchr(p, '\n', end - p);
    size_t body = nl ? (size_t)(nl - f.data) + 1 : f.size;
    const int nt = f.size > parse_split_bytes() ? (threads > 0 ? threads : hardware_threads()) : 1;
    std::vector<size_t> cut = line_chunks(f.data, body, f.size, nt);
    const int nc = (int)cut.size() - 1;
    // scan 1: reads and alignments per chunk (the first number of every line); scan 2: the numbers, straight into place
    // (no per-chunk vectors in between: see parse_read_file)
    std::vector<uint64_t> nrow(nc, 0), nhit(nc, 0);
    parallel_for(nc, [&](int ci) {
        const char* q = f.data + cut[ci];
        const char* e = f.data + cut[ci + 1];
        prefault_text(q, (size_t)(e - q));
        uint64_t r = 0, h = 0;
        while (q < e) {
            const char* le = (const char*)memchr(q, '\n', e - q);
            if (!le) le = e;
            long long k;
            if (parse_long(q, le, k)) {
                if (k <= 0) die("%s: a read without alignments", path.c_str());
                // every alignment takes at least " s p": a count the line cannot hold is a corrupt file, said here and not by an
                // allocation of that many records
                if ((unsigned long long)k > (unsigned long long)(le - q) / 2 + 1) die("%s: Cannot read alignments (a line announces %lld of them and is %lld bytes long)", path.c_str(), k, (long long)(le - q));
                ++r;
                h += (uint64_t)k;
            }
            q = le + 1;
        }
        nrow[ci] = r; nhit[ci] = h;
    });
    std::vector<uint64_t> r0(nc + 1, 0), h0(nc + 1, 0);
    for (int c = 0; c < nc; c++) { r0[c + 1] = r0[c] + nrow[c]; h0[c + 1] = h0[c] + nhit[c]; }
    uint64_t* rp = D.row_ptr.alloc(r0[nc] + 1);
    int32_t* sidp = D.sid_signed.alloc(h0[nc]);
    int32_t* posp = D.pos.alloc(h0[nc]);
    int32_t* insp = pe ? D.insertL.alloc(h0[nc]) : nullptr;
    rp[r0[nc]] = h0[nc];
    parallel_for(nc, [&](int ci) {
        const char* q = f.data + cut[ci];
        const char* e = f.data + cut[ci + 1];
        uint64_t r = r0[ci], o = h0[ci];
        while (q < e) {
            const char* le = (const char*)memchr(q, '\n', e - q);
            if (!le) le = e;
            long long k;
            if (parse_long(q, le, k)) {
                rp[r++] = o;
                for (long long t = 0; t < k; t++) {
                    long long sv, ps, il = 0;
                    if (!parse_long(q, le, sv) || !parse_long(q, le, ps) || (pe && !parse_long(q, le, il)))
                        die("Cannot read alignments from .dat file!");
                    sidp[o] = (int32_t)sv;
                    posp[o] = (int32_t)ps;
                    if (pe) insp[o] = (int32_t)il;
                    ++o;
                }
            }
            q = le + 1;
        }
    });
    if (D.row_ptr.size() - 1 != D.N1) die("Number of alignable reads does not match!");
    return D;
}

}  // namespace rsemh
