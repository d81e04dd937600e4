// files.hpp -- readers / writers for the files rsem-run-em and rsem-run-gibbs exchange with the
// rest of the RSEM pipeline (formats: SURVEY.md Appendix A; each function cites the reference
// code that defines the format).  Host side of the product; plain C++17, no GPU code here.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

namespace rsemh {

constexpr double kEpsilon = 1e-300;  // utils.h:19
constexpr double kMinEel = 1.0;      // utils.h:20
constexpr int kOLen = 25;            // utils.h:23

// the reference's error convention: message on stderr, exit(-1)  (my_assert.h:89-96)
[[noreturn]] inline void die(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    exit(-1);
}

// read-only memory map of a whole file
struct MappedFile {
    const char* data = nullptr;
    size_t size = 0;
    int fd = -1;
    bool open(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); fd = -1; return false; }
        size = (size_t)st.st_size;
        if (size == 0) { data = ""; return true; }
        void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) { ::close(fd); fd = -1; return false; }
        madvise(p, size, MADV_SEQUENTIAL);
        data = (const char*)p;
        return true;
    }
    ~MappedFile() {
        // (the pages' entries are dropped under the address space's SHARED lock first: a munmap of a 10 GB text file holds the exclusive one
        // for its whole walk, and the other parsers' page faults wait for it)
        if (data && size >= ((size_t)64 << 20)) (void)madvise((void*)data, size, MADV_DONTNEED);
        if (data && size) munmap((void*)data, size);
        if (fd >= 0) ::close(fd);
    }
};

// A typed array that either owns its storage (uninitialised on allocation: the parsers fill it from many threads, which
// also spreads the first-touch page faults) or views memory owned elsewhere (a mapped file).
// Big arrays are 2 MB-aligned and advised as huge pages (transparent_hugepage = madvise / always): 512 x fewer page faults
// while the parsers fill them and 512 x fewer pages to give back at exit -- at BASELINE configs[2] the process holds
// 35 GB of parsed arrays, and tearing down that many 4 KB pages was 3 s of wall clock after main() had returned.
struct ArrFree { void operator()(void* q) const { free(q); } };
template <typename T>
struct Arr {
    std::unique_ptr<T[], ArrFree> own;
    std::shared_ptr<MappedFile> keep;
    const T* p = nullptr;
    size_t n = 0;
    const T* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    const T& operator[](size_t i) const { return p[i]; }
    T* alloc(size_t m) {
        static_assert(std::is_trivial<T>::value, "Arr holds plain data");
        const size_t bytes = (m ? m : 1) * sizeof(T);
        void* q = nullptr;
        if (bytes >= ((size_t)32 << 20)) {
            const size_t huge = (size_t)2 << 20, rounded = (bytes + huge - 1) / huge * huge;
            if (posix_memalign(&q, huge, rounded) != 0) q = nullptr;
            else madvise(q, rounded, MADV_HUGEPAGE);
        } else q = malloc(bytes);
        if (!q) die("Out of memory (%zu bytes)!", bytes);
        own.reset((T*)q); p = own.get(); n = m;
        return own.get();
    }
    void view(const T* q, size_t m, std::shared_ptr<MappedFile> k) { own.reset(); keep = std::move(k); p = q; n = m; }
    void release() { own.reset(); keep.reset(); p = nullptr; n = 0; }
    // the owned storage as a byte range (empty for a view or a small array): what give_back_pages() below takes
    std::pair<char*, size_t> owned_bytes() const {
        const size_t bytes = n * sizeof(T);
        return (own && bytes >= ((size_t)32 << 20)) ? std::make_pair((char*)own.get(), bytes) : std::make_pair((char*)nullptr, (size_t)0);
    }
};

// The CPU time the process's cgroup grants, in cores (cpu.max of cgroup v2, the CFS quota of v1); 0 = no limit found.  A container may
// show 256 hardware threads and grant 16 cores of time (the gpurun boxes do: profiles/r06q_call.log); stages whose threads only compete
// for that time are faster with as many threads as cores granted (the -b pass: 7.7 s with 16 threads, 8.5-8.7 with 64, profiles/r06ao_call.log).
inline int cgroup_cpu_cores() {
    static const int cores = []() -> int {
        double q = 0, per = 0;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char a[64] = {0};
            const int n = fscanf(f, "%63s %lf", a, &per);
            fclose(f);
            if (n == 2 && strcmp(a, "max") != 0 && per > 0) { q = atof(a); return q > 0 ? (int)ceil(q / per) : 0; }
            return 0;
        }
        FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
        FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        int r = 0;
        if (f1 && f2 && fscanf(f1, "%lf", &q) == 1 && fscanf(f2, "%lf", &per) == 1 && q > 0 && per > 0) r = (int)ceil(q / per);
        if (f1) fclose(f1);
        if (f2) fclose(f2);
        return r;
    }();
    return cores;
}

inline int hardware_threads() {  // (NOT cut to the cgroup's cores: the parsers share these among five files read at once, and with the boxes' 16 cores the
    // longest of them got three threads: 10.5-11.8 s at configs[2] where the hardware's count gives 9.5-9.8, profiles/r06aq_call.log)
    unsigned n = std::thread::hardware_concurrency();
    return (int)std::min<unsigned>(std::max<unsigned>(n, 1), 64);
}
// threads for ONE of several files that are parsed at the same time (run_em.cpp parses .dat and the read files concurrently)
inline int file_parse_threads(int files_in_flight) {
    const unsigned n = std::max(1u, std::thread::hardware_concurrency());
    return (int)std::min<unsigned>(std::max<unsigned>(n / (unsigned)std::max(1, files_in_flight), 1), 64);
}

// split [begin, end) into ~n chunks that end right after a '\n'
inline std::vector<size_t> line_chunks(const char* buf, size_t begin, size_t end, int n) {
    std::vector<size_t> cut{begin};
    for (int i = 1; i < n; i++) {
        size_t p = begin + (end - begin) / n * i;
        if (p <= cut.back()) continue;
        const char* nl = (const char*)memchr(buf + p, '\n', end - p);
        if (!nl) break;
        size_t q = (size_t)(nl - buf) + 1;
        if (q > cut.back() && q < end) cut.push_back(q);
    }
    cut.push_back(end);
    return cut;
}

inline void parallel_for(int n, const std::function<void(int)>& fn) {
    if (n <= 1) { if (n == 1) fn(0); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < n; i++) th.emplace_back(fn, i);
    for (auto& t : th) t.join();
}

// Give the pages of big arrays back to the system on several threads BEFORE their owners free them.  The boxes' kernels clear every
// page they take back (51 ms per GB whatever the page size: tools/thp_probe.cpp), so the 35 GB of parsed inputs a run of rsem-run-em holds
// at configs[2] are 1.7 s of kernel work for whoever frees them -- one thread doing it through munmap held the address space's lock for
// most of that time and the thread that launches the device loop waited for it (0.8 s of a 5.1-second loop, profiles/r06ai_*).
// MADV_DONTNEED takes that lock shared, in slices on `threads` threads; the free() that follows finds nothing left to clear.
inline void give_back_pages(const std::vector<std::pair<char*, size_t>>& ranges, int threads) {
    const size_t slice = (size_t)256 << 20, huge = (size_t)2 << 20;
    std::vector<std::pair<char*, size_t>> work;
    for (const auto& r : ranges) {
        if (!r.first || !r.second) continue;
        char* b = (char*)(((uintptr_t)r.first + huge - 1) / huge * huge);
        char* e = (char*)(((uintptr_t)r.first + r.second) / huge * huge);
        for (; b < e; b += slice) work.push_back({b, (size_t)std::min<ptrdiff_t>((ptrdiff_t)slice, e - b)});
    }
    if (work.empty()) return;
    std::atomic<size_t> next{0};
    const int nt = std::max(1, std::min<int>(threads, (int)work.size()));
    parallel_for(nt, [&](int) {
        for (size_t k; (k = next.fetch_add(1)) < work.size();) (void)madvise(work[k].first, work[k].second, MADV_DONTNEED);
    });
}

// One line of cells first..last (inclusive) separated by `sep` and closed by a newline; cell(buf, i) writes cell i into a buffer of
// kCellBuf bytes and returns its length as snprintf does (a "%.2f" of 1e300 has 304 characters: the buffer holds them; a cell that
// still does not fit ends the program rather than the line with garbage).  A long line -- the per-transcript rows of .theta, .model and the result files: 200 k numbers
// each, 0.1-0.2 s of printf per file when done by one thread -- is formatted in pieces on the host's threads and written in order.
constexpr int kCellBuf = 352;
template <typename Cell>
inline void write_cells_line(FILE* f, long first, long last, char sep, Cell cell) {
    const long n = last - first + 1;
    if (n <= 0) { fputc('\n', f); return; }
    const int nt = n >= 50000 ? std::max(1, std::min(32, hardware_threads())) : 1;
    std::vector<std::string> part(nt);
    parallel_for(nt, [&](int t) {
        const long lo = first + n * t / nt, hi = first + n * (t + 1) / nt;
        std::string& out = part[t];
        out.reserve((size_t)(hi - lo) * 12);
        char buf[kCellBuf];
        for (long i = lo; i < hi; i++) {
            const int k = cell(buf, i);
            if (k < 0 || k >= kCellBuf) die("write_cells_line: a cell of %d characters does not fit", k);
            out.append(buf, (size_t)k);
            out.push_back(i < last ? sep : '\n');
        }
    });
    for (const std::string& o : part) fwrite(o.data(), 1, o.size(), f);
}

// fast decimal integer (optionally signed); advances p; skips leading blanks (not newlines)
inline bool parse_long(const char*& p, const char* end, long long& v) {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
    if (p >= end) return false;
    bool neg = false;
    if (*p == '-') { neg = true; ++p; } else if (*p == '+') ++p;
    if (p >= end || *p < '0' || *p > '9') return false;
    long long x = 0;
    while (p < end && *p >= '0' && *p <= '9') { x = x * 10 + (*p - '0'); ++p; }
    v = neg ? -x : x;
    return true;
}

// ---- reference files -------------------------------------------------------------------------

struct RefInfo {  // ref.seq (RefSeq.h:108-138, Refs.h:118-145)
    int M = 0;
    bool has_polyA = false;
    std::vector<int32_t> fullLen, totLen;        // [M+1], index 0 unused
    std::vector<std::string> seq;                // forward strand incl. poly(A) tail (empty when not loaded)
    std::vector<std::vector<uint32_t>> masks;    // fmasks words
    bool mask_at(int sid, int pos) const { return masks[sid][pos / 32] & (1u << (pos % 32)); }
};

inline RefInfo load_refs(const std::string& path, bool with_seq) {
    MappedFile f;
    if (!f.open(path)) die("Cannot open %s! It may not exist.", path.c_str());
    RefInfo R;
    R.fullLen.push_back(0); R.totLen.push_back(0); R.seq.emplace_back(); R.masks.emplace_back();
    const char* p = f.data;
    const char* end = f.data + f.size;
    auto next_line = [&](const char*& b, const char*& e) -> bool {
        if (p >= end) return false;
        b = p;
        const char* nl = (const char*)memchr(p, '\n', end - p);
        e = nl ? nl : end;
        p = nl ? nl + 1 : end;
        return true;
    };
    const char *b, *e;
    while (next_line(b, e)) {
        if (b == e) continue;
        long long fl, tl;
        const char* q = b;
        if (!parse_long(q, e, fl) || !parse_long(q, e, tl)) break;
        if (!next_line(b, e)) break;  // name
        if (!next_line(b, e)) break;  // sequence
        if (with_seq) R.seq.emplace_back(b, e - b); else R.seq.emplace_back();
        if (!next_line(b, e)) break;  // mask words
        std::vector<uint32_t> m;
        q = b;
        long long w;
        while (parse_long(q, e, w)) m.push_back((uint32_t)w);
        R.fullLen.push_back((int32_t)fl);
        R.totLen.push_back((int32_t)tl);
        R.masks.push_back(std::move(m));
        R.has_polyA = R.has_polyA || fl < tl;
        ++R.M;
    }
    return R;
}

struct TranscriptInfo {  // ref.ti (Transcripts.h:81-94, Transcript.h:119-148)
    std::string transcript_id, transcript_name, gene_id, gene_name, seqname;
    int length = 0;
};

struct Transcripts {
    int M = 0, type = 0;
    std::vector<TranscriptInfo> t;  // [M+1]
};

inline Transcripts load_transcripts(const std::string& path) {
    FILE* fi = fopen(path.c_str(), "r");
    if (!fi) die("Cannot open %s! It may not exist.", path.c_str());
    Transcripts T;
    if (fscanf(fi, "%d %d", &T.M, &T.type) != 2) die("%s: bad header", path.c_str());
    char* line = nullptr;
    size_t cap = 0;
    auto getl = [&](std::string& s) {
        ssize_t n = getline(&line, &cap, fi);
        if (n < 0) die("%s: truncated", path.c_str());
        while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) --n;
        s.assign(line, n);
    };
    std::string s;
    getl(s);  // rest of the header line
    T.t.resize(T.M + 1);
    for (int i = 1; i <= T.M; i++) {
        TranscriptInfo& x = T.t[i];
        getl(s);
        size_t tab = s.find('\t');
        x.transcript_id = s.substr(0, tab);
        x.transcript_name = tab == std::string::npos ? "" : s.substr(tab + 1);
        getl(s);
        tab = s.find('\t');
        x.gene_id = s.substr(0, tab);
        x.gene_name = tab == std::string::npos ? "" : s.substr(tab + 1);
        getl(x.seqname);
        getl(s);  // strand length
        char strand;
        if (sscanf(s.c_str(), " %c %d", &strand, &x.length) != 2) die("%s: bad transcript record %d", path.c_str(), i);
        getl(s);  // structure
        getl(s);  // left
    }
    free(line);
    fclose(fi);
    return T;
}

struct GroupInfo {  // ref.grp / .gt / .ta  (GroupInfo.h:34-53)
    int m = 0;
    std::vector<int32_t> starts;  // [m+1]
    bool load(const std::string& path) {
        FILE* fi = fopen(path.c_str(), "r");
        if (!fi) return false;
        int v;
        starts.clear();
        while (fscanf(fi, "%d", &v) == 1) starts.push_back(v);
        fclose(fi);
        m = (int)starts.size() - 1;
        return m >= 0;
    }
};

inline bool file_exists(const std::string& p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}

// ---- length distribution (shared by the model file reader and the result math) --------------------

struct LenDist {  // LenDist.h: support (lb, ub], pdf/cdf indexed 1..span
    int lb = 0, ub = 1, span = 1;
    std::vector<double> pdf, cdf;
    LenDist() { reset(1, 1000); }
    void reset(int minL, int maxL) {  // LenDist.h:20-34 (uniform initial parameters)
        lb = minL - 1; ub = maxL; span = ub - lb;
        pdf.assign(span + 1, 0.0); cdf.assign(span + 1, 0.0);
        for (int i = 1; i <= span; i++) { pdf[i] = 1.0 / span; cdf[i] = i * 1.0 / span; }
    }
    int minL() const { return lb + 1; }
    int maxL() const { return ub; }
    void trim() {  // LenDist.h:265-294
        int newlb, newub;
        for (newlb = 1; newlb <= span && pdf[newlb] < kEpsilon; newlb++) {}
        newlb--;
        for (newub = span; newub > newlb && pdf[newub] < kEpsilon; newub--) {}
        if (!(newlb < newub)) die("LenDist::trim: empty distribution");
        if (newlb == 0 && newub == span) return;
        int nspan = newub - newlb;
        std::vector<double> np(nspan + 1, 0.0), nc(nspan + 1, 0.0);
        for (int i = 1; i <= nspan; i++) { np[i] = pdf[i + newlb]; nc[i] = cdf[i + newlb]; }
        pdf.swap(np); cdf.swap(nc);
        span = nspan; lb += newlb; ub = lb + span;
    }
    void finish() {  // LenDist.h:186-199
        double sum = 0.0;
        for (int i = 1; i <= span; i++) sum += pdf[i];
        if (sum <= kEpsilon) die("No valid read to estimate the length distribution!");
        for (int i = 1; i <= span; i++) { pdf[i] = pdf[i] / sum; cdf[i] = cdf[i - 1] + pdf[i]; }
        trim();
    }
    void read(FILE* fi) {  // LenDist.h:218-233
        if (fscanf(fi, "%d %d %d", &lb, &ub, &span) != 3) die("model file: bad length distribution");
        pdf.assign(span + 1, 0.0); cdf.assign(span + 1, 0.0);
        for (int i = 1; i <= span; i++) {
            if (fscanf(fi, "%lf", &pdf[i]) != 1) die("model file: bad length distribution");
            cdf[i] = cdf[i - 1] + pdf[i];
        }
        trim();
    }
    void write(FILE* fo) const {  // LenDist.h:235-241
        fprintf(fo, "%d %d %d\n", lb, ub, span);
        for (int i = 1; i < span; i++) fprintf(fo, "%.10g ", pdf[i]);
        fprintf(fo, "%.10g\n", pdf[span]);
    }
};

// ---- result arithmetic (WriteResults.h:24-104) ---------------------------------------------------

inline std::vector<double> calc_eel(int M, const RefInfo& R, const LenDist& gld) {
    std::vector<double> clen(gld.span + 1, 0.0), eel(M + 1, 0.0);
    for (int i = 1; i <= gld.span; i++) clen[i] = clen[i - 1] + gld.pdf[i] * (gld.lb + i);
    for (int i = 1; i <= M; i++) {
        int totLen = R.totLen[i], fullLen = R.fullLen[i];
        int pos1 = std::max(std::min(totLen - fullLen + 1, gld.ub) - gld.lb, 0);
        int pos2 = std::max(std::min(totLen, gld.ub) - gld.lb, 0);
        if (pos2 == 0) { eel[i] = 0.0; continue; }
        eel[i] = fullLen * gld.cdf[pos1] + ((gld.cdf[pos2] - gld.cdf[pos1]) * (totLen + 1) - (clen[pos2] - clen[pos1]));
        if (eel[i] < kMinEel) eel[i] = 0.0;
    }
    return eel;
}

inline void polish_theta(int M, std::vector<double>& theta, const std::vector<double>& eel, const double* mw) {
    double sum = 0.0;
    for (int i = 0; i <= M; i++) {
        if (i > 0 && (mw[i] < kEpsilon || eel[i] < kEpsilon)) { theta[i] = 0.0; continue; }
        theta[i] = theta[i] / mw[i];
        sum += theta[i];
    }
    if (!(sum >= kEpsilon)) die("No effective length is no less than %.6f !", kMinEel);
    for (int i = 0; i <= M; i++) theta[i] /= sum;
}

inline void calc_expression(int M, const std::vector<double>& theta, const std::vector<double>& eel,
                            std::vector<double>& tpm, std::vector<double>& fpkm) {
    double denom = 0.0;
    std::vector<double> frac(M + 1, 0.0);
    for (int i = 1; i <= M; i++)
        if (eel[i] >= kEpsilon) { frac[i] = theta[i]; denom += frac[i]; }
    if (denom < kEpsilon) denom = 1.0;
    for (int i = 1; i <= M; i++) frac[i] /= denom;
    fpkm.assign(M + 1, 0.0);
    for (int i = 1; i <= M; i++)
        if (eel[i] >= kEpsilon) fpkm[i] = frac[i] * 1e9 / eel[i];
    tpm.assign(M + 1, 0.0);
    denom = 0.0;
    for (int i = 1; i <= M; i++) denom += fpkm[i];
    if (denom < kEpsilon) denom = 1.0;
    for (int i = 1; i <= M; i++) tpm[i] = fpkm[i] / denom * 1e6;
}

// ---- Gibbs inputs -------------------------------------------------------------------------------

struct OfgData {  // imd.ofg (EM.cpp:435-457): items CSR incl. the noise column
    int M = 0;
    uint64_t N0 = 0;
    Arr<uint64_t> row_ptr;
    Arr<int32_t> sid;
    Arr<double> conprb;
};

inline OfgData load_ofg(const std::string& path) {
    MappedFile f;
    if (!f.open(path)) die("Cannot open %s!", path.c_str());
    OfgData D;
    const char* p = f.data;
    const char* end = f.data + f.size;
    long long M, N0;
    if (!parse_long(p, end, M) || !parse_long(p, end, N0)) die("%s: bad header", path.c_str());
    D.M = (int)M; D.N0 = (uint64_t)N0;
    const char* nl = (const char*)memchr(p, '\n', end - p);
    size_t body = nl ? (size_t)(nl - f.data) + 1 : f.size;
    int nt = f.size > (64u << 20) ? hardware_threads() : 1;
    std::vector<size_t> cut = line_chunks(f.data, body, f.size, nt);
    int nc = (int)cut.size() - 1;
    struct Part { std::vector<uint32_t> lens; std::vector<int32_t> sid; std::vector<double> val; };
    std::vector<Part> parts(nc);
    parallel_for(nc, [&](int c) {
        const char* q = f.data + cut[c];
        const char* e = f.data + cut[c + 1];
        Part& P = parts[c];
        while (q < e) {
            const char* le = (const char*)memchr(q, '\n', e - q);
            if (!le) le = e;
            uint32_t n = 0;
            while (q < le) {
                long long s;
                if (!parse_long(q, le, s)) break;
                char* ep;
                double v = strtod(q, &ep);  // the line ends in '\n' (or the buffer in a mapped page): strtod stops there
                if (ep == q) break;
                q = ep;
                P.sid.push_back((int32_t)s);
                P.val.push_back(v);
                ++n;
            }
            // Gibbs.cpp:121-132: every line of the file is a read, even an empty one
            P.lens.push_back(n);
            q = le + 1;
        }
    });
    std::vector<uint64_t> r0(nc + 1, 0), h0(nc + 1, 0);
    for (int c = 0; c < nc; c++) { r0[c + 1] = r0[c] + parts[c].lens.size(); h0[c + 1] = h0[c] + parts[c].sid.size(); }
    uint64_t* rp = D.row_ptr.alloc(r0[nc] + 1);
    int32_t* sp = D.sid.alloc(h0[nc]);
    double* vp = D.conprb.alloc(h0[nc]);
    rp[r0[nc]] = h0[nc];
    parallel_for(nc, [&](int c) {  // the chunks' pieces go to their final places in parallel
        Part& P = parts[c];
        uint64_t o = h0[c];
        for (size_t i = 0; i < P.lens.size(); i++) { rp[r0[c] + i] = o; o += P.lens[i]; }
        if (!P.sid.empty()) {
            memcpy(sp + h0[c], P.sid.data(), sizeof(int32_t) * P.sid.size());
            memcpy(vp + h0[c], P.val.data(), sizeof(double) * P.val.size());
        }
        P = Part();
    });
    return D;
}

// first number of each of the 4 entries of stat.cnt line 1 (EM.cpp:607-613)
inline void load_cnt(const std::string& path, uint64_t& N0, uint64_t& N1, uint64_t& N2, uint64_t& Ntot) {
    FILE* fi = fopen(path.c_str(), "r");
    if (!fi) die("Cannot open %s! It may not exist.", path.c_str());
    unsigned long long a, b, c, d;
    if (fscanf(fi, "%llu %llu %llu %llu", &a, &b, &c, &d) != 4) die("%s: bad first line", path.c_str());
    fclose(fi);
    N0 = a; N1 = b; N2 = c; Ntot = d;
}

}  // namespace rsemh
