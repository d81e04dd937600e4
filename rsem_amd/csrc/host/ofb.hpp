// ofb.hpp -- the binary hand-off between rsem-run-em --gibbs-out and rsem-run-gibbs.
//
// The reference passes the Gibbs sampler's input as TEXT: imdName.ofg, one line per read of "sid conprb" pairs printed
// with 15 significant digits (EM.cpp:421-458), read back with istream >> (Gibbs.cpp:101-137).  At BASELINE configs[2] /
// configs[3] size that is 16.5 GB of decimal text: 12.8 s to print and about as long to parse again, between two programs
// whose own work takes seconds.  imdName.ofb/ holds the same items as arrays:
//
//   hdr               OfbHeader (written last: a directory without it is incomplete)
//   row_ptr  u64[N1+1]   first item of every read that has one (a read without items has no line in .ofg either)
//   sid      i32[n]      0 = the noise transcript, first in its read
//   val      f64[n]      EXACTLY the doubles a reader of the text file would get: every value has gone through the
//                        15-digit decimal form (printed, parsed back), so the chains drawn from .ofb are the chains
//                        drawn from .ofg, bit for bit
#pragma once
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <charconv>

#include "files.hpp"

namespace rsemh {

struct OfbHeader {
    char magic[8];  // "RSEMOFB1"
    uint32_t version;
    int32_t M;
    uint64_t N0, N1, nitems;
};

inline std::string ofb_dir(const std::string& imdName) { return imdName + ".ofb"; }

inline void remove_ofb(const std::string& imdName) {
    const std::string dir = ofb_dir(imdName);
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d)) {
            if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
            unlink((dir + "/" + e->d_name).c_str());
        }
        closedir(d);
        rmdir(dir.c_str());
    }
}

// what `is >> double` makes of the text `os << setprecision(15) << v` wrote
inline double through_15_digits(double v) {
    char tmp[40];
    auto r = std::to_chars(tmp, tmp + sizeof(tmp), v, std::chars_format::general, 15);
    double out = v;
    std::from_chars(tmp, r.ptr, out);
    return out;
}

// One writer thread's share of the reads, already filtered and rounded; the pieces are concatenated in thread order.
struct OfbPart {
    std::vector<uint32_t> lens;
    std::vector<int32_t> sid;
    std::vector<double> val;
};

// The arrays are written by write_ofb_arrays, the header by write_ofb_header: the header is the completeness marker AND
// the time stamp ofb_present() compares with a .ofg beside it, so a writer of both hand-offs writes the header LAST,
// after the text file is closed (otherwise the text would always be the newer one and the arrays would never be used).
inline OfbHeader write_ofb_arrays(const std::string& imdName, int M, uint64_t N0, std::vector<OfbPart>& parts) {
    const std::string dir = ofb_dir(imdName);
    remove_ofb(imdName);
    if (mkdir(dir.c_str(), 0777) != 0) die("Cannot create %s!", dir.c_str());
    const int nt = (int)parts.size();
    std::vector<uint64_t> r0(nt + 1, 0), h0(nt + 1, 0);
    for (int t = 0; t < nt; t++) { r0[t + 1] = r0[t] + parts[t].lens.size(); h0[t + 1] = h0[t] + parts[t].sid.size(); }
    auto open_w = [&](const char* name) {
        const int fd = ::open((dir + "/" + name).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) die("Cannot open %s/%s for writing!", dir.c_str(), name);
        return fd;
    };
    const int f_rp = open_w("row_ptr"), f_sid = open_w("sid"), f_val = open_w("val");
    std::vector<char> okv(nt, 1);
    auto put = [&](int fd, const void* p, uint64_t bytes, uint64_t off) -> bool {
        const char* q = (const char*)p;
        while (bytes > 0) {
            const ssize_t k = ::pwrite(fd, q, (size_t)std::min<uint64_t>(bytes, (uint64_t)1 << 30), (off_t)off);
            if (k <= 0) return false;
            q += k; off += (uint64_t)k; bytes -= (uint64_t)k;
        }
        return true;
    };
    parallel_for(nt, [&](int t) {
        OfbPart& P = parts[t];
        std::vector<uint64_t> rp(P.lens.size());
        uint64_t o = h0[t];
        for (size_t i = 0; i < P.lens.size(); i++) { rp[i] = o; o += P.lens[i]; }
        okv[t] = put(f_rp, rp.data(), rp.size() * 8, r0[t] * 8) && put(f_sid, P.sid.data(), P.sid.size() * 4, h0[t] * 4) &&
                 put(f_val, P.val.data(), P.val.size() * 8, h0[t] * 8);
        P = OfbPart();
    });
    bool ok = put(f_rp, &h0[nt], 8, r0[nt] * 8);
    for (char o : okv) ok = ok && o;
    if (::close(f_rp) != 0 || ::close(f_sid) != 0 || ::close(f_val) != 0 || !ok) die("Cannot write %s (disk full?)!", dir.c_str());
    OfbHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "RSEMOFB1", 8);
    h.version = 1; h.M = M; h.N0 = N0; h.N1 = r0[nt]; h.nitems = h0[nt];
    return h;
}

inline void write_ofb_header(const std::string& imdName, const OfbHeader& h) {
    const std::string path = ofb_dir(imdName) + "/hdr";
    const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) die("Cannot open %s for writing!", path.c_str());
    const bool ok = ::pwrite(fd, &h, sizeof(h), 0) == (ssize_t)sizeof(h);
    if (::close(fd) != 0 || !ok) die("Cannot write %s!", path.c_str());
}

inline void write_ofb(const std::string& imdName, int M, uint64_t N0, std::vector<OfbPart>& parts) {
    write_ofb_header(imdName, write_ofb_arrays(imdName, M, N0, parts));
}

// used when the header exists and is not older than a text file lying next to it (a kept sample.temp may hold the arrays
// of an earlier run beside the .ofg of a later one written by the reference's rsem-run-em)
inline bool ofb_present(const std::string& imdName) {
    struct stat hb, tb;
    if (stat((ofb_dir(imdName) + "/hdr").c_str(), &hb) != 0) return false;
    if (stat((imdName + ".ofg").c_str(), &tb) != 0) return true;
    const bool newer_text = tb.st_mtim.tv_sec > hb.st_mtim.tv_sec || (tb.st_mtim.tv_sec == hb.st_mtim.tv_sec && tb.st_mtim.tv_nsec > hb.st_mtim.tv_nsec);
    if (newer_text) fprintf(stderr, "Warning: %s.ofg is newer than %s: the text file is used.\n", imdName.c_str(), ofb_dir(imdName).c_str());
    return !newer_text;
}

template <typename T>
inline void ofb_map(const std::string& path, size_t count, Arr<T>& a) {
    auto f = std::make_shared<MappedFile>();
    if (!f->open(path)) die("Cannot open %s! It may not exist.", path.c_str());
    if (f->size != count * sizeof(T)) die("%s holds %zu bytes, the header promises %zu!", path.c_str(), f->size, count * sizeof(T));
    a.view((const T*)f->data, count, f);
}

inline OfgData load_ofb(const std::string& imdName) {
    const std::string dir = ofb_dir(imdName);
    OfbHeader h;
    {
        MappedFile f;
        if (!f.open(dir + "/hdr") || f.size != sizeof(OfbHeader)) die("%s/hdr is missing or damaged!", dir.c_str());
        memcpy(&h, f.data, sizeof(h));
    }
    if (memcmp(h.magic, "RSEMOFB1", 8) != 0 || h.version != 1) die("%s is not an RSEM Gibbs hand-off directory of version 1!", dir.c_str());
    OfgData D;
    D.M = h.M;
    D.N0 = h.N0;
    ofb_map(dir + "/row_ptr", (size_t)h.N1 + 1, D.row_ptr);
    ofb_map(dir + "/sid", (size_t)h.nitems, D.sid);
    ofb_map(dir + "/val", (size_t)h.nitems, D.conprb);
    if (D.row_ptr[0] != 0 || D.row_ptr[h.N1] != h.nitems) die("%s/row_ptr does not match the header!", dir.c_str());
    for (uint64_t i = 0; i < h.N1; i++)
        if (D.row_ptr[i + 1] < D.row_ptr[i]) die("%s/row_ptr is damaged (not increasing at read %llu)!", dir.c_str(), (unsigned long long)i);
    return D;
}

}  // namespace rsemh
