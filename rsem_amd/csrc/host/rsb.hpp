// rsb.hpp -- the binary hand-off between rsem-parse-alignments and rsem-run-em.
//
// The reference passes the parsed alignments and the reads to the EM stage as TEXT: imdName.dat (one line of
// "sid pos [insertL]" triples per read, written by parseIt.cpp:139-147 / HitWrapper, read back with istream >> in
// HitContainer.h:63-79) and FASTA/FASTQ files per read category (utils.h:129-149), which rsem-run-em parses again in
// every one of its first rounds.  Once the EM itself takes seconds, printing and re-parsing tens of GB of decimal text
// is what is left.  imdName.rsb/ holds the same information as the arrays the device wants, so that rsem-run-em maps
// the files and uploads them:
//
//   hdr                       RsbHeader (written last: a directory without it is incomplete)
//   row_ptr   u64[N1+1]       first alignment of every alignable read (the CSR of HitContainer::s)
//   sid       i32[nHits]      transcript id, negative = reverse strand (SingleHit.h:24-26)
//   pos       i32[nHits]
//   ins       i32[nHits]      fragment length, paired-end only (PairedEndHit.h:8-34)
//   off_<c>_<m> u64[N_c+1]    per read category c (0 unalignable, 1 alignable, 2 filtered) and mate m: base offsets
//   seq_<c>_<m> u8[...]       base ids A C G T N = 0..4 (utils.h:36-50), 255 = a letter get_base_id rejects
//   qual_<c>_<m> u8[...]      quality character - 33 (QProfile.h:44), FASTQ models only
//
// All files are plain little-endian arrays; everything is appended in read order, so the writer needs no seeking.
// Read names are not stored: the EM stage never uses them (error messages fall back to the read's index).
#pragma once
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include "reads.hpp"

namespace rsemh {

struct RsbHeader {
    char magic[8];       // "RSEMRSB1"
    uint32_t version;    // 1
    uint32_t read_type;  // 0..3
    uint64_t N[3];       // reads per category
    uint64_t nHits;
    uint64_t nBases[3][2];
};

inline std::string rsb_dir(const std::string& imdName) { return imdName + ".rsb"; }

// remove imdName.rsb/ and the plain files in it (the writer creates nothing else there); no shell involved
inline void remove_rsb(const std::string& imdName) {
    const std::string dir = rsb_dir(imdName);
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d)) {
            if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
            unlink((dir + "/" + e->d_name).c_str());
        }
        closedir(d);
        rmdir(dir.c_str());
    }
}

class RsbWriter {
  public:
    RsbWriter(const std::string& imdName, int read_type) : dir_(rsb_dir(imdName)), read_type_(read_type) {
        const bool pe = read_type >= 2, q = (read_type == 1 || read_type == 3);
        remove_rsb(imdName);
        if (mkdir(dir_.c_str(), 0777) != 0) die("Cannot create %s!", dir_.c_str());
        memset(&h_, 0, sizeof(h_));
        f_rp_ = open_("row_ptr"); f_sid_ = open_("sid"); f_pos_ = open_("pos");
        if (pe) f_ins_ = open_("ins");
        for (int c = 0; c < 3; c++)
            for (int m = 0; m < (pe ? 2 : 1); m++) {
                const std::string t = "_" + std::to_string(c) + "_" + std::to_string(m);
                f_off_[c][m] = open_("off" + t);
                f_seq_[c][m] = open_("seq" + t);
                if (q) f_qual_[c][m] = open_("qual" + t);
                const uint64_t zero = 0;
                put_(&zero, sizeof(zero), 1, f_off_[c][m]);
            }
        const uint64_t zero = 0;
        put_(&zero, sizeof(zero), 1, f_rp_);
    }
    // alignments of consecutive alignable reads: lens[n] alignments each
    void append_hits(const uint32_t* lens, size_t n, const int32_t* sid, const int32_t* pos, const int32_t* ins) {
        tmp_.resize(n);
        const uint64_t before = h_.nHits;
        for (size_t i = 0; i < n; i++) { h_.nHits += lens[i]; tmp_[i] = h_.nHits; }
        const size_t k = (size_t)(h_.nHits - before);
        put_(tmp_.data(), sizeof(uint64_t), n, f_rp_);
        put_(sid, sizeof(int32_t), k, f_sid_);
        put_(pos, sizeof(int32_t), k, f_pos_);
        if (f_ins_) put_(ins, sizeof(int32_t), k, f_ins_);
    }
    // consecutive reads of category c, mate m: lens[n] bases each, packed back to back
    void append_reads(int c, int m, const uint32_t* lens, size_t n, const uint8_t* seq, const uint8_t* qual) {
        tmp_.resize(n);
        uint64_t tot = 0;
        for (size_t i = 0; i < n; i++) { tot += lens[i]; tmp_[i] = h_.nBases[c][m] + tot; }
        put_(tmp_.data(), sizeof(uint64_t), n, f_off_[c][m]);
        put_(seq, 1, tot, f_seq_[c][m]);
        if (f_qual_[c][m]) put_(qual, 1, tot, f_qual_[c][m]);
        h_.nBases[c][m] += tot;
        if (m == 0) h_.N[c] += n;
    }
    void finish() {
        for (FILE* f : {f_rp_, f_sid_, f_pos_, f_ins_})
            if (f && fclose(f) != 0) die("Cannot write %s!", dir_.c_str());
        for (int c = 0; c < 3; c++)
            for (int m = 0; m < 2; m++)
                for (FILE* f : {f_off_[c][m], f_seq_[c][m], f_qual_[c][m]})
                    if (f && fclose(f) != 0) die("Cannot write %s!", dir_.c_str());
        memcpy(h_.magic, "RSEMRSB1", 8);
        h_.version = 1;
        h_.read_type = (uint32_t)read_type_;
        FILE* f = open_("hdr");
        put_(&h_, sizeof(h_), 1, f);
        if (fclose(f) != 0) die("Cannot write %s/hdr!", dir_.c_str());
    }
    const RsbHeader& header() const { return h_; }

  private:
    void put_(const void* p, size_t size, size_t n, FILE* f) {  // a full disk is an error here, not a surprise at load time
        if (n && fwrite(p, size, n, f) != n) die("Cannot write %s (disk full?)!", dir_.c_str());
    }
    FILE* open_(const std::string& name) {
        FILE* f = fopen((dir_ + "/" + name).c_str(), "wb");
        if (!f) die("Cannot open %s/%s for writing!", dir_.c_str(), name.c_str());
        setvbuf(f, nullptr, _IOFBF, 1 << 22);
        return f;
    }
    std::string dir_;
    int read_type_;
    RsbHeader h_;
    std::vector<uint64_t> tmp_;
    FILE *f_rp_ = nullptr, *f_sid_ = nullptr, *f_pos_ = nullptr, *f_ins_ = nullptr;
    FILE* f_off_[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    FILE* f_seq_[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    FILE* f_qual_[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
};

// The binary hand-off is used when its header exists AND it is not older than a text hand-off lying next to it: a kept
// sample.temp directory may hold arrays of an earlier --binary run beside the .dat of a later text run of another parser
// (this repo's parser removes the stale directory itself; the reference's does not know about it).
inline bool rsb_present(const std::string& imdName) {
    struct stat hb, db;
    if (stat((rsb_dir(imdName) + "/hdr").c_str(), &hb) != 0) return false;
    if (stat((imdName + ".dat").c_str(), &db) != 0) return true;
    const bool newer_dat = db.st_mtim.tv_sec > hb.st_mtim.tv_sec || (db.st_mtim.tv_sec == hb.st_mtim.tv_sec && db.st_mtim.tv_nsec > hb.st_mtim.tv_nsec);
    if (newer_dat) fprintf(stderr, "Warning: %s.dat is newer than %s: the text files are used.\n", imdName.c_str(), rsb_dir(imdName).c_str());
    return !newer_dat;
}

struct ReadSetFiles {  // the three categories of reads (utils.h:129-149): un, alignable, max
    ReadFile mate[3][2];
    bool present[3] = {false, false, false};
};

template <typename T>
inline void rsb_map(const std::string& path, size_t count, Arr<T>& a) {
    auto f = std::make_shared<MappedFile>();
    if (!f->open(path)) die("Cannot open %s! It may not exist.", path.c_str());
    if (f->size != count * sizeof(T)) die("%s holds %zu bytes, the header promises %zu!", path.c_str(), f->size, count * sizeof(T));
    a.view((const T*)f->data, count, f);
}

// Map imdName.rsb/ into the structures the text parsers fill (load_dat, parse_read_file).  The per-read low-quality
// flags (SingleRead::calc_lq, needs seedLen and the reference's poly(A) setting) and the validity checks the text path
// makes while parsing (unknown base letters, quality characters) are one parallel pass over the bases.
inline void load_rsb(const std::string& imdName, int read_type, bool hasPolyA, int seedLen, DatData& D, ReadSetFiles& rs) {
    const std::string dir = rsb_dir(imdName);
    RsbHeader h;
    {
        MappedFile f;
        if (!f.open(dir + "/hdr") || f.size != sizeof(RsbHeader)) die("%s/hdr is missing or damaged!", dir.c_str());
        memcpy(&h, f.data, sizeof(h));
    }
    if (memcmp(h.magic, "RSEMRSB1", 8) != 0 || h.version != 1) die("%s is not an RSEM binary hand-off directory of version 1!", dir.c_str());
    if ((int)h.read_type != read_type) die("Data file (.dat) does not have the right read type!");
    const bool pe = read_type >= 2, q = (read_type == 1 || read_type == 3);
    D.N1 = h.N[1]; D.nHits = h.nHits; D.read_type = read_type;
    rsb_map(dir + "/row_ptr", (size_t)h.N[1] + 1, D.row_ptr);
    rsb_map(dir + "/sid", (size_t)h.nHits, D.sid_signed);
    rsb_map(dir + "/pos", (size_t)h.nHits, D.pos);
    if (pe) rsb_map(dir + "/ins", (size_t)h.nHits, D.insertL);
    if (D.row_ptr[0] != 0 || D.row_ptr[h.N[1]] != h.nHits) die("%s/row_ptr does not match the header!", dir.c_str());
    for (uint64_t i = 0; i < h.N[1]; i++)
        if (D.row_ptr[i + 1] < D.row_ptr[i]) die("%s/row_ptr is damaged (not increasing at read %llu)!", dir.c_str(), (unsigned long long)i);
    for (int c = 0; c < 3; c++) {
        if (h.N[c] == 0) continue;
        rs.present[c] = true;
        for (int m = 0; m < (pe ? 2 : 1); m++) {
            ReadFile& R = rs.mate[c][m];
            const std::string t = "_" + std::to_string(c) + "_" + std::to_string(m);
            R.n = h.N[c];
            rsb_map(dir + "/off" + t, (size_t)h.N[c] + 1, R.off);
            rsb_map(dir + "/seq" + t, (size_t)h.nBases[c][m], R.seq);
            if (q) rsb_map(dir + "/qual" + t, (size_t)h.nBases[c][m], R.qual);
            if (R.off[0] != 0 || R.off[R.n] != h.nBases[c][m]) die("%s/off%s does not match the header!", dir.c_str(), t.c_str());
            R.lq1.assign(R.n, 0);
            const int nt = R.n > 200000 ? hardware_threads() : 1;
            std::vector<int> bad(nt, 0);
            parallel_for(nt, [&](int th) {
                const uint64_t lo = R.n * th / nt, hi = R.n * (th + 1) / nt;
                for (uint64_t i = lo; i < hi; i++) {
                    if (R.off[i + 1] < R.off[i] || R.off[i + 1] > h.nBases[c][m] || R.off[i + 1] - R.off[i] > 0x7fffffffull) { bad[th] = 3; break; }
                    const uint8_t* s = R.seq.data() + R.off[i];
                    const int len = R.len(i);
                    for (int k = 0; k < len; k++)
                        if (s[k] > 4) bad[th] = 1;
                    if (q) {
                        const uint8_t* ql = R.qual.data() + R.off[i];
                        for (int k = 0; k < len; k++)
                            if (ql[k] > 93) bad[th] = 2;
                    }
                    R.lq1[i] = calc_lq_single_ids(s, len, hasPolyA, seedLen) ? 1 : 0;
                }
            });
            for (int b : bad) {
                if (b == 1) die("Found unknown sequence letter at function get_base_id! (%s/seq%s)", dir.c_str(), t.c_str());
                if (b == 2) die("%s/qual%s: quality character out of range", dir.c_str(), t.c_str());
                if (b == 3) die("%s/off%s is damaged (offsets not increasing or past the end)!", dir.c_str(), t.c_str());
            }
        }
    }
}

}  // namespace rsemh
