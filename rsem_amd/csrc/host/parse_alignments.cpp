// rsem-parse-alignments -- drop-in for the reference's alignment parser (parseIt.cpp:1-260, SamParser.h:28-330):
// SAM/BAM in, the EM stage's input files out:
//
//   rsem-parse-alignments refName imdName statName alignF read_type [-t fai_file] [-tag tagName] [-q]
//
//   imdName.dat                        N1 nHits read_type / one line of hits per alignable read
//   imdName_{un,alignable,max}[_1|_2].{fa,fq}   reads by category (empty categories are removed)
//   imdName.omit                       transcripts the alignment header does not declare
//   statName.cnt                       N0 N1 N2 / nUnique nMulti nIsoMulti / nHits read_type / hits-per-read histogram
//
// Host-only stage (byte shuffling; no GPU work): it is the caller of the EM hot path, SURVEY.md section 8 "next".
// Input decoding is bam_io.hpp (zlib only, no htslib); `-t` is accepted and ignored (it only matters for CRAM).
#include <map>

#include "bam_io.hpp"
#include "files.hpp"
#include "reads.hpp"

using namespace rsemh;

namespace {

struct Rec {  // field view of one BAM record (SAM spec 4.2)
    const uint8_t* d;
    size_t n;
    int32_t tid() const { int32_t v; memcpy(&v, d, 4); return v; }
    int32_t pos() const { int32_t v; memcpy(&v, d + 4, 4); return v; }
    int l_name() const { return d[8]; }
    int n_cigar() const { return d[12] | (d[13] << 8); }
    int flag() const { return d[14] | (d[15] << 8); }
    int32_t l_seq() const { int32_t v; memcpy(&v, d + 16, 4); return v; }
    const char* qname() const { return (const char*)d + 32; }
    const uint8_t* cigar() const { return d + 32 + l_name(); }
    const uint8_t* seq() const { return cigar() + 4 * n_cigar(); }
    const uint8_t* qual() const { return seq() + (l_seq() + 1) / 2; }
    const uint8_t* aux() const { return qual() + l_seq(); }
    bool paired() const { return flag() & 1; }
    bool mapped() const { return !(flag() & 4); }
    bool rev() const { return flag() & 16; }
    bool read1() const { return flag() & 64; }
    bool read2() const { return flag() & 128; }
};

// first whitespace-delimited word of QNAME (sam_utils.h:54-61)
std::string canonical_name(const Rec& r) {
    const char* q = r.qname();
    const char* w = strpbrk(q, " \t\n\r\f\v");
    return w ? std::string(q, w - q) : std::string(q);
}

// one ungapped M/=/X operation covering the whole read (sam_utils.h:64-70)
bool check_cigar(const Rec& r) {
    uint32_t c;
    memcpy(&c, r.cigar(), 4);
    const int op = c & 15;
    return r.n_cigar() == 1 && (op == 0 || op == 7 || op == 8) && (int32_t)(c >> 4) == r.l_seq();
}

// read as sequenced: reverse-complemented back when the alignment is on the reverse strand (sam_utils.h:78-112)
void read_seq(const Rec& r, std::string& s) {
    const int L = r.l_seq();
    const uint8_t* p = r.seq();
    s.resize(L);
    const bool rev = r.rev();
    for (int i = 0; i < L; i++) {
        const int k = rev ? L - 1 - i : i;
        const int code = (p[k >> 1] >> ((~k & 1) << 2)) & 15;
        char b;
        switch (code) {
            case 1: b = rev ? 'T' : 'A'; break;
            case 2: b = rev ? 'G' : 'C'; break;
            case 4: b = rev ? 'C' : 'G'; break;
            case 8: b = rev ? 'A' : 'T'; break;
            case 15: b = 'N'; break;
            default: die("Read %s: base code %d is not one of A, C, G, T, N!", r.qname(), code);
        }
        s[i] = b;
    }
}

void read_qual(const Rec& r, std::string& s) {  // sam_utils.h:114-132
    const int L = r.l_seq();
    const uint8_t* p = r.qual();
    s.resize(L);
    if (r.rev()) for (int i = 0; i < L; i++) s[i] = (char)(p[L - 1 - i] + 33);
    else for (int i = 0; i < L; i++) s[i] = (char)(p[i] + 33);
}

// integer value of an optional field, or `absent` (bam_aux_get + bam_aux2i)
long long aux_int(const Rec& r, const char* tag, bool& found) {
    found = false;
    const uint8_t* p = r.aux();
    const uint8_t* e = r.d + r.n;
    while (p + 3 <= e) {
        const bool hit = p[0] == (uint8_t)tag[0] && p[1] == (uint8_t)tag[1];
        const char type = (char)p[2];
        p += 3;
        size_t sz = 0;
        long long v = 0;
        switch (type) {
            case 'A': case 'c': v = (int8_t)p[0]; sz = 1; break;
            case 'C': v = p[0]; sz = 1; break;
            case 's': { int16_t x; memcpy(&x, p, 2); v = x; sz = 2; break; }
            case 'S': { uint16_t x; memcpy(&x, p, 2); v = x; sz = 2; break; }
            case 'i': { int32_t x; memcpy(&x, p, 4); v = x; sz = 4; break; }
            case 'I': { uint32_t x; memcpy(&x, p, 4); v = x; sz = 4; break; }
            case 'f': sz = 4; break;
            case 'd': sz = 8; break;
            case 'Z': case 'H': sz = strlen((const char*)p) + 1; break;
            case 'B': {
                const char sub = (char)p[0];
                int32_t cnt; memcpy(&cnt, p + 1, 4);
                const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                sz = 5 + w * (size_t)cnt;
                break;
            }
            default: return 0;  // malformed: stop scanning
        }
        if (hit) { found = type != 'f' && type != 'd' && type != 'Z' && type != 'H' && type != 'B'; return v; }
        p += sz;
    }
    return 0;
}

struct Mate {
    std::string name, seq, qual;
    int len = 0;
};

struct Out {  // one buffered read file
    FILE* f = nullptr;
    std::string path;
};

void write_mate(FILE* f, const Mate& m, bool has_q) {  // SingleRead.h:52-55, SingleReadQ.h:57-60
    fputc(has_q ? '@' : '>', f);
    fwrite(m.name.data(), 1, m.name.size(), f);
    fputc('\n', f);
    fwrite(m.seq.data(), 1, m.seq.size(), f);
    if (has_q) {
        fputs("\n+\n", f);
        fwrite(m.qual.data(), 1, m.qual.size(), f);
    }
    fputc('\n', f);
}

}  // namespace

int main(int argc, char* argv[]) {
    if (argc < 6) {
        printf("Usage : rsem-parse-alignments refName imdName statName alignF read_type [-t fai_file] [-tag tagName] [-q]\n");
        exit(-1);
    }
    const std::string refName = argv[1], imdName = argv[2], statName = argv[3], alignF = argv[4];
    const int read_type = atoi(argv[5]);
    if (read_type < 0 || read_type > 3) die("read_type must be 0, 1, 2 or 3!");
    bool verbose = true;
    std::string rt_tag;
    for (int i = 6; i < argc; i++) {
        if (!strcmp(argv[i], "-tag") && i + 1 < argc) rt_tag = argv[i + 1];
        if (!strcmp(argv[i], "-q")) verbose = false;
    }
    const bool paired = read_type >= 2, has_q = read_type & 1;

    GroupInfo gi;
    if (!gi.load(refName + ".grp")) die("Cannot open %s.grp! It may not exist.", refName.c_str());
    std::vector<int32_t> gid_of(gi.starts.empty() ? 1 : gi.starts.back() + 1, 0);
    for (int g = 0; g < gi.m; g++)
        for (int s = gi.starts[g]; s < gi.starts[g + 1]; s++) gid_of[s] = g;
    const Transcripts T = load_transcripts(refName + ".ti");
    const int M = T.M;

    AlignmentReader in;
    in.open(alignF);

    // external (header order) -> internal transcript ids, and the .omit list (Transcripts.h:105-143)
    const int n_targets = (int)in.header.names.size();
    if (n_targets <= 0) die("The SAM/BAM file declares less than one reference sequence!");
    if (n_targets > M) die("The SAM/BAM file declares more reference sequences (%d) than RSEM knows (%d)!", n_targets, M);
    if (n_targets < M)
        fprintf(stderr, "Warning: The SAM/BAM file declares less reference sequences (%d) than RSEM knows (%d)! Please make sure that you aligned your reads against transcript sequences instead of genome.\n", n_targets, M);
    std::vector<int32_t> e2i(n_targets, 0);
    {
        std::map<std::string, int> dict;
        for (int i = 1; i <= M; i++) {
            const std::string& tid = T.type == 2 ? T.t[i].seqname : T.t[i].transcript_id;
            if (!dict.emplace(tid, i).second) die("RSEM's indices might be corrupted, %s appears more than once!", tid.c_str());
        }
        std::vector<char> appeared(M + 1, 0);
        for (int i = 0; i < n_targets; i++) {
            auto it = dict.find(in.header.names[i]);
            if (it == dict.end()) die("RSEM can not recognize reference sequence name %s!", in.header.names[i].c_str());
            if (it->second <= 0) die("Reference sequence name %s appears more than once in the SAM/BAM file!", in.header.names[i].c_str());
            e2i[i] = it->second;
            appeared[it->second] = 1;
            it->second = -1;
        }
        FILE* fo = fopen((imdName + ".omit").c_str(), "w");
        if (!fo) die("Cannot open %s.omit for writing!", imdName.c_str());
        for (int i = 1; i <= M; i++)
            if (!appeared[i]) fprintf(fo, "%d\n", i);
        fclose(fo);
    }

    // category files: 0 unalignable, 1 alignable, 2 filtered ("max") (utils.h genReadFileNames)
    const int n_os = paired ? 2 : 1;
    Out cat[3][2];
    for (int c = 0; c < 3; c++) {
        const std::vector<std::string> names = read_file_names(imdName, c, read_type);
        for (int j = 0; j < n_os; j++) {
            cat[c][j].path = names[j];
            cat[c][j].f = fopen(names[j].c_str(), "w");
            if (!cat[c][j].f) die("Cannot open %s for writing!", names[j].c_str());
            setvbuf(cat[c][j].f, nullptr, _IOFBF, 1 << 22);
        }
    }
    FILE* fdat = fopen((imdName + ".dat").c_str(), "w");
    if (!fdat) die("Cannot open %s.dat for writing!", imdName.c_str());
    setvbuf(fdat, nullptr, _IOFBF, 1 << 22);
    fprintf(fdat, "%-99s\n", "");  // patched once the totals are known (parseIt.cpp:195-204)

    long long N[3] = {0, 0, 0}, nHits = 0, nMulti = 0, nIsoMulti = 0, cnt = 0;
    std::map<long long, long long> counter;
    int n_warns = 0;

    auto read_category = [&](const Rec& a, const Rec* b) -> int {  // SamParser.h:61-82
        if (a.mapped() && (!b || b->mapped())) return 1;
        if (rt_tag.empty()) return 0;
        bool found;
        long long v = aux_int(a, rt_tag.c_str(), found);
        if (found && v > 0) return 2;
        if (b) {
            v = aux_int(*b, rt_tag.c_str(), found);
            if (found && v > 0) return 2;
        }
        return 0;
    };

    Mate cur[2];        // the read whose hits are being collected
    int cur_val = -2;   // its category; -2 = none yet
    std::string hits;   // its .dat line body
    std::vector<int32_t> hit_gids;
    char tmp[64];

    auto flush = [&]() {  // parseIt.cpp:92-118
        if (cur_val >= 0) {
            for (int j = 0; j < n_os; j++) write_mate(cat[cur_val][j].f, cur[j], has_q);
            ++N[cur_val];
        }
        if (cur_val != 1 && !hit_gids.empty()) die("Read %s is both unalignable and alignable according to the input file!", cur[0].name.c_str());
        if (cur_val == 1) {
            const long long k = (long long)hit_gids.size();
            nHits += k;
            std::sort(hit_gids.begin(), hit_gids.end());
            if (std::unique(hit_gids.begin(), hit_gids.end()) - hit_gids.begin() > 1) ++nMulti;
            if (k > 1) ++nIsoMulti;
            if (k > 0) {  // HitContainer::updateRI drops a read without hits (cannot happen for category 1)
                int n = snprintf(tmp, sizeof(tmp), "%lld", k);
                fwrite(tmp, 1, n, fdat);
                fwrite(hits.data(), 1, hits.size(), fdat);
                fputc('\n', fdat);
            }
            ++counter[k];
        }
        hits.clear();
        hit_gids.clear();
    };

    AlnRecord ra, rb;
    std::string name, name2;
    for (;;) {
        if (!in.next(ra)) break;
        if (paired && !in.next(rb)) break;
        Rec a{ra.d.data(), ra.d.size()}, b{rb.d.data(), rb.d.size()};
        if (paired && !a.read1()) std::swap(a, b);
        name = canonical_name(a);
        if (!paired) {
            if (a.paired()) die("Read %s: Find a paired end read in the file!", name.c_str());
        } else {
            if (!(a.paired() && b.paired())) die("Read %s: One of the mate is not paired-end! (RSEM assumes the two mates of a paired-end read should be adjacent)", name.c_str());
            if (!(a.read1() && b.read2())) die("Read %s: The adjacent two lines do not represent the two mates of a paired-end read! (RSEM assumes the two mates of a paired-end read should be adjacent)", name.c_str());
            if (a.mapped() != b.mapped()) die("Read %s: RSEM currently does not support partial alignments!", name.c_str());
            name2 = canonical_name(b);
            if (name != name2 && ++n_warns <= 50)
                fprintf(stderr, "Warning: Detected a read pair whose two mates have different names--%s and %s!\n", name.c_str(), name2.c_str());
        }
        const int rt = read_category(a, paired ? &b : nullptr);
        if (rt != 1 || cur[0].name != name) {  // a new read (SamParser.h:128-131); before the first one the name is ""
            flush();
            cur_val = rt;
            cur[0].name = name;
            cur[0].len = a.l_seq();
            read_seq(a, cur[0].seq);
            if (has_q) read_qual(a, cur[0].qual);
            if (paired) {
                cur[1].name = name2;
                cur[1].len = b.l_seq();
                read_seq(b, cur[1].seq);
                if (has_q) read_qual(b, cur[1].qual);
            }
        } else {
            if (!paired) {
                if (cur[0].len != a.l_seq()) die("Read %s has alignments with inconsistent read lengths!", name.c_str());
            } else if (cur[0].len != a.l_seq() || cur[1].len != b.l_seq())
                die("Paired-end read %s has alignments with inconsistent mate lengths!", name.c_str());
        }
        if (rt == 1) {
            if (!check_cigar(a) || (paired && !check_cigar(b))) die("Read %s: RSEM currently does not support gapped alignments, sorry!\n", name.c_str());
            if (paired && a.tid() != b.tid()) die("Read %s: The two mates do not align to a same transcript! RSEM does not support discordant alignments.", name.c_str());
            const int tid = a.tid();
            if (tid < 0 || tid >= n_targets) die("Read %s: alignment to an undeclared reference sequence!", name.c_str());
            const int sid = e2i[tid];
            const int tlen = in.header.lens[tid];
            int n;
            if (!paired) {
                if (a.rev()) n = snprintf(tmp, sizeof(tmp), " %d %d", -sid, tlen - a.pos() - a.l_seq());
                else n = snprintf(tmp, sizeof(tmp), " %d %d", sid, a.pos());
            } else {
                if (a.rev()) n = snprintf(tmp, sizeof(tmp), " %d %d %d", -sid, tlen - a.pos() - a.l_seq(), a.pos() + a.l_seq() - b.pos());
                else n = snprintf(tmp, sizeof(tmp), " %d %d %d", sid, a.pos(), b.pos() + b.l_seq() - a.pos());
            }
            hits.append(tmp, n);
            hit_gids.push_back(gid_of[sid]);
        }
        ++cnt;
        if (verbose && cnt % 1000000 == 0) { printf("Parsed %lld entries\n", cnt); fflush(stdout); }
    }
    flush();
    if (n_warns > 0) fprintf(stderr, "Warning: Detected %d lines containing read pairs whose two mates have different names.\n", n_warns);
    const long long nUnique = N[1] - nMulti;

    fflush(fdat);
    fseek(fdat, 0, SEEK_SET);
    fprintf(fdat, "%lld %lld %d", N[1], nHits, read_type);
    fclose(fdat);

    FILE* fc = fopen((statName + ".cnt").c_str(), "w");
    if (!fc) die("Cannot open %s.cnt for writing!", statName.c_str());
    fprintf(fc, "%lld %lld %lld %lld\n", N[0], N[1], N[2], N[0] + N[1] + N[2]);
    fprintf(fc, "%lld %lld %lld\n", nUnique, nMulti, nIsoMulti);
    fprintf(fc, "%lld %d\n", nHits, read_type);
    fprintf(fc, "0\t%lld\n", N[0]);
    for (auto& kv : counter) fprintf(fc, "%lld\t%lld\n", kv.first, kv.second);
    fprintf(fc, "Inf\t%lld\n", N[2]);
    fclose(fc);

    for (int c = 0; c < 3; c++)
        for (int j = 0; j < n_os; j++) {
            fclose(cat[c][j].f);
            if (N[c] == 0) remove(cat[c][j].path.c_str());
        }
    if (verbose) printf("Done!\n");
    return 0;
}
