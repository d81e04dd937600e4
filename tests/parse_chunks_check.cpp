// parse_chunks_check.cpp -- TEST INFRASTRUCTURE: the text parsers of rsem_amd/csrc/host/reads.hpp (three scans, every thread a run
// of whole records written straight into place) must give the same arrays whatever the number of threads.  Writes small read
// files / .dat files with the features that stress the chunk boundaries -- reads of very different lengths, CRLF line ends,
// a last line without a newline, stray empty lines at the end -- and parses each with 1, 2, 3, 7 and 16 threads
// (RSEM_HIP_PARSE_SPLIT_BYTES=64 in the environment makes the parsers split even these files).  exit 0 = all equal.
#include <random>

#include "../rsem_amd/csrc/host/reads.hpp"

using namespace rsemh;

static bool same(const ReadFile& a, const ReadFile& b) {
    if (a.n != b.n || a.seq.size() != b.seq.size() || a.qual.size() != b.qual.size() || a.lq1 != b.lq1) return false;
    for (uint64_t i = 0; i <= a.n; i++) if (a.off[i] != b.off[i]) return false;
    if (memcmp(a.seq.data(), b.seq.data(), a.seq.size())) return false;
    if (a.qual.size() && memcmp(a.qual.data(), b.qual.data(), a.qual.size())) return false;
    return true;
}
static bool same(const DatData& a, const DatData& b) {
    if (a.N1 != b.N1 || a.sid_signed.size() != b.sid_signed.size() || a.insertL.size() != b.insertL.size()) return false;
    for (uint64_t i = 0; i <= a.N1; i++) if (a.row_ptr[i] != b.row_ptr[i]) return false;
    for (size_t j = 0; j < a.sid_signed.size(); j++)
        if (a.sid_signed[j] != b.sid_signed[j] || a.pos[j] != b.pos[j] || (a.insertL.size() && a.insertL[j] != b.insertL[j])) return false;
    return true;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    std::mt19937_64 rng(11);
    int bad = 0;
    const char L[4] = {'A', 'C', 'G', 'T'};
    for (int variant = 0; variant < 8; variant++) {
        const bool fastq = variant & 1, crlf = variant & 2, ragged_end = variant & 4;
        const std::string path = dir + "/reads" + std::to_string(variant) + (fastq ? ".fq" : ".fa");
        const int n = 157 + variant;
        FILE* f = fopen(path.c_str(), "w");
        const char* nl = crlf ? "\r\n" : "\n";
        for (int i = 0; i < n; i++) {
            const int len = 1 + (int)(rng() % (i % 13 == 0 ? 400 : 40));
            std::string s(len, 'A'), q(len, 'I');
            for (int k = 0; k < len; k++) { s[k] = rng() % 50 == 0 ? 'N' : L[rng() & 3]; q[k] = (char)(33 + rng() % 60); }
            const bool last = i == n - 1;
            if (fastq) fprintf(f, "@r%d%s%s%s+%s%s%s", i, nl, s.c_str(), nl, nl, q.c_str(), (last && ragged_end) ? "" : nl);
            else fprintf(f, ">r%d some description%s%s%s", i, nl, s.c_str(), (last && ragged_end) ? "" : nl);
        }
        if (!ragged_end && variant >= 2) fprintf(f, "%s%s", nl, nl);  // stray empty lines at the end
        fclose(f);
        const ReadFile one = parse_read_file(path, fastq, false, 25, 1);
        if (one.n != (uint64_t)n) { printf("variant %d: %llu reads with one thread, %d written\n", variant, (unsigned long long)one.n, n); ++bad; }
        for (int nt : {2, 3, 7, 16}) {
            const ReadFile many = parse_read_file(path, fastq, false, 25, nt);
            if (!same(one, many)) { printf("read file variant %d: %d threads differ from one\n", variant, nt); ++bad; }
        }
    }
    for (int pe = 0; pe < 2; pe++) {
        const std::string path = dir + "/hits" + std::to_string(pe) + ".dat";
        const int n = 211;
        std::string body;
        uint64_t hits = 0;
        for (int i = 0; i < n; i++) {
            const int k = 1 + (int)(rng() % (i % 17 == 0 ? 120 : 6));
            body += std::to_string(k);
            for (int t = 0; t < k; t++) {
                body += " " + std::to_string((int)(rng() % 5000 + 1) * ((rng() & 1) ? 1 : -1)) + " " + std::to_string((int)(rng() % 3000));
                if (pe) body += " " + std::to_string((int)(rng() % 400 + 50));
            }
            body += "\n";
            hits += k;
        }
        FILE* f = fopen(path.c_str(), "w");
        char head[128];
        snprintf(head, sizeof(head), "%d %llu %d", n, (unsigned long long)hits, pe ? 3 : 1);
        fprintf(f, "%-99s\n%s", head, body.c_str());
        fclose(f);
        const DatData one = load_dat(path, pe ? 3 : 1, 1);
        if (one.sid_signed.size() != hits) { printf(".dat %d: %zu alignments parsed, %llu written\n", pe, one.sid_signed.size(), (unsigned long long)hits); ++bad; }
        for (int nt : {2, 3, 7, 16}) {
            const DatData many = load_dat(path, pe ? 3 : 1, nt);
            if (!same(one, many)) { printf(".dat %d: %d threads differ from one\n", pe, nt); ++bad; }
        }
    }
    printf(bad ? "parse_chunks_check: %d mismatches\n" : "parse_chunks_check: ok%.0d\n", bad);
    return bad ? 1 : 0;
}
