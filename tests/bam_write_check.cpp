// bam_write_check.cpp -- TEST INFRASTRUCTURE for rsem_amd/csrc/host/bam_io.hpp (the -b pass of rsem-run-em) without a GPU: takes
// the alignment weights from the ZW:f tags of a transcript.bam the REFERENCE wrote (tests/golden/*/golden.transcript.bam), runs
// write_transcript_bam on the fixture's SAM (or BAM) input with them, and prints the decompressed output stream's size and a
// checksum; the Python side compares thread counts / chunk sizes with each other and the records with the golden file's.
//   bam_write_check <ref.ti> <input .sam|.bam> <golden.transcript.bam> <out.bam> <paired 0|1> <threads>
#include <chrono>

#include "../rsem_amd/csrc/host/bam_io.hpp"

using namespace rsemh;

static void read_all_records(const std::string& path, AlnHeader& H, std::vector<AlnRecord>& recs) {
    BgzfReader z;
    if (!z.open(path)) die("cannot open %s", path.c_str());
    char magic[4];
    int32_t l_text, n_ref;
    z.read(magic, 4);
    z.read(&l_text, 4);
    H.text.resize(l_text);
    z.read(&H.text[0], l_text);
    z.read(&n_ref, 4);
    for (int i = 0; i < n_ref; i++) {
        int32_t l_name, l_ref;
        z.read(&l_name, 4);
        std::string nm(l_name, '\0');
        z.read(&nm[0], l_name);
        nm.resize(strlen(nm.c_str()));
        z.read(&l_ref, 4);
        H.names.push_back(nm);
        H.lens.push_back(l_ref);
    }
    int32_t bs;
    while (z.read(&bs, 4)) {
        AlnRecord r;
        r.d.resize(bs);
        if (!z.read(r.d.data(), bs)) die("truncated %s", path.c_str());
        recs.push_back(std::move(r));
    }
}

static bool zw_of(const AlnRecord& r, float& v) {
    const std::vector<uint8_t>& d = r.d;
    uint16_t n_cig; int32_t l_seq;
    memcpy(&n_cig, d.data() + 12, 2);
    memcpy(&l_seq, d.data() + 16, 4);
    size_t p = 32 + d[8] + 4 * (size_t)n_cig + (size_t)(l_seq + 1) / 2 + (size_t)l_seq;
    while (p + 3 <= d.size()) {
        const bool zw = d[p] == 'Z' && d[p + 1] == 'W';
        const char type = (char)d[p + 2];
        if (zw) { memcpy(&v, d.data() + p + 3, 4); return true; }
        p += 3;
        switch (type) {
            case 'A': case 'c': case 'C': p += 1; break;
            case 's': case 'S': p += 2; break;
            case 'i': case 'I': case 'f': p += 4; break;
            case 'Z': case 'H': while (p < d.size() && d[p]) ++p; ++p; break;
            case 'B': { const char sub = (char)d[p]; int32_t n; memcpy(&n, d.data() + p + 1, 4); p += 5 + (size_t)n * ((sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4); break; }
            default: die("unknown aux type");
        }
    }
    return false;
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: bam_write_check ref.ti input golden.bam out.bam paired threads\n"); return 2; }
    Transcripts T = load_transcripts(argv[1]);
    const bool paired = atoi(argv[5]) != 0;
    const int threads = atoi(argv[6]);
    AlnHeader gh;
    std::vector<AlnRecord> gold;
    read_all_records(argv[3], gh, gold);
    std::vector<std::pair<std::string, int>> dict;
    for (int i = 1; i <= T.M; i++) dict.push_back({T.type == 2 ? T.t[i].seqname : T.t[i].transcript_id, i});
    std::sort(dict.begin(), dict.end());
    auto sid_of = [&](int refID) { auto it = std::lower_bound(dict.begin(), dict.end(), std::make_pair(gh.names[refID], -1)); return it->second; };
    std::vector<int32_t> sids;
    std::vector<double> w;
    for (size_t i = 0; i < gold.size(); i += paired ? 2 : 1) {
        const AlnRecord& a = gold[i];
        const bool m = paired ? (a.mapped() && gold[i + 1].mapped()) : a.mapped();
        if (!m) continue;
        float v = 0;
        if (!zw_of(a, v)) die("golden record without ZW");
        sids.push_back(sid_of(a.refID()));
        w.push_back((double)v);
    }
    if (const char* rep = getenv("BAM_CHECK_REPEAT")) {  // throughput mode: the input holds the fixture's alignment lines `rep` times over
        const int n = atoi(rep);
        const size_t k = w.size();
        for (int i = 1; i < n; i++) { sids.insert(sids.end(), sids.begin(), sids.begin() + k); w.insert(w.end(), w.begin(), w.begin() + k); }
        const auto t0 = std::chrono::steady_clock::now();
        write_transcript_bam(argv[2], argv[4], paired, sids.data(), w.data(), w.size(), T, threads);
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %d: %zu weighted alignments, %.2f s, %.2f M records/s\n", threads, w.size(), el, (double)gold.size() * n / el * 1e-6);
        return 0;
    }
    write_transcript_bam(argv[2], argv[4], paired, sids.data(), w.data(), w.size(), T, threads);
    AlnHeader oh;
    std::vector<AlnRecord> out;
    read_all_records(argv[4], oh, out);
    // FNV-1a over the decompressed record stream
    unsigned long long hsh = 1469598103934665603ull, bytes = 0;
    for (const AlnRecord& r : out) for (uint8_t c : r.d) { hsh = (hsh ^ c) * 1099511628211ull; ++bytes; }
    size_t same = 0, close_ = 0;
    if (out.size() != gold.size()) { printf("RECORDS %zu GOLD %zu\n", out.size(), gold.size()); return 1; }
    for (size_t i = 0; i < out.size(); i++) {
        if (out[i].d == gold[i].d) { ++same; continue; }
        // only MAPQ may differ by rounding of the weight (the ZW float is the golden's own)
        std::vector<uint8_t> a = out[i].d, b = gold[i].d;
        if (a.size() == b.size() && abs((int)a[9] - (int)b[9]) <= 1) { a[9] = b[9]; if (a == b) { ++close_; continue; } }
        // a weight within float precision of 1 loses its distance from 1 in the ZW float this harness takes it from: MAPQ = -10 log10(1 - w)
        // cannot be recomputed from it (the GPU test compares with the weights' doubles); everything else must still be equal
        float zv = 0;
        if (a.size() == b.size() && zw_of(out[i], zv) && zv > 0.99999f) { a[9] = b[9]; if (a == b) { ++close_; continue; } }
        printf("RECORD %zu DIFFERS (mapq %d vs %d)\n", i, (int)out[i].d[9], (int)gold[i].d[9]);
        return 1;
    }
    printf("records %zu identical %zu mapq_off_by_one %zu header_equal %d stream_bytes %llu fnv %016llx\n", out.size(), same, close_, (int)(oh.text == gh.text && oh.names == gh.names), bytes, hsh);
    return 0;
}
