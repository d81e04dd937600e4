// temp_to_rsb.cpp -- TEST / BENCH INFRASTRUCTURE: rewrite the reference's text hand-off files (imdName.dat, the
// FASTA/FASTQ read categories) as the binary directory imdName.rsb/ that `rsem-parse-alignments --binary` writes
// (rsem_amd/csrc/host/rsb.hpp).  Used to (a) check that the parser's binary output equals the conversion of its own
// text output and (b) give the end-to-end scripts binary inputs at scales where no SAM file is generated.
//   temp_to_rsb imdName statName read_type
#include "../rsem_amd/csrc/host/rsb.hpp"

using namespace rsemh;

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: temp_to_rsb imdName statName read_type\n"); return 1; }
    const std::string imdName = argv[1], statName = argv[2];
    const int read_type = atoi(argv[3]);
    const bool pe = read_type >= 2, q = (read_type == 1 || read_type == 3);
    uint64_t N0, N1, N2, Ntot;
    load_cnt(statName + ".cnt", N0, N1, N2, Ntot);
    const uint64_t Ncat[3] = {N0, N1, N2};
    RsbWriter w(imdName, read_type);
    if (N1 > 0) {
        DatData D = load_dat(imdName + ".dat", read_type);
        std::vector<uint32_t> lens(D.N1);
        for (uint64_t i = 0; i < D.N1; i++) lens[i] = (uint32_t)(D.row_ptr[i + 1] - D.row_ptr[i]);
        w.append_hits(lens.data(), D.N1, D.sid_signed.data(), D.pos.data(), pe ? D.insertL.data() : nullptr);
    }
    for (int c = 0; c < 3; c++) {
        if (Ncat[c] == 0) continue;
        std::vector<std::string> names = read_file_names(imdName, c, read_type);
        for (size_t m = 0; m < names.size(); m++) {
            ReadFile R = parse_read_file(names[m], q, false, 0);
            std::vector<uint32_t> lens(R.n);
            for (uint64_t i = 0; i < R.n; i++) lens[i] = (uint32_t)R.len(i);
            w.append_reads(c, (int)m, lens.data(), R.n, R.seq.data(), q ? R.qual.data() : nullptr);
        }
    }
    w.finish();
    printf("temp_to_rsb: N = %llu %llu %llu, nHits = %llu\n", (unsigned long long)w.header().N[0], (unsigned long long)w.header().N[1],
           (unsigned long long)w.header().N[2], (unsigned long long)w.header().nHits);
    return 0;
}
