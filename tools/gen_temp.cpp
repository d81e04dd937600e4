// gen_temp.cpp -- seeded generator of a complete rsem-run-em input directory at benchmark scale
// (".temp generator without SAM").  TEST / BENCH INFRASTRUCTURE, not product.
//
//   gen_temp <outdir> <n_reads> <M> <read_type 0|1|2|3> [seed] [read_len] [sam|nosam] [kmin-kmax]
//
// read_type 0 / 2 (SingleModel / PairedEndModel): the same reads without qualities, as FASTA (utils.h:129-149,
// SingleRead.h:37-52); the bases still carry errors drawn at the phred rate of the hidden quality chain.
//
// kmin-kmax = isoforms per gene (default 2-9, ~6 alignments per read; 6-20 gives the ~12 per read of BASELINE configs[2])
//
// with a 7th argument "sam" it also writes <outdir>/aln.sam (the same reads and alignments as SAM records, in the
// order rsem-parse-alignments would need to reproduce the files above byte for byte).
//
// writes  <outdir>/ref.{seq,ti,grp}, <outdir>/temp/s.{dat,mparams,omit}, s_alignable*.fq, s_un*.fq,
// <outdir>/stat/s.cnt   in the formats of SURVEY.md Appendix A, so that BOTH the reference binary
// (oracle/_ref/rsem-run-em, after rsem-build-read-index) and rsem_amd/bin/rsem-run-em can run on it.
//
// Model of the data: genes own 2..9 isoforms; every isoform is a contiguous sub-interval of its gene's
// sequence (heavy overlap), so a read drawn from one isoform aligns, at shifted offsets, to every
// sibling that contains it; true theta ~ lognormal(0,2) with 30% zeros; 5% noise reads; phred qualities
// from a small Markov chain, base errors at the phred rate.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>

static const char BASES[4] = {'A', 'C', 'G', 'T'};
static char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }

struct Tx { int gene; int a, b; };  // interval of the gene sequence

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: gen_temp outdir n_reads M read_type(1|3) [seed] [read_len]\n"); return 1; }
    const std::string out = argv[1];
    const long long N = atoll(argv[2]);
    const int M = atoi(argv[3]), read_type = atoi(argv[4]);
    const unsigned seed = argc > 5 ? (unsigned)atoll(argv[5]) : 20250925u;
    const int L = argc > 6 ? atoi(argv[6]) : 100;
    const bool pe = read_type >= 2, hasq = (read_type & 1) != 0;
    const std::string ext = hasq ? ".fq" : ".fa";
    const bool want_sam = argc > 7 && std::string(argv[7]) == "sam";
    int kmin = 2, kmax = 9;
    if (argc > 8 && sscanf(argv[8], "%d-%d", &kmin, &kmax) != 2) { fprintf(stderr, "isoforms per gene: kmin-kmax\n"); return 1; }
    if (kmin < 1 || kmax < kmin) { fprintf(stderr, "isoforms per gene: 1 <= kmin <= kmax\n"); return 1; }
    if (read_type < 0 || read_type > 3) { fprintf(stderr, "read_type must be 0..3\n"); return 1; }
    std::mt19937_64 rng(seed);
    auto uni = [&](double a, double b) { return std::uniform_real_distribution<double>(a, b)(rng); };
    auto irand = [&](int a, int b) { return std::uniform_int_distribution<int>(a, b)(rng); };

    // genes / transcripts
    std::vector<std::string> gseq;
    std::vector<Tx> tx(1);
    std::vector<int> gstart;  // first transcript id of each gene
    while ((int)tx.size() - 1 < M) {
        int k = std::min(irand(kmin, kmax), M - ((int)tx.size() - 1));
        int Lg = irand(1500, 4000);
        std::string s(Lg, 'A');
        for (int i = 0; i < Lg; i++) s[i] = BASES[rng() & 3];
        gstart.push_back((int)tx.size());
        for (int j = 0; j < k; j++) tx.push_back(Tx{(int)gseq.size(), irand(0, 300), Lg - irand(0, 300)});
        gseq.push_back(std::move(s));
    }
    gstart.push_back(M + 1);
    const int m = (int)gseq.size();
    auto tlen = [&](int t) { return tx[t].b - tx[t].a; };

    std::string cmd = "mkdir -p " + out + "/temp " + out + "/stat";
    if (system(cmd.c_str()) != 0) return 1;
    {   // ref.seq (RefSeq.h:108-138), ref.ti (Transcript.h:119-148), ref.grp (GroupInfo.h:34-53)
        FILE* fs = fopen((out + "/ref.seq").c_str(), "w");
        FILE* ft = fopen((out + "/ref.ti").c_str(), "w");
        FILE* fg = fopen((out + "/ref.grp").c_str(), "w");
        fprintf(ft, "%d 1\n", M);
        for (int t = 1; t <= M; t++) {
            const int len = tlen(t);
            fprintf(fs, "%d %d\nt%d\n", len, len, t);
            fwrite(gseq[tx[t].gene].data() + tx[t].a, 1, len, fs);
            fputc('\n', fs);
            const int words = (len - 1) / 32 + 1;
            for (int w = 0; w < words; w++) fprintf(fs, w + 1 < words ? "0 " : "0\n");
            fprintf(ft, "t%d\ng%d\nt%d\n+ %d\n1 1 %d\n\n", t, tx[t].gene, t, len, len);
        }
        for (int g = 0; g <= m; g++) fprintf(fg, "%d\n", gstart[g]);
        fclose(fs); fclose(ft); fclose(fg);
    }
    // expression
    std::vector<double> cdf(M + 1, 0.0);
    {
        std::normal_distribution<double> nd(0.0, 2.0);
        for (int t = 1; t <= M; t++) {
            double th = exp(nd(rng));
            if (uni(0, 1) < 0.3) th = 0.0;
            cdf[t] = cdf[t - 1] + th * tlen(t);
        }
    }
    // quality Markov chain: states 2..40, drift to high quality
    auto next_q = [&](int q) { int d = irand(-5, 3); int v = q + d; if (v > 40) v = 40 - irand(0, 3); if (v < 2) v = 2 + irand(0, 3); return v; };

    const long long N0 = N / 20;
    const long long N1 = N - N0;
    FILE* fdat = fopen((out + "/temp/s.dat").c_str(), "w");
    FILE* fq1 = fopen((out + (pe ? "/temp/s_alignable_1" + ext : "/temp/s_alignable" + ext)).c_str(), "w");
    FILE* fq2 = pe ? fopen((out + "/temp/s_alignable_2" + ext).c_str(), "w") : nullptr;
    setvbuf(fdat, nullptr, _IOFBF, 1 << 22); setvbuf(fq1, nullptr, _IOFBF, 1 << 22); if (fq2) setvbuf(fq2, nullptr, _IOFBF, 1 << 22);
    FILE* fsam = nullptr;
    if (want_sam) {
        fsam = fopen((out + "/aln.sam").c_str(), "w");
        setvbuf(fsam, nullptr, _IOFBF, 1 << 22);
        fprintf(fsam, "@HD\tVN:1.0\tSO:unsorted\n");
        for (int t = 1; t <= M; t++) fprintf(fsam, "@SQ\tSN:t%d\tLN:%d\n", t, tlen(t));
        fprintf(fsam, "@PG\tID:gen_temp\n");
    }
    fprintf(fdat, "%-99s\n", "");  // header is patched at the end (parseIt.cpp:195-199)
    // Reads are generated in tasks of kTask reads, each from its own generator seeded by (seed, task), on all host
    // threads; the tasks' text is written in task order, so the files do not depend on the number of threads.
    constexpr long long kTask = 16384;
    struct TaskOut { std::string dat, q1, q2, sam; long long hits = 0; };
    auto gen_task = [&](long long task, TaskOut& O) {
        std::mt19937_64 rg(((uint64_t)seed << 20) ^ (uint64_t)(task + 1) * 0x9e3779b97f4a7c15ull);
        auto tuni = [&](double a, double b) { return std::uniform_real_distribution<double>(a, b)(rg); };
        auto tirand = [&](int a, int b) { return std::uniform_int_distribution<int>(a, b)(rg); };
        auto tnext_q = [&](int q) { int d = tirand(-5, 3); int v = q + d; if (v > 40) v = 40 - tirand(0, 3); if (v < 2) v = 2 + tirand(0, 3); return v; };
        std::string seq(L, 'A'), qual(L, 'I'), seq2(L, 'A'), qual2(L, 'I');
        char tmp[160];
        auto revcomp = [&](const std::string& x) { std::string r(x.rbegin(), x.rend()); for (char& c : r) c = comp(c); return r; };
        auto rev = [&](const std::string& x) { return std::string(x.rbegin(), x.rend()); };
        auto sam_rec = [&](long long id, int flag, int sv, int fwd, int mate_fwd, int tl, const std::string& sq, const std::string& ql, bool is_rev) {
            if (sv > 0) snprintf(tmp, sizeof(tmp), "r%lld\t%d\tt%d\t%d\t255\t%dM\t%s\t%d\t%d\t", id, flag, sv, fwd + 1, L, pe ? "=" : "*", pe ? mate_fwd + 1 : 0, tl);
            else snprintf(tmp, sizeof(tmp), "r%lld\t%d\t*\t0\t0\t*\t*\t0\t0\t", id, flag);
            O.sam += tmp;
            O.sam += is_rev ? revcomp(sq) : sq; O.sam += '\t';
            O.sam += is_rev ? rev(ql) : ql; O.sam += '\n';
        };
        auto emit_read = [&](std::string& f, long long id, const std::string& sq, const std::string& q) {
            snprintf(tmp, sizeof(tmp), "%cr%lld\n", hasq ? '@' : '>', id);
            f += tmp; f += sq;
            if (hasq) { f += "\n+\n"; f += q; }
            f += '\n';
        };
        auto make_read = [&](int t, int dir, int spos, std::string& sq, std::string& q) {
            const std::string& g = gseq[tx[t].gene];
            const int tl = tlen(t);
            int qv = tirand(25, 40);
            for (int i = 0; i < L; i++) {
                char c = dir == 0 ? g[tx[t].a + spos + i] : comp(g[tx[t].a + (tl - 1 - (spos + i))]);
                if (tuni(0, 1) < pow(10.0, -qv / 10.0)) c = BASES[rg() & 3];
                sq[i] = c;
                q[i] = (char)(qv + 33);
                qv = tnext_q(qv);
            }
        };
        const long long r0 = task * kTask, r1 = std::min(N1, r0 + kTask);
        O.dat.reserve((size_t)(r1 - r0) * 80);
        O.q1.reserve((size_t)(r1 - r0) * (2 * L + 20));
        if (pe) O.q2.reserve((size_t)(r1 - r0) * (2 * L + 20));
        std::string line;
        for (long long r = r0; r < r1; r++) {
            double u = tuni(0, cdf[M]);
            int t = (int)(std::upper_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
            t = std::min(std::max(t, 1), M);
            const int tl = tlen(t);
            const int dir = (rg() & 1);
            int frag = L;
            if (pe) { frag = (int)std::lround(std::normal_distribution<double>(200, 30)(rg)); frag = std::min(std::max(frag, L), std::min(tl, 400)); }
            const int fpos = tirand(0, tl - frag);             // forward coordinate of the fragment in t
            const int gpos = tx[t].a + fpos;                   // gene coordinate
            const int spos = dir == 0 ? fpos : tl - fpos - frag;  // position on the strand of alignment
            make_read(t, dir, spos, seq, qual);
            if (pe) make_read(t, !dir, tl - spos - frag, seq2, qual2);
            emit_read(O.q1, r, seq, qual);
            if (pe) emit_read(O.q2, r, seq2, qual2);
            // alignments: every isoform of the gene that contains [gpos, gpos+frag)
            line.clear();
            int k = 0;
            const int g = tx[t].gene;
            for (int sv = gstart[g]; sv < gstart[g + 1]; sv++) {
                if (tx[sv].a <= gpos && gpos + frag <= tx[sv].b) {
                    const int sl = tlen(sv), f2 = gpos - tx[sv].a;
                    const int p = dir == 0 ? f2 : sl - f2 - frag;
                    if (pe) snprintf(tmp, sizeof(tmp), " %d %d %d", dir == 0 ? sv : -sv, p, frag);
                    else snprintf(tmp, sizeof(tmp), " %d %d", dir == 0 ? sv : -sv, p);
                    line += tmp;
                    ++k;
                    if (want_sam) {
                        if (!pe) sam_rec(r, dir == 0 ? 0 : 16, sv, f2, 0, 0, seq, qual, dir != 0);
                        else if (dir == 0) {
                            sam_rec(r, 99, sv, f2, f2 + frag - L, frag, seq, qual, false);
                            sam_rec(r, 147, sv, f2 + frag - L, f2, -frag, seq2, qual2, true);
                        } else {
                            sam_rec(r, 83, sv, f2 + frag - L, f2, -frag, seq, qual, true);
                            sam_rec(r, 163, sv, f2, f2 + frag - L, frag, seq2, qual2, false);
                        }
                    }
                }
            }
            snprintf(tmp, sizeof(tmp), "%d", k);
            O.dat += tmp; O.dat += line; O.dat += '\n';
            O.hits += k;
        }
    };
    long long nHits = 0;
    {
        const long long ntasks = (N1 + kTask - 1) / kTask;
        const int nthr = (int)std::max(1u, std::min(std::thread::hardware_concurrency(), 256u));
        for (long long w0 = 0; w0 < ntasks; w0 += nthr) {
            const int nw = (int)std::min<long long>(nthr, ntasks - w0);
            std::vector<TaskOut> outs(nw);
            std::vector<std::thread> th;
            for (int i = 0; i < nw; i++) th.emplace_back([&, i]() { gen_task(w0 + i, outs[i]); });
            for (auto& x : th) x.join();
            for (int i = 0; i < nw; i++) {
                fwrite(outs[i].dat.data(), 1, outs[i].dat.size(), fdat);
                fwrite(outs[i].q1.data(), 1, outs[i].q1.size(), fq1);
                if (fq2) fwrite(outs[i].q2.data(), 1, outs[i].q2.size(), fq2);
                if (fsam) fwrite(outs[i].sam.data(), 1, outs[i].sam.size(), fsam);
                nHits += outs[i].hits;
            }
        }
    }
    fseek(fdat, 0, SEEK_SET);
    fprintf(fdat, "%lld %lld %d", N1, nHits, read_type);
    fclose(fdat); fclose(fq1); if (fq2) fclose(fq2);
    std::string seq(L, 'A'), qual(L, 'I');
    auto revcomp = [&](const std::string& x) { std::string r(x.rbegin(), x.rend()); for (char& c : r) c = comp(c); return r; };
    auto rev = [&](const std::string& x) { return std::string(x.rbegin(), x.rend()); };
    auto sam_rec = [&](long long id, int flag, int sv, int fwd, int mate_fwd, int tl, const std::string& sq, const std::string& ql, bool is_rev) {
        (void)sv; (void)fwd; (void)mate_fwd; (void)tl;
        fprintf(fsam, "r%lld\t%d\t*\t0\t0\t*\t*\t0\t0\t", id, flag);
        const std::string a2 = is_rev ? revcomp(sq) : sq, b2 = is_rev ? rev(ql) : ql;
        fwrite(a2.data(), 1, a2.size(), fsam); fputc('\t', fsam);
        fwrite(b2.data(), 1, b2.size(), fsam); fputc('\n', fsam);
    };
    auto emit_read = [&](FILE* f, long long id, const std::string& sq, const std::string& q) {
        fprintf(f, "%cr%lld\n", hasq ? '@' : '>', id);
        fwrite(sq.data(), 1, sq.size(), f);
        if (hasq) { fputs("\n+\n", f); fwrite(q.data(), 1, q.size(), f); }
        fputc('\n', f);
    };
    {   // unalignable reads
        FILE* fu1 = fopen((out + (pe ? "/temp/s_un_1" + ext : "/temp/s_un" + ext)).c_str(), "w");
        FILE* fu2 = pe ? fopen((out + "/temp/s_un_2" + ext).c_str(), "w") : nullptr;
        for (long long r = 0; r < N0; r++) {
            for (int mth = 0; mth < (pe ? 2 : 1); mth++) {
                int qv = irand(10, 40);
                for (int i = 0; i < L; i++) { seq[i] = BASES[rng() & 3]; qual[i] = (char)(qv + 33); qv = next_q(qv); }
                emit_read(mth ? fu2 : fu1, N1 + r, seq, qual);
                if (fsam) sam_rec(N1 + r, pe ? (mth ? 141 : 77) : 4, 0, 0, 0, 0, seq, qual, false);
            }
        }
        fclose(fu1); if (fu2) fclose(fu2);
        if (fsam) fclose(fsam);
    }
    FILE* f = fopen((out + "/stat/s.cnt").c_str(), "w");
    fprintf(f, "%lld %lld 0 %lld\n%lld 0 %lld\n%lld %d\n0\t%lld\nInf\t0\n", N0, N1, N, N1, N1, nHits, read_type, N0);
    fclose(f);
    f = fopen((out + "/temp/s.mparams").c_str(), "w");
    fprintf(f, "1 1000\n0.5\n0\n20\n1 1000\n-1 0\n25\n");
    fclose(f);
    f = fopen((out + "/temp/s.omit").c_str(), "w");
    fclose(f);
    printf("gen_temp: N0=%lld N1=%lld nHits=%lld (%.2f per read) M=%d genes=%d\n", N0, N1, nHits, (double)nHits / N1, M, m);
    return 0;
}
