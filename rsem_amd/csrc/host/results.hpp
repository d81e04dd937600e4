// results.hpp -- the result files of rsem-run-em / rsem-run-gibbs: imd.iso_res, imd.gene_res, imd.allele_res.
//
// Format (the pipeline's next stage, rsem-calculate-expression, parses these by ROW): every file is a matrix written
// row-major, one row per quantity, values tab-separated and printed with %.2f (WriteResults.h:125-479).  The numbers
// are one hierarchy seen at different levels -- alleles roll up into transcripts (ref.ta), transcripts into genes
// (ref.grp, or ref.gt over the transcripts of an allele-specific reference) -- so everything here is built from one
// roll-up routine and one row writer instead of a function per file.
#pragma once
#include "files.hpp"

namespace rsemh {

inline bool is_allele_specific(const std::string& refName) {  // WriteResults.h:106-123
    return file_exists(refName + ".gt") && file_exists(refName + ".ta");
}

// One level of the hierarchy: parent i owns children [starts[i], starts[i+1]).
//   counts / tpm / fpkm of a parent = sums over its children (in child order);
//   share[child] = child tpm / parent tpm (0 when the parent's tpm is below EPS);
//   a parent's length and effective length = share-weighted means of its children's, or plain means when the
//   parent is not expressed (WriteResults.h:150-181 for genes, 184-224 for transcripts over alleles).
struct Level {
    std::vector<double> counts, tpm, fpkm, len, eel;  // per parent
    std::vector<double> share;                        // per child (indexed like the child arrays)
};

template <typename LenArray>
inline Level roll_up(const std::vector<int>& starts, int n_parents, size_t n_children, const double* counts, const double* tpm,
                     const double* fpkm, const LenArray* child_len, const double* child_eel) {
    Level L;
    L.counts.assign(n_parents, 0.0); L.tpm.assign(n_parents, 0.0); L.fpkm.assign(n_parents, 0.0);
    L.share.assign(n_children, 0.0);
    const bool lengths = child_len != nullptr;
    if (lengths) { L.len.assign(n_parents, 0.0); L.eel.assign(n_parents, 0.0); }
    for (int p = 0; p < n_parents; p++) {
        const int first = starts[p], last = starts[p + 1];
        for (int c = first; c < last; c++) { L.counts[p] += counts[c]; L.tpm[p] += tpm[c]; L.fpkm[p] += fpkm[c]; }
        const bool expressed = !(L.tpm[p] < kEpsilon);
        for (int c = first; c < last; c++) {
            if (expressed) L.share[c] = L.tpm[p] > kEpsilon ? tpm[c] / L.tpm[p] : 0.0;
            if (lengths) {
                const double w = expressed ? L.share[c] : 1.0 / (last - first);
                L.len[p] += child_len[c] * w;
                L.eel[p] += child_eel[c] * w;
            }
        }
    }
    return L;
}

// rows of one file; cells are produced by a callable so that names, integers and %.2f values share the plumbing
class RowFile {
  public:
    RowFile(const std::string& path, const char* mode) : f_(fopen(path.c_str(), mode)), path_(path) {
        if (!f_) die("Cannot open %s%s", path.c_str(), mode[0] == 'w' ? " for writing!" : "!");
    }
    ~RowFile() { if (f_) fclose(f_); }
    template <typename Cell>
    void row(int first, int last, Cell cell) {  // cells first..last inclusive
        for (int i = first; i <= last; i++) {
            cell(f_, i);
            fputc(i < last ? '\t' : '\n', f_);
        }
    }
    void values(int first, int last, const double* v, double scale = 1.0) {
        write_cells_line(f_, first, last, '\t', [&](char* b, long i) { return snprintf(b, rsemh::kCellBuf, "%.2f", v[i] * scale); });
    }
    void roots(int first, int last, const double* v) {
        write_cells_line(f_, first, last, '\t', [&](char* b, long i) { return snprintf(b, rsemh::kCellBuf, "%.2f", sqrt(v[i])); });
    }

  private:
    FILE* f_;
    std::string path_;
};

// imd.iso_res / imd.gene_res [/ imd.allele_res] after the EM (WriteResults.h:125-355)
inline void write_results_em(int M, const std::string& refName, const std::string& imdName, const Transcripts& T,
                             const std::vector<double>& theta, const std::vector<double>& eel, const double* counts,
                             bool appendNames) {
    GroupInfo genes, gene_over_trans, trans_over_alleles;
    if (!genes.load(refName + ".grp")) die("Cannot open %s.grp! It may not exist.", refName.c_str());
    const bool alleleS = is_allele_specific(refName);
    if (alleleS && (!gene_over_trans.load(refName + ".gt") || !trans_over_alleles.load(refName + ".ta")))
        die("Cannot load %s.gt / %s.ta!", refName.c_str(), refName.c_str());
    std::vector<double> tpm, fpkm;
    calc_expression(M, theta, eel, tpm, fpkm);
    std::vector<int> tlens(M + 1, 0);
    for (int j = 1; j <= M; j++) tlens[j] = T.t[j].length;

    const int m = genes.m;
    const Level G = roll_up(genes.starts, m, (size_t)M + 1, counts, tpm.data(), fpkm.data(), tlens.data(), eel.data());
    auto with_name = [&](FILE* f, const std::string& id, const std::string& name) {
        fputs(id.c_str(), f);
        if (appendNames && !name.empty()) fprintf(f, "_%s", name.c_str());
    };

    if (alleleS) {
        const int mt = trans_over_alleles.m;
        const Level A = roll_up(trans_over_alleles.starts, mt, (size_t)M + 1, counts, tpm.data(), fpkm.data(), tlens.data(), eel.data());
        // share of a transcript in its gene (WriteResults.h:218-224): transcripts are the children of ref.gt
        std::vector<double> in_gene(mt, 0.0);
        for (int g = 0; g < m; g++)
            if (!(G.tpm[g] < kEpsilon))
                for (int t = gene_over_trans.starts[g]; t < gene_over_trans.starts[g + 1]; t++) in_gene[t] = G.tpm[g] > kEpsilon ? A.tpm[t] / G.tpm[g] : 0.0;
        {
            RowFile fa(imdName + ".allele_res", "w");  // WriteResults.h:262-290
            fa.row(1, M, [&](FILE* f, int i) { fputs(T.t[i].seqname.c_str(), f); });
            fa.row(1, M, [&](FILE* f, int i) { fputs(T.t[i].transcript_id.c_str(), f); });
            fa.row(1, M, [&](FILE* f, int i) { fputs(T.t[i].gene_id.c_str(), f); });
            fa.row(1, M, [&](FILE* f, int i) { fprintf(f, "%d", tlens[i]); });
            fa.values(1, M, eel.data());
            fa.values(1, M, counts);
            fa.values(1, M, tpm.data());
            fa.values(1, M, fpkm.data());
            fa.values(1, M, A.share.data(), 1e2);
            fa.values(1, M, G.share.data(), 1e2);
        }
        RowFile fi(imdName + ".iso_res", "w");  // WriteResults.h:292-315: one column per transcript = group of alleles
        fi.row(0, mt - 1, [&](FILE* f, int i) { fputs(T.t[trans_over_alleles.starts[i]].transcript_id.c_str(), f); });
        fi.row(0, mt - 1, [&](FILE* f, int i) { fputs(T.t[trans_over_alleles.starts[i]].gene_id.c_str(), f); });
        fi.values(0, mt - 1, A.len.data());
        fi.values(0, mt - 1, A.eel.data());
        fi.values(0, mt - 1, A.counts.data());
        fi.values(0, mt - 1, A.tpm.data());
        fi.values(0, mt - 1, A.fpkm.data());
        fi.values(0, mt - 1, in_gene.data(), 1e2);
    } else {
        RowFile fi(imdName + ".iso_res", "w");
        fi.row(1, M, [&](FILE* f, int i) { with_name(f, T.t[i].transcript_id, T.t[i].transcript_name); });
        fi.row(1, M, [&](FILE* f, int i) { with_name(f, T.t[i].gene_id, T.t[i].gene_name); });
        fi.row(1, M, [&](FILE* f, int i) { fprintf(f, "%d", tlens[i]); });
        fi.values(1, M, eel.data());
        fi.values(1, M, counts);
        fi.values(1, M, tpm.data());
        fi.values(1, M, fpkm.data());
        fi.values(1, M, G.share.data(), 1e2);
    }

    RowFile fg(imdName + ".gene_res", "w");
    fg.row(0, m - 1, [&](FILE* f, int g) {
        const TranscriptInfo& t = T.t[genes.starts[g]];
        with_name(f, t.gene_id, t.gene_name);
    });
    fg.row(0, m - 1, [&](FILE* f, int g) {  // the gene's transcript ids, each once (alleles repeat theirs)
        const std::string* prev = nullptr;
        for (int j = genes.starts[g]; j < genes.starts[g + 1]; j++) {
            const std::string& tid = T.t[j].transcript_id;
            if (prev && *prev == tid) continue;
            if (prev) fputc(',', f);
            with_name(f, tid, T.t[j].transcript_name);
            prev = &tid;
        }
    });
    fg.values(0, m - 1, G.len.data());
    fg.values(0, m - 1, G.eel.data());
    fg.values(0, m - 1, G.counts.data());
    fg.values(0, m - 1, G.tpm.data());
    fg.values(0, m - 1, G.fpkm.data());
}

// rows appended by rsem-run-gibbs: posterior mean count, its standard deviation, posterior mean TPM / FPKM and the
// shares recomputed from the posterior means (WriteResults.h:357-479)
inline void write_results_gibbs(int M, const GroupInfo& genes, const std::string& imdName, const std::vector<double>& pme_c,
                                const std::vector<double>& pme_fpkm, const std::vector<double>& pme_tpm,
                                const std::vector<double>& pve_c, const std::vector<double>& pve_c_genes,
                                bool alleleS = false, const GroupInfo* gene_over_trans = nullptr, const GroupInfo* trans_over_alleles = nullptr,
                                const std::vector<double>* pve_c_trans = nullptr) {
    const int m = genes.m;
    const Level G = roll_up<int>(genes.starts, m, (size_t)M + 1, pme_c.data(), pme_tpm.data(), pme_fpkm.data(), nullptr, nullptr);
    if (alleleS) {
        const int mt = trans_over_alleles->m;
        const Level A = roll_up<int>(trans_over_alleles->starts, mt, (size_t)M + 1, pme_c.data(), pme_tpm.data(), pme_fpkm.data(), nullptr, nullptr);
        std::vector<double> in_gene(mt, 0.0);
        for (int g = 0; g < m; g++)
            if (!(G.tpm[g] < kEpsilon))
                for (int t = gene_over_trans->starts[g]; t < gene_over_trans->starts[g + 1]; t++) in_gene[t] = A.tpm[t] / G.tpm[g];
        {
            RowFile fa(imdName + ".allele_res", "a");
            fa.values(1, M, pme_c.data());
            fa.roots(1, M, pve_c.data());
            fa.values(1, M, pme_tpm.data());
            fa.values(1, M, pme_fpkm.data());
            fa.values(1, M, A.share.data(), 1e2);
            fa.values(1, M, G.share.data(), 1e2);
        }
        RowFile fi(imdName + ".iso_res", "a");
        fi.values(0, mt - 1, A.counts.data());
        fi.roots(0, mt - 1, pve_c_trans->data());
        fi.values(0, mt - 1, A.tpm.data());
        fi.values(0, mt - 1, A.fpkm.data());
        fi.values(0, mt - 1, in_gene.data(), 1e2);
    } else {
        RowFile fi(imdName + ".iso_res", "a");
        fi.values(1, M, pme_c.data());
        fi.roots(1, M, pve_c.data());
        fi.values(1, M, pme_tpm.data());
        fi.values(1, M, pme_fpkm.data());
        fi.values(1, M, G.share.data(), 1e2);
    }
    RowFile fg(imdName + ".gene_res", "a");
    fg.values(0, m - 1, G.counts.data());
    fg.roots(0, m - 1, pve_c_genes.data());
    fg.values(0, m - 1, G.tpm.data());
    fg.values(0, m - 1, G.fpkm.data());
}

}  // namespace rsemh
