// results.hpp -- writers of the result files (WriteResults.h:125-479, EM.cpp:484-500, Gibbs.cpp:257-262).
#pragma once
#include "files.hpp"

namespace rsemh {

inline bool is_allele_specific(const std::string& refName) {  // WriteResults.h:106-123
    return file_exists(refName + ".gt") && file_exists(refName + ".ta");
}

// imd.iso_res / imd.gene_res, row-major (WriteResults.h:125-355, non-allele-specific branch)
inline void write_results_em(int M, const std::string& refName, const std::string& imdName, const Transcripts& T,
                             const std::vector<double>& theta, const std::vector<double>& eel, const double* counts,
                             bool appendNames) {
    if (is_allele_specific(refName)) die("Allele-specific references (%s.ta/.gt) are not supported by this build yet.", refName.c_str());
    GroupInfo gi;
    if (!gi.load(refName + ".grp")) die("Cannot open %s.grp! It may not exist.", refName.c_str());
    const int m = gi.m;
    std::vector<double> tpm, fpkm;
    calc_expression(M, theta, eel, tpm, fpkm);
    std::vector<double> isopct(M + 1, 0.0), glens(m, 0.0), gene_eels(m, 0.0), gene_counts(m, 0.0), gene_tpm(m, 0.0), gene_fpkm(m, 0.0);
    std::vector<int> tlens(M + 1, 0);
    for (int i = 0; i < m; i++) {
        int b = gi.starts[i], e = gi.starts[i + 1];
        for (int j = b; j < e; j++) {
            tlens[j] = T.t[j].length;
            gene_counts[i] += counts[j];
            gene_tpm[i] += tpm[j];
            gene_fpkm[i] += fpkm[j];
        }
        if (gene_tpm[i] < kEpsilon) {
            double frac = 1.0 / (e - b);
            for (int j = b; j < e; j++) { glens[i] += tlens[j] * frac; gene_eels[i] += eel[j] * frac; }
        } else {
            for (int j = b; j < e; j++) {
                isopct[j] = gene_tpm[i] > kEpsilon ? tpm[j] / gene_tpm[i] : 0.0;
                glens[i] += tlens[j] * isopct[j];
                gene_eels[i] += eel[j] * isopct[j];
            }
        }
    }
    FILE* fo = fopen((imdName + ".iso_res").c_str(), "w");
    if (!fo) die("Cannot open %s.iso_res for writing!", imdName.c_str());
    for (int i = 1; i <= M; i++) {
        fprintf(fo, "%s", T.t[i].transcript_id.c_str());
        if (appendNames && !T.t[i].transcript_name.empty()) fprintf(fo, "_%s", T.t[i].transcript_name.c_str());
        fprintf(fo, "%c", (i < M ? '\t' : '\n'));
    }
    for (int i = 1; i <= M; i++) {
        fprintf(fo, "%s", T.t[i].gene_id.c_str());
        if (appendNames && !T.t[i].gene_name.empty()) fprintf(fo, "_%s", T.t[i].gene_name.c_str());
        fprintf(fo, "%c", (i < M ? '\t' : '\n'));
    }
    for (int i = 1; i <= M; i++) fprintf(fo, "%d%c", tlens[i], (i < M ? '\t' : '\n'));
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", eel[i], (i < M ? '\t' : '\n'));
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", counts[i], (i < M ? '\t' : '\n'));
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", tpm[i], (i < M ? '\t' : '\n'));
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", fpkm[i], (i < M ? '\t' : '\n'));
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", isopct[i] * 1e2, (i < M ? '\t' : '\n'));
    fclose(fo);

    fo = fopen((imdName + ".gene_res").c_str(), "w");
    if (!fo) die("Cannot open %s.gene_res for writing!", imdName.c_str());
    for (int i = 0; i < m; i++) {
        const TranscriptInfo& t = T.t[gi.starts[i]];
        fprintf(fo, "%s", t.gene_id.c_str());
        if (appendNames && !t.gene_name.empty()) fprintf(fo, "_%s", t.gene_name.c_str());
        fprintf(fo, "%c", (i < m - 1 ? '\t' : '\n'));
    }
    for (int i = 0; i < m; i++) {
        int b = gi.starts[i], e = gi.starts[i + 1];
        std::string curtid;
        for (int j = b; j < e; j++) {
            const std::string& tid = T.t[j].transcript_id;
            if (curtid != tid) {
                if (!curtid.empty()) fprintf(fo, ",");
                fprintf(fo, "%s", tid.c_str());
                if (appendNames && !T.t[j].transcript_name.empty()) fprintf(fo, "_%s", T.t[j].transcript_name.c_str());
                curtid = tid;
            }
        }
        fprintf(fo, "%c", (i < m - 1 ? '\t' : '\n'));
    }
    for (int i = 0; i < m; i++) fprintf(fo, "%.2f%c", glens[i], (i < m - 1 ? '\t' : '\n'));
    for (int i = 0; i < m; i++) fprintf(fo, "%.2f%c", gene_eels[i], (i < m - 1 ? '\t' : '\n'));
    for (int i = 0; i < m; i++) fprintf(fo, "%.2f%c", gene_counts[i], (i < m - 1 ? '\t' : '\n'));
    for (int i = 0; i < m; i++) fprintf(fo, "%.2f%c", gene_tpm[i], (i < m - 1 ? '\t' : '\n'));
    for (int i = 0; i < m; i++) fprintf(fo, "%.2f%c", gene_fpkm[i], (i < m - 1 ? '\t' : '\n'));
    fclose(fo);
}

// rows appended by rsem-run-gibbs (WriteResults.h:357-479, non-allele-specific branch)
inline void write_results_gibbs(int M, const GroupInfo& gi, const std::string& imdName, const std::vector<double>& pme_c,
                                const std::vector<double>& pme_fpkm, const std::vector<double>& pme_tpm,
                                const std::vector<double>& pve_c, const std::vector<double>& pve_c_genes) {
    const int m = gi.m;
    std::vector<double> isopct(M + 1, 0.0), gene_counts(m, 0.0), gene_tpm(m, 0.0), gene_fpkm(m, 0.0);
    for (int i = 0; i < m; i++) {
        int b = gi.starts[i], e = gi.starts[i + 1];
        for (int j = b; j < e; j++) { gene_counts[i] += pme_c[j]; gene_tpm[i] += pme_tpm[j]; gene_fpkm[i] += pme_fpkm[j]; }
        if (gene_tpm[i] < kEpsilon) continue;
        for (int j = b; j < e; j++) isopct[j] = pme_tpm[j] / gene_tpm[i];
    }
    FILE* fo = fopen((imdName + ".iso_res").c_str(), "a");
    if (!fo) die("Cannot open %s.iso_res!", imdName.c_str());
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", pme_c[i], (i < M ? '\t' : '\n'));
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", sqrt(pve_c[i]), (i < M ? '\t' : '\n'));
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", pme_tpm[i], (i < M ? '\t' : '\n'));
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", pme_fpkm[i], (i < M ? '\t' : '\n'));
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", isopct[i] * 1e2, (i < M ? '\t' : '\n'));
    fclose(fo);
    fo = fopen((imdName + ".gene_res").c_str(), "a");
    if (!fo) die("Cannot open %s.gene_res!", imdName.c_str());
    for (int i = 0; i < m; i++) fprintf(fo, "%.2f%c", gene_counts[i], (i < m - 1 ? '\t' : '\n'));
    for (int i = 0; i < m; i++) fprintf(fo, "%.2f%c", sqrt(pve_c_genes[i]), (i < m - 1 ? '\t' : '\n'));
    for (int i = 0; i < m; i++) fprintf(fo, "%.2f%c", gene_tpm[i], (i < m - 1 ? '\t' : '\n'));
    for (int i = 0; i < m; i++) fprintf(fo, "%.2f%c", gene_fpkm[i], (i < m - 1 ? '\t' : '\n'));
    fclose(fo);
}

}  // namespace rsemh
