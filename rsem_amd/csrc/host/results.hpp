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
    GroupInfo gi, gt, ta;
    if (!gi.load(refName + ".grp")) die("Cannot open %s.grp! It may not exist.", refName.c_str());
    const int m = gi.m;
    const bool alleleS = is_allele_specific(refName);  // WriteResults.h:106-123
    if (alleleS && (!gt.load(refName + ".gt") || !ta.load(refName + ".ta"))) die("Cannot load %s.gt / %s.ta!", refName.c_str(), refName.c_str());
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
    // allele-specific aggregation (WriteResults.h:184-224)
    int m_trans = 0;
    std::vector<double> trans_lens, trans_eels, trans_counts, trans_tpm, trans_fpkm, ta_pct, gt_pct;
    if (alleleS) {
        m_trans = ta.m;
        ta_pct.assign(M + 1, 0.0);
        trans_lens.assign(m_trans, 0.0); trans_eels.assign(m_trans, 0.0);
        trans_counts.assign(m_trans, 0.0); trans_tpm.assign(m_trans, 0.0); trans_fpkm.assign(m_trans, 0.0);
        for (int i = 0; i < m_trans; i++) {
            int b = ta.starts[i], e = ta.starts[i + 1];
            for (int j = b; j < e; j++) { trans_counts[i] += counts[j]; trans_tpm[i] += tpm[j]; trans_fpkm[i] += fpkm[j]; }
            if (trans_tpm[i] < kEpsilon) {
                double frac = 1.0 / (e - b);
                for (int j = b; j < e; j++) { trans_lens[i] += tlens[j] * frac; trans_eels[i] += eel[j] * frac; }
            } else {
                for (int j = b; j < e; j++) {
                    ta_pct[j] = trans_tpm[i] > kEpsilon ? tpm[j] / trans_tpm[i] : 0.0;
                    trans_lens[i] += tlens[j] * ta_pct[j];
                    trans_eels[i] += eel[j] * ta_pct[j];
                }
            }
        }
        gt_pct.assign(m_trans, 0.0);
        for (int i = 0; i < m; i++)
            if (gene_tpm[i] >= kEpsilon)
                for (int j = gt.starts[i]; j < gt.starts[i + 1]; j++) gt_pct[j] = gene_tpm[i] > kEpsilon ? trans_tpm[j] / gene_tpm[i] : 0.0;
    }
    FILE* fo = nullptr;
    if (alleleS) {
        fo = fopen((imdName + ".allele_res").c_str(), "w");  // WriteResults.h:262-290
        if (!fo) die("Cannot open %s.allele_res for writing!", imdName.c_str());
        for (int i = 1; i <= M; i++) fprintf(fo, "%s%c", T.t[i].seqname.c_str(), (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fo, "%s%c", T.t[i].transcript_id.c_str(), (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fo, "%s%c", T.t[i].gene_id.c_str(), (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fo, "%d%c", tlens[i], (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", eel[i], (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", counts[i], (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", tpm[i], (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", fpkm[i], (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", ta_pct[i] * 1e2, (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", isopct[i] * 1e2, (i < M ? '\t' : '\n'));
        fclose(fo);
        fo = fopen((imdName + ".iso_res").c_str(), "w");  // WriteResults.h:292-315
        if (!fo) die("Cannot open %s.iso_res for writing!", imdName.c_str());
        for (int i = 0; i < m_trans; i++) fprintf(fo, "%s%c", T.t[ta.starts[i]].transcript_id.c_str(), (i < m_trans - 1 ? '\t' : '\n'));
        for (int i = 0; i < m_trans; i++) fprintf(fo, "%s%c", T.t[ta.starts[i]].gene_id.c_str(), (i < m_trans - 1 ? '\t' : '\n'));
        for (int i = 0; i < m_trans; i++) fprintf(fo, "%.2f%c", trans_lens[i], (i < m_trans - 1 ? '\t' : '\n'));
        for (int i = 0; i < m_trans; i++) fprintf(fo, "%.2f%c", trans_eels[i], (i < m_trans - 1 ? '\t' : '\n'));
        for (int i = 0; i < m_trans; i++) fprintf(fo, "%.2f%c", trans_counts[i], (i < m_trans - 1 ? '\t' : '\n'));
        for (int i = 0; i < m_trans; i++) fprintf(fo, "%.2f%c", trans_tpm[i], (i < m_trans - 1 ? '\t' : '\n'));
        for (int i = 0; i < m_trans; i++) fprintf(fo, "%.2f%c", trans_fpkm[i], (i < m_trans - 1 ? '\t' : '\n'));
        for (int i = 0; i < m_trans; i++) fprintf(fo, "%.2f%c", gt_pct[i] * 1e2, (i < m_trans - 1 ? '\t' : '\n'));
        fclose(fo);
    }
    if (!alleleS) {
    fo = fopen((imdName + ".iso_res").c_str(), "w");
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
    }

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
                                const std::vector<double>& pve_c, const std::vector<double>& pve_c_genes,
                                bool alleleS = false, const GroupInfo* gt = nullptr, const GroupInfo* ta = nullptr,
                                const std::vector<double>* pve_c_trans = nullptr) {
    const int m = gi.m;
    std::vector<double> isopct(M + 1, 0.0), gene_counts(m, 0.0), gene_tpm(m, 0.0), gene_fpkm(m, 0.0);
    for (int i = 0; i < m; i++) {
        int b = gi.starts[i], e = gi.starts[i + 1];
        for (int j = b; j < e; j++) { gene_counts[i] += pme_c[j]; gene_tpm[i] += pme_tpm[j]; gene_fpkm[i] += pme_fpkm[j]; }
        if (gene_tpm[i] < kEpsilon) continue;
        for (int j = b; j < e; j++) isopct[j] = pme_tpm[j] / gene_tpm[i];
    }
    if (alleleS) {  // WriteResults.h:390-404, 430-468
        const int m_trans = ta->m;
        std::vector<double> ta_pct(M + 1, 0.0), trans_counts(m_trans, 0.0), trans_tpm(m_trans, 0.0), trans_fpkm(m_trans, 0.0), gt_pct(m_trans, 0.0);
        for (int i = 0; i < m_trans; i++) {
            int b = ta->starts[i], e = ta->starts[i + 1];
            for (int j = b; j < e; j++) { trans_counts[i] += pme_c[j]; trans_tpm[i] += pme_tpm[j]; trans_fpkm[i] += pme_fpkm[j]; }
            if (trans_tpm[i] < kEpsilon) continue;
            for (int j = b; j < e; j++) ta_pct[j] = pme_tpm[j] / trans_tpm[i];
        }
        for (int i = 0; i < m; i++)
            if (gene_tpm[i] >= kEpsilon)
                for (int j = gt->starts[i]; j < gt->starts[i + 1]; j++) gt_pct[j] = trans_tpm[j] / gene_tpm[i];
        FILE* fa = fopen((imdName + ".allele_res").c_str(), "a");
        if (!fa) die("Cannot open %s.allele_res!", imdName.c_str());
        for (int i = 1; i <= M; i++) fprintf(fa, "%.2f%c", pme_c[i], (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fa, "%.2f%c", sqrt(pve_c[i]), (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fa, "%.2f%c", pme_tpm[i], (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fa, "%.2f%c", pme_fpkm[i], (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fa, "%.2f%c", ta_pct[i] * 1e2, (i < M ? '\t' : '\n'));
        for (int i = 1; i <= M; i++) fprintf(fa, "%.2f%c", isopct[i] * 1e2, (i < M ? '\t' : '\n'));
        fclose(fa);
        fa = fopen((imdName + ".iso_res").c_str(), "a");
        if (!fa) die("Cannot open %s.iso_res!", imdName.c_str());
        for (int i = 0; i < m_trans; i++) fprintf(fa, "%.2f%c", trans_counts[i], (i < m_trans - 1 ? '\t' : '\n'));
        for (int i = 0; i < m_trans; i++) fprintf(fa, "%.2f%c", sqrt((*pve_c_trans)[i]), (i < m_trans - 1 ? '\t' : '\n'));
        for (int i = 0; i < m_trans; i++) fprintf(fa, "%.2f%c", trans_tpm[i], (i < m_trans - 1 ? '\t' : '\n'));
        for (int i = 0; i < m_trans; i++) fprintf(fa, "%.2f%c", trans_fpkm[i], (i < m_trans - 1 ? '\t' : '\n'));
        for (int i = 0; i < m_trans; i++) fprintf(fa, "%.2f%c", gt_pct[i] * 1e2, (i < m_trans - 1 ? '\t' : '\n'));
        fclose(fa);
    }
    FILE* fo = nullptr;
    if (!alleleS) {
    fo = fopen((imdName + ".iso_res").c_str(), "a");
    if (!fo) die("Cannot open %s.iso_res!", imdName.c_str());
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", pme_c[i], (i < M ? '\t' : '\n'));
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", sqrt(pve_c[i]), (i < M ? '\t' : '\n'));
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", pme_tpm[i], (i < M ? '\t' : '\n'));
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", pme_fpkm[i], (i < M ? '\t' : '\n'));
    for (int i = 1; i <= M; i++) fprintf(fo, "%.2f%c", isopct[i] * 1e2, (i < M ? '\t' : '\n'));
    fclose(fo);
    }
    fo = fopen((imdName + ".gene_res").c_str(), "a");
    if (!fo) die("Cannot open %s.gene_res!", imdName.c_str());
    for (int i = 0; i < m; i++) fprintf(fo, "%.2f%c", gene_counts[i], (i < m - 1 ? '\t' : '\n'));
    for (int i = 0; i < m; i++) fprintf(fo, "%.2f%c", sqrt(pve_c_genes[i]), (i < m - 1 ? '\t' : '\n'));
    for (int i = 0; i < m; i++) fprintf(fo, "%.2f%c", gene_tpm[i], (i < m - 1 ? '\t' : '\n'));
    for (int i = 0; i < m; i++) fprintf(fo, "%.2f%c", gene_fpkm[i], (i < m - 1 ? '\t' : '\n'));
    fclose(fo);
}

}  // namespace rsemh
