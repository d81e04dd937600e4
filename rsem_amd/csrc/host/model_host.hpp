// model_host.hpp -- host side of RSEM's four read models (SingleModel.h, SingleQModel.h,
// PairedEndModel.h, PairedEndQModel.h and their parts Orientation.h, LenDist.h, RSPD.h, Profile.h,
// QProfile.h, NoiseProfile.h, NoiseQProfile.h, QualDist.h): the small dense tables, their
// normalisation between EM rounds (init / collect / finish), the mask weights (calcMW) and the
// .model file.  The per-read / per-alignment work (getConPrb, update) runs on the GPU
// (rsem_amd/csrc/model.hip); what is here is O(table size) or O(masked positions) per round.
#pragma once
#include "files.hpp"

namespace rsemh {

constexpr int kQSize = 100;   // QProfile.h:36, NoiseQProfile.h:43, QualDist.h:31
constexpr int kNCodes = 5;
constexpr double kOriValve = 0.1;  // utils.h:21
constexpr int kRange = 201;        // utils.h:22

struct ModelParams {  // imd.mparams (rsem-calculate-expression:606-615, EM.cpp:647-658)
    int minL = 1, maxL = 1000;
    double probF = 0.5;
    bool estRSPD = false;
    int B = 20;
    int mate_minL = 1, mate_maxL = 1000;
    double mean = -1, sd = 0;
    int seedLen = 0;
};

inline ModelParams load_mparams(const std::string& path) {
    FILE* fi = fopen(path.c_str(), "r");
    if (!fi) die("Cannot open %sIt may not exist.", path.c_str());
    ModelParams P;
    int est;
    if (fscanf(fi, "%d %d %lf %d %d %d %d %lf %lf %d", &P.minL, &P.maxL, &P.probF, &est, &P.B, &P.mate_minL, &P.mate_maxL,
               &P.mean, &P.sd, &P.seedLen) != 10)
        die("%s: malformed", path.c_str());
    P.estRSPD = est != 0;
    fclose(fi);
    return P;
}

struct RSPD {  // RSPD.h
    bool est = false;
    int B = 20;
    std::vector<double> pdf, cdf;  // [B+2]
    void reset(bool estRSPD, int B_) {
        est = estRSPD; B = B_;
        pdf.assign(B + 2, 0.0); cdf.assign(B + 2, 0.0);
        for (int i = 1; i <= B; i++) { pdf[i] = 1.0 / B; cdf[i] = i * 1.0 / B; }
    }
    double evalCDF(int fpos, int fullLen) const {  // RSPD.h:63-68
        int i = (int)(((long long)fpos) * B / fullLen);
        double val = fpos * 1.0 / fullLen * B;
        return cdf[i] + (val - i) * pdf[i + 1];
    }
    double adjusted(int fpos, int effL, int fullLen) const {  // RSPD.h:70-75
        if (!est) return 1.0 / effL;
        double denom = evalCDF(effL, fullLen);
        return denom >= kEpsilon ? (evalCDF(fpos + 1, fullLen) - evalCDF(fpos, fullLen)) / denom : 0.0;
    }
    void finish() {  // RSPD.h:116-127
        double sum = 0.0;
        for (int i = 1; i <= B; i++) sum += pdf[i];
        for (int i = 1; i <= B; i++) { pdf[i] /= sum; cdf[i] = cdf[i - 1] + pdf[i]; }
    }
    void read(FILE* fi) {  // RSPD.h:137-167
        int val;
        if (fscanf(fi, "%d", &val) != 1) die("model file: bad RSPD");
        if (val != 0) {
            if (fscanf(fi, "%d", &B) != 1) die("model file: bad RSPD");
            est = true;
            pdf.assign(B + 2, 0.0); cdf.assign(B + 2, 0.0);
            for (int i = 1; i <= B; i++) {
                if (fscanf(fi, "%lf", &pdf[i]) != 1) die("model file: bad RSPD");
                cdf[i] = cdf[i - 1] + pdf[i];
            }
        } else reset(false, 20);
    }
    void write(FILE* fo) const {  // RSPD.h:169-178
        fprintf(fo, "%d\n", est ? 1 : 0);
        if (est) {
            fprintf(fo, "%d\n", B);
            for (int i = 1; i < B; i++) fprintf(fo, "%.10g ", pdf[i]);
            fprintf(fo, "%.10g\n", pdf[B]);
        }
    }
};

// All tables of one model instance, laid out flat so they can be handed to the device as they are.
struct Model {
    int type = 0;  // 0 Single, 1 SingleQ, 2 PairedEnd, 3 PairedEndQ
    int M = 0;
    ModelParams P;
    double probF = 0.5;
    LenDist gld, mld;
    bool has_mld = false;         // SE: only with --fragment-length-mean; PE: always
    RSPD rspd;
    // quality models (types 1, 3)
    std::vector<double> qd_init, qd_tran;  // [100], [100*100]   QualDist.h
    std::vector<double> qpro;              // [100*5*5]          QProfile.h
    std::vector<double> nq_c, nq_p;        // [100*5] counts in N0 reads / probabilities   NoiseQProfile.h
    // no-quality models (types 0, 2)
    int proLen = 0;
    std::vector<double> pro;               // [proLen*5*5]       Profile.h
    double np_c[kNCodes] = {0, 0, 0, 0, 0}, np_p[kNCodes] = {0, 0, 0, 0, 0};  // NoiseProfile.h
    std::vector<double> mw;                // [M+1]
    std::vector<char> needs_mw;            // [M+1] the transcript has a masked position or a poly(A) tail (found once: calc_mw runs every round)
    bool needCalcConPrb = true;

    bool hasQ() const { return type == 1 || type == 3; }
    bool paired() const { return type >= 2; }

    // master-model construction (SingleQModel.h:56-81, PairedEndQModel.h:53-77 and the no-Q twins)
    void init_master(int type_, int M_, const ModelParams& P_) {
        type = type_; M = M_; P = P_;
        probF = P.probF;
        gld.reset(P.minL, P.maxL);
        if (paired()) { has_mld = true; mld.reset(P.mate_minL, P.mate_maxL); }
        else if (P.mean >= kEpsilon) { has_mld = true; mld.reset(P.mate_minL, P.mate_maxL); }
        rspd.reset(P.estRSPD, P.estRSPD ? P.B : 20);
        if (hasQ()) {
            qd_init.assign(kQSize, 0.0); qd_tran.assign(kQSize * kQSize, 0.0);
            qpro.assign(kQSize * 25, 0.0);
            default_qprofile();
            nq_c.assign(kQSize * kNCodes, 0.0); nq_p.assign(kQSize * kNCodes, 0.0);
        } else {
            proLen = P.maxL;  // Profile(params.maxL), SingleModel.h:76
            pro.assign((size_t)proLen * 25, 0.0);
            default_profile();
        }
    }

    void default_qprofile() {  // QProfile.h:52-80
        const int N = kNCodes - 1;
        const double probN = 1e-5;
        for (int i = 0; i < kQSize; i++) {
            for (int j = 0; j < kNCodes - 1; j++) {
                qpro[(i * 5 + j) * 5 + N] = probN;
                double probO = exp(-i / 10.0 * log(10.0));
                double probC = 1.0 - probO;
                probO /= (kNCodes - 2);
                probC *= (1.0 - probN);
                probO *= (1.0 - probN);
                for (int k = 0; k < kNCodes - 1; k++) qpro[(i * 5 + j) * 5 + k] = (j == k) ? probC : probO;
            }
            qpro[(i * 5 + N) * 5 + N] = probN;
            for (int k = 0; k < kNCodes - 1; k++) qpro[(i * 5 + N) * 5 + k] = (1.0 - probN) / (kNCodes - 1);
        }
    }
    void default_profile() {  // Profile.h:48-72
        const int N = kNCodes - 1;
        const double probN = 1e-5, portionC = 0.99;
        for (int i = 0; i < proLen; i++) {
            for (int j = 0; j < kNCodes - 1; j++) {
                pro[((size_t)i * 5 + j) * 5 + N] = probN;
                double probC = portionC * (1.0 - probN);
                double probO = (1.0 - portionC) / (kNCodes - 2) * (1.0 - probN);
                for (int k = 0; k < kNCodes - 1; k++) pro[((size_t)i * 5 + j) * 5 + k] = (j == k) ? probC : probO;
            }
            pro[((size_t)i * 5 + N) * 5 + N] = probN;
            for (int k = 0; k < kNCodes - 1; k++) pro[((size_t)i * 5 + N) * 5 + k] = (1.0 - probN) / (kNCodes - 1);
        }
    }

    // ---- between rounds: model.init(); collect(helpers); finish()  (EM.cpp:400-404) ----------------
    // `acc` holds the sums the device accumulated this round (the helpers' tables, already merged).
    struct Accum {
        std::vector<double> prof;   // [100*25] or [proLen*25]
        std::vector<double> noise;  // [100*5] or [5]
        std::vector<double> rspd;   // [B+2]
        std::vector<double> gld;    // [span0+1] over the ORIGINAL (minL-1, maxL] support (PE only)
    };

    void finish_round(const Accum& acc, const RefInfo& R) {
        // (N)(Q)Profile::finish -- row-normalise p[.][r][.]  (QProfile.h:95-109, Profile.h:98-112)
        std::vector<double>& T = hasQ() ? qpro : pro;
        const size_t rows = T.size() / 5;
        for (size_t r = 0; r < rows; r++) {
            double sum = 0.0;
            for (int k = 0; k < 5; k++) sum += acc.prof[r * 5 + k];
            if (sum < kEpsilon) { for (int k = 0; k < 5; k++) T[r * 5 + k] = 0.0; continue; }
            for (int k = 0; k < 5; k++) T[r * 5 + k] = acc.prof[r * 5 + k] / sum;
        }
        if (hasQ()) {  // NoiseQProfile::finish (NoiseQProfile.h:81-97)
            for (int i = 0; i < kQSize; i++) {
                double sum = 0.0;
                for (int j = 0; j < kNCodes; j++) sum += (acc.noise[i * 5 + j] + nq_c[i * 5 + j]);
                if (sum <= 0.0) { for (int j = 0; j < kNCodes; j++) nq_p[i * 5 + j] = acc.noise[i * 5 + j]; continue; }
                for (int j = 0; j < kNCodes; j++) nq_p[i * 5 + j] = (acc.noise[i * 5 + j] + nq_c[i * 5 + j]) / sum;
            }
        } else {  // NoiseProfile::finish (NoiseProfile.h:70-82)
            double sum = 0.0;
            for (int i = 0; i < kNCodes; i++) sum += (acc.noise[i] + np_c[i]);
            if (sum <= kEpsilon) { for (int i = 0; i < kNCodes; i++) np_p[i] = acc.noise[i]; }
            else for (int i = 0; i < kNCodes; i++) np_p[i] = (acc.noise[i] + np_c[i]) / sum;
        }
        if (P.estRSPD) {
            for (int i = 0; i <= rspd.B + 1; i++) { rspd.pdf[i] = acc.rspd[i]; rspd.cdf[i] = 0.0; }
            rspd.finish();
        }
        if (paired()) {  // gld re-estimated from the fragments (PairedEndQModel.h:161-168,299-306; LenDist::collect)
            gld.lb = P.minL - 1; gld.ub = P.maxL; gld.span = gld.ub - gld.lb;
            gld.pdf.assign(gld.span + 1, 0.0); gld.cdf.assign(gld.span + 1, 0.0);
            for (int i = 1; i <= gld.span; i++) gld.pdf[i] = acc.gld[i];
            gld.finish();
        }
        needCalcConPrb = true;
        if (paired() || P.estRSPD) calc_mw(R);  // PairedEndQModel.h:305; SingleQModel.h:340
    }

    // ---- mask weights (SingleQModel.h:482-544, PairedEndQModel.h:445-479) -----------------------------
    static double adj_prob(const LenDist& d, int len, int refL) {  // LenDist.h:63-68
        if (len <= d.lb || len > d.ub || refL <= d.lb) return 0.0;
        return d.pdf[len - d.lb] / d.cdf[std::min(d.ub, refL) - d.lb];
    }
    static double adj_cum(const LenDist& d, int len, int refL) {   // LenDist.h:72-77
        return d.cdf[len - d.lb] / d.cdf[std::min(d.ub, refL) - d.lb];
    }
    void calc_mw(const RefInfo& R) {
        mw.assign(M + 1, 0.0);
        mw[0] = 1.0;
        if ((int)needs_mw.size() != M + 1) {  // (the references do not change between the rounds)
            needs_mw.assign(M + 1, 0);
            for (int i = 1; i <= M; i++) {
                bool any_mask = R.totLen[i] != R.fullLen[i];
                for (size_t w = 0; !any_mask && w < R.masks[i].size(); w++) any_mask = R.masks[i][w] != 0;
                needs_mw[i] = any_mask ? 1 : 0;
            }
        }
        const double probR = 1.0 - probF;
        const int seedLen = P.seedLen;
        for (int i = 1; i <= M; i++) {
            const int totLen = R.totLen[i], fullLen = R.fullLen[i];
            double value = 0.0;
            // nothing to integrate for a transcript without masked positions and without a poly(A) tail: every loop
            // below is either guarded by getMask() or runs over [fullLen, totLen) -- mw = 1 exactly as in the reference
            if (!needs_mw[i]) { mw[i] = 1.0; continue; }
            if (paired()) {
                const int end = std::min(fullLen, totLen - gld.minL() + 1);
                for (int seedPos = 0; seedPos < end; seedPos++)
                    if (R.mask_at(i, seedPos)) {
                        const int maxL = std::min(gld.maxL(), totLen - seedPos);
                        for (int fragLen = gld.minL(); fragLen <= maxL; fragLen++) {
                            int effL = std::min(fullLen, totLen - fragLen + 1);
                            value += adj_prob(gld, fragLen, totLen) * rspd.adjusted(seedPos, effL, fullLen);
                        }
                    }
            } else {
                const int end = std::min(fullLen, totLen - seedLen + 1);
                for (int seedPos = 0; seedPos < end; seedPos++)
                    if (R.mask_at(i, seedPos)) {
                        int minL = gld.minL();
                        int maxL = std::min(gld.maxL(), totLen - seedPos);
                        for (int fragLen = minL; fragLen <= maxL; fragLen++) {  // forward
                            int effL = std::min(fullLen, totLen - fragLen + 1);
                            double factor = has_mld ? adj_cum(mld, std::min(mld.maxL(), fragLen), fragLen) : 1.0;
                            value += probF * adj_prob(gld, fragLen, totLen) * rspd.adjusted(seedPos, effL, fullLen) * factor;
                        }
                        maxL = std::min(gld.maxL(), seedPos + seedLen);
                        for (int fragLen = minL; fragLen <= maxL; fragLen++) {  // reverse
                            int pfpos = seedPos - (fragLen - seedLen);
                            int effL = std::min(fullLen, totLen - fragLen + 1);
                            double factor = has_mld ? adj_cum(mld, std::min(mld.maxL(), fragLen), fragLen) : 1.0;
                            value += probR * adj_prob(gld, fragLen, totLen) * rspd.adjusted(pfpos, effL, fullLen) * factor;
                        }
                    }
                for (int seedPos = end; seedPos <= totLen - seedLen; seedPos++) {  // reverse strand masking
                    int minL = std::max(gld.minL(), seedPos + seedLen - fullLen + 1);
                    int maxL = std::min(gld.maxL(), seedPos + seedLen);
                    for (int fragLen = minL; fragLen <= maxL; fragLen++) {
                        int pfpos = seedPos - (fragLen - seedLen);
                        int effL = std::min(fullLen, totLen - fragLen + 1);
                        double factor = has_mld ? adj_cum(mld, std::min(mld.maxL(), fragLen), fragLen) : 1.0;
                        value += probR * adj_prob(gld, fragLen, totLen) * rspd.adjusted(pfpos, effL, fullLen) * factor;
                    }
                }
            }
            mw[i] = 1.0 - value;
            if (mw[i] < 1e-8) mw[i] = 0.0;
        }
    }

    // ---- .model file (SingleQModel.h:349-411 and twins; model_file_description.txt) -------------------
    void write(const std::string& path) const {
        FILE* fo = fopen(path.c_str(), "w");
        if (!fo) die("Cannot open %s for writing!", path.c_str());
        fprintf(fo, "%d\n\n", type);
        fprintf(fo, "%.10g\n\n", probF);
        gld.write(fo); fprintf(fo, "\n");
        if (!paired()) {
            if (has_mld) { fprintf(fo, "1\n"); mld.write(fo); } else fprintf(fo, "0\n");
            fprintf(fo, "\n");
        } else { mld.write(fo); fprintf(fo, "\n"); }
        rspd.write(fo); fprintf(fo, "\n");
        if (hasQ()) {
            fprintf(fo, "%d\n", kQSize);  // QualDist::write (QualDist.h:97-106)
            for (int i = 0; i < kQSize - 1; i++) fprintf(fo, "%.10g ", qd_init[i]);
            fprintf(fo, "%.10g\n", qd_init[kQSize - 1]);
            for (int i = 0; i < kQSize; i++) {
                for (int j = 0; j < kQSize - 1; j++) fprintf(fo, "%.10g ", qd_tran[i * kQSize + j]);
                fprintf(fo, "%.10g\n", qd_tran[i * kQSize + kQSize - 1]);
            }
            fprintf(fo, "\n");
            fprintf(fo, "%d %d\n", kQSize, kNCodes);  // QProfile::write (QProfile.h:139-149)
            for (int i = 0; i < kQSize; i++) {
                for (int j = 0; j < kNCodes; j++) {
                    for (int k = 0; k < kNCodes - 1; k++) fprintf(fo, "%.10g ", qpro[(i * 5 + j) * 5 + k]);
                    fprintf(fo, "%.10g\n", qpro[(i * 5 + j) * 5 + kNCodes - 1]);
                }
                if (i < kQSize - 1) fprintf(fo, "\n");
            }
            fprintf(fo, "\n");
            fprintf(fo, "%d %d\n", kQSize, kNCodes);  // NoiseQProfile::write (NoiseQProfile.h:154-160)
            for (int i = 0; i < kQSize; i++) {
                for (int j = 0; j < kNCodes - 1; j++) fprintf(fo, "%.10g ", nq_p[i * 5 + j]);
                fprintf(fo, "%.10g\n", nq_p[i * 5 + kNCodes - 1]);
            }
        } else {
            fprintf(fo, "%d %d\n", proLen, kNCodes);  // Profile::write (Profile.h:150-160)
            for (int i = 0; i < proLen; i++) {
                for (int j = 0; j < kNCodes; j++) {
                    for (int k = 0; k < kNCodes - 1; k++) fprintf(fo, "%.10g ", pro[((size_t)i * 5 + j) * 5 + k]);
                    fprintf(fo, "%.10g\n", pro[((size_t)i * 5 + j) * 5 + kNCodes - 1]);
                }
                if (i < proLen - 1) fprintf(fo, "\n");
            }
            fprintf(fo, "\n");
            fprintf(fo, "%d\n", kNCodes);  // NoiseProfile::write (NoiseProfile.h:119-125)
            for (int i = 0; i < kNCodes - 1; i++) fprintf(fo, "%.10g ", np_p[i]);
            fprintf(fo, "%.10g\n", np_p[kNCodes - 1]);
        }
        if (!mw.empty()) {
            fprintf(fo, "\n%d\n", M);
            write_cells_line(fo, 0, M, ' ', [&](char* b, long i) { return snprintf(b, rsemh::kCellBuf, "%.15g", mw[i]); });
        }
        fclose(fo);
    }

    // read(): what Gibbs / calcCI need is gld (for eel) and mw; every table is parsed so the position is right
    void read(const std::string& path, int M_expected) {
        FILE* fi = fopen(path.c_str(), "r");
        if (!fi) die("Cannot open %s! It may not exist.", path.c_str());
        if (fscanf(fi, "%d", &type) != 1) die("%s: empty model file", path.c_str());
        if (fscanf(fi, "%lf", &probF) != 1) die("%s: bad model file", path.c_str());
        gld.read(fi);
        has_mld = false;
        if (!paired()) {
            int val;
            if (fscanf(fi, "%d", &val) != 1) die("%s: bad model file", path.c_str());
            if (val > 0) { has_mld = true; mld.read(fi); }
        } else { has_mld = true; mld.read(fi); }
        rspd.read(fi);
        auto rd = [&](std::vector<double>& v, size_t n) {
            v.resize(n);
            for (size_t i = 0; i < n; i++)
                if (fscanf(fi, "%lf", &v[i]) != 1) die("%s: bad model file", path.c_str());
        };
        int a, b;
        if (hasQ()) {
            if (fscanf(fi, "%d", &a) != 1 || a != kQSize) die("%s: bad QualDist", path.c_str());
            rd(qd_init, kQSize); rd(qd_tran, (size_t)kQSize * kQSize);
            if (fscanf(fi, "%d %d", &a, &b) != 2 || a != kQSize || b != kNCodes) die("%s: bad QProfile", path.c_str());
            rd(qpro, (size_t)kQSize * 25);
            if (fscanf(fi, "%d %d", &a, &b) != 2 || a != kQSize || b != kNCodes) die("%s: bad NoiseQProfile", path.c_str());
            rd(nq_p, (size_t)kQSize * 5);
            nq_c.assign((size_t)kQSize * 5, 0.0);
        } else {
            if (fscanf(fi, "%d %d", &proLen, &b) != 2 || b != kNCodes) die("%s: bad Profile", path.c_str());
            rd(pro, (size_t)proLen * 25);
            if (fscanf(fi, "%d", &b) != 1 || b != kNCodes) die("%s: bad NoiseProfile", path.c_str());
            for (int i = 0; i < kNCodes; i++)
                if (fscanf(fi, "%lf", &np_p[i]) != 1) die("%s: bad NoiseProfile", path.c_str());
        }
        mw.clear();
        int val;
        if (fscanf(fi, "%d", &val) == 1) {
            M = val;
            if (M_expected == 0 || M_expected == val) rd(mw, (size_t)val + 1);
        }
        fclose(fi);
    }
};

}  // namespace rsemh
