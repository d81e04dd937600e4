/* rsem_oracle.c -- TEST INFRASTRUCTURE ONLY (see rsem_oracle.h).
 *
 * Each function names the reference lines (under /root/reference) whose arithmetic it restates.
 * Compiled with -ffp-contract=off and without -ffast-math: every sum is taken in the reference's
 * left-to-right order, so results are reproducible bit for bit across runs.
 */
#include "rsem_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EPSILON 1e-300 /* utils.h:19 */
#define ORC_MINEEL 1.0     /* utils.h:20 */
#define ORC_STOP 0.001     /* EM.cpp:53 */

/* EM.cpp:198-244 */
void orc_em_estep(int32_t M, uint64_t N1, const uint64_t* row_ptr, const int32_t* sid,
                  const double* conprb, const double* ncp, const double* theta,
                  double* counts, double* w, double* w_noise) {
    uint64_t maxlen = 0;
    for (uint64_t i = 0; i < N1; i++)
        if (row_ptr[i + 1] - row_ptr[i] > maxlen) maxlen = row_ptr[i + 1] - row_ptr[i];
    double* fracs = (double*)malloc(sizeof(double) * (maxlen + 1));
    memset(counts, 0, sizeof(double) * ((size_t)M + 1));
    for (uint64_t i = 0; i < N1; i++) {
        uint64_t fr = row_ptr[i], to = row_ptr[i + 1];
        double sum = 0.0;
        fracs[0] = theta[0] * ncp[i]; /* EM.cpp:211 */
        if (fracs[0] < ORC_EPSILON) fracs[0] = 0.0;
        sum += fracs[0];
        for (uint64_t j = fr; j < to; j++) { /* EM.cpp:214-221 */
            double f = theta[sid[j]] * conprb[j];
            if (f < ORC_EPSILON) f = 0.0;
            fracs[j - fr + 1] = f;
            sum += f;
        }
        if (sum >= ORC_EPSILON) { /* EM.cpp:223-236 */
            fracs[0] /= sum;
            counts[0] += fracs[0];
            if (w_noise) w_noise[i] = fracs[0];
            for (uint64_t j = fr; j < to; j++) {
                double f = fracs[j - fr + 1] / sum;
                counts[sid[j]] += f;
                if (w) w[j] = f;
            }
        } else { /* EM.cpp:237-243 */
            if (w_noise) w_noise[i] = 0.0;
            if (w)
                for (uint64_t j = fr; j < to; j++) w[j] = 0.0;
        }
    }
    free(fracs);
}

/* EM.cpp:391-413 */
void orc_em_mstep(int32_t M, double N0, double* counts, const double* theta_old,
                  double* theta_new, double* sum_out, double* bChange_out, int32_t* totNum_out) {
    counts[0] += N0;
    double sum = 0.0;
    for (int32_t i = 0; i <= M; i++) sum += counts[i];
    for (int32_t i = 0; i <= M; i++) theta_new[i] = counts[i] / sum;
    double bChange = 0.0;
    int32_t totNum = 0;
    for (int32_t i = 0; i <= M; i++)
        if (theta_old[i] >= 1e-7) {
            double change = fabs(theta_new[i] - theta_old[i]) / theta_old[i];
            if (change >= ORC_STOP) ++totNum;
            if (bChange < change) bChange = change;
        }
    *sum_out = sum;
    *bChange_out = bChange;
    *totNum_out = totNum;
}

/* EM.cpp:365-416 (rounds with needCalcConPrb == updateModel == false) */
int orc_em_run(int32_t M, uint64_t N1, const uint64_t* row_ptr, const int32_t* sid,
               const double* conprb, const double* ncp, double N0, double* theta,
               int round0, int min_round, int max_round, double* bChange, int32_t* totNum) {
    double* probv = (double*)malloc(sizeof(double) * ((size_t)M + 1));
    double* counts = (double*)malloc(sizeof(double) * ((size_t)M + 1));
    int ROUND = round0;
    double sum;
    do {
        ++ROUND;
        memcpy(probv, theta, sizeof(double) * ((size_t)M + 1));
        orc_em_estep(M, N1, row_ptr, sid, conprb, ncp, probv, counts, NULL, NULL);
        orc_em_mstep(M, N0, counts, probv, theta, &sum, bChange, totNum);
    } while (ROUND < min_round || (*totNum > 0 && ROUND < max_round));
    free(probv);
    free(counts);
    return ROUND;
}

/* WriteResults.h:24-53 */
void orc_calc_eel(int32_t M, const int32_t* fullLen, const int32_t* totLen, int lb, int ub, int span,
                  const double* pdf, const double* cdf, double* eel) {
    double* clen = (double*)malloc(sizeof(double) * ((size_t)span + 1));
    clen[0] = 0.0;
    for (int i = 1; i <= span; i++) clen[i] = clen[i - 1] + pdf[i] * (lb + i);
    eel[0] = 0.0;
    for (int32_t i = 1; i <= M; i++) {
        int tl = totLen[i], fl = fullLen[i];
        int a = tl - fl + 1;
        int pos1 = (a < ub ? a : ub) - lb;
        if (pos1 < 0) pos1 = 0;
        int pos2 = (tl < ub ? tl : ub) - lb;
        if (pos2 < 0) pos2 = 0;
        if (pos2 == 0) { eel[i] = 0.0; continue; }
        eel[i] = fl * cdf[pos1] + ((cdf[pos2] - cdf[pos1]) * (tl + 1) - (clen[pos2] - clen[pos1]));
        if (eel[i] < ORC_MINEEL) eel[i] = 0.0;
    }
    free(clen);
}

/* WriteResults.h:55-75 */
int orc_polish_theta(int32_t M, double* theta, const double* eel, const double* mw) {
    double sum = 0.0;
    for (int32_t i = 0; i <= M; i++) {
        if (i > 0 && (mw[i] < ORC_EPSILON || eel[i] < ORC_EPSILON)) { theta[i] = 0.0; continue; }
        theta[i] = theta[i] / mw[i];
        sum += theta[i];
    }
    if (!(sum >= ORC_EPSILON)) return -1;
    for (int32_t i = 0; i <= M; i++) theta[i] /= sum;
    return 0;
}

/* WriteResults.h:77-104 */
void orc_calc_expression(int32_t M, const double* theta, const double* eel, double* tpm, double* fpkm) {
    double denom = 0.0;
    double* frac = (double*)calloc((size_t)M + 1, sizeof(double));
    for (int32_t i = 1; i <= M; i++)
        if (eel[i] >= ORC_EPSILON) { frac[i] = theta[i]; denom += frac[i]; }
    if (denom < ORC_EPSILON) denom = 1.0;
    for (int32_t i = 1; i <= M; i++) frac[i] /= denom;
    for (int32_t i = 0; i <= M; i++) fpkm[i] = 0.0;
    for (int32_t i = 1; i <= M; i++)
        if (eel[i] >= ORC_EPSILON) fpkm[i] = frac[i] * 1e9 / eel[i];
    for (int32_t i = 0; i <= M; i++) tpm[i] = 0.0;
    denom = 0.0;
    for (int32_t i = 1; i <= M; i++) denom += fpkm[i];
    if (denom < ORC_EPSILON) denom = 1.0;
    for (int32_t i = 1; i <= M; i++) tpm[i] = fpkm[i] / denom * 1e6;
    free(frac);
}

/* boost::random::mt19937 (Boost 1.55 mersenne_twister.hpp): the published MT19937 with
 * seeding x[i] = 1812433253 * (x[i-1] ^ (x[i-1] >> 30)) + i. */
void orc_mt_seed(orc_mt19937* g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

uint32_t orc_mt_next(orc_mt19937* g) {
    if (g->idx >= 624) {
        for (int k = 0; k < 624; k++) {
            uint32_t y = (g->mt[k] & 0x80000000u) | (g->mt[(k + 1) % 624] & 0x7fffffffu);
            g->mt[k] = g->mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* sampling.h:24-38: successive outputs of the seed engine, skipping repeats */
void orc_chain_seeds(uint32_t seed, int nchains, uint32_t* out) {
    orc_mt19937 g;
    orc_mt_seed(&g, seed);
    int n = 0;
    while (n < nchains) {
        uint32_t s = orc_mt_next(&g);
        int dup = 0;
        for (int i = 0; i < n; i++) if (out[i] == s) { dup = 1; break; }
        if (!dup) out[n++] = s;
    }
}

/* sampling.h:50-65 with uniform_01 = mt() * 2^-32 (boost/random/uniform_01.hpp:91-102) */
static int orc_sample(orc_mt19937* g, const double* arr, int len) {
    double u = (double)orc_mt_next(g) * (1.0 / 4294967296.0);
    double prb = u * arr[len - 1];
    int l = 0, r = len - 1;
    while (l <= r) {
        int mid = (l + r) / 2;
        if (arr[mid] <= prb) l = mid + 1; else r = mid - 1;
    }
    if (l >= len) l = len - 1; /* the reference asserts here; unreachable for arr[len-1] > 0 */
    return l;
}

/* Gibbs.cpp:265-353 */
void orc_gibbs_chain(int32_t M, uint64_t N1, const uint64_t* s, const int32_t* sid,
                     const double* conprb, const int32_t* init_counts, const double* alpha,
                     double pseudoC, double totc, uint64_t N0, const double* eel, const double* mw,
                     int32_t m, const int32_t* grp, uint32_t mt_seed, int burnin, int nsamples,
                     int gap, int32_t* count_vectors, double* pme_c, double* pve_c,
                     double* pme_tpm, double* pme_fpkm, double* pve_c_genes) {
    orc_mt19937 g;
    orc_mt_seed(&g, mt_seed);
    uint64_t maxlen = 1;
    for (uint64_t i = 0; i < N1; i++) if (s[i + 1] - s[i] > maxlen) maxlen = s[i + 1] - s[i];
    double* arr = (double*)malloc(sizeof(double) * maxlen);
    int32_t* z = (int32_t*)calloc(N1 ? N1 : 1, sizeof(int32_t));
    int32_t* counts = (int32_t*)malloc(sizeof(int32_t) * ((size_t)M + 1));
    double* theta = (double*)calloc((size_t)M + 1, sizeof(double));
    double* tpm = (double*)calloc((size_t)M + 1, sizeof(double));
    double* fpkm = (double*)calloc((size_t)M + 1, sizeof(double));
    memcpy(counts, init_counts, sizeof(int32_t) * ((size_t)M + 1));
    counts[0] += (int32_t)N0; /* Gibbs.cpp:279 */

    for (uint64_t i = 0; i < N1; i++) { /* Gibbs.cpp:281-291 */
        uint64_t fr = s[i], to = s[i + 1];
        int len = (int)(to - fr);
        for (uint64_t j = fr; j < to; j++) {
            arr[j - fr] = conprb[j];
            if (j > fr) arr[j - fr] += arr[j - fr - 1];
        }
        z[i] = sid[fr + orc_sample(&g, arr, len)];
        ++counts[z[i]];
    }

    int chainlen = 1 + (nsamples - 1) * gap;
    int kept = 0;
    for (int ROUND = 1; ROUND <= burnin + chainlen; ROUND++) { /* Gibbs.cpp:295 */
        for (uint64_t i = 0; i < N1; i++) {                   /* Gibbs.cpp:297-311 */
            --counts[z[i]];
            uint64_t fr = s[i], to = s[i + 1];
            int len = (int)(to - fr);
            for (uint64_t j = fr; j < to; j++) {
                double a = alpha ? alpha[sid[j]] : pseudoC;
                arr[j - fr] = (counts[sid[j]] + a) * conprb[j];
                if (j > fr) arr[j - fr] += arr[j - fr - 1];
            }
            z[i] = sid[fr + orc_sample(&g, arr, len)];
            ++counts[z[i]];
        }
        if (ROUND > burnin && (ROUND - burnin - 1) % gap == 0) { /* Gibbs.cpp:313-346 */
            if (count_vectors) memcpy(count_vectors + (size_t)kept * ((size_t)M + 1), counts, sizeof(int32_t) * ((size_t)M + 1));
            ++kept;
            for (int32_t i = 0; i <= M; i++) {
                double a = alpha ? alpha[i] : pseudoC;
                theta[i] = (counts[i] < 0 ? 0.0 : (counts[i] + a) / totc);
            }
            orc_polish_theta(M, theta, eel, mw);
            orc_calc_expression(M, theta, eel, tpm, fpkm);
            for (int32_t i = 0; i <= M; i++) {
                pme_c[i] += counts[i];
                pve_c[i] += (double)counts[i] * counts[i];
                pme_tpm[i] += tpm[i];
                pme_fpkm[i] += fpkm[i];
            }
            for (int32_t i = 0; i < m; i++) {
                double c = 0.0;
                for (int32_t j = grp[i]; j < grp[i + 1]; j++) c += counts[j];
                pve_c_genes[i] += c * c;
            }
        }
    }
    free(arr); free(z); free(counts); free(theta); free(tpm); free(fpkm);
}


/* ---- credibility intervals ----------------------------------------------------------------------------- */

static int cmp_float(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* calcCI.cpp:216-284 */
void orc_calc_ci(int nSamples, float* samples, double confidence, float* lb_out, float* ub_out, float* cqv_out) {
    int p, q, newp, newq;
    const int threshold = nSamples - ((int)(confidence * nSamples - 1e-8) + 1);
    int nOutside = 0;
    float lb, ub;

    qsort(samples, (size_t)nSamples, sizeof(float), cmp_float);

    p = 0; q = nSamples - 1;
    newq = nSamples - 1;
    do {
        q = newq;
        while (newq > 0 && samples[newq - 1] == samples[newq]) newq--;
        newq--;
    } while (newq >= 0 && nSamples - (newq + 1) <= threshold);

    nOutside = nSamples - (q + 1);

    lb = -1e30f; ub = 1e30f;
    do {
        if (samples[q] - samples[p] < ub - lb) {
            lb = samples[p];
            ub = samples[q];
        }
        newp = p;
        while (newp < nSamples - 1 && samples[newp] == samples[newp + 1]) newp++;
        newp++;
        if (newp <= threshold) {
            nOutside += newp - p;
            p = newp;
            while (nOutside > threshold && q < nSamples - 1) {
                newq = q + 1;
                while (newq < nSamples - 1 && samples[newq] == samples[newq + 1]) newq++;
                nOutside -= newq - q;
                q = newq;
            }
        } else p = newp;
    } while (p <= threshold);

    {
        float Q1, Q3;
        const int quotient = nSamples / 4, residue = nSamples % 4;
        if (residue == 0) {
            Q1 = (float)((samples[quotient - 1] + samples[quotient]) / 2.0);
            Q3 = (float)((samples[3 * quotient - 1] + samples[3 * quotient]) / 2.0);
        } else if (residue == 3) {
            Q1 = (float)((samples[quotient] + samples[quotient + 1]) / 2.0);
            Q3 = (float)((samples[quotient * 3 + 1] + samples[quotient * 3 + 2]) / 2.0);
        } else {
            Q1 = samples[quotient];
            Q3 = samples[3 * quotient];
        }
        *cqv_out = (float)(Q3 - Q1 > 0.0 ? (Q3 - Q1) / (Q3 + Q1) : 0.0);
    }
    *lb_out = lb; *ub_out = ub;
}

/* calcCI.cpp:129-149 */
float orc_ci_transform(int32_t M, const double* gam, const int32_t* cvec, const double* eel, const double* mw,
                       float* tpm) {
    const double EPSILON = 1e-300;
    double* theta = (double*)malloc(sizeof(double) * ((size_t)M + 1));
    double sum = 0.0;
    float l_bar;
    int j;
    for (j = 0; j <= M; j++) {
        theta[j] = ((j == 0 || (cvec[j] >= 0 && eel[j] >= EPSILON && mw[j] >= EPSILON)) ? gam[j] / mw[j] : 0.0);
        sum += theta[j];
    }
    if (!(sum >= EPSILON)) { free(theta); return -1.0f; }
    for (j = 0; j <= M; j++) theta[j] /= sum;
    sum = 0.0;
    tpm[0] = 0.0f;
    for (j = 1; j <= M; j++) {
        if (eel[j] >= EPSILON) {
            tpm[j] = (float)(theta[j] / eel[j]);
            sum += tpm[j];
        } else tpm[j] = 0.0f;
    }
    if (!(sum >= EPSILON)) { free(theta); return -1.0f; }
    l_bar = 0.0f;
    for (j = 1; j <= M; j++) {
        tpm[j] = (float)(tpm[j] / sum);
        l_bar = (float)(l_bar + tpm[j] * eel[j]);
        tpm[j] = (float)(tpm[j] * 1e6);
    }
    free(theta);
    return l_bar;
}


/* ---- data-augmentation sampler (CPU model of the drop-in's PARALLEL Gibbs mode) ---------------------------- */

static double da_u01(orc_mt19937* g) {  /* (0,1), 53 bits */
    const uint32_t a = orc_mt_next(g) >> 5, b = orc_mt_next(g) >> 6;
    return ((double)a * 67108864.0 + (double)b + 0.5) * (1.0 / 9007199254740992.0);
}

static double da_gamma(orc_mt19937* g, double a) {  /* Marsaglia & Tsang (2000) */
    double boost = 1.0, d, c;
    if (a < 1.0) { boost = exp(log(da_u01(g)) / a); a += 1.0; }
    d = a - 1.0 / 3.0; c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        const double u1 = da_u01(g), u2 = da_u01(g);
        const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        double v = 1.0 + c * x, u, x2;
        if (v <= 0.0) continue;
        v = v * v * v;
        u = da_u01(g);
        x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2) return d * v * boost;
        if (log(u) < 0.5 * x2 + d * (1.0 - v + log(v))) return d * v * boost;
    }
}

void orc_gibbs_da_chain(int32_t M, uint64_t N1, const uint64_t* row_ptr, const int32_t* sid,
                        const double* conprb, const int32_t* init_counts, double pseudoC, uint64_t N0,
                        uint32_t mt_seed, int burnin, int nsamples, int gap, int thin, double* pme_c, double* pve_c) {
    orc_mt19937 g;
    double* th = (double*)malloc(sizeof(double) * ((size_t)M + 1));
    int32_t* counts = (int32_t*)malloc(sizeof(int32_t) * ((size_t)M + 1));
    const int chainlen = 1 + (nsamples - 1) * gap;
    int round, t, first = 1;
    int32_t j;
    uint64_t i, k;
    orc_mt_seed(&g, mt_seed);
    for (j = 0; j <= M; j++) th[j] = 1.0;  /* initial state: z ~ conprb alone, as Gibbs.cpp:283-293 */
    for (round = 0; round <= burnin + chainlen; round++) {
        const int sweeps = first ? 1 : thin;
        for (t = 0; t < sweeps; t++) {
            if (!first)
                for (j = 0; j <= M; j++) th[j] = counts[j] < 0 ? 0.0 : da_gamma(&g, (double)counts[j] + pseudoC);
            for (j = 0; j <= M; j++) counts[j] = init_counts[j];
            counts[0] += (int32_t)N0;
            for (i = 0; i < N1; i++) {
                double tot = 0.0, target, run = 0.0;
                int32_t pick = -1;
                for (k = row_ptr[i]; k < row_ptr[i + 1]; k++) tot += th[sid[k]] * conprb[k];
                if (!(tot > 0.0)) continue;
                target = da_u01(&g) * tot;
                for (k = row_ptr[i]; k < row_ptr[i + 1]; k++) {
                    const double f = th[sid[k]] * conprb[k];
                    run += f;
                    if (f > 0.0) pick = sid[k];
                    if (target < run) break;
                }
                if (pick >= 0) ++counts[pick];
            }
            first = 0;
        }
        if (round > burnin && (round - burnin - 1) % gap == 0)
            for (j = 0; j <= M; j++) { pme_c[j] += counts[j]; pve_c[j] += (double)counts[j] * counts[j]; }
    }
    free(th); free(counts);
}
