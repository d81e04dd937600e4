// host-only check of the general-G shape policy and index helpers of sell_layout.hpp (compiled with -DRSEM_GENERAL_G=1)
#include <cstdio>
#include <set>
#include "../rsem_amd/csrc/sell_layout.hpp"
int main() {
    uint16_t t0[257], t1[257];
    shape_policy_table(0, t0);
    shape_policy_table(1, t1);
    double b0 = 0, b1 = 0;
    int bad = 0;
    for (int L = 1; L <= 256; L++) {
        for (int pol = 0; pol < 2; pol++) {
            const int id = pol ? t1[L] : t0[L];
            const int G = id / 4 + 1, K = id % 4 + 1;
            if (G * K < L || G > 64 || K > 4) { printf("BAD L=%d pol=%d G=%d K=%d\n", L, pol, G, K); bad++; }
            Shape S{};
            int lg = 0; while ((1 << lg) < G) ++lg;
            S.lg = ((1 << lg) == G) ? lg : -G; S.K = K;
            if (shape_G(S) != G || shape_R(S) != (uint32_t)(64 / G)) { printf("BAD accessors L=%d\n", L); bad++; }
            // fill offsets of one slice: R reads, distinct entries inside K planes of 64
            std::set<int> used;
            const int R = 64 / G;
            for (int r = 0; r < R; r++) for (int c = 0; c < L; c++) {
                const int off = (c / G) * 64 + r * G + (c % G);
                if (off < 0 || off >= K * 64 || !used.insert(off).second) { printf("BAD offset L=%d pol=%d r=%d c=%d\n", L, pol, r, c); bad++; }
            }
            const double bytes = K * 64.0 / R;
            (pol ? b1 : b0) += bytes / L;
        }
        if (L <= 24) printf("L=%3d pow2 (G=%2d,K=%d) %.2f/L  minbytes (G=%2d,K=%d) %.2f/L\n", L, t0[L] / 4 + 1, t0[L] % 4 + 1, (t0[L] % 4 + 1) * 64.0 / (64 / (t0[L] / 4 + 1)) / L,
                            t1[L] / 4 + 1, t1[L] % 4 + 1, (t1[L] % 4 + 1) * 64.0 / (64 / (t1[L] / 4 + 1)) / L);
    }
    // row_to_slot: a bijection of the rows of a shape onto (slice, slot) with slot < R
    for (int G : {3, 5, 7, 12}) {
        Shape S{}; S.lg = -G; S.K = 2; S.n_rows = 1000;
        const uint32_t T = 16, R = shape_R(S);
        std::set<std::pair<uint32_t, uint32_t>> seen;
        for (uint32_t q = 0; q < S.n_rows; q++) {
            uint32_t sl, r; row_to_slot(S, T, q, sl, r);
            if (r >= R || !seen.insert({sl, r}).second || sl >= (S.n_rows + R - 1) / R + T) { printf("BAD row_to_slot G=%d q=%u\n", G, q); bad++; }
        }
    }
    printf("mean entries per alignment over L=1..256: pow2 %.4f, min-bytes %.4f; bad=%d\n", b0 / 256, b1 / 256, bad);
    return bad != 0;
}
