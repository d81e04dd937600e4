// Host-only: rsem_amd/csrc/host/posterior_moments.hpp on sums read from stdin -> means / variances on stdout
// (tests/test_capi_cpu.py compares with numpy's statement of Gibbs.cpp:389-423).
//   input: n_samples M1 m, then M1 s1, M1 s2, m+1 group starts, m group s2
#include <cstdio>
#include <vector>
#include "../rsem_amd/csrc/host/posterior_moments.hpp"
int main() {
    long n; int M1, m;
    if (scanf("%ld %d %d", &n, &M1, &m) != 3) return 2;
    std::vector<double> s1(M1), s2(M1), g2(m);
    std::vector<int> starts(m + 1);
    for (double& v : s1) if (scanf("%lf", &v) != 1) return 2;
    for (double& v : s2) if (scanf("%lf", &v) != 1) return 2;
    for (int& v : starts) if (scanf("%d", &v) != 1) return 2;
    for (double& v : g2) if (scanf("%lf", &v) != 1) return 2;
    rsem_host::finish_per_transcript(n, s1, s2);
    rsem_host::finish_per_group(n, s1, starts, g2);
    for (double v : s1) printf("%.17g\n", v);
    for (double v : s2) printf("%.17g\n", v);
    for (double v : g2) printf("%.17g\n", v);
    return 0;
}
