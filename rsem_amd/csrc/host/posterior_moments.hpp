// posterior_moments.hpp -- what is left of Gibbs.cpp's release() (Gibbs.cpp:389-423) once the per-chain accumulators have been
// summed on the device / over the GPUs: sample means and unbiased sample variances from sums and sums of squares.
//
//   mean = S1 / n,   var = (S2 - n * mean^2) / (n - 1), floored at 0 (cancellation can leave -1e-17)
//
// The count of a group (gene, or transcript of an allele-specific reference) is the sum of its members' counts, so its mean is
// the sum of their means; its S2 was accumulated per kept sample by the sampler.  Expressions are evaluated in the
// reference's order (n * mean * mean, left to right), so the printed values are the same digits.
#pragma once
#include <cstddef>
#include <vector>

namespace rsem_host {

struct Moments {
    double n;  // kept samples over all chains
    explicit Moments(long n_samples) : n((double)n_samples) {}
    double mean(double s1) const { return s1 / n; }
    double variance(double s2, double mean_value) const {
        const double v = (s2 - n * mean_value * mean_value) / (n - 1.0);
        return v < 0.0 ? 0.0 : v;
    }
};

// in: sums / sums of squares per transcript; out: means / variances in place
inline void finish_per_transcript(long n_samples, std::vector<double>& s1_to_mean, std::vector<double>& s2_to_var) {
    const Moments mo(n_samples);
    for (size_t i = 0; i < s1_to_mean.size(); i++) {
        s1_to_mean[i] = mo.mean(s1_to_mean[i]);
        s2_to_var[i] = mo.variance(s2_to_var[i], s1_to_mean[i]);
    }
}

inline void finish_means(long n_samples, std::vector<double>& s1_to_mean) {
    const Moments mo(n_samples);
    for (double& v : s1_to_mean) v = mo.mean(v);
}

// group g = members [starts[g], starts[g+1]) of mean_counts; in: sums of squares of the group counts, out: variances
inline void finish_per_group(long n_samples, const std::vector<double>& mean_counts, const std::vector<int>& starts,
                             std::vector<double>& s2_to_var) {
    const Moments mo(n_samples);
    for (size_t g = 0; g + 1 < starts.size() && g < s2_to_var.size(); g++) {
        double group_mean = 0.0;
        for (int j = starts[g]; j < starts[g + 1]; j++) group_mean += mean_counts[j];
        s2_to_var[g] = mo.variance(s2_to_var[g], group_mean);
    }
}

}  // namespace rsem_host
