// PhredScale restated (reference phred.h:29-35,51-78; phred.cpp:24-72).
//
// The reference materialises substitution_lookup[quality << 8 | expected << 4 | observed], 32768
// doubles.  Every entry is one of: true_positive_quality[q] (expected == observed, both in ACGT),
// q itself (both in ACGT, different), UNIFORM_BASE_QUALITY (anything else), and the whole q == 0 row
// is 0.0 because the builder loops start at q = 1 on a zero-initialised static.  Only the 128 entry
// column and the two constants are kept; the kernels rebuild the three-way choice with a compare.
#pragma once

#include <cstdint>

namespace phq {

struct PhredScale {
    double uniform_base_quality;            // 10 log10 4
    double phred_probability_base;          // 10^-0.1
    double log2_base_hi;                    // log2(phred_probability_base) = hi + lo, for the device 2^x
    double log2_base_lo;
    double false_positive_probability[128];
    double true_positive_probability[128];
    double true_positive_quality[128];
    double match_probability[128];          // pow(base, true_positive_quality[q]): P(observed | expected) for a match
    double mismatch_probability[128];       // pow(base, q): for a mismatch between two ACGT bases
    double uniform_probability;             // pow(base, uniform_base_quality): anything involving a base outside ACGT
    static const PhredScale& get_instance();
    double substitution_quality(uint8_t expected, uint8_t observed, uint8_t quality) const;   // phred.h:58-60
private:
    PhredScale();
};

}  // namespace phq
