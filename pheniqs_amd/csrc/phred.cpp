#include "phred.hpp"

#include <cmath>

namespace phq {

PhredScale::PhredScale() {
    uniform_base_quality = 10.0 * std::log10(4);
    phred_probability_base = std::pow(10.0, -0.1);
    // log2 of the DOUBLE the reference holds, to 106 bits.  For fl(10^-0.1) = 0x1.96b230bcdc434p-1 (what glibc's pow
    // returns) the pair is a constant computed once with 80 digit arithmetic; any other value of the base falls back
    // to the 64 bit mantissa of the host's log2l (the low word then only has 11 good bits).
    const long double log2_base = log2l(static_cast<long double>(phred_probability_base));
    log2_base_hi = static_cast<double>(log2_base);
    log2_base_lo = static_cast<double>(log2_base - static_cast<long double>(log2_base_hi));
    if(phred_probability_base == 0x1.96b230bcdc434p-1 && log2_base_hi == -0x1.542a5a12e1c5bp-2) {
        log2_base_lo = -0x1.7d4a7c345d92fp-58;
    }
    false_positive_probability[0] = 0;
    true_positive_probability[0] = 0;
    true_positive_quality[0] = 0;
    // same libm calls, same order, as phred.cpp:24-38 so the tables are bit identical on one host
    for(int q = 1; q < 128; ++q) { false_positive_probability[q] = std::pow(phred_probability_base, q); }
    for(int q = 1; q < 128; ++q) { true_positive_probability[q] = 1.0 - false_positive_probability[q]; }
    for(int q = 1; q < 128; ++q) { true_positive_quality[q] = -10.0 * std::log10(true_positive_probability[q]); }
    // per position factors of pow(base, sigma) = prod pow(base, cost): what the product form kernel multiplies
    match_probability[0] = 1;
    mismatch_probability[0] = 1;
    for(int q = 1; q < 128; ++q) {
        match_probability[q] = std::pow(phred_probability_base, true_positive_quality[q]);
        mismatch_probability[q] = std::pow(phred_probability_base, static_cast<double>(q));
    }
    uniform_probability = std::pow(phred_probability_base, uniform_base_quality);
}

const PhredScale& PhredScale::get_instance() {
    static const PhredScale instance;
    return instance;
}

double PhredScale::substitution_quality(uint8_t expected, uint8_t observed, uint8_t quality) const {
    const uint8_t q = quality & 0x7f;
    if(q == 0) { return 0.0; }
    const bool strict_e = expected == 1 || expected == 2 || expected == 4 || expected == 8;
    const bool strict_o = observed == 1 || observed == 2 || observed == 4 || observed == 8;
    if(strict_e && expected == observed) { return true_positive_quality[q]; }
    if(strict_e && strict_o) { return q; }
    return uniform_base_quality;
}

}  // namespace phq
