// Per channel quality control accumulators of the output reads (the last optional piece):
//
//   reference multiplex.h:28-181    AveragePhreadAccumulator, NucleotideAccumulator, CycleAccumulator, SegmentAccumulator,
//                                   ReadAccumulator: what Channel::push counts when `enable quality control` is set
//   reference multiplex.cpp:26-262  finalize (quantiles, whiskers), operator+= (the merge), encode_value (the report)
//   reference multiplex.cpp:405-431 Multiplexer::collect / finalize / encode: the report's "multiplex" array
//
// They count the OUTPUT read (the template's segments, after Multiplexer::push has picked the channel), which only the
// host side of the transcoder holds as text; so they are filled by its record threads — one private copy per thread, as
// the reference keeps one per TranscodingThread — and merged at the end of a run (multiplex.cpp:405-411).  Every count
// is an integer; the one sum of doubles the reference keeps (the reads' mean qualities) is held as a 2^-24 fixed point
// integer, so the merge is exact and independent of the order of the threads.
#pragma once

#include <cstdint>
#include <vector>

#include "json.hpp"

namespace phq {

constexpr int QC_PHRED_RANGE = 42;            // phred.h:32 EFFECTIVE_PHRED_RANGE
constexpr int QC_CODES = 16;                  // iupac.h:57 IUPAC_CODE_SIZE

struct QcSegment {                            // SegmentAccumulator, multiplex.h:126-164
    int32_t capacity = 0;                     // the longest segment seen = cycles held
    int32_t shortest = 0x7fffffff;
    uint64_t nucleic[QC_CODES] = {0};         // [0] = every base
    uint64_t average_count = 0;               // AveragePhreadAccumulator, multiplex.h:28-59
    uint64_t average_sum_fixed = 0;           // sum of the reads' mean quality, 2^-24 fixed point
    double average_max = 0;                   // (the reference's min starts at 0 and so stays 0)
    uint64_t average_distribution[QC_PHRED_RANGE] = {0};
    std::vector<uint64_t> cycle;              // [cycle][code][phred]

    // one output segment of one read; qualities past the range the reference's vectors hold (it indexes them unchecked)
    // go to the last bin; an empty segment has no mean quality and leaves the average alone
    void increment(const uint8_t* code, const uint8_t* quality, int32_t length);
    void merge(const QcSegment& other);       // operator+=, multiplex.cpp:178-192
    Json encode() const;                      // finalize + encode_value, multiplex.cpp:77-104,169-176,193-262
};

struct QcState {
    int32_t channels = 0;                     // undetermined + the multiplexing decoder's barcodes
    int32_t segments = 0;                     // output segment cardinality
    std::vector<QcSegment> cell;              // [channel][segment]
    void shape(int32_t n_channel, int32_t n_segment) { channels = n_channel; segments = n_segment; cell.assign(static_cast<size_t>(n_channel) * static_cast<size_t>(n_segment), QcSegment()); }
    bool empty() const { return cell.empty(); }
    QcSegment& at(int32_t channel, int32_t segment) { return cell[static_cast<size_t>(channel) * static_cast<size_t>(segments) + static_cast<size_t>(segment)]; }
    void merge(const QcState& other);
    Json encode() const;                      // Multiplexer::encode: the "multiplex" array
    // a flat image for a merge across processes: {channels, segments} then per cell {capacity, shortest, nucleic[16],
    // average count, sum, max as its bit pattern, distribution[42], capacity x 16 x 42 cycle words}
    std::vector<uint64_t> to_words() const;
    static QcState from_words(const uint64_t* words, size_t n);
};

}  // namespace phq
