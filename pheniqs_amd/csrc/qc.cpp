// Quality control accumulators: see qc.hpp.
#include "qc.hpp"

#include <cmath>

#include <algorithm>
#include <cstring>

#include "error.hpp"
#include "job.hpp"

namespace phq {

void QcSegment::increment(const uint8_t* code, const uint8_t* quality, int32_t length) {
    if(length > capacity) {                                                    // multiplex.h:147-150
        cycle.resize(static_cast<size_t>(length) * QC_CODES * QC_PHRED_RANGE, 0);
        capacity = length;
    }
    if(length < shortest) { shortest = length; }
    uint64_t total = 0;
    for(int32_t i = 0; i < length; ++i) {
        const uint32_t c = code[i] & 0xfu;
        const uint32_t q = quality[i] < QC_PHRED_RANGE ? quality[i] : QC_PHRED_RANGE - 1;
        ++nucleic[0];                                                          // multiplex.h:155-156
        ++nucleic[c];
        ++cycle[(static_cast<size_t>(i) * QC_CODES + c) * QC_PHRED_RANGE + q];
        total += quality[i];
    }
    if(length > 0) {                                                           // AveragePhreadAccumulator::increment, multiplex.h:44-55
        const double value = static_cast<double>(total) / static_cast<double>(length);
        ++average_count;
        average_sum_fixed += static_cast<uint64_t>(value * 16777216.0 + 0.5);
        average_max = std::max(average_max, value);
        const size_t bin = static_cast<size_t>(value);
        ++average_distribution[bin < QC_PHRED_RANGE ? bin : QC_PHRED_RANGE - 1];
    }
}

void QcSegment::merge(const QcSegment& other) {
    if(other.capacity > capacity) {
        cycle.resize(static_cast<size_t>(other.capacity) * QC_CODES * QC_PHRED_RANGE, 0);
        capacity = other.capacity;
    }
    shortest = std::min(shortest, other.shortest);
    for(int c = 0; c < QC_CODES; ++c) { nucleic[c] += other.nucleic[c]; }
    for(size_t k = 0; k < other.cycle.size(); ++k) { cycle[k] += other.cycle[k]; }
    average_count += other.average_count;
    average_sum_fixed += other.average_sum_fixed;
    average_max = std::max(average_max, other.average_max);
    for(int q = 0; q < QC_PHRED_RANGE; ++q) { average_distribution[q] += other.average_distribution[q]; }
}

namespace {

// NucleotideAccumulator after finalize (multiplex.cpp:77-104), from its distribution alone
struct NucleotideFinal {
    uint64_t count = 0;
    uint8_t min_quality = 0, max_quality = 0, Q1 = 0, Q3 = 0, IQR = 0, LW = 0, RW = 0, median_quality = 0;
    double mean_quality = 0;
};

uint8_t quantile(const uint64_t (&distribution)[QC_PHRED_RANGE], uint64_t count, double portion) {       // multiplex.h:99-113
    uint64_t position = static_cast<uint64_t>(portion * static_cast<double>(count));
    int phred = 0;
    while(position > 0) {
        if(distribution[phred] >= position) { break; }
        position -= distribution[phred];
        ++phred;
        while(phred < QC_PHRED_RANGE - 1 && distribution[phred] == 0) { ++phred; }
        if(phred >= QC_PHRED_RANGE - 1) { break; }
    }
    return static_cast<uint8_t>(phred);
}

NucleotideFinal finalize(const uint64_t (&distribution)[QC_PHRED_RANGE]) {
    NucleotideFinal f;
    for(int q = 0; q < QC_PHRED_RANGE; ++q) { f.count += distribution[q]; }
    if(f.count > 0) {
        uint64_t sum = 0;
        for(int q = 0; q < QC_PHRED_RANGE; ++q) {
            const uint64_t value = distribution[q];
            sum += value * static_cast<uint64_t>(q);
            if(value != 0) {
                f.max_quality = static_cast<uint8_t>(q);
                if(f.min_quality == 0) { f.min_quality = static_cast<uint8_t>(q); }
            }
        }
        f.mean_quality = static_cast<double>(sum) / static_cast<double>(f.count);
        f.median_quality = quantile(distribution, f.count, 0.5);
        f.Q1 = quantile(distribution, f.count, 0.25);
        f.Q3 = quantile(distribution, f.count, 0.75);
        f.IQR = static_cast<uint8_t>(f.Q3 - f.Q1);
        double W = f.Q1 - f.IQR * 1.5;
        f.LW = (W < f.min_quality) ? f.min_quality : static_cast<uint8_t>(W);
        W = f.Q3 + f.IQR * 1.5;
        f.RW = (W > f.max_quality) ? f.max_quality : static_cast<uint8_t>(W);
    }
    return f;
}

}  // namespace

Json QcSegment::encode() const {
    Json container = Json::object();
    container.set("min sequence length", Json::integer(shortest == 0x7fffffff ? 0 : shortest));       // multiplex.cpp:170-172
    container.set("max sequence length", Json::integer(capacity));
    Json by_cycle = Json::object();
    Json by_nucleotide = Json::array();
    for(int n = 0; n < QC_CODES; ++n) {
        if(nucleic[n] == 0) { continue; }
        const char* const names[10] = {"cycle count", "cycle quality first quartile", "cycle quality third quartile", "cycle quality interquartile range",
                                       "cycle quality left whisker", "cycle quality right whisker", "cycle quality min", "cycle quality max",
                                       "cycle quality mean", "cycle quality median"};
        Json column[10];
        for(auto& c : column) { c = Json::array(); }
        for(int32_t c = 0; c < capacity; ++c) {
            uint64_t distribution[QC_PHRED_RANGE];
            if(n == 0) {
                // CycleAccumulator::finalize, multiplex.cpp:134-143: code 0 holds every nucleotide of the cycle
                for(int q = 0; q < QC_PHRED_RANGE; ++q) {
                    distribution[q] = 0;
                    for(int code = 0; code < QC_CODES; ++code) { distribution[q] += cycle[(static_cast<size_t>(c) * QC_CODES + static_cast<size_t>(code)) * QC_PHRED_RANGE + static_cast<size_t>(q)]; }
                }
            } else {
                std::memcpy(distribution, &cycle[(static_cast<size_t>(c) * QC_CODES + static_cast<size_t>(n)) * QC_PHRED_RANGE], sizeof(distribution));
            }
            const NucleotideFinal f = finalize(distribution);
            column[0].push(Json::uinteger(f.count));
            column[1].push(Json::uinteger(f.Q1)); column[2].push(Json::uinteger(f.Q3)); column[3].push(Json::uinteger(f.IQR));
            column[4].push(Json::uinteger(f.LW)); column[5].push(Json::uinteger(f.RW));
            column[6].push(Json::uinteger(f.min_quality)); column[7].push(Json::uinteger(f.max_quality));
            column[8].push(Json::real(f.mean_quality)); column[9].push(Json::uinteger(f.median_quality));
        }
        Json distribution_report = Json::object();
        for(int k = 0; k < 10; ++k) { distribution_report.set(names[k], column[k]); }
        if(n > 0) {
            Json nucleotide_report = Json::object();
            nucleotide_report.set("nucleotide count", Json::uinteger(nucleic[n]));
            nucleotide_report.set("nucleotide", Json::string(std::string(1, bam_to_ascii(static_cast<uint8_t>(n)))));
            nucleotide_report.set("cycle quality distribution", distribution_report);
            by_nucleotide.push(nucleotide_report);
        } else {
            by_cycle.set("cycle quality distribution", distribution_report);
        }
    }
    container.set("quality control by nucleotide", by_nucleotide);
    container.set("quality control by cycle", by_cycle);
    Json average = Json::object();
    average.set("average phred score min", Json::real(0.0));                 // AveragePhreadAccumulator: min_value starts at 0 (multiplex.cpp:27)
    average.set("average phred score max", Json::real(average_max));
    average.set("average phred score mean", Json::real(average_count > 0 ? static_cast<double>(average_sum_fixed) / 16777216.0 / static_cast<double>(average_count) : 0.0));
    Json distribution = Json::array();
    for(int q = 0; q < QC_PHRED_RANGE; ++q) { distribution.push(Json::uinteger(average_distribution[q])); }
    average.set("average phred score distribution", distribution);
    container.set("average phred score report", average);
    return container;
}

void QcState::merge(const QcState& other) {
    if(other.empty()) { return; }
    if(empty()) { *this = other; return; }
    if(other.channels != channels || other.segments != segments) { throw InternalError("quality control accumulators of different shapes"); }
    for(size_t k = 0; k < cell.size(); ++k) { cell[k].merge(other.cell[k]); }
}

Json QcState::encode() const {
    Json channel_array = Json::array();                                       // Multiplexer::encode, multiplex.cpp:419-431
    for(int32_t c = 0; c < channels; ++c) {
        Json by_segment = Json::array();
        for(int32_t s = 0; s < segments; ++s) { by_segment.push(cell[static_cast<size_t>(c) * static_cast<size_t>(segments) + static_cast<size_t>(s)].encode()); }
        Json channel_report = Json::object();
        channel_report.set("quality control by segment", by_segment);         // Channel::encode, multiplex.cpp:349-368
        channel_array.push(channel_report);
    }
    return channel_array;
}

std::vector<uint64_t> QcState::to_words() const {
    std::vector<uint64_t> words = {static_cast<uint64_t>(channels), static_cast<uint64_t>(segments)};
    for(const auto& cell_of : cell) {
        words.push_back(static_cast<uint64_t>(cell_of.capacity));
        words.push_back(static_cast<uint64_t>(static_cast<uint32_t>(cell_of.shortest)));
        words.insert(words.end(), cell_of.nucleic, cell_of.nucleic + QC_CODES);
        words.push_back(cell_of.average_count);
        words.push_back(cell_of.average_sum_fixed);
        uint64_t bits;
        std::memcpy(&bits, &cell_of.average_max, sizeof(bits));
        words.push_back(bits);
        words.insert(words.end(), cell_of.average_distribution, cell_of.average_distribution + QC_PHRED_RANGE);
        words.insert(words.end(), cell_of.cycle.begin(), cell_of.cycle.end());
    }
    return words;
}

// An image gathered from another rank or process may be cut short or damaged: nothing is allocated on the word of its
// header alone — every cell's room is checked against what the image still holds before it is resized — and an average that
// is not a finite number is refused (it would poison the maximum of a merge).
QcState QcState::from_words(const uint64_t* words, size_t n) {
    QcState state;
    if(n < 2) { throw InternalError("quality control image too short"); }
    size_t at = 0;
    auto take = [&]() { if(at >= n) { throw InternalError("quality control image too short"); } return words[at++]; };
    const uint64_t channels = take(), segments = take();
    if(channels > (1u << 20) || segments > 64) { throw InternalError("quality control image out of range"); }
    constexpr size_t CELL_FIXED = 2 + QC_CODES + 3 + QC_PHRED_RANGE;               // what a cell holds besides its cycles
    const uint64_t cells = channels * segments;
    if(cells > (n - at) / CELL_FIXED) { throw InternalError("quality control image too short for its " + std::to_string(cells) + " cells"); }
    state.shape(static_cast<int32_t>(channels), static_cast<int32_t>(segments));
    for(auto& cell_of : state.cell) {
        const uint64_t capacity = take();
        if(capacity > (1u << 20)) { throw InternalError("quality control image out of range"); }
        const size_t cycle_words = static_cast<size_t>(capacity) * QC_CODES * QC_PHRED_RANGE;
        if(CELL_FIXED - 1 + cycle_words > n - at) { throw InternalError("quality control image too short"); }      // before anything is resized
        cell_of.capacity = static_cast<int32_t>(capacity);
        cell_of.shortest = static_cast<int32_t>(static_cast<uint32_t>(take()));
        for(int c = 0; c < QC_CODES; ++c) { cell_of.nucleic[c] = take(); }
        cell_of.average_count = take();
        cell_of.average_sum_fixed = take();
        const uint64_t bits = take();
        std::memcpy(&cell_of.average_max, &bits, sizeof(bits));
        if(!std::isfinite(cell_of.average_max)) { throw InternalError("quality control image: an average that is not a number"); }
        for(int q = 0; q < QC_PHRED_RANGE; ++q) { cell_of.average_distribution[q] = take(); }
        cell_of.cycle.resize(cycle_words);
        for(auto& w : cell_of.cycle) { w = take(); }
    }
    return state;
}

}  // namespace phq
