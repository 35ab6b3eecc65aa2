// Host-side statistics model: AccumulatingOption / AccumulatingSelector (reference selector.h:32-92)
// rebuilt from the integer counter buffer the kernels fill, then finalize and encode exactly as
//   AccumulatingOption::finalize / encode      selector.cpp:78-135
//   AccumulatingSelector::finalize / encode    selector.cpp:197-247
//   Classifier<T>::finalize / encode           classifier.h:94-124,161-177
//   Decoder<T>::finalize, PamlDecoder finalize decoder.h:77-83, pamld.h:40-48
//   Barcode::encode                            barcode.cpp:55-69
//   TranscodingDecoder::finalize / encode      transcode.cpp:180-231
#pragma once

#include <cstdint>
#include <vector>

#include "job.hpp"
#include "json.hpp"

namespace phq {

struct OptionCounters {                    // selector.h:34-41
    uint64_t count = 0;
    uint64_t pf_count = 0;
    uint64_t accumulated_distance = 0;
    uint64_t accumulated_pf_distance = 0;
    uint64_t low_conditional_confidence_count = 0;
    uint64_t low_confidence_count = 0;
    double accumulated_confidence = 0;
    double accumulated_pf_confidence = 0;
};

struct OptionFinal {                       // selector.h:43-52
    double pf_fraction = 0;
    double average_distance = 0;
    double average_confidence = 0;
    double average_pf_distance = 0;
    double average_pf_confidence = 0;
    double pooled_fraction = 0;
    double pf_pooled_fraction = 0;
    double pooled_classified_fraction = 0;
    double pf_pooled_classified_fraction = 0;
    double estimated_concentration_prior = 0;
};

struct SelectorFinal {                     // selector.h:60-79
    uint64_t count = 0;
    uint64_t pf_count = 0;
    uint64_t classified_count = 0;
    uint64_t pf_classified_count = 0;
    uint64_t accumulated_classified_distance = 0;
    uint64_t accumulated_pf_classified_distance = 0;
    uint64_t low_conditional_confidence_count = 0;
    uint64_t low_confidence_count = 0;
    double accumulated_classified_confidence = 0;
    double accumulated_pf_classified_confidence = 0;
    double pf_fraction = 0;
    double classified_fraction = 0;
    double average_classified_distance = 0;
    double average_classified_confidence = 0;
    double pf_classified_fraction = 0;
    double classified_pf_fraction = 0;
    double average_pf_classified_distance = 0;
    double average_pf_classified_confidence = 0;
    double estimated_noise_prior = 0;
};

// raw words of one decoder -> per barcode counters (bin 0 = unclassified)
std::vector<OptionCounters> decode_counters(const DecoderConfig& decoder, const uint64_t* words);

void finalize_decoder(const DecoderConfig& decoder, const std::vector<OptionCounters>& option,
                      SelectorFinal& selector, std::vector<OptionFinal>& final_option);

// the "sample" / "molecular"[i] / "cellular"[i] node of the report
Json encode_decoder(const DecoderConfig& decoder, const std::vector<OptionCounters>& option);

// whole report: incoming (if count > 0), outgoing, sample, molecular, cellular; keys sorted
Json encode_report(const JobConfig& job, const uint64_t* words, uint64_t incoming_count, uint64_t incoming_pf_count);

// Classifier<T>::adjust_prior over a job document (classifier.h:125-160, Transcode::apply_prior_adjustment
// transcode.cpp:1910-1941): every decoder's "noise" becomes its estimated noise prior and every codec record's
// "concentration" the estimated concentration prior of the barcode with the same segments; keys sorted
Json adjust_prior(const JobConfig& job, const uint64_t* words, const Json& document);

}  // namespace phq
