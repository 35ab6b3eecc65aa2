#include "stats.hpp"

#include "error.hpp"

#include <cmath>

namespace phq {

namespace {

double confidence_from_limbs(uint64_t low, uint64_t high) {
    // value = (high * 2^26 + low) / 2^52, assembled in extended precision so that the only rounding is the
    // final conversion to double
    const long double scaled = static_cast<long double>(high) * static_cast<long double>(1ull << CONFIDENCE_LIMB_BITS) + static_cast<long double>(low);
    return static_cast<double>(std::ldexp(scaled, -CONFIDENCE_FRACTION_BITS));
}

}  // namespace

std::vector<OptionCounters> decode_counters(const DecoderConfig& decoder, const uint64_t* words) {
    std::vector<OptionCounters> out(decoder.codec.size() + 1);
    const uint64_t* base = words + decoder.stats_offset;
    for(size_t b = 0; b < out.size(); ++b) {
        const uint64_t* fail = base + b * STAT_WORDS_PER_BIN;
        const uint64_t* pass = fail + STAT_FIELDS;
        OptionCounters& o = out[b];
        o.count = fail[FIELD_COUNT] + pass[FIELD_COUNT];
        o.pf_count = pass[FIELD_COUNT];
        o.accumulated_distance = fail[FIELD_DISTANCE] + pass[FIELD_DISTANCE];
        o.accumulated_pf_distance = pass[FIELD_DISTANCE];
        o.low_conditional_confidence_count = fail[FIELD_LOW_CONDITIONAL_CONFIDENCE] + pass[FIELD_LOW_CONDITIONAL_CONFIDENCE];
        o.low_confidence_count = fail[FIELD_LOW_CONFIDENCE] + pass[FIELD_LOW_CONFIDENCE];
        o.accumulated_confidence = confidence_from_limbs(fail[FIELD_CONFIDENCE_LOW] + pass[FIELD_CONFIDENCE_LOW],
                                                         fail[FIELD_CONFIDENCE_HIGH] + pass[FIELD_CONFIDENCE_HIGH]);
        o.accumulated_pf_confidence = confidence_from_limbs(pass[FIELD_CONFIDENCE_LOW], pass[FIELD_CONFIDENCE_HIGH]);
    }
    return out;
}

void finalize_decoder(const DecoderConfig& decoder, const std::vector<OptionCounters>& option,
                      SelectorFinal& sel, std::vector<OptionFinal>& final_option) {
    (void)decoder;
    sel = SelectorFinal();
    final_option.assign(option.size(), OptionFinal());
    const size_t n = option.size();
    // pamld.h:40-48 (all zero for the other algorithms)
    for(size_t b = 1; b < n; ++b) {
        sel.accumulated_classified_confidence += option[b].accumulated_confidence;
        sel.accumulated_pf_classified_confidence += option[b].accumulated_pf_confidence;
        sel.low_conditional_confidence_count += option[b].low_conditional_confidence_count;
        sel.low_confidence_count += option[b].low_confidence_count;
    }
    // decoder.h:77-83
    for(size_t b = 1; b < n; ++b) {
        sel.accumulated_classified_distance += option[b].accumulated_distance;
        sel.accumulated_pf_classified_distance += option[b].accumulated_pf_distance;
    }
    // classifier.h:94-124
    for(size_t b = 1; b < n; ++b) {
        sel.classified_count += option[b].count;
        sel.pf_classified_count += option[b].pf_count;
    }
    sel.count = sel.classified_count + option[0].count;
    sel.pf_count = sel.pf_classified_count + option[0].pf_count;

    double estimated_noise_count = static_cast<double>(sel.low_conditional_confidence_count);
    const double confident_noise_ratio = estimated_noise_count / (estimated_noise_count + sel.pf_classified_count);
    if(sel.low_confidence_count > 0) {
        estimated_noise_count += static_cast<double>(sel.low_confidence_count) * confident_noise_ratio;
    }
    sel.estimated_noise_prior = estimated_noise_count / static_cast<double>(sel.count);

    const double estimated_not_noise_prior = 1.0 - sel.estimated_noise_prior;
    for(size_t b = 0; b < n; ++b) {
        const OptionCounters& x = option[b];
        OptionFinal& f = final_option[b];
        if(x.count > 0) {                                                   // selector.cpp:78-101
            f.average_distance = x.accumulated_distance / static_cast<double>(x.count);
            f.average_confidence = x.accumulated_confidence / static_cast<double>(x.count);
            if(sel.count > 0) { f.pooled_fraction = static_cast<double>(x.count) / static_cast<double>(sel.count); }
            if(sel.classified_count > 0) { f.pooled_classified_fraction = static_cast<double>(x.count) / static_cast<double>(sel.classified_count); }
        }
        if(x.pf_count > 0) {
            f.pf_fraction = static_cast<double>(x.pf_count) / static_cast<double>(x.count);
            f.average_pf_distance = x.accumulated_pf_distance / static_cast<double>(x.pf_count);
            f.average_pf_confidence = x.accumulated_pf_confidence / static_cast<double>(x.pf_count);
            if(sel.pf_count > 0) { f.pf_pooled_fraction = static_cast<double>(x.pf_count) / static_cast<double>(sel.pf_count); }
            if(sel.pf_classified_count > 0) { f.pf_pooled_classified_fraction = static_cast<double>(x.pf_count) / static_cast<double>(sel.pf_classified_count); }
        }
        if(b > 0) { f.estimated_concentration_prior = estimated_not_noise_prior * f.pf_pooled_classified_fraction; }
    }
    if(sel.count > 0) {                                                     // selector.cpp:197-214
        sel.pf_fraction = static_cast<double>(sel.pf_count) / static_cast<double>(sel.count);
        sel.classified_fraction = static_cast<double>(sel.classified_count) / static_cast<double>(sel.count);
    }
    if(sel.pf_count > 0) { sel.pf_classified_fraction = static_cast<double>(sel.pf_classified_count) / static_cast<double>(sel.pf_count); }
    if(sel.classified_count > 0) {
        sel.average_classified_distance = sel.accumulated_classified_distance / static_cast<double>(sel.classified_count);
        sel.average_classified_confidence = sel.accumulated_classified_confidence / static_cast<double>(sel.classified_count);
        sel.classified_pf_fraction = static_cast<double>(sel.pf_classified_count) / static_cast<double>(sel.classified_count);
    }
    if(sel.pf_classified_count > 0) {
        sel.average_pf_classified_distance = sel.accumulated_pf_classified_distance / static_cast<double>(sel.pf_classified_count);
        sel.average_pf_classified_confidence = sel.accumulated_pf_classified_confidence / static_cast<double>(sel.pf_classified_count);
    }
}

namespace {

Json encode_option(const OptionCounters& o, const OptionFinal& f, const BarcodeConfig* barcode) {
    Json e = Json::object();                                                // selector.cpp:102-135
    e.set("count", Json::uinteger(o.count));
    if(f.average_distance > 0) { e.set("average distance", Json::real(f.average_distance)); }
    if(f.average_confidence > 0) { e.set("average confidence", Json::real(f.average_confidence)); }
    if(o.low_conditional_confidence_count > 0) { e.set("low conditional confidence count", Json::uinteger(o.low_conditional_confidence_count)); }
    if(o.low_confidence_count > 0) { e.set("low confidence count", Json::uinteger(o.low_confidence_count)); }
    e.set("pooled fraction", Json::real(f.pooled_fraction));
    if(f.pooled_classified_fraction > 0) { e.set("pooled classified fraction", Json::real(f.pooled_classified_fraction)); }
    e.set("pf count", Json::uinteger(o.pf_count));
    if(f.average_pf_distance > 0) { e.set("average pf distance", Json::real(f.average_pf_distance)); }
    if(f.average_pf_confidence > 0) { e.set("average pf confidence", Json::real(f.average_pf_confidence)); }
    e.set("pf fraction", Json::real(f.pf_fraction));
    e.set("pf pooled fraction", Json::real(f.pf_pooled_fraction));
    if(f.pf_pooled_classified_fraction > 0) { e.set("pf pooled classified fraction", Json::real(f.pf_pooled_classified_fraction)); }
    if(f.estimated_concentration_prior > 0) { e.set("estimated concentration", Json::real(f.estimated_concentration_prior)); }
    // barcode.cpp:55-69
    e.set("index", Json::integer(barcode ? barcode->index : 0));
    if(barcode != nullptr && barcode->index > 0) {
        e.set("concentration", Json::real(barcode->concentration));
        Json segments = Json::array();
        for(const auto& s : barcode->segment) { segments.push(Json::string(s)); }
        e.set("barcode", segments);
    }
    return e;
}

}  // namespace

Json encode_decoder(const DecoderConfig& decoder, const std::vector<OptionCounters>& option) {
    SelectorFinal sel;
    std::vector<OptionFinal> fin;
    finalize_decoder(decoder, option, sel, fin);
    Json out = Json::object();                                              // selector.cpp:215-247
    out.set("index", Json::integer(decoder.index));
    out.set("count", Json::uinteger(sel.count));
    out.set("pf count", Json::uinteger(sel.pf_count));
    out.set("classified count", Json::uinteger(sel.classified_count));
    if(sel.low_conditional_confidence_count > 0) { out.set("low conditional confidence count", Json::uinteger(sel.low_conditional_confidence_count)); }
    if(sel.low_confidence_count > 0) { out.set("low confidence count", Json::uinteger(sel.low_confidence_count)); }
    out.set("pf classified count", Json::uinteger(sel.pf_classified_count));
    out.set("pf fraction", Json::real(sel.pf_fraction));
    out.set("classified fraction", Json::real(sel.classified_fraction));
    if(sel.average_classified_distance > 0) { out.set("average classified distance", Json::real(sel.average_classified_distance)); }
    if(sel.average_classified_confidence > 0) { out.set("average classified confidence", Json::real(sel.average_classified_confidence)); }
    out.set("pf classified fraction", Json::real(sel.pf_classified_fraction));
    out.set("classified pf fraction", Json::real(sel.classified_pf_fraction));
    if(sel.average_pf_classified_distance > 0) { out.set("average pf classified distance", Json::real(sel.average_pf_classified_distance)); }
    if(sel.average_pf_classified_confidence > 0) { out.set("average pf classified confidence", Json::real(sel.average_pf_classified_confidence)); }
    if(sel.estimated_noise_prior > 0) { out.set("estimated noise", Json::real(sel.estimated_noise_prior)); }
    // classifier.h:161-177
    out.set("unclassified", encode_option(option[0], fin[0], &decoder.undetermined));
    if(!decoder.codec.empty()) {
        Json classified = Json::array();
        for(size_t b = 0; b < decoder.codec.size(); ++b) {
            classified.push(encode_option(option[b + 1], fin[b + 1], &decoder.codec[b]));
        }
        out.set("classified", classified);
    }
    return out;
}

Json encode_report(const JobConfig& job, const uint64_t* words, uint64_t incoming_count, uint64_t incoming_pf_count) {
    Json report = Json::object();
    if(incoming_count > 0) {                                                // transcode.cpp:1821-1828
        Json e = Json::object();
        e.set("count", Json::uinteger(incoming_count));
        e.set("pf count", Json::uinteger(incoming_pf_count));
        e.set("pf fraction", Json::real(static_cast<double>(incoming_pf_count) / static_cast<double>(incoming_count)));
        report.set("incoming", e);
    }
    const uint64_t count = words[0] + words[1];
    const uint64_t pf_count = words[1];
    if(count > 0) {                                                         // transcode.cpp:196-204
        Json e = Json::object();
        e.set("count", Json::uinteger(count));
        e.set("pf count", Json::uinteger(pf_count));
        e.set("pf fraction", Json::real(static_cast<double>(pf_count) / static_cast<double>(count)));
        report.set("outgoing", e);
    }
    Json molecular = Json::array();
    Json cellular = Json::array();
    for(const auto& d : job.decoder) {
        Json node = encode_decoder(d, decode_counters(d, words));
        switch(d.kind) {
            case Kind::SAMPLE: report.set("sample", node); break;
            case Kind::MOLECULAR: molecular.push(node); break;
            case Kind::CELLULAR: cellular.push(node); break;
        }
    }
    if(!molecular.elements().empty()) { report.set("molecular", molecular); }
    if(!cellular.elements().empty()) { report.set("cellular", cellular); }
    report.sort_keys();                                                     // transcode.cpp:1862 sort_json_value
    return report;
}

namespace {

void adjust_decoder(const DecoderConfig& decoder, const uint64_t* words, Json& node) {
    if(!node.is_object()) { return; }
    const std::vector<OptionCounters> option = decode_counters(decoder, words);
    SelectorFinal selector;
    std::vector<OptionFinal> final_option;
    finalize_decoder(decoder, option, selector, final_option);
    node.set("noise", Json::real(selector.estimated_noise_prior));
    Json* codec = node.find("codec");
    if(codec == nullptr || !codec->is_object()) { return; }
    for(auto& member : codec->members()) {
        const Json* segments = member.second.find("barcode");
        if(segments == nullptr || !segments->is_array()) { continue; }
        for(size_t b = 0; b < decoder.codec.size(); ++b) {
            const auto& mine = decoder.codec[b].segment;
            bool same = mine.size() == segments->elements().size();
            for(size_t k = 0; same && k < mine.size(); ++k) {
                same = segments->elements()[k].is_string() && segments->elements()[k].as_string() == mine[k];
            }
            if(same) { member.second.set("concentration", Json::real(final_option[b + 1].estimated_concentration_prior)); break; }
        }
    }
}

}  // namespace

Json adjust_prior(const JobConfig& job, const uint64_t* words, const Json& document) {
    if(!document.is_object()) { throw ConfigurationError("job document must be a dictionary"); }
    Json adjusted = document;
    size_t k = 0;
    for(const char* topic : {"sample", "molecular", "cellular"}) {              // the order of job.decoder (transcode.h:51-60)
        Json* node = adjusted.find(topic);
        if(node == nullptr || node->is_null()) { continue; }
        if(node->is_object()) {
            if(k >= job.decoder.size()) { throw ConfigurationError("the document has more decoders than the decoder set"); }
            adjust_decoder(job.decoder[k], words, *node);
            ++k;
        } else if(node->is_array()) {
            for(auto& element : node->elements()) {
                if(k >= job.decoder.size()) { throw ConfigurationError("the document has more decoders than the decoder set"); }
                adjust_decoder(job.decoder[k], words, element);
                ++k;
            }
        }
    }
    if(k != job.decoder.size()) { throw ConfigurationError("the document does not hold the decoders of the decoder set"); }
    adjusted.sort_keys();
    return adjusted;
}

}  // namespace phq
