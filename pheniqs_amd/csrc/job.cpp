#include "job.hpp"

#include <cmath>
#include <limits>

#include "error.hpp"

namespace phq {

uint8_t ascii_to_bam(uint8_t c) {
    switch(c) {
        case '0': return 1;  case '1': return 2;  case '2': return 4;  case '3': return 8;
        case '=': return 0;
        case 'A': case 'a': return 1;   case 'C': case 'c': return 2;
        case 'G': case 'g': return 4;   case 'T': case 't': return 8;
        case 'M': case 'm': return 3;   case 'R': case 'r': return 5;
        case 'S': case 's': return 6;   case 'V': case 'v': return 7;
        case 'W': case 'w': return 9;   case 'Y': case 'y': return 10;
        case 'H': case 'h': return 11;  case 'K': case 'k': return 12;
        case 'D': case 'd': return 13;  case 'B': case 'b': return 14;
        default: return 15;
    }
}
uint8_t reverse_complement_bam(uint8_t c) {
    // swap A<->T (bits 0,3) and C<->G (bits 1,2): bit reversal of the 4 bit code
    c &= 0xf;
    return static_cast<uint8_t>(((c & 1) << 3) | ((c & 2) << 1) | ((c & 4) >> 1) | ((c & 8) >> 3));
}
char bam_to_ascii(uint8_t c) {
    static const char table[] = "=ACMGRSVTWYHKDBN";
    return table[c & 0xf];
}

int32_t TransformConfig::absolute_end(int32_t length) const {
    if(end_terminated) {
        if(end < 0) { int32_t v = length + end; return v < 0 ? 0 : v; }
        return end > length ? length : end;
    }
    return length;
}
int32_t TransformConfig::absolute_start(int32_t length) const {
    if(start < 0) { int32_t v = length + start; return v < 0 ? 0 : v; }
    return start > length ? 0 : start;
}
int32_t TransformConfig::length() const {
    bool constant = end_terminated ? ((start >= 0 && end >= 0) || (start < 0 && end < 0)) : start < 0;
    if(!constant) { return -1; }
    if(end_terminated) { return start >= end ? 0 : end - start; }
    return -start;
}

namespace {

const Json& require(const Json& node, const char* key) {
    const Json* value = node.find(key);
    if(value == nullptr || value->is_null()) { throw ConfigurationError(std::string("no element ") + key + " found"); }
    return *value;
}
int32_t get_int(const Json& node, const char* key) {
    const Json& v = require(node, key);
    if(!v.is_number()) { throw ConfigurationError(std::string(key) + " element must be a number"); }
    return static_cast<int32_t>(v.as_int());
}
int32_t get_int(const Json& node, const char* key, int32_t fallback) {
    const Json* v = node.find(key);
    if(v == nullptr || v->is_null()) { return fallback; }
    if(!v->is_number()) { throw ConfigurationError(std::string(key) + " element must be a number"); }
    return static_cast<int32_t>(v->as_int());
}
double get_double(const Json& node, const char* key, double fallback) {
    const Json* v = node.find(key);
    if(v == nullptr || v->is_null()) { return fallback; }
    if(!v->is_number()) { throw ConfigurationError(std::string(key) + " element must be a number"); }
    return v->as_double();
}
bool get_bool(const Json& node, const char* key, bool fallback) {
    const Json* v = node.find(key);
    if(v == nullptr || v->is_null()) { return fallback; }
    if(!v->is_bool()) { throw ConfigurationError(std::string(key) + " element must be a boolean"); }
    return v->as_bool();
}
std::vector<std::string> get_string_array(const Json& node, const char* key) {
    std::vector<std::string> out;
    const Json* found = node.find(key);
    if(found == nullptr || found->is_null()) { return out; }
    if(!found->is_array()) { throw ConfigurationError(std::string(key) + " element must be an array"); }
    for(const auto& e : found->elements()) {
        if(!e.is_string()) { throw ConfigurationError(std::string(key) + " element must be an array of strings"); }
        out.push_back(e.as_string());
    }
    return out;
}

std::vector<int32_t> get_int_array(const Json& node, const char* key) {
    std::vector<int32_t> out;
    const Json* v = node.find(key);
    if(v == nullptr || v->is_null()) { return out; }
    if(!v->is_array()) { throw ConfigurationError(std::string(key) + " element must be an array"); }
    for(const auto& e : v->elements()) {
        if(!e.is_number()) { throw ConfigurationError(std::string(key) + " element must hold numbers"); }
        out.push_back(static_cast<int32_t>(e.as_int()));
    }
    return out;
}

Algorithm parse_algorithm(const Json& node, Kind kind) {
    const Json* v = node.find("algorithm");
    if(v == nullptr || !v->is_string()) { throw ConfigurationError("decoder must declare an algorithm"); }
    const std::string& name = v->as_string();
    if(name == "pamld") { return Algorithm::PAMLD; }
    if(name == "mdd") { return Algorithm::MDD; }
    if(name == "passthrough") { return Algorithm::PASSTHROUGH; }
    if(name == "naive") {
        if(kind == Kind::MOLECULAR) { return Algorithm::NAIVE; }    // transcode.cpp:100-103
    }
    static const char* topic[] = {"sample", "molecular", "cellular"};
    throw ConfigurationError(std::string("unsupported ") + topic[static_cast<int>(kind)] + " decoder algorithm " + name);
}

// token syntax `segment:start:end`  (reference transform.cpp:31,118-174)
struct TokenConfig { int32_t input_segment_index; int32_t start; int32_t end; bool end_terminated; };

TokenConfig parse_token(const std::string& pattern) {
    size_t first = pattern.find(':');
    size_t second = first == std::string::npos ? std::string::npos : pattern.find(':', first + 1);
    if(first == std::string::npos || second == std::string::npos || pattern.find(':', second + 1) != std::string::npos) {
        throw ConfigurationError("illegal token syntax " + pattern);
    }
    auto integer = [&](const std::string& text, bool allow_sign) -> int32_t {
        if(text.empty()) { throw ConfigurationError("illegal token syntax " + pattern); }
        size_t i = 0;
        if(text[0] == '-') { if(!allow_sign || text.size() == 1) { throw ConfigurationError("illegal token syntax " + pattern); } i = 1; }
        for(; i < text.size(); ++i) { if(text[i] < '0' || text[i] > '9') { throw ConfigurationError("illegal token syntax " + pattern); } }
        return static_cast<int32_t>(std::stol(text));
    };
    TokenConfig token{};
    std::string head = pattern.substr(0, first);
    std::string start = pattern.substr(first + 1, second - first - 1);
    std::string end = pattern.substr(second + 1);
    if(head == "s") { token.input_segment_index = -1; }
    else if(head == "c") { token.input_segment_index = -2; }
    else if(head == "m") { token.input_segment_index = -3; }
    else { token.input_segment_index = integer(head, false); }
    token.start = start.empty() ? 0 : integer(start, true);
    token.end_terminated = !end.empty();
    token.end = end.empty() ? 0 : integer(end, true);
    return token;
}

// knit syntax `[~]token[:[~]token...]` per observation segment (reference transform.cpp:252-331)
std::vector<TransformConfig> parse_rule(const Json& decoder, int32_t& output_segment_cardinality, const char* key = "transform") {
    const Json& rule = require(decoder, key);
    if(!rule.is_object()) { throw ConfigurationError("element transform must be a dictionary"); }
    const Json& token_node = require(rule, "token");
    if(!token_node.is_array()) { throw ConfigurationError("token element must be an array"); }
    std::vector<TokenConfig> token;
    for(const auto& e : token_node.elements()) {
        if(!e.is_string()) { throw ConfigurationError("token element must be a string"); }
        token.push_back(parse_token(e.as_string()));
    }
    std::vector<std::string> knit;
    const Json* knit_node = rule.find("knit");
    if(knit_node == nullptr || knit_node->is_null() || (knit_node->is_array() && knit_node->elements().empty())) {
        for(size_t i = 0; i < token.size(); ++i) { knit.push_back(std::to_string(i)); }   // transcode.cpp:749-757
    } else {
        if(!knit_node->is_array()) { throw ConfigurationError("rule observation element must be an array"); }
        for(const auto& e : knit_node->elements()) {
            if(!e.is_string()) { throw ConfigurationError("transform element must be a string"); }
            knit.push_back(e.as_string());
        }
    }
    std::vector<TransformConfig> out;
    output_segment_cardinality = 0;
    for(const auto& pattern : knit) {
        bool reverse_complement = false;
        int64_t reference = -1;
        auto emit = [&]() {
            if(reference < 0) { throw ConfigurationError("transform must explicitly specify a token reference"); }
            if(reference >= static_cast<int64_t>(token.size())) {
                throw ConfigurationError("invalid token reference " + std::to_string(reference) + " in transform");
            }
            const TokenConfig& t = token[static_cast<size_t>(reference)];
            out.push_back(TransformConfig{t.input_segment_index, t.start, t.end, t.end_terminated, output_segment_cardinality, reverse_complement});
            reference = -1;
            reverse_complement = false;
        };
        for(char c : pattern) {
            if(c == ':') { emit(); }
            else if(c == '~') {
                if(reference >= 0) { throw ConfigurationError(std::string("illegal right hand side operator in transform ") + c); }
                reverse_complement = true;
            } else if(c >= '0' && c <= '9') { reference = (reference < 0 ? 0 : reference * 10) + (c - '0'); }
            else { throw ConfigurationError(std::string("illegal character in transform ") + c); }
        }
        emit();
        ++output_segment_cardinality;
    }
    return out;
}

BarcodeConfig parse_barcode(const std::string& key, const Json& node, int32_t segment_cardinality, bool is_undetermined) {
    BarcodeConfig barcode;
    barcode.key = key;
    barcode.index = get_int(node, "index");
    barcode.concentration = get_double(node, "concentration", 0);
    const Json* id = node.find("ID");
    if(id != nullptr && id->is_string()) { barcode.id = id->as_string(); }
    for(const char* tag : {"ID", "BC", "CN", "DS", "DT", "FO", "KS", "LB", "PG", "PI", "PL", "PM", "PU", "SM"}) {
        const Json* field = node.find(tag);
        if(field != nullptr && field->is_string() && !field->as_string().empty()) { barcode.read_group.emplace_back(tag, field->as_string()); }
    }
    barcode.output = get_string_array(node, "output");
    barcode.filter_outgoing_qc_fail = get_bool(node, "filter outgoing qc fail", false);
    const Json* segments = node.find("barcode");
    if(segments != nullptr && !segments->is_null()) {
        if(!segments->is_array() || static_cast<int32_t>(segments->elements().size()) != segment_cardinality) {
            throw ConfigurationError("barcode must have exactly " + std::to_string(segment_cardinality) + " segments");   // barcode.cpp:41
        }
        int32_t i = 0;
        for(const auto& e : segments->elements()) {
            if(!e.is_string()) { throw ConfigurationError("barcode segment " + std::to_string(i) + " must be a string"); }
            barcode.segment.push_back(e.as_string());
            ++i;
        }
    } else if(!is_undetermined) {
        throw ConfigurationError("barcode " + key + " has no sequence");
    }
    return barcode;
}

DecoderConfig parse_decoder(const Json& node, Kind kind) {
    if(!node.is_object()) { throw ConfigurationError("decoder element must be a dictionary"); }
    DecoderConfig d{};
    d.kind = kind;
    d.algorithm = parse_algorithm(node, kind);
    d.index = get_int(node, "index");
    d.multiplexing_classifier = get_bool(node, "multiplexing classifier", false);
    d.enable_quality_control = get_bool(node, "enable quality control", false);
    d.corrected_quality = get_int(node, "corrected quality", 30);

    const Json* undetermined = node.find("undetermined");
    if(undetermined == nullptr || !undetermined->is_object()) {
        throw ConfigurationError("classifier must declare an undetermined element");          // classifier.cpp:96
    }
    if(d.algorithm == Algorithm::PASSTHROUGH && node.find("transform") == nullptr) {
        // a bare Classifier<Barcode> never reads the transform (transcode.cpp:77-80)
        d.segment_cardinality = get_int(node, "segment cardinality", 0);
        d.nucleotide_cardinality = 0;
    } else {
        int32_t knit_segments = 0;
        d.transform = parse_rule(node, knit_segments);
        d.segment_cardinality = get_int(node, "segment cardinality");
        d.nucleotide_cardinality = get_int(node, "nucleotide cardinality");
        if(knit_segments != d.segment_cardinality) {
            throw ConfigurationError("transform knits " + std::to_string(knit_segments) + " segments but segment cardinality is " + std::to_string(d.segment_cardinality));
        }
        if(d.algorithm == Algorithm::PASSTHROUGH) { d.nucleotide_cardinality = 0; d.transform.clear(); }
    }
    if(d.segment_cardinality > MAX_SEGMENT) { throw ConfigurationError("more than " + std::to_string(MAX_SEGMENT) + " barcode segments are not supported"); }
    d.barcode_length = get_int_array(node, "barcode length");
    if(d.nucleotide_cardinality > 0) {
        if(static_cast<int32_t>(d.barcode_length.size()) != d.segment_cardinality) {
            throw ConfigurationError("barcode length cardinality inconsistant with segment cardinality");
        }
        int32_t total = 0;
        for(int32_t n : d.barcode_length) { d.segment_offset.push_back(total); total += n; }
        if(total != d.nucleotide_cardinality) { throw ConfigurationError("barcode length does not sum to nucleotide cardinality"); }
        // every observation segment must be filled by fixed width tokens of exactly that many cycles (transcode.cpp:834-849)
        std::vector<int32_t> filled(static_cast<size_t>(d.segment_cardinality), 0);
        for(const auto& t : d.transform) {
            if(t.length() <= 0) { throw ConfigurationError("token is empty or not fixed width"); }
            filled[static_cast<size_t>(t.output_segment_index)] += t.length();
        }
        for(int32_t i = 0; i < d.segment_cardinality; ++i) {
            if(filled[static_cast<size_t>(i)] != d.barcode_length[static_cast<size_t>(i)]) {
                throw ConfigurationError("tokens of segment " + std::to_string(i) + " do not add up to the barcode length");
            }
        }
        if(d.closed() && d.nucleotide_cardinality > MAX_NUCLEOTIDE) {
            // (33 to 128 nt take the general kernels — pamld_literal_long.hip, mdd_kernel over 8 or 16 words — at their stated rate)
            throw ConfigurationError("barcodes longer than " + std::to_string(MAX_NUCLEOTIDE) + " nucleotides are not supported");
        }
    }
    d.high_quality_threshold = get_int(node, "high quality threshold", 30);
    d.high_quality_distance_threshold = get_int(node, "high quality distance threshold", 0);
    d.quality_masking_threshold = get_int(node, "quality masking threshold", 0);
    d.distance_tolerance = get_int_array(node, "distance tolerance");
    d.noise = get_double(node, "noise", 0);
    d.confidence_threshold = get_double(node, "confidence threshold", 0);
    d.random_barcode_probability = get_double(node, "random barcode probability", 0);
    if(d.algorithm == Algorithm::PAMLD) {
        // PamlDecoder reads all three unconditionally (pamld.cpp:26-28)
        require(node, "noise"); require(node, "confidence threshold"); require(node, "random barcode probability");
        if(d.noise < 0 || d.noise > 1) { throw ConfigurationError("noise value " + std::to_string(d.noise) + " not between 0 and 1"); }              // transcode.cpp:1545-1553
        if(d.confidence_threshold < 0 || d.confidence_threshold > 1) { throw ConfigurationError("confidence threshold not between 0 and 1"); }
    }
    if(d.algorithm == Algorithm::MDD) {
        if(static_cast<int32_t>(d.distance_tolerance.size()) != d.segment_cardinality) {
            throw ConfigurationError(std::to_string(d.distance_tolerance.size()) + " distance tolerance cardinality inconsistant with " +
                                     std::to_string(d.segment_cardinality) + " barcode segment cardinality");   // metric.h:233-237
        }
    }

    d.undetermined = parse_barcode("undetermined", *undetermined, d.segment_cardinality, true);
    const Json* codec = node.find("codec");
    if(codec != nullptr && !codec->is_null()) {
        if(!codec->is_object()) { throw ConfigurationError("codec element must be a dictionary"); }
        for(const auto& member : codec->members()) {
            d.codec.push_back(parse_barcode(member.first, member.second, d.segment_cardinality, false));
        }
    }
    d.all_strict = true;
    d.barcode_code.assign(d.codec.size() * static_cast<size_t>(d.nucleotide_cardinality), 0);
    for(size_t b = 0; b < d.codec.size(); ++b) {
        size_t column = 0;
        for(int32_t s = 0; s < d.segment_cardinality; ++s) {
            const std::string& text = d.codec[b].segment[static_cast<size_t>(s)];
            if(static_cast<int32_t>(text.size()) != d.barcode_length[static_cast<size_t>(s)]) {
                throw ConfigurationError("expected " + std::to_string(d.barcode_length[static_cast<size_t>(s)]) + " but found " + std::to_string(text.size()) +
                                         " nucleotides in segment " + std::to_string(s) + " of barcode " + d.codec[b].key);
            }
            for(char c : text) {
                uint8_t code = ascii_to_bam(static_cast<uint8_t>(c));
                if(!is_strict_bam(code)) { d.all_strict = false; }
                d.barcode_code[b * static_cast<size_t>(d.nucleotide_cardinality) + column++] = code;
            }
        }
    }
    return d;
}

void parse_topic(const Json& document, const char* key, Kind kind, std::vector<DecoderConfig>& out) {
    const Json* node = document.find(key);
    if(node == nullptr || node->is_null()) { return; }
    if(node->is_object()) { out.push_back(parse_decoder(*node, kind)); }
    else if(node->is_array()) {
        if(kind == Kind::SAMPLE) { throw ConfigurationError("sample element must be a dictionary"); }
        for(const auto& e : node->elements()) { out.push_back(parse_decoder(e, kind)); }
    } else { throw ConfigurationError(std::string(key) + " element must be a dictionary or an array"); }
}

}  // namespace

JobConfig JobConfig::from_json(const Json& document) {
    if(!document.is_object()) { throw ConfigurationError("job document must be a dictionary"); }
    JobConfig job{};
    try {
        parse_topic(document, "sample", Kind::SAMPLE, job.decoder);
        parse_topic(document, "molecular", Kind::MOLECULAR, job.decoder);
        parse_topic(document, "cellular", Kind::CELLULAR, job.decoder);
    } catch(const Error& e) {
        throw Error(e.code, "TranscodingDecoder", e.what());     // error.push("TranscodingDecoder") transcode.cpp:44-47
    }
    job.input_segment_cardinality = get_int(document, "input segment cardinality", 0);
    job.leading_segment_index = get_int(document, "leading segment index", 0);
    job.filter_incoming_qc_fail = get_bool(document, "filter incoming qc fail", false);
    job.min_input_length = get_int_array(document, "min input length");
    job.float_precision = get_int(document, "float precision", 15);
    job.input_phred_offset = get_int(document, "input phred offset", 33);
    job.output_phred_offset = get_int(document, "output phred offset", 33);
    job.output_segment_cardinality = get_int(document, "output segment cardinality", 0);
    if(const Json* node = document.find("template")) {
        if(node->is_object() && node->find("transform") != nullptr) {
            int32_t segments = 0;
            job.template_transform = parse_rule(*node, segments);
            if(job.output_segment_cardinality == 0) { job.output_segment_cardinality = segments; }
        }
    }

    job.input = get_string_array(document, "input");
    job.output = get_string_array(document, "output");
    job.platform = "ILLUMINA";
    if(const Json* platform = document.find("platform")) { if(platform->is_string()) { job.platform = platform->as_string(); } }
    job.multiplexing_decoder = -1;
    job.enable_quality_control = get_bool(document, "enable quality control", false);

    int32_t multiplexing = 0;
    int32_t column = 0;
    int32_t nucleotides = 0;
    int32_t slot = 0;
    int64_t word = STAT_JOB_WORDS;
    for(auto& d : job.decoder) {
        if(d.multiplexing_classifier) {
            ++multiplexing;
            job.multiplexing_decoder = static_cast<int32_t>(&d - job.decoder.data());
            job.enable_quality_control = d.enable_quality_control;
        }
        d.plane_stride = (d.nucleotide_cardinality + 3) / 4 * 4;
        d.plane_prefix = column;
        column += d.plane_stride;
        nucleotides += d.nucleotide_cardinality;
        d.result_slot = d.closed() ? slot++ : -1;
        d.stats_offset = word;
        word += static_cast<int64_t>(d.codec.size() + 1) * STAT_WORDS_PER_BIN;
        if(job.input_segment_cardinality > 0) {
            for(const auto& t : d.transform) {
                if(!(t.input_segment_index < job.input_segment_cardinality) || t.input_segment_index < 0) {
                    throw ConfigurationError("invalid input feed reference " + std::to_string(t.input_segment_index) + " in token");
                }
            }
        }
    }
    if(multiplexing > 1) { throw ConfigurationError("more than one decoder is marked as the multiplexing classifier"); }
    job.plane_width = column;
    job.nucleotide_width = nucleotides;
    if(job.decoder.size() > DEV_MAX_DECODER) { throw ConfigurationError("more than " + std::to_string(DEV_MAX_DECODER) + " decoders are not supported"); }
    job.result_count = slot;
    job.stats_words = word;
    return job;
}

}  // namespace phq
