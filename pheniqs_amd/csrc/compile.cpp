#include "compile.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "error.hpp"
#include "url.hpp"

namespace phq {

namespace {

const char* const TOPICS[] = {"sample", "molecular", "cellular"};

// merge_json_value, json.cpp:780-803: members of the base fill in what the derived value does not define
void merge(const Json& base, Json& value) {
    if(base.is_null()) { return; }
    if(value.is_null()) { value = base; return; }
    if(!base.is_object()) { return; }
    if(!value.is_object()) { throw ConfigurationError("element is not a dictionary"); }
    for(const auto& member : base.members()) {
        Json* existing = value.find(member.first);
        if(existing != nullptr) {
            try {
                merge(member.second, *existing);
            } catch(ConfigurationError& error) {
                throw ConfigurationError(member.first + " " + error.message);
            }
        } else {
            value.set(member.first, member.second);
        }
    }
}

// compute_inheritence_depth, transcode.cpp:235-253
int depth_of(const std::string& key, const std::map<std::string, Json*>& by_key, std::map<std::string, int>& depth, int guard) {
    auto known = depth.find(key);
    if(known != depth.end()) { return known->second; }
    auto record = by_key.find(key);
    if(record == by_key.end()) { throw ConfigurationError("referencing an unknown parent " + key); }
    if(guard > static_cast<int>(by_key.size())) { throw ConfigurationError(key + " is part of a circular inheritance"); }
    int d = 0;
    const Json* base = record->second->find("base");
    if(base != nullptr && base->is_string()) {
        if(base->as_string() == key) { throw ConfigurationError(key + " references itself as parent"); }
        d = depth_of(base->as_string(), by_key, depth, guard + 1) + 1;
    }
    depth[key] = d;
    return d;
}

// Transcode::apply_repository_inheritence, transcode.cpp:351-396
void apply_repository_inheritance(Json& document) {
    Json* repository = document.find("decoder");
    if(repository == nullptr || !repository->is_object()) { return; }
    std::map<std::string, Json*> by_key;
    for(auto& member : repository->members()) {
        if(!member.second.is_null()) { by_key[member.first] = &member.second; }
    }
    std::map<std::string, int> depth;
    int max_depth = 0;
    for(const auto& record : by_key) { max_depth = std::max(max_depth, depth_of(record.first, by_key, depth, 0)); }
    for(int level = 1; level <= max_depth; ++level) {
        for(const auto& record : by_key) {
            if(depth[record.first] != level) { continue; }
            const Json* base = record.second->find("base");
            if(base != nullptr && base->is_string()) {
                const std::string parent = base->as_string();
                merge(*by_key[parent], *record.second);
                record.second->erase("base");
            }
        }
    }
}

// Transcode::apply_decoder_inheritance, transcode.cpp:419-436
void apply_decoder_inheritance(const Json* repository, Json& decoder) {
    if(!decoder.is_object()) { return; }
    const Json* base = decoder.find("base");
    if(base != nullptr && base->is_string() && repository != nullptr && repository->is_object()) {
        const Json* parent = repository->find(base->as_string());
        if(parent == nullptr) { throw ConfigurationError("reference to an unknown base " + base->as_string()); }
        merge(*parent, decoder);
    }
    decoder.erase("base");
}

struct Token {                                       // transform.h:35-88, transform.cpp:31
    int32_t input_segment_index = 0;
    int32_t start = 0;
    int32_t end = 0;
    bool end_terminated = false;

    explicit Token(const std::string& pattern) {
        // (s|c|m|[0-9]+):(-?[0-9]+)?:(-?[0-9]+)?
        const size_t first = pattern.find(':');
        const size_t second = first == std::string::npos ? std::string::npos : pattern.find(':', first + 1);
        if(first == std::string::npos || second == std::string::npos || pattern.find(':', second + 1) != std::string::npos || first == 0) {
            throw ConfigurationError("illegal token syntax " + pattern);
        }
        const std::string head = pattern.substr(0, first);
        const std::string a = pattern.substr(first + 1, second - first - 1);
        const std::string b = pattern.substr(second + 1);
        if(head == "s") { input_segment_index = -1; }
        else if(head == "c") { input_segment_index = -2; }
        else if(head == "m") { input_segment_index = -3; }
        else { input_segment_index = parse(head, false, pattern); }
        start = a.empty() ? 0 : parse(a, true, pattern);
        end_terminated = !b.empty();
        end = b.empty() ? 0 : parse(b, true, pattern);
    }
    static int32_t parse(const std::string& text, bool sign_allowed, const std::string& pattern) {
        size_t i = 0;
        if(sign_allowed && !text.empty() && text[0] == '-') { i = 1; }
        if(i >= text.size()) { throw ConfigurationError("illegal token syntax " + pattern); }
        for(size_t k = i; k < text.size(); ++k) {
            if(text[k] < '0' || text[k] > '9') { throw ConfigurationError("illegal token syntax " + pattern); }
        }
        return static_cast<int32_t>(std::stol(text));
    }
    bool constant() const {
        if(end_terminated) { return (start >= 0 && end >= 0) || (start < 0 && end < 0); }
        return start < 0;
    }
    bool empty() const {
        return (end_terminated && start >= end) && ((start >= 0 && end >= 0) || (start < 0 && end < 0));
    }
    int32_t length() const {
        if(constant()) {
            if(end_terminated) { return empty() ? 0 : end - start; }
            return -start;
        }
        return -1;
    }
};

double number_of(const Json& value, const std::string& key, double fallback) {
    const Json* found = value.find(key);
    return (found != nullptr && found->is_number()) ? found->as_double() : fallback;
}

void set_default(Json& value, const std::string& key, const Json& fallback) {
    if(value.find(key) == nullptr) { value.set(key, fallback); }
}

// WordMetric::find_shannon_bound, metric.h:87-111, over the distinct words of one segment
int32_t shannon_bound(std::vector<std::string> words, int32_t barcode_length) {
    std::sort(words.begin(), words.end());
    words.erase(std::unique(words.begin(), words.end()), words.end());
    int32_t min_distance = barcode_length;
    for(size_t i = 0; i < words.size(); ++i) {
        for(size_t j = i + 1; j < words.size(); ++j) {
            int32_t distance = 0;
            const size_t n = std::min(words[i].size(), words[j].size());
            for(size_t k = 0; k < n; ++k) { if(words[i][k] != words[j][k]) { ++distance; } }
            min_distance = std::min(min_distance, distance);
        }
    }
    return (min_distance - 1) / 2;                   // integer division truncates toward zero
}

// The "<type>:decoder" and "<type>:barcode" projections of configuration.json:419-490: the keys a decoder takes
// from the job's root (when it does not define them) and a barcode from its decoder, with their defaults
// (null = no default).  Transcode::compile_topic, transcode.cpp:769-800; project_json_value, json.cpp:804-835.
struct Projected { const char* key; Json fallback; };

std::vector<Projected> decoder_projection(const std::string& kind) {
    std::vector<Projected> keys = {
        {"algorithm", Json::string(kind == "molecular" ? "naive" : "pamld")}, {"codec", Json()}, {"confidence threshold", Json::real(0.95)},
        {"corrected quality", Json()}, {"distance tolerance", Json()}, {"high quality distance threshold", Json::integer(0)},
        {"high quality threshold", Json::integer(30)}, {"noise", Json::real(0.01)}, {"quality masking threshold", Json::integer(0)},
        {"undetermined", Json()}};
    if(kind == "sample") {
        for(const char* key : {"CN", "DT", "LB", "PG", "PI", "PL", "PM", "SM", "flowcell id", "flowcell lane number"}) { keys.push_back({key, Json()}); }
    }
    return keys;
}

std::vector<Projected> barcode_projection(const std::string& kind) {
    std::vector<Projected> keys = {{"algorithm", Json()}, {"concentration", Json::integer(1)}, {"segment cardinality", Json()}};
    if(kind == "sample") {
        for(const char* key : {"CN", "DT", "LB", "PG", "PI", "PL", "PM", "SM", "flowcell id", "flowcell lane number"}) { keys.push_back({key, Json()}); }
    }
    return keys;
}

// every projected key: the source's value when it has one, else the fallback; nulls are dropped (clean_json_value)
Json project(const std::vector<Projected>& keys, const Json& source, const Json* second_fallback) {
    Json out = Json::object();
    for(const auto& projected : keys) {
        const Json* found = source.find(projected.key);
        if(found != nullptr && !found->is_null()) { out.set(projected.key, *found); continue; }
        const Json* inherited = second_fallback != nullptr ? second_fallback->find(projected.key) : nullptr;
        if(inherited != nullptr && !inherited->is_null()) { out.set(projected.key, *inherited); continue; }
        if(!projected.fallback.is_null()) { out.set(projected.key, projected.fallback); }
    }
    return out;
}

// Transcode::infer_PU / infer_ID, transcode.cpp:1224-1260
void infer_PU_and_ID(Json& barcode, bool undetermined) {
    if(barcode.find("PU") == nullptr || !barcode.find("PU")->is_string()) {
        std::string suffix;
        if(!undetermined) {
            const Json* segments = barcode.find("barcode");
            if(segments != nullptr && segments->is_array()) {
                for(const auto& segment : segments->elements()) { if(segment.is_string()) { suffix += segment.as_string(); } }
            }
        } else {
            suffix = "undetermined";
        }
        if(!suffix.empty()) {
            std::string unit;
            const Json* flowcell = barcode.find("flowcell id");
            if(flowcell != nullptr && flowcell->is_string()) {
                unit = flowcell->as_string();
                unit.push_back(':');
                const Json* lane = barcode.find("flowcell lane number");
                if(lane != nullptr && lane->is_number()) { unit += std::to_string(lane->as_int()); unit.push_back(':'); }
            }
            barcode.set("PU", Json::string(unit + suffix));
        }
    }
    const Json* id = barcode.find("ID");
    if(id == nullptr || !id->is_string()) {
        const Json* unit = barcode.find("PU");
        if(unit != nullptr && unit->is_string()) { barcode.set("ID", *unit); }
    }
}

Json compile_decoder(const Json& spec, const Json& root, const std::string& kind, int32_t index, int32_t input_segment_cardinality) {
    if(!spec.is_object()) { throw ConfigurationError(kind + " decoder element must be a dictionary"); }
    Json value = spec;
    value.set("index", Json::integer(index));
    // overlay on top of the default decoder: the projection of the job's root (transcode.cpp:945-947)
    {
        Json corrected = Json::object();
        corrected.set("corrected quality", Json::integer(30));                      // configuration.json "corrected quality"
        merge(project(decoder_projection(kind), root, &corrected), value);
    }
    const std::string algorithm = value.find("algorithm")->is_string() ? value.find("algorithm")->as_string() : "";

    // transcode.cpp:735-763
    Json* transform = value.find("transform");
    if(transform == nullptr || !transform->is_object()) { throw ConfigurationError("decoder is missing a transform"); }
    const Json* token_list = transform->find("token");
    if(token_list == nullptr || !token_list->is_array() || token_list->elements().empty()) {
        throw ConfigurationError("transform is missing a token array");
    }
    const Json* knit_found = transform->find("knit");
    if(knit_found == nullptr || !knit_found->is_array() || knit_found->elements().empty()) {
        Json knit = Json::array();
        for(size_t i = 0; i < token_list->elements().size(); ++i) { knit.push(Json::string(std::to_string(i))); }
        transform->set("knit", knit);
    }
    token_list = transform->find("token");
    const Json* knit = transform->find("knit");

    // transcode.cpp:824-863
    std::vector<Token> tokens;
    for(size_t i = 0; i < token_list->elements().size(); ++i) {
        const Json& element = token_list->elements()[i];
        if(!element.is_string()) { throw ConfigurationError("token must be a string"); }
        tokens.emplace_back(element.as_string());
        const Token& token = tokens.back();
        if(input_segment_cardinality >= 0 && !(token.input_segment_index < input_segment_cardinality)) {
            throw ConfigurationError("invalid input feed reference " + std::to_string(token.input_segment_index) + " in token " + std::to_string(i));
        }
        if(token.empty()) { throw ConfigurationError("token " + element.as_string() + " is empty"); }
        if(!token.constant()) { throw ConfigurationError("token " + element.as_string() + " is not fixed width"); }
    }
    const int32_t segment_cardinality = static_cast<int32_t>(knit->elements().size());
    std::vector<int32_t> barcode_length(static_cast<size_t>(segment_cardinality), 0);
    int32_t nucleotide_cardinality = 0;
    for(int32_t output = 0; output < segment_cardinality; ++output) {
        const Json& pattern = knit->elements()[static_cast<size_t>(output)];
        if(!pattern.is_string()) { throw ConfigurationError("transform must be a string"); }
        const std::string& text = pattern.as_string();
        size_t position = 0;
        while(position <= text.size()) {
            size_t next = text.find(':', position);
            if(next == std::string::npos) { next = text.size(); }
            std::string piece = text.substr(position, next - position);
            if(!piece.empty() && piece[0] == '~') { piece = piece.substr(1); }
            if(piece.empty() || piece.find_first_not_of("0123456789") != std::string::npos) {
                throw ConfigurationError("illegal character in transform " + text);
            }
            const size_t reference = static_cast<size_t>(std::stoul(piece));
            if(reference >= tokens.size()) { throw ConfigurationError("invalid token reference " + std::to_string(reference) + " in transform"); }
            barcode_length[static_cast<size_t>(output)] += tokens[reference].length();
            nucleotide_cardinality += tokens[reference].length();
            position = next + 1;
        }
    }
    value.set("segment cardinality", Json::integer(segment_cardinality));
    value.set("nucleotide cardinality", Json::integer(nucleotide_cardinality));
    Json lengths = Json::array();
    for(int32_t n : barcode_length) { lengths.push(Json::integer(n)); }
    value.set("barcode length", lengths);
    const double lower_bound = 1.0 / std::pow(4.0, static_cast<double>(nucleotide_cardinality));
    const Json* rbp = value.find("random barcode probability");
    if(rbp != nullptr && rbp->is_number()) {
        if(rbp->as_double() < lower_bound) { throw ConfigurationError("random barcode probability is smaller than lower bound"); }
    } else {
        value.set("random barcode probability", Json::real(lower_bound));
    }

    // the default barcode induced by the codec: the barcode projection of the job's root, overridden by the decoder's
    // own values (transcode.cpp:949-962); algorithm and segment cardinality reach the barcodes this way
    const Json barcode_from_root = project(barcode_projection(kind), root, nullptr);
    const Json default_codec_barcode = project(barcode_projection(kind), value, &barcode_from_root);

    // transcode.cpp:867-882, 974-986
    const double noise = number_of(value, "noise", 0.01);
    {
        Json undetermined = value.find("undetermined") != nullptr && value.find("undetermined")->is_object() ? *value.find("undetermined") : Json::object();
        merge(default_codec_barcode, undetermined);
        Json barcode = Json::array();
        for(int32_t n : barcode_length) { barcode.push(Json::string(std::string(static_cast<size_t>(n), '='))); }
        undetermined.set("barcode", barcode);
        undetermined.set("segment cardinality", Json::integer(segment_cardinality));
        undetermined.set("index", Json::integer(0));
        undetermined.set("algorithm", Json::string(algorithm));
        infer_PU_and_ID(undetermined, true);
        undetermined.set("concentration", Json::real(noise));
        value.set("undetermined", undetermined);
    }

    Json* codec = value.find("codec");
    if(codec != nullptr && codec->is_object() && !codec->members().empty()) {
        // sort_json_value json.cpp:875-893: members are ordered by key before compile_decoder walks them
        std::stable_sort(codec->members().begin(), codec->members().end(),
                         [](const std::pair<std::string, Json>& a, const std::pair<std::string, Json>& b) { return a.first < b.first; });
        std::set<std::string> seen;
        std::set<std::string> unique_id;
        if(const Json* id = value.find("undetermined")->find("ID")) { if(id->is_string()) { unique_id.insert(id->as_string()); } }
        int32_t barcode_index = 1;
        double total_concentration = 0;
        std::vector<std::vector<std::string>> words(static_cast<size_t>(segment_cardinality));
        for(auto& member : codec->members()) {
            Json& record = member.second;
            if(!record.is_object()) { throw ConfigurationError("barcode " + member.first + " must be a dictionary"); }
            const Json* segments = record.find("barcode");
            if(segments == nullptr || !segments->is_array() || static_cast<int32_t>(segments->elements().size()) != segment_cardinality) {
                throw ConfigurationError("expected " + std::to_string(segment_cardinality) + " segments in barcode " + member.first);
            }
            std::string flat;
            std::string joined;
            for(int32_t s = 0; s < segment_cardinality; ++s) {
                const Json& segment = segments->elements()[static_cast<size_t>(s)];
                if(!segment.is_string()) { throw ConfigurationError("barcode segment must be a string in " + member.first); }
                if(static_cast<int32_t>(segment.as_string().size()) != barcode_length[static_cast<size_t>(s)]) {
                    throw ConfigurationError("expected " + std::to_string(barcode_length[static_cast<size_t>(s)]) + " but found " +
                                             std::to_string(segment.as_string().size()) + " nucleotides in segment " + std::to_string(s) + " of barcode " + member.first);
                }
                flat += segment.as_string();
                if(s > 0) { joined.push_back('-'); }
                joined += segment.as_string();
                words[static_cast<size_t>(s)].push_back(segment.as_string());
            }
            if(!seen.insert(flat).second) { throw ConfigurationError("duplicate barcode sequence " + flat); }
            merge(default_codec_barcode, record);
            record.set("segment cardinality", Json::integer(segment_cardinality));
            record.set("algorithm", Json::string(algorithm));
            record.set("index", Json::integer(barcode_index));
            infer_PU_and_ID(record, false);
            if(const Json* id = record.find("ID")) {
                if(id->is_string() && !unique_id.insert(id->as_string()).second) { throw ConfigurationError("duplicate " + member.first + " barcode"); }
            }
            const double concentration = number_of(record, "concentration", 1.0);
            if(concentration < 0) { throw ConfigurationError("barcode concentration must be a positive number"); }
            total_concentration += concentration;
            record.set("BC", Json::string(joined));
            ++barcode_index;
        }
        value.set("barcode cardinality", Json::integer(barcode_index));
        if(!(total_concentration > 0)) { throw ConfigurationError("total pool concentration is not a positive number"); }
        const double factor = (1.0 - noise) / total_concentration;                 // transcode.cpp:1024-1029
        codec = value.find("codec");
        for(auto& member : codec->members()) {
            member.second.set("concentration", Json::real(number_of(member.second, "concentration", 1.0) * factor));
        }

        // metric.h:216-242
        std::vector<int32_t> bound;
        Json bound_json = Json::array();
        for(int32_t s = 0; s < segment_cardinality; ++s) {
            bound.push_back(shannon_bound(words[static_cast<size_t>(s)], barcode_length[static_cast<size_t>(s)]));
            bound_json.push(Json::integer(bound.back()));
        }
        value.set("shannon bound", bound_json);
        const Json* tolerance = value.find("distance tolerance");
        if(tolerance != nullptr && !tolerance->is_null()) {
            if(!tolerance->is_array() || static_cast<int32_t>(tolerance->elements().size()) != segment_cardinality) {
                throw ConfigurationError("distance tolerance cardinality inconsistant with barcode segment cardinality");
            }
            for(int32_t s = 0; s < segment_cardinality; ++s) {
                if(tolerance->elements()[static_cast<size_t>(s)].as_int() > bound[static_cast<size_t>(s)]) {
                    throw ConfigurationError("barcode tolerance for segment " + std::to_string(s) + " is higher than shannon bound " +
                                             std::to_string(bound[static_cast<size_t>(s)]));
                }
            }
        } else {
            value.set("distance tolerance", bound_json);
        }
    }
    return value;
}

bool decoder_is_compiled(const Json& decoder) {
    return decoder.is_object() && decoder.find("nucleotide cardinality") != nullptr && decoder.find("segment cardinality") != nullptr &&
           decoder.find("index") != nullptr && decoder.find("barcode length") != nullptr && decoder.find("base") == nullptr;
}

// The IO directives: Transcode::compile_input (transcode.cpp:500-557, explicit layout 683-733), compile_template
// (1419-1445) and compile_output (1261-1417) as far as FASTQ / SAM text feeds go.  Input paths go under `base input
// url`; a job without a template routes every input segment to the output verbatim; `output` and `filter outgoing qc
// fail` project from the root through the multiplexing decoder onto each of its barcodes (projections
// "multiplex:decoder" / "multiplex:barcode", configuration.json:453-463), output paths go under `base output url`, a
// single output path is repeated for every output segment.
std::string string_member(const Json& value, const char* key) {
    const Json* found = value.find(key);
    return found != nullptr && found->is_string() ? found->as_string() : std::string();
}

void relocate_array(Json& array, const std::string& base) {
    if(!array.is_array()) { throw ConfigurationError("URL array element must be an array"); }
    for(auto& element : array.elements()) {
        if(element.is_null()) { throw ConfigurationError("URL element is null"); }
        if(!element.is_string()) { throw ConfigurationError("URL element must be a string"); }
        Url url = Url::parse(element.as_string());
        url.relocate_child(base);
        element = Json::string(url.encoded());
    }
}

Json* multiplexing_decoder_of(Json& job) {
    for(const char* topic : TOPICS) {
        Json* node = job.find(topic);
        if(node == nullptr) { continue; }
        auto is_marked = [](const Json& d) { const Json* m = d.find("multiplexing classifier"); return m != nullptr && m->is_bool() && m->as_bool(); };
        if(node->is_object()) { if(is_marked(*node)) { return node; } }
        else if(node->is_array()) { for(auto& element : node->elements()) { if(is_marked(element)) { return &element; } } }
    }
    return nullptr;
}

void compile_io(Json& job, int32_t input_segment_cardinality) {
    if(Json* input = job.find("input")) {
        if(!input->is_null()) { relocate_array(*input, string_member(job, "base input url")); }
    }

    // compile_template
    if(job.find("template") == nullptr || job.find("template")->is_null()) { job.set("template", Json::object()); }
    Json& pattern = *job.find("template");
    if(!pattern.is_object()) { throw ConfigurationError("template element must be a dictionary"); }
    if(pattern.find("transform") == nullptr || pattern.find("transform")->is_null()) { pattern.set("transform", Json::object()); }
    Json& transform = *pattern.find("transform");
    if(!transform.is_object()) { throw ConfigurationError("template transform element must be a dictionary"); }
    if(transform.find("token") == nullptr) {
        if(input_segment_cardinality < 0) { return; }            // nothing states the input layout: classification only
        Json token = Json::array();
        for(int32_t i = 0; i < input_segment_cardinality; ++i) { token.push(Json::string(std::to_string(i) + "::")); }
        transform.set("token", token);
    }
    const Json& token = *transform.find("token");
    if(!token.is_array()) { throw ConfigurationError("transform token element is not an array"); }
    const Json* knit = transform.find("knit");
    if(knit == nullptr || knit->is_null() || (knit->is_array() && knit->elements().empty())) {
        Json observation = Json::array();
        for(size_t i = 0; i < token.elements().size(); ++i) { observation.push(Json::string(std::to_string(i))); }
        transform.set("knit", observation);
    }
    const int32_t output_segment_cardinality = static_cast<int32_t>(transform.find("knit")->elements().size());
    job.set("output segment cardinality", Json::integer(output_segment_cardinality));

    // compile_output; a job that names no output writes SAM to the standard output (configuration.json:13-15)
    if(job.find("output") == nullptr || job.find("output")->is_null()) {
        Json standard = Json::array();
        standard.push(Json::string("/dev/stdout"));
        job.set("output", standard);
    }
    Json* decoder = multiplexing_decoder_of(job);
    if(decoder == nullptr) { return; }
    auto inherit = [](Json& to, const Json& from, const char* key) {
        const Json* own = to.find(key);
        if(own != nullptr && !own->is_null()) { return; }
        const Json* parent = from.find(key);
        if(parent != nullptr && !parent->is_null()) { to.set(key, *parent); }
    };
    for(const char* key : {"base output url", "filter outgoing qc fail", "enable quality control", "output"}) { inherit(*decoder, job, key); }
    const std::string base = string_member(*decoder, "base output url");
    std::vector<Json*> channel;
    if(Json* undetermined = decoder->find("undetermined")) { if(undetermined->is_object()) { channel.push_back(undetermined); } }
    if(Json* codec = decoder->find("codec")) {
        if(codec->is_object()) { for(auto& member : codec->members()) { if(member.second.is_object()) { channel.push_back(&member.second); } } }
    }
    for(Json* record : channel) {
        for(const char* key : {"filter outgoing qc fail", "enable quality control", "output"}) { inherit(*record, *decoder, key); }
        record->set("TC", Json::integer(output_segment_cardinality));
        Json* output = record->find("output");
        if(output == nullptr || output->is_null()) { continue; }
        relocate_array(*output, base);
        for(const auto& element : output->elements()) {
            const Url url = Url::parse(element.as_string());
            if(url.is_stdin()) { throw ConfigurationError("output stream can not be set to standard input"); }
            if(url.is_stderr()) { throw ConfigurationError("output stream can not be set to standard error"); }
        }
        // pad_url_array_by_key, transcode.cpp:1446-1460
        const size_t stated = output->elements().size();
        if(stated > 0 && static_cast<int32_t>(stated) != output_segment_cardinality) {
            if(stated != 1) { throw ConfigurationError("incorrect number of output URLs in channel"); }
            const Json first = output->elements().front();
            while(static_cast<int32_t>(output->elements().size()) < output_segment_cardinality) { output->push(first); }
        }
    }
}

}  // namespace

bool is_compiled_job(const Json& document) {
    if(!document.is_object()) { return true; }      // JobConfig::from_json reports the error
    if(document.find("decoder") != nullptr) { return false; }
    for(const char* topic : TOPICS) {
        const Json* node = document.find(topic);
        if(node == nullptr || node->is_null()) { continue; }
        if(node->is_object()) { if(!decoder_is_compiled(*node)) { return false; } }
        else if(node->is_array()) { for(const auto& element : node->elements()) { if(!decoder_is_compiled(element)) { return false; } } }
    }
    return true;
}

Json assemble_job(const Json& document) {
    if(!document.is_object()) { throw ConfigurationError("job document must be a dictionary"); }
    Json job = document.find("import") != nullptr ? resolve_imports(document, ".") : document;      // a document in memory: imports relative to the working directory
    apply_repository_inheritance(job);
    const Json repository = job.find("decoder") != nullptr ? *job.find("decoder") : Json();
    for(const char* topic : TOPICS) {
        Json* node = job.find(topic);
        if(node == nullptr || node->is_null()) { continue; }
        try {
            if(node->is_object()) { apply_decoder_inheritance(repository.is_null() ? nullptr : &repository, *node); }
            else if(node->is_array()) {
                int32_t index = 0;
                for(auto& element : node->elements()) {
                    try { apply_decoder_inheritance(repository.is_null() ? nullptr : &repository, element); }
                    catch(ConfigurationError& error) { throw ConfigurationError(std::string(topic) + " decoder at " + std::to_string(index) + " : " + error.message); }
                    ++index;
                }
            }
        } catch(ConfigurationError& error) {
            if(error.message.compare(0, std::string(topic).size(), topic) == 0) { throw; }
            throw ConfigurationError(std::string(topic) + " decoder : " + error.message);
        }
    }
    job.erase("decoder");
    return job;
}

Json compile_job(const Json& document) {
    if(!document.is_object()) { throw ConfigurationError("job document must be a dictionary"); }
    Json job = document.find("import") != nullptr ? resolve_imports(document, ".") : document;      // a document in memory: imports relative to the working directory
    apply_repository_inheritance(job);
    const Json repository = job.find("decoder") != nullptr ? *job.find("decoder") : Json();

    // the number of input segments: stated, or one per entry of "input" (transcode.cpp:693-701)
    int32_t input_segment_cardinality = -1;
    if(const Json* stated = job.find("input segment cardinality")) {
        if(stated->is_number()) { input_segment_cardinality = static_cast<int32_t>(stated->as_int()); }
    } else if(const Json* input = job.find("input")) {
        if(input->is_array()) {
            input_segment_cardinality = static_cast<int32_t>(input->elements().size());
            job.set("input segment cardinality", Json::integer(input_segment_cardinality));
        }
    }

    for(const char* topic : TOPICS) {
        Json* node = job.find(topic);
        if(node == nullptr) { continue; }
        if(node->is_null()) { job.erase(topic); continue; }
        try {
            if(node->is_object()) {
                apply_decoder_inheritance(repository.is_null() ? nullptr : &repository, *node);
                *node = compile_decoder(*node, job, topic, 0, input_segment_cardinality);
            } else if(node->is_array()) {
                int32_t index = 0;
                for(auto& element : node->elements()) {
                    try {
                        apply_decoder_inheritance(repository.is_null() ? nullptr : &repository, element);
                        element = compile_decoder(element, job, topic, index, input_segment_cardinality);
                    } catch(ConfigurationError& error) {
                        throw ConfigurationError(std::string(topic) + " decoder at " + std::to_string(index) + " : " + error.message);
                    }
                    ++index;
                }
            } else {
                throw ConfigurationError("element must be a dictionary or an array");
            }
        } catch(ConfigurationError& error) {
            if(error.message.compare(0, std::string(topic).size(), topic) == 0) { throw; }
            throw ConfigurationError(std::string(topic) + " decoder : " + error.message);
        }
    }
    job.erase("decoder");                              // the repository is not part of the compiled instruction

    // Transcode::compile_multiplexing_decoder transcode.cpp:1087-1223, its common outcome: an explicitly marked
    // decoder wins, otherwise the sample decoder drives output routing
    bool marked = false;
    for(const char* topic : TOPICS) {
        const Json* node = job.find(topic);
        if(node == nullptr) { continue; }
        auto is_marked = [](const Json& d) { const Json* m = d.find("multiplexing classifier"); return m != nullptr && m->is_bool() && m->as_bool(); };
        if(node->is_object()) { marked = marked || is_marked(*node); }
        else if(node->is_array()) { for(const auto& element : node->elements()) { marked = marked || is_marked(element); } }
    }
    if(!marked) {
        Json* sample = job.find("sample");
        if(sample != nullptr && sample->is_object()) { sample->set("multiplexing classifier", Json::boolean(true)); }
    }
    set_default(job, "filter incoming qc fail", Json::boolean(false));
    set_default(job, "leading segment index", Json::integer(0));
    compile_io(job, input_segment_cardinality);
    return job;
}

// ---- file imports (job.cpp:163-224, json.cpp:780-803) ---------------------------------------------------------

namespace {

// merge_json_value(base, ontology): members of `base` that `ontology` lacks are copied in, shared dictionaries recurse
void merge_json(const Json& base, Json& ontology) {
    if(base.is_null()) { return; }
    if(ontology.is_null()) { ontology = base; return; }
    if(!base.is_object()) { return; }
    if(!ontology.is_object()) { throw ConfigurationError("element is not a dictionary"); }
    for(const auto& member : base.members()) {
        Json* mine = ontology.find(member.first);
        if(mine != nullptr) {
            try { merge_json(member.second, *mine); }
            catch(ConfigurationError& error) { throw ConfigurationError(member.first + " " + error.message); }
        } else {
            ontology.set(member.first, member.second);
        }
    }
}

std::string directory_of(const std::string& path) {
    const size_t slash = path.find_last_of('/');
    return slash == std::string::npos ? std::string(".") : (slash == 0 ? std::string("/") : path.substr(0, slash));
}

// lexical normal form of a path (no file system access): what tells two spellings of one import apart
std::string normal_path(const std::string& path) {
    std::vector<std::string> part;
    const bool absolute = !path.empty() && path[0] == '/';
    size_t at = 0;
    while(at <= path.size()) {
        const size_t next = std::min(path.find('/', at), path.size());
        const std::string piece = path.substr(at, next - at);
        if(piece == "..") { if(!part.empty() && part.back() != "..") { part.pop_back(); } else if(!absolute) { part.push_back(piece); } }
        else if(!piece.empty() && piece != ".") { part.push_back(piece); }
        at = next + 1;
    }
    std::string out = absolute ? "/" : "";
    for(size_t k = 0; k < part.size(); ++k) { out += (k ? "/" : "") + part[k]; }
    return out.empty() ? std::string(".") : out;
}

std::string expand_home(const std::string& path) {
    if(!path.empty() && path[0] == '~' && (path.size() == 1 || path[1] == '/')) {
        const char* home = std::getenv("HOME");
        if(home != nullptr) { return std::string(home) + path.substr(1); }
    }
    return path;
}

Json load_with_import(const std::string& path, std::vector<std::string>& visited);

Json resolve(Json document, const std::string& directory, std::vector<std::string>& visited) {
    if(!document.is_object()) { throw ConfigurationError("instruction root must be a dictionary"); }
    const Json* import = document.find("import");
    if(import != nullptr) {
        if(!import->is_array()) { throw ConfigurationError("import element must be an array of paths"); }
        Json aggregated;
        for(const auto& record : import->elements()) {
            if(!record.is_string()) { throw ConfigurationError("import element must be an array of paths"); }
            std::string path = expand_home(record.as_string());
            if(path.empty() || path[0] != '/') { path = directory + "/" + path; }     // relative to the importing document
            path = normal_path(path);
            // a path is only visited once, the first time a depth first walk meets it: no cyclical imports
            if(std::find(visited.begin(), visited.end(), path) == visited.end()) {
                Json imported = load_with_import(path, visited);
                merge_json(aggregated, imported);                                        // the later import wins
                aggregated = std::move(imported);
            }
        }
        merge_json(aggregated, document);                                                // the importing document wins
        document.erase("import");
    }
    return document;
}

Json load_with_import(const std::string& path, std::vector<std::string>& visited) {
    std::FILE* file = std::fopen(path.c_str(), "rb");
    if(file == nullptr) { throw ConfigurationError("unable to read instruction file from " + path); }
    std::string text;
    char buffer[1 << 16];
    size_t got;
    while((got = std::fread(buffer, 1, sizeof(buffer), file)) > 0) { text.append(buffer, got); }
    std::fclose(file);
    visited.push_back(path);
    return resolve(Json::parse(text), directory_of(path), visited);
}

}  // namespace

Json load_job_document(const std::string& path) {
    std::vector<std::string> visited;
    return load_with_import(normal_path(expand_home(path)), visited);
}

Json resolve_imports(Json document, const std::string& directory) {
    std::vector<std::string> visited;
    return resolve(std::move(document), directory.empty() ? std::string(".") : directory, visited);
}

}  // namespace phq
