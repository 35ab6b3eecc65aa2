// extern "C" surface of libpheniqs_amd.so (declared in include/pheniqs_amd.h).  Exceptions stop here:
// every entry point converts an Error into its reference ErrorCode (error.h:32-44) and stores the text.
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/pheniqs_amd.h"
#include "compile.hpp"
#include "decoder_set.hpp"
#include "error.hpp"
#include "stats.hpp"
#include "transcoder.hpp"

struct phq_decoder_set {
    phq::DecoderSet impl;
    phq_decoder_set(const char* json, int32_t device) : impl(json, device) {}
};

namespace {

thread_local std::string global_error;

template <typename F>
int guarded(phq_decoder_set* set, F&& body) {
    std::string* sink = set ? &set->impl.last_error : &global_error;
    try {
        body();
        return PHQ_OK;
    } catch(const phq::Error& e) {
        *sink = e.what();
        return static_cast<int>(e.code);
    } catch(const std::bad_alloc&) {
        *sink = "Out of memory error";
        return PHQ_OUT_OF_MEMORY_ERROR;
    } catch(const std::exception& e) {
        *sink = std::string("Unknown error : ") + e.what();
        return PHQ_UNKNOWN_ERROR;
    } catch(...) {
        *sink = "Unknown error";
        return PHQ_UNKNOWN_ERROR;
    }
}

void require(const void* pointer, const char* name) {
    if(pointer == nullptr) { throw phq::InternalError(std::string(name) + " is NULL"); }
}

static_assert(sizeof(phq_result_t) == sizeof(phq::DevResult), "result record layout");
static_assert(sizeof(phq_result_t) == 16, "result record is 16 bytes");

}  // namespace

extern "C" {

int phq_create(const char* compiled_job_json, int32_t device, phq_decoder_set** out) {
    if(out != nullptr) { *out = nullptr; }
    return guarded(nullptr, [&]() {
        require(compiled_job_json, "compiled_job_json");
        require(out, "out");
        *out = new phq_decoder_set(compiled_job_json, device);
    });
}

void phq_destroy(phq_decoder_set* set) { delete set; }

const char* phq_last_error(const phq_decoder_set* set) {
    return set ? set->impl.last_error.c_str() : global_error.c_str();
}

int phq_get_layout(const phq_decoder_set* set, phq_layout_t* layout) {
    return guarded(const_cast<phq_decoder_set*>(set), [&]() {
        require(set, "set"); require(layout, "layout");
        const phq::JobConfig& job = set->impl.job();
        std::memset(layout, 0, sizeof(*layout));
        layout->n_decoder = static_cast<int32_t>(job.decoder.size());
        layout->n_result = job.result_count;
        layout->plane_width = job.plane_width;
        layout->nucleotide_width = job.nucleotide_width;
        layout->stats_words = job.stats_words;
        layout->input_segment_cardinality = job.input_segment_cardinality;
    });
}

int phq_get_decoder_info(const phq_decoder_set* set, int32_t decoder, phq_decoder_info_t* info) {
    return guarded(const_cast<phq_decoder_set*>(set), [&]() {
        require(set, "set"); require(info, "info");
        const phq::JobConfig& job = set->impl.job();
        if(decoder < 0 || decoder >= static_cast<int32_t>(job.decoder.size())) { throw phq::InternalError("decoder ordinal out of range"); }
        const phq::DecoderConfig& d = job.decoder[static_cast<size_t>(decoder)];
        std::memset(info, 0, sizeof(*info));
        info->kind = static_cast<int32_t>(d.kind);
        info->algorithm = static_cast<int32_t>(d.algorithm);
        info->index = d.index;
        info->n_barcode = static_cast<int32_t>(d.codec.size());
        info->nucleotide_cardinality = d.nucleotide_cardinality;
        info->segment_cardinality = d.segment_cardinality;
        info->plane_stride = d.plane_stride;
        info->plane_prefix = d.plane_prefix;
        info->result_slot = d.result_slot;
        info->multiplexing = d.multiplexing_classifier ? 1 : 0;
        info->stats_offset = d.stats_offset;
    });
}

int phq_set_mode(phq_decoder_set* set, int32_t mode) {
    return guarded(set, [&]() { require(set, "set"); set->impl.set_mode(mode); });
}

int phq_stage_reads(const phq_decoder_set* set, int64_t n_read, int32_t n_segment,
                    const uint8_t* const* code, const uint8_t* const* quality,
                    const int64_t* const* offset, const int32_t* const* length, const uint8_t* in_flag,
                    int64_t batch_capacity, uint8_t* base_planes, uint8_t* quality_planes,
                    uint8_t* out_flag, int64_t* kept, int64_t* n_kept,
                    uint64_t* incoming_count, uint64_t* incoming_pf_count) {
    return guarded(const_cast<phq_decoder_set*>(set), [&]() {
        require(set, "set"); require(code, "code"); require(quality, "quality"); require(offset, "offset"); require(length, "length");
        set->impl.stage_reads(n_read, n_segment, code, quality, offset, length, in_flag, batch_capacity, base_planes, quality_planes,
                              out_flag, kept, n_kept, incoming_count, incoming_pf_count);
    });
}

int phq_classify_device(phq_decoder_set* set, int64_t batch_capacity, int64_t first_read, int64_t n_read,
                        const uint8_t* base_planes, const uint8_t* quality_planes, const uint8_t* flag,
                        phq_result_t* result, uint32_t* word, void* hip_stream) {
    return guarded(set, [&]() {
        require(set, "set");
        set->impl.classify_device(batch_capacity, first_read, n_read, base_planes, quality_planes, flag,
                                  reinterpret_cast<phq::DevResult*>(result), word, hip_stream);
    });
}

int phq_classify_host(phq_decoder_set* set, int64_t batch_capacity, int64_t n_read,
                      const uint8_t* base_planes, const uint8_t* quality_planes, const uint8_t* flag,
                      phq_result_t* result, uint32_t* word) {
    return guarded(set, [&]() {
        require(set, "set");
        set->impl.classify_host(batch_capacity, n_read, base_planes, quality_planes, flag, reinterpret_cast<phq::DevResult*>(result), word);
    });
}

int phq_pack_planes(phq_decoder_set* set, int64_t batch_capacity, int64_t n_read, const uint8_t* base_planes, const uint8_t* quality_planes,
                    uint8_t* packed_planes, uint64_t* escapes, int64_t escape_capacity, int64_t* n_escape) {
    if(n_escape != nullptr) { *n_escape = 0; }
    return guarded(set, [&]() {
        require(set, "set");
        require(base_planes, "base_planes");
        require(quality_planes, "quality_planes");
        require(packed_planes, "packed_planes");
        require(n_escape, "n_escape");
        *n_escape = set->impl.pack_planes(batch_capacity, n_read, base_planes, quality_planes, packed_planes, escapes, escape_capacity);
    });
}

int phq_classify_host_packed(phq_decoder_set* set, int64_t batch_capacity, int64_t n_read, const uint8_t* packed_planes,
                             const uint64_t* escapes, int64_t n_escape, const uint8_t* flag, phq_result_t* result, uint32_t* word) {
    return guarded(set, [&]() {
        require(set, "set");
        set->impl.classify_host_packed(batch_capacity, n_read, packed_planes, escapes, n_escape, flag, reinterpret_cast<phq::DevResult*>(result), word);
    });
}

int phq_stats_reset(phq_decoder_set* set, void* hip_stream) {
    return guarded(set, [&]() { require(set, "set"); set->impl.stats_reset(hip_stream); });
}

int phq_stats_device_pointer(phq_decoder_set* set, uint64_t** device_words) {
    return guarded(set, [&]() { require(set, "set"); require(device_words, "device_words"); *device_words = set->impl.stats_device_pointer(); });
}

int phq_stats_export(phq_decoder_set* set, uint64_t* host_words) {
    return guarded(set, [&]() { require(set, "set"); require(host_words, "host_words"); set->impl.stats_export(host_words); });
}

int phq_stats_import(phq_decoder_set* set, const uint64_t* host_words) {
    return guarded(set, [&]() { require(set, "set"); require(host_words, "host_words"); set->impl.stats_import(host_words); });
}

int phq_stats_merge(phq_decoder_set* dst, phq_decoder_set* src) {
    return guarded(dst, [&]() {
        require(dst, "dst"); require(src, "src");
        const int64_t words = dst->impl.job().stats_words;
        if(words != src->impl.job().stats_words) { throw phq::ConfigurationError("cannot merge statistics of different jobs"); }
        std::vector<uint64_t> a(static_cast<size_t>(words)), b(static_cast<size_t>(words));
        dst->impl.stats_export(a.data());
        src->impl.stats_export(b.data());
        for(int64_t i = 0; i < words; ++i) { a[static_cast<size_t>(i)] += b[static_cast<size_t>(i)]; }
        dst->impl.stats_import(a.data());
    });
}

int phq_stats_allreduce(phq_decoder_set* set, void* rccl_comm, void* hip_stream) {
    return guarded(set, [&]() { require(set, "set"); require(rccl_comm, "rccl_comm"); set->impl.stats_allreduce(rccl_comm, hip_stream); });
}

int phq_summarize(phq_decoder_set* set, int64_t batch_capacity, int64_t n_read, const phq_result_t* result, uint32_t* distance, double* confidence) {
    return guarded(set, [&]() {
        require(set, "set");
        set->impl.summarize(batch_capacity, n_read, reinterpret_cast<const phq::DevResult*>(result), distance, confidence);
    });
}

int phq_stats_options(phq_decoder_set* set, const uint64_t* host_words, int32_t decoder, phq_option_t* option) {
    return guarded(set, [&]() {
        require(set, "set"); require(host_words, "host_words"); require(option, "option");
        const phq::JobConfig& job = set->impl.job();
        if(decoder < 0 || decoder >= static_cast<int32_t>(job.decoder.size())) { throw phq::InternalError("decoder ordinal out of range"); }
        const auto counters = phq::decode_counters(job.decoder[static_cast<size_t>(decoder)], host_words);
        for(size_t b = 0; b < counters.size(); ++b) {
            option[b].count = counters[b].count;
            option[b].pf_count = counters[b].pf_count;
            option[b].accumulated_distance = counters[b].accumulated_distance;
            option[b].accumulated_pf_distance = counters[b].accumulated_pf_distance;
            option[b].low_conditional_confidence_count = counters[b].low_conditional_confidence_count;
            option[b].low_confidence_count = counters[b].low_confidence_count;
            option[b].accumulated_confidence = counters[b].accumulated_confidence;
            option[b].accumulated_pf_confidence = counters[b].accumulated_pf_confidence;
        }
    });
}

int phq_report_json(phq_decoder_set* set, const uint64_t* host_words, uint64_t incoming_count,
                    uint64_t incoming_pf_count, const char** json) {
    return guarded(set, [&]() {
        require(set, "set"); require(host_words, "host_words"); require(json, "json");
        *json = set->impl.report_json(host_words, incoming_count, incoming_pf_count).c_str();
    });
}

int phq_profile_enable(phq_decoder_set* set, int32_t enable) {
    return guarded(set, [&]() { require(set, "set"); set->impl.profile_enable(enable != 0); });
}

int phq_profile_collect(phq_decoder_set* set, double* milliseconds, int64_t* launches) {
    return guarded(set, [&]() {
        require(set, "set"); require(milliseconds, "milliseconds"); require(launches, "launches");
        set->impl.profile_collect(milliseconds, launches);
    });
}

int phq_copy_bandwidth(int32_t device, uint64_t bytes, int32_t repeats, double* gigabytes_per_second) {
    return guarded(nullptr, [&]() {
        require(gigabytes_per_second, "gigabytes_per_second");
        *gigabytes_per_second = 0;
        if(bytes < 16 || bytes % 16 != 0 || repeats < 1) { throw phq::InternalError("copy probe: a multiple of 16 bytes, at least one repeat"); }
        const int status = phq::copy_bandwidth(device, static_cast<size_t>(bytes), repeats, gigabytes_per_second);
        if(status != 0) { throw phq::InternalError("copy probe: HIP error " + std::to_string(status)); }
    });
}

int phq_device_path_counters(uint64_t counters[4]) {
    return guarded(nullptr, [&]() {
        require(counters, "counters");
        phq::device_path_counters(counters);
    });
}

int phq_plan_chain(phq_decoder_set* set, int64_t n_read, int32_t mode, char** chain_json) {
    if(chain_json != nullptr) { *chain_json = nullptr; }
    return guarded(set, [&]() {
        require(set, "set"); require(chain_json, "chain_json");
        const std::string text = set->impl.plan_chain(n_read, mode);
        char* const out = static_cast<char*>(std::malloc(text.size() + 1));
        if(out == nullptr) { throw std::bad_alloc(); }
        std::memcpy(out, text.c_str(), text.size() + 1);
        *chain_json = out;
    });
}

int phq_profile_chain(phq_decoder_set* set, char** chain_json) {
    if(chain_json != nullptr) { *chain_json = nullptr; }
    return guarded(set, [&]() {
        require(set, "set"); require(chain_json, "chain_json");
        const std::string text = set->impl.profile_chain();
        char* const out = static_cast<char*>(std::malloc(text.size() + 1));
        if(out == nullptr) { throw std::bad_alloc(); }
        std::memcpy(out, text.c_str(), text.size() + 1);
        *chain_json = out;
    });
}

int phq_transcode_fastq_to_sam(phq_decoder_set* set, int32_t n_input, const char* const* input_path, const char* sam_output_path,
                               int64_t batch_capacity, uint64_t* incoming_count, uint64_t* incoming_pf_count) {
    return guarded(set, [&]() {
        require(set, "set"); require(input_path, "input_path"); require(sam_output_path, "sam_output_path");
        require(incoming_count, "incoming_count"); require(incoming_pf_count, "incoming_pf_count");
        if(batch_capacity <= 0) { throw phq::InternalError("batch capacity must be positive"); }
        std::vector<std::string> input;
        for(int32_t i = 0; i < n_input; ++i) { require(input_path[i], "input_path[i]"); input.emplace_back(input_path[i]); }
        const bool trace = std::getenv("PHQ_TRANSCODE_TRACE") != nullptr;
        auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        double t1 = 0, t2 = 0;
        {
            phq::Transcoder transcoder(set->impl, phq::TranscodePlan::explicit_files(set->impl.job(), input, sam_output_path), batch_capacity);
            t1 = now();
            transcoder.run(*incoming_count, *incoming_pf_count);
            t2 = now();
        }
        if(trace) { std::fprintf(stderr, "transcode call (s): set up %.3f run %.3f tear down %.3f\n", t1 - t0, t2 - t1, now() - t2); }
    });
}

int phq_transcode_job(phq_decoder_set* set, int64_t batch_capacity, uint64_t* incoming_count, uint64_t* incoming_pf_count) {
    return guarded(set, [&]() {
        require(set, "set"); require(incoming_count, "incoming_count"); require(incoming_pf_count, "incoming_pf_count");
        if(batch_capacity <= 0) { throw phq::InternalError("batch capacity must be positive"); }
        phq::Transcoder transcoder(set->impl, phq::TranscodePlan::from_job(set->impl.job()), batch_capacity);
        transcoder.run(*incoming_count, *incoming_pf_count);
    });
}

int phq_transcode_job_copies(phq_decoder_set* set, int32_t n_copy, phq_decoder_set* const* copy, int64_t batch_capacity, uint64_t* incoming_count, uint64_t* incoming_pf_count) {
    return guarded(set, [&]() {
        require(set, "set"); require(incoming_count, "incoming_count"); require(incoming_pf_count, "incoming_pf_count");
        if(batch_capacity <= 0) { throw phq::InternalError("batch capacity must be positive"); }
        if(n_copy < 0 || (n_copy > 0 && copy == nullptr)) { throw phq::InternalError("copies missing"); }
        phq::Transcoder transcoder(set->impl, phq::TranscodePlan::from_job(set->impl.job()), batch_capacity);
        for(int32_t k = 0; k < n_copy; ++k) { require(copy[k], "copy"); transcoder.add_copy(copy[k]->impl); }
        transcoder.run(*incoming_count, *incoming_pf_count);
    });
}

int phq_host_alloc(uint64_t bytes, void** pointer) {
    return guarded(nullptr, [&]() { require(pointer, "pointer"); *pointer = phq::host_alloc(static_cast<size_t>(bytes)); });
}

void phq_host_free(void* pointer) { phq::host_free(pointer); }

int phq_compile_job(const char* job_json, char** compiled_json) {
    if(compiled_json != nullptr) { *compiled_json = nullptr; }
    return guarded(nullptr, [&]() {
        require(job_json, "job_json");
        require(compiled_json, "compiled_json");
        const phq::Json document = phq::Json::parse(job_json);
        const std::string text = (phq::is_compiled_job(document) ? document : phq::compile_job(document)).dump(324);   // no truncation: round trip digits
        char* out = static_cast<char*>(std::malloc(text.size() + 1));
        if(out == nullptr) { throw std::bad_alloc(); }
        std::memcpy(out, text.c_str(), text.size() + 1);
        *compiled_json = out;
    });
}

int phq_load_job(const char* path, char** job_json) {
    if(job_json != nullptr) { *job_json = nullptr; }
    return guarded(nullptr, [&]() {
        require(path, "path");
        require(job_json, "job_json");
        const std::string text = phq::load_job_document(path).dump(324);
        char* out = static_cast<char*>(std::malloc(text.size() + 1));
        if(out == nullptr) { throw std::bad_alloc(); }
        std::memcpy(out, text.c_str(), text.size() + 1);
        *job_json = out;
    });
}

void phq_free_string(char* text) { std::free(text); }

int phq_transcode_plan(phq_decoder_set* set, char** plan_json) {
    if(plan_json != nullptr) { *plan_json = nullptr; }
    return guarded(set, [&]() {
        require(set, "set"); require(plan_json, "plan_json");
        const phq::TranscodePlan plan = phq::TranscodePlan::from_job(set->impl.job());
        phq::Json document = phq::Json::object();
        phq::Json input_feed = phq::Json::array();
        for(size_t f = 0; f < plan.input_feed.size(); ++f) {
            phq::Json feed = phq::Json::object();
            feed.set("index", phq::Json::integer(static_cast<int64_t>(f)));
            feed.set("url", phq::Json::string(plan.input_feed[f].path));
            feed.set("resolution", phq::Json::integer(plan.input_feed[f].resolution));
            input_feed.push(feed);
        }
        phq::Json by_segment = phq::Json::array();
        for(const auto& where : plan.input_segment) {
            phq::Json pair = phq::Json::array();
            pair.push(phq::Json::integer(where.first));
            pair.push(phq::Json::integer(where.second));
            by_segment.push(pair);
        }
        phq::Json output_feed = phq::Json::array();
        for(size_t f = 0; f < plan.output_feed.size(); ++f) {
            const auto& from = plan.output_feed[f];
            phq::Json feed = phq::Json::object();
            feed.set("index", phq::Json::integer(static_cast<int64_t>(f)));
            feed.set("url", phq::Json::string(from.path));
            feed.set("format", phq::Json::string(from.format == phq::FileFormat::FASTQ ? "fastq" : "sam"));
            feed.set("compression", phq::Json::string(from.gzip ? "gz" : "none"));
            feed.set("compression level", phq::Json::integer(from.level));
            feed.set("resolution", phq::Json::integer(from.resolution));
            output_feed.push(feed);
        }
        phq::Json channel = phq::Json::array();
        for(size_t c = 0; c < plan.channel.size(); ++c) {
            phq::Json record = phq::Json::object();
            record.set("index", phq::Json::integer(static_cast<int64_t>(c)));
            phq::Json feeds = phq::Json::array();
            for(const int32_t f : plan.channel[c].feed_by_segment) { feeds.push(phq::Json::integer(f)); }
            record.set("output feed by segment", feeds);
            record.set("filter outgoing qc fail", phq::Json::boolean(plan.channel[c].filter_outgoing_qc_fail));
            channel.push(record);
        }
        document.set("input feed", input_feed);
        document.set("input feed by segment", by_segment);
        document.set("output feed", output_feed);
        document.set("channel", channel);
        const std::string text = document.dump(15);
        char* out = static_cast<char*>(std::malloc(text.size() + 1));
        if(out == nullptr) { throw std::bad_alloc(); }
        std::memcpy(out, text.c_str(), text.size() + 1);
        *plan_json = out;
    });
}

int phq_adjust_prior(phq_decoder_set* set, const uint64_t* host_words, const char* job_json, char** adjusted_json) {
    if(adjusted_json != nullptr) { *adjusted_json = nullptr; }
    return guarded(set, [&]() {
        require(set, "set"); require(host_words, "host_words"); require(job_json, "job_json"); require(adjusted_json, "adjusted_json");
        const phq::JobConfig& job = set->impl.job();
        const std::string text = phq::adjust_prior(job, host_words, phq::assemble_job(phq::Json::parse(job_json))).dump(job.float_precision);
        char* out = static_cast<char*>(std::malloc(text.size() + 1));
        if(out == nullptr) { throw std::bad_alloc(); }
        std::memcpy(out, text.c_str(), text.size() + 1);
        *adjusted_json = out;
    });
}

int phq_format_double(double value, int32_t precision, char* buffer, int32_t capacity) {
    return guarded(nullptr, [&]() {
        require(buffer, "buffer");
        const std::string text = phq::format_double(value, precision);
        if(static_cast<int32_t>(text.size()) + 1 > capacity) { throw phq::OverflowError("buffer too small"); }
        std::memcpy(buffer, text.c_str(), text.size() + 1);
    });
}

int phq_qc_export(phq_decoder_set* set, uint64_t* words, int64_t capacity_words, int64_t* n_words) {
    return guarded(set, [&]() {
        require(set, "set"); require(n_words, "n_words");
        const std::vector<uint64_t> image = set->impl.quality_control().to_words();
        *n_words = static_cast<int64_t>(image.size());
        if(words == nullptr) { return; }
        if(capacity_words < *n_words) { throw phq::OverflowError("buffer too small for the quality control image"); }
        std::memcpy(words, image.data(), image.size() * sizeof(uint64_t));
    });
}

int phq_qc_merge(phq_decoder_set* set, const uint64_t* words, int64_t n_words) {
    return guarded(set, [&]() {
        require(set, "set"); require(words, "words");
        if(n_words < 2) { throw phq::InternalError("quality control image too short"); }
        set->impl.quality_control_merge(phq::QcState::from_words(words, static_cast<size_t>(n_words)));
    });
}

const char* phq_build_id(void) {
    static const char id[] =
#include "build/build_id.inc"
        ;
    return id;
}

}  // extern "C"
