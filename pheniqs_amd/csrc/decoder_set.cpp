#include "decoder_set.hpp"
#include "unpack.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <limits>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "compile.hpp"
#include "host_parallel.hpp"
#include "error.hpp"
#include "phred.hpp"
#include "stats.hpp"

namespace phq {

// pageable caller buffers bounce through pinned ones: the copy runs on the host threads (one thread moves ~10 GB/s,
// a third of what the PCIe link takes)
static void host_copy(void* to, const void* from, size_t bytes) {
    if(bytes < (size_t(1) << 22)) { std::memcpy(to, from, bytes); return; }
    parallel_ranges(static_cast<int64_t>(bytes), int64_t(1) << 21, host_threads(), [&](int64_t first, int64_t last, int) {
        std::memcpy(static_cast<char*>(to) + first, static_cast<const char*>(from) + first, static_cast<size_t>(last - first));
    });
}

namespace {

void check(hipError_t status, const char* what) {
    if(status == hipSuccess) { return; }
    std::string message = std::string(what) + " : " + hipGetErrorString(status);
    if(status == hipErrorOutOfMemory) { throw OutOfMemoryError(message); }
    throw InternalError(message);
}

}  // namespace

// hipEventRecord right before and right after each kernel launch, on the launch stream
struct DecoderSet::EventTimer : LaunchTimer {
    struct Span { int32_t kind; hipEvent_t start; hipEvent_t stop; int32_t decoder; const char* name; };
    std::vector<Span> spans;
    void begin(int32_t kind, void* stream, const char* name) override {
        Span span{kind, nullptr, nullptr, decoder, name};
        check(hipEventCreate(&span.start), "hipEventCreate");
        check(hipEventCreate(&span.stop), "hipEventCreate");
        check(hipEventRecord(span.start, static_cast<hipStream_t>(stream)), "hipEventRecord");
        spans.push_back(span);
    }
    void end(void* stream) override {
        check(hipEventRecord(spans.back().stop, static_cast<hipStream_t>(stream)), "hipEventRecord");
    }
    void collect(double* milliseconds, int64_t* launches) {
        for(int k = 0; k < KERNEL_KINDS; ++k) { milliseconds[k] = 0; launches[k] = 0; }
        for(auto& span : spans) {
            check(hipEventSynchronize(span.stop), "hipEventSynchronize");
            float ms = 0;
            check(hipEventElapsedTime(&ms, span.start, span.stop), "hipEventElapsedTime");
            milliseconds[span.kind] += ms;
            launches[span.kind] += 1;
            (void)hipEventDestroy(span.start);
            (void)hipEventDestroy(span.stop);
        }
        spans.clear();
    }
    // every span since the last collect, in launch order: [{"decoder", "kernel", "kind", "ms"}]; forgets them
    std::string chain() {
        std::string out = "[";
        char number[64];
        for(size_t k = 0; k < spans.size(); ++k) {
            auto& span = spans[k];
            check(hipEventSynchronize(span.stop), "hipEventSynchronize");
            float ms = 0;
            check(hipEventElapsedTime(&ms, span.start, span.stop), "hipEventElapsedTime");
            std::snprintf(number, sizeof(number), "%.6f", static_cast<double>(ms));
            out += std::string(k ? ", " : "") + "{\"decoder\": " + std::to_string(span.decoder) + ", \"kernel\": \"" + span.name + "\", \"kind\": " + std::to_string(span.kind) + ", \"ms\": " + number + "}";
            (void)hipEventDestroy(span.start);
            (void)hipEventDestroy(span.stop);
        }
        spans.clear();
        return out + "]";
    }
    ~EventTimer() override {
        for(auto& span : spans) { (void)hipEventDestroy(span.start); (void)hipEventDestroy(span.stop); }
    }
};

namespace {

// bins whose counters fit a workgroup histogram (12 words of 8 bytes per bin) next to the Phred table
constexpr int LDS_HISTOGRAM_MAX_BINS = 340;          // 340 * 96 B = 32 KiB

}  // namespace

void* host_alloc(size_t bytes) {
    void* pointer = nullptr;
    check(hipHostMalloc(&pointer, bytes, hipHostMallocDefault), "hipHostMalloc");
    return pointer;
}
void host_free(void* pointer) {
    if(pointer != nullptr) { (void)hipHostFree(pointer); }
}

// a job document that did not go through the compiler yet (decoder repository, "base" references, no cardinalities)
// is compiled first, so a user level Pheniqs configuration runs unchanged
static Json compiled_document(const std::string& text) {
    Json document = Json::parse(text);
    return is_compiled_job(document) ? document : compile_job(document);
}

DecoderSet::DecoderSet(const std::string& compiled_job_json, int32_t device) :
    job_(JobConfig::from_json(compiled_document(compiled_job_json))),
    device_(device),
    mode_(MODE_AUTO),
    max_blocks_(2048),
    device_decoder_(nullptr),
    device_stats_(nullptr),
    device_worklist_(nullptr),
    worklist_capacity_(0),
    timer_(nullptr),
    chunk_capacity_(0) {
    if(device_ >= 0) {
        try {
            build_device_images();
        } catch(...) {
            release();
            throw;
        }
    }
}

DecoderSet::~DecoderSet() {
    delete timer_;
    timer_ = nullptr;
    release();
}

void DecoderSet::profile_enable(bool enable) {
    require_device();
    if(enable && timer_ == nullptr) { timer_ = new EventTimer(); }
    if(!enable) { delete timer_; timer_ = nullptr; }
}

std::string DecoderSet::profile_chain() {
    require_device();
    if(timer_ == nullptr) { throw InternalError("profiling is not enabled"); }
    return timer_->chain();
}

void DecoderSet::profile_collect(double* milliseconds, int64_t* launches) {
    require_device();
    if(timer_ == nullptr) { throw InternalError("profiling is not enabled"); }
    timer_->collect(milliseconds, launches);
}

void DecoderSet::release() {
    if(device_ >= 0) { (void)hipSetDevice(device_); }
    for(auto& slot : slot_) {
        if(slot.stream) { (void)hipStreamSynchronize(static_cast<hipStream_t>(slot.stream)); }
        (void)hipFree(slot.device_base); (void)hipFree(slot.device_quality); (void)hipFree(slot.device_flag);
        (void)hipFree(slot.device_result); (void)hipFree(slot.device_word); (void)hipFree(slot.device_worklist);
        (void)hipFree(slot.device_packed); (void)hipFree(slot.device_escape); host_free(slot.pinned_escape);
        host_free(slot.pinned_base); host_free(slot.pinned_quality); host_free(slot.pinned_flag);
        host_free(slot.pinned_result); host_free(slot.pinned_word);
        if(slot.done) { (void)hipEventDestroy(static_cast<hipEvent_t>(slot.done)); }
        if(slot.uploaded) { (void)hipEventDestroy(static_cast<hipEvent_t>(slot.uploaded)); }
        if(slot.consumed) { (void)hipEventDestroy(static_cast<hipEvent_t>(slot.consumed)); }
        if(slot.stream) { (void)hipStreamDestroy(static_cast<hipStream_t>(slot.stream)); }
        slot = Slot();
    }
    if(upload_stream_) { (void)hipStreamDestroy(static_cast<hipStream_t>(upload_stream_)); upload_stream_ = nullptr; }
    release_host_stage();
    if(order_event_) { (void)hipEventDestroy(static_cast<hipEvent_t>(order_event_)); order_event_ = nullptr; order_pending_ = false; }
    for(void* p : device_allocation_) { (void)hipFree(p); }
    device_allocation_.clear();
    for(void* p : host_image_) { std::free(p); }
    host_image_.clear();
    (void)hipFree(device_worklist_);
    device_worklist_ = nullptr;
    worklist_capacity_ = 0;
    device_decoder_ = nullptr;
    device_stats_ = nullptr;
}

void DecoderSet::require_device() const {
    if(device_ < 0) { throw InternalError("this decoder set was created without a device (PHQ_DEVICE_NONE)"); }
    check(hipSetDevice(device_), "hipSetDevice");
}

void DecoderSet::set_mode(int32_t mode) {
    if(mode != MODE_LITERAL && mode != MODE_FAST && mode != MODE_AUTO) { throw ConfigurationError("unknown classification mode " + std::to_string(mode)); }
    mode_ = mode;
}

// (barcode, position) cells a read costs the literal kernel, over the job's PAMLD decoders (SURVEY.md §8d's secondary figure)
int64_t DecoderSet::literal_cells_per_read() const {
    int64_t cells = 0;
    for(const auto& c : job_.decoder) {
        if(c.algorithm == Algorithm::PAMLD) { cells += static_cast<int64_t>(c.codec.size()) * static_cast<int64_t>(c.nucleotide_cardinality); }
    }
    return cells;
}

// The tables of the neighbour list kernels (pamld_near.hpp) for N words of L nucleotides given as code bit planes (bit j =
// position j): the word -> barcode table over every word at most one mismatch from one of them, and per word the bit
// planes of the words nearest to it, what bounds the others, and `tag` (what the kernel reports for the word).
// the barcodes of an ACGT codec as two bit planes per barcode: bit j of low / high = the low / high bit of the 2-bit code
// (A, C, G, T = 0..3) of position j — the key format of the lookup tables of the first tiers
void code_planes(const DecoderConfig& c, std::vector<uint32_t>& low, std::vector<uint32_t>& high) {
    const size_t N = c.codec.size();
    const size_t L = static_cast<size_t>(c.nucleotide_cardinality);
    low.assign(N, 0u);
    high.assign(N, 0u);
    for(size_t b = 0; b < N; ++b) {
        for(size_t j = 0; j < L; ++j) {
            const uint32_t code = c.barcode_code[b * L + j];
            const uint32_t two = code == 1 ? 0u : code == 2 ? 1u : code == 4 ? 2u : 3u;
            low[b] |= (two & 1u) << j;
            high[b] |= (two >> 1) << j;
        }
    }
}

struct NearTables {
    std::vector<uint16_t> table;
    std::vector<uint32_t> record;
    std::vector<int32_t> listed; // near_list(L) ordinals per word, in the order of its record (the word itself first)
    int32_t usable = 0;          // words whose radius is at least 3
};
// prior: the barcodes' priors when they differ (then `noise` = the prior adjusted noise probability, pamld.cpp:29), else null
NearTables build_near_tables(const std::vector<uint32_t>& low, const std::vector<uint32_t>& high, int32_t L, const std::vector<uint32_t>& tag,
                             const std::vector<double>* prior = nullptr, double noise = 0) {
    NearTables out;
    const int32_t N = static_cast<int32_t>(low.size());
    const int32_t list = near_list(L);
    const int32_t field = near_field_bits(L);
    const int32_t per_word = 32 / field;
    const int32_t record_words = near_record_words(L, prior != nullptr);
    const int32_t weight_at = near_record_words(L);
    auto distance = [&](int32_t a, int32_t b) {
        return __builtin_popcount((low[static_cast<size_t>(a)] ^ low[static_cast<size_t>(b)]) | (high[static_cast<size_t>(a)] ^ high[static_cast<size_t>(b)]));
    };
    // a read more than one mismatch away from every word is not finished by the kernels: only those words
    out.table.assign(static_cast<size_t>(1) << (2 * L), NEAR_NO_BARCODE);
    for(int32_t b = N - 1; b >= 0; --b) {                     // the lowest ordinal last: it wins a shared word
        for(int32_t j = 0; j < L; ++j) {
            for(uint32_t other = 0; other < 4; ++other) {
                const uint32_t l = (low[static_cast<size_t>(b)] & ~(1u << j)) | ((other & 1u) << j);
                const uint32_t h = (high[static_cast<size_t>(b)] & ~(1u << j)) | ((other >> 1) << j);
                uint16_t& slot = out.table[l | (h << L)];
                const bool is_exact = l == low[static_cast<size_t>(b)] && h == high[static_cast<size_t>(b)];
                // an exact word is never taken over by a neighbour of another word
                if(slot == NEAR_NO_BARCODE || is_exact || !(low[slot] == l && high[slot] == h)) { slot = static_cast<uint16_t>(b); }
            }
        }
    }
    if(N < list) { return out; }                             // the table alone (mdd_near.hpp): no lists of a set this small (N == list: every word is listed, nothing left to bound)
    out.record.assign(static_cast<size_t>(N) * static_cast<size_t>(record_words), 0u);
    out.listed.assign(static_cast<size_t>(N) * static_cast<size_t>(list), 0);
    std::vector<std::pair<int, int32_t>> order(static_cast<size_t>(N));
    for(int32_t b = 0; b < N; ++b) {
        for(int32_t o = 0; o < N; ++o) { order[static_cast<size_t>(o)] = {o == b ? -1 : distance(b, o), o}; }
        std::sort(order.begin(), order.end());               // b itself first, then by (distance, ordinal)
        uint32_t* record = &out.record[static_cast<size_t>(b) * static_cast<size_t>(record_words)];
        for(int32_t k = 0; k < list; ++k) {
            const int32_t o = order[static_cast<size_t>(k)].second;
            record[2 * (k / per_word)] |= low[static_cast<size_t>(o)] << (field * (k % per_word));
            record[2 * (k / per_word) + 1] |= high[static_cast<size_t>(o)] << (field * (k % per_word));
            out.listed[static_cast<size_t>(b) * static_cast<size_t>(list) + static_cast<size_t>(k)] = o;
        }
        // the first word not listed sets the radius, at most 5 (the kernels form their bounds for 4 and 5)
        const int radius = N > list ? std::min(order[static_cast<size_t>(list)].first, 5) : 5;
        // what is not listed, at the radius and beyond it: how many — or, with priors that differ, their priors over b's
        double at_radius = 0;
        double beyond = 0;
        const double own = prior ? (*prior)[static_cast<size_t>(b)] : 1.0;
        for(int32_t k = list; k < N; ++k) {
            const double weight = prior ? (*prior)[static_cast<size_t>(order[static_cast<size_t>(k)].second)] / own : 1.0;
            if(order[static_cast<size_t>(k)].first <= radius) { at_radius += weight; } else { beyond += weight; }
        }
        const float up = prior ? 1.0f + 1e-6f : 1.0f;          // the sums bound: rounded up
        const float cnt_r = static_cast<float>(at_radius) * up;
        const float cnt_rest = static_cast<float>(beyond) * up;
        if(prior) {
            for(int32_t k = 0; k < list; ++k) {
                const float weight = k == 0 ? static_cast<float>(noise / own)
                                            : static_cast<float>((*prior)[static_cast<size_t>(order[static_cast<size_t>(k)].second)] / own);
                std::memcpy(&record[weight_at + k], &weight, sizeof(float));
            }
        }
        uint32_t* meta = record + 2 * (list / per_word);
        meta[0] = static_cast<uint32_t>(radius);
        std::memcpy(&meta[1], &cnt_r, sizeof(float));
        std::memcpy(&meta[2], &cnt_rest, sizeof(float));
        meta[3] = tag[static_cast<size_t>(b)];
        if(radius >= 3) { ++out.usable; }                     // at 3 only the reads ON the word are finished by the kernels
    }
    return out;
}

void DecoderSet::build_device_images() {
    // a set without a device (plan_chain): the same tables, in host memory — the dispatcher's predicates ask which of them
    // there are; an MI355X's 256 CUs for the grids
    const bool host_only = device_ < 0;
    if(host_only) { max_blocks_ = 256 * GRID_BLOCKS_PER_CU; }
    else {
        int count = 0;
        check(hipGetDeviceCount(&count), "hipGetDeviceCount");
        if(device_ >= count) { throw ConfigurationError("no HIP device " + std::to_string(device_)); }
        check(hipSetDevice(device_), "hipSetDevice");
        hipDeviceProp_t properties;
        check(hipGetDeviceProperties(&properties, device_), "hipGetDeviceProperties");
        // the largest grid of any stage kernel, 16 workgroups per CU; stage.hip picks each kernel's own share of it
        // (well past the 256 CUs either way, guide §6 G11)
        max_blocks_ = properties.multiProcessorCount * GRID_BLOCKS_PER_CU;
    }

    auto upload = [&](const void* source, size_t bytes) -> void* {
        if(host_only) {
            void* const image = std::malloc(bytes > 0 ? bytes : 16);
            if(image == nullptr) { throw std::bad_alloc(); }
            host_image_.push_back(image);
            if(bytes > 0) { std::memcpy(image, source, bytes); }
            return image;
        }
        void* pointer = nullptr;
        check(hipMalloc(&pointer, bytes > 0 ? bytes : 16), "hipMalloc");
        device_allocation_.push_back(pointer);
        if(bytes > 0) { check(hipMemcpy(pointer, source, bytes, hipMemcpyHostToDevice), "hipMemcpy"); }
        return pointer;
    };

    const PhredScale& scale = PhredScale::get_instance();
    const double* device_tpq = static_cast<const double*>(upload(scale.true_positive_quality, sizeof(scale.true_positive_quality)));
    const double* device_match = static_cast<const double*>(upload(scale.match_probability, sizeof(scale.match_probability)));
    const double* device_mismatch = static_cast<const double*>(upload(scale.mismatch_probability, sizeof(scale.mismatch_probability)));

    std::vector<double> factor_pairs(129 * 2);
    for(int k = 0; k < 129; ++k) {
        factor_pairs[static_cast<size_t>(2 * k)] = k < 128 ? scale.match_probability[k] : scale.uniform_probability;
        factor_pairs[static_cast<size_t>(2 * k + 1)] = k < 128 ? scale.mismatch_probability[k] : scale.uniform_probability;
    }
    const double* device_factor_pairs = static_cast<const double*>(upload(factor_pairs.data(), factor_pairs.size() * sizeof(double)));

    host_decoder_.assign(job_.decoder.size(), DevDecoder());
    for(size_t k = 0; k < job_.decoder.size(); ++k) {
        const DecoderConfig& c = job_.decoder[k];
        DevDecoder& d = host_decoder_[k];
        std::memset(&d, 0, sizeof(d));
        d.algorithm = static_cast<int32_t>(c.algorithm);
        d.kind = static_cast<int32_t>(c.kind);
        d.n_barcode = static_cast<int32_t>(c.codec.size());
        d.nucleotide_cardinality = c.nucleotide_cardinality;
        d.n_segment = c.segment_cardinality;
        d.n_word = (c.nucleotide_cardinality + 7) / 8;
        d.plane_stride = c.plane_stride;
        d.result_slot = c.result_slot;
        d.multiplexing = c.multiplexing_classifier ? 1 : 0;
        d.all_strict = c.all_strict ? 1 : 0;
        d.high_quality_threshold = c.high_quality_threshold;
        d.high_quality_distance_threshold = c.high_quality_distance_threshold;
        d.quality_masking_threshold = c.quality_masking_threshold;
        d.use_lds_histogram = (d.n_barcode + 1) <= LDS_HISTOGRAM_MAX_BINS ? 1 : 0;
        d.adjusted_noise_probability = c.noise * c.random_barcode_probability;        // pamld.cpp:29
        d.confidence_threshold = c.confidence_threshold;
        d.random_barcode_probability = c.random_barcode_probability;
        d.uniform_base_quality = scale.uniform_base_quality;
        d.log2_base_hi = scale.log2_base_hi;
        d.log2_base_lo = scale.log2_base_lo;
        d.stats_offset = c.stats_offset;
        d.true_positive_quality = device_tpq;
        d.match_probability = device_match;
        d.mismatch_probability = device_mismatch;
        d.uniform_probability = scale.uniform_probability;
        d.factor_pairs = device_factor_pairs;
        if(c.closed()) {
            for(int32_t s = 0; s < c.segment_cardinality; ++s) {
                d.distance_tolerance[s] = s < static_cast<int32_t>(c.distance_tolerance.size()) ? c.distance_tolerance[static_cast<size_t>(s)] : 0;
                const int32_t begin = c.segment_offset[static_cast<size_t>(s)];
                const int32_t end = begin + c.barcode_length[static_cast<size_t>(s)];
                for(int32_t j = begin; j < end; ++j) {
                    d.segment_mask[s][j >> 3] |= 1u << (4 * (j & 7));
                    d.full_mask[j >> 3] |= 1u << (4 * (j & 7));
                }
                if(end > begin) { d.segment_end_bits[(end - 1) >> 5] |= 1u << ((end - 1) & 31); }
            }
            std::vector<uint32_t> packed(static_cast<size_t>(d.n_barcode) * static_cast<size_t>(d.n_word), 0);
            std::vector<double> concentration(static_cast<size_t>(d.n_barcode));
            std::vector<int32_t> index(static_cast<size_t>(d.n_barcode));
            for(int32_t b = 0; b < d.n_barcode; ++b) {
                for(int32_t j = 0; j < c.nucleotide_cardinality; ++j) {
                    const uint32_t code = c.barcode_code[static_cast<size_t>(b) * static_cast<size_t>(c.nucleotide_cardinality) + static_cast<size_t>(j)];
                    packed[static_cast<size_t>(b) * static_cast<size_t>(d.n_word) + static_cast<size_t>(j >> 3)] |= (code & 0xfu) << (4 * (j & 7));
                }
                concentration[static_cast<size_t>(b)] = c.codec[static_cast<size_t>(b)].concentration;
                index[static_cast<size_t>(b)] = c.codec[static_cast<size_t>(b)].index;
            }
            d.uniform_prior = 1;
            for(double c0 : concentration) { if(c0 != concentration[0]) { d.uniform_prior = 0; } }
            d.identity_index = 1;
            for(int32_t b = 0; b < d.n_barcode; ++b) { if(index[static_cast<size_t>(b)] != b + 1) { d.identity_index = 0; } }
            // The product form PAMLD kernel (pamld_lut.hpp) lives on two workgroups per CU, and its per read tables
            // leave little LDS: a workgroup histogram that would cost it a workgroup is not worth it — the counters of
            // classified reads are then rebuilt from the result records (count_results_kernel), as for large codecs.
            // Measured at 192 barcodes of 8 nt: 12.1 -> 13.7 Greads/s.
            if(c.algorithm == Algorithm::PAMLD && c.all_strict && c.nucleotide_cardinality <= 16 && d.identity_index && d.use_lds_histogram) {
                const size_t groups = static_cast<size_t>(c.nucleotide_cardinality + 3) / 4;
                const size_t threads = c.nucleotide_cardinality <= 8 ? 512 : 256;
                const size_t padded = static_cast<size_t>((d.n_barcode + 7) & ~7);
                const size_t fixed = (threads / 64) * groups * 16 * 64 * 4 + 3120 + 16 + (c.nucleotide_cardinality <= 8 ? padded / 2 : padded) * 8;
                const size_t with_histogram = fixed + (STAT_JOB_WORDS + static_cast<size_t>(d.n_barcode + 1) * STAT_WORDS_PER_BIN) * 8;
                const size_t without = fixed + (STAT_JOB_WORDS + STAT_WORDS_PER_BIN) * 8;
                constexpr size_t LDS_PER_CU = 160 * 1024;
                // ... when it is the second workgroup that is lost; a third gains less than the extra pass costs (C4's 10 nt
                // decoder: 4.56 -> 4.41 Greads/s)
                if(LDS_PER_CU / with_histogram < 2 && LDS_PER_CU / without >= 2) { d.use_lds_histogram = 0; }
            }
            if(!d.use_lds_histogram && d.identity_index && d.n_barcode + 1 <= MAX_DEFERRED_BINS) {
                // count_results_kernel's staging array (counters.hip), zero between launches
                const std::vector<unsigned long long> zeros(static_cast<size_t>(d.n_barcode + 1) * 4, 0ull);
                d.count_staging = static_cast<unsigned long long*>(upload(zeros.data(), zeros.size() * sizeof(unsigned long long)));
            }
            d.barcode_packed = static_cast<const uint32_t*>(upload(packed.data(), packed.size() * sizeof(uint32_t)));
            d.concentration = static_cast<const double*>(upload(concentration.data(), concentration.size() * sizeof(double)));
            d.barcode_index = static_cast<const int32_t*>(upload(index.data(), index.size() * sizeof(int32_t)));
            if(c.all_strict && c.algorithm == Algorithm::MDD && d.n_barcode > 0 && c.nucleotide_cardinality <= DEV_TIER_NUCLEOTIDE) {
                // exact match hash (mdd.cpp:29-31,43-46): keys are the full length barcodes, 2 bits per base
                uint32_t slots = 16;
                while(slots < 4u * static_cast<uint32_t>(d.n_barcode)) { slots <<= 1; }
                std::vector<uint32_t> table(static_cast<size_t>(slots) * 4, 0u);
                int32_t longest = 1;
                for(int32_t b = 0; b < d.n_barcode; ++b) {
                    uint32_t key[2] = {0, 0};
                    for(int32_t j = 0; j < c.nucleotide_cardinality; ++j) {
                        const uint32_t code = c.barcode_code[static_cast<size_t>(b) * static_cast<size_t>(c.nucleotide_cardinality) + static_cast<size_t>(j)];
                        const uint32_t two = code == 1 ? 0u : code == 2 ? 1u : code == 4 ? 2u : 3u;
                        key[j >> 4] |= two << (2 * (j & 15));
                    }
                    uint32_t h = (key[0] * 0x9e3779b1u) ^ (key[1] * 0x85ebca77u);
                    h ^= h >> 15;
                    int32_t probe = 0;
                    while(table[static_cast<size_t>((h + static_cast<uint32_t>(probe)) & (slots - 1)) * 4 + 2] != 0u) { ++probe; }
                    uint32_t* slot = &table[static_cast<size_t>((h + static_cast<uint32_t>(probe)) & (slots - 1)) * 4];
                    slot[0] = key[0]; slot[1] = key[1]; slot[2] = static_cast<uint32_t>(b + 1);
                    slot[3] = static_cast<uint32_t>(index[static_cast<size_t>(b)]);       // Classifier index, saves the kernel a gather
                    longest = std::max(longest, probe + 1);
                }
                d.exact_table = static_cast<const uint32_t*>(upload(table.data(), table.size() * sizeof(uint32_t)));
                d.exact_slots = slots;
                d.exact_probes = longest + 1;
            }
            if(c.all_strict && c.nucleotide_cardinality <= 16 && d.n_barcode > 0) {
                const int32_t padded = (d.n_barcode + 7) / 8 * 8;        // whole blocks of 8 barcodes (pamld_lut.hpp)
                const bool paired = c.nucleotide_cardinality <= 8;
                std::vector<uint32_t> packed2(static_cast<size_t>(paired ? padded / 2 : padded), 0);
                std::vector<float> prior(static_cast<size_t>(padded), 0.0f);
                for(int32_t b = 0; b < padded; ++b) {
                    const int32_t source = b < d.n_barcode ? b : 0;          // padding repeats barcode 0; its probability is zeroed
                    uint32_t word = 0;
                    for(int32_t j = 0; j < c.nucleotide_cardinality; ++j) {
                        const uint32_t code = c.barcode_code[static_cast<size_t>(source) * static_cast<size_t>(c.nucleotide_cardinality) + static_cast<size_t>(j)];
                        const uint32_t two = code == 1 ? 0u : code == 2 ? 1u : code == 4 ? 2u : 3u;
                        // bit-sliced: byte (2 * (b odd) + j / 4) of the word, low bit at 2 + j % 4, high bit 4 places up (mod 32)
                        const uint32_t byte = (paired ? 2u * static_cast<uint32_t>(b & 1) : 0u) + static_cast<uint32_t>(j >> 2);
                        const uint32_t low = 8u * byte + 2u + static_cast<uint32_t>(j & 3);
                        word |= (two & 1u) << low;
                        word |= (two >> 1) << ((low + 4u) & 31u);
                    }
                    if(paired) { packed2[static_cast<size_t>(b >> 1)] |= word; }
                    else { packed2[static_cast<size_t>(b)] = word; }
                    if(b < d.n_barcode) { prior[static_cast<size_t>(b)] = static_cast<float>(concentration[static_cast<size_t>(b)]); }
                }
                d.barcode_packed2 = static_cast<const uint32_t*>(upload(packed2.data(), packed2.size() * sizeof(uint32_t)));
                d.concentration_single = static_cast<const float*>(upload(prior.data(), prior.size() * sizeof(float)));
            }
            // ratios of priors are formed below; a barcode whose own prior is zero (nothing seen of it by an estimation pass)
            // gets infinite ones: the kernels' certificates fail on them and a read that names it goes to the next tier
            bool positive_priors = true;
            for(double c0 : concentration) { if(!(c0 >= 0) || !std::isfinite(c0)) { positive_priors = false; } }
            if(c.all_strict && c.algorithm == Algorithm::PAMLD && (d.uniform_prior || positive_priors) && c.nucleotide_cardinality >= NEAR_MIN_LENGTH && c.nucleotide_cardinality <= NEAR_MAX_LENGTH &&
               d.n_barcode > near_list(c.nucleotide_cardinality) && d.n_barcode <= NEAR_MAX_BARCODES) {
                // neighbour list tier (pamld_near.hpp): the barcode of every word at most one mismatch away from one, and
                // per barcode the bit planes of the barcodes nearest to it with what bounds the others
                const int32_t L = c.nucleotide_cardinality;
                const int32_t N = d.n_barcode;
                std::vector<uint32_t> low, high;
                code_planes(c, low, high);
                std::vector<uint32_t> tag(static_cast<size_t>(N));
                for(int32_t b = 0; b < N; ++b) { tag[static_cast<size_t>(b)] = static_cast<uint32_t>(index[static_cast<size_t>(b)]); }
                const NearTables tables = build_near_tables(low, high, L, tag, d.uniform_prior ? nullptr : &concentration, d.adjusted_noise_probability);
                d.near_table = static_cast<const uint16_t*>(upload(tables.table.data(), tables.table.size() * sizeof(uint16_t)));
                d.near_record = static_cast<const uint32_t*>(upload(tables.record.data(), tables.record.size() * sizeof(uint32_t)));
                d.near_ready = 2 * tables.usable >= N ? 1 : 0;          // a denser set leaves the kernel nothing to finish
                const bool stream_priors = !d.uniform_prior;
                // a decoder without a workgroup histogram streams with blocks that hold no bin (its counters come from the records)
                const bool stream_bins = d.use_lds_histogram != 0;
                d.stream_binless = stream_bins ? 0 : 1;
                const uint32_t stream_stride = static_cast<uint32_t>(stream_record_words(stream_priors, stream_bins)) * 4u;
                if(!tables.record.empty() && N < static_cast<int32_t>(STREAM_UNKNOWN) &&
                   stream_block_at(STREAM_WAVES, N, stream_priors, L) + static_cast<uint32_t>(N) * stream_stride < (1u << 18)) {
                    // the streaming kernel (pamld_stream.hpp): the same table and lists in pamld_lut.hpp's bit order; the table
                    // names a barcode by the LDS address of its block and its index
                    const uint32_t first_block = stream_block_at(STREAM_WAVES, N, stream_priors, L);
                    const int32_t near_listed = near_list(L);                 // tables.listed holds that many per barcode, nearest first
                    const int32_t list = stream_list(L);                      // 16 up to 8 nt, two to a plane word; 8 beyond, one to a word
                    const bool one_per_word = L > 8;
                    const int32_t record_words = stream_record_words(stream_priors, stream_bins);
                    auto half = [](uint32_t bits) { return ((bits & 0xfu) << 2) | (((bits >> 4) & 0xfu) << 10) | (((bits >> 8) & 0x3u) << 18); };      // bit j = position j -> bit 8 (j / 4) + 2 + j % 4
                    std::vector<uint32_t> record(static_cast<size_t>(N) * static_cast<size_t>(record_words), 0u);
                    std::vector<char> usable(static_cast<size_t>(N), 1);
                    bool indexable = true;
                    for(int32_t b = 0; b < N; ++b) {
                        uint32_t* const to = &record[static_cast<size_t>(b) * static_cast<size_t>(record_words)];
                        const double own = concentration[static_cast<size_t>(b)];
                        // a barcode nothing was seen of (prior zero) has no ratios: a word that names it stays unknown to the lookup
                        if(stream_priors && !(own > 0)) { usable[static_cast<size_t>(b)] = 0; }
                        std::vector<bool> is_listed(static_cast<size_t>(N), false);
                        for(int32_t k = 0; k < list; ++k) {
                            const size_t o = static_cast<size_t>(tables.listed[static_cast<size_t>(b) * static_cast<size_t>(near_listed) + static_cast<size_t>(k)]);
                            is_listed[o] = true;
                            if(one_per_word) {
                                to[k] = half(low[o]);
                                to[STREAM_PLANE_WORDS / 2 + k] = half(high[o]);
                            } else {
                                to[k / 2] |= half(low[o]) << (16 * (k & 1));
                                to[STREAM_PLANE_WORDS / 2 + k / 2] |= half(high[o]) << (16 * (k & 1));
                            }
                            if(stream_priors) {
                                // slot 0: the noise term over this barcode's prior; then each listed barcode's prior over it
                                const float ratio = k == 0 ? static_cast<float>(d.adjusted_noise_probability / own) : static_cast<float>(concentration[o] / own);
                                std::memcpy(&to[STREAM_PLANE_WORDS + 4 + k], &ratio, sizeof(float));
                            }
                        }
                        // the barcodes that are not listed, by their distance from b: 3, 4, 5, 6 or more — how many, or with
                        // priors that differ the sum of their priors over b's, rounded up; one closer than 3 leaves the bound
                        // nothing to certify
                        double weight[4] = {0, 0, 0, 0};
                        bool crowded = false;
                        for(int32_t o = 0; o < N; ++o) {
                            if(is_listed[static_cast<size_t>(o)]) { continue; }
                            const int away = __builtin_popcount((low[static_cast<size_t>(b)] ^ low[static_cast<size_t>(o)]) | (high[static_cast<size_t>(b)] ^ high[static_cast<size_t>(o)]));
                            if(away < 3) { crowded = true; }
                            else { weight[std::min(away, 6) - 3] += stream_priors ? concentration[static_cast<size_t>(o)] / own : 1.0; }
                        }
                        for(int k = 0; k < 4; ++k) {
                            const float rounded = crowded ? std::numeric_limits<float>::infinity() : static_cast<float>(weight[k]) * (stream_priors ? 1.0f + 1e-6f : 1.0f);
                            std::memcpy(&to[STREAM_PLANE_WORDS + k], &rounded, sizeof(float));
                        }
                        if(index[static_cast<size_t>(b)] < 0 || index[static_cast<size_t>(b)] >= static_cast<int32_t>(STREAM_UNKNOWN)) { indexable = false; }
                    }
                    // up to 8 nt the key is 16 bits whatever the length and an entry names the barcode by its block's LDS address and
                    // its index; beyond, the bits of positions 8 and 9 follow them and an entry is the barcode's ordinal in 16 bits
                    const int32_t tail = L > 8 ? L - 8 : 0;
                    const uint32_t positions = (1u << L) - 1u;
                    auto slot_of = [&](uint32_t key) {
                        const uint32_t l = key & positions;
                        const uint32_t h = key >> L;
                        return (l & 0xfu) | ((h & 0xfu) << 4) | (((l >> 4) & 0xfu) << 8) | (((h >> 4) & 0xfu) << 12) | ((l >> 8) << 16) | ((h >> 8) << (16 + tail));
                    };
                    if(L > 8) {
                        std::vector<uint16_t> table(static_cast<size_t>(1) << (2 * L), static_cast<uint16_t>(0xffffu));
                        for(uint32_t key = 0; key < (1u << (2 * L)); ++key) {
                            const uint16_t named = tables.table[key];
                            if(named == NEAR_NO_BARCODE || !usable[named]) { continue; }
                            table[slot_of(key)] = named;
                        }
                        d.stream_table = upload(table.data(), table.size() * sizeof(uint16_t));
                        if(!d.identity_index) { indexable = false; }
                    } else {
                        std::vector<uint32_t> table(65536, (STREAM_UNKNOWN << 18) | first_block);
                        for(uint32_t key = 0; key < (1u << (2 * L)); ++key) {
                            const uint16_t named = tables.table[key];
                            if(named == NEAR_NO_BARCODE || !usable[named]) { continue; }
                            table[slot_of(key)] = (static_cast<uint32_t>(index[named]) << 18) | (first_block + static_cast<uint32_t>(named) * stream_stride);
                        }
                        d.stream_table = upload(table.data(), table.size() * sizeof(uint32_t));
                    }
                    d.stream_record = static_cast<const uint32_t*>(upload(record.data(), record.size() * sizeof(uint32_t)));
                    d.stream_ready = indexable ? d.near_ready : 0;
                }
            }
            if(c.all_strict && c.algorithm == Algorithm::MDD && c.segment_cardinality == 1 && c.nucleotide_cardinality >= NEAR_MIN_LENGTH && c.nucleotide_cardinality <= NEAR_MAX_LENGTH &&
               d.n_barcode >= 1 && d.n_barcode <= NEAR_MAX_BARCODES && d.distance_tolerance[0] <= 1 && c.quality_masking_threshold <= 128) {
                // one lookup MDD (mdd_near.hpp): the same table — the word itself, else the lowest ordinal one mismatch away —
                // and per barcode its code bits and index
                const int32_t L = c.nucleotide_cardinality;
                const int32_t N = d.n_barcode;
                std::vector<uint32_t> low, high;
                code_planes(c, low, high);
                std::vector<uint32_t> record(static_cast<size_t>(N) * 4, 0u);
                for(int32_t b = 0; b < N; ++b) {
                    record[static_cast<size_t>(b) * 4] = low[static_cast<size_t>(b)];
                    record[static_cast<size_t>(b) * 4 + 1] = high[static_cast<size_t>(b)];
                    record[static_cast<size_t>(b) * 4 + 3] = static_cast<uint32_t>(index[static_cast<size_t>(b)]);
                }
                // two barcodes with the same word: the reference's map keeps one of them for the exact match; not taken here
                bool distinct = true;
                for(int32_t b = 0; b < N && distinct; ++b) {
                    for(int32_t o = b + 1; o < N; ++o) { if(low[static_cast<size_t>(b)] == low[static_cast<size_t>(o)] && high[static_cast<size_t>(b)] == high[static_cast<size_t>(o)]) { distinct = false; break; } }
                }
                if(distinct) {
                    const NearTables tables = build_near_tables(low, high, L, std::vector<uint32_t>(static_cast<size_t>(N), 0u));
                    d.near_table = static_cast<const uint16_t*>(upload(tables.table.data(), tables.table.size() * sizeof(uint16_t)));
                    d.mdd_near_record = static_cast<const uint32_t*>(upload(record.data(), record.size() * sizeof(uint32_t)));
                    d.mdd_near_ready = 1;
                }
            }
            const bool paired_mdd = c.algorithm == Algorithm::MDD && c.quality_masking_threshold <= 128;
            // A single index of 11 to 20 nt (past the streaming kernel's 10) is taken as two PARTS of its one segment — 8 nt and the
            // rest up to 16 nt, the last 10 beyond — by the same tier: it scores one barcode over all positions and bounds the others
            // by their distance, wherever a segment ends.  96 barcodes of 12 nt: 16 -> 3x Greads/s (the product form over all
            // barcodes was the first tier of such sets).
            const bool two_parts = (c.algorithm == Algorithm::PAMLD || paired_mdd) && c.segment_cardinality == 1 && c.nucleotide_cardinality >= 11 && c.nucleotide_cardinality <= 2 * NEAR_MAX_LENGTH &&
                                   std::getenv("PHQ_NO_PAIR_PARTS") == nullptr;
            int32_t part_start[2] = {0, 0}, part_length[2] = {0, 0};
            if(two_parts) {
                const int32_t L = c.nucleotide_cardinality;
                part_length[0] = L <= 16 ? std::min(8, L - 4) : L - NEAR_MAX_LENGTH;
                part_length[1] = L - part_length[0];
                part_start[0] = c.segment_offset[0];
                part_start[1] = c.segment_offset[0] + part_length[0];
            } else if(c.segment_cardinality == 2) {
                for(int32_t s = 0; s < 2; ++s) { part_start[s] = c.segment_offset[static_cast<size_t>(s)]; part_length[s] = c.barcode_length[static_cast<size_t>(s)]; }
                // segments of 11 to 16 nt (pairs of 12 + 12 or 16 + 16 nt): the lookup takes the first 10 nt of each — a table of 4^12 words
                // is 32 MB — and the tier scores the barcode it names over ALL its positions; the prefixes must still be all different.
                // PAMLD only: the MDD lookup decides by the whole segment's tolerance.  (Round 5: such pairs had the two word product
                // form over all barcodes as their first tier — 96 pairs of 12 + 12 nt 4.5 Greads/s, profiles/r05_shape_grid.md.)
                if(c.algorithm == Algorithm::PAMLD && c.nucleotide_cardinality <= 32) {
                    for(int32_t s = 0; s < 2; ++s) {
                        // a part is the first 10 nt of a longer segment: the pair tier scores every position through full_mask whichever barcode the prefix names
                        if(part_length[s] > NEAR_MAX_LENGTH) { part_length[s] = NEAR_MAX_LENGTH; }
                    }
                }
            }
            if(c.all_strict && ((c.algorithm == Algorithm::PAMLD && (d.uniform_prior || positive_priors)) || paired_mdd) && (c.segment_cardinality == 2 || two_parts) && c.nucleotide_cardinality <= 32 &&
               d.n_barcode > 1 && d.n_barcode <= NEAR_MAX_BARCODES &&
               part_length[0] >= 4 && part_length[0] <= NEAR_MAX_LENGTH && part_length[1] >= 4 && part_length[1] <= NEAR_MAX_LENGTH) {
                // paired barcode tier (pamld_pair.hpp): per segment the barcode of every word at most two mismatches from
                // one of its words — which must be all different in both segments — and per barcode its code bits and
                // the distance to the nearest other barcode
                const int32_t N = d.n_barcode;
                std::vector<uint32_t> low, high;
                code_planes(c, low, high);
                bool distinct = true;
                int32_t naming = 0;                           // parts whose words name a barcode each
                std::vector<uint8_t> nameless(static_cast<size_t>(d.n_barcode), 0);      // in how many parts the barcode's word is another barcode's too
                std::vector<std::vector<uint16_t>> table(2);
                for(int32_t s = 0; s < 2 && distinct; ++s) {
                    const int32_t start = part_start[s];
                    const int32_t length = part_length[s];
                    const uint32_t mask = (1u << length) - 1u;
                    std::vector<uint16_t>& nearest = table[static_cast<size_t>(s)];
                    nearest.assign(static_cast<size_t>(1) << (2 * length), NEAR_NO_BARCODE);
                    std::vector<uint32_t> word_low(static_cast<size_t>(N)), word_high(static_cast<size_t>(N));
                    std::vector<uint8_t> shared(static_cast<size_t>(N), 0);          // the barcode's word of this part is another barcode's too
                    for(int32_t b = 0; b < N; ++b) {
                        word_low[static_cast<size_t>(b)] = (low[static_cast<size_t>(b)] >> start) & mask;
                        word_high[static_cast<size_t>(b)] = (high[static_cast<size_t>(b)] >> start) & mask;
                        uint16_t& exact = nearest[word_low[static_cast<size_t>(b)] | (word_high[static_cast<size_t>(b)] << length)];
                        if(exact != NEAR_NO_BARCODE) {
                            // two barcodes share this segment word: an MDD lookup decides by the segment's own tolerance and needs a barcode
                            // per word; under PAMLD (and for the parts of a single index) a shared word names no one, as below
                            if(!two_parts && c.algorithm != Algorithm::PAMLD) { distinct = false; break; }
                            // a part of a single index: a word that several barcodes share names no one — the other part's word decides
                            shared[static_cast<size_t>(b)] = 1;
                            if(exact != PAIR_ANY_BARCODE) { shared[exact] = 1; }
                            exact = PAIR_ANY_BARCODE;
                            continue;
                        }
                        exact = static_cast<uint16_t>(b);
                    }
                    if(std::count(shared.begin(), shared.end(), 0) > 0) { ++naming; }
                    for(int32_t b = 0; b < N; ++b) { if(shared[static_cast<size_t>(b)]) { ++nameless[static_cast<size_t>(b)]; } }
                    // every word up to two mismatches from a segment word names the barcode of the NEAREST one, the lowest
                    // ordinal among equally near ones: distance 1 over distance 2, and within a distance the lowest ordinal last
                    std::vector<uint8_t> reached(nearest.size(), 3);
                    for(int32_t b = 0; b < N && distinct; ++b) { reached[word_low[static_cast<size_t>(b)] | (word_high[static_cast<size_t>(b)] << length)] = 0; }
                    for(int32_t away = 1; away <= 2 && distinct; ++away) {
                        for(int32_t b = N - 1; b >= 0; --b) {
                            const uint32_t bl = word_low[static_cast<size_t>(b)], bh = word_high[static_cast<size_t>(b)];
                            for(int32_t j = 0; j < length; ++j) {
                                for(uint32_t other = 0; other < 4; ++other) {
                                    const uint32_t l1 = (bl & ~(1u << j)) | ((other & 1u) << j);
                                    const uint32_t h1 = (bh & ~(1u << j)) | ((other >> 1) << j);
                                    if(l1 == bl && h1 == bh) { continue; }
                                    if(away == 1) {
                                        const size_t at = l1 | (h1 << length);
                                        if(reached[at] >= 1) { reached[at] = 1; nearest[at] = shared[static_cast<size_t>(b)] ? PAIR_ANY_BARCODE : static_cast<uint16_t>(b); }
                                        continue;
                                    }
                                    for(int32_t i = j + 1; i < length; ++i) {
                                        for(uint32_t second = 0; second < 4; ++second) {
                                            const uint32_t l2 = (l1 & ~(1u << i)) | ((second & 1u) << i);
                                            const uint32_t h2 = (h1 & ~(1u << i)) | ((second >> 1) << i);
                                            if(((l2 ^ bl) | (h2 ^ bh)) >> i & 1u) {
                                                const size_t at = l2 | (h2 << length);
                                                if(reached[at] >= 2) { reached[at] = 2; nearest[at] = shared[static_cast<size_t>(b)] ? PAIR_ANY_BARCODE : static_cast<uint16_t>(b); }
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                if(distinct && naming > 0) {
                    std::vector<uint32_t> record(static_cast<size_t>(N) * PAIR_RECORD_WORDS, 0u);
                    int32_t usable = 0;
                    for(int32_t b = 0; b < N; ++b) {
                        auto away = [&](int32_t o) { return __builtin_popcount((low[static_cast<size_t>(b)] ^ low[static_cast<size_t>(o)]) | (high[static_cast<size_t>(b)] ^ high[static_cast<size_t>(o)])); };
                        int radius = 31;
                        for(int32_t o = 0; o < N; ++o) { if(o != b) { radius = std::min(radius, away(o)); } }
                        uint32_t* const entry = &record[static_cast<size_t>(b) * PAIR_RECORD_WORDS];
                        entry[0] = low[static_cast<size_t>(b)];
                        entry[1] = high[static_cast<size_t>(b)];
                        entry[2] = static_cast<uint32_t>(radius);
                        entry[3] = static_cast<uint32_t>(index[static_cast<size_t>(b)]);
                        // what scales the bound on the other barcodes — those at the radius, one further, and all the rest —, and the
                        // noise term, in units of this barcode's prior
                        const double own = concentration[static_cast<size_t>(b)];
                        double weight[3] = {0, 0, 0};
                        for(int32_t o = 0; o < N; ++o) {
                            if(o != b) { weight[std::min(std::min(away(o), 31) - radius, 2)] += d.uniform_prior ? 1.0 : concentration[static_cast<size_t>(o)] / own; }
                        }
                        const float up = d.uniform_prior ? 1.0f : 1.0f + 1e-6f;          // sums of ratios bound: rounded up
                        const float others[3] = {static_cast<float>(weight[0]) * up, static_cast<float>(weight[1]) * up, static_cast<float>(weight[2]) * up};
                        const float noise_over_prior = static_cast<float>(d.adjusted_noise_probability / own);
                        std::memcpy(&entry[4], &others[0], sizeof(float));
                        std::memcpy(&entry[5], &noise_over_prior, sizeof(float));
                        std::memcpy(&entry[6], &others[1], sizeof(float));
                        std::memcpy(&entry[7], &others[2], sizeof(float));
                        // (a single index in two parts pays from a radius of 4 on: 96 barcodes of 11 nt 16 -> 28 Greads/s; the product form
                        // over all barcodes is what such a set takes otherwise)
                        if(radius >= (two_parts ? 4 : 6)) { ++usable; }
                    }
                    for(int32_t s = 0; s < 2; ++s) {
                        d.pair_table[s] = static_cast<const uint16_t*>(upload(table[static_cast<size_t>(s)].data(), table[static_cast<size_t>(s)].size() * sizeof(uint16_t)));
                        d.pair_start[s] = part_start[s];
                        d.pair_length[s] = part_length[s];
                    }
                    d.pair_record = static_cast<const uint32_t*>(upload(record.data(), record.size() * sizeof(uint32_t)));
                    if(c.algorithm == Algorithm::PAMLD) {
                        // the list tier behind the first look (pamld_pair.hpp, SECOND): per barcode its PAIR_LISTED nearest barcodes (the
                        // lowest ordinal among equally near ones first), and for the bound on everyone else the distance of the nearest
                        // barcode that is NOT listed with the others' priors over its own in three classes from there on
                        std::vector<uint32_t> lists(static_cast<size_t>(N) * PAIR_LIST_WORDS, 0u);
                        std::vector<std::pair<int, int32_t>> by_distance(static_cast<size_t>(N));
                        for(int32_t b = 0; b < N; ++b) {
                            auto away = [&](int32_t o) { return std::min(31, __builtin_popcount((low[static_cast<size_t>(b)] ^ low[static_cast<size_t>(o)]) | (high[static_cast<size_t>(b)] ^ high[static_cast<size_t>(o)]))); };
                            by_distance.clear();
                            for(int32_t o = 0; o < N; ++o) { if(o != b) { by_distance.emplace_back(away(o), o); } }
                            std::sort(by_distance.begin(), by_distance.end());
                            uint32_t* const entry = &lists[static_cast<size_t>(b) * PAIR_LIST_WORDS];
                            const double own = concentration[static_cast<size_t>(b)];
                            for(int32_t o = 0; o < PAIR_LISTED; ++o) {
                                const bool there = static_cast<size_t>(o) < by_distance.size();
                                const int32_t other = there ? by_distance[static_cast<size_t>(o)].second : b;
                                entry[o] = low[static_cast<size_t>(other)];
                                entry[PAIR_LISTED + o] = high[static_cast<size_t>(other)];
                                const float ratio = !there ? 0.0f : d.uniform_prior ? 1.0f : static_cast<float>(concentration[static_cast<size_t>(other)] / own);
                                std::memcpy(&entry[2 * PAIR_LISTED + 4 + o], &ratio, sizeof(float));
                            }
                            const int beyond = by_distance.size() > static_cast<size_t>(PAIR_LISTED) ? by_distance[PAIR_LISTED].first : 31;
                            double weight[3] = {0, 0, 0};
                            for(size_t o = PAIR_LISTED; o < by_distance.size(); ++o) {
                                weight[std::min(by_distance[o].first - beyond, 2)] += d.uniform_prior ? 1.0 : concentration[static_cast<size_t>(by_distance[o].second)] / own;
                            }
                            const float up = d.uniform_prior ? 1.0f : 1.0f + 1e-6f;          // sums of ratios bound: rounded up
                            entry[2 * PAIR_LISTED] = static_cast<uint32_t>(beyond);
                            for(int k = 0; k < 3; ++k) {
                                const float others = static_cast<float>(weight[k]) * up;
                                std::memcpy(&entry[2 * PAIR_LISTED + 1 + k], &others, sizeof(float));
                            }
                        }
                        d.pair_list = static_cast<const uint32_t*>(upload(lists.data(), lists.size() * sizeof(uint32_t)));
                    }
                    d.pair_ready = (c.algorithm == Algorithm::PAMLD && 2 * usable >= N) ? 1 : 0;              // barcodes closer than that leave the bound nothing to certify
                    // one lookup per segment is the reference's first fit (mdd.cpp:50-80) only while at most ONE word of a segment
                    // can be within its tolerance of an observation: tolerance <= (smallest distance between two of the
                    // segment's words - 1) / 2, the bound the compiler enforces (metric.h:228) and a hand written compiled
                    // document may exceed — such a codec takes the scan kernels
                    bool within_bound = paired_mdd;           // (asked of two real segments only: the parts of a single index are PAMLD's)
                    for(int32_t s = 0; s < 2 && paired_mdd; ++s) {
                        const uint32_t mask = (c.barcode_length[static_cast<size_t>(s)] >= 32 ? 0xffffffffu : ((1u << c.barcode_length[static_cast<size_t>(s)]) - 1u)) << c.segment_offset[static_cast<size_t>(s)];
                        int closest = 32;
                        for(int32_t b = 0; b < N; ++b) {
                            for(int32_t o = b + 1; o < N; ++o) {
                                closest = std::min(closest, __builtin_popcount(((low[static_cast<size_t>(b)] ^ low[static_cast<size_t>(o)]) | (high[static_cast<size_t>(b)] ^ high[static_cast<size_t>(o)])) & mask));
                            }
                        }
                        const int32_t tolerance = d.distance_tolerance[s];
                        if(tolerance < 0 || 2 * tolerance + 1 > closest) { within_bound = false; }
                    }
                    d.mdd_pair_ready = (paired_mdd && within_bound) ? 1 : 0;
                    if(two_parts && paired_mdd) {
                        // one segment in two parts: the tolerance is the whole barcode's, and at most one barcode can be within it
                        // (the same bound over all positions); a read no part names, or whose named barcode is too far, is scanned
                        int closest = 32;
                        for(int32_t b = 0; b < N; ++b) {
                            for(int32_t o = b + 1; o < N; ++o) {
                                closest = std::min(closest, __builtin_popcount((low[static_cast<size_t>(b)] ^ low[static_cast<size_t>(o)]) | (high[static_cast<size_t>(b)] ^ high[static_cast<size_t>(o)])));
                            }
                        }
                        const int32_t tolerance = d.distance_tolerance[0];
                        d.mdd_pair_parts = 1;
                        // ... and every barcode must be named by a part of its own: the scan behind the lookup takes the reference's order
                        // for granted — the exact match (mdd.cpp:43-46) before any masking — which the lookup settles only for barcodes it
                        // can name
                        const bool all_named = std::count(nameless.begin(), nameless.end(), 2) == 0;
                        d.mdd_pair_ready = (all_named && tolerance >= 0 && tolerance <= 2 && 2 * tolerance + 1 <= closest) ? 1 : 0;
                    }
                }
            }
            if(c.all_strict && c.algorithm == Algorithm::PAMLD && (d.uniform_prior || positive_priors) && c.segment_cardinality == 2 && d.n_barcode > 1 && d.n_barcode <= 256 &&
               c.barcode_length[0] >= 4 && c.barcode_length[0] <= 8 && c.barcode_length[1] >= 4 && c.barcode_length[1] <= 8) {
                // plates (pamld_plate.hip): few words a segment, every combination of them or a part
                constexpr int32_t SLOTS = 16;
                const int32_t N = d.n_barcode;
                std::vector<uint32_t> low, high;
                code_planes(c, low, high);
                std::vector<std::pair<uint32_t, uint32_t>> words[2];           // (low plane, high plane) of the segment's distinct words, first appearance order
                std::vector<int32_t> word_of[2];
                bool fits = true;
                for(int32_t s = 0; s < 2 && fits; ++s) {
                    const int32_t start = c.segment_offset[static_cast<size_t>(s)];
                    const uint32_t mask = (1u << c.barcode_length[static_cast<size_t>(s)]) - 1u;
                    word_of[s].resize(static_cast<size_t>(N));
                    for(int32_t b = 0; b < N && fits; ++b) {
                        const std::pair<uint32_t, uint32_t> word((low[static_cast<size_t>(b)] >> start) & mask, (high[static_cast<size_t>(b)] >> start) & mask);
                        auto found = std::find(words[s].begin(), words[s].end(), word);
                        if(found == words[s].end()) {
                            if(static_cast<int32_t>(words[s].size()) >= SLOTS) { fits = false; break; }
                            words[s].push_back(word);
                            found = words[s].end() - 1;
                        }
                        word_of[s][static_cast<size_t>(b)] = static_cast<int32_t>(found - words[s].begin());
                    }
                }
                const int32_t n0 = static_cast<int32_t>(words[0].size()), n1 = static_cast<int32_t>(words[1].size());
                std::vector<int32_t> barcode_of(static_cast<size_t>(n0) * static_cast<size_t>(std::max(n1, 1)), -1);
                for(int32_t b = 0; b < N && fits; ++b) {
                    int32_t& at = barcode_of[static_cast<size_t>(word_of[0][static_cast<size_t>(b)]) * static_cast<size_t>(n1) + static_cast<size_t>(word_of[1][static_cast<size_t>(b)])];
                    if(at >= 0) { fits = false; }                 // the same barcode twice
                    at = b;
                }
                // a plate, not a list of pairs: words are shared between barcodes (a set of all different words is the paired tier's)
                if(fits && n0 >= 2 && n1 >= 2 && N > std::max(n0, n1)) {
                    std::vector<uint32_t> block(static_cast<size_t>(n0 + n1) * 3 * SLOTS, 0u);
                    std::vector<std::vector<int32_t>> slot_word[2];          // per word: the word each slot of its block holds (-1: none)
                    for(int32_t s = 0; s < 2; ++s) {
                        const int32_t length = c.barcode_length[static_cast<size_t>(s)];
                        const int32_t ns = static_cast<int32_t>(words[s].size());
                        slot_word[s].assign(static_cast<size_t>(ns), std::vector<int32_t>(SLOTS, -1));
                        for(int32_t u = 0; u < ns; ++u) {
                            uint32_t* const to = &block[(static_cast<size_t>(s == 0 ? 0 : n0) + static_cast<size_t>(u)) * 3 * SLOTS];
                            int32_t slot = 0;
                            auto put = [&](int32_t w) {
                                to[slot] = words[s][static_cast<size_t>(w)].first;
                                to[SLOTS + slot] = words[s][static_cast<size_t>(w)].second;
                                const float one = 1.0f;
                                std::memcpy(&to[2 * SLOTS + slot], &one, sizeof(float));
                                slot_word[s][static_cast<size_t>(u)][static_cast<size_t>(slot)] = w;
                                ++slot;
                            };
                            put(u);
                            for(int32_t w = 0; w < ns; ++w) { if(w != u) { put(w); } }
                        }
                        // the lookup: the word itself and every word one mismatch from it; a key two words claim names none
                        std::vector<uint8_t> table(static_cast<size_t>(1) << (2 * length), static_cast<uint8_t>(0xffu));
                        std::vector<uint8_t> reached(table.size(), 2);
                        for(int32_t u = 0; u < ns; ++u) {
                            const size_t at = words[s][static_cast<size_t>(u)].first | (words[s][static_cast<size_t>(u)].second << length);
                            table[at] = static_cast<uint8_t>(u);
                            reached[at] = 0;
                        }
                        for(int32_t u = 0; u < ns; ++u) {
                            const uint32_t wl = words[s][static_cast<size_t>(u)].first, wh = words[s][static_cast<size_t>(u)].second;
                            for(int32_t j = 0; j < length; ++j) {
                                for(uint32_t other = 0; other < 4; ++other) {
                                    const uint32_t l1 = (wl & ~(1u << j)) | ((other & 1u) << j);
                                    const uint32_t h1 = (wh & ~(1u << j)) | ((other >> 1) << j);
                                    if(l1 == wl && h1 == wh) { continue; }
                                    const size_t at = l1 | (h1 << length);
                                    if(reached[at] == 0) { continue; }                           // a word of the set
                                    if(reached[at] == 1 && table[at] != static_cast<uint8_t>(u)) { table[at] = static_cast<uint8_t>(0xffu); reached[at] = 3; continue; }
                                    if(reached[at] == 2) { table[at] = static_cast<uint8_t>(u); reached[at] = 1; }
                                }
                            }
                        }
                        // ... and, round 5, the word TWO mismatches away where no word is nearer and no other word as near (the kernel scores
                        // every word of the segment whichever one is named: a read two low quality calls off its word — 3 % of H94's
                        // reads — was handed on for the name alone)
                        for(int32_t u = 0; u < ns; ++u) {
                            const uint32_t wl = words[s][static_cast<size_t>(u)].first, wh = words[s][static_cast<size_t>(u)].second;
                            for(int32_t j = 0; j < length; ++j) {
                                for(int32_t i = j + 1; i < length; ++i) {
                                    for(uint32_t pair_of = 0; pair_of < 16; ++pair_of) {
                                        const uint32_t first = pair_of & 3u, second = pair_of >> 2;
                                        const uint32_t l2 = (wl & ~((1u << j) | (1u << i))) | ((first & 1u) << j) | ((second & 1u) << i);
                                        const uint32_t h2 = (wh & ~((1u << j) | (1u << i))) | ((first >> 1) << j) | ((second >> 1) << i);
                                        if(!((((l2 ^ wl) | (h2 ^ wh)) >> j) & 1u) || !((((l2 ^ wl) | (h2 ^ wh)) >> i) & 1u)) { continue; }       // both positions differ
                                        const size_t at = l2 | (h2 << length);
                                        if(reached[at] == 2) { table[at] = static_cast<uint8_t>(u); reached[at] = 4; }
                                        else if(reached[at] == 4 && table[at] != static_cast<uint8_t>(u)) { table[at] = static_cast<uint8_t>(0xffu); reached[at] = 5; }
                                    }
                                }
                            }
                        }
                        d.plate_table[s] = static_cast<const uint8_t*>(upload(table.data(), table.size()));
                        d.plate_unique[s] = ns;
                        d.plate_start[s] = c.segment_offset[static_cast<size_t>(s)];
                        d.plate_length[s] = length;
                    }
                    double largest = 0;
                    for(int32_t b = 0; b < N; ++b) { largest = std::max(largest, concentration[static_cast<size_t>(b)]); }
                    std::vector<float> ratio(static_cast<size_t>(n0) * static_cast<size_t>(n1) * 2 * SLOTS, 0.0f);
                    std::vector<uint32_t> info(static_cast<size_t>(n0) * static_cast<size_t>(n1) * 2, 0u);
                    auto prior_of = [&](int32_t u0, int32_t u1) {
                        const int32_t b = (u0 < 0 || u1 < 0) ? -1 : barcode_of[static_cast<size_t>(u0) * static_cast<size_t>(n1) + static_cast<size_t>(u1)];
                        return b < 0 ? 0.0 : concentration[static_cast<size_t>(b)];
                    };
                    for(int32_t u0 = 0; u0 < n0; ++u0) {
                        for(int32_t u1 = 0; u1 < n1; ++u1) {
                            const size_t combination = static_cast<size_t>(u0) * static_cast<size_t>(n1) + static_cast<size_t>(u1);
                            float* const to = &ratio[combination * 2 * SLOTS];
                            const int32_t b = barcode_of[combination];
                            const double own = prior_of(u0, u1);
                            if(b < 0 || !(own > 0)) {
                                for(int32_t k = 0; k < 2 * SLOTS; ++k) { to[k] = std::numeric_limits<float>::infinity(); }      // not in the set: nothing is finished on it
                                continue;
                            }
                            to[0] = static_cast<float>(d.adjusted_noise_probability / own);
                            to[SLOTS] = static_cast<float>(largest / own * (1.0 + 1e-6));                      // a bound: rounded up
                            for(int32_t k = 1; k < SLOTS; ++k) {
                                to[k] = static_cast<float>(prior_of(u0, slot_word[1][static_cast<size_t>(u1)][static_cast<size_t>(k)]) / own);
                                to[SLOTS + k] = static_cast<float>(prior_of(slot_word[0][static_cast<size_t>(u0)][static_cast<size_t>(k)], u1) / own);
                            }
                            info[combination * 2] = static_cast<uint32_t>(b + 1);
                            info[combination * 2 + 1] = static_cast<uint32_t>(index[static_cast<size_t>(b)]);
                        }
                    }
                    d.plate_block = static_cast<const uint32_t*>(upload(block.data(), block.size() * sizeof(uint32_t)));
                    d.plate_ratio = static_cast<const uint32_t*>(upload(ratio.data(), ratio.size() * sizeof(float)));
                    d.plate_info = static_cast<const uint32_t*>(upload(info.data(), info.size() * sizeof(uint32_t)));
                    d.plate_ready = 1;
                }
            }
            if(c.all_strict && c.algorithm == Algorithm::PAMLD && c.nucleotide_cardinality > 16 && c.nucleotide_cardinality <= 32 && d.n_barcode > 0) {
                // two words of bit-sliced 2-bit bases per barcode for the long barcode product form (pamld_wide.hpp)
                const int32_t padded = (d.n_barcode + 7) / 8 * 8;
                std::vector<uint32_t> wide(static_cast<size_t>(padded) * 2, 0u);
                std::vector<float> prior(static_cast<size_t>(padded), 0.0f);
                for(int32_t b = 0; b < padded; ++b) {
                    const int32_t source = b < d.n_barcode ? b : 0;          // padding repeats barcode 0; its probability is zeroed
                    for(int32_t j = 0; j < c.nucleotide_cardinality; ++j) {
                        const uint32_t code = c.barcode_code[static_cast<size_t>(source) * static_cast<size_t>(c.nucleotide_cardinality) + static_cast<size_t>(j)];
                        const uint32_t two = code == 1 ? 0u : code == 2 ? 1u : code == 4 ? 2u : 3u;
                        const uint32_t low = 8u * static_cast<uint32_t>((j >> 2) & 3) + 1u + static_cast<uint32_t>(j & 3);
                        uint32_t& word = wide[static_cast<size_t>(b) * 2 + static_cast<size_t>(j >> 4)];
                        word |= (two & 1u) << low;
                        word |= (two >> 1) << ((low + 4u) & 31u);
                    }
                    if(b < d.n_barcode) { prior[static_cast<size_t>(b)] = static_cast<float>(concentration[static_cast<size_t>(b)]); }
                }
                d.barcode_wide = static_cast<const uint32_t*>(upload(wide.data(), wide.size() * sizeof(uint32_t)));
                d.concentration_single = static_cast<const float*>(upload(prior.data(), prior.size() * sizeof(float)));
                d.factor_pairs = device_factor_pairs;
            }
            if(c.all_strict && ((c.algorithm == Algorithm::PAMLD && (d.uniform_prior || c.segment_cardinality == 2)) || c.algorithm == Algorithm::MDD) &&
               d.n_barcode > 1 && c.segment_cardinality >= 1 && c.segment_cardinality <= CROSS_MAX_SEGMENT && c.nucleotide_cardinality <= DEV_TIER_NUCLEOTIDE) {
                // A combinatorial barcode set (every combination of per segment sets, once)?  Then P(r|b) factorises
                // over the segments and pamld_cross.hpp scans the segments' distinct words instead of the barcodes.
                // A single segment set is the trivial case: it only uses the exact tier of that file.
                const int32_t S = c.segment_cardinality;
                std::vector<std::vector<uint32_t>> distinct(static_cast<size_t>(S));       // 2 bits per base keys, first appearance order
                std::vector<int32_t> id(static_cast<size_t>(d.n_barcode) * static_cast<size_t>(S));
                bool supported = true;
                int32_t groups = 1;
                for(int32_t s = 0; s < S && supported; ++s) {
                    const int32_t begin = c.segment_offset[static_cast<size_t>(s)];
                    const int32_t length = c.barcode_length[static_cast<size_t>(s)];
                    if(length < 1 || length > 16) { supported = false; break; }
                    groups = std::max(groups, (length + 3) / 4);
                    for(int32_t b = 0; b < d.n_barcode; ++b) {
                        uint32_t key = 0;
                        for(int32_t j = 0; j < length; ++j) {
                            const uint32_t code = c.barcode_code[static_cast<size_t>(b) * static_cast<size_t>(c.nucleotide_cardinality) + static_cast<size_t>(begin + j)];
                            key |= (code == 1 ? 0u : code == 2 ? 1u : code == 4 ? 2u : 3u) << (2 * j);
                        }
                        auto& set = distinct[static_cast<size_t>(s)];
                        auto found = std::find(set.begin(), set.end(), key);
                        if(found == set.end()) {
                            if(set.size() >= static_cast<size_t>(CROSS_MAX_UNIQUE)) { supported = false; break; }
                            set.push_back(key);
                            found = set.end() - 1;
                        }
                        id[static_cast<size_t>(b) * static_cast<size_t>(S) + static_cast<size_t>(s)] = static_cast<int32_t>(found - set.begin());
                    }
                }
                int64_t combinations = 1;
                for(int32_t s = 0; s < S && supported; ++s) { combinations *= static_cast<int64_t>(distinct[static_cast<size_t>(s)].size()); }
                if(supported && combinations == d.n_barcode) {
                    int32_t stride[CROSS_MAX_SEGMENT] = {0, 0, 0};
                    int32_t weight = 1;
                    for(int32_t s = S - 1; s >= 0; --s) { stride[s] = weight; weight *= static_cast<int32_t>(distinct[static_cast<size_t>(s)].size()); }
                    std::vector<int32_t> ordinal(static_cast<size_t>(d.n_barcode), -1);
                    for(int32_t b = 0; b < d.n_barcode && supported; ++b) {
                        int32_t flat = 0;
                        for(int32_t s = 0; s < S; ++s) { flat += id[static_cast<size_t>(b) * static_cast<size_t>(S) + static_cast<size_t>(s)] * stride[s]; }
                        if(ordinal[static_cast<size_t>(flat)] >= 0) { supported = false; }      // the same barcode twice
                        ordinal[static_cast<size_t>(flat)] = b;
                    }
                    if(supported) {
                        std::vector<uint32_t> words(static_cast<size_t>(CROSS_MAX_SEGMENT) * CROSS_MAX_UNIQUE, 0u);
                        for(int32_t s = 0; s < S; ++s) {
                            const int32_t length = c.barcode_length[static_cast<size_t>(s)];
                            const auto& set = distinct[static_cast<size_t>(s)];
                            for(size_t u = 0; u < set.size(); ++u) {
                                uint32_t word = 0;
                                for(int32_t j = 0; j < length; ++j) {
                                    const uint32_t two = (set[u] >> (2 * j)) & 3u;
                                    const uint32_t low = 8u * static_cast<uint32_t>(j >> 2) + 2u + static_cast<uint32_t>(j & 3);
                                    word |= (two & 1u) << low;
                                    word |= (two >> 1) << ((low + 4u) & 31u);
                                }
                                words[static_cast<size_t>(s) * CROSS_MAX_UNIQUE + u] = word;
                            }
                            d.cross_start[s] = c.segment_offset[static_cast<size_t>(s)];
                            d.cross_length[s] = length;
                            d.cross_unique[s] = static_cast<int32_t>(set.size());
                            d.cross_stride[s] = stride[s];
                        }
                        d.cross_segments = S;
                        d.cross_groups = groups;
                        d.cross_identity = 1;
                        for(int32_t b = 0; b < d.n_barcode; ++b) { if(ordinal[static_cast<size_t>(b)] != b) { d.cross_identity = 0; } }
                        d.cross_words = static_cast<const uint32_t*>(upload(words.data(), words.size() * sizeof(uint32_t)));
                        d.cross_ordinal = static_cast<const int32_t*>(upload(ordinal.data(), ordinal.size() * sizeof(int32_t)));
                        if(S == 2 && c.algorithm == Algorithm::PAMLD && d.cross_identity && d.identity_index &&
                           c.barcode_length[0] == c.barcode_length[1] && c.barcode_length[0] >= NEAR_MIN_LENGTH && c.barcode_length[0] <= NEAR_MAX_LENGTH &&
                           static_cast<int32_t>(distinct[0].size()) >= near_list(c.barcode_length[0]) && static_cast<int32_t>(distinct[1].size()) >= near_list(c.barcode_length[0])) {
                            // neighbour list tier of the set (pamld_cross_near.hpp): pamld_near.hpp's tables per segment
                            const int32_t length = c.barcode_length[0];
                            int32_t usable = 0;
                            std::vector<int32_t> listed[2];
                            std::vector<uint16_t> near_table_of[2];
                            for(int32_t s2 = 0; s2 < 2; ++s2) {
                                const auto& set = distinct[static_cast<size_t>(s2)];
                                std::vector<uint32_t> low(set.size(), 0u), high(set.size(), 0u), tag(set.size());
                                for(size_t u = 0; u < set.size(); ++u) {
                                    for(int32_t j = 0; j < length; ++j) {
                                        const uint32_t two = (set[u] >> (2 * j)) & 3u;
                                        low[u] |= (two & 1u) << j;
                                        high[u] |= (two >> 1) << j;
                                    }
                                    tag[u] = static_cast<uint32_t>(u) * static_cast<uint32_t>(stride[s2]);
                                }
                                const NearTables tables = build_near_tables(low, high, length, tag);
                                d.cross_near_table[s2] = static_cast<const uint16_t*>(upload(tables.table.data(), tables.table.size() * sizeof(uint16_t)));
                                d.cross_near_record[s2] = static_cast<const uint32_t*>(upload(tables.record.data(), tables.record.size() * sizeof(uint32_t)));
                                if(2 * tables.usable >= static_cast<int32_t>(set.size())) { ++usable; }
                                listed[s2] = tables.listed;
                                near_table_of[s2] = tables.table;
                            }
                            d.cross_near_ready = usable == 2 ? 1 : 0;
                            if(d.cross_near_ready && length >= 8) {          // (priors that differ: pass A of the same kernel, with the weights below)
                                // the streaming kernel (pamld_cross_stream.hpp): per segment the lookup table and the blocks of
                                // pamld_stream.hpp over the segment's words, one listed word per plane word
                                const int32_t near_listed = near_list(length);
                                const int32_t list = stream_list(length);
                                const int32_t block_words = cross_stream_block_words(length);
                                const int32_t unique0 = static_cast<int32_t>(distinct[0].size()), unique1 = static_cast<int32_t>(distinct[1].size());
                                const int32_t tail = length > 8 ? length - 8 : 0;
                                auto half = [](uint32_t bits) { return ((bits & 0xfu) << 2) | (((bits >> 4) & 0xfu) << 10) | (((bits >> 8) & 0x3u) << 18); };
                                bool fits = cross_stream_block_at(CROSS_STREAM_WAVES, length, unique0, unique1, 1) + static_cast<uint32_t>(unique1) * static_cast<uint32_t>(block_words) * 4u < (1u << 18);
                                for(int32_t s2 = 0; s2 < 2 && fits; ++s2) {
                                    const auto& set = distinct[static_cast<size_t>(s2)];
                                    const int32_t N = static_cast<int32_t>(set.size());
                                    if(N >= static_cast<int32_t>(CROSS_STREAM_UNKNOWN)) { fits = false; break; }      // an ordinal is a byte of the table
                                    std::vector<uint32_t> low(set.size(), 0u), high(set.size(), 0u);
                                    for(size_t u = 0; u < set.size(); ++u) {
                                        for(int32_t j = 0; j < length; ++j) {
                                            const uint32_t two = (set[u] >> (2 * j)) & 3u;
                                            low[u] |= (two & 1u) << j;
                                            high[u] |= (two >> 1) << j;
                                        }
                                    }
                                    std::vector<uint32_t> record(static_cast<size_t>(N) * static_cast<size_t>(block_words), 0u);
                                    for(int32_t b = 0; b < N; ++b) {
                                        uint32_t* const to = &record[static_cast<size_t>(b) * static_cast<size_t>(block_words)];
                                        std::vector<bool> is_listed(static_cast<size_t>(N), false);
                                        for(int32_t k = 0; k < list; ++k) {
                                            const size_t o = static_cast<size_t>(listed[s2][static_cast<size_t>(b) * static_cast<size_t>(near_listed) + static_cast<size_t>(k)]);
                                            is_listed[o] = true;
                                            to[k] = half(low[o]);
                                            to[list + k] = half(high[o]);
                                        }
                                        // the words that are not listed, by their distance from b: 3, 4, 5, 6 or more; one closer than 3
                                        // leaves the bound nothing to certify
                                        double weight[4] = {0, 0, 0, 0};
                                        bool crowded = false;
                                        for(int32_t o = 0; o < N; ++o) {
                                            if(is_listed[static_cast<size_t>(o)]) { continue; }
                                            const int away = __builtin_popcount((low[static_cast<size_t>(b)] ^ low[static_cast<size_t>(o)]) | (high[static_cast<size_t>(b)] ^ high[static_cast<size_t>(o)]));
                                            if(away < 3) { crowded = true; } else { weight[std::min(away, 6) - 3] += 1.0; }
                                        }
                                        for(int k = 0; k < 4; ++k) {
                                            const float rounded = crowded ? std::numeric_limits<float>::infinity() : static_cast<float>(weight[k]);
                                            std::memcpy(&to[2 * list + k], &rounded, sizeof(float));
                                        }
                                    }
                                    std::vector<uint8_t> table(length > 8 ? (static_cast<size_t>(1) << (2 * length)) : 65536, CROSS_STREAM_UNKNOWN);      // a word's ordinal
                                    const uint32_t positions = (1u << length) - 1u;
                                    const std::vector<uint16_t>& named_by = near_table_of[s2];
                                    for(uint32_t key = 0; key < (1u << (2 * length)); ++key) {
                                        const uint32_t l = key & positions;
                                        const uint32_t h = key >> length;
                                        const uint16_t named = named_by[key];
                                        if(named == NEAR_NO_BARCODE) { continue; }
                                        table[(l & 0xfu) | ((h & 0xfu) << 4) | (((l >> 4) & 0xfu) << 8) | (((h >> 4) & 0xfu) << 12) | ((l >> 8) << 16) | ((h >> 8) << (16 + tail))] = static_cast<uint8_t>(named);
                                    }
                                    d.cross_stream_table[s2] = upload(table.data(), table.size() * sizeof(uint8_t));
                                    d.cross_stream_record[s2] = static_cast<const uint32_t*>(upload(record.data(), record.size() * sizeof(uint32_t)));
                                }
                                d.cross_stream_ready = fits ? 1 : 0;
                            }
                            if(!d.uniform_prior && d.cross_near_ready) {
                                // priors that differ: the listed words' priors over the combination's, along the row and the
                                // column of every combination (device_types.hpp, cross_near_weight).  A combination whose own
                                // prior is zero (nothing seen of it by the estimation pass) gets infinities: the kernel's
                                // certificate fails on them and the read goes to the weighted scans.
                                const size_t n0 = distinct[0].size(), n1 = distinct[1].size();
                                const size_t list = static_cast<size_t>(near_list(length));
                                auto prior_of = [&](size_t u0, size_t u1) {
                                    return concentration[static_cast<size_t>(ordinal[u0 * static_cast<size_t>(stride[0]) + u1 * static_cast<size_t>(stride[1])])];
                                };
                                double largest = 0;
                                for(int32_t b = 0; b < d.n_barcode; ++b) { largest = std::max(largest, concentration[static_cast<size_t>(b)]); }
                                std::vector<float> along_row(n0 * n1 * CROSS_NEAR_WEIGHTS, 0.0f), along_column(n0 * n1 * CROSS_NEAR_WEIGHTS, 0.0f);
                                for(size_t u0 = 0; u0 < n0; ++u0) {
                                    for(size_t u1 = 0; u1 < n1; ++u1) {
                                        const double own = prior_of(u0, u1);
                                        float* const row = &along_row[(u0 * n1 + u1) * CROSS_NEAR_WEIGHTS];
                                        float* const column = &along_column[(u0 * n1 + u1) * CROSS_NEAR_WEIGHTS];
                                        row[0] = static_cast<float>(d.adjusted_noise_probability / own);
                                        column[0] = static_cast<float>(largest / own * (1.0 + 1e-6));          // a bound: rounded up
                                        for(size_t k = 1; k < list; ++k) {
                                            row[k] = static_cast<float>(prior_of(u0, static_cast<size_t>(listed[1][u1 * list + k])) / own);
                                            column[k] = static_cast<float>(prior_of(static_cast<size_t>(listed[0][u0 * list + k]), u1) / own);
                                        }
                                    }
                                }
                                d.cross_near_weight[0] = static_cast<const float*>(upload(along_row.data(), along_row.size() * sizeof(float)));
                                d.cross_near_weight[1] = static_cast<const float*>(upload(along_column.data(), along_column.size() * sizeof(float)));
                            }
                        }
                        if(S == 2 && c.algorithm == Algorithm::MDD && d.cross_identity && c.quality_masking_threshold <= 128 &&
                           d.distance_tolerance[0] <= 1 && d.distance_tolerance[1] <= 1 &&
                           c.barcode_length[0] >= NEAR_MIN_LENGTH && c.barcode_length[0] <= NEAR_MAX_LENGTH &&
                           c.barcode_length[1] >= NEAR_MIN_LENGTH && c.barcode_length[1] <= NEAR_MAX_LENGTH) {
                            // one lookup per segment (mdd_near.hpp, COMBINATORIAL): pamld_near.hpp's table over each segment's
                            // distinct words, and the words' code bits
                            std::vector<uint32_t> planes;
                            bool within_bound = true;          // as for the paired kind: at most one word of a segment within its tolerance
                            for(int32_t s2 = 0; s2 < 2; ++s2) {
                                const int32_t length = c.barcode_length[static_cast<size_t>(s2)];
                                const auto& set = distinct[static_cast<size_t>(s2)];
                                std::vector<uint32_t> low(set.size(), 0u), high(set.size(), 0u);
                                for(size_t u = 0; u < set.size(); ++u) {
                                    for(int32_t j = 0; j < length; ++j) {
                                        const uint32_t two = (set[u] >> (2 * j)) & 3u;
                                        low[u] |= (two & 1u) << j;
                                        high[u] |= (two >> 1) << j;
                                    }
                                    planes.push_back(low[u]);
                                    planes.push_back(high[u]);
                                }
                                int closest = 32;
                                for(size_t u = 0; u < set.size(); ++u) {
                                    for(size_t o = u + 1; o < set.size(); ++o) { closest = std::min(closest, __builtin_popcount((low[u] ^ low[o]) | (high[u] ^ high[o]))); }
                                }
                                const int32_t tolerance = d.distance_tolerance[s2];
                                if(tolerance < 0 || 2 * tolerance + 1 > closest) { within_bound = false; }
                                const NearTables tables = build_near_tables(low, high, length, std::vector<uint32_t>(set.size(), 0u));
                                d.cross_near_table[s2] = static_cast<const uint16_t*>(upload(tables.table.data(), tables.table.size() * sizeof(uint16_t)));
                            }
                            d.mdd_cross_word = static_cast<const uint32_t*>(upload(planes.data(), planes.size() * sizeof(uint32_t)));
                            d.mdd_cross_ready = within_bound ? 1 : 0;
                        }
                        if(!d.uniform_prior && S == 2 && c.algorithm == Algorithm::PAMLD) {
                            // priors that differ between the barcodes: by combination, single precision, for the weighted scans
                            const size_t n0 = distinct[0].size();
                            const size_t n1 = distinct[1].size();
                            std::vector<float> by_combination(n0 * CROSS_MAX_UNIQUE, 0.0f);            // [u0][u1], rows zero padded
                            std::vector<float> transposed(n1 * CROSS_MAX_UNIQUE, 0.0f);                // [u1][u0]
                            std::vector<double> by_combination_double(ordinal.size());
                            double largest = 0;
                            for(size_t u0 = 0; u0 < n0; ++u0) {
                                for(size_t u1 = 0; u1 < n1; ++u1) {
                                    const size_t flat = u0 * static_cast<size_t>(stride[0]) + u1 * static_cast<size_t>(stride[1]);
                                    const double prior = concentration[static_cast<size_t>(ordinal[flat])];
                                    by_combination[u0 * CROSS_MAX_UNIQUE + u1] = static_cast<float>(prior);
                                    transposed[u1 * CROSS_MAX_UNIQUE + u0] = static_cast<float>(prior);
                                    by_combination_double[flat] = prior;
                                    largest = std::max(largest, prior);
                                }
                            }
                            d.cross_prior = static_cast<const float*>(upload(by_combination.data(), by_combination.size() * sizeof(float)));
                            d.cross_prior_transposed = static_cast<const float*>(upload(transposed.data(), transposed.size() * sizeof(float)));
                            d.cross_prior_double = static_cast<const double*>(upload(by_combination_double.data(), by_combination_double.size() * sizeof(double)));
                            d.cross_prior_max = largest * (1.0 + 1e-6);          // covers the rounding of the single precision copies
                        }
                    }
                }
            }
            if(d.cross_segments == 0 && c.all_strict && c.algorithm == Algorithm::PAMLD && d.n_barcode > 1 &&
               d.n_barcode <= EXACT_SINGLE_UNIQUE && c.nucleotide_cardinality >= 1 && c.nucleotide_cardinality <= 16) {
                // Not every combination of per segment sets (unique dual index PAIRS), or more distinct words than the
                // factorised kernels keep: the exact tier still serves the codec with its whole barcode — the segments
                // one after the other, up to 16 nt — taken as ONE word per barcode, EXACT_SINGLE_UNIQUE of them at most.
                std::vector<uint32_t> words(static_cast<size_t>(EXACT_SINGLE_UNIQUE), 0u);
                std::vector<int32_t> ordinal(static_cast<size_t>(d.n_barcode));
                std::vector<uint32_t> seen;
                bool distinct_words = true;
                for(int32_t b = 0; b < d.n_barcode && distinct_words; ++b) {
                    uint32_t word = 0;
                    uint32_t key = 0;
                    for(int32_t j = 0; j < c.nucleotide_cardinality; ++j) {
                        const uint32_t code = c.barcode_code[static_cast<size_t>(b) * static_cast<size_t>(c.nucleotide_cardinality) + static_cast<size_t>(j)];
                        const uint32_t two = code == 1 ? 0u : code == 2 ? 1u : code == 4 ? 2u : 3u;
                        const uint32_t low = 8u * static_cast<uint32_t>(j >> 2) + 2u + static_cast<uint32_t>(j & 3);
                        word |= (two & 1u) << low;
                        word |= (two >> 1) << ((low + 4u) & 31u);
                        key |= two << (2 * j);
                    }
                    if(std::find(seen.begin(), seen.end(), key) != seen.end()) { distinct_words = false; }      // the same barcode twice
                    seen.push_back(key);
                    words[static_cast<size_t>(b)] = word;
                    ordinal[static_cast<size_t>(b)] = b;
                }
                if(distinct_words) {
                    d.cross_start[0] = 0;
                    d.cross_length[0] = c.nucleotide_cardinality;
                    d.cross_unique[0] = d.n_barcode;
                    d.cross_stride[0] = 1;
                    d.cross_segments = 1;
                    d.cross_groups = (c.nucleotide_cardinality + 3) / 4;
                    d.cross_identity = 1;
                    d.cross_words = static_cast<const uint32_t*>(upload(words.data(), words.size() * sizeof(uint32_t)));
                    d.cross_ordinal = static_cast<const int32_t*>(upload(ordinal.data(), ordinal.size() * sizeof(int32_t)));
                }
            }
        }
    }
    device_decoder_ = static_cast<DevDecoder*>(upload(host_decoder_.data(), host_decoder_.size() * sizeof(DevDecoder)));

    if(host_only) { return; }
    void* stats = nullptr;
    check(hipMalloc(&stats, static_cast<size_t>(job_.stats_words) * sizeof(uint64_t)), "hipMalloc");
    device_allocation_.push_back(stats);
    device_stats_ = static_cast<unsigned long long*>(stats);
    check(hipMemset(device_stats_, 0, static_cast<size_t>(job_.stats_words) * sizeof(uint64_t)), "hipMemset");
}

// ---- extraction: Rule::apply over a batch (transform.h:142-169) -----------------------------------------

void DecoderSet::stage_reads(int64_t n_read, int32_t n_segment, const uint8_t* const* code, const uint8_t* const* quality,
                             const int64_t* const* offset, const int32_t* const* length, const uint8_t* in_flag,
                             int64_t batch_capacity, uint8_t* base_planes, uint8_t* quality_planes,
                             uint8_t* out_flag, int64_t* kept, int64_t* n_kept,
                             uint64_t* incoming_count, uint64_t* incoming_pf_count) const {
    if(job_.input_segment_cardinality > 0 && n_segment != job_.input_segment_cardinality) {
        throw ConfigurationError("expected " + std::to_string(job_.input_segment_cardinality) + " input segments but got " + std::to_string(n_segment));
    }
    for(const auto& d : job_.decoder) {
        for(const auto& t : d.transform) {
            if(t.input_segment_index < 0 || t.input_segment_index >= n_segment) {
                throw ConfigurationError("invalid input feed reference " + std::to_string(t.input_segment_index) + " in token");
            }
        }
    }
    // pass 1, in order: the pre-filters and the incoming counters decide which reads take a row
    int64_t out = 0;
    uint64_t count = 0;
    uint64_t pf_count = 0;
    std::vector<int64_t> local_kept;
    int64_t* const source = kept != nullptr ? kept : (local_kept.resize(static_cast<size_t>(std::min(n_read, batch_capacity))), local_kept.data());
    for(int64_t i = 0; i < n_read; ++i) {
        const bool qcfail = in_flag != nullptr && (in_flag[i] & 1);
        ++count;                                                        // transcode.cpp:304-309
        if(!qcfail) { ++pf_count; }
        if(job_.filter_incoming_qc_fail && qcfail) { continue; }       // transcode.h:205
        bool too_short = false;                                         // transcode.h:193-200, from segment 1 on
        for(int32_t s = 1; s < n_segment && s < static_cast<int32_t>(job_.min_input_length.size()); ++s) {
            if(length[s][i] < job_.min_input_length[static_cast<size_t>(s)]) { too_short = true; break; }
        }
        if(too_short) { continue; }
        if(out >= batch_capacity) { throw OverflowError("more reads than the batch capacity " + std::to_string(batch_capacity)); }
        source[out] = i;
        ++out;
    }

    // pass 2, rows are independent: Rule::apply of every decoder into its plane, a few threads on large batches
    parallel_ranges(out, 1 << 15, host_threads(), [&](int64_t first, int64_t last, int) {
        for(int64_t row = first; row < last; ++row) {
            const int64_t i = source[row];
            for(const auto& d : job_.decoder) {
                if(d.plane_stride == 0) { continue; }
                uint8_t* row_code = base_planes + batch_capacity * d.plane_prefix + row * d.plane_stride;
                uint8_t* row_quality = quality_planes + batch_capacity * d.plane_prefix + row * d.plane_stride;
                std::memset(row_code, 0, static_cast<size_t>(d.plane_stride));
                std::memset(row_quality, 0, static_cast<size_t>(d.plane_stride));
                int32_t filled[MAX_SEGMENT] = {0};
                for(const auto& t : d.transform) {
                    const int32_t from_length = length[t.input_segment_index][i];
                    const uint8_t* from_code = code[t.input_segment_index] + offset[t.input_segment_index][i];
                    const uint8_t* from_quality = quality[t.input_segment_index] + offset[t.input_segment_index][i];
                    const int32_t start = t.absolute_start(from_length);
                    const int32_t end = t.absolute_end(from_length);
                    const int32_t size = end - start;
                    if(size <= 0) { continue; }
                    const int32_t o = t.output_segment_index;
                    const int32_t room = d.barcode_length[static_cast<size_t>(o)] - filled[o];
                    const int32_t take = size < room ? size : room;
                    uint8_t* to_code = row_code + d.segment_offset[static_cast<size_t>(o)] + filled[o];
                    uint8_t* to_quality = row_quality + d.segment_offset[static_cast<size_t>(o)] + filled[o];
                    if(t.reverse_complement) {
                        for(int32_t j = 0; j < take; ++j) {
                            to_code[j] = reverse_complement_bam(from_code[end - j - 1]);
                            to_quality[j] = from_quality[end - j - 1];
                        }
                    } else {
                        std::memcpy(to_code, from_code + start, static_cast<size_t>(take));
                        std::memcpy(to_quality, from_quality + start, static_cast<size_t>(take));
                    }
                    filled[o] += take;
                }
            }
            if(out_flag != nullptr) { out_flag[row] = (in_flag != nullptr && (in_flag[i] & 1)) ? 1 : 0; }
        }
    });
    if(n_kept != nullptr) { *n_kept = out; }
    if(incoming_count != nullptr) { *incoming_count += count; }
    if(incoming_pf_count != nullptr) { *incoming_pf_count += pf_count; }
}

// ---- classification ------------------------------------------------------------------------------------

void DecoderSet::classify_device(int64_t batch_capacity, int64_t first_read, int64_t n_read,
                                 const uint8_t* base_planes, const uint8_t* quality_planes, const uint8_t* flag,
                                 DevResult* result, uint32_t* word, void* stream) {
    require_device();
    if(n_read <= 0) { return; }
    if(first_read < 0 || first_read + n_read > batch_capacity) { throw OverflowError("read range exceeds the batch capacity"); }
    // worklists, ring entries and the worklist stride hold read ordinals of one call in 32 (31) bits
    if(n_read > (int64_t(1) << 31) - (int64_t(1) << 20)) { throw OverflowError("more than 2^31 reads in one call: classify the batch in pieces"); }
    if(word == nullptr) { throw InternalError("the status word buffer is mandatory"); }
    if(job_.plane_width > 0 && (base_planes == nullptr || quality_planes == nullptr)) { throw InternalError("observation planes are missing"); }
    const size_t n_decoder = job_.decoder.size();
    if(n_decoder == 0) {
        throw ConfigurationError("the job declares no decoder");
    }
    uint32_t* worklist = nullptr;
    if(needs_worklist()) {
        if(worklist_capacity_ < n_read) {
            // grown outside of any stream capture; sized for the worst case (every read deferred)
            check(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize");
            if(order_pending_ && order_event_ != nullptr) { check(hipEventSynchronize(static_cast<hipEvent_t>(order_event_)), "hipEventSynchronize"); }      // the previous call may have run on another stream
            (void)hipFree(device_worklist_);
            device_worklist_ = nullptr;
            check(hipMalloc(reinterpret_cast<void**>(&device_worklist_), worklist_words(n_read) * sizeof(uint32_t)), "hipMalloc");
            worklist_capacity_ = n_read;
        }
        worklist = device_worklist_;
    }
    // The worklist, the carry region, the per workgroup counters in front of them and the counter staging words of
    // count_results_kernel (DevDecoder::count_staging) belong to the set, not to the call: two calls on two streams must
    // not run at once, with or without a worklist.  A call on another stream than the previous one first waits
    // (on the device) for that one's kernels; calls on one stream are ordered by the stream itself.
    if(order_event_ == nullptr) {
        hipEvent_t event;
        check(hipEventCreateWithFlags(&event, hipEventDisableTiming), "hipEventCreate");
        order_event_ = event;
    }
    if(order_pending_ && last_stream_ != stream) {
        check(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(order_event_), 0), "hipStreamWaitEvent");
    }
    run_stages(batch_capacity, first_read, n_read, base_planes, quality_planes, flag, result, word, stream, worklist);
    check(hipEventRecord(static_cast<hipEvent_t>(order_event_), static_cast<hipStream_t>(stream)), "hipEventRecord");
    order_pending_ = true;
    last_stream_ = stream;
}

// counters + segments: every workgroup owns ceil(tiles / grid) * 256 slots, the worst case "every read deferred"
size_t DecoderSet::worklist_words(int64_t n_read) const {
    // whatever the grid (at most max_blocks_ workgroups of at most 1024 lanes): grid * ceil(tiles / grid) * threads
    // is less than n_read + (grid + 1) * threads
    return worklist_entries(n_read) + carry_words(n_read);
}

size_t DecoderSet::worklist_entries(int64_t n_read) const {
    const size_t worst = static_cast<size_t>(n_read) + (static_cast<size_t>(max_blocks_) + 1) * 1024;
    return static_cast<size_t>(max_blocks_) + worst + static_cast<size_t>(max_blocks_) * 1024 + 512;      // + the 4 tile passes of the MDD exact kernel
}

// the inputs of deferred reads beside the first quarter of every worklist segment (StageArgs::carry), for jobs with a
// decoder the neighbour list tier takes
size_t DecoderSet::carry_words(int64_t n_read) const {
    bool any = false;
    for(const auto& d : host_decoder_) { if(d.near_ready) { any = true; } }
    if(!any) { return 0; }
    const size_t worst = static_cast<size_t>(n_read) + (static_cast<size_t>(max_blocks_) + 1) * 1024;
    return CARRY_PLANES * (worst / 4 + static_cast<size_t>(max_blocks_) * 512);
}

std::string DecoderSet::plan_chain(int64_t n_read, int32_t mode) {
    if(mode != MODE_LITERAL && mode != MODE_FAST && mode != MODE_AUTO) { throw ConfigurationError("unknown classification mode " + std::to_string(mode)); }
    if(n_read <= 0) { return "[]"; }
    if(host_decoder_.empty() && !job_.decoder.empty()) {
        if(device_ >= 0) { throw InternalError("the decoder set holds no tables"); }
        build_device_images();
    }
    struct Recorder : LaunchTimer {
        std::string out = "[";
        void begin(int32_t kind, void*, const char* name) override {
            out += std::string(out.size() > 1 ? ", " : "") + "{\"decoder\": " + std::to_string(decoder) + ", \"kernel\": \"" + name + "\", \"kind\": " + std::to_string(kind) + "}";
        }
        void end(void*) override {}
    } recorder;
    struct PlanOnly { PlanOnly() { stage_plan_only(true); } ~PlanOnly() { stage_plan_only(false); } } guard;
    // the arguments the dispatcher looks at: which buffers there are (not what they hold), the batch's size, the stage's place
    static unsigned char present[64];
    const size_t n_decoder = job_.decoder.size();
    for(size_t k = 0; k < n_decoder; ++k) {
        const DecoderConfig& c = job_.decoder[k];
        size_t folded = 0;
        if(c.algorithm == Algorithm::PAMLD || c.algorithm == Algorithm::MDD) {
            while(folded < 4 && k + folded + 1 < n_decoder &&
                  (job_.decoder[k + folded + 1].algorithm == Algorithm::PASSTHROUGH || job_.decoder[k + folded + 1].algorithm == Algorithm::NAIVE)) { ++folded; }
        }
        StageArgs args;
        std::memset(&args, 0, sizeof(args));
        const bool worklist = needs_worklist();
        args.worklist_count = worklist ? reinterpret_cast<uint32_t*>(present) : nullptr;
        args.worklist = worklist ? reinterpret_cast<uint32_t*>(present) : nullptr;
        args.carry = (worklist && carry_words(n_read) > 0) ? reinterpret_cast<uint32_t*>(present) : nullptr;
        args.decoder = device_decoder_ + k;
        args.n_read = n_read;
        args.base = c.plane_stride ? present : nullptr;
        args.quality = c.plane_stride ? present : nullptr;
        args.input_flag = present;
        args.result = c.result_slot >= 0 ? reinterpret_cast<DevResult*>(present) : nullptr;
        args.word = reinterpret_cast<uint32_t*>(present);
        args.stats = reinterpret_cast<unsigned long long*>(present);
        args.first_stage = k == 0 ? 1 : 0;
        args.job_outgoing = k + folded + 1 == n_decoder ? 1 : 0;
        args.last_stage = (args.job_outgoing || folded > 0) ? 1 : 0;
        args.echo_count = static_cast<int32_t>(folded);
        recorder.decoder = static_cast<int32_t>(k);
        const int32_t effective = mode != MODE_AUTO ? mode : static_cast<int32_t>(MODE_FAST);
        const int status = launch_stage(args, host_decoder_[k], host_decoder_[k].algorithm != 1 && mode == MODE_AUTO ? static_cast<int32_t>(MODE_FAST) : effective, max_blocks_, nullptr, &recorder);
        if(status != 0) { throw InternalError("the stage dispatcher failed on decoder " + std::to_string(k)); }
        k += folded;
    }
    return recorder.out + "]";
}

bool DecoderSet::needs_worklist() const {
    for(const auto& d : host_decoder_) { if(stage_uses_worklist(d, mode_ == MODE_LITERAL ? MODE_LITERAL : MODE_FAST)) { return true; } }
    return false;
}

// one launch (two in PHQ_MODE_FAST) per decoder, in TranscodingDecoder::classify order (transcode.h:51-60)
void DecoderSet::run_stages(int64_t batch_capacity, int64_t first_read, int64_t n_read,
                            const uint8_t* base_planes, const uint8_t* quality_planes, const uint8_t* flag,
                            DevResult* result, uint32_t* word, void* stream, uint32_t* worklist) {
    const size_t n_decoder = job_.decoder.size();
    static const bool no_fold = std::getenv("PHQ_NO_FOLD") != nullptr;
    for(size_t k = 0; k < n_decoder; ++k) {
        const DecoderConfig& c = job_.decoder[k];
        // the decoders behind this one that only count (passthrough, naive): their reads are this stage's outgoing reads, so
        // their two counts are added by this stage's kernels and their launches dropped (StageArgs::echo_offset)
        size_t folded = 0;
        if(!no_fold && (c.algorithm == Algorithm::PAMLD || c.algorithm == Algorithm::MDD)) {
            while(folded < 4 && k + folded + 1 < n_decoder &&
                  (job_.decoder[k + folded + 1].algorithm == Algorithm::PASSTHROUGH || job_.decoder[k + folded + 1].algorithm == Algorithm::NAIVE)) { ++folded; }
        }
        StageArgs args;
        std::memset(&args, 0, sizeof(args));
        args.worklist_count = worklist;                                   // one counter per workgroup ...
        args.worklist = worklist ? worklist + max_blocks_ : nullptr;      // ... then one segment per workgroup
        args.carry = (worklist && carry_words(n_read) > 0) ? worklist + worklist_entries(n_read) : nullptr;       // stage.hip sets the stride
        args.decoder = device_decoder_ + k;
        args.n_read = n_read;
        args.base = c.plane_stride ? base_planes + batch_capacity * c.plane_prefix + first_read * c.plane_stride : nullptr;
        args.quality = c.plane_stride ? quality_planes + batch_capacity * c.plane_prefix + first_read * c.plane_stride : nullptr;
        args.input_flag = flag ? flag + first_read : nullptr;
        args.result = (c.result_slot >= 0 && result != nullptr) ? result + static_cast<int64_t>(c.result_slot) * batch_capacity + first_read : nullptr;
        args.word = word + first_read;
        args.stats = device_stats_;
        args.first_stage = k == 0 ? 1 : 0;
        args.job_outgoing = k + folded + 1 == n_decoder ? 1 : 0;
        args.last_stage = (args.job_outgoing || folded > 0) ? 1 : 0;      // the stage counts its outgoing reads
        args.echo_count = static_cast<int32_t>(folded);
        for(size_t e = 0; e < folded; ++e) { args.echo_offset[e] = host_decoder_[k + 1 + e].stats_offset; }
        if(timer_) { timer_->decoder = static_cast<int32_t>(k); }
        // (MDD is integer work in either mode: only PAMLD stages turn literal for exact records)
        const int32_t stage_mode = (mode_ == MODE_AUTO && host_decoder_[k].algorithm != 1) ? static_cast<int32_t>(MODE_FAST) : effective_mode();
        check(static_cast<hipError_t>(launch_stage(args, host_decoder_[k], stage_mode, max_blocks_, stream, timer_)), "kernel launch");
        k += folded;
    }
}

void DecoderSet::ensure_pipeline(int64_t chunk_capacity) {
    if(chunk_capacity_ >= chunk_capacity && slot_[0].stream != nullptr) { return; }
    for(auto& slot : slot_) {
        if(slot.stream) { check(hipStreamSynchronize(static_cast<hipStream_t>(slot.stream)), "hipStreamSynchronize"); }
        (void)hipFree(slot.device_base); (void)hipFree(slot.device_quality); (void)hipFree(slot.device_flag);
        (void)hipFree(slot.device_result); (void)hipFree(slot.device_word); (void)hipFree(slot.device_worklist);
        (void)hipFree(slot.device_packed); (void)hipFree(slot.device_escape);
        host_free(slot.pinned_base); host_free(slot.pinned_quality); host_free(slot.pinned_flag);
        host_free(slot.pinned_result); host_free(slot.pinned_word); host_free(slot.pinned_escape);
        // nothing of the old buffers is left in the slot should an allocation below throw (release() frees what it finds)
        { Slot fresh; fresh.stream = slot.stream; fresh.done = slot.done; fresh.uploaded = slot.uploaded; fresh.consumed = slot.consumed; slot = fresh; }
        chunk_capacity_ = 0;
        const size_t planes = (static_cast<size_t>(chunk_capacity) * static_cast<size_t>(std::max(job_.plane_width, 4)) + 15) / 16 * 16;      // whole units of 16 bytes (unpack_planes)
        const size_t results = static_cast<size_t>(chunk_capacity) * static_cast<size_t>(std::max(job_.result_count, 1)) * sizeof(DevResult);
        check(hipMalloc(reinterpret_cast<void**>(&slot.device_base), planes), "hipMalloc");
        check(hipMalloc(reinterpret_cast<void**>(&slot.device_quality), planes), "hipMalloc");
        check(hipMalloc(reinterpret_cast<void**>(&slot.device_flag), static_cast<size_t>(chunk_capacity)), "hipMalloc");
        check(hipMalloc(reinterpret_cast<void**>(&slot.device_result), results), "hipMalloc");
        check(hipMalloc(reinterpret_cast<void**>(&slot.device_word), static_cast<size_t>(chunk_capacity) * sizeof(uint32_t)), "hipMalloc");
        check(hipMalloc(reinterpret_cast<void**>(&slot.device_worklist), worklist_words(chunk_capacity) * sizeof(uint32_t)), "hipMalloc");
        check(hipMalloc(reinterpret_cast<void**>(&slot.device_packed), planes), "hipMalloc");
        check(hipMalloc(reinterpret_cast<void**>(&slot.device_escape), static_cast<size_t>(chunk_capacity) * sizeof(uint64_t)), "hipMalloc");
        slot.pinned_escape = static_cast<uint64_t*>(host_alloc(static_cast<size_t>(chunk_capacity) * sizeof(uint64_t)));
        slot.pinned_base = static_cast<uint8_t*>(host_alloc(planes));
        slot.pinned_quality = static_cast<uint8_t*>(host_alloc(planes));
        slot.pinned_flag = static_cast<uint8_t*>(host_alloc(static_cast<size_t>(chunk_capacity)));
        slot.pinned_result = static_cast<DevResult*>(host_alloc(results));
        slot.pinned_word = static_cast<uint32_t*>(host_alloc(static_cast<size_t>(chunk_capacity) * sizeof(uint32_t)));
        if(slot.stream == nullptr) {
            hipStream_t stream;
            check(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
            slot.stream = stream;
            hipEvent_t event;
            check(hipEventCreateWithFlags(&event, hipEventDisableTiming), "hipEventCreate");
            slot.done = event;
            check(hipEventCreateWithFlags(&event, hipEventDisableTiming), "hipEventCreate");
            slot.uploaded = event;
            check(hipEventCreateWithFlags(&event, hipEventDisableTiming), "hipEventCreate");
            slot.consumed = event;
        }
        slot.busy = false;
    }
    if(upload_stream_ == nullptr) {
        hipStream_t stream;
        check(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
        upload_stream_ = stream;
    }
    chunk_capacity_ = chunk_capacity;
}

// Host batch -> pinned slot -> device -> pinned slot -> host batch (the reference's per-read Transcode::pull /
// Multiplexer::push hand-offs, transcode.cpp:287-316 and multiplex.h:198-222, become two async copies per chunk).
// The link moves bytes BOTH ways at once (profiles/tools/micro/pcie_duplex.py: 57 GB/s either way alone, 96 GB/s together),
// so the pipeline is laid out for that: every chunk's uploads go through ONE stream, one after the other at the link's full
// rate, and a chunk's kernels and downloads through its slot's stream behind an event — while chunk k comes down, chunk
// k + 1 goes up.  (Round 3 had each slot upload, compute and download on its own stream: two chunks then ran in step, both
// up and then both down, and the two directions took turns: 55 GB/s for both together.)  Three slots of 2 Mi reads.
void DecoderSet::classify_host(int64_t batch_capacity, int64_t n_read, const uint8_t* base_planes, const uint8_t* quality_planes, const uint8_t* flag,
                               DevResult* result, uint32_t* word) {
    classify_host_planes(batch_capacity, n_read, base_planes, quality_planes, nullptr, nullptr, 0, flag, result, word);
}

// The same call for planes of ONE byte per nucleotide (unpack.hpp): 9 instead of 17 bytes per read of an 8 nt index go up
// the link; a chunk's packed planes are expanded into the slot's observation planes by a kernel in front of the stages,
// the few nucleotides that do not pack are patched in from the list beside them.
void DecoderSet::classify_host_packed(int64_t batch_capacity, int64_t n_read, const uint8_t* packed_planes, const uint64_t* escapes, int64_t n_escape,
                                      const uint8_t* flag, DevResult* result, uint32_t* word) {
    if(packed_planes == nullptr) { throw InternalError("packed_planes is NULL"); }
    if(n_escape < 0 || (n_escape > 0 && escapes == nullptr)) { throw InternalError("escapes"); }
    for(int64_t i = 1; i < n_escape; ++i) {
        if((escapes[i] >> 32) < (escapes[i - 1] >> 32)) { throw InternalError("the escape list is not in read order"); }
    }
    classify_host_planes(batch_capacity, n_read, nullptr, nullptr, packed_planes, escapes, n_escape, flag, result, word);
}

int64_t DecoderSet::pack_planes(int64_t batch_capacity, int64_t n_read, const uint8_t* base_planes, const uint8_t* quality_planes, uint8_t* packed_planes,
                                uint64_t* escapes, int64_t escape_capacity) const {
    if(n_read <= 0) { return 0; }
    if(n_read > batch_capacity) { throw OverflowError("read count exceeds the batch capacity"); }
    if(job_.plane_width > 0xffff) { throw OverflowError("plane width"); }
    const int parts = host_threads();
    std::vector<std::vector<uint64_t>> found(static_cast<size_t>(parts));
    parallel_ranges(n_read, 1 << 14, parts, [&](int64_t first, int64_t last, int part) {
        std::vector<uint64_t>& mine = found[static_cast<size_t>(part)];
        for(int64_t k = first; k < last; ++k) {
            for(const auto& c : job_.decoder) {
                if(c.plane_stride == 0) { continue; }
                const int64_t row = batch_capacity * c.plane_prefix + k * c.plane_stride;
                for(int32_t j = 0; j < c.plane_stride; ++j) {
                    const uint8_t code = base_planes[row + j], quality = quality_planes[row + j];
                    const uint8_t packed = pack_nucleotide(code, quality);
                    packed_planes[row + j] = packed;
                    if(packed == PACKED_ESCAPE) { mine.push_back(escape_entry(k, c.plane_prefix + j, code, quality)); }
                }
            }
        }
    });
    int64_t total = 0;
    for(const auto& part : found) { total += static_cast<int64_t>(part.size()); }
    if(total > escape_capacity || (total > 0 && escapes == nullptr)) { return total; }
    int64_t at = 0;
    for(const auto& part : found) {                        // the parts are ranges of reads in order: so is the list
        if(!part.empty()) { std::memcpy(escapes + at, part.data(), part.size() * sizeof(uint64_t)); }
        at += static_cast<int64_t>(part.size());
    }
    return total;
}

void DecoderSet::classify_host_planes(int64_t batch_capacity, int64_t n_read, const uint8_t* base_planes, const uint8_t* quality_planes, const uint8_t* packed_planes,
                                      const uint64_t* escapes, int64_t n_escape, const uint8_t* flag, DevResult* result, uint32_t* word) {
    require_device();
    if(n_read <= 0) { return; }
    if(n_read > batch_capacity) { throw OverflowError("read count exceeds the batch capacity"); }
    static const int64_t chunk_reads = []() { const char* text = std::getenv("PHQ_HOST_CHUNK"); const int64_t asked = text ? std::atoll(text) : 0; return asked >= 4096 ? asked : int64_t(1) << 21; }();      // sweeps
    const int64_t chunk = std::min<int64_t>(n_read, chunk_reads);
    ensure_pipeline(chunk);
    hipStream_t upload = static_cast<hipStream_t>(upload_stream_);
    // batches that already live in pinned memory (phq_host_alloc) are copied straight from / to the caller's
    // buffers; pageable ones bounce through the slot's pinned buffers
    auto pinned = [](const void* pointer) {
        if(pointer == nullptr) { return true; }
        hipPointerAttribute_t attributes;
        if(hipPointerGetAttributes(&attributes, pointer) != hipSuccess) { (void)hipGetLastError(); return false; }
        return attributes.type == hipMemoryTypeHost;
    };
    const bool direct_in = pinned(base_planes) && pinned(quality_planes) && pinned(packed_planes) && pinned(flag);
    int32_t plane_prefix[UNPACK_MAX_DECODER] = {0}, plane_stride[UNPACK_MAX_DECODER] = {0};
    if(packed_planes != nullptr) {
        if(job_.decoder.size() > static_cast<size_t>(UNPACK_MAX_DECODER)) { throw OverflowError("too many decoders for packed planes"); }
        for(size_t d = 0; d < job_.decoder.size(); ++d) { plane_prefix[d] = job_.decoder[d].plane_prefix; plane_stride[d] = job_.decoder[d].plane_stride; }
    }
    int64_t escape_at = 0;                          // the list is in read order: a chunk's entries follow the previous chunk's
    const bool direct_out = pinned(result) && pinned(word);
    if(packed_planes != nullptr) {
        // PHQ_OVERFLOW_ERROR is the documented, recoverable answer to a batch that does not pack (the caller falls back to
        // phq_classify_host): it is given BEFORE anything is enqueued, so no chunk of this call is in flight when it is
        int64_t at = 0;
        for(int64_t first = 0; first < n_read; first += chunk) {
            const int64_t end = first + std::min(chunk, n_read - first);
            int64_t in_chunk = 0;
            while(at < n_escape && static_cast<int64_t>(escapes[at] >> 32) < end) { ++at; ++in_chunk; }
            if(in_chunk > chunk_capacity_) { throw OverflowError("more nucleotides outside the packed form than reads in a chunk: use phq_classify_host"); }
        }
    }
    // whatever else is thrown part way (a HIP error): nothing of this call stays in flight behind it — the caller's buffers
    // are not written after the call has returned, and the next call finds every slot idle
    struct Quiesce {
        DecoderSet* set; hipStream_t upload; bool armed;
        ~Quiesce() {
            if(!armed) { return; }
            (void)hipStreamSynchronize(upload);
            for(auto& slot : set->slot_) {
                if(slot.stream) { (void)hipStreamSynchronize(static_cast<hipStream_t>(slot.stream)); }
                slot.busy = false;
            }
        }
    } quiesce{this, upload, true};

    auto drain = [&](Slot& slot) {
        if(!slot.busy) { return; }
        check(hipEventSynchronize(static_cast<hipEvent_t>(slot.done)), "hipEventSynchronize");
        if(direct_out) { slot.busy = false; return; }
        for(const auto& c : job_.decoder) {
            if(c.result_slot >= 0 && result != nullptr) {
                host_copy(result + static_cast<int64_t>(c.result_slot) * batch_capacity + slot.first,
                            slot.pinned_result + static_cast<int64_t>(c.result_slot) * chunk_capacity_,
                            static_cast<size_t>(slot.count) * sizeof(DevResult));
            }
        }
        if(word != nullptr) { host_copy(word + slot.first, slot.pinned_word, static_cast<size_t>(slot.count) * sizeof(uint32_t)); }
        slot.busy = false;
    };

    int which = 0;
    for(int64_t first = 0; first < n_read; first += chunk, which = (which + 1) % SLOTS) {
        Slot& slot = slot_[which];
        drain(slot);
        const int64_t count = std::min(chunk, n_read - first);
        hipStream_t stream = static_cast<hipStream_t>(slot.stream);
        // the slot's device planes are free once its previous chunk's kernels are through (the host has waited for that
        // chunk's downloads just above; the event orders the two streams on the device as well)
        if(slot.used) { check(hipStreamWaitEvent(upload, static_cast<hipEvent_t>(slot.consumed), 0), "hipStreamWaitEvent"); }
        for(const auto& c : job_.decoder) {
            if(c.plane_stride == 0) { continue; }
            const size_t bytes = static_cast<size_t>(count) * static_cast<size_t>(c.plane_stride);
            const int64_t source = batch_capacity * c.plane_prefix + first * c.plane_stride;
            const int64_t target = chunk_capacity_ * c.plane_prefix;
            if(packed_planes != nullptr) {
                const uint8_t* from = packed_planes + source;
                if(!direct_in) { host_copy(slot.pinned_base + target, from, bytes); from = slot.pinned_base + target; }
                check(hipMemcpyAsync(slot.device_packed + target, from, bytes, hipMemcpyHostToDevice, upload), "hipMemcpyAsync");
            } else if(direct_in) {
                check(hipMemcpyAsync(slot.device_base + target, base_planes + source, bytes, hipMemcpyHostToDevice, upload), "hipMemcpyAsync");
                check(hipMemcpyAsync(slot.device_quality + target, quality_planes + source, bytes, hipMemcpyHostToDevice, upload), "hipMemcpyAsync");
            } else {
                host_copy(slot.pinned_base + target, base_planes + source, bytes);
                host_copy(slot.pinned_quality + target, quality_planes + source, bytes);
                check(hipMemcpyAsync(slot.device_base + target, slot.pinned_base + target, bytes, hipMemcpyHostToDevice, upload), "hipMemcpyAsync");
                check(hipMemcpyAsync(slot.device_quality + target, slot.pinned_quality + target, bytes, hipMemcpyHostToDevice, upload), "hipMemcpyAsync");
            }
        }
        if(flag != nullptr) {
            if(direct_in) {
                check(hipMemcpyAsync(slot.device_flag, flag + first, static_cast<size_t>(count), hipMemcpyHostToDevice, upload), "hipMemcpyAsync");
            } else {
                host_copy(slot.pinned_flag, flag + first, static_cast<size_t>(count));
                check(hipMemcpyAsync(slot.device_flag, slot.pinned_flag, static_cast<size_t>(count), hipMemcpyHostToDevice, upload), "hipMemcpyAsync");
            }
        }
        int64_t chunk_escapes = 0;
        if(packed_planes != nullptr) {
            while(escape_at + chunk_escapes < n_escape && static_cast<int64_t>(escapes[escape_at + chunk_escapes] >> 32) < first + count) { ++chunk_escapes; }
            if(chunk_escapes > chunk_capacity_) { throw OverflowError("more nucleotides outside the packed form than reads in a chunk: use phq_classify_host"); }
            if(chunk_escapes > 0) {
                std::memcpy(slot.pinned_escape, escapes + escape_at, static_cast<size_t>(chunk_escapes) * sizeof(uint64_t));
                check(hipMemcpyAsync(slot.device_escape, slot.pinned_escape, static_cast<size_t>(chunk_escapes) * sizeof(uint64_t), hipMemcpyHostToDevice, upload), "hipMemcpyAsync");
            }
        }
        check(hipEventRecord(static_cast<hipEvent_t>(slot.uploaded), upload), "hipEventRecord");
        check(hipStreamWaitEvent(stream, static_cast<hipEvent_t>(slot.uploaded), 0), "hipStreamWaitEvent");
        if(packed_planes != nullptr) {
            const size_t planes = static_cast<size_t>(chunk_capacity_) * static_cast<size_t>(std::max(job_.plane_width, 4));
            check(static_cast<hipError_t>(unpack_planes(slot.device_packed, slot.device_base, slot.device_quality, planes, stream)), "unpack_planes");
            check(static_cast<hipError_t>(patch_escapes(slot.device_escape, chunk_escapes, first, chunk_capacity_, static_cast<int32_t>(job_.decoder.size()), plane_prefix, plane_stride,
                                                        slot.device_base, slot.device_quality, stream)), "patch_escapes");
            escape_at += chunk_escapes;
        }
        run_stages(chunk_capacity_, 0, count, slot.device_base, slot.device_quality, flag ? slot.device_flag : nullptr,
                   slot.device_result, slot.device_word, stream, slot.device_worklist);
        check(hipEventRecord(static_cast<hipEvent_t>(slot.consumed), stream), "hipEventRecord");
        slot.used = true;
        for(const auto& c : job_.decoder) {
            if(c.result_slot >= 0 && result != nullptr) {
                const int64_t plane = static_cast<int64_t>(c.result_slot) * chunk_capacity_;
                DevResult* target = direct_out ? result + static_cast<int64_t>(c.result_slot) * batch_capacity + first : slot.pinned_result + plane;
                check(hipMemcpyAsync(target, slot.device_result + plane, static_cast<size_t>(count) * sizeof(DevResult),
                                     hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
            }
        }
        if(word != nullptr || !direct_out) {
            check(hipMemcpyAsync(direct_out ? word + first : slot.pinned_word, slot.device_word, static_cast<size_t>(count) * sizeof(uint32_t),
                                 hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
        }
        check(hipEventRecord(static_cast<hipEvent_t>(slot.done), stream), "hipEventRecord");
        slot.busy = true;
        slot.first = first;
        slot.count = count;
    }
    for(auto& slot : slot_) { drain(slot); }
    quiesce.armed = false;
}

// ---- statistics ----------------------------------------------------------------------------------------

void DecoderSet::stats_reset(void* stream) {
    qc_ = QcState();                       // the quality control accumulators go with the counters
    require_device();
    check(hipMemsetAsync(device_stats_, 0, static_cast<size_t>(job_.stats_words) * sizeof(uint64_t), static_cast<hipStream_t>(stream)), "hipMemsetAsync");
}

uint64_t* DecoderSet::stats_device_pointer() {
    require_device();
    return reinterpret_cast<uint64_t*>(device_stats_);
}

void DecoderSet::stats_export(uint64_t* host_words) {
    require_device();
    check(hipDeviceSynchronize(), "hipDeviceSynchronize");
    check(hipMemcpy(host_words, device_stats_, static_cast<size_t>(job_.stats_words) * sizeof(uint64_t), hipMemcpyDeviceToHost), "hipMemcpy");
}

void DecoderSet::stats_import(const uint64_t* host_words) {
    require_device();
    check(hipDeviceSynchronize(), "hipDeviceSynchronize");
    check(hipMemcpy(device_stats_, host_words, static_cast<size_t>(job_.stats_words) * sizeof(uint64_t), hipMemcpyHostToDevice), "hipMemcpy");
}

void DecoderSet::stats_take(DecoderSet& other) {
    if(other.job_.stats_words != job_.stats_words) { throw ConfigurationError("cannot merge statistics of different jobs"); }
    std::vector<uint64_t> mine(static_cast<size_t>(job_.stats_words)), theirs(mine.size());
    other.stats_export(theirs.data());
    stats_export(mine.data());
    for(size_t i = 0; i < mine.size(); ++i) { mine[i] += theirs[i]; }
    stats_import(mine.data());
    std::fill(theirs.begin(), theirs.end(), 0);
    other.stats_import(theirs.data());
}

// RCCL is resolved at first use so that single GPU users never load it.
void DecoderSet::stats_allreduce(void* rccl_comm, void* stream) {
    require_device();
    typedef int (*all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static all_reduce_t all_reduce = nullptr;
    if(all_reduce == nullptr) {
        void* library = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if(library == nullptr) { library = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL); }
        if(library == nullptr) { throw InternalError(std::string("cannot load librccl.so : ") + dlerror()); }
        all_reduce = reinterpret_cast<all_reduce_t>(dlsym(library, "ncclAllReduce"));
        if(all_reduce == nullptr) { throw InternalError("librccl.so has no ncclAllReduce"); }
    }
    constexpr int NCCL_UINT64 = 5;     // ncclUint64 (rccl.h ncclDataType_t)
    constexpr int NCCL_SUM = 0;        // ncclSum
    const int status = all_reduce(device_stats_, device_stats_, static_cast<size_t>(job_.stats_words), NCCL_UINT64, NCCL_SUM, rccl_comm, static_cast<hipStream_t>(stream));
    if(status != 0) { throw InternalError("ncclAllReduce failed with code " + std::to_string(status)); }
}

// Read::update_* / set_* as the decoders' wrappers call them, after the last decoder (pamld.cpp:133-180, mdd.cpp:96-138)
void summarize_read(const JobConfig& job, const DevResult* result, int64_t batch_capacity, int64_t k, uint32_t (&distance)[3], double (&confidence)[3]) {
    for(int t = 0; t < 3; ++t) { distance[t] = 0; confidence[t] = 1; }               // read.h:176-183
    for(const auto& d : job.decoder) {
        if(d.result_slot < 0) { continue; }                                          // naive / passthrough: raw barcode only
        const DevResult& res = result[static_cast<int64_t>(d.result_slot) * batch_capacity + k];
        const int kind = static_cast<int>(d.kind);
        const bool classified = res.index > 0;
        if(d.algorithm == Algorithm::PAMLD) {
            if(d.kind == Kind::SAMPLE || classified) {
                distance[kind] += res.distance;
                confidence[kind] = confidence[kind] == 1 ? res.confidence : confidence[kind] * res.confidence;     // read.h:279-285
            } else {
                confidence[kind] = 0;
                if(d.kind == Kind::CELLULAR) { distance[kind] = 0; }                  // read.h:319-321: the molecular setter adds
            }
        } else {
            if(d.kind == Kind::SAMPLE || classified) { distance[kind] += res.distance; }
            else { distance[static_cast<int>(Kind::CELLULAR)] = 0; }                   // mdd.cpp:118,136: both reset the cellular one
        }
    }
}

const DecoderSet::HostStage& DecoderSet::host_stage(int64_t capacity) {
    if(host_stage_.capacity < capacity) {
        release_host_stage();
        // pinned for a set with a device; a host only set (configuration, staging, the sanitizer build) takes plain memory
        const bool pinned = device_ >= 0;
        auto take = [&](size_t bytes) -> void* {
            if(pinned) { return host_alloc(bytes); }
            void* pointer = std::malloc(bytes > 0 ? bytes : 16);
            if(pointer == nullptr) { throw OutOfMemoryError("host staging planes"); }
            return pointer;
        };
        const size_t planes = static_cast<size_t>(capacity) * static_cast<size_t>(std::max(job_.plane_width, 4));
        host_stage_.pinned = pinned;
        host_stage_.base = static_cast<uint8_t*>(take(planes));
        host_stage_.quality = static_cast<uint8_t*>(take(planes));
        host_stage_.flag = static_cast<uint8_t*>(take(static_cast<size_t>(capacity)));
        host_stage_.result = static_cast<DevResult*>(take(static_cast<size_t>(capacity) * static_cast<size_t>(std::max(job_.result_count, 1)) * sizeof(DevResult)));
        host_stage_.word = static_cast<uint32_t*>(take(static_cast<size_t>(capacity) * sizeof(uint32_t)));
        host_stage_.capacity = capacity;
    }
    return host_stage_;
}

void DecoderSet::release_host_stage() {
    void* const all[] = {host_stage_.base, host_stage_.quality, host_stage_.flag, host_stage_.result, host_stage_.word};
    for(void* pointer : all) { if(host_stage_.pinned) { host_free(pointer); } else { std::free(pointer); } }
    host_stage_ = HostStage();
}

void DecoderSet::summarize(int64_t batch_capacity, int64_t n_read, const DevResult* result, uint32_t* distance, double* confidence) const {
    if(n_read < 0 || n_read > batch_capacity) { throw OverflowError("read range exceeds the batch capacity"); }
    if(result == nullptr && job_.result_count > 0) { throw InternalError("result planes are missing"); }
    for(int64_t k = 0; k < n_read; ++k) {
        uint32_t d[3];
        double c[3];
        summarize_read(job_, result, batch_capacity, k, d, c);
        for(int t = 0; t < 3; ++t) {
            if(distance) { distance[3 * k + t] = d[t]; }
            if(confidence) { confidence[3 * k + t] = c[t]; }
        }
    }
}

const std::string& DecoderSet::report_json(const uint64_t* host_words, uint64_t incoming_count, uint64_t incoming_pf_count) {
    Json report = encode_report(job_, host_words, incoming_count, incoming_pf_count);
    if(job_.enable_quality_control) {                                          // transcode.cpp:1830-1833, multiplex.cpp:419-431
        if(qc_.empty()) { qc_.shape(job_.multiplexing_decoder >= 0 ? static_cast<int32_t>(job_.decoder[static_cast<size_t>(job_.multiplexing_decoder)].codec.size()) + 1 : 1, job_.output_segment_cardinality); }
        report.set("multiplex", qc_.encode());
        report.sort_keys();
    }
    report_ = report.dump(job_.float_precision);
    return report_;
}

void DecoderSet::quality_control_merge(const QcState& other) { qc_.merge(other); }

}  // namespace phq
