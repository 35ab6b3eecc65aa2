// DecoderSet: the MI355X stand-in for one TranscodingDecoder (reference transcode.h:40-77).
//
// It owns what the reference's decoders own — deep copies of the barcode tables, priors and thresholds
// (classifier.h:68-77) — as immutable device images, plus the device counter buffer that plays the role
// of every AccumulatingOption.  One set serves one GPU; it is single-caller like the reference's
// per-thread decoder copies (transcode.cpp:1769-1775).
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "device_types.hpp"
#include "job.hpp"
#include "qc.hpp"
#include "json.hpp"

namespace phq {

class DecoderSet {
public:
    DecoderSet(const std::string& compiled_job_json, int32_t device);
    ~DecoderSet();
    DecoderSet(const DecoderSet&) = delete;
    DecoderSet& operator=(const DecoderSet&) = delete;

    const JobConfig& job() const { return job_; }
    int32_t device() const { return device_; }
    void set_mode(int32_t mode);
    int32_t mode() const { return mode_; }
    // what MODE_AUTO means right now: literal while a transcoding run that writes SAM records of a small job holds the set (device_types.hpp)
    int32_t effective_mode() const { return mode_ != MODE_AUTO ? mode_ : records_exact_ ? MODE_LITERAL : MODE_FAST; }
    void set_records_exact(bool on) { records_exact_ = on; }
    int64_t literal_cells_per_read() const;
    // the kernels a batch of n_read reads would take in `mode`, decoder by decoder, as the stage dispatcher names them — decided by
    // the dispatcher itself with its launches switched off, so this is the chain phq_profile_chain reports after a run, without
    // one; works on a set without a device (the tables' images are then built in host memory, for their presence alone)
    std::string plan_chain(int64_t n_read, int32_t mode);

    // Rule::apply + pre-filters + incoming counters (see include/pheniqs_amd.h phq_stage_reads)
    void stage_reads(int64_t n_read, int32_t n_segment, const uint8_t* const* code, const uint8_t* const* quality,
                     const int64_t* const* offset, const int32_t* const* length, const uint8_t* in_flag,
                     int64_t batch_capacity, uint8_t* base_planes, uint8_t* quality_planes,
                     uint8_t* out_flag, int64_t* kept, int64_t* n_kept,
                     uint64_t* incoming_count, uint64_t* incoming_pf_count) const;

    void classify_device(int64_t batch_capacity, int64_t first_read, int64_t n_read,
                         const uint8_t* base_planes, const uint8_t* quality_planes, const uint8_t* flag,
                         DevResult* result, uint32_t* word, void* stream);
    void classify_host(int64_t batch_capacity, int64_t n_read, const uint8_t* base_planes, const uint8_t* quality_planes, const uint8_t* flag,
                       DevResult* result, uint32_t* word);
    // one byte per nucleotide across the link (unpack.hpp): pack_planes returns the number of escape entries it needs — more than
    // escape_capacity: nothing but the count is valid; classify_host_packed is classify_host for planes packed that way
    int64_t pack_planes(int64_t batch_capacity, int64_t n_read, const uint8_t* base_planes, const uint8_t* quality_planes, uint8_t* packed_planes,
                        uint64_t* escapes, int64_t escape_capacity) const;
    void classify_host_packed(int64_t batch_capacity, int64_t n_read, const uint8_t* packed_planes, const uint64_t* escapes, int64_t n_escape,
                              const uint8_t* flag, DevResult* result, uint32_t* word);

    void stats_reset(void* stream);
    uint64_t* stats_device_pointer();
    void stats_export(uint64_t* host_words);
    void stats_take(DecoderSet& other);           // this += other, other = 0 (selector.cpp:68-77: operator+= of every accumulator)
    void stats_import(const uint64_t* host_words);
    void stats_allreduce(void* rccl_comm, void* stream);
    void summarize(int64_t batch_capacity, int64_t n_read, const DevResult* result, uint32_t* distance, double* confidence) const;
    // pinned host planes for the transcoder's batches, kept by the set from one run to the next (pinning tens of
    // megabytes costs more than classifying a batch)
    struct HostStage { int64_t capacity = 0; bool pinned = true; uint8_t* base = nullptr; uint8_t* quality = nullptr; uint8_t* flag = nullptr; DevResult* result = nullptr; uint32_t* word = nullptr; };
    const HostStage& host_stage(int64_t capacity);
    void release_host_stage();
    const std::string& report_json(const uint64_t* host_words, uint64_t incoming_count, uint64_t incoming_pf_count);

    // HIP event timing of every kernel launch (bench.py roofline): enable, run, then collect sums per kernel kind
    void profile_enable(bool enable);
    void profile_collect(double* milliseconds, int64_t* launches);
    std::string profile_chain();

    // quality control accumulators of the output reads (qc.hpp), filled by the transcoder's record threads when the job
    // enables them; cleared with the counters (stats_reset), merged across sets or processes through their flat image
    QcState& quality_control() { return qc_; }
    void quality_control_merge(const QcState& other);
    std::string last_error;
    std::shared_ptr<void> device_fastq_cache;      // transcoder.cpp: the buffers of the device FASTQ path, kept from run to run

private:
    QcState qc_;
    void require_device() const;
    void run_stages(int64_t batch_capacity, int64_t first_read, int64_t n_read,
                    const uint8_t* base_planes, const uint8_t* quality_planes, const uint8_t* flag,
                    DevResult* result, uint32_t* word, void* stream, uint32_t* worklist);
    bool needs_worklist() const;
    size_t worklist_words(int64_t n_read) const;
    size_t worklist_entries(int64_t n_read) const;
    size_t carry_words(int64_t n_read) const;
    void build_device_images();
    void release();
    void ensure_pipeline(int64_t chunk_capacity);

    JobConfig job_;
    int32_t device_;
    int32_t mode_;
    bool records_exact_ = false;
    int32_t max_blocks_;
    std::vector<DevDecoder> host_decoder_;      // host copies holding DEVICE table pointers
    DevDecoder* device_decoder_;                // device array, one per decoder
    std::vector<void*> device_allocation_;
    std::vector<void*> host_image_;             // a set without a device: the tables of build_device_images in host memory (plan_chain)
    unsigned long long* device_stats_;
    HostStage host_stage_;
    void* order_event_ = nullptr;               // recorded behind the kernels of the last phq_classify_device call
    void* last_stream_ = nullptr;
    bool order_pending_ = false;
    uint32_t* device_worklist_;                 // read ordinals deferred by the product form kernel + 1 counter word in front
    int64_t worklist_capacity_;
    std::string report_;
    struct EventTimer;
    EventTimer* timer_;

    // host staged pipeline (classify_host): two slots of device buffers + pinned bounce buffers
    struct Slot {
        uint8_t* device_base = nullptr;
        uint8_t* device_quality = nullptr;
        uint8_t* device_flag = nullptr;
        DevResult* device_result = nullptr;
        uint32_t* device_word = nullptr;
        uint8_t* pinned_base = nullptr;
        uint8_t* pinned_quality = nullptr;
        uint8_t* pinned_flag = nullptr;
        DevResult* pinned_result = nullptr;
        uint32_t* pinned_word = nullptr;
        uint32_t* device_worklist = nullptr;
        uint8_t* device_packed = nullptr;       // classify_host_packed: the chunk's planes as they cross the link
        uint64_t* device_escape = nullptr;      // ... and the nucleotides that do not pack, chunk_capacity entries at most
        uint64_t* pinned_escape = nullptr;
        void* stream = nullptr;
        void* done = nullptr;                   // hipEvent_t recorded after the slot's last D2H
        void* uploaded = nullptr;               // ... after its chunk's last H2D, on the upload stream
        void* consumed = nullptr;               // ... after its chunk's kernels: the device planes may take the next chunk
        bool busy = false;
        bool used = false;                      // a chunk has gone through the slot: `consumed` has been recorded
        int64_t first = 0;
        int64_t count = 0;
    };
    void classify_host_planes(int64_t batch_capacity, int64_t n_read, const uint8_t* base_planes, const uint8_t* quality_planes, const uint8_t* packed_planes,
                              const uint64_t* escapes, int64_t n_escape, const uint8_t* flag, DevResult* result, uint32_t* word);
    static constexpr int SLOTS = 3;
    Slot slot_[SLOTS];
    void* upload_stream_ = nullptr;             // every chunk's host to device copies, one after the other
    int64_t chunk_capacity_;
};

void* host_alloc(size_t bytes);   // pinned host memory
int copy_bandwidth(int device, size_t bytes, int repeats, double* gigabytes_per_second);      // copy_probe.hip: a hipError_t value
void host_free(void* pointer);

// the per type accumulators of one read (read.h:152-165) from its result records; shared with the transcoder
void summarize_read(const JobConfig& job, const DevResult* result, int64_t batch_capacity, int64_t k, uint32_t (&distance)[3], double (&confidence)[3]);

}  // namespace phq
