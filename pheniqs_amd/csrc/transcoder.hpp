// FASTQ in -> classification -> FASTQ / SAM text out, around one DecoderSet.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "decoder_set.hpp"
#include "url.hpp"

namespace phq {

// What Transcode::compile_input / compile_output leave in `feed` and on the channels (transcode.cpp:683-733,1261-1417):
// the input feeds with their resolution, the output feeds, and for every channel of the multiplexing decoder the
// feed each output segment is pushed to (multiplex.h:198-240).
struct TranscodePlan {
    struct InputFeed { std::string path; int32_t resolution; };
    struct OutputFeed {
        std::string path;
        FileFormat format;
        bool gzip;
        int32_t level;
        int32_t resolution;                                    // output segments of one read pushed to this feed
    };
    struct Channel { std::vector<int32_t> feed_by_segment; bool filter_outgoing_qc_fail; };

    std::vector<InputFeed> input_feed;
    std::vector<std::pair<int32_t, int32_t>> input_segment;    // (feed, slot inside its resolution) per input segment
    std::vector<OutputFeed> output_feed;
    std::vector<Channel> channel;                              // 0 = undetermined, then the codec of the multiplexing decoder in order
    bool sam_header;

    // the job's own `input` and per channel `output` directives
    static TranscodePlan from_job(const JobConfig& job);
    // one FASTQ file per input segment, every channel into one SAM file ("-" = stdout)
    static TranscodePlan explicit_files(const JobConfig& job, const std::vector<std::string>& input, const std::string& sam_output);
};

// rounds and reads decoded on the device, bytes of records assembled there, runs handed to the host reader part way: since the process started
void device_path_counters(uint64_t* counters);

struct DeviceText;                     // what a run that writes its records on the device needs (transcoder.cpp, record_device.hpp)

class Transcoder {
public:
    Transcoder(DecoderSet& set, const TranscodePlan& plan, int64_t batch_capacity);
    ~Transcoder();
    Transcoder(const Transcoder&) = delete;
    Transcoder& operator=(const Transcoder&) = delete;
    // runs to the end of the input; adds every read pulled to the incoming counters (transcode.cpp:304-309)
    void run(uint64_t& incoming_count, uint64_t& incoming_pf_count);
    // A seam for the sanitizer build (check/transcoder_check.cpp): what fills the result planes and status words of a
    // staged batch.  The product never sets it — a batch is classified by DecoderSet::classify_host on the GPU, and a set
    // without a device fails there; the check injects planes of its own so that the reader and the record writers run
    // under AddressSanitizer, where no GPU code can.
    typedef std::function<void(int64_t capacity, int64_t n_read, const uint8_t* base, const uint8_t* quality, const uint8_t* flag, DevResult* result, uint32_t* word)> Classifier;
    void set_classifier(Classifier classifier);
    // a further copy of the decoder set (same job), typically on another device: batches are dealt round robin to the set
    // and its copies, whose counters are added into the set's (and zeroed) when the run ends.  The report only path that
    // decodes FASTQ text on the device (run_on_device) stays on the set's own device.
    void add_copy(DecoderSet& copy);

private:
    struct Impl;
    Impl* impl_;
    // FASTQ text decoded — and, with `text`, the output records assembled — on the device (fastq_device.hpp, record_device.hpp).
    // NOT_TAKEN: the run goes through the host reader from its start, nothing was counted or written; DONE: the whole input
    // went through the device; RESUMED: the host reader continues at the byte of every input file the device path stopped at
    // (an input that turns out of the ordinary part way into a run that writes).
    enum class OnDevice { NOT_TAKEN, DONE, RESUMED };
    OnDevice run_on_device(uint64_t& incoming_count, uint64_t& incoming_pf_count, const DeviceText* text);
};

}  // namespace phq
