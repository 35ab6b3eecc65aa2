// FASTQ in -> classification -> SAM text out, around one DecoderSet.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "decoder_set.hpp"

namespace phq {

class Transcoder {
public:
    // one input path per input segment (plain or gzip FASTQ); sam_output "-" = stdout
    Transcoder(DecoderSet& set, const std::vector<std::string>& input, const std::string& sam_output, int64_t batch_capacity);
    ~Transcoder();
    Transcoder(const Transcoder&) = delete;
    Transcoder& operator=(const Transcoder&) = delete;
    // runs to the end of the input; adds every read pulled to the incoming counters (transcode.cpp:304-309)
    void run(uint64_t& incoming_count, uint64_t& incoming_pf_count);

private:
    struct Impl;
    Impl* impl_;
};

}  // namespace phq
