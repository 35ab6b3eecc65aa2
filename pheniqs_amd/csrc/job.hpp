// Compiled job model: what the reference's decoder constructors read from the compiled ontology.
//
//   Classifier<T>      index, undetermined, codec, multiplexing classifier, corrected quality   classifier.h:55-60
//   Decoder<T>         transform, nucleotide cardinality, high quality threshold,
//                      high quality distance threshold, segment cardinality                     decoder.h:46-50
//   PamlDecoder<T>     noise, confidence threshold, random barcode probability                  pamld.cpp:26-28
//   MdDecoder<T>       quality masking threshold, distance tolerance                            mdd.cpp:26-27
//   Barcode            segment cardinality, index, concentration, barcode[]                     barcode.cpp:25-41
//   TranscodingDecoder algorithm switch per topic                                               transcode.cpp:62-161
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "device_types.hpp"
#include "json.hpp"

namespace phq {

constexpr int MAX_SEGMENT = DEV_MAX_SEGMENT;       // observation segments per decoder
constexpr int MAX_NUCLEOTIDE = DEV_MAX_NUCLEOTIDE;   // nucleotide cardinality of a PAMLD / MDD decoder the kernels unroll for

enum class Algorithm : int32_t { PASSTHROUGH = 0, PAMLD = 1, MDD = 2, NAIVE = 3 };   // atom.h:348-355 subset
enum class Kind : int32_t { SAMPLE = 0, MOLECULAR = 1, CELLULAR = 2 };                // classifier.h:29-34

uint8_t ascii_to_bam(uint8_t c);              // iupac.h:153-171
uint8_t reverse_complement_bam(uint8_t c);    // iupac.h:107-124
char bam_to_ascii(uint8_t c);                 // iupac.h:62-79
inline bool is_strict_bam(uint8_t c) { return c == 1 || c == 2 || c == 4 || c == 8; }

// Token + Transform flattened (transform.h:35-118)
struct TransformConfig {
    int32_t input_segment_index;
    int32_t start;
    int32_t end;
    bool end_terminated;
    int32_t output_segment_index;
    bool reverse_complement;
    int32_t absolute_start(int32_t length) const;   // transform.h:73-80
    int32_t absolute_end(int32_t length) const;     // transform.h:65-72
    int32_t length() const;                         // transform.h:50-56, -1 when not constant
};

struct BarcodeConfig {
    std::string key;                      // codec member name
    int32_t index;
    double concentration;
    std::vector<std::string> segment;     // IUPAC text per segment
    std::string id;
    std::vector<std::pair<std::string, std::string>> read_group;   // the @RG fields of the record in HeadRGAtom::encode order (atom.cpp:906-965)
    std::vector<std::string> output;      // one URL per output segment when this barcode is a channel of the multiplexer
    bool filter_outgoing_qc_fail;
};

struct DecoderConfig {
    Kind kind;
    Algorithm algorithm;
    int32_t index;                        // position inside its topic
    bool multiplexing_classifier;
    bool enable_quality_control;          // multiplex.cpp:303,393: of the multiplexing decoder (inherited from the root by the compiler)
    int32_t corrected_quality;
    int32_t segment_cardinality;
    int32_t nucleotide_cardinality;
    std::vector<int32_t> barcode_length;
    std::vector<int32_t> segment_offset;
    int32_t high_quality_threshold;
    int32_t high_quality_distance_threshold;
    int32_t quality_masking_threshold;
    std::vector<int32_t> distance_tolerance;
    double noise;
    double confidence_threshold;
    double random_barcode_probability;
    std::vector<TransformConfig> transform;
    BarcodeConfig undetermined;
    std::vector<BarcodeConfig> codec;     // tag_array order == document member order
    std::vector<uint8_t> barcode_code;    // codec.size() x nucleotide_cardinality BAM codes
    bool all_strict;                      // every barcode base is one of A C G T

    // derived layout
    int32_t plane_stride;                 // bytes per read in this decoder's observation plane: nucleotide cardinality rounded up to 4
    int32_t plane_prefix;                 // sum of plane_stride over the decoders in front of this one
    int32_t result_slot;                  // -1 when the decoder emits no result record (naive, passthrough)
    int64_t stats_offset;                 // first u64 word of this decoder in the counter buffer
    bool closed() const { return algorithm == Algorithm::PAMLD || algorithm == Algorithm::MDD; }
};

struct JobConfig {
    std::vector<DecoderConfig> decoder;   // sample, molecular[], cellular[]: the order of transcode.h:51-60
    int32_t input_segment_cardinality;
    int32_t leading_segment_index;
    bool filter_incoming_qc_fail;
    std::vector<int32_t> min_input_length;
    int32_t float_precision;
    int32_t input_phred_offset;            // fastq.h:63-77 subtracts it, sequence.h:450-459 adds the output one back
    int32_t output_phred_offset;
    int32_t output_segment_cardinality;
    std::vector<TransformConfig> template_transform;   // TemplateRule (transform.h:172-226); input segment -1/-2/-3 = corrected sample / cellular / molecular barcode
    int32_t plane_width;                  // bytes per read over all observation planes
    int32_t nucleotide_width;             // sum of nucleotide cardinality over all decoders
    int32_t result_count;                 // number of PAMLD / MDD decoders
    int64_t stats_words;                  // u64 words in the counter buffer
    std::vector<std::string> input;        // one URL per input segment; a URL stated k times is one interleaved feed (transcode.cpp:693-701)
    std::vector<std::string> output;       // root level fallback for a job without a multiplexing decoder
    std::string platform;                  // "ILLUMINA" by default (configuration.json:375): decides the FASTQ comment (fastq.h:176-194)
    int32_t multiplexing_decoder;          // position in `decoder`, -1 when none is marked
    bool enable_quality_control;           // the multiplexer's (multiplex.cpp:393), the root's without one

    static JobConfig from_json(const Json& document);
};

}  // namespace phq
