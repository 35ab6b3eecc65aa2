// Output records — SAM or FASTQ text — assembled ON THE DEVICE from the FASTQ chunks, the observation planes and the result
// records of a round: one stream, or a set of files per barcode of the multiplexing decoder (the host
// copies the finished text into them, and compresses the gzip ones).
//
//   reference read.h:187-237        Read::flush: the per type accumulators into the auxiliary tags
//   reference auxiliary.cpp:324-354 Auxiliary::encode: the order of the tags (FI TC RG BC QT XB RX QX OX BZ XM CB CR CY XC)
//   reference sequence.h:382-398    append_corrected: the corrected barcode and its qualities
//   reference transform.h:190-226   TemplateRule::apply: the output segments from the template tokens
//   reference fastq.h:144-194       FastqRecord::encode and the Illumina comment
//   reference multiplex.h:198-240   Multiplexer::push -> Channel::push: which channel takes the read, whether it is written
//
// Two passes over the reads of a round, one lane per read: the first leaves every read's byte count, a scan turns the counts
// into places, the second writes the text where it belongs.  The host only copies the finished text to the output.
#pragma once

#include <cstddef>
#include <cstdint>

#include "device_types.hpp"
#include "fastq_device.hpp"

namespace phq {

constexpr int RECORD_MAX_TOKEN = 32;
constexpr int RECORD_MAX_OUT = 8;
constexpr int RECORD_MAX_SINK = 1024;       // output files of a run whose records are assembled on the device
// what a sink's records are: SAM lines, FASTQ records, or SAM lines the host turns into BAM records (bam_record.hpp) — those state
// XB / XM / XC as '#' and the eight hexadecimal digits of the float's bits, since a BAM record holds the float and not its `%g`
constexpr uint8_t RECORD_FORMAT_SAM = 0;
constexpr uint8_t RECORD_FORMAT_FASTQ = 1;
constexpr uint8_t RECORD_FORMAT_SAM_FOR_BAM = 2;

struct RecordDecoder {
    int32_t kind;                   // 0 sample, 1 molecular, 2 cellular
    int32_t plane_prefix;
    int32_t plane_stride;
    int32_t result_slot;            // -1: no result record (naive, passthrough)
    int32_t segment_cardinality;
    int32_t nucleotide_cardinality;
    int32_t closed;                 // PAMLD or MDD: the decoder corrects
    int32_t pamld;
    int32_t corrected_quality;
    int32_t multiplexing;
    int32_t n_ordinal;              // ints[ordinal_at + index], index < n_ordinal: the barcode's ordinal in tag_array order, -1 none
    int32_t ordinal_at;
    int32_t code_at;                // bytes[code_at + ordinal * nucleotide_cardinality + position]: the barcode's BAM codes
    int32_t segment_offset[DEV_MAX_SEGMENT];
    int32_t barcode_length[DEV_MAX_SEGMENT];
};

struct RecordToken {                // one token of the template transform
    int32_t source;                 // >= 0: input segment; -1, -2, -3: the corrected sample, cellular, molecular barcode
    int32_t start;
    int32_t end;
    int32_t end_terminated;
    int32_t reverse_complement;
};

struct RecordPlan {
    int32_t n_decoder;
    int32_t n_out;                  // output segments
    int32_t n_channel;
    int32_t n_sink;                 // output files the records go to
    int32_t sink_at;                // ints[sink_at + channel * n_out + o]: the sink of output segment o of a channel, -1 none
    int32_t format_at;              // bytes[format_at + sink]: RECORD_FORMAT_* of the sink
    int32_t illumina_comment;
    int32_t quality_shift;          // output Phred offset - input Phred offset
    int32_t input_phred_offset;
    int32_t output_phred_offset;
    int32_t n_group;                // read group ids: bytes[ints[group_at + g] .. ints[group_at + g + 1])
    int32_t group_at;
    int32_t channel_at;             // bytes[channel_at + channel]: bit 0 the channel is written, bit 1 it filters outgoing QC fail
    int32_t out_first[RECORD_MAX_OUT + 1];      // out_token[out_first[o] .. out_first[o + 1]): the tokens of output segment o, in order
    int32_t out_token[RECORD_MAX_TOKEN];
    int32_t columns_size[RECORD_MAX_OUT][2];
    char columns[RECORD_MAX_OUT][2][32];        // "\t<flag>\t*\t0\t0\t*\t*\t0\t0\t" per output segment and filter flag
    RecordDecoder decoder[FASTQ_MAX_DECODER];
    RecordToken token[RECORD_MAX_TOKEN];
    char text_of_bam[16];
    char canonical_text[256];       // a FASTQ letter as the output prints it: BamToAmbiguousAscii of its BAM code
    char complement_text[256];
    uint8_t complement_of_bam[16];
    const int32_t* ints;            // device
    const uint8_t* bytes;           // device
};

struct RecordRound {
    const RecordPlan* plan;         // device
    const DevResult* result;        // device: [slot][capacity]
    const uint32_t* word;           // device: the summary word of every row (bit 31: the read failed QC)
    uint32_t* size;                 // device scratch: n words
    unsigned long long* block;      // device scratch: n / 256 + 2 words
    unsigned long long* total;      // device: bytes of the round's text (after record_sizes)
    char* text;                     // device: at least *total bytes (record_write)
    // a plan of several sinks (record_sizes_by_sink / record_write_by_sink): every sink's records are consecutive in `text`,
    // sink s from byte sink_base[s] on; sink_base[n_sink] is the whole text
    uint32_t* piece_size;           // device scratch: n x n_out words, the bytes of the record of output segment o of read r
    unsigned long long* sink_block; // device scratch: (n / 32 + 2) x n_sink words: bytes per workgroup's reads and sink, then their prefix sums
    unsigned long long* sink_base;  // device: n_sink + 1 words
    unsigned long long* quality_control;        // device: the accumulators (record_quality_control), or NULL
    int32_t cycles;                 // ... and the cycles every cell of them holds
};

// The quality control accumulators of the output reads on the device (reference multiplex.h:28-181: what Channel::push
// counts when `enable quality control` is set; csrc/qc.hpp is the host's form).  One cell per (channel, output segment),
// QC_CELL_HEAD words — [0] the longest segment seen, [1] 0x7fffffff - the shortest, [2] reads with a mean quality,
// [3] the sum of those means in 2^-24 units, [4] the largest mean (the bits of a double), [5 .. 46] the distribution of the
// means — then cycles x 16 codes x 42 Phred values of counts.  The per code totals are sums over the cycles: the host forms them.
constexpr int QC_CELL_HEAD = 64;
constexpr int QC_CELL_CODES = 16;
constexpr int QC_CELL_PHRED = 42;
constexpr size_t qc_cell_words(int32_t cycles) { return static_cast<size_t>(QC_CELL_HEAD) + static_cast<size_t>(cycles) * QC_CELL_CODES * QC_CELL_PHRED; }

// every function returns a hipError_t value; all work goes to `stream`.  `keep` / `keep_block` are what fastq_decode left
// (a plan that filters), rows are the compacted ones the planes and results hold; `on_device`, `plan`, `slot` as for fastq_decode.
int record_sizes(const FastqTables* on_device, const FastqPlan& plan, int32_t slot, uint32_t n, int64_t capacity, const uint8_t* base_planes, const uint8_t* quality_planes,
                 const uint8_t* keep, const uint32_t* keep_block, const RecordRound& round, void* stream);
int record_write(const FastqTables* on_device, const FastqPlan& plan, int32_t slot, uint32_t n, int64_t capacity, const uint8_t* base_planes, const uint8_t* quality_planes,
                 const uint8_t* keep, const uint32_t* keep_block, const RecordRound& round, void* stream);
// the same two passes for a plan of several sinks: the records of a read go to the files of its channel, every file's
// records stay in input order
int record_sizes_by_sink(const FastqTables* on_device, const FastqPlan& plan, int32_t slot, uint32_t n, int64_t capacity, const uint8_t* base_planes, const uint8_t* quality_planes,
                         const uint8_t* keep, const uint32_t* keep_block, const RecordRound& round, int32_t n_out, int32_t n_sink, void* stream);
int record_write_by_sink(const FastqTables* on_device, const FastqPlan& plan, int32_t slot, uint32_t n, int64_t capacity, const uint8_t* base_planes, const uint8_t* quality_planes,
                         const uint8_t* keep, const uint32_t* keep_block, const RecordRound& round, int32_t n_out, int32_t n_sink, void* stream);
// every read the pull loop kept is counted in the cell of its channel, written or not (multiplex.h:219-221); an output
// segment longer than round.cycles is not counted and sets *overflow (device)
int record_quality_control(const FastqTables* on_device, const FastqPlan& plan, int32_t slot, uint32_t n, int64_t capacity, const uint8_t* base_planes, const uint8_t* quality_planes,
                           const uint8_t* keep, const uint32_t* keep_block, const RecordRound& round, uint32_t* overflow, void* stream);

}  // namespace phq
