// FASTQ text -> decoder planes ON THE DEVICE: for runs that write nothing (report only: every
// output /dev/null, reference feed.h:105 NullFeed — the workload of docs/illumina_api_vignette.md's prior estimation) and
// for runs whose records are assembled on the device too (record_device.hpp).
//
//   reference fastq.h:55-78      FastqRecord::decode: ASCII -> BAM code, quality - phred offset
//   reference fastq.h:253-285    the Illumina comment's filter flag
//   reference read.h:253-264     Read::validate: every segment of a read carries the same identifier
//   reference transform.h:142-169, 65-80   Rule::apply, Token::absolute_start / absolute_end
//
// The host threads only move raw file bytes into a pinned buffer (inflating them on the way when the file is a gzip stream); kernels find the line ends, check the framing of every
// record, compare the names across the input files, read the filter flag and cut the decoders' tokens straight into the
// observation planes phq_classify_device takes.  Anything out of the ordinary — a record that is not framed by '@' and
// '+', a carriage return, a blank line, names out of sync, sequence and quality of different lengths — raises a flag and
// the caller runs the input through the host reader instead, which tolerates or reports it exactly as the reference does.
#pragma once

#include <cstdint>

namespace phq {

constexpr int FASTQ_MAX_FEED = 8;
constexpr int FASTQ_MAX_TOKEN = 64;
constexpr int FASTQ_MAX_DECODER = 16;

struct FastqToken {                 // one token of a decoder's transform (TransformConfig), resolved against the plan
    int32_t segment;                // input segment
    int32_t start;
    int32_t end;
    int32_t end_terminated;
    int32_t reverse_complement;
    int32_t output_offset;          // segment_offset of the token's output segment inside the decoder's observation
    int32_t output_room;            // barcode_length of that output segment
    int32_t output_segment;
};

struct FastqDecoderPlan {
    int32_t plane_prefix;
    int32_t plane_stride;
    int32_t first_token;
    int32_t n_token;
};

struct FastqPlan {
    int32_t n_feed;
    int32_t n_decoder;
    int32_t n_segment;              // input segments: segment s is record (r * resolution + slot) of its feed (transcode.cpp:693-701)
    int32_t segment_feed[FASTQ_MAX_FEED];
    int32_t segment_slot[FASTQ_MAX_FEED];
    int32_t feed_resolution[FASTQ_MAX_FEED];      // consecutive records of the file that make one read (an interleaved feed)
    int32_t leader_segment;         // whose comment carries the filter flag
    int32_t phred_offset;
    int32_t illumina_comment;
    // the pre-filters of the pull loop (transcode.h:193-205): a read is dropped — pulled, counted as incoming, never decoded —
    // when its filter flag is set and the job filters incoming QC fail, or when a segment past the first is shorter than the
    // job's minimum for it.  With either in force the kept reads are compacted: read r goes to the row of its rank among them.
    int32_t filter_incoming_qc_fail;
    int32_t min_length[FASTQ_MAX_FEED];
    int32_t filtering;              // either of the two
    // how the kernels share the work out (fastq_device_inner.hpp): a wave takes `reads_per_wave` consecutive reads (64, 32, 16
    // or 8) and copies their text into `input_room` bytes of LDS; fastq_plan_waves sets both from the files' bytes per read
    int32_t reads_per_wave;
    int32_t input_room;
    FastqDecoderPlan decoder[FASTQ_MAX_DECODER];
    FastqToken token[FASTQ_MAX_TOKEN];
};

// what the planning kernel leaves for the host after a round's chunks are indexed
struct FastqRound {
    uint32_t lines[FASTQ_MAX_FEED];          // complete lines found in the feed's chunk
    uint32_t consumed[FASTQ_MAX_FEED];       // bytes of the chunk the round's records take
    uint32_t records;                        // READS every feed holds whole (a read is `resolution` records of an interleaved feed), at most the batch capacity
    uint32_t irregular;                      // a feed's newline index overflowed (lines shorter than any FASTQ has)
};

struct FastqFeedChunk {
    const uint8_t* raw;             // device: the chunk's bytes, zero padded to a multiple of 16
    uint32_t bytes;
    uint32_t* newline;              // device: offsets of the chunk's newlines, in order
    uint32_t newline_capacity;
    uint32_t* block_count;          // device: scratch, one word per 4 KiB of the chunk (+ 1)
    uint32_t* total;                // device: number of newlines found
};

// What the decode and record kernels read through a pointer, in device memory for the whole run: passed by value, the 3 KiB
// of it were copied to every lane's scratch memory by the compiler (the kernels index them at run time) and read from there.
struct FastqTables {
    FastqPlan plan;
    FastqFeedChunk chunk[2][FASTQ_MAX_FEED];     // the feeds' chunks in the two slots rounds alternate between (raw, newline)
    uint8_t bam_of_ascii[256];
    uint8_t complement_of_bam[16];
};
void fastq_fill_tables(FastqTables& tables);     // the two code tables (host)

// reads_per_wave and input_room of a plan whose reads take about `bytes_per_read` bytes of the input files together
void fastq_plan_waves(FastqPlan& plan, double bytes_per_read);

// every function returns a hipError_t value; all work goes to `stream`
int fastq_index(const FastqFeedChunk& chunk, void* stream);
int fastq_plan_round(const FastqFeedChunk* chunk, int32_t n_feed, const int32_t* resolution, uint32_t capacity, FastqRound* round_on_device, void* stream);
// reads [0, n) of the round -> planes laid out for `capacity` reads; *irregular |= 1 on anything out of the ordinary;
// *flagged += reads whose filter flag is set.  A plan that filters needs the scratch `keep` (n bytes) and `keep_block`
// (n / 32 + 2 words) and leaves the number of rows it wrote — the reads it kept, in input order — in *kept (device);
// a plan that does not writes row r for read r and leaves *kept alone.  `longest` (device, FASTQ_MAX_FEED words) may be NULL:
// else longest[s] = max(longest[s], the longest segment s of the reads decoded).
// `on_device` is a device copy of the tables whose plan is `plan`; the round's chunks are those of `slot`
int fastq_decode(const FastqTables* on_device, const FastqPlan& plan, int32_t slot, uint32_t n, int64_t capacity, uint8_t* base_planes, uint8_t* quality_planes,
                 uint8_t* flag, uint32_t* irregular, unsigned long long* flagged, uint8_t* keep, uint32_t* keep_block, uint32_t* kept, uint32_t* longest, void* stream);

}  // namespace phq
