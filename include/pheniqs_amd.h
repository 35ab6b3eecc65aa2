/* pheniqs_amd.h — C ABI of the MI355X barcode classification path (libpheniqs_amd.so).
 *
 * The reference (Pheniqs 2.1.0) has no FFI: its classification surface is the C++ class family
 * Classifier<T> / Decoder<T> / PamlDecoder<T> / MdDecoder<T> owned by a TranscodingDecoder and driven
 * one read at a time by TranscodingThread::run.  Every entry point below names the reference
 * interface it stands in for (file:line in the reference tree).  Reads cross the boundary in batches
 * of structure-of-arrays planes instead of one `Read` object at a time; everything else — the compiled
 * JSON decoder configuration, the order decoders run in, the sticky QC-fail flag, the counters that
 * make the report — keeps the reference's meaning.
 *
 * All functions return an int in the reference's ErrorCode space (error.h:32-44) and never throw.
 * Pointers are plain; no C++ or torch type appears in a signature.
 */
#ifndef PHENIQS_AMD_H
#define PHENIQS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference error.h:32-44 */
#define PHQ_OK                      0
#define PHQ_UNKNOWN_ERROR           1
#define PHQ_INTERNAL_ERROR          2
#define PHQ_CONFIGURATION_ERROR     3
#define PHQ_OUT_OF_MEMORY_ERROR     4
#define PHQ_COMMAND_LINE_ERROR      5
#define PHQ_IO_ERROR                6
#define PHQ_SEQUENCE_ERROR          7
#define PHQ_OVERFLOW_ERROR          8

#define PHQ_ALGORITHM_PASSTHROUGH   0
#define PHQ_ALGORITHM_PAMLD         1
#define PHQ_ALGORITHM_MDD           2
#define PHQ_ALGORITHM_NAIVE         3
#define PHQ_KIND_SAMPLE             0
#define PHQ_KIND_MOLECULAR          1
#define PHQ_KIND_CELLULAR           2

#define PHQ_STATUS_PASS             0   /* classified, confidence above threshold (may still fail the high quality filter) */
#define PHQ_STATUS_LOW_CONFIDENCE   1   /* pamld.cpp:110-113: stays assigned, QC fail */
#define PHQ_STATUS_NOISE            2   /* pamld.cpp:114-121: failed the noise filter, unclassified, QC fail */
#define PHQ_STATUS_UNCLASSIFIED     3   /* mdd.cpp:82-84: no barcode within tolerance, QC fail */

#define PHQ_MODE_LITERAL            0   /* every barcode scored with the reference's own Kahan recurrence: the posterior within
                                           1e-12 of the reference's */
#define PHQ_MODE_FAST               1   /* PAMLD: tiers of kernels, each finishing the reads it can certify (every integer the
                                           reference's, the posterior within 1e-8 of it; the contract is 1e-6) and handing the
                                           others on, down to the literal kernel; MDD: table lookups, the rest scanned */

#define PHQ_MODE_AUTO               2   /* the default: PHQ_MODE_FAST — but a phq_transcode_* run that writes SAM records of a job of at
                                           most 4 096 (barcode, position) cells a read over its PAMLD decoders classifies those decoders in
                                           PHQ_MODE_LITERAL, so that XB / XM / XC, which print float(1 - posterior) to six digits
                                           (read.h:187-199, auxiliary.cpp:327-354), carry the reference's own digits: 1e-8 of the
                                           posterior is the fourth digit of a 1e-5 tag.  phq_set_mode(PHQ_MODE_FAST) keeps such a run fast */

#define PHQ_DEVICE_NONE            (-1) /* host only set: configuration, staging, statistics, report */

/* What one PAMLD / MDD decoder leaves on a read: the members Decoder<T>/PamlDecoder<T> hold after
 * classify() and the wrappers copy into the output Read (pamld.cpp:133-180, mdd.cpp:96-138). */
typedef struct phq_result {
    int32_t index;          /* decoded->index, 0 = undetermined            barcode.h:37-45      */
    uint8_t distance;       /* edit_distance                                decoder.h:35         */
    uint8_t hq_distance;    /* high_quality_edit_distance                   decoder.h:36         */
    uint8_t status;         /* PHQ_STATUS_*                                                     */
    uint8_t qcfail;         /* output.qcfail() right after this decoder     read.h:96-98         */
    double  confidence;     /* decoding_confidence (0 for MDD)              pamld.h:34           */
} phq_result_t;

/* Shape of the buffers one decoder set exchanges. */
typedef struct phq_layout {
    int32_t n_decoder;          /* sample + molecular[] + cellular[], in classify order (transcode.h:51-60) */
    int32_t n_result;           /* PAMLD / MDD decoders: each owns one result plane                         */
    int32_t plane_width;        /* bytes per read over all observation planes                               */
    int32_t nucleotide_width;   /* sum of nucleotide cardinality: the algorithmic base (= quality) bytes    */
    int64_t stats_words;        /* uint64 words in the counter buffer                                       */
    int32_t input_segment_cardinality;
    int32_t reserved;
} phq_layout_t;

typedef struct phq_decoder_info {
    int32_t kind;               /* PHQ_KIND_*                                                    */
    int32_t algorithm;          /* PHQ_ALGORITHM_*                                               */
    int32_t index;              /* position inside its topic ("index" member)                    */
    int32_t n_barcode;          /* codec size, undetermined excluded                             */
    int32_t nucleotide_cardinality;
    int32_t segment_cardinality;
    int32_t plane_stride;       /* bytes per read in this decoder's plane (cardinality rounded up to 4) */
    int32_t plane_prefix;       /* plane starts at byte offset batch_capacity * plane_prefix     */
    int32_t result_slot;        /* -1: no result plane (naive, passthrough)                      */
    int32_t multiplexing;       /* drives channel_index (classifier.h:83-85)                     */
    int64_t stats_offset;       /* first counter word of this decoder                            */
} phq_decoder_info_t;

/* AccumulatingOption raw counters, selector.h:34-41; record 0 = unclassified, then tag_array order */
typedef struct phq_option {
    uint64_t count;
    uint64_t pf_count;
    uint64_t accumulated_distance;
    uint64_t accumulated_pf_distance;
    uint64_t low_conditional_confidence_count;
    uint64_t low_confidence_count;
    double   accumulated_confidence;
    double   accumulated_pf_confidence;
} phq_option_t;

typedef struct phq_decoder_set phq_decoder_set;

/* ---- construction ---------------------------------------------------------------------------------
 * Replaces TranscodingDecoder::TranscodingDecoder(const Value& ontology) and the per algorithm
 * constructors it switches to (transcode.cpp:31-161; PamlDecoder pamld.cpp:24-36, MdDecoder
 * mdd.cpp:24-36, Decoder decoder.h:44-60, Classifier classifier.h:53-66, Barcode barcode.cpp:24-46).
 * `compiled_job_json` is the compiled ontology (`pheniqs mux --compile`, e.g. reference
 * test/BDGGG/valid/compile_annotated.out); members outside sample / molecular / cellular and the
 * few job level keys listed in INTEGRATION.md are ignored.  `device` is a HIP device ordinal or
 * PHQ_DEVICE_NONE.  On failure *out is NULL and phq_last_error(NULL) holds the message. */
int phq_create(const char* compiled_job_json, int32_t device, phq_decoder_set** out);
void phq_destroy(phq_decoder_set* set);                                   /* ~TranscodingDecoder transcode.cpp:48-61 */
const char* phq_last_error(const phq_decoder_set* set);                    /* Error::what error.h:71-73               */

int phq_get_layout(const phq_decoder_set* set, phq_layout_t* layout);
int phq_get_decoder_info(const phq_decoder_set* set, int32_t decoder, phq_decoder_info_t* info);
int phq_set_mode(phq_decoder_set* set, int32_t mode);                      /* PHQ_MODE_*, default PHQ_MODE_AUTO */

/* ---- extraction -----------------------------------------------------------------------------------
 * Rule::apply(const Read&, Observation&) for every decoder (transform.h:142-169 with
 * Token::absolute_start/end transform.h:65-80 and BamToReverseComplementBam iupac.h:107-124), fused
 * with the pre-filters of TranscodingThread::run (transcode.h:193-205) and the incoming counters of
 * Transcode::pull (transcode.cpp:304-309).
 *
 * Input: n_read reads of n_segment input segments; segment s of read i is the length[s][i] bytes at
 * code[s] + offset[s][i] (BAM 4-bit codes, one per byte) and quality[s] + offset[s][i] (Phred, offset
 * removed).  in_flag[i] bit 0 is the leading segment's QC fail (fastq.h:264-272).
 * Output: planes for up to batch_capacity reads.  Decoder d's plane starts at byte
 * batch_capacity * plane_prefix[d] of `base` (and of `quality`); read k of the plane is plane_stride[d]
 * bytes, positions past the observed length are (0, 0).  kept[k] receives the input
 * ordinal of staged read k, out_flag[k] its incoming QC fail.  *n_kept is the number of staged reads;
 * incoming_count / incoming_pf_count are incremented for every input read before filtering. */
int phq_stage_reads(const phq_decoder_set* set, int64_t n_read, int32_t n_segment,
                    const uint8_t* const* code, const uint8_t* const* quality,
                    const int64_t* const* offset, const int32_t* const* length, const uint8_t* in_flag,
                    int64_t batch_capacity, uint8_t* base_planes, uint8_t* quality_planes,
                    uint8_t* out_flag, int64_t* kept, int64_t* n_kept,
                    uint64_t* incoming_count, uint64_t* incoming_pf_count);

/* ---- classification -------------------------------------------------------------------------------
 * TranscodingDecoder::classify(const Read&, Read&) (transcode.h:51-65) over reads
 * [first_read, first_read + n_read) of a batch whose planes were laid out for batch_capacity reads.
 * Every pointer is DEVICE memory of the set's device.  result holds n_result planes of
 * batch_capacity records (plane r at result + r * batch_capacity); word[i] = channel_index
 * (classifier.h:83-85) | final QC fail << 31 and is also the scratch that carries the sticky QC-fail
 * flag from decoder to decoder, so it must not be NULL.  flag may be NULL (no incoming QC fail).
 * The call only enqueues work on `hip_stream` (a hipStream_t, NULL = default stream); counters
 * accumulate in the set's device counter buffer.  n_read is at most 2^31 - 2^20 per call (read ordinals travel in
 * 32 bits between the kernels of a stage): PHQ_OVERFLOW_ERROR beyond that, classify a larger batch in pieces
 * with first_read.  The scratch the kernels of one call hand reads through (worklists, counters) belongs to the
 * set: calls on ONE stream are ordered by the stream; a call on a different stream than the previous call's first
 * waits, on the device, for that call's kernels (hipStreamWaitEvent) — correct from any stream, concurrent on none.
 * A host that wants two batches in flight uses two sets (or phq_classify_host, which owns two slots). */
int phq_classify_device(phq_decoder_set* set, int64_t batch_capacity, int64_t first_read, int64_t n_read,
                        const uint8_t* base_planes, const uint8_t* quality_planes, const uint8_t* flag,
                        phq_result_t* result, uint32_t* word, void* hip_stream);

/* Same call for HOST buffers laid out for batch_capacity reads (what phq_stage_reads fills): decodes
 * reads [0, n_read), chunk by chunk through the set's pinned double buffers, overlapping hipMemcpyAsync
 * in both directions with compute.  Blocks until results are in host memory.  word may be NULL: it is then not copied
 * back (for a job of one decoder it repeats what the record says). */
int phq_classify_host(phq_decoder_set* set, int64_t batch_capacity, int64_t n_read,
                      const uint8_t* base_planes, const uint8_t* quality_planes, const uint8_t* flag,
                      phq_result_t* result, uint32_t* word);

/* The same call with ONE byte per nucleotide across the link, for callers the link bounds (37 bytes per read of an 8 nt index
 * move 2.2 Greads/s at this pool's 96 GB/s for both directions together).  A nucleotide is A, C, G or T with a quality of at
 * most 62 in all but a few cases (FastqRecord::decode, fastq.h:55-78, leaves a BAM code and a Phred value, a byte each):
 * packed byte = ordinal (A 0, C 1, G 2, T 3) << 6 | quality; (0, 0) — past the observed length, and the columns a plane's
 * stride adds — is the byte 0x7F.  Every other nucleotide — N and the other ambiguity codes, qualities of 63 and more — is
 * the byte 0x3F and an entry of the list beside the planes: read << 32 | column << 16 | code << 8 | quality, column = the byte's place among the plane_width bytes of its read
 * (info[d].plane_prefix + position), the list in read order.
 *   phq_pack_planes       host planes (phq_stage_reads' layout) -> packed planes of the same layout and the list; *n_escape is
 *                         the number of entries the batch needs: larger than escape_capacity, only that number is valid.
 *   phq_classify_host_packed   phq_classify_host on such planes: a chunk's packed planes are expanded on the device in front
 *                         of the stages, results are the same bytes.  word may be NULL (it is then not copied back); a chunk
 *                         of 2 Mi reads holds at most as many list entries as reads (PHQ_OVERFLOW_ERROR beyond). */
int phq_pack_planes(phq_decoder_set* set, int64_t batch_capacity, int64_t n_read, const uint8_t* base_planes, const uint8_t* quality_planes,
                    uint8_t* packed_planes, uint64_t* escapes, int64_t escape_capacity, int64_t* n_escape);
int phq_classify_host_packed(phq_decoder_set* set, int64_t batch_capacity, int64_t n_read, const uint8_t* packed_planes,
                             const uint64_t* escapes, int64_t n_escape, const uint8_t* flag, phq_result_t* result, uint32_t* word);

/* The per type accumulators of the output Read that Read::flush turns into XB / XM / XC (read.h:152-165,187-199):
 * for reads [0, n_read) of HOST result planes laid out for batch_capacity reads (what phq_classify_host fills, or a
 * copy of the device planes), distance[3 * i + kind] and confidence[3 * i + kind], kind = PHQ_KIND_SAMPLE /
 * MOLECULAR / CELLULAR, as the decoders' wrappers leave them after the last decoder:
 *   PAMLD   sample: distance += edit_distance, confidence updated (x 1 -> c, else x *= c)   pamld.cpp:133-140
 *           cellular / molecular: the same when classified, else confidence = 0 and the cellular distance
 *           reset to 0 (set_molecular_distance ADDS its argument: read.h:319-321)        pamld.cpp:149-160,169-180
 *   MDD     sample: distance += edit_distance; cellular / molecular: += when classified, else the CELLULAR
 *           distance is reset, by both wrappers (mdd.cpp:118,136); confidence stays 1     mdd.cpp:96-138
 * Either output may be NULL.  A host side loop over the records; needs no device. */
int phq_summarize(phq_decoder_set* set, int64_t batch_capacity, int64_t n_read, const phq_result_t* result,
                  uint32_t* distance, double* confidence);

/* ---- the steps either side of the path -----------------------------------
 * Transcode::pull + TranscodingThread::run for FASTQ inputs and one SAM text output: n_input synchronised
 * FASTQ files (plain or gzip), one per input segment (fastq.h:55-78,253-285,427-440; read.h:253-264),
 * batches of batch_capacity reads through phq_stage_reads and phq_classify_host, then per read the output
 * segments of the job's `template` (transform.h:190-226) with the tags Read::flush sets (read.h:187-237)
 * in Auxiliary::encode order (auxiliary.cpp:327-354), printed the way htslib prints SAM, after the header
 * lines the reference writes: @HD VN:1.0 SO:unknown GO:query, one @RG per barcode of the sample decoder
 * (atom.cpp:454-472,906-965; transcode.cpp:1679-1700) and a @PG line naming this library (the reference's own
 * test drops @PG before comparing, test/BDGGG/function.sh:29).  Counters accumulate in the set; the incoming
 * counters of Transcode::pull (transcode.cpp:304-309) are added to *incoming_count / *incoming_pf_count. */
int phq_transcode_fastq_to_sam(phq_decoder_set* set, int32_t n_input, const char* const* input_path, const char* sam_output_path,
                               int64_t batch_capacity, uint64_t* incoming_count, uint64_t* incoming_pf_count);

/* The same run driven by the job's own IO directives, what `pheniqs mux --config job.json` does for FASTQ and SAM text
 * (transcode.cpp:500-557,683-733 input; 1261-1417 output; multiplex.h:198-240 routing):
 *   input   `input` URLs under `base input url`, plain or gzip FASTQ; a URL stated k times is ONE interleaved feed
 *           whose consecutive records are k segments of a read (transcode.cpp:693-701).
 *   output  every barcode of the multiplexing decoder (and its undetermined) is a channel; its `output` — own,
 *           else the decoder's, else the root's, else /dev/stdout — names one URL per output segment (a single URL
 *           is repeated, pad_url_array_by_key), under `base output url`.  A read goes to the channel of its decoded
 *           barcode (classifier.h:83-85) unless the channel sets `filter outgoing qc fail` and the read failed.
 *           Format by `?format=` or extension: fastq / fq -> FASTQ records with the Illumina comment
 *           `segment:Y|N:0:raw sample barcode` (fastq.h:144-194), anything else SAM text; `.gz` or
 *           `?compression=gz` -> gzip (one member per batch part, compressed on all host threads: up to the reference's default
 *           level 5 — configuration.json:6 — by the library's own deflate, csrc/fast_deflate.hpp, 4x zlib's rate with 5 % more
 *           bytes; `&level=6` and more, or PHQ_GZIP_ZLIB=1, by zlib).  gzip INPUT: BGZF blocks and files of many small members
 *           inflate member parallel (csrc/gzip_members.hpp); a file of one member — what `gzip` writes — is inflated chunk
 *           parallel from block headers found by search, or on one thread where it is small (csrc/gzip_stream.hpp).
 * `.bam` (or `?format=bam`) -> unaligned BAM, written without htslib: records as the reference's bam1_t holds them (hts.h:169-239,
 * hts.cpp:26-44, auxiliary.cpp:320-354), BGZF blocks by the library's own deflate, the host's record writer.  CRAM (htslib) and
 * bz2 / xz are refused with a configuration error.
 * Records keep the input order inside every file. */
int phq_transcode_job(phq_decoder_set* set, int64_t batch_capacity, uint64_t* incoming_count, uint64_t* incoming_pf_count);

/* The same run with n_copy further copies of the decoder set behind the one pull loop — what the reference does with one
 * TranscodingDecoder per thread (transcode.cpp:1769-1775,2287-2300), here one copy per GPU of a node driven by one process:
 * copy[k] is a set created from the same job on another device (phq_create(job, device k + 1, ...)); batch b is classified
 * by the set (b mod (n_copy + 1) == 0) or by copy (b mod (n_copy + 1)) - 1, records leave in input order, and when the input
 * ends every copy's counters are added into the set's and zeroed (transcode.cpp:1816-1818 -> selector.cpp:68-77), so
 * phq_stats_export / phq_report_json on `set` then speak for the whole run.  With processes per GPU instead, the same merge
 * is phq_stats_allreduce.  A report only run that decodes its FASTQ text on the device stays on the set's own device.
 * NOT a speed-up as it stands: every batch's classify call blocks the one pull loop, so one device is busy at a time; the
 * call shows the dealing of batches, the input order of the records and the counter merge.  A node's GPUs are put to work
 * by one process per GPU over shards of the input (pheniqs_amd/distributed.py, phq_stats_allreduce). */
int phq_transcode_job_copies(phq_decoder_set* set, int32_t n_copy, phq_decoder_set* const* copy, int64_t batch_capacity,
                             uint64_t* incoming_count, uint64_t* incoming_pf_count);

/* What phq_transcode_job would open, without opening it: the counterpart of the `feed` section the reference's compile
 * step leaves in the job (transcode.cpp:683-733 "input feed" / "input feed by segment", 1373-1413 "output feed") and of
 * the channels' output routing (multiplex.h:198-240).  *plan_json (release with phq_free_string):
 *   {"input feed": [{index, url, resolution}], "input feed by segment": [[feed, position in its resolution]],
 *    "output feed": [{index, url, format fastq|sam, compression none|gz, compression level, resolution}],
 *    "channel": [{index (0 = undetermined, then the multiplexing decoder's barcodes in order),
 *                 "output feed by segment": [feed per output segment, empty = discarded], "filter outgoing qc fail"}]}
 * The same configuration errors as phq_transcode_job; needs no device. */
int phq_transcode_plan(phq_decoder_set* set, char** plan_json);

/* Measurement aid (the reference has no timers): when enabled, every kernel launch of
 * phq_classify_* is bracketed by HIP events on its own stream.  phq_profile_collect waits for them, returns
 * the summed milliseconds and the launch count per kernel kind, and forgets them. */
#define PHQ_KERNEL_PAMLD_PRODUCT    0
#define PHQ_KERNEL_PAMLD_LITERAL    1
#define PHQ_KERNEL_MDD              2
#define PHQ_KERNEL_COUNT            3
#define PHQ_KERNEL_MDD_EXACT        4
#define PHQ_KERNEL_PAMLD_EXACT      5   /* factorised exact tier of combinatorial barcode sets */
#define PHQ_KERNEL_PAMLD_NEAR       6   /* neighbour list tier in front of the product form kernel */
#define PHQ_KERNEL_KINDS            7
int phq_profile_enable(phq_decoder_set* set, int32_t enable);
int phq_profile_collect(phq_decoder_set* set, double* milliseconds /* [PHQ_KERNEL_KINDS] */, int64_t* launches /* [PHQ_KERNEL_KINDS] */);
/* The same spans one by one, in launch order, under the name of the kernel (or tier) the stage dispatcher took for the
 * decoder — the names are given where its predicates choose, so this is the chain that really ran, not a description of
 * it: *chain_json = [{"decoder": k, "kernel": "pamld_stream_kernel", "kind": PHQ_KERNEL_*, "ms": t}, ...] (release with
 * phq_free_string).  Forgets the spans like phq_profile_collect; call one or the other after a batch. */
int phq_profile_chain(phq_decoder_set* set, char** chain_json);
/* The chain a batch of n_read reads WOULD take in `mode` (PHQ_MODE_*), without running it: the stage dispatcher itself walks its
 * predicates for every decoder with its launches switched off, so the names are the ones phq_profile_chain reports after a run —
 * *chain_json = [{"decoder": k, "kernel": "...", "kind": PHQ_KERNEL_*}, ...] (release with phq_free_string).  The one table
 * shape -> chain there is: tests/test_shape_grid_plan.py walks a grid of decoder shapes through it.  Works on a set created with
 * PHQ_DEVICE_NONE (the tables are then built in host memory, for their presence alone; grids are sized for an MI355X's 256 CUs).
 * No reference counterpart: the reference has one code path per algorithm (pamld.cpp:37-123, mdd.cpp:37-86). */
int phq_plan_chain(phq_decoder_set* set, int64_t n_read, int32_t mode, char** chain_json);
/* The node's practical HBM ceiling next to the 8 TB/s rooflines are priced against: a plain streaming copy of `bytes` (a
 * multiple of 16), 16 bytes a lane, `repeats` times; *gigabytes_per_second counts bytes read + bytes written. */
int phq_copy_bandwidth(int32_t device, uint64_t bytes, int32_t repeats, double* gigabytes_per_second);
/* Which way the transcoding calls of this process went, summed since it started: counters[0] = rounds whose FASTQ text was
 * decoded on the device, counters[1] = the reads of those rounds, counters[2] = bytes of output records assembled on the
 * device, counters[3] = runs the device path handed to the host reader part way (an input out of the ordinary). */
int phq_device_path_counters(uint64_t counters[4]);

/* Pinned host memory for batches handed to phq_classify_host (hipHostMalloc / hipHostFree). */
int phq_host_alloc(uint64_t bytes, void** pointer);
void phq_host_free(void* pointer);

/* ---- statistics -----------------------------------------------------------------------------------
 * The counter buffer is the merged state of every AccumulatingOption of every decoder plus the
 * outgoing count / pf count of the TranscodingDecoder, as integers (layout: INTEGRATION.md). */
int phq_stats_reset(phq_decoder_set* set, void* hip_stream);
int phq_stats_device_pointer(phq_decoder_set* set, uint64_t** device_words); /* for an in-place all-reduce       */
int phq_stats_export(phq_decoder_set* set, uint64_t* host_words);            /* device -> host, synchronises     */
int phq_stats_import(phq_decoder_set* set, const uint64_t* host_words);      /* host -> device, replaces          */
/* TranscodingDecoder::collect (transcode.cpp:162-179) -> Classifier::collect (classifier.h:87-93)
 * -> AccumulatingOption::collect (selector.cpp:68-77): dst += src, both on the host side of the sets */
int phq_stats_merge(phq_decoder_set* dst, phq_decoder_set* src);
/* The end of run merge over GPUs (reference Transcode::finalize transcode.cpp:1816-1818 over threads):
 * one ncclAllReduce(sum, uint64) in place on the device counter buffer.  rccl_comm is an ncclComm_t. */
int phq_stats_allreduce(phq_decoder_set* set, void* rccl_comm, void* hip_stream);
/* per barcode counters of one decoder, n_barcode + 1 records */
int phq_stats_options(phq_decoder_set* set, const uint64_t* host_words, int32_t decoder, phq_option_t* option);
/* Transcode::finalize + TranscodingDecoder::finalize/encode + print_json (transcode.cpp:180-231,
 * 1812-1863; classifier.h:94-124,161-177; selector.cpp:78-135,197-247; json.cpp:39-46): the report
 * as sorted, pretty printed JSON with doubles truncated to `float precision` decimals.
 * The string belongs to the set and is valid until the next call on it. */
int phq_report_json(phq_decoder_set* set, const uint64_t* host_words, uint64_t incoming_count,
                    uint64_t incoming_pf_count, const char** json);

/* How print_json lays a double out under PrettyWriter::SetMaxDecimalPlaces(precision) (json.cpp:39-46):
 * shortest round trip digits, plain decimal, truncated.  Writes a NUL terminated string; returns
 * PHQ_OVERFLOW_ERROR when `capacity` is too small. */
int phq_format_double(double value, int32_t precision, char* buffer, int32_t capacity);

/* ---- next row f4: the configuration compiler, decoder topics only ----------------------------------------
 * Replaces, for the sample / molecular / cellular decoders of a job document, Transcode::apply_inheritance
 * (reference transcode.cpp:329-437: the "decoder" repository and "base" references), compile_transformation
 * (735-763), compile_decoder_transformation (824-940), compile_decoder (941-1039) and
 * CodecMetric::compile_barcode_tolerance (metric.h:87-111,216-242): defaults, cardinalities, random barcode
 * probability, undetermined, barcode index by sorted key, concentration (1 - noise) / sum, BC / ID, shannon
 * bound and default distance tolerance.  Errors are the reference's ConfigurationError texts (code 3).
 * phq_create runs it by itself on a document that is not compiled yet.  `*compiled_json` is a NUL terminated
 * string (doubles printed with round trip digits) owned by the library: release it with phq_free_string. */
int phq_compile_job(const char* job_json, char** compiled_json);

/* Job::read_instruction_document (job.cpp:163-224): the job document at `path` with its `import` list resolved, as
 * JSON text for phq_create / phq_compile_job.  Imported documents are loaded the same way, depth first, a path only
 * the first time it is met; relative paths are taken against the directory of the importing document (a leading ~ is
 * the home directory); a later import takes precedence over an earlier one and the importing document over all of
 * them (merge_json_value, json.cpp:780-803).  So reference test/BDGGG/BDGGG_annotated.json runs as it is.  A document
 * handed over as text may carry `import` too: its paths are then relative to the working directory. */
int phq_load_job(const char* path, char** job_json);
void phq_free_string(char* text);

/* ---- next row f3: the prior estimator ---------------------------------------------------------------------
 * Replaces Transcode::apply_prior_adjustment (reference transcode.cpp:1888-1941) -> Classifier<T>::adjust_prior
 * (classifier.h:125-160): a copy of `job_json` (the user level or the compiled document of this decoder set; its
 * decoder repository and `base` references are resolved first, as Transcode::assemble does) in which every
 * decoder's "noise" is its estimated noise prior and every codec record's "concentration" the
 * estimated concentration prior of its barcode, from the counters in `host_words` (phq_stats_export / merge /
 * allreduce).  Printed like the report: keys sorted, doubles truncated at `float precision` decimals.
 * Release `*adjusted_json` with phq_free_string. */
int phq_adjust_prior(phq_decoder_set* set, const uint64_t* host_words, const char* job_json, char** adjusted_json);

/* Quality control accumulators of the output reads (reference multiplex.h:28-181): when the job sets `enable quality
 * control`, phq_transcode_fastq_to_sam / phq_transcode_job count every output segment of every read by channel, cycle,
 * nucleotide and quality — one private copy per record thread, merged at the end of the run as Multiplexer::collect does
 * (multiplex.cpp:405-411) — and phq_report_json adds the report's "multiplex" array (multiplex.cpp:419-431).
 * phq_stats_reset clears them with the counters.  To merge several sets or processes (the reference's per thread merge,
 * transcode.cpp:319): phq_qc_export writes a flat image — {channels, segments}, then per channel and segment {cycles,
 * shortest, nucleotide counts[16], mean quality count, sum (2^-24 fixed point), max (a double's bits), distribution[42],
 * cycles x 16 x 42 counts} — and returns its size in words in *n_words (call with words == NULL to ask for the size);
 * phq_qc_merge adds such an image to the set's (counts add, shortest / max take the min / max).  A plain SUM all-reduce
 * does NOT merge an image (two of its fields are a min and a max): gather the images and merge them on one rank. */
int phq_qc_export(phq_decoder_set* set, uint64_t* words, int64_t capacity_words, int64_t* n_words);
int phq_qc_merge(phq_decoder_set* set, const uint64_t* words, int64_t n_words);

/* SHA-256 (hex) over the sources this library was built from (the .cpp, .hip and .hpp files of pheniqs_amd/csrc, its Makefile and
 * this header, sorted by name): ties a measurement or a committed profile summary to a build.  No reference counterpart
 * (the reference prints its version from version.h through `pheniqs --version`, interface.cpp). */
const char* phq_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* PHENIQS_AMD_H */
