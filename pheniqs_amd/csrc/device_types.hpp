// Plain-old-data images shared by the host library and the HIP kernels.
#pragma once

#include <cstdint>

namespace phq {

constexpr int DEV_MAX_DECODER = 16;
constexpr int DEV_MAX_SEGMENT = 8;
constexpr int DEV_MAX_WORD = 16;         // 8 nucleotides per 32-bit word of 4-bit BAM codes
constexpr int DEV_MAX_NUCLEOTIDE = 8 * DEV_MAX_WORD;      // 128: the reference loops over any length (barcode.h:147-161); every tier but the
                                                          // general kernels (pamld_literal_long.hip, mdd_kernel) stops at DEV_TIER_NUCLEOTIDE
constexpr int DEV_TIER_NUCLEOTIDE = 32;
constexpr int CROSS_MAX_SEGMENT = 3;     // factorised PAMLD kernel (pamld_cross.hpp): segments of a combinatorial barcode set
constexpr int CROSS_MAX_UNIQUE = 128;    // distinct words per segment
constexpr int EXACT_SINGLE_UNIQUE = 2048;  // words of a single segment set the exact tiers hold (round 6: 512 until then)
constexpr int EXACT_SINGLE_MEDIUM = 512;   // ... and the size up to which pamld_exact_lanes keeps its smaller image of them
// neighbour list kernel (pamld_near.hpp), by barcode length: flag bits per barcode in a plane word, barcodes scored per
// read (c and the nearest to it; a whole number of plane words), words of a barcode's record
constexpr int near_field_bits(int length) { return length <= 8 ? 8 : 10; }
constexpr int near_list(int length) { return length <= 8 ? 16 : 15; }
constexpr int CROSS_NEAR_WEIGHTS = 16;          // floats per combination in cross_near_weight (near_list(L), padded)
constexpr int near_record_words(int length) { return length <= 8 ? 12 : 16; }      // 2 x list words + {radius, cnt_r, cnt_rest, index}, whole quads
// ... with priors that differ: behind those, one float per listed barcode — its prior over c's; slot 0 (c itself) holds
// the noise term over c's prior — again whole quads
constexpr int near_record_words(int length, bool priors) { return near_record_words(length) + (priors ? (length <= 8 ? 16 : 16) : 0); }
constexpr int MAX_DEFERRED_BINS = 9472;        // count_results_kernel: 16 bytes of LDS each
constexpr int PAIR_LISTED = 4;                 // pamld_pair.hpp, the list tier behind the first look: the nearest barcodes of a barcode that are scored exactly
constexpr int PAIR_LIST_WORDS = 16;            // ... {4 low planes, 4 high planes, distance of the nearest barcode NOT listed, the others' priors over its own from there: at that distance / one further / the rest, 4 x the listed barcode's prior over its own (0: no such barcode)}
constexpr int PAIR_RECORD_WORDS = 8;           // pamld_pair.hpp: {low plane, high plane, radius, barcode index, others at the radius / prior, noise / prior, others one further / prior, the rest / prior}
// pamld_stream.hpp.  A barcode's block in LDS: its 16 listed barcodes as 8 low + 8 high plane words, the four weights of
// the bound, its counter bin (8 x 64 bits), padding to 11 quads.  STREAM_WAVES waves in its one workgroup per CU, each
// with 8 KiB of per read tables and a 64 entry ring in LDS.  The LDS map in front of the blocks is fixed by the wave count
// alone, so the host can put a block's LDS ADDRESS into the lookup table (stream_block_at).
// with priors that differ a block also holds the prior of each listed barcode over c's (16 floats; slot 0: the noise term
// over c's prior) between the weights and the bin: 15 quads; and the codec's single precision priors sit behind its words
// (round 4) without the bin — a codec whose counters are rebuilt from the records — 5 quads, 9 with prior ratios
constexpr int stream_record_words(bool priors, bool bins = true) { return bins ? (priors ? 60 : 44) : (priors ? 36 : 20); }
#ifndef PHQ_STREAM_WAVES
#define PHQ_STREAM_WAVES 12
#endif
constexpr int STREAM_WAVES = PHQ_STREAM_WAVES;
#ifndef PHQ_CROSS_STREAM_WAVES
#define PHQ_CROSS_STREAM_WAVES 12
#endif
constexpr int CROSS_STREAM_WAVES = PHQ_CROSS_STREAM_WAVES;      // pamld_cross_stream.hpp (its lookup tables hold ordinals: the host's tables do not depend on it)
// 9 and 10 nt: two tabled groups of four positions and one or two positions picked by masks, 8 listed barcodes one per plane
// word (the 16 plane words of a block either way), three words per input row, the codec's words one barcode per word
constexpr int stream_list(int length) { return length > 8 ? 8 : 16; }
constexpr int STREAM_PLANE_WORDS = 16;
constexpr int stream_row_words(int length) { return length > 8 ? 3 : 2; }
constexpr uint32_t stream_ring_bytes(int length) { return static_cast<uint32_t>(2 * stream_row_words(length) + 2) * 64u * 4u; }      // code words, quality words, carried state, ordinal
constexpr uint32_t stream_words_bytes(int n_barcode, bool priors, int length = 8) {
    return static_cast<uint32_t>((n_barcode + 7) & ~7) * (length > 8 ? 2u : 1u) * 4u + (priors ? static_cast<uint32_t>((n_barcode + 7) & ~7) * 4u : 0u);
}
constexpr uint32_t stream_block_at(int waves, int n_barcode, bool priors, int length = 8) {
    return static_cast<uint32_t>(waves) * 8192u + 1040u + 129u * 16u + static_cast<uint32_t>(waves) * stream_ring_bytes(length) + stream_words_bytes(n_barcode, priors, length);
}
// pamld_cross_stream.hpp (two segment combinatorial sets, L nt each): a word's block is its list — stream_list(L) low plane
// words then as many high plane words, one listed word per plane word — and the four weights of the bound; the LDS map in
// front of the blocks: tables, factor rows, three rings a wave (state and ordinal), the two segments' words
constexpr int cross_stream_block_words(int length) { return 2 * stream_list(length) + 4; }
constexpr uint32_t cross_stream_block_at(int waves, int length, int unique0, int unique1, int segment) {
    return static_cast<uint32_t>(waves) * 8192u + 1040u + 129u * 16u + static_cast<uint32_t>(waves) * 3u * 2u * 64u * 4u +
           static_cast<uint32_t>((unique0 + 7) & ~7) * 8u + static_cast<uint32_t>((unique1 + 7) & ~7) * 8u +
           (segment ? static_cast<uint32_t>(unique0) * static_cast<uint32_t>(cross_stream_block_words(length)) * 4u : 0u);
}
constexpr uint32_t STREAM_UNKNOWN = 0x3fffu;   // the index field of a lookup that names no barcode
constexpr uint8_t CROSS_STREAM_UNKNOWN = 0xffu; // pamld_cross_stream.hpp's byte tables: a key that names no word
static_assert(CROSS_MAX_UNIQUE < CROSS_STREAM_UNKNOWN, "a word's ordinal and the unknown mark share a byte");
constexpr int NEAR_MIN_LENGTH = 5;
constexpr int NEAR_MAX_LENGTH = 10;
constexpr int NEAR_MAX_BARCODES = 2048;
constexpr uint16_t PAIR_ANY_BARCODE = 0xfffeu;  // pamld_pair.hpp: this part of the barcode names no one of its own (its words are shared): the other part's word decides
constexpr uint16_t NEAR_NO_BARCODE = 0xffffu; // ... of a set taken as ONE segment by the exact PAMLD tier (a codec that is not combinatorial)

// Counter buffer geometry (u64 words).  Word 0/1: outgoing reads by [qc-fail | pass-filter]
// (TranscodingDecoder::count/pf_count, transcode.h:61-64).  Then per decoder, per barcode bin
// (bin 0 = unclassified), per [qc-fail | pass-filter] half, STAT_FIELDS words.  Every word is a plain
// sum over reads, so merging threads, streams or GPUs is one integer sum (selector.cpp:68-77).
// Confidence is accumulated exactly as a 2^-52 fixed point number split in two 26-bit limbs.
constexpr int STAT_FIELDS = 6;
constexpr int STAT_WORDS_PER_BIN = 2 * STAT_FIELDS;
constexpr int STAT_JOB_WORDS = 2;

// Grid sizes, in workgroups per CU (the kernels are persistent: grid stride over tiles).  max_blocks = 16 per CU is
// what DecoderSet sizes the worklist for; each stage kernel takes the share that measured best on MI355X — more
// workgroups even out the tail, fewer amortise a kernel's set up (its LDS tables) over more tiles:
//   product form PAMLD (pamld_lut.hpp) 12: C2 28.3 (8) -> 29.0 (12) -> 28.6 (16) Greads/s
//   fused MDD (mdd_fused.hip)          16: C2 55.4 (8) -> 57.6 (12) -> 60.2 (16) Greads/s
//   factorised / wide PAMLD, the rest   8: C3 8.02 (8) -> 7.73 (12) -> 7.49 (16) Greads/s
constexpr int GRID_BLOCKS_PER_CU = 16;
constexpr int GRID_SHARE_LUT = 12;
constexpr int GRID_SHARE_CONSUMER = 12;    // the product form kernel behind a neighbour list tier takes the segments in groups when this is below the producer's share (measured: no gain on C2, 0.47 ms either way)
constexpr int GRID_SHARE_NEAR = 12;        // the neighbour list tier and the tiers behind it
constexpr int GRID_SHARE_MDD_FUSED = 16;
constexpr int GRID_SHARE_DEFAULT = 8;
constexpr int EXACT_GROUP_MAX = 16;      // worklist segments one workgroup of the exact PAMLD tier visits, at most (StageArgs::worklist_group)

constexpr int CONFIDENCE_FRACTION_BITS = 52;
constexpr int CONFIDENCE_LIMB_BITS = 26;
enum StatField : int32_t {
    FIELD_COUNT = 0, FIELD_DISTANCE = 1, FIELD_CONFIDENCE_LOW = 2, FIELD_CONFIDENCE_HIGH = 3,
    FIELD_LOW_CONDITIONAL_CONFIDENCE = 4, FIELD_LOW_CONFIDENCE = 5
};
enum ResultStatus : uint8_t { STATUS_PASS = 0, STATUS_LOW_CONFIDENCE = 1, STATUS_NOISE = 2, STATUS_UNCLASSIFIED = 3 };

// 16-byte per (read, closed decoder) record; layout fixed by include/pheniqs_amd.h phq_result_t
struct alignas(16) DevResult {
    int32_t index;
    uint8_t distance;
    uint8_t hq_distance;
    uint8_t status;
    uint8_t qcfail;
    double confidence;
};

// Everything one decoder launch needs besides the read planes.  Lives in device memory, is read only
// through the constant address space (scalar loads), and is immutable after phq_create.
struct DevDecoder {
    int32_t algorithm;                    // phq::Algorithm
    int32_t kind;                         // phq::Kind
    int32_t n_barcode;
    int32_t nucleotide_cardinality;
    int32_t n_segment;
    int32_t n_word;                       // ceil(nucleotide_cardinality / 8)
    int32_t plane_stride;                 // bytes per read in this decoder's observation plane
    int32_t result_slot;
    int32_t multiplexing;
    int32_t all_strict;
    int32_t high_quality_threshold;
    int32_t high_quality_distance_threshold;
    int32_t quality_masking_threshold;
    int32_t use_lds_histogram;            // bins fit in LDS: workgroup histogram, else global atomics
    uint32_t segment_end_bits[DEV_MAX_NUCLEOTIDE / 32];      // bit j (of word j / 32) set when position j is the last of its segment
    int32_t uniform_prior;                // every concentration is the same double
    int32_t identity_index;               // barcode_index[b] == b + 1 for every barcode (the compiler's numbering)
    int32_t distance_tolerance[DEV_MAX_SEGMENT];
    uint32_t segment_mask[DEV_MAX_SEGMENT][DEV_MAX_WORD];   // nibble flags (bit 4*(j%8) of word j/8) per segment
    uint32_t full_mask[DEV_MAX_WORD];     // nibble flags of every position
    double adjusted_noise_probability;    // noise * random barcode probability (pamld.cpp:29)
    double confidence_threshold;
    double random_barcode_probability;
    double uniform_base_quality;          // 10 log10 4                         phred.h:33
    double log2_base_hi;                  // log2 of the double PHRED_PROBABILITY_BASE (phred.h:34), hi + lo
    double log2_base_lo;
    int64_t stats_offset;                 // first u64 word of this decoder in the counter buffer
    const uint32_t* barcode_packed;       // n_barcode x n_word words of 4-bit BAM codes
    const double* concentration;          // n_barcode
    const int32_t* barcode_index;         // n_barcode
    const double* true_positive_quality;  // 128 entries, -10 log10(1 - 10^(-q/10)), [0] = 0   phred.cpp:34-38
    const double* match_probability;      // 128 entries, pow(base, true_positive_quality[q]), [0] = 1
    const double* mismatch_probability;   // 128 entries, pow(base, q), [0] = 1
    double uniform_probability;           // pow(base, 10 log10 4) ~ 1/4
    // product form kernel only (ACGT barcodes of up to 16 nt): bit-sliced 2-bit codes A,C,G,T = 0..3, position
    // 4f + k with its low bit at 8f + 2 + k and its high bit 4 places up, wrapping (pamld_lut.hip); up to 8 nt two
    // barcodes share a word (even barcode in the low half); padded to whole blocks of 8 barcodes
    const uint32_t* barcode_packed2;
    const float* concentration_single;    // padded with zeros to whole blocks of 8 barcodes
    // neighbour list kernel only (pamld_near.hpp; ACGT barcodes of 5 to 10 nt, uniform priors): near_table[key] = the
    // barcode that the word whose 2-bit codes are key (low code bits of positions 0..L-1 in bits 0..L-1, high code
    // bits in bits L..2L-1) equals or is one mismatch away from (lowest ordinal on ties), NEAR_NO_BARCODE for every
    // other word; 4^L entries.  near_record[c] = near_record_words(L) words: the near_list(L) barcodes {c, the nearest
    // to c} as {low plane word, high plane word} pairs (near_field_bits(L) bits per barcode, bit j = position j),
    // then the radius (distance from c of the nearest barcode not listed, at most 5), the number of barcodes not
    // listed at that distance and further away (two floats; with priors that differ the sums of their priors over c's),
    // barcode_index[c], and padding; with priors that differ near_record_words(L, true) words, the listed barcodes' prior
    // ratios behind the rest
    const uint16_t* near_table;
    const uint32_t* near_record;
    int32_t near_ready;                   // both present and the radius is at least 4 for most barcodes
    // streaming kernel (pamld_stream.hpp; 5 to 10 nt): the same two things in pamld_lut.hpp's bit order (9 and 10 nt: the
    // table has 4^L entries of 16 BITS, a barcode's ordinal — hence `const void*`: 32 bit entries up to 8 nt, 16 bit ones past
    // it — and a block lists 8 barcodes one per plane word — stream_list).
    // stream_table[key], 65536 entries: key = bits 2..17 of the bit-sliced observation (positions 0..3: low code bits in
    // key bits 0..3, high code bits in 4..7; positions 4..7 in bits 8..15); an entry = the LDS address of the barcode's
    // block (18 bits) | barcode_index << 18, index field STREAM_UNKNOWN for a word that names none.  stream_record[c] =
    // stream_record_words(priors differ) words, the image of the block: the listed barcodes two per word (the even one in the low half)
    // as 8 low plane words then 8 high plane words, then how many barcodes that are not listed sit 3, 4, 5 and 6 or more
    // mismatches from c (floats; with priors that differ the sums of their priors over c's; infinite if one sits closer than 3),
    // with priors that differ the 16 prior ratios of the listed barcodes (slot 0: the noise term over c's prior), then zeros (the bin, padding)
    const void* stream_table;
    const uint32_t* stream_record;
    int32_t stream_ready;
    int32_t stream_binless;               // the blocks hold no counter bin (the decoder has no workgroup histogram): stream_record_words(priors, false)
    // one lookup MDD (mdd_near.hpp; one segment of 5 to 10 nt, tolerance 0 or 1): near_table as above and per barcode
    // {low plane, high plane, -, barcode index}
    const uint32_t* mdd_near_record;
    int32_t mdd_near_ready;
    int32_t mdd_pair_ready;               // an MDD codec of two segments with pair_table / pair_record (mdd_near.hpp, PAIRED)
    int32_t mdd_pair_parts;               // ... or of ONE segment of 11 to 20 nt taken as two parts: one tolerance over the whole barcode, what no part names is scanned
    // a combinatorial MDD codec of two segments (mdd_near.hpp, COMBINATORIAL): cross_near_table[s] over the segment's
    // distinct words and {low plane, high plane} of segment 0's words, then of segment 1's
    const uint32_t* mdd_cross_word;
    int32_t mdd_cross_ready;
    // paired barcode kernel only (pamld_pair.hpp; two segments of up to 10 nt whose words are all different, uniform
    // priors): pair_table[s] = the table above — reaching two mismatches — for the words of segment s (key over the segment's positions, entry =
    // the barcode), pair_record[c] = {low code bits, high code bits (bit j = position j of the whole barcode), distance
    // to the nearest other barcode (at most 31), barcode_index[c]}
    // neighbour list kernel of two segment combinatorial sets (pamld_cross_near.hpp; equal segment lengths, barcodes in
    // lexicographic order of their words, identity index): near_table / near_record of each segment's distinct words,
    // a record's last word = the word's share of the barcode ordinal (its ordinal times cross_stride)
    const uint16_t* cross_near_table[2];
    const uint32_t* cross_near_record[2];
    int32_t cross_near_ready;
    // the streaming kernel of such sets (pamld_cross_stream.hpp; uniform priors, 8 to 10 nt words): per segment a lookup
    // table over the segment's distinct words — ONE BYTE an entry (4^L of them; 65 536 up to 8 nt): the ordinal in its segment
    // of the word the key names, CROSS_STREAM_UNKNOWN (0xff) for a key that names none, so a segment holds at most 254 words
    // (CROSS_MAX_UNIQUE is 128: checked below) — and the image of the blocks, cross_stream_block_words(L) words each
    const void* cross_stream_table[2];
    const uint32_t* cross_stream_record[2];
    int32_t cross_stream_ready;
    // ... with priors that differ: per combination (c0, c1), 16 floats each.  cross_near_weight[0]: the noise term over
    // the combination's prior, then prior(c0, w) / prior(c0, c1) for the words w listed with c1 in its record;
    // cross_near_weight[1]: the largest prior over the combination's, then prior(w, c1) / prior(c0, c1) for the words
    // listed with c0.  Null with uniform priors.
    const float* cross_near_weight[2];
    // count_results_kernel (counters.hip): [bin][count, distance, confidence low limb, confidence high limb] of the
    // pass filter halves, where its workgroups add their LDS bins with dense atomics before one small kernel moves the
    // sums into the counter buffer (zero between launches); null when the decoder's bins fit a workgroup histogram
    unsigned long long* count_staging;
    // plates (pamld_plate.hip): two segments of 4 to 8 nt with at most 16 distinct words each, every combination or a part of
    // them.  plate_table[s]: 4^length entries of one byte, the ordinal of the segment's word at most one mismatch from the key
    // (key = low code bits | high code bits << length), 0xff for none; plate_block: per word of segment 0 then of segment 1,
    // 48 words — 16 low code planes (bit j = position j of the segment), 16 high code planes, 16 floats 1 / 0: slot 0 the word
    // itself, then the segment's other words in order, zero past them; plate_ratio: per combination u0 * unique[1] + u1, 32
    // floats — [0] the noise term over the combination's prior, [k] prior(u0, slot k of u1's block) / its own, [16] the largest
    // prior over its own (rounded up), [16 + k] prior(slot k of u0's block, u1) / its own; infinities for a combination that
    // is not in the set; plate_info: per combination {counter bin = barcode ordinal + 1 (0: not in the set), barcode index}
    const uint8_t* plate_table[2];
    const uint32_t* plate_block;
    const uint32_t* plate_ratio;
    const uint32_t* plate_info;
    int32_t plate_unique[2];
    int32_t plate_start[2];
    int32_t plate_length[2];
    int32_t plate_ready;
    const uint16_t* pair_table[2];
    const uint32_t* pair_record;
    const uint32_t* pair_list;            // PAIR_LIST_WORDS per barcode (null: no list tier)
    int32_t pair_start[2];
    int32_t pair_length[2];
    int32_t pair_ready;
    // MDD exact match (mdd.cpp:43-46): open addressing hash of the ACGT barcodes; slot = {key low, key high,
    // ordinal + 1 (0 = empty), barcode index}, key = 2 bits per base, position j at bits 2j
    const uint32_t* exact_table;
    uint32_t exact_slots;                 // power of two
    int32_t exact_probes;                 // longest probe sequence + 1
    // product form kernel for 17 to 32 nt (pamld_wide.hpp): two words per barcode, position j of a word with its low
    // bit at 8 (j / 4) + 1 + j % 4 and its high bit 4 places up (wrapping); padded to whole blocks of 8 barcodes
    const uint32_t* barcode_wide;
    const double* factor_pairs;           // 129 x [match | mismatch] probabilities, row 128 = a base outside ACGT
    // factorised PAMLD kernel only: the barcode set is EVERY combination of per segment sets (dual / triple index)
    int32_t cross_segments;               // 0 = not combinatorial (or not supported), else n_segment
    int32_t cross_groups;                 // groups of 4 positions of the longest segment
    int32_t cross_identity;               // cross_ordinal is the identity: barcodes in lexicographic order of the segment ordinals
    int32_t cross_start[CROSS_MAX_SEGMENT];      // first position of the segment in the observation
    int32_t cross_length[CROSS_MAX_SEGMENT];
    int32_t cross_unique[CROSS_MAX_SEGMENT];     // distinct words of the segment
    int32_t cross_stride[CROSS_MAX_SEGMENT];     // mixed radix weight of the segment in cross_ordinal
    const uint32_t* cross_words;          // [CROSS_MAX_SEGMENT][CROSS_MAX_UNIQUE] bit-sliced 2-bit words, positions relative to the segment
    const int32_t* cross_ordinal;         // combination -> barcode ordinal
    const float* cross_prior;             // [word of segment 0][CROSS_MAX_UNIQUE] priors, single precision, zero padded
    const float* cross_prior_transposed;  // [word of segment 1][CROSS_MAX_UNIQUE], the same transposed
    const double* cross_prior_double;     // the same, double precision
    double cross_prior_max;               // the largest prior
};

// MODE_AUTO (the default): MODE_FAST, but for a transcoding run that writes SAM records of a job small enough for it, which takes
// MODE_LITERAL: XB / XM / XC print float(1 - posterior) to six digits (read.h:187-199, auxiliary.cpp:327-354), and 1e-8 of the
// posterior is the fourth digit of a 1e-5 tag — with the literal kernel's posterior (the reference's to ~1e-16) the records are the
// reference's byte for byte
enum LaunchMode : int32_t { MODE_LITERAL = 0, MODE_FAST = 1, MODE_AUTO = 2 };
constexpr int64_t RECORD_LITERAL_CELLS = 4096;      // ... "small enough": at most this many (barcode, position) cells a read over the job's PAMLD decoders (C2 768, C4 2 880, C5 2 328; C3's 184 320 stay fast)

// One decoder stage over reads [0, n_read) of its planes.
struct StageArgs {
    const DevDecoder* decoder;            // device pointer
    int64_t n_read;
    const uint8_t* base;                  // n_read x plane_stride BAM codes (null for passthrough)
    const uint8_t* quality;               // n_read x plane_stride Phred values, offset removed
    const uint8_t* input_flag;            // first stage only: bit 0 = incoming QC fail; may be null
    DevResult* result;                    // n_read records of this decoder; may be null
    uint32_t* word;                       // channel | qcfail << 31, carried from stage to stage
    unsigned long long* stats;            // counter buffer
    int32_t first_stage;                  // initialise word from input_flag
    int32_t last_stage;                   // count outgoing reads (transcode.h:61-64)
    // ... into the job words when job_outgoing is set (the stage is the job's last), and into bin 0 of the passthrough
    // decoders folded into this stage (round 3): a decoder that only counts (the naive UMI decoder behind C4's and C5's
    // classifiers) sees exactly this stage's outgoing reads, so its launch — a pass over the state words of every read — is
    // dropped and its two counts are added here.  echo_offset[e] = stats_offset of the e-th such decoder.
    int32_t job_outgoing;
    int32_t echo_count;
    int64_t echo_offset[4];
    // Two tier PAMLD (PHQ_MODE_FAST): workgroup g of the product form kernel appends the reads it cannot
    // certify to its own segment worklist[g * worklist_stride ...] and leaves their number in
    // worklist_count[g]; workgroup g of the literal kernel (same grid) then visits exactly those.
    // No global atomic: one shared counter saturates at ~88 increments per microsecond on this part.
    uint32_t* worklist;
    uint32_t* worklist_count;
    int32_t worklist_role;                // 0 none, 1 producer, 2 consumer
    int32_t worklist_stride;
    // The exact tier (pamld_cross_exact_kernel) runs fewer, fuller workgroups: workgroup c visits the segments
    // [c * worklist_group, (c + 1) * worklist_group) below worklist_segments as one list, and compacts what it leaves
    // back into each read's own segment, so the literal tier keeps its one segment per workgroup.
    int32_t worklist_segments;            // segments the producer wrote = its grid
    int32_t worklist_group;               // segments per workgroup of the exact tier (1 elsewhere)
    // Large codecs (no workgroup histogram): the kernels leave count / distance / confidence of the classified
    // reads alone and count_results_kernel (counters.hip) builds them from the result records afterwards —
    // scattered global atomics run at the memory side on this part, ~3.3 of them per read cost as much as the
    // whole classification
    int32_t defer_counters;
    // Round 5: the product form kernel (pamld_lut.hpp, NEAR_TOP) hands a noise read on for nothing but a tied argmax with bit 31 of
    // its worklist entry set and the two blocks of eight barcodes that hold every candidate in its placeholder record; the exact
    // tier's first pass (pamld_exact_lanes.hip, TIE_PACE) settles those from the literal sums of those barcodes alone and leaves
    // the other entries, without the bit, to its scan.  Set by the dispatcher only where that tier follows (pamld_exact_lanes_ties).
    int32_t flag_ties;
    // The neighbour list tier (pamld_near.hpp) hands the reads it defers to the product form kernel WITH their inputs:
    // beside entry e < carry_stride of workgroup g's worklist segment, plane p (0..2 the code words, 3..5 the quality
    // words, 6 the carried state: flag byte or status word) holds its word at carry[(g * 7 + p) * carry_stride + e],
    // so that the consumer reads whole lines instead of gathering 8 byte rows from HBM.  Entries past carry_stride
    // (a quarter of the segment, a multiple of 512) are gathered as before.
    uint32_t* carry;
    int32_t carry_stride;
};
constexpr int CARRY_PLANES = 7;

// Optional per launch timing: HIP events recorded on the launch stream right around each kernel.
enum KernelKind : int32_t { KERNEL_PAMLD_PRODUCT = 0, KERNEL_PAMLD_LITERAL = 1, KERNEL_MDD = 2, KERNEL_COUNT = 3, KERNEL_MDD_EXACT = 4, KERNEL_PAMLD_EXACT = 5, KERNEL_PAMLD_NEAR = 6, KERNEL_KINDS = 7 };
struct LaunchTimer {
    int32_t decoder = 0;                  // the decoder whose stage is being launched (set by the caller of launch_stage)
    virtual ~LaunchTimer() {}
    virtual void begin(int32_t kind, void* stream, const char* name) = 0;      // name: a literal, the kernel the dispatcher took
    virtual void end(void* stream) = 0;
};

// implemented in the kernel translation units; all return a hipError_t value
int launch_stage(const StageArgs& args, const DevDecoder& host_decoder, int32_t mode, int32_t max_blocks, void* stream, LaunchTimer* timer);
// on: launch_stage names the kernels of a stage to the timer and launches none (this thread only; DecoderSet::plan_chain)
void stage_plan_only(bool on);
bool stage_uses_worklist(const DevDecoder& host_decoder, int32_t mode);

}  // namespace phq
