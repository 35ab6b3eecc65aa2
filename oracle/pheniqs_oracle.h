/* oracle/pheniqs_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded, literal restatement of the Pheniqs barcode
 * classification hot path.  It exists to CHECK the HIP path (tests/,
 * __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing under
 * pheniqs_amd/ may include, link or call it.
 *
 * Parity pin: PAMLD / naive / chain / counters / finalize are pinned against
 * the reference's own golden files test/BDGGG/valid/annotated.{out,err}
 * (tests/test_oracle_golden.py) and the known-answer values of SURVEY.md
 * Appendix A.  MDD, multi-segment, reverse-complement and quality masking are
 * NOT exercised by any reference test: for those rows parity is UNPINNED
 * beyond the two MDD known answers recorded in SURVEY.md Appendix A.
 * The reference itself is unbuildable here (include.h:129-165 needs htslib and
 * rapidjson headers that are absent), so there is no oracle/_ref build.
 *
 * All file:line citations are into the reference tree (/root/reference).
 */
#ifndef PHENIQS_ORACLE_H
#define PHENIQS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_SEGMENT   8
#define ORC_MAX_TRANSFORM 16

enum { ORC_ALG_PASSTHROUGH = 0, ORC_ALG_PAMLD = 1, ORC_ALG_MDD = 2, ORC_ALG_NAIVE = 3 };
enum { ORC_KIND_SAMPLE = 0, ORC_KIND_MOLECULAR = 1, ORC_KIND_CELLULAR = 2 };
enum { ORC_STATUS_PASS = 0, ORC_STATUS_LOW_CONFIDENCE = 1, ORC_STATUS_NOISE = 2, ORC_STATUS_UNCLASSIFIED = 3 };

/* one token applied to one observation segment: transform.h:95-118 */
typedef struct {
    int32_t input_segment_index;
    int32_t start;
    int32_t end;
    int32_t end_terminated;
    int32_t output_segment_index;
    int32_t reverse_complement;
} orc_transform_t;

/* what Decoder<T>/Classifier<T>/PamlDecoder/MdDecoder read from the compiled
   decoder JSON: classifier.h:55-60, decoder.h:46-50, pamld.cpp:26-28, mdd.cpp:26-27 */
typedef struct {
    int32_t algorithm;
    int32_t kind;
    int32_t n_barcode;                       /* codec size, undetermined excluded */
    int32_t n_segment;                       /* segment cardinality */
    int32_t segment_length[ORC_MAX_SEGMENT]; /* barcode length[] */
    int32_t nucleotide_cardinality;
    int32_t multiplexing_classifier;
    int32_t high_quality_threshold;
    int32_t high_quality_distance_threshold;
    int32_t quality_masking_threshold;
    int32_t distance_tolerance[ORC_MAX_SEGMENT];
    int32_t n_transform;
    orc_transform_t transform[ORC_MAX_TRANSFORM];
    double noise;
    double confidence_threshold;
    double random_barcode_probability;
    const uint8_t* barcode;        /* n_barcode x nucleotide_cardinality BAM codes, segments concatenated */
    const double*  concentration;  /* n_barcode */
    const int32_t* barcode_index;  /* n_barcode: the "index" member of every codec entry */
} orc_decoder_t;

/* 16-byte result record shared with the product boundary (include/pheniqs_amd.h) */
typedef struct {
    int32_t index;
    uint8_t distance;
    uint8_t hq_distance;
    uint8_t status;
    uint8_t qcfail;       /* output QC-fail state right after this decoder (sticky) */
    double  confidence;
} orc_result_t;

/* AccumulatingOption raw counters: selector.h:34-41 */
typedef struct {
    uint64_t count;
    uint64_t pf_count;
    uint64_t accumulated_distance;
    uint64_t accumulated_pf_distance;
    uint64_t low_conditional_confidence_count;
    uint64_t low_confidence_count;
    double   accumulated_confidence;
    double   accumulated_pf_confidence;
} orc_option_t;

/* AccumulatingOption derived values: selector.cpp:78-101, classifier.h:117-120 */
typedef struct {
    double pf_fraction;
    double average_distance;
    double average_confidence;
    double average_pf_distance;
    double average_pf_confidence;
    double pooled_fraction;
    double pf_pooled_fraction;
    double pooled_classified_fraction;
    double pf_pooled_classified_fraction;
    double estimated_concentration_prior;
} orc_option_final_t;

/* AccumulatingSelector after finalize: selector.h:58-79 */
typedef struct {
    uint64_t count;
    uint64_t pf_count;
    uint64_t classified_count;
    uint64_t pf_classified_count;
    uint64_t accumulated_classified_distance;
    uint64_t accumulated_pf_classified_distance;
    uint64_t low_conditional_confidence_count;
    uint64_t low_confidence_count;
    double accumulated_classified_confidence;
    double accumulated_pf_classified_confidence;
    double pf_fraction;
    double classified_fraction;
    double average_classified_distance;
    double average_classified_confidence;
    double pf_classified_fraction;
    double classified_pf_fraction;
    double average_pf_classified_distance;
    double average_pf_classified_confidence;
    double estimated_noise_prior;
} orc_selector_final_t;

typedef struct orc_job orc_job;

void   orc_phred_init(void);
double orc_substitution_quality(uint8_t expected, uint8_t observed, uint8_t quality);
double orc_true_positive_quality(uint8_t quality);
double orc_uniform_base_quality(void);
double orc_phred_probability_base(void);
uint8_t orc_ascii_to_bam(uint8_t c);
uint8_t orc_reverse_complement_bam(uint8_t c);

/* Barcode::compensated_decoding_probability, barcode.h:131-164, on one staged observation */
void orc_compensated_decoding_probability(const uint8_t* expected, const uint8_t* observed_code,
                                          const uint8_t* observed_quality, int32_t length,
                                          int32_t high_quality_threshold,
                                          double* probability, int32_t* distance, int32_t* hq_distance,
                                          double* sigma_q);

orc_job* orc_job_create(const orc_decoder_t* decoder, int32_t n_decoder);
void     orc_job_destroy(orc_job* job);
void     orc_job_reset(orc_job* job);
int32_t  orc_job_row_width(const orc_job* job);          /* sum of nucleotide cardinality over all decoders */
int32_t  orc_job_result_count(const orc_job* job);       /* number of PAMLD/MDD decoders */
int32_t  orc_job_decoder_offset(const orc_job* job, int32_t decoder);

/* Rule::apply (transform.h:142-169) for every decoder of the job on one read.
   segment_code/segment_quality/segment_length describe the input segments of ONE read. */
void orc_stage_read(const orc_job* job, const uint8_t* const* segment_code, const uint8_t* const* segment_quality,
                    const int32_t* segment_length, int32_t n_input_segment, uint8_t* row_code, uint8_t* row_quality);

/* TranscodingDecoder::classify (transcode.h:51-65) over n staged reads, in order. */
void orc_classify(orc_job* job, int64_t n_read, const uint8_t* row_code, const uint8_t* row_quality,
                  const uint8_t* input_flag, orc_result_t* result, uint32_t* word);

/* same, plus the per type accumulators of read.h:152-165 (sample, molecular, cellular) that Read::flush
   (read.h:187-199) turns into XB/XM/XC: type_distance[n][3], type_confidence[n][3] */
void orc_classify_summary(orc_job* job, int64_t n_read, const uint8_t* row_code, const uint8_t* row_quality,
                          const uint8_t* input_flag, orc_result_t* result, uint32_t* word,
                          uint32_t* type_distance, double* type_confidence);

/* raw counters, (n_barcode + 1) records per decoder, record 0 = unclassified */
const orc_option_t* orc_job_option(const orc_job* job, int32_t decoder);
void orc_job_outgoing(const orc_job* job, uint64_t* count, uint64_t* pf_count);
/* Classifier::collect / TranscodingDecoder::collect: classifier.h:87-93, transcode.cpp:162-179 */
void orc_job_collect(orc_job* job, const orc_job* other);
/* finalize chain: pamld.h:40-48 -> decoder.h:77-83 -> classifier.h:94-124 -> selector.cpp:197-214 */
void orc_job_finalize(const orc_job* job, int32_t decoder, orc_selector_final_t* selector, orc_option_final_t* option);

#ifdef __cplusplus
}
#endif
#endif
