/* oracle/pheniqs_oracle.c — TEST INFRASTRUCTURE ONLY (see pheniqs_oracle.h).
 *
 * Literal CPU restatement of the reference hot path: every loop keeps the
 * reference's operation order (Kahan sums exactly as written, libm pow/log10),
 * so that on x86-64 without FMA contraction it computes what the reference
 * computes.  Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 */
#include "pheniqs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- iupac.h:107-124,153-171 : BAM 4-bit tables -------------------------------------------- */

static const uint8_t REVERSE_COMPLEMENT_BAM[16] = {
    0x0, 0x8, 0x4, 0xc, 0x2, 0xa, 0x6, 0xe, 0x1, 0x9, 0x5, 0xd, 0x3, 0xb, 0x7, 0xf
};

uint8_t orc_ascii_to_bam(uint8_t c) {
    /* iupac.h:153-171: '0'-'3' -> 1,2,4,8; '=' -> 0; IUPAC letters either case; everything else 15 */
    switch(c) {
        case '0': return 1;  case '1': return 2;  case '2': return 4;  case '3': return 8;
        case '=': return 0;
        case 'A': case 'a': return 1;   case 'C': case 'c': return 2;
        case 'G': case 'g': return 4;   case 'T': case 't': return 8;
        case 'M': case 'm': return 3;   case 'R': case 'r': return 5;
        case 'S': case 's': return 6;   case 'V': case 'v': return 7;
        case 'W': case 'w': return 9;   case 'Y': case 'y': return 10;
        case 'H': case 'h': return 11;  case 'K': case 'k': return 12;
        case 'D': case 'd': return 13;  case 'B': case 'b': return 14;
        default: return 15;
    }
}
uint8_t orc_reverse_complement_bam(uint8_t c) { return REVERSE_COMPLEMENT_BAM[c & 0xf]; }

/* ---- phred.h:29-35,58-60 ; phred.cpp:24-72 : PhredScale ------------------------------------- */

static double UNIFORM_BASE_QUALITY;
static double PHRED_PROBABILITY_BASE;
static double substitution_lookup[0x8000];     /* static storage: row q=0 stays 0.0, as in the reference */
static double false_positive_probability[0x80];
static double true_positive_probability[0x80];
static double true_positive_quality[0x80];
static int phred_ready = 0;

void orc_phred_init(void) {
    if(phred_ready) { return; }
    UNIFORM_BASE_QUALITY = 10.0 * log10(4);
    PHRED_PROBABILITY_BASE = pow(10.0, -0.1);
    for(int q = 1; q < 0x80; ++q) { false_positive_probability[q] = pow(PHRED_PROBABILITY_BASE, q); }
    for(int q = 1; q < 0x80; ++q) { true_positive_probability[q] = 1.0 - false_positive_probability[q]; }
    for(int q = 1; q < 0x80; ++q) { true_positive_quality[q] = -10.0 * log10(true_positive_probability[q]); }
    for(int q = 1; q < 0x80; ++q) {
        for(int e = 0; e < 0x10; ++e) {
            for(int o = 0; o < 0x10; ++o) {
                int key = e << 4 | o;
                int strict_e = (e == 1 || e == 2 || e == 4 || e == 8);
                int strict_o = (o == 1 || o == 2 || o == 4 || o == 8);
                double value;
                if(strict_e && e == o)            { value = true_positive_quality[q]; }
                else if(strict_e && strict_o)     { value = q; }
                else                              { value = UNIFORM_BASE_QUALITY; }
                substitution_lookup[q << 8 | key] = value;
            }
        }
    }
    phred_ready = 1;
}
double orc_substitution_quality(uint8_t expected, uint8_t observed, uint8_t quality) {
    /* phred.h:58-60; quality is masked to the 128 rows the table has */
    return substitution_lookup[(quality & 0x7f) << 8 | (expected & 0xf) << 4 | (observed & 0xf)];
}
double orc_true_positive_quality(uint8_t quality) { return true_positive_quality[quality & 0x7f]; }
double orc_uniform_base_quality(void) { return UNIFORM_BASE_QUALITY; }
double orc_phred_probability_base(void) { return PHRED_PROBABILITY_BASE; }

/* ---- barcode.h:131-164 ---------------------------------------------------------------------- */

void orc_compensated_decoding_probability(const uint8_t* expected, const uint8_t* observed_code,
                                          const uint8_t* observed_quality, int32_t length,
                                          int32_t high_quality_threshold,
                                          double* probability, int32_t* distance, int32_t* hq_distance,
                                          double* sigma_out) {
    /* The reference walks segment by segment, position by position; segments are stored
       concatenated here so one loop over the nucleotide cardinality is the same order. */
    double y = 0;
    double t = 0;
    double sigma_q = 0;
    double compensation = 0;
    int32_t d = 0;
    int32_t hqd = 0;
    for(int32_t j = 0; j < length; ++j) {
        y = orc_substitution_quality(expected[j], observed_code[j], observed_quality[j]) - compensation;
        t = sigma_q + y;
        compensation = (t - sigma_q) - y;
        sigma_q = t;
        if(observed_code[j] != expected[j]) {
            ++d;
            if(observed_quality[j] >= high_quality_threshold) { ++hqd; }
        }
    }
    *probability = pow(PHRED_PROBABILITY_BASE, sigma_q);
    *distance = d;
    *hq_distance = hqd;
    if(sigma_out) { *sigma_out = sigma_q; }
}

/* ---- job: the decoder set of one TranscodingDecoder (transcode.h:40-77) --------------------- */

typedef struct {
    orc_decoder_t cfg;
    uint8_t* barcode;
    double* concentration;
    int32_t* barcode_index;
    int32_t offset;              /* first column of this decoder in a staged row */
    int32_t result_slot;         /* -1 for naive / passthrough */
    int32_t segment_offset[ORC_MAX_SEGMENT];
    orc_option_t* option;        /* [0] unclassified, [1..N] tag_array */
    int32_t decoded;             /* option slot of Classifier::decoded; persists across reads */
} decoder_state;

struct orc_job {
    int32_t n_decoder;
    int32_t row_width;
    int32_t n_result;
    decoder_state* decoder;
    uint64_t count;
    uint64_t pf_count;
};

orc_job* orc_job_create(const orc_decoder_t* decoder, int32_t n_decoder) {
    orc_phred_init();
    orc_job* job = (orc_job*)calloc(1, sizeof(orc_job));
    job->n_decoder = n_decoder;
    job->decoder = (decoder_state*)calloc((size_t)n_decoder, sizeof(decoder_state));
    int32_t offset = 0;
    int32_t slot = 0;
    for(int32_t i = 0; i < n_decoder; ++i) {
        decoder_state* s = &job->decoder[i];
        s->cfg = decoder[i];
        const int32_t n = s->cfg.n_barcode;
        const int32_t L = s->cfg.nucleotide_cardinality;
        s->barcode = (uint8_t*)malloc((size_t)(n > 0 ? n : 1) * (size_t)(L > 0 ? L : 1));
        s->concentration = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
        s->barcode_index = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
        if(n > 0) {
            memcpy(s->barcode, decoder[i].barcode, (size_t)n * (size_t)L);
            memcpy(s->concentration, decoder[i].concentration, sizeof(double) * (size_t)n);
            memcpy(s->barcode_index, decoder[i].barcode_index, sizeof(int32_t) * (size_t)n);
        }
        s->cfg.barcode = s->barcode;
        s->cfg.concentration = s->concentration;
        s->cfg.barcode_index = s->barcode_index;
        s->offset = offset;
        int32_t so = 0;
        for(int32_t k = 0; k < s->cfg.n_segment; ++k) { s->segment_offset[k] = so; so += s->cfg.segment_length[k]; }
        offset += L;
        s->result_slot = (s->cfg.algorithm == ORC_ALG_PAMLD || s->cfg.algorithm == ORC_ALG_MDD) ? slot++ : -1;
        s->option = (orc_option_t*)calloc((size_t)n + 1, sizeof(orc_option_t));
        s->decoded = 0;
    }
    job->row_width = offset;
    job->n_result = slot;
    return job;
}
void orc_job_destroy(orc_job* job) {
    if(!job) { return; }
    for(int32_t i = 0; i < job->n_decoder; ++i) {
        free(job->decoder[i].barcode);
        free(job->decoder[i].concentration);
        free(job->decoder[i].barcode_index);
        free(job->decoder[i].option);
    }
    free(job->decoder);
    free(job);
}
void orc_job_reset(orc_job* job) {
    for(int32_t i = 0; i < job->n_decoder; ++i) {
        memset(job->decoder[i].option, 0, sizeof(orc_option_t) * ((size_t)job->decoder[i].cfg.n_barcode + 1));
        job->decoder[i].decoded = 0;
    }
    job->count = 0;
    job->pf_count = 0;
}
int32_t orc_job_row_width(const orc_job* job) { return job->row_width; }
int32_t orc_job_result_count(const orc_job* job) { return job->n_result; }
int32_t orc_job_decoder_offset(const orc_job* job, int32_t decoder) { return job->decoder[decoder].offset; }
const orc_option_t* orc_job_option(const orc_job* job, int32_t decoder) { return job->decoder[decoder].option; }
void orc_job_outgoing(const orc_job* job, uint64_t* count, uint64_t* pf_count) { *count = job->count; *pf_count = job->pf_count; }

/* ---- transform.h:65-80,142-169 : Token::absolute_start/end and Rule::apply ------------------ */

static int32_t absolute_end(const orc_transform_t* t, int32_t length) {
    if(t->end_terminated) {
        if(t->end < 0) { int32_t v = length + t->end; return v < 0 ? 0 : v; }
        else { return t->end > length ? length : t->end; }
    } else { return length; }
}
static int32_t absolute_start(const orc_transform_t* t, int32_t length) {
    if(t->start < 0) { int32_t v = length + t->start; return v < 0 ? 0 : v; }
    else { return t->start > length ? 0 : t->start; }
}

void orc_stage_read(const orc_job* job, const uint8_t* const* segment_code, const uint8_t* const* segment_quality,
                    const int32_t* segment_length, int32_t n_input_segment, uint8_t* row_code, uint8_t* row_quality) {
    /* Observation buffers are zero-filled past the observed length (SURVEY S1 convention). */
    memset(row_code, 0, (size_t)job->row_width);
    memset(row_quality, 0, (size_t)job->row_width);
    for(int32_t i = 0; i < job->n_decoder; ++i) {
        const decoder_state* s = &job->decoder[i];
        int32_t filled[ORC_MAX_SEGMENT];
        for(int32_t k = 0; k < ORC_MAX_SEGMENT; ++k) { filled[k] = 0; }
        for(int32_t k = 0; k < s->cfg.n_transform; ++k) {
            const orc_transform_t* t = &s->cfg.transform[k];
            if(t->input_segment_index < 0 || t->input_segment_index >= n_input_segment) { continue; }
            const uint8_t* from_code = segment_code[t->input_segment_index];
            const uint8_t* from_quality = segment_quality[t->input_segment_index];
            const int32_t from_length = segment_length[t->input_segment_index];
            const int32_t start = absolute_start(t, from_length);
            const int32_t end = absolute_end(t, from_length);
            const int32_t size = end - start;
            if(size > 0) {
                const int32_t o = t->output_segment_index;
                uint8_t* to_code = row_code + s->offset + s->segment_offset[o];
                uint8_t* to_quality = row_quality + s->offset + s->segment_offset[o];
                const int32_t room = s->cfg.segment_length[o] - filled[o];
                const int32_t take = size < room ? size : room;  /* compiled tokens are fixed width, never exceeds */
                if(t->reverse_complement) {
                    for(int32_t j = 0; j < take; ++j) {
                        to_code[filled[o] + j] = REVERSE_COMPLEMENT_BAM[from_code[end - j - 1] & 0xf];
                        to_quality[filled[o] + j] = from_quality[end - j - 1];
                    }
                } else {
                    memcpy(to_code + filled[o], from_code + start, (size_t)take);
                    memcpy(to_quality + filled[o], from_quality + start, (size_t)take);
                }
                filled[o] += take;
            }
        }
    }
}

/* observed length of one staged observation segment: positions before the trailing (code 0, quality 0) run */
static int32_t observed_length(const uint8_t* code, const uint8_t* quality, int32_t capacity) {
    int32_t length = capacity;
    while(length > 0 && code[length - 1] == 0 && quality[length - 1] == 0) { --length; }
    return length;
}

/* ---- per read state: read.h:152-165,269-348 ------------------------------------------------- */

typedef struct {
    int qcfail;
    int32_t channel_index;
    uint32_t distance[3];
    double confidence[3];
} read_state;

static void update_confidence(read_state* r, int kind, double confidence) {
    /* read.h:279-285,306-312,333-339 */
    if(r->confidence[kind] == 1) { r->confidence[kind] = confidence; }
    else { r->confidence[kind] *= confidence; }
}

/* decoder.h:68-76 then classifier.h:78-86 */
static void decoder_tail(decoder_state* s, read_state* r, int32_t edit_distance, int with_distance) {
    orc_option_t* decoded = &s->option[s->decoded];
    if(with_distance && s->decoded > 0 && edit_distance) {
        decoded->accumulated_distance += (uint64_t)edit_distance;
        if(!r->qcfail) { decoded->accumulated_pf_distance += (uint64_t)edit_distance; }
    }
    ++decoded->count;
    if(!r->qcfail) { ++decoded->pf_count; }
    if(s->cfg.multiplexing_classifier) {
        r->channel_index = s->decoded > 0 ? s->cfg.barcode_index[s->decoded - 1] : 0;
    }
}

/* pamld.cpp:37-123 */
static void pamld_classify(decoder_state* s, const uint8_t* code, const uint8_t* quality, read_state* r, orc_result_t* out) {
    const orc_decoder_t* c = &s->cfg;
    const double adjusted_noise_probability = c->noise * c->random_barcode_probability;   /* pamld.cpp:29 */
    double p = 0, y = 0, t = 0;
    int32_t d = 0, hqd = 0;
    double sigma_p = 0;
    double compensation = 0;
    double conditional_probability = 0;
    double adjusted_conditional_decoding_probability = 0;
    double conditional_decoding_probability = 0;
    int32_t edit_distance = 0;
    int32_t high_quality_edit_distance = 0;
    uint8_t status = ORC_STATUS_PASS;

    for(int32_t b = 0; b < c->n_barcode; ++b) {
        orc_compensated_decoding_probability(c->barcode + (size_t)b * (size_t)c->nucleotide_cardinality, code, quality,
            c->nucleotide_cardinality, c->high_quality_threshold, &conditional_probability, &d, &hqd, NULL);
        p = conditional_probability * c->concentration[b];
        y = p - compensation;
        t = sigma_p + y;
        compensation = (t - sigma_p) - y;
        sigma_p = t;
        if(p > adjusted_conditional_decoding_probability) {
            s->decoded = b + 1;
            edit_distance = d;
            high_quality_edit_distance = hqd;
            adjusted_conditional_decoding_probability = p;
            conditional_decoding_probability = conditional_probability;
        }
    }
    y = adjusted_noise_probability - compensation;
    t = sigma_p + y;
    compensation = (t - sigma_p) - y;
    sigma_p = t;

    double decoding_confidence = adjusted_conditional_decoding_probability / sigma_p;
    orc_option_t* decoded = &s->option[s->decoded];
    if(conditional_decoding_probability > c->random_barcode_probability) {
        if(decoding_confidence > c->confidence_threshold) {
            decoded->accumulated_confidence += decoding_confidence;
            if(c->high_quality_distance_threshold > 0 && high_quality_edit_distance >= c->high_quality_distance_threshold) {
                r->qcfail = 1;
            }
            if(!r->qcfail) { decoded->accumulated_pf_confidence += decoding_confidence; }
        } else {
            ++decoded->low_confidence_count;
            r->qcfail = 1;
            status = ORC_STATUS_LOW_CONFIDENCE;
        }
    } else {
        ++decoded->low_conditional_confidence_count;
        r->qcfail = 1;
        s->decoded = 0;
        edit_distance = 0;
        high_quality_edit_distance = 0;
        decoding_confidence = 0;
        status = ORC_STATUS_NOISE;
    }
    decoder_tail(s, r, edit_distance, 1);

    /* pamld.cpp:133-140,149-160,169-180 */
    const int kind = c->kind;
    if(kind == ORC_KIND_SAMPLE) {
        r->distance[kind] += (uint32_t)edit_distance;
        update_confidence(r, kind, decoding_confidence);
    } else {
        if(s->decoded > 0) {
            update_confidence(r, kind, decoding_confidence);
            r->distance[kind] += (uint32_t)edit_distance;
        } else {
            r->confidence[kind] = 0;
            if(kind == ORC_KIND_CELLULAR) { r->distance[kind] = 0; }   /* read.h:319-321: set_molecular_distance adds */
        }
    }
    out->index = s->decoded > 0 ? c->barcode_index[s->decoded - 1] : 0;
    out->distance = (uint8_t)(edit_distance > 255 ? 255 : edit_distance);
    out->hq_distance = (uint8_t)(high_quality_edit_distance > 255 ? 255 : high_quality_edit_distance);
    out->status = status;
    out->qcfail = (uint8_t)r->qcfail;
    out->confidence = decoding_confidence;
}

/* mdd.cpp:37-86 */
static void mdd_classify(decoder_state* s, const uint8_t* code, const uint8_t* quality, read_state* r, orc_result_t* out) {
    const orc_decoder_t* c = &s->cfg;
    int32_t length[ORC_MAX_SEGMENT];
    int32_t total_observed = 0;
    for(int32_t k = 0; k < c->n_segment; ++k) {
        length[k] = observed_length(code + s->segment_offset[k], quality + s->segment_offset[k], c->segment_length[k]);
        total_observed += length[k];
    }
    s->decoded = 0;
    int32_t edit_distance = 0;

    /* exact match of the whole observation: the keys are full length barcodes (mdd.cpp:29-31, barcode.h:46-56) */
    int32_t exact = -1;
    if(total_observed == c->nucleotide_cardinality) {
        for(int32_t b = 0; b < c->n_barcode && exact < 0; ++b) {
            if(memcmp(c->barcode + (size_t)b * (size_t)c->nucleotide_cardinality, code, (size_t)c->nucleotide_cardinality) == 0) { exact = b; }
        }
    }
    if(exact >= 0) {
        s->decoded = exact + 1;
    } else {
        for(int32_t b = 0; b < c->n_barcode; ++b) {
            const uint8_t* expected = c->barcode + (size_t)b * (size_t)c->nucleotide_cardinality;
            int32_t distance = 0;
            int successful = 1;
            for(int32_t k = 0; k < c->n_segment; ++k) {
                const uint8_t* oc = code + s->segment_offset[k];
                const uint8_t* oq = quality + s->segment_offset[k];
                const uint8_t* ec = expected + s->segment_offset[k];
                int32_t error = 0;
                if(c->quality_masking_threshold > 0) {
                    /* sequence.h:321-332 */
                    for(int32_t i = 0; i < length[k]; ++i) {
                        if(oq[i] < c->quality_masking_threshold) { ++error; }
                        else if(oc[i] != ec[i]) { ++error; }
                    }
                } else {
                    /* sequence.h:90-98 */
                    for(int32_t i = 0; i < length[k]; ++i) { if(oc[i] != ec[i]) { ++error; } }
                }
                if(error > c->distance_tolerance[k]) { successful = 0; break; }
                else { distance += error; }
            }
            if(successful) {
                edit_distance = distance;
                s->decoded = b + 1;
                break;
            }
        }
    }
    uint8_t status = ORC_STATUS_PASS;
    if(s->decoded == 0) { r->qcfail = 1; status = ORC_STATUS_UNCLASSIFIED; }
    decoder_tail(s, r, edit_distance, 1);

    /* mdd.cpp:96-102,111-120,129-138 */
    const int kind = c->kind;
    if(kind == ORC_KIND_SAMPLE) {
        r->distance[kind] += (uint32_t)edit_distance;
    } else if(s->decoded > 0) {
        r->distance[kind] += (uint32_t)edit_distance;
    } else {
        /* both the cellular and the molecular wrapper call set_cellular_distance(0) (mdd.cpp:118,136) */
        r->distance[ORC_KIND_CELLULAR] = 0;
    }
    out->index = s->decoded > 0 ? c->barcode_index[s->decoded - 1] : 0;
    out->distance = (uint8_t)(edit_distance > 255 ? 255 : edit_distance);
    out->hq_distance = 0;
    out->status = status;
    out->qcfail = (uint8_t)r->qcfail;
    out->confidence = 0;
}

typedef struct {
    uint32_t distance[3];
    double confidence[3];
} orc_summary_t;

static void classify_read(orc_job* job, const uint8_t* row_code, const uint8_t* row_quality, uint8_t input_flag,
                          orc_result_t* result, uint32_t* word, orc_summary_t* summary) {
    read_state r;
    r.qcfail = (input_flag & 1) ? 1 : 0;        /* transcode.h:207-211 */
    r.channel_index = 0;
    for(int k = 0; k < 3; ++k) { r.distance[k] = 0; r.confidence[k] = 1; }   /* read.h:176-183 */

    /* transcode.h:51-60: sample, molecular[], cellular[]; the job stores decoders in that order */
    for(int32_t i = 0; i < job->n_decoder; ++i) {
        decoder_state* s = &job->decoder[i];
        const uint8_t* code = row_code + s->offset;
        const uint8_t* quality = row_quality + s->offset;
        switch(s->cfg.algorithm) {
            case ORC_ALG_PAMLD: pamld_classify(s, code, quality, &r, &result[s->result_slot]); break;
            case ORC_ALG_MDD:   mdd_classify(s, code, quality, &r, &result[s->result_slot]); break;
            case ORC_ALG_NAIVE:
                /* naive.h:40-45 -> decoder.h:68-76 with decoded == unclassified */
                s->decoded = 0;
                decoder_tail(s, &r, 0, 1);
                break;
            default:
                /* transcode.cpp:77-80: bare Classifier<Barcode>, decoded stays unclassified */
                s->decoded = 0;
                decoder_tail(s, &r, 0, 0);
                break;
        }
    }
    ++job->count;                                  /* transcode.h:61-64 */
    if(!r.qcfail) { ++job->pf_count; }
    if(word) { *word = ((uint32_t)r.channel_index & 0x7fffffffu) | (r.qcfail ? 0x80000000u : 0u); }
    if(summary) {
        for(int k = 0; k < 3; ++k) { summary->distance[k] = r.distance[k]; summary->confidence[k] = r.confidence[k]; }
    }
}

void orc_classify(orc_job* job, int64_t n_read, const uint8_t* row_code, const uint8_t* row_quality,
                  const uint8_t* input_flag, orc_result_t* result, uint32_t* word) {
    orc_result_t scratch[64];
    for(int64_t i = 0; i < n_read; ++i) {
        classify_read(job, row_code + i * job->row_width, row_quality + i * job->row_width,
                      input_flag ? input_flag[i] : 0,
                      result ? result + i * job->n_result : scratch,
                      word ? word + i : NULL, NULL);
    }
}

/* same, also returning the per type accumulators of read.h:152-165 that Read::flush turns into XB/XM/XC */
void orc_classify_summary(orc_job* job, int64_t n_read, const uint8_t* row_code, const uint8_t* row_quality,
                          const uint8_t* input_flag, orc_result_t* result, uint32_t* word,
                          uint32_t* type_distance, double* type_confidence) {
    orc_result_t scratch[64];
    orc_summary_t summary;
    for(int64_t i = 0; i < n_read; ++i) {
        classify_read(job, row_code + i * job->row_width, row_quality + i * job->row_width,
                      input_flag ? input_flag[i] : 0,
                      result ? result + i * job->n_result : scratch,
                      word ? word + i : NULL, &summary);
        for(int k = 0; k < 3; ++k) {
            if(type_distance) { type_distance[i * 3 + k] = summary.distance[k]; }
            if(type_confidence) { type_confidence[i * 3 + k] = summary.confidence[k]; }
        }
    }
}

/* classifier.h:87-93, selector.cpp:68-77, transcode.cpp:162-179 */
void orc_job_collect(orc_job* job, const orc_job* other) {
    job->count += other->count;
    job->pf_count += other->pf_count;
    for(int32_t i = 0; i < job->n_decoder; ++i) {
        for(int32_t b = 0; b <= job->decoder[i].cfg.n_barcode; ++b) {
            orc_option_t* x = &job->decoder[i].option[b];
            const orc_option_t* o = &other->decoder[i].option[b];
            x->count += o->count;
            x->pf_count += o->pf_count;
            x->accumulated_distance += o->accumulated_distance;
            x->accumulated_confidence += o->accumulated_confidence;
            x->low_conditional_confidence_count += o->low_conditional_confidence_count;
            x->low_confidence_count += o->low_confidence_count;
            x->accumulated_pf_distance += o->accumulated_pf_distance;
            x->accumulated_pf_confidence += o->accumulated_pf_confidence;
        }
    }
}

void orc_job_finalize(const orc_job* job, int32_t decoder, orc_selector_final_t* sel, orc_option_final_t* option) {
    const decoder_state* s = &job->decoder[decoder];
    const int32_t n = s->cfg.n_barcode;
    memset(sel, 0, sizeof(*sel));
    memset(option, 0, sizeof(orc_option_final_t) * ((size_t)n + 1));

    /* pamld.h:40-48 (only a PAMLD decoder has these; for the others the per barcode values are all zero) */
    for(int32_t b = 1; b <= n; ++b) {
        sel->accumulated_classified_confidence += s->option[b].accumulated_confidence;
        sel->accumulated_pf_classified_confidence += s->option[b].accumulated_pf_confidence;
        sel->low_conditional_confidence_count += s->option[b].low_conditional_confidence_count;
        sel->low_confidence_count += s->option[b].low_confidence_count;
    }
    /* decoder.h:77-83 */
    for(int32_t b = 1; b <= n; ++b) {
        sel->accumulated_classified_distance += s->option[b].accumulated_distance;
        sel->accumulated_pf_classified_distance += s->option[b].accumulated_pf_distance;
    }
    /* classifier.h:94-124 */
    for(int32_t b = 1; b <= n; ++b) {
        sel->classified_count += s->option[b].count;
        sel->pf_classified_count += s->option[b].pf_count;
    }
    sel->count = sel->classified_count + s->option[0].count;
    sel->pf_count = sel->pf_classified_count + s->option[0].pf_count;

    double estimated_noise_count = (double)sel->low_conditional_confidence_count;
    double confident_noise_ratio = estimated_noise_count / (estimated_noise_count + sel->pf_classified_count);
    if(sel->low_confidence_count > 0) {
        estimated_noise_count += (double)sel->low_confidence_count * confident_noise_ratio;
    }
    sel->estimated_noise_prior = estimated_noise_count / (double)sel->count;

    double estimated_not_noise_prior = 1.0 - sel->estimated_noise_prior;
    for(int32_t b = 0; b <= n; ++b) {
        /* selector.cpp:78-101 */
        const orc_option_t* x = &s->option[b];
        orc_option_final_t* f = &option[b];
        if(x->count > 0) {
            f->average_distance = x->accumulated_distance / (double)x->count;
            f->average_confidence = x->accumulated_confidence / (double)x->count;
            if(sel->count > 0) { f->pooled_fraction = (double)x->count / (double)sel->count; }
            if(sel->classified_count > 0) { f->pooled_classified_fraction = (double)x->count / (double)sel->classified_count; }
        }
        if(x->pf_count > 0) {
            f->pf_fraction = (double)x->pf_count / (double)x->count;
            f->average_pf_distance = x->accumulated_pf_distance / (double)x->pf_count;
            f->average_pf_confidence = x->accumulated_pf_confidence / (double)x->pf_count;
            if(sel->pf_count > 0) { f->pf_pooled_fraction = (double)x->pf_count / (double)sel->pf_count; }
            if(sel->pf_classified_count > 0) { f->pf_pooled_classified_fraction = (double)x->pf_count / (double)sel->pf_classified_count; }
        }
        if(b > 0) { f->estimated_concentration_prior = estimated_not_noise_prior * f->pf_pooled_classified_fraction; }
    }
    /* selector.cpp:197-214 */
    if(sel->count > 0) {
        sel->pf_fraction = (double)sel->pf_count / (double)sel->count;
        sel->classified_fraction = (double)sel->classified_count / (double)sel->count;
    }
    if(sel->pf_count > 0) { sel->pf_classified_fraction = (double)sel->pf_classified_count / (double)sel->pf_count; }
    if(sel->classified_count > 0) {
        sel->average_classified_distance = sel->accumulated_classified_distance / (double)sel->classified_count;
        sel->average_classified_confidence = sel->accumulated_classified_confidence / (double)sel->classified_count;
        sel->classified_pf_fraction = (double)sel->pf_classified_count / (double)sel->classified_count;
    }
    if(sel->pf_classified_count > 0) {
        sel->average_pf_classified_distance = sel->accumulated_pf_classified_distance / (double)sel->pf_classified_count;
        sel->average_pf_classified_confidence = sel->accumulated_pf_classified_confidence / (double)sel->pf_classified_count;
    }
}
