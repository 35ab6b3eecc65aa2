// K1f: PamlDecoder<T>::classify for TINY codecs (at most 8 ACGT barcodes of up to 16 nt), fast mode tier 1.
//
//   reference pamld.cpp:37-123   posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164  P(r|b) = pow(base, sum of Phred costs), distance, high quality distance
//
// The table driven product form kernel (pamld_lut.hpp) pays ~450 vector instructions per 64 reads before and after
// its scan — the per read tables, the winning block again, the fp64 winner — whatever the codec holds.  A SPLiT-seq
// round of 4 barcodes (C5) is all overhead there.  Here every barcode's probability is the fp64 product of the per
// position factors (P(match) and P(mismatch) of the position's quality: the same host tables, built with the same
// libm calls), no tables per read, the barcodes in SGPRs.  The product is within ~1e-15 of the reference's pow of a
// Kahan sum, so a read is finished here when nothing hangs on less than that: the runner up at most (1 - 1e-10) of the
// winner (no near tie for the argmax), the noise filter and the confidence filter both further than 1e-9 from their
// thresholds.  Everything else goes to this workgroup's worklist segment for the literal kernel.
#include "kernel_common.hpp"

namespace phq {
namespace {

constexpr int TINY_MAX_BARCODES = 8;

// L: the codec's nucleotide cardinality, a template parameter since round 3 — the loops over the positions used to run to
// the row's capacity (8 for a 6 nt codec) under run time guards: a quarter of C5's fourth round for nothing
template <int L>
__global__ __launch_bounds__(BLOCK) void pamld_tiny_kernel(StageArgs args) {
    constexpr int ROW_WORDS = (L + 3) / 4;
    constexpr int LMAX = L;
    __shared__ double factor[129][2];                 // [row][0 match | 1 mismatch]; row 128 = a base outside ACGT
    __shared__ uint32_t deferred_total;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw);
    DecoderRef d = *constant_cast(args.decoder);
    for(int i = threadIdx.x; i < 129; i += BLOCK) {
        factor[i][0] = i < 128 ? d.match_probability[i] : d.uniform_probability;
        factor[i][1] = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
    }
    if(threadIdx.x == 0) { deferred_total = 0; }
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const PHQ_CONSTANT uint32_t* packed = constant_cast(d.barcode_packed);      // 4-bit BAM codes, 8 per word
    const PHQ_CONSTANT double* concentration = constant_cast(d.concentration);
    const int n_barcode = d.n_barcode;
    const int n_word = d.n_word;
    const int64_t n = args.n_read;
    const int64_t tiles = (n + BLOCK - 1) / BLOCK;

    for(int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t i = tile * BLOCK + threadIdx.x;
        const bool active = i < n;
        ReadState r{false, 0};
        bool defer = false;
        if(active) {
            r = load_read_state(args, i);
            const uint32_t* code_row = reinterpret_cast<const uint32_t*>(args.base + i * (4 * ROW_WORDS));
            const uint32_t* quality_row = reinterpret_cast<const uint32_t*>(args.quality + i * (4 * ROW_WORDS));
            uint32_t code_words[ROW_WORDS];
            uint32_t quality_words[ROW_WORDS];
#pragma unroll
            for(int w = 0; w < ROW_WORDS; ++w) { code_words[w] = code_row[w]; quality_words[w] = quality_row[w]; }

            // per position: the observed code, its two factors, the high quality flag
            uint32_t observed[LMAX];
            double match[LMAX];
            double mismatch[LMAX];
            uint32_t high_quality = 0;
#pragma unroll
            for(int j = 0; j < LMAX; ++j) {
                if(j < L) {
                    const uint32_t c = (code_words[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    const uint32_t q8 = (quality_words[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    const uint32_t q = q8 & 0x7fu;                                   // the table has 128 rows
                    // outside ACGT: the uniform quality against every barcode base, nothing at quality 0 (phred.cpp:62-64)
                    const uint32_t row = strict_code(c) ? q : (q ? 128u : 0u);
                    observed[j] = c;
                    match[j] = factor[row][0];
                    mismatch[j] = factor[row][1];
                    if(static_cast<int>(q8) >= d.high_quality_threshold) { high_quality |= 1u << j; }
                }
            }

            double sum = 0;
            double top = 0;
            double second = 0;
            double top_conditional = 0;
            int best = -1;
            uint32_t best_flags = 0;
            for(int b = 0; b < n_barcode; ++b) {                       // wave uniform: the barcode's codes are scalars
                double conditional = 1.0;
                uint32_t flags = 0;
#pragma unroll
                for(int j = 0; j < LMAX; ++j) {
                    if(j < L) {
                        const uint32_t e = (packed[b * n_word + (j >> 3)] >> (4 * (j & 7))) & 0xfu;
                        const bool differs = observed[j] != e;
                        conditional *= differs ? mismatch[j] : match[j];
                        flags |= differs ? (1u << j) : 0u;
                    }
                }
                const double p = conditional * concentration[b];                         // pamld.cpp:66
                sum += p;
                const bool takes_over = p > top;                                          // strict: the first keeps a tie
                second = fmax(second, takes_over ? top : p);
                best = takes_over ? b : best;
                best_flags = takes_over ? flags : best_flags;
                top_conditional = takes_over ? conditional : top_conditional;
                top = takes_over ? p : top;
            }

            if(best < 0) {
                defer = true;
            } else {
                int distance = __popc(best_flags);
                int hq_distance = __popc(best_flags & high_quality);
                const double total = sum + d.adjusted_noise_probability;                  // pamld.cpp:61-86
                double confidence = top / total;                                          // pamld.cpp:92
                const double noise_margin = top_conditional - d.random_barcode_probability;
                const double confidence_margin = confidence - d.confidence_threshold;
                const bool noise_clear = fabs(noise_margin) > 1e-9 * d.random_barcode_probability;
                defer = !(second <= top * (1.0 - 1e-10)) || !noise_clear || (noise_margin > 0 && !(fabs(confidence_margin) > 1e-9));
                if(!defer) {
                    int bin = best + 1;
                    uint8_t status = STATUS_PASS;
                    if(noise_margin > 0) {                                                // pamld.cpp:96
                        if(confidence_margin > 0) {                                       // pamld.cpp:99
                            if(d.high_quality_distance_threshold > 0 && hq_distance >= d.high_quality_distance_threshold) { r.qcfail = true; }
                            stat_add_confidence(d, sink, bin, r.qcfail ? 0 : 1, confidence);
                        } else {
                            stat_add(d, sink, bin, 0, FIELD_LOW_CONFIDENCE, 1ull);
                            r.qcfail = true;
                            status = STATUS_LOW_CONFIDENCE;
                        }
                    } else {
                        // charged to the argmax barcode, counted under unclassified
                        stat_add(d, sink, bin, 0, FIELD_LOW_CONDITIONAL_CONFIDENCE, 1ull);
                        r.qcfail = true;
                        bin = 0;
                        distance = 0;
                        hq_distance = 0;
                        confidence = 0;
                        status = STATUS_NOISE;
                    }
                    decoder_tail(d, sink, r, bin, distance);
                    if(args.result) {
                        DevResult out;
                        out.index = bin > 0 ? d.barcode_index[bin - 1] : 0;
                        out.distance = static_cast<uint8_t>(distance > 255 ? 255 : distance);
                        out.hq_distance = static_cast<uint8_t>(hq_distance > 255 ? 255 : hq_distance);
                        out.status = status;
                        out.qcfail = r.qcfail ? 1 : 0;
                        out.confidence = confidence;
                        stream_result(&args.result[i], out);
                    }
                    stream_read_state(args, i, r);
                }
            }
            if(defer) { stream_placeholder(args, i, r); }                                 // the literal kernel overwrites it
        }
        // deferred reads: one LDS add per wave reserves a run of slots in this workgroup's worklist segment
        const unsigned long long deferred = __ballot(active && defer);
        if(deferred) {
            const int lane = threadIdx.x & 63;
            uint32_t first = 0;
            if(lane == 0) { first = atomicAdd(&deferred_total, static_cast<uint32_t>(__popcll(deferred))); }
            first = __shfl(first, 0);
            if(active && defer) { segment[first + __popcll(deferred & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(i); }
        }
        count_outgoing(args, histogram, active && !defer, r);
    }
    __syncthreads();
    if(threadIdx.x == 0) { args.worklist_count[blockIdx.x] = deferred_total; }
    histogram_flush(args, d, histogram);
}

}  // namespace

// ACGT barcodes, at most 8 of them, of up to 16 nt
bool pamld_tiny_supports(const DevDecoder& host_decoder) {
    return host_decoder.algorithm == 1 && host_decoder.all_strict && host_decoder.n_barcode >= 1 && host_decoder.n_barcode <= TINY_MAX_BARCODES &&
           host_decoder.nucleotide_cardinality >= 1 && host_decoder.nucleotide_cardinality <= 16 && host_decoder.barcode_packed != nullptr;
}

namespace {
template <int LO, int HI>
int launch_tiny_range(const StageArgs& args, int length, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if constexpr(LO > HI) { (void)args; (void)length; (void)grid; (void)histogram_bytes; (void)stream; return static_cast<int>(hipErrorInvalidValue); }
    else {
        if(length == LO) {
            hipLaunchKernelGGL((pamld_tiny_kernel<LO>), grid, dim3(BLOCK), histogram_bytes, stream, args);
            return static_cast<int>(hipGetLastError());
        }
        return launch_tiny_range<LO + 1, HI>(args, length, grid, histogram_bytes, stream);
    }
}
}  // namespace

int launch_pamld_tiny(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(host_decoder.plane_stride != 4 * ((host_decoder.nucleotide_cardinality + 3) / 4)) { return static_cast<int>(hipErrorInvalidValue); }
    return launch_tiny_range<1, 16>(args, host_decoder.nucleotide_cardinality, grid, histogram_bytes, stream);
}

}  // namespace phq
