// K1 (literal form, any length): PamlDecoder<T>::classify for barcodes of 33 to 128 nt, one read per lane.
//
//   reference pamld.cpp:37-123   posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164  Barcode::compensated_decoding_probability: loops over ANY expected.length, any segment_array
//   reference phred.h:58-60      PhredScale::substitution_quality
//
// The reference's cost is length agnostic; every tier of this library stops at 32 nt (two words of a bit sliced barcode,
// pamld_literal.hpp's fully unrolled L).  Past them this kernel keeps the drop-in: the same Kahan recurrence, operation for
// operation, and the same argmax rules as pamld_literal.hpp, with the length at run time — a read's code and quality row
// (up to 128 bytes each) wait in LDS, word w of lane t at [w][t] so that a wave's reads of one word never share a bank, and
// the barcodes come as packed 4 bit codes through the constant address space (wave uniform: scalar loads).  Every read walks
// every barcode, eight of them side by side through the read's positions.  Rates: profiles/r06_shape_grid.md (96 pairs of 20 + 20 nt).
#include <atomic>

#include "kernel_common.hpp"

namespace phq {
namespace {

__global__ __launch_bounds__(BLOCK) void pamld_literal_long_kernel(StageArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    DecoderRef d = *constant_cast(args.decoder);
    const int L = d.nucleotide_cardinality;
    const int row_words = d.plane_stride / 4;                                     // (L + 3) / 4
    double* const tpq = reinterpret_cast<double*>(lds_raw);                       // 128 true positive qualities (phred.cpp:34-38)
    uint32_t* const code_rows = reinterpret_cast<uint32_t*>(lds_raw + 128 * sizeof(double));       // [row_words][BLOCK]
    uint32_t* const quality_rows = code_rows + row_words * BLOCK;
    unsigned long long* const histogram = reinterpret_cast<unsigned long long*>(quality_rows + row_words * BLOCK);
    for(int i = threadIdx.x; i < 128; i += BLOCK) { tpq[i] = d.true_positive_quality[i]; }
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    const PHQ_CONSTANT uint32_t* packed = constant_cast(d.barcode_packed);
    const PHQ_CONSTANT double* concentration = constant_cast(d.concentration);
    const int n_barcode = d.n_barcode;
    const int n_word = d.n_word;
    const bool all_strict = d.all_strict != 0;
    const double uniform = d.uniform_base_quality;
    const double log2_base_hi = d.log2_base_hi;
    const double log2_base_lo = d.log2_base_lo;
    const int64_t n = args.n_read;
    const int64_t tiles = (n + BLOCK - 1) / BLOCK;
    uint32_t* const my_code = code_rows + threadIdx.x;
    uint32_t* const my_quality = quality_rows + threadIdx.x;

    for(int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t i = tile * BLOCK + threadIdx.x;
        const bool active = i < n;
        ReadState r{false, 0};
        if(active) {
            r = load_read_state(args, i);
            const uint32_t* const code_at = reinterpret_cast<const uint32_t*>(args.base + i * d.plane_stride);
            const uint32_t* const quality_at = reinterpret_cast<const uint32_t*>(args.quality + i * d.plane_stride);
            for(int w = 0; w < row_words; ++w) { my_code[w * BLOCK] = code_at[w]; my_quality[w * BLOCK] = quality_at[w]; }      // (a lane reads its own words only: no barrier)

            // sigma_q of the barcodes b0 .. b0 + GROUP - 1: the reference's Kahan walk over the positions (barcode.h:147-161) for
            // each of them, side by side — a position's observed code, its table lookup (phred.cpp:44-68; the quality byte masked to the
            // table's 128 rows) and its mismatch cost are formed once for the group, every barcode then takes its own
            // four dependent additions: ~9 instructions a (read, barcode, position) where one barcode at a time took ~20
            constexpr int GROUP = 8;
            auto kahan_of_group = [&](int b0, int count, double (&sigma)[GROUP]) {
                double c[GROUP];
#pragma unroll
                for(int g = 0; g < GROUP; ++g) { sigma[g] = 0; c[g] = 0; }
                for(int w = 0; w < row_words; ++w) {
                    const uint32_t code_word = my_code[w * BLOCK];
                    const uint32_t quality_word = my_quality[w * BLOCK];
                    uint32_t expected_word[GROUP];
#pragma unroll
                    for(int g = 0; g < GROUP; ++g) { expected_word[g] = g < count ? packed[(b0 + g) * n_word + (w >> 1)] >> (16 * (w & 1)) : 0u; }
#pragma unroll
                    for(int k = 0; k < 4; ++k) {
                        if(4 * w + k < L) {
                            const uint32_t o = (code_word >> (8 * k)) & 0xffu;
                            const uint32_t q = (quality_word >> (8 * k)) & 0x7fu;
                            const double u = q ? uniform : 0.0;
                            const double match = tpq[q];
                            const double mismatch = strict_code(o) ? static_cast<double>(q) : u;
#pragma unroll
                            for(int g = 0; g < GROUP; ++g) {
                                const uint32_t e = (expected_word[g] >> (4 * k)) & 0xfu;
                                double x = (o == e) ? match : mismatch;
                                if(!all_strict && !strict_code(e)) { x = u; }
                                const double y = x - c[g];
                                const double t = sigma[g] + y;
                                c[g] = (t - sigma[g]) - y;
                                sigma[g] = t;
                            }
                        }
                    }
                }
            };
            auto kahan_of = [&](int b) {
                double sigma[GROUP];
                kahan_of_group(b, 1, sigma);
                return sigma[0];
            };

            // the sum over the barcodes and the argmax exactly as pamld_literal.hpp: quickly rounded powers for the sum, the
            // carefully rounded one where two barcodes come within 1e-15 of each other at the top with sums that differ
            double sigma_p = 0;
            double compensation = 0;
            double best_p = 0;
            double best_conditional = 0;
            int best = -1;
            double running = 0;
            int near_index = 0;
            double near_sigma = 0;
            double near_prior = 0;
            bool ambiguous = false;
            for(int b0 = 0; b0 < n_barcode; b0 += GROUP) {
                const int count = min(GROUP, n_barcode - b0);
                double sigma[GROUP];
                kahan_of_group(b0, count, sigma);
#pragma unroll
                for(int g = 0; g < GROUP; ++g) {
                    if(g < count) {                                    // (in barcode order: the sum and the first of equals are the one at a time walk's)
                        const int b = b0 + g;
                        const double sigma_q = sigma[g];
                        const double prior = concentration[b];
                        const double p = phred_probability_fast(sigma_q, log2_base_hi, log2_base_lo) * prior;
                        const double y = p - compensation;
                        const double t = sigma_p + y;
                        compensation = (t - sigma_p) - y;
                        sigma_p = t;
                        const bool clears = p * (1.0 - 1e-15) > running;
                        const bool close = !clears && p >= running * (1.0 - 1e-15) && p > 0;
                        if(clears) { near_index = b; near_sigma = sigma_q; near_prior = prior; ambiguous = false; }
                        ambiguous = ambiguous || (close && (sigma_q != near_sigma || prior != near_prior));
                        running = p > running ? p : running;
                    }
                }
            }
            double best_summand = 0;
            if(!ambiguous) {
                if(running > 0) {
                    best = near_index;
                    best_conditional = phred_probability(near_sigma, log2_base_hi, log2_base_lo);
                    best_p = best_conditional * concentration[near_index];
                    best_summand = running;
                }
            } else {
                for(int b = 0; b < n_barcode; ++b) {
                    const double sigma_q = kahan_of(b);
                    const double p = phred_probability_fast(sigma_q, log2_base_hi, log2_base_lo) * concentration[b];
                    if(p >= running * (1.0 - 1e-15)) {
                        const double conditional = phred_probability(sigma_q, log2_base_hi, log2_base_lo);
                        const double careful = conditional * concentration[b];
                        if(careful > best_p) { best = b; best_p = careful; best_conditional = conditional; best_summand = p; }     // strict: the lowest index keeps a tie
                    }
                }
            }
            {
                const double y = d.adjusted_noise_probability - compensation;     // pamld.cpp:83-86
                const double t = sigma_p + y;
                compensation = (t - sigma_p) - y;
                sigma_p = t;
            }

            // distance and high quality distance of the winner only (barcode.h:155-160)
            int distance = 0;
            int hq_distance = 0;
            if(best >= 0) {
                for(int w = 0; w < row_words; ++w) {
                    const uint32_t code_word = my_code[w * BLOCK];
                    const uint32_t quality_word = my_quality[w * BLOCK];
                    const uint32_t expected_word = packed[best * n_word + (w >> 1)] >> (16 * (w & 1));
#pragma unroll
                    for(int k = 0; k < 4; ++k) {
                        if(4 * w + k < L) {
                            const bool mismatch = ((code_word >> (8 * k)) & 0xffu) != ((expected_word >> (4 * k)) & 0xfu);
                            distance += mismatch ? 1 : 0;
                            hq_distance += (mismatch && static_cast<int>((quality_word >> (8 * k)) & 0xffu) >= d.high_quality_threshold) ? 1 : 0;
                        }
                    }
                }
            }

            double confidence = best_summand / sigma_p;                            // pamld.cpp:92
            int bin = best + 1;
            uint8_t status = STATUS_PASS;
            if(best_conditional > d.random_barcode_probability) {                 // pamld.cpp:96
                if(confidence > d.confidence_threshold) {                          // pamld.cpp:99
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
                args.result[i] = out;
            }
            store_read_state(args, i, r);
        }
        count_outgoing(args, histogram, active, r);
    }
    __syncthreads();
    histogram_flush(args, d, histogram);
}

}  // namespace

// lds_bytes: the caller's 128 doubles + histogram; the rows come on top
int launch_pamld_literal_long(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    if(args.worklist_role == 2) { return static_cast<int>(hipErrorInvalidValue); }       // no tier in front of it: it sees every read
    const size_t rows = static_cast<size_t>(2) * static_cast<size_t>(host_decoder.plane_stride / 4) * BLOCK * sizeof(uint32_t);
    static std::atomic<uint64_t> prepared{0};
    if(const int status = allow_large_lds(&pamld_literal_long_kernel, prepared)) { return status; }
    hipLaunchKernelGGL(pamld_literal_long_kernel, grid, dim3(BLOCK), lds_bytes + rows, stream, args);
    return static_cast<int>(hipGetLastError());
}

}  // namespace phq
