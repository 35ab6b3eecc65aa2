// K1 (literal form): PamlDecoder<T>::classify for one decoder, one read per lane.
//
//   reference pamld.cpp:37-123   posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164  Barcode::compensated_decoding_probability (Kahan sum of Phred costs, pow)
//   reference phred.h:58-60      PhredScale::substitution_quality, table built in phred.cpp:39-72
//
// "Literal" means sigma_q of every barcode is the reference's Kahan recurrence, operation for
// operation, in fp64 with contraction off: the argmax (ties included: strict '>' keeps the lowest
// index) is decided on the same sums the reference decides on.  Only pow() differs (in-kernel
// 2^x, kernel_common.hpp: ~0.502 ulp where the argmax is decided, ~1 ulp in the sum), which moves the posterior by ~1e-16.
//
// Template parameters: L = nucleotide cardinality (fully unrolled), STRICT = every barcode base is
// one of A, C, G, T (the common case; drops the IUPAC branch from the inner loop).
#pragma once

#include "kernel_common.hpp"

namespace phq {

// Round 5: what reaches this kernel as a CONSUMER is a handful of reads (S384: ONE of 100 M, H94: 51), and a lane that walks
// every barcode of the codec for its read alone is one chain of dependent fp64 operations 384 barcodes long: 0.16 ms for that one
// read.  A worklist segment of at most LITERAL_SHARED_READS reads is shared out instead: the workgroup's lanes compute the Kahan
// sums and the quickly rounded powers of the barcodes between them, into LDS, and lane 0 alone goes through them in barcode order
// — the sum over the barcodes, the argmax and its ties exactly as below, on the same values.
constexpr int LITERAL_SHARED_READS = 4;
constexpr int LITERAL_SHARED_MOST = 1024;           // barcodes: 16 KiB of LDS

template <int L, bool STRICT>
__global__ __launch_bounds__(BLOCK) void pamld_literal_kernel(StageArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* tpq = reinterpret_cast<double*>(lds_raw);
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw + 128 * sizeof(double));
    __shared__ double shared_sigma[LITERAL_SHARED_MOST];
    __shared__ double shared_p[LITERAL_SHARED_MOST];
    DecoderRef d = *constant_cast(args.decoder);

    // PhredScale in LDS: 128 true positive qualities; the other two values a lookup can return are the
    // Phred value itself and the constant 10 log10 4 (phred.cpp:44-68).  Row q = 0 is all zeros.
    for(int i = threadIdx.x; i < 128; i += BLOCK) { tpq[i] = d.true_positive_quality[i]; }
    histogram_clear(d, histogram);
    __syncthreads();

    constexpr int W = (L + 7) / 8;
    constexpr int LP = (L + 3) / 4 * 4;
    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    const PHQ_CONSTANT uint32_t* packed = constant_cast(d.barcode_packed);
    const PHQ_CONSTANT double* concentration = constant_cast(d.concentration);
    const int n_barcode = d.n_barcode;
    const double uniform = d.uniform_base_quality;
    const double log2_base_hi = d.log2_base_hi;
    const double log2_base_lo = d.log2_base_lo;
    // as the second tier of PHQ_MODE_FAST workgroup g visits only the reads workgroup g of the product form
    // kernel deferred (same grid): its segment of the worklist, worklist_count[g] entries
    const bool consumer = args.worklist_role == 2;
    const uint32_t* const segment = consumer ? args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride : nullptr;
    const int64_t n = consumer ? static_cast<int64_t>(args.worklist_count[blockIdx.x]) : args.n_read;
    const int64_t tiles = (n + BLOCK - 1) / BLOCK;
    const int64_t tile_first = consumer ? 0 : blockIdx.x;
    const int64_t tile_step = consumer ? 1 : gridDim.x;

    const bool shared = consumer && n > 0 && n <= LITERAL_SHARED_READS && n_barcode <= LITERAL_SHARED_MOST;      // (the workgroup's: LITERAL_SHARED_READS above)

    for(int64_t tile = tile_first; tile < (shared ? n : tiles); tile += tile_step) {
        if(shared) { __syncthreads(); }        // lane 0 is done with the read before
        const int64_t k = shared ? tile : tile * BLOCK + threadIdx.x;
        const bool active = shared || k < n;
        ReadState r{false, 0};
        if(active) {
            const int64_t i = consumer ? static_cast<int64_t>(segment[k]) : k;
            r = load_read_state(args, i);
            uint8_t code[LP];
            uint8_t quality[LP];
            load_plane_row<L>(args.base, i, code);
            load_plane_row<L>(args.quality, i, quality);

            uint32_t observed[L];
            double match_cost[L];
            double mismatch_cost[L];
            uint32_t nonzero_quality = 0;       // only read when a barcode holds a base outside ACGT
#pragma unroll
            for(int j = 0; j < L; ++j) {
                const uint32_t c = code[j];
                const uint32_t q = quality[j] & 0x7fu;      // the table has 128 rows
                observed[j] = c;
                match_cost[j] = tpq[q];
                const double u = q ? uniform : 0.0;
                mismatch_cost[j] = strict_code(c) ? static_cast<double>(q) : u;
                if(q) { nonzero_quality |= 1u << j; }
            }

            double sigma_p = 0;
            double compensation = 0;
            double best_p = 0;
            double best_conditional = 0;
            int best = -1;
            // Every barcode's probability within ~1 ulp (phred_probability_fast) for the sum.  The argmax needs the
            // carefully rounded power only where two barcodes come within 1e-15 of each other at the top with Kahan sums
            // (or priors) that are NOT the same double — equal sums and priors are equal products under any pow: the
            // first keeps the tie —: the same multiset of costs summed in another order, ~1 read in 10 000.  Those reads
            // take a second walk.
            double running = 0;                 // the largest quickly rounded probability so far
            int near_index = 0;                 // the barcode that holds it (the first of equals)
            double near_sigma = 0;
            double near_prior = 0;
            bool ambiguous = false;
            auto kahan_of = [&](int b) {
                uint32_t expected[W];
#pragma unroll
                for(int w = 0; w < W; ++w) { expected[w] = packed[b * W + w]; }
                double sigma_q = 0;
                double c = 0;
#pragma unroll
                for(int j = 0; j < L; ++j) {
                    const uint32_t e = (expected[j >> 3] >> (4 * (j & 7))) & 0xfu;
                    double x = (observed[j] == e) ? match_cost[j] : mismatch_cost[j];
                    if(!STRICT) {
                        if(!strict_code(e)) { x = ((nonzero_quality >> j) & 1u) ? uniform : 0.0; }
                    }
                    const double y = x - c;
                    const double t = sigma_q + y;
                    c = (t - sigma_q) - y;
                    sigma_q = t;
                }
                return sigma_q;
            };
            if(shared) {
                for(int b = threadIdx.x; b < n_barcode; b += BLOCK) {
                    const double sigma_q = kahan_of(b);
                    shared_sigma[b] = sigma_q;
                    shared_p[b] = phred_probability_fast(sigma_q, log2_base_hi, log2_base_lo) * concentration[b];
                }
                __syncthreads();
                if(threadIdx.x != 0) { continue; }
            }
#pragma unroll 2
            for(int b = 0; b < n_barcode; ++b) {
                const double sigma_q = shared ? shared_sigma[b] : kahan_of(b);
                const double prior = concentration[b];
                const double p = shared ? shared_p[b] : phred_probability_fast(sigma_q, log2_base_hi, log2_base_lo) * prior;
                const double y = p - compensation;
                const double t = sigma_p + y;
                compensation = (t - sigma_p) - y;
                sigma_p = t;
                const bool clears = p * (1.0 - 1e-15) > running;                    // a new maximum well above the old one
                const bool close = !clears && p >= running * (1.0 - 1e-15) && p > 0;
                if(clears) { near_index = b; near_sigma = sigma_q; near_prior = prior; ambiguous = false; }
                ambiguous = ambiguous || (close && (sigma_q != near_sigma || prior != near_prior));
                running = p > running ? p : running;
            }
            // best_summand: the winner's term as it entered the sum.  The reference's posterior divides one of the
            // summands by the sum (pamld.cpp:61-92), so two tied barcodes and no noise give exactly 1/2; dividing the
            // carefully rounded power by a sum of quickly rounded ones would miss that by an ulp and flip a decision
            // that sits exactly on the threshold.
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
                    const double sigma_q = shared ? shared_sigma[b] : kahan_of(b);
                    const double p = shared ? shared_p[b] : phred_probability_fast(sigma_q, log2_base_hi, log2_base_lo) * concentration[b];
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
                const uint32_t* winner = d.barcode_packed + static_cast<int64_t>(best) * W;
#pragma unroll
                for(int j = 0; j < L; ++j) {
                    const uint32_t e = (winner[j >> 3] >> (4 * (j & 7))) & 0xfu;
                    const bool mismatch = observed[j] != e;
                    distance += mismatch ? 1 : 0;
                    hq_distance += (mismatch && static_cast<int>(quality[j]) >= d.high_quality_threshold) ? 1 : 0;
                }
            }

            double confidence = best_summand / sigma_p;                            // pamld.cpp:92
            int bin = best + 1;
            uint8_t status = STATUS_PASS;
            if(best_conditional > d.random_barcode_probability) {                 // pamld.cpp:96
                if(confidence > d.confidence_threshold) {                          // pamld.cpp:99
                    if(d.high_quality_distance_threshold > 0 && hq_distance >= d.high_quality_distance_threshold) {
                        r.qcfail = true;
                    }
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

// instantiate <L, true> and <L, false> for L in [LO, HI] and dispatch on the decoder
template <int LO, int HI>
int launch_pamld_literal_range(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    if constexpr(LO > HI) {
        (void)args; (void)host_decoder; (void)grid; (void)lds_bytes; (void)stream;
        return static_cast<int>(hipErrorInvalidValue);
    } else {
        if(host_decoder.nucleotide_cardinality == LO) {
            if(host_decoder.all_strict) { hipLaunchKernelGGL((pamld_literal_kernel<LO, true>), grid, dim3(BLOCK), lds_bytes, stream, args); }
            else { hipLaunchKernelGGL((pamld_literal_kernel<LO, false>), grid, dim3(BLOCK), lds_bytes, stream, args); }
            return static_cast<int>(hipGetLastError());
        }
        return launch_pamld_literal_range<LO + 1, HI>(args, host_decoder, grid, lds_bytes, stream);
    }
}

}  // namespace phq
