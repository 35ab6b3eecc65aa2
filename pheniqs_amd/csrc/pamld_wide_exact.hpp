// K1w (exact tier, long barcodes): the reads the two word product form (pamld_wide.hpp) could not certify — a second
// plausible barcode, ties, noise reads — decided in fp64 before the literal kernel sees them.
//
//   reference pamld.cpp:37-123   posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164  P(r|b) = pow(base, sum of Phred costs), distance, high quality distance
//
// The method of pamld_cross_exact_kernel's table form (pamld_cross.hpp), over barcodes of two words instead of segment
// words: per read one fp64 table of 4 entries per PAIR of positions (entry m = the product of the pair's match or
// mismatch factors by the bits of m; 16 entry tables per group of four would leave room for two waves per CU, and the tier
// is bound by latency: measured 0.34 ms on the 20 nt pairs of the sci-RNA-seq shape), so a barcode's P(r|b) is L / 2 table
// reads and as many multiplications;
// every barcode's prior * P enters the sum, the maximum and a list — a bit mask over the barcodes, in registers: the
// barcode's ordinal is wave uniform — of the barcodes within 1e-10 of the maximum, the only ones the reference's argmax
// can fall on.  Those are evaluated LITERALLY (the reference's Kahan recurrence over the positions and the 2^x of the
// literal kernel) and compared as the reference's loop does (strict '>' in barcode order).  The sum is a plain fp64 sum
// (~1e-14), so only a confidence within 1e-12 of its threshold goes on to the literal kernel.
// 320 to 512 bytes of tables per lane: workgroups of one wave, three or four per CU; the tier sees ~1.5 % of the reads.
#pragma once

#include <type_traits>

#include "kernel_common.hpp"
#include "pamld_wide.hpp"

namespace phq {
namespace wide {

constexpr int EXACT_LANES = 64;
constexpr int EXACT_MAX_BARCODES = 2048;      // one list bit each: sixteen mask registers up to 512 barcodes (what the wide product form takes), thirty two and sixty four past them (round 6)
__host__ __device__ constexpr size_t exact_table_bytes(int groups) { return static_cast<size_t>(groups) * 8u * sizeof(double) * EXACT_LANES; }      // two pair tables of four entries per group

template <int G, int EXACT_MASK_WORDS = 16>
__global__ __launch_bounds__(EXACT_LANES) void pamld_wide_exact_kernel(StageArgs args) {
    constexpr int WORDS = 2;
    static_assert(G >= 5 && G <= 8, "17 to 32 nucleotides");
    static_assert(EXACT_LANES * sizeof(double) == 512, "an entry of a table is 512 bytes");
    __shared__ double factor_double[FACTOR_ROWS][2];                          // probabilities [row][0 match | 1 mismatch]
    __shared__ double cost_double[FACTOR_ROWS][2];                            // Phred costs, same layout (phred.cpp:44-68)
    __shared__ uint32_t kept;                                                 // reads left in the segment for the literal tier
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw);
    DecoderRef d = *constant_cast(args.decoder);
    const int n_barcode = d.n_barcode;
    // the barcodes' two words (for the candidate walk) and their priors behind the tables, sized by the codec: as static
    // arrays for 512 barcodes they took 8 KiB, and with them three of these one wave workgroups fit a CU, not four — the
    // tier is latency bound (pamld_cross.hpp, PHQ_EXACT_PAD_LDS)
    double* const lds_prior = reinterpret_cast<double*>(lds_raw + ((static_cast<size_t>(histogram_words(d)) * sizeof(unsigned long long) + 15u) & ~static_cast<size_t>(15u)) + exact_table_bytes(G));
    uint32_t* const lds_word = reinterpret_cast<uint32_t*>(lds_prior + n_barcode);

    for(int i = threadIdx.x; i < FACTOR_ROWS; i += EXACT_LANES) {
        factor_double[i][0] = i < 128 ? d.match_probability[i] : d.uniform_probability;
        factor_double[i][1] = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        cost_double[i][0] = i < 128 ? d.true_positive_quality[i] : d.uniform_base_quality;
        cost_double[i][1] = i < 128 ? static_cast<double>(i) : d.uniform_base_quality;
    }
    for(int b = threadIdx.x; b < n_barcode; b += EXACT_LANES) {
        lds_word[2 * b] = d.barcode_wide[2 * b];
        lds_word[2 * b + 1] = d.barcode_wide[2 * b + 1];
        lds_prior[b] = d.concentration[b];
    }
    if(threadIdx.x == 0) { kept = 0; }
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const int64_t n = static_cast<int64_t>(args.worklist_count[blockIdx.x]);
    const int64_t tiles = (n + EXACT_LANES - 1) / EXACT_LANES;
    const int L = d.nucleotide_cardinality;
    uint32_t full_mask[WORDS] = {0u, 0u};                                     // the flag bit of each of the L positions
#pragma unroll
    for(int j = 0; j < 4 * G; ++j) { if(j < L) { full_mask[j >> 4] |= 1u << low_bit(j); } }
    constexpr double near = 1.0 - 1e-10;
    const char* const factor_bytes = reinterpret_cast<const char*>(&factor_double[0][0]);
    const char* const cost_bytes = reinterpret_cast<const char*>(&cost_double[0][0]);
    // this lane's column of the per read tables [group][entry][lane], behind the histogram in the dynamic block
    double* const group_table = reinterpret_cast<double*>(lds_raw + ((static_cast<size_t>(histogram_words(d)) * sizeof(unsigned long long) + 15u) & ~static_cast<size_t>(15u))) + threadIdx.x;
    const char* const table_bytes = reinterpret_cast<const char*>(group_table);
    const PHQ_CONSTANT uint32_t* const word_list = constant_cast(d.barcode_wide);
    const PHQ_CONSTANT double* const prior_list = constant_cast(d.concentration);

    uint32_t entry_ahead = tiles > 0 ? segment[min(static_cast<int64_t>(threadIdx.x), n - 1)] : 0u;
    for(int64_t tile = 0; tile < tiles; ++tile) {
        const bool active = tile * EXACT_LANES + threadIdx.x < n;
        const int64_t i = static_cast<int64_t>(entry_ahead);
        {   // the next tile's entries a tile ahead; what this tile compacts lands below the entries already read
            const int64_t next = (tile + 1 < tiles ? tile + 1 : tile) * EXACT_LANES + threadIdx.x;
            entry_ahead = segment[next < n ? next : n - 1];
        }
        __syncthreads();
        ReadState r = load_read_state(args, i);
        const uint32_t* const code_row = reinterpret_cast<const uint32_t*>(args.base + i * (4 * G));
        const uint32_t* const quality_row = reinterpret_cast<const uint32_t*>(args.quality + i * (4 * G));
        uint32_t code_words[G];
        uint32_t quality_words[G];
#pragma unroll
        for(int g = 0; g < G; ++g) { code_words[g] = code_row[g]; quality_words[g] = quality_row[g]; }

        uint32_t row_offset[4 * G];            // byte offset of the position's factor / cost row
        uint32_t observed[WORDS] = {0u, 0u};
        uint32_t outside[WORDS] = {0u, 0u};
        uint32_t high_quality[WORDS] = {0u, 0u};
#pragma unroll
        for(int j = 0; j < 4 * G; ++j) {
            const uint32_t c = (code_words[j >> 2] >> (8 * (j & 3))) & 0xffu;
            const uint32_t q8 = (quality_words[j >> 2] >> (8 * (j & 3))) & 0xffu;
            const uint32_t q = q8 & 0x7fu;
            // outside ACGT: the uniform quality against every barcode base (phred.cpp:62-64); absent: row 0 = 1, 1
            const uint32_t k = strict_code(c) ? q : (q ? 128u : 0u);
            row_offset[j] = k * 16u;
            const uint32_t two = (0x30210u >> (2 * (c & 0xfu))) & 3u;                // A,C,G,T (1,2,4,8) -> 0..3
            observed[j >> 4] |= ((two & 1u) << low_bit(j)) | ((two >> 1) << high_bit(j));
            if(static_cast<int>(q8) >= d.high_quality_threshold) { high_quality[j >> 4] |= 1u << low_bit(j); }
            if(!strict_code(c)) { outside[j >> 4] |= 1u << low_bit(j); }
        }
        // tables: entry m of pair t = product over positions 2 t, 2 t + 1 of the match (bit clear) or mismatch factor
#pragma unroll
        for(int t = 0; t < 2 * G; ++t) {
            const double* const first = reinterpret_cast<const double*>(factor_bytes + row_offset[2 * t]);
            const double* const second = reinterpret_cast<const double*>(factor_bytes + row_offset[2 * t + 1]);
            const double m0 = first[0], e0 = first[1], m1 = second[0], e1 = second[1];
            group_table[(t * 4 + 0) * EXACT_LANES] = m0 * m1;
            group_table[(t * 4 + 1) * EXACT_LANES] = e0 * m1;
            group_table[(t * 4 + 2) * EXACT_LANES] = m0 * e1;
            group_table[(t * 4 + 3) * EXACT_LANES] = e0 * e1;
        }
        // mismatch flags of a barcode: bit low_bit(j) of word j / 16 = position j differs (or the base is outside ACGT)
        auto flags_of = [&](uint32_t first, uint32_t second, uint32_t (&flags)[WORDS]) {
            const uint32_t x0 = observed[0] ^ first;
            const uint32_t x1 = observed[1] ^ second;
            flags[0] = ((x0 | __builtin_amdgcn_alignbit(x0, x0, 4)) | outside[0]) & full_mask[0];
            flags[1] = ((x1 | __builtin_amdgcn_alignbit(x1, x1, 4)) | outside[1]) & full_mask[1];
        };
        // pair t = 2 g + h of group g (f = g mod 4 within its word) has its two flags at bits 8 f + 1 + 2 h and the next:
        // entry * 512 bytes = (flags moved to bits 9..10) & 0x600
        auto probability_of = [&](uint32_t first, uint32_t second) {
            uint32_t flags[WORDS];
            flags_of(first, second, flags);
            double p = 1.0;
#pragma unroll
            for(int t = 0; t < 2 * G; ++t) {
                const uint32_t word = flags[t >> 3];
                const int at = 8 * ((t >> 1) & 3) + 1 + 2 * (t & 1);
                const uint32_t moved = at < 9 ? word << (9 - at) : word >> (at - 9);
                const double v = *reinterpret_cast<const double*>(table_bytes + t * 4 * EXACT_LANES * sizeof(double) + (moved & 0x600u));
                p = t == 0 ? v : p * v;
            }
            return p;
        };

        // one pass: the sum, the maximum, and the list of every barcode within 1e-10 of the running maximum (emptied when
        // a new maximum clears the old one by more than that; barcodes that crept in while the maximum was lower are
        // harmless extras: the literal evaluation below decides)
        double top = 0;
        double sum = 0;
        uint32_t list_mask[EXACT_MASK_WORDS];
        int first_listed = 0;
#pragma unroll
        for(int e = 0; e < EXACT_MASK_WORDS; ++e) { list_mask[e] = 0u; }
        auto scan_chunk = [&](auto which) {
            constexpr int c = decltype(which)::value;
            if(32 * c >= n_barcode) { return; }                                         // wave uniform
            const int here = min(n_barcode - 32 * c, 32);
            uint32_t current = 0;
            auto visit = [&](int b, uint32_t bit) {
                const double p = probability_of(word_list[2 * b], word_list[2 * b + 1]) * prior_list[b];      // scalar loads; pamld.cpp:66
                sum += p;
                const bool clears = p * near > top;
                if constexpr (EXACT_MASK_WORDS > 32) {
                    // sixty four words: what was listed in front of b is left where it is and the walk below starts at b (taking it
                    // out of the words in front as the shorter masks do — measured the faster way for those — is 63 selects a barcode)
                    first_listed = clears ? b : first_listed;
                } else {
#pragma unroll
                    for(int e = 0; e < c; ++e) { list_mask[e] = clears ? 0u : list_mask[e]; }
                }
                current = clears ? 0u : current;
                current |= p >= top * near ? bit : 0u;
                top = fmax(top, p);
            };
            int k = 0;
            for(; k + 4 <= here; k += 4) {
#pragma unroll
                for(int h = 0; h < 4; ++h) { visit(32 * c + k + h, 1u << (k + h)); }
            }
            for(; k < here; ++k) { visit(32 * c + k, 1u << k); }
            list_mask[c] = current;
        };
        for_each_chunk(scan_chunk, std::make_integer_sequence<int, EXACT_MASK_WORDS>{});
        bool defer = !(top > 0);

        // the listed barcode at or after ordinal `from`, -1 when there is none
        auto next_listed = [&](int from) {
            int found = -1;
#pragma unroll
            for(int e = EXACT_MASK_WORDS - 1; e >= 0; --e) {
                const int low = from - 32 * e;
                uint32_t m = list_mask[e];
                m = low >= 32 ? 0u : low > 0 ? m & (~0u << low) : m;
                found = m ? 32 * e + __ffs(static_cast<int>(m)) - 1 : found;
            }
            return found;
        };
        // the candidates literally: Kahan sum of the costs over the barcode's positions, then pow (barcode.h:131-164),
        // in barcode order with the reference's strict '>' (pamld.cpp:66-79)
        int best = -1;
        double best_conditional = 0;
        double best_p = -1.0;
        uint32_t best_flags[WORDS] = {0u, 0u};
        for(int b = defer ? -1 : next_listed(first_listed); b >= 0; b = next_listed(b + 1)) {
            uint32_t flags[WORDS];
            flags_of(lds_word[2 * b], lds_word[2 * b + 1], flags);
            double sigma = 0;
            double compensation = 0;
#pragma unroll
            for(int j = 0; j < 4 * G; ++j) {
                if(j < L) {
                    const double x = *reinterpret_cast<const double*>(cost_bytes + (row_offset[j] + ((flags[j >> 4] >> low_bit(j)) & 1u) * 8u));
                    const double y = x - compensation;
                    const double t = sigma + y;
                    compensation = (t - sigma) - y;
                    sigma = t;
                }
            }
            const double conditional = phred_probability(sigma, d.log2_base_hi, d.log2_base_lo);
            const double p = conditional * lds_prior[b];                             // pamld.cpp:66
            if(p > best_p) {
                best = b;
                best_p = p;
                best_conditional = conditional;
                best_flags[0] = flags[0];
                best_flags[1] = flags[1];
            }
        }
        if(best < 0) { defer = true; best = 0; best_p = 0; }

        int distance = __popc(best_flags[0]) + __popc(best_flags[1]);
        int hq_distance = __popc(best_flags[0] & high_quality[0]) + __popc(best_flags[1] & high_quality[1]);
        const double total = sum + d.adjusted_noise_probability;                     // pamld.cpp:61-86
        double confidence = best_p / total;                                          // pamld.cpp:92
        const bool is_noise = !(best_conditional > d.random_barcode_probability);    // pamld.cpp:96, on the literal value
        if(!is_noise && !(fabs(confidence - d.confidence_threshold) > 1e-12)) { defer = true; }
        if(!defer && active) {
            int bin = best + 1;
            uint8_t status = STATUS_PASS;
            if(!is_noise) {
                if(confidence > d.confidence_threshold) {                            // pamld.cpp:99
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
        // what is left goes back to the front of the segment (few reads: one LDS atomic each); it never holds more
        // leftovers than entries already read from it
        if(active && defer) {
            const uint32_t slot = atomicAdd(&kept, 1u);
            segment[slot] = static_cast<uint32_t>(i);
        }
        count_outgoing(args, histogram, active && !defer, r);
    }
    __syncthreads();
    if(threadIdx.x == 0) { args.worklist_count[blockIdx.x] = kept; }
    histogram_flush(args, d, histogram);
}

template <int G, int MASK_WORDS>
int launch_exact_masked(const StageArgs& args, int n_barcode, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    const size_t bytes = ((histogram_bytes + 15u) & ~static_cast<size_t>(15u)) + exact_table_bytes(G) + static_cast<size_t>(n_barcode) * (sizeof(double) + 2 * sizeof(uint32_t));
    if(bytes > 64u * 1024u) {
        static std::atomic<uint64_t> prepared{0};
        // at most 32 KiB of histogram + 32 KiB of tables + 16 KiB of words and priors; the kernel also has ~4 KiB of static LDS
        if(const int status = allow_large_lds(&pamld_wide_exact_kernel<G, MASK_WORDS>, prepared, 128 * 1024)) { return status; }
    }
    hipLaunchKernelGGL((pamld_wide_exact_kernel<G, MASK_WORDS>), grid, dim3(EXACT_LANES), bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}

// (the 64 word masks — 1 025 to 2 048 barcodes — are instantiated in a unit of their own, pamld_wide_exact_big.hip: they take as
// long to compile as the others together)
template <int G>
int launch_exact_groups(const StageArgs& args, int n_barcode, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return n_barcode <= 512 ? launch_exact_masked<G, 16>(args, n_barcode, grid, histogram_bytes, stream)
                            : launch_exact_masked<G, 32>(args, n_barcode, grid, histogram_bytes, stream);
}

}  // namespace wide
}  // namespace phq
