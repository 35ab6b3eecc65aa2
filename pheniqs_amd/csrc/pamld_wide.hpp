// K1 (product form, long barcodes): PamlDecoder<T>::classify for ACGT barcodes of 17 to 32 nt — paired dual
// indices that are NOT every combination of their segments (those factorise: pamld_cross.hpp), long cellular
// barcodes.  Same method as pamld_lut.hpp, which has the argument and the certificate; what differs:
//
//   * a barcode is TWO words of bit-sliced 2-bit bases (16 positions each); position j of a word has its low
//     code bit at 8 (j / 4) + 1 + j % 4 and its high bit 4 places up (wrapping), so X | rotr(X, 4) leaves
//     index * 2 of group f in bits 1..4 of byte f;
//   * 5 to 8 sixteen entry tables per read = 320 to 512 bytes of LDS per lane, so a workgroup is 128 lanes and
//     the tables are [group][index][128 lanes]: byte 1 of an entry address is (lane >> 6) | index << 1, i.e. a
//     byte of the masked field word OR-ed with a per lane constant — one v_perm_b32 per address — and the
//     group is the instruction's immediate offset.  The 5 group shape (17-20 nt: the 10+10 dual index) takes
//     192 lanes instead — two workgroups = 12 waves per CU rather than three = 6 — and pays a bit field extract
//     and a multiply-add per address for its 768 byte entry rows;
//   * the packed barcode words and the single precision priors sit in LDS and are read at wave uniform
//     addresses (one ds_read_b128 serves two barcodes), not in a wave wide register.
//
//   reference pamld.cpp:37-123, barcode.h:131-164, phred.cpp:44-68
#pragma once

#include "kernel_common.hpp"

namespace phq {
namespace wide {

// others <= dominance(G) * winner (pamld_lut.hpp): 9e-9 over the worst relative error of the single precision sum
__host__ __device__ constexpr double dominance(int groups) { return 9e-9 / (groups * 4.3e-7 + 1e-6); }
constexpr float RUNNER_UP = 0.9999f;          // runner up <= RUNNER_UP * winner: the argmax is certain in single precision
constexpr float SMALLEST_CERTAIN = 1e-30f;    // ... for a winner well inside the normal range (a subnormal one has lost its digits)
constexpr int FACTOR_ROWS = 129;              // 0..127 = quality of an ACGT base, 128 = a base outside ACGT with q >= 1
// NARROW: the 5 group shape on 128 lanes as well — for a consumer whose codec's histogram leaves 192 lane workgroups only
// one per CU (196 barcodes: 84 KiB; 64 KiB on 128 lanes, two per CU)
template <int G, bool NARROW = false> struct Geometry {
    static constexpr int THREADS = (G == 5 && !NARROW) ? 192 : 128;
    static constexpr uint32_t ENTRY_STEP = THREADS * sizeof(float);
    static constexpr uint32_t GROUP_STEP = 16u * ENTRY_STEP;
};
inline int threads_of(int groups) { return groups == 5 ? 192 : 128; }
constexpr int FOLD = 8;                       // barcodes per block of the scan
constexpr uint32_t FIELD_MASK = 0x1e1e1e1eu;

__host__ __device__ constexpr uint32_t low_bit(int position) { return 8u * ((position >> 2) & 3) + 1u + (position & 3); }
__host__ __device__ constexpr uint32_t high_bit(int position) { return (low_bit(position) + 4u) & 31u; }

// the two code bits in two plane words, every other bit zero (pamld_lut.hpp): fields = (lo ^ lo') | (hi ^ hi'), and
// what is OR-ed into the unused bits of the observed low plane (a lane's share of the table address) rides along
__device__ __forceinline__ uint32_t low_plane(uint32_t sliced) { return sliced & FIELD_MASK; }
__device__ __forceinline__ uint32_t high_plane(uint32_t sliced) { return __builtin_amdgcn_alignbit(sliced, sliced, 4) & FIELD_MASK; }

typedef __attribute__((address_space(3))) const float* LdsFloats;

// one hand laid dynamic LDS block at LDS address 0 (no static LDS; pamld_lut.hpp)
struct Layout {
    uint32_t words_at;        // padded barcodes x 2 words x 2 planes
    uint32_t prior_at;        // padded barcodes floats
    uint32_t match_at;
    uint32_t mismatch_at;
    uint32_t deferred_at;
    uint32_t histogram_at;
};
__host__ __device__ inline Layout layout_of(int groups, int threads, int padded_barcodes) {
    Layout l;
    l.words_at = static_cast<uint32_t>(groups) * 16u * static_cast<uint32_t>(threads) * 4u;
    l.prior_at = l.words_at + static_cast<uint32_t>(padded_barcodes) * 16u;
    l.match_at = l.prior_at + static_cast<uint32_t>(padded_barcodes) * 4u;
    l.mismatch_at = l.match_at + 528u;
    l.deferred_at = l.mismatch_at + 528u;
    l.histogram_at = l.deferred_at + 16u;
    return l;
}

// G = groups of 4 positions = words of the plane row (plane stride = 4 G bytes, zero filled past the barcode)
// CONSUMER: the kernel visits the reads an earlier tier (pamld_pair.hpp) left in this workgroup's worklist segment
// instead of every read, and compacts what it defers itself back into the segment (pamld_lut.hpp).
template <int G, bool CONSUMER, bool NARROW>
__global__ __launch_bounds__((Geometry<G, NARROW>::THREADS)) void pamld_wide_kernel(StageArgs args) {
    constexpr int THREADS = Geometry<G, NARROW>::THREADS;
    constexpr uint32_t GROUP_STEP = Geometry<G, NARROW>::GROUP_STEP;
    constexpr uint32_t ENTRY_STEP = Geometry<G, NARROW>::ENTRY_STEP;
    constexpr bool PERMUTED = THREADS == 128;
    constexpr int WORDS = (G + 3) / 4;        // words of 16 positions: 2
    static_assert(G >= 5 && G <= 8, "17 to 32 nucleotides");
    // Round 5: the tables are scaled by 2^(96 / G) a group, every barcode's probability by 2^(G (96 / G)): twelve Q37 mismatches
    // of a 20 nt barcode are 4e-45, under a float's normal range — and a noise read, whose argmax is all that matters (it is
    // charged to it and leaves unclassified, pamld.cpp:96,113-119), sits that far from every barcode: every one of them went to
    // the exact tier for a winner the floats had lost (C4's paired decoder: its exact tier 0.27 ms per 50 M reads for the 1 % of
    // noise reads).  The sums are scaled back in fp64 where they meet the winner's own product.
    constexpr int GROUP_SCALE_LOG2 = 96 / G;
    constexpr float GROUP_SCALE = static_cast<float>(1ull << (GROUP_SCALE_LOG2 > 62 ? 62 : GROUP_SCALE_LOG2));
    static_assert(GROUP_SCALE_LOG2 <= 62, "one shift");
    constexpr double PRODUCT_UNSCALE = 1.0 / (static_cast<double>(1ull << (GROUP_SCALE_LOG2 * G / 2)) * static_cast<double>(1ull << (GROUP_SCALE_LOG2 * G - GROUP_SCALE_LOG2 * G / 2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    DecoderRef d = *constant_cast(args.decoder);
    const int n_barcode = d.n_barcode;
    const int padded_barcodes = (n_barcode + FOLD - 1) / FOLD * FOLD;
    const Layout at = layout_of(G, THREADS, padded_barcodes);
    float* const table = reinterpret_cast<float*>(lds_raw);
    uint4* const barcode_words = reinterpret_cast<uint4*>(lds_raw + at.words_at);        // {low 0, low 1, high 0, high 1}
    float* const prior_single = reinterpret_cast<float*>(lds_raw + at.prior_at);
    float* const match_single = reinterpret_cast<float*>(lds_raw + at.match_at);
    float* const mismatch_single = reinterpret_cast<float*>(lds_raw + at.mismatch_at);
    // the fp64 factor rows stay in global memory (20 to 32 cached loads per read, once): with them in LDS three
    // workgroups of the 20 nt shape would no longer fit a CU
    const double* const factor_pairs = d.factor_pairs;
    uint32_t& deferred_total = *reinterpret_cast<uint32_t*>(lds_raw + at.deferred_at);
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw + at.histogram_at);

    for(int i = threadIdx.x; i < FACTOR_ROWS; i += THREADS) {
        const double m = i < 128 ? d.match_probability[i] : d.uniform_probability;
        const double e = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        match_single[i] = static_cast<float>(m);
        mismatch_single[i] = static_cast<float>(e);
    }
    for(int b = threadIdx.x; b < padded_barcodes; b += THREADS) {
        const uint32_t first = d.barcode_wide[2 * b];
        const uint32_t second = d.barcode_wide[2 * b + 1];
        barcode_words[b] = make_uint4(low_plane(first), low_plane(second), high_plane(first), high_plane(second));
        prior_single[b] = d.concentration_single[b];                 // zero for the padding
    }
    if(threadIdx.x == 0) { deferred_total = 0; }
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const int L = d.nucleotide_cardinality;
    const bool uniform_prior = d.uniform_prior != 0;
    const int64_t n = CONSUMER ? static_cast<int64_t>(args.worklist_count[blockIdx.x]) : args.n_read;
    const int64_t tiles = (n + THREADS - 1) / THREADS;
    const uint32_t lane_low = (threadIdx.x & 63u) << 2;                       // byte 0 of an entry address
    const uint32_t lane_high_bytes = ((threadIdx.x >> 6) & 1u) * 0x01010101u;  // bit 0 of byte 1 of an entry address
    const uint32_t lane_offset = threadIdx.x * static_cast<uint32_t>(sizeof(float));
    uint32_t full_mask[WORDS];                                                // the flag bit of each of the L positions
#pragma unroll
    for(int w = 0; w < WORDS; ++w) { full_mask[w] = 0; }
#pragma unroll
    for(int j = 0; j < 4 * G; ++j) { if(j < L) { full_mask[j >> 4] |= 1u << low_bit(j); } }

    for(int64_t tile = CONSUMER ? 0 : blockIdx.x; tile < tiles; tile += CONSUMER ? 1 : gridDim.x) {
        const bool active = tile * THREADS + threadIdx.x < n;
        int64_t i = active ? tile * THREADS + threadIdx.x : n - 1;
        if(CONSUMER) {
            i = static_cast<int64_t>(segment[i]);
            __syncthreads();                   // every entry of this tile is read before any wave compacts into it
        }
        ReadState r = load_read_state(args, i);
        bool defer = false;
        {
            uint32_t code_words[G];
            uint32_t quality_words[G];
            {
                const uint32_t* code_row = reinterpret_cast<const uint32_t*>(args.base + i * (4 * G));
                const uint32_t* quality_row = reinterpret_cast<const uint32_t*>(args.quality + i * (4 * G));
#pragma unroll
                for(int g = 0; g < G; ++g) { code_words[g] = code_row[g]; quality_words[g] = quality_row[g]; }
            }
            // per position: factor row, single precision factors, 2-bit base, high quality flag
            float fm[4 * G];
            float fe[4 * G];
            uint32_t row[G];
            uint32_t observed[WORDS];
            uint32_t high_quality[WORDS];
            uint32_t outside[WORDS];
#pragma unroll
            for(int w = 0; w < WORDS; ++w) { observed[w] = 0; high_quality[w] = 0; outside[w] = 0; }
#pragma unroll
            for(int g = 0; g < G; ++g) { row[g] = 0; }
#pragma unroll
            for(int j = 0; j < 4 * G; ++j) {
                const uint32_t c = (code_words[j >> 2] >> (8 * (j & 3))) & 0xffu;
                const uint32_t q8 = (quality_words[j >> 2] >> (8 * (j & 3))) & 0xffu;
                const uint32_t q = q8 & 0x7fu;
                // outside ACGT: the uniform quality against every barcode base (phred.cpp:62-64); absent: row 0 = 1, 1
                const uint32_t k = strict_code(c) ? q : (q ? 128u : 0u);
                fm[j] = match_single[k];
                fe[j] = mismatch_single[k];
                row[j >> 2] |= k << (8 * (j & 3));
                const uint32_t two = (0x30210u >> (2 * (c & 0xfu))) & 3u;                // A,C,G,T (1,2,4,8) -> 0..3
                observed[j >> 4] |= ((two & 1u) << low_bit(j)) | ((two >> 1) << high_bit(j));
                if(static_cast<int>(q8) >= d.high_quality_threshold) { high_quality[j >> 4] |= 1u << low_bit(j); }
                if(!strict_code(c)) { outside[j >> 4] |= 1u << low_bit(j); }
            }
            // tables: entry m of group g = product over the group's positions of (bit ? mismatch : match)
#pragma unroll
            for(int g = 0; g < G; ++g) {
                const int base = 4 * g;
                float lo[4];
                float hi[4];
                lo[0] = fm[base] * fm[base + 1];
                lo[1] = fe[base] * fm[base + 1];
                lo[2] = fm[base] * fe[base + 1];
                lo[3] = fe[base] * fe[base + 1];
                // (round 5: every group's entries scaled by 2^(96 / G) — see GROUP_SCALE — four multiplications a group)
                hi[0] = (fm[base + 2] * GROUP_SCALE) * fm[base + 3];
                hi[1] = (fe[base + 2] * GROUP_SCALE) * fm[base + 3];
                hi[2] = (fm[base + 2] * GROUP_SCALE) * fe[base + 3];
                hi[3] = (fe[base + 2] * GROUP_SCALE) * fe[base + 3];
#pragma unroll
                for(int m = 0; m < 16; ++m) {
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(table) + g * GROUP_STEP + m * ENTRY_STEP + lane_offset) = lo[m & 3] * hi[m >> 2];
                }
            }

            uint32_t observed_low[WORDS];
            uint32_t observed_high[WORDS];
#pragma unroll
            for(int w = 0; w < WORDS; ++w) {
                observed_low[w] = low_plane(observed[w]) | (PERMUTED ? lane_high_bytes : 0u);
                observed_high[w] = high_plane(observed[w]);
            }
            // P(r|b) of one barcode from its plane words; also leaves the mismatch fields
            auto probability_of = [&](const uint4 quad, uint32_t (&fields)[WORDS]) {
                fields[0] = (observed_low[0] ^ quad.x) | (observed_high[0] ^ quad.z);
                fields[1] = (observed_low[1] ^ quad.y) | (observed_high[1] ^ quad.w);
                float value = 1.0f;
#pragma unroll
                for(int g = 0; g < G; ++g) {
                    uint32_t address;
                    if(PERMUTED) {
                        // { 0, 0, fields.byte[g % 4], lane_low.byte[0] }: byte 1 = (lane >> 6) | index << 1
                        address = __builtin_amdgcn_perm(fields[g >> 2], lane_low, 0x0c0c0400u | (static_cast<uint32_t>(g & 3) << 8));
                    } else {
                        const uint32_t index = __builtin_amdgcn_ubfe(fields[g >> 2], 8 * (g & 3) + 1, 4);
                        address = index * ENTRY_STEP + lane_offset;                         // v_mad_u32_u24
                    }
                    const float entry = ((LdsFloats)static_cast<uintptr_t>(address))[g * (GROUP_STEP / sizeof(float))];
                    value = g == 0 ? entry : value * entry;
                }
                return value;
            };

            // single precision scan, eight barcodes at a time (pamld_lut.hpp has the bookkeeping's rationale)
            float others_sum = 0;
            float held_sum = 0;
            float best_value = 0;
            float runner_up = 0;           // the largest block maximum but the running one
            int best_fold = -1;
            const int n_fold = padded_barcodes / FOLD;
            auto values_of_block = [&](int f, float (&p)[FOLD]) {
#pragma unroll
                for(int k = 0; k < FOLD; ++k) {
                    uint32_t fields[WORDS];
                    p[k] = probability_of(barcode_words[FOLD * f + k], fields);           // wave uniform address in the scan
                }
                if(!uniform_prior) {
#pragma unroll
                    for(int k = 0; k < FOLD; ++k) { p[k] *= prior_single[FOLD * f + k]; }
                }
                if(f == n_fold - 1) {
#pragma unroll
                    for(int k = 1; k < FOLD; ++k) { if(FOLD * f + k >= n_barcode) { p[k] = 0.0f; } }
                }
            };
            for(int f = 0; f < n_fold; ++f) {
                float p[FOLD];
                values_of_block(f, p);
                const float block_max = fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7])));
                const float block_sum = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
                const bool takes_over = block_max > best_value;                          // strict: the first block keeps a tie
                runner_up = __builtin_amdgcn_fmed3f(best_value, block_max, runner_up);   // runner_up <= best_value: the median
                others_sum += takes_over ? held_sum : block_sum;
                held_sum = takes_over ? block_sum : held_sum;
                best_fold = takes_over ? f : best_fold;
                best_value = takes_over ? block_max : best_value;
            }

            // inside the winning block (pamld_lut.hpp): the winner is the first member EQUAL to the maximum the scan
            // found, the rest the sum of the members that are not; two members equal to the maximum show in the block's
            // sum and hand the read on
            int best = -1;
            uint32_t best_flags[WORDS];
#pragma unroll
            for(int w = 0; w < WORDS; ++w) { best_flags[w] = 0; }
            float sigma = others_sum;
            const float winner_single = best_value;
            {
                const int block = best_fold < 0 ? 0 : best_fold;
                float p[FOLD];
                values_of_block(block, p);
                int within = -1;
                float q[FOLD];
#pragma unroll
                for(int k = FOLD - 1; k >= 0; --k) {
                    const bool is_top = p[k] == best_value;
                    within = is_top ? k : within;
                    q[k] = is_top ? 0.0f : p[k];
                }
                const float rest = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
                const float second = fmaxf(fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3])), fmaxf(fmaxf(q[4], q[5]), fmaxf(q[6], q[7])));
                const bool crowded = held_sum - rest >= 1.5f * best_value;               // the block's sum holds the maximum twice
                runner_up = fmaxf(runner_up, crowded ? best_value : second);
                sigma += crowded ? best_value : rest;
                if(best_fold >= 0 && best_value > 0.0f && within >= 0) {
                    best = FOLD * block + within;
                    const uint4 quad = barcode_words[best];
                    best_flags[0] = (((observed_low[0] ^ quad.x) | (observed_high[0] ^ quad.z)) | outside[0]) & full_mask[0];
                    best_flags[1] = (((observed_low[1] ^ quad.y) | (observed_high[1] ^ quad.w)) | outside[1]) & full_mask[1];
                }
            }

            if(best < 0) {
                defer = true;
            } else {
                int distance = 0;
                int hq_distance = 0;
#pragma unroll
                for(int w = 0; w < WORDS; ++w) {
                    distance += __popc(best_flags[w]);
                    hq_distance += __popc(best_flags[w] & high_quality[w]);
                }
                double best_conditional = 1.0;
#pragma unroll
                for(int j = 0; j < 4 * G; ++j) {
                    const uint32_t k = (row[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    best_conditional *= factor_pairs[2 * k + ((best_flags[j >> 4] >> low_bit(j)) & 1u)];
                }
                const double best_p = best_conditional * d.concentration[best];
                const double others = (uniform_prior ? static_cast<double>(sigma) * d.concentration[best] : static_cast<double>(sigma)) * PRODUCT_UNSCALE;
                const double total = (best_p + others) + d.adjusted_noise_probability;   // pamld.cpp:61-86
                double confidence = best_p / total;                                       // pamld.cpp:92
                const double noise_margin = best_conditional - d.random_barcode_probability;
                const double confidence_margin = confidence - d.confidence_threshold;
                // a read that clearly fails the noise filter only needs its argmax (it is charged to it and leaves
                // unclassified, pamld.cpp:96,113-119): the runner up further than single precision can blur;
                // otherwise the posterior matters: a dominant winner, no decision within 1e-9 of its threshold
                const bool noise_clear = fabs(noise_margin) > 1e-9 * d.random_barcode_probability;
                defer = !noise_clear
                     || (noise_margin <= 0 ? !(winner_single >= SMALLEST_CERTAIN && runner_up <= RUNNER_UP * winner_single)
                                           : (!(others <= dominance(G) * best_p) || !(fabs(confidence_margin) > 2e-8)));
                if(!defer && active) {
                    int bin = best + 1;
                    uint8_t status = STATUS_PASS;
                    if(noise_margin > 0) {                                                // pamld.cpp:96
                        if(confidence_margin > 0) {                                       // pamld.cpp:99
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
                        stream_result(&args.result[i], out);
                    }
                    stream_read_state(args, i, r);
                } else if(active) {
                    stream_placeholder(args, i, r);                                   // the next tier overwrites it
                }
            }
        }
        // deferred reads: one LDS add per wave reserves a run of slots in this workgroup's worklist segment
        const unsigned long long deferred = __ballot(active && defer);
        if(deferred) {
            const int lane = threadIdx.x & 63;
            uint32_t first = 0;
            if(lane == 0) { first = atomicAdd(&deferred_total, static_cast<uint32_t>(__popcll(deferred))); }
            first = __shfl(first, 0);
            if(active && defer) {
                segment[first + __popcll(deferred & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(i);
            }
        }
        count_outgoing(args, histogram, active && !defer, r);
    }
    __syncthreads();
    if(threadIdx.x == 0) { args.worklist_count[blockIdx.x] = deferred_total; }
    histogram_flush(args, d, histogram);
}

template <int G, bool CONSUMER, bool NARROW>
int launch_shape(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    static std::atomic<uint64_t> prepared{0};
    if(const int status = allow_large_lds(&pamld_wide_kernel<G, CONSUMER, NARROW>, prepared)) { return status; }
    const int padded = (host_decoder.n_barcode + FOLD - 1) / FOLD * FOLD;
    const size_t lds_bytes = layout_of(G, Geometry<G, NARROW>::THREADS, padded).histogram_at + histogram_bytes;
    hipLaunchKernelGGL((pamld_wide_kernel<G, CONSUMER, NARROW>), grid, dim3(Geometry<G, NARROW>::THREADS), lds_bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}

template <int G, bool CONSUMER>
int launch_role(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if constexpr(G == 5 && CONSUMER) {
        // a consumer's workgroup size is its own business (it walks a worklist segment): narrower if that keeps two per CU
        const int padded = (host_decoder.n_barcode + FOLD - 1) / FOLD * FOLD;
        const size_t broad = layout_of(G, Geometry<G, false>::THREADS, padded).histogram_at + histogram_bytes;
        const size_t narrow = layout_of(G, Geometry<G, true>::THREADS, padded).histogram_at + histogram_bytes;
        if((160u * 1024u) / broad < 2 && (160u * 1024u) / narrow >= 2) { return launch_shape<G, CONSUMER, true>(args, host_decoder, grid, histogram_bytes, stream); }
    }
    return launch_shape<G, CONSUMER, false>(args, host_decoder, grid, histogram_bytes, stream);
}

template <int G>
int launch_groups(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return args.worklist_role == 2 ? launch_role<G, true>(args, host_decoder, grid, histogram_bytes, stream)
                                   : launch_role<G, false>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace wide
}  // namespace phq
