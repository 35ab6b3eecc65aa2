// K1 (factorised product form): PamlDecoder<T>::classify for a COMBINATORIAL barcode set — every combination
// of S per segment sets (dual / triple indexing: U1 x U2 [x U3] barcodes), uniform prior.
//
//   reference pamld.cpp:37-123     posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164    P(r|b) = pow(base, sum over ALL segments' positions of the Phred cost)
//
// pow(base, sum) factorises over the segments, so for b = (u1, .., uS)
//     P(r|b)        = P1(u1) * .. * PS(uS)
//     sum_b P(r|b)  = (sum P1) * .. * (sum PS)
//     argmax_b      = (argmax P1, .., argmax PS)
// and U1 + .. + US per segment evaluations replace U1 * .. * US per barcode ones (192 instead of 9 216 for a
// 96 x 96 dual index).  Each segment is scanned with the machinery of pamld_lut.hpp: per read 16 entry tables
// of 4-position factor products in LDS, bit-sliced 2-bit bases, X | rotr(X, 4) mismatch fields, the segment's
// distinct words in a wave wide register read with v_readlane.  The segments reuse the same LDS tables one
// after the other.
//
// A read is FINISHED here only when (pamld_lut.hpp has the argument)
//   * every segment's best word beats its runner up by more than single precision can blur (1e-4), so the
//     argmax is certain, and
//   * either the winner clearly fails the noise filter (then nothing but the argmax matters: the read is
//     charged to it and leaves unclassified, pamld.cpp:96,113-119), or the winner dominates (everything
//     else <= 5e-4 of it: the posterior, recomputed in fp64 for the winner, holds ~1e-9) and the confidence
//     is not within 1e-9 of its threshold.
// Every other read goes to this workgroup's worklist segment for the literal kernel.
#pragma once

#include <cstdlib>
#include <type_traits>

#include "kernel_common.hpp"

namespace phq {
namespace cross {

constexpr double DOMINANCE = 5e-4;
// the weighted scans: their single precision sums carry ~2.3e-6 of `others` (pamld_lut.hpp, dominance), so others up to
// this share of the winner move the posterior by 4e-9; the bounded part adds at most 2e-9 (PHQ_MODE_FAST: 1e-8)
constexpr double DOMINANCE_WEIGHTED = 1.5e-3;
constexpr double ELSEWHERE_WEIGHTED = 4e-9;
constexpr float RUNNER_UP = 0.9999f;          // second <= RUNNER_UP * top: the argmax of a segment is certain
constexpr float SMALLEST_CERTAIN = 1e-30f;    // ... for a top well inside the normal range (a subnormal one has lost its digits)
constexpr int FACTOR_ROWS = 129;              // 0..127 = quality of an ACGT base, 128 = a base outside ACGT with q >= 1
constexpr int THREADS = 256;
constexpr int TABLE_REGISTERS = CROSS_MAX_UNIQUE / 64;

__host__ __device__ constexpr uint32_t low_bit(int position) { return 8u * (position >> 2) + 2u + (position & 3); }
__host__ __device__ constexpr uint32_t high_bit(int position) { return (low_bit(position) + 4u) & 31u; }

__device__ __forceinline__ uint32_t mismatch_fields(uint32_t x) {
    return x | __builtin_amdgcn_alignbit(x, x, 4);                  // x | rotr(x, 4)
}
// the two code bits in two plane words, every other bit zero (pamld_lut.hpp): fields = (lo ^ lo') | (hi ^ hi')
__device__ __forceinline__ uint32_t low_plane(uint32_t sliced) { return sliced & 0x3c3c3c3cu; }
__device__ __forceinline__ uint32_t high_plane(uint32_t sliced) { return __builtin_amdgcn_alignbit(sliced, sliced, 4) & 0x3c3c3c3cu; }

typedef __attribute__((address_space(3))) const float* LdsFloats;         // an LDS address as the hardware takes it
// bytes of the scan kernel's LDS block in front of the counter histogram: tables, factor columns, worklist counter
__host__ __device__ constexpr uint32_t scan_fixed_bytes(int groups, int segments) {
    return static_cast<uint32_t>(groups) * 16u * THREADS * 4u + 1056u + 2u * static_cast<uint32_t>(segments) * CROSS_MAX_UNIQUE * 4u + 16u;
}

// G = groups of 4 positions of the longest segment (segments are padded with absent positions: factor 1 either way)
//
// WEIGHTED (two segments only): the priors differ between the barcodes (estimated priors, phq_adjust_prior), so the
// sum over the barcodes does not factorise.  Three scans instead of two: segment 0 plain, which names u0; segment 1
// with every word weighted by prior[u0][.], which names u1 and gives the exact rest of ROW u0; segment 0 again
// weighted by prior[.][u1], the exact rest of COLUMN u1 (and u0 must win again).  What lies in neither is bounded
// by the largest prior times the two plain rests and has to be negligible (1e-9 of the winner) for the read to be
// finished here.
template <int G, int S, bool WEIGHTED>
__global__ __launch_bounds__(THREADS) void pamld_cross_kernel(StageArgs args) {
    static_assert(!WEIGHTED || S == 2, "weighted scans are for two segment codecs");
    constexpr int PASSES = WEIGHTED ? 3 : S;
    constexpr uint32_t GROUP_BYTES = 16 * THREADS * sizeof(float);            // [index][256 lanes]
    constexpr uint32_t ENTRY_BYTES = THREADS * sizeof(float);
    constexpr uint32_t INDEX_MASK = 0x3c00u;                                  // index * 1024 at byte 1 of a field word
    // one hand laid dynamic LDS block at LDS address 0 and no static LDS, as in pamld_lut.hpp: a table address is
    // (field & mask) | lane offset, one instruction
    constexpr uint32_t MATCH_AT = G * GROUP_BYTES;                            // FACTOR_ROWS x {match, mismatch} floats
    constexpr uint32_t WORDS_AT = MATCH_AT + 1056;                            // the segments' distinct words
    constexpr uint32_t DEFERRED_AT = WORDS_AT + 2 * S * CROSS_MAX_UNIQUE * 4;  // two planes per word
    static_assert(DEFERRED_AT + 16 == scan_fixed_bytes(G, S), "launch_segments sizes the block");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    float* const table = reinterpret_cast<float*>(lds_raw);                   // [group][index][lane], one segment at a time
    float2* const factor_single = reinterpret_cast<float2*>(lds_raw + MATCH_AT);
    // [segment][block of 8 words][low plane words | high plane words]
    uint32_t* const lds_words = reinterpret_cast<uint32_t*>(lds_raw + WORDS_AT);
    uint32_t& deferred_total = *reinterpret_cast<uint32_t*>(lds_raw + DEFERRED_AT);
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw + DEFERRED_AT + 16);
    DecoderRef d = *constant_cast(args.decoder);
    // the fp64 factor rows stay in global memory (4 G S cached loads per read, once): with them in LDS a third
    // workgroup of the 10 + 10 nt shape would not fit a CU
    const double* const factor_pairs = d.factor_pairs;

    for(int i = threadIdx.x; i < FACTOR_ROWS; i += THREADS) {
        const double m = i < 128 ? d.match_probability[i] : d.uniform_probability;
        const double e = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        factor_single[i] = make_float2(static_cast<float>(m), static_cast<float>(e));
    }
    for(int w = threadIdx.x; w < S * CROSS_MAX_UNIQUE; w += THREADS) {                                        // zero past a segment's words
        const uint32_t sliced = d.cross_words[w];
        const int at = 16 * (w / 8) + (w % 8);
        lds_words[at] = low_plane(sliced);
        lds_words[at + 8] = high_plane(sliced);
    }
    if(threadIdx.x == 0) { deferred_total = 0; }
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    // quality >= high quality threshold, byte parallel (pamld_lut.hpp): thresholds of 0..128 (pamld_cross_supports)
    const int high_quality_threshold = min(max(d.high_quality_threshold, 0), 128);
    const uint32_t high_quality_addend = static_cast<uint32_t>(128 - high_quality_threshold) * 0x01010101u;

    // per segment, wave uniform: geometry, byte masks of the padded words, flag mask
    int first_word[S];
    uint32_t byte_shift[S];
    uint32_t byte_mask[S][G];
    uint32_t full_mask[S];
    int unique[S];
    int radix[S];
    const int row_words = d.plane_stride >> 2;
#pragma unroll
    for(int s = 0; s < S; ++s) {
        const int start = d.cross_start[s];
        const int length = d.cross_length[s];
        first_word[s] = start >> 2;
        byte_shift[s] = static_cast<uint32_t>(start & 3);
        unique[s] = d.cross_unique[s];
        radix[s] = d.cross_stride[s];
        full_mask[s] = 0;
#pragma unroll
        for(int j = 0; j < 4 * G; ++j) { if(j < length) { full_mask[s] |= 1u << low_bit(j); } }
#pragma unroll
        for(int g = 0; g < G; ++g) {
            const int valid = length - 4 * g;
            byte_mask[s][g] = valid >= 4 ? 0xffffffffu : valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u);
        }
    }

    // behind the neighbour list kernel of combinatorial sets (pamld_cross_near.hpp) the kernel is a CONSUMER: it visits
    // the reads of this workgroup's worklist segment and compacts what it defers itself back into it (pamld_lut.hpp)
    const bool consumer = args.worklist_role == 2;
    const int64_t n = consumer ? static_cast<int64_t>(args.worklist_count[blockIdx.x]) : args.n_read;
    const int64_t tiles = (n + THREADS - 1) / THREADS;
    const uint32_t lane_offset = threadIdx.x * static_cast<uint32_t>(sizeof(float));
    const int64_t stride = d.plane_stride;

    for(int64_t tile = consumer ? 0 : blockIdx.x; tile < tiles; tile += consumer ? 1 : gridDim.x) {
        // every lane stays enabled (cross lane reads of the wave wide tables): lanes past the end of the batch
        // recompute its last read and are kept from every store and counter
        const bool active = tile * THREADS + threadIdx.x < n;
        int64_t i = active ? tile * THREADS + threadIdx.x : n - 1;
        if(consumer) {
            i = static_cast<int64_t>(segment[i]);
            __syncthreads();                   // every entry of this tile is read before any wave compacts into it
        }
        ReadState r = load_read_state(args, i);
        const uint32_t* const code_row = reinterpret_cast<const uint32_t*>(args.base + i * stride);
        const uint32_t* const quality_row = reinterpret_cast<const uint32_t*>(args.quality + i * stride);

        // the padded words of every segment: G + 1 aligned words each, funnel shifted to the segment's first byte
        uint32_t code_words[S][G];
        uint32_t quality_words[S][G];
#pragma unroll
        for(int s = 0; s < S; ++s) {
            uint32_t c[G + 1];
            uint32_t q[G + 1];
#pragma unroll
            for(int k = 0; k <= G; ++k) {
                const int w = min(first_word[s] + k, row_words - 1);          // never past the row
                c[k] = code_row[w];
                q[k] = quality_row[w];
            }
#pragma unroll
            for(int k = 0; k < G; ++k) {
                code_words[s][k] = __builtin_amdgcn_alignbyte(c[k + 1], c[k], byte_shift[s]) & byte_mask[s][k];
                quality_words[s][k] = __builtin_amdgcn_alignbyte(q[k + 1], q[k], byte_shift[s]) & byte_mask[s][k];
            }
        }

        double best_conditional = 1.0;         // the winner's P(r|b), fp64
        double winner_product = 1.0;           // the same from the single precision scan
        double others = 0.0;                   // sum over every other combination of P(r|b)
        int flat = 0;                          // mixed radix index of the winning combination
        int distance = 0;
        int hq_distance = 0;
        bool certain = true;
        // weighted scans: what each pass leaves for the combination below
        float pass_top[PASSES];                // the winner's value (weighted in the weighted passes)
        float pass_plain[PASSES];              // the winner's P, without the weight
        float pass_rest[PASSES];               // the sum of the other words (weighted in the weighted passes)
        float pass_second[PASSES];             // the runner up (weighted in the weighted passes)
        float pass_plain_sum[PASSES];          // the plain sum over every word of the segment
        float pass_plain_other[PASSES];        // weighted passes: the largest plain P among the words other than the winner
        int pass_best[PASSES];

#pragma unroll
        for(int pass = 0; pass < PASSES; ++pass) {
            const int s = (WEIGHTED && pass == 2) ? 0 : pass;      // the third weighted pass scans segment 0 again
            const bool weigh = WEIGHTED && pass > 0;
            // the weights of this pass, contiguous and zero padded to CROSS_MAX_UNIQUE: the row of the winner of
            // segment 0 (pass 1) or, from the transposed copy, the column of the winner of segment 1 (pass 2)
            const float* const weights = !WEIGHTED ? nullptr
                                       : pass == 1 ? d.cross_prior + pass_best[0] * CROSS_MAX_UNIQUE
                                       : pass == 2 ? d.cross_prior_transposed + pass_best[1] * CROSS_MAX_UNIQUE : d.cross_prior;
            // per position, four at a time and byte parallel (pamld_lut.hpp): factor row, 2-bit base, flags
            float fm[4 * G];
            float fe[4 * G];
            uint32_t row[G];
            uint32_t observed = 0;
            uint32_t high_quality = 0;
            uint32_t outside = 0;
            auto gather = [](uint32_t planes) {            // bits 8 j and 8 j + 4 -> bits j and 4 + j
                const uint32_t t = planes | (planes >> 7);
                return (t | (t >> 14)) & 0xffu;
            };
#pragma unroll
            for(int g = 0; g < G; ++g) {
                const uint32_t valid = byte_mask[s][g];                              // the bytes of positions inside the segment
                const uint32_t x = code_words[s][g] & 0x0f0f0f0fu;
                const uint32_t q8 = quality_words[s][g];
                const uint32_t q7 = q8 & 0x7f7f7f7fu;
                const uint32_t z = x & ((x | 0x80808080u) - 0x01010101u);
                const uint32_t strict = (x + 0x7f7f7f7fu) & ~(z + 0x7f7f7f7fu) & 0x80808080u;
                const uint32_t strict_bytes = (strict - (strict >> 7)) | strict;
                // outside ACGT: the uniform quality against every barcode base (row 128), nothing at quality 0 or for
                // an absent position (row 0 holds 1 and 1)   phred.cpp:62-64
                const uint32_t nonzero = (q7 + 0x7f7f7f7fu) & 0x80808080u;
                row[g] = (q7 & strict_bytes) | (nonzero & ~strict_bytes);
                const uint32_t codes = gather((((x >> 1) | (x >> 3)) & 0x01010101u) | (((x << 2) | (x << 1)) & 0x10101010u));
                observed |= __builtin_amdgcn_alignbit(codes, codes, (32 - (8 * g + 2)) & 31);
                const uint32_t high = ((q7 + high_quality_addend) | q8) & 0x80808080u & valid;
                const uint32_t flags = gather(((~strict & 0x80808080u & valid) >> 7) | (high >> 3));
                outside |= (flags & 0xfu) << (8 * g + 2);
                high_quality |= (flags >> 4) << (8 * g + 2);
            }
#pragma unroll
            for(int j = 0; j < 4 * G; ++j) {
                const float2 pair = factor_single[(row[j >> 2] >> (8 * (j & 3))) & 0xffu];
                fm[j] = pair.x;
                fe[j] = pair.y;
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
                hi[0] = fm[base + 2] * fm[base + 3];
                hi[1] = fe[base + 2] * fm[base + 3];
                hi[2] = fm[base + 2] * fe[base + 3];
                hi[3] = fe[base + 2] * fe[base + 3];
#pragma unroll
                for(int m = 0; m < 16; ++m) {
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(table) + g * GROUP_BYTES + m * ENTRY_BYTES + lane_offset) = lo[m & 3] * hi[m >> 2];
                }
            }

            // scan the segment's distinct words eight at a time: the block's maximum and sum; across blocks the
            // running maximum, which block holds it, that block's sum set aside, the sum of every other block and the
            // largest maximum of the others (pamld_lut.hpp); then the winner inside its block
            const uint32_t observed_low = low_plane(observed);
            const uint32_t observed_high = high_plane(observed);
            auto value_of = [&](uint32_t low_word, uint32_t high_word) {
                const uint32_t fields = (observed_low ^ low_word) | (observed_high ^ high_word);
                float value = 1.0f;
#pragma unroll
                for(int g = 0; g < G; ++g) {
                    const uint32_t source = g == 0 ? fields << 8 : g == 1 ? fields : g == 2 ? fields >> 8 : fields >> 16;
                    const float entry = ((LdsFloats)static_cast<uintptr_t>((source & INDEX_MASK) | lane_offset))[g * (GROUP_BYTES / sizeof(float))];
                    value = g == 0 ? entry : value * entry;
                }
                return value;
            };
            constexpr int FOLD = 8;
            const uint32_t* const segment_words = lds_words + 2 * s * CROSS_MAX_UNIQUE;
            const uint4* const blocks = reinterpret_cast<const uint4*>(segment_words);
            const int n_block = (unique[s] + FOLD - 1) / FOLD;
            float others_sum = 0;
            float held_sum = 0;
            float best_value = 0;
            float second = 0;
            float plain_sum = 0;
            float plain_top = 0;               // weighted passes: the two largest plain P
            float plain_second = 0;
            int best_block = -1;
            auto values_of_block = [&](int f, float (&p)[FOLD], float (&plain)[FOLD]) {
                const uint4 a = blocks[4 * f];                                      // wave uniform address: a broadcast
                const uint4 b = blocks[4 * f + 1];
                const uint4 c = blocks[4 * f + 2];                                  // the high planes of the same eight words
                const uint4 e = blocks[4 * f + 3];
                plain[0] = value_of(a.x, c.x); plain[1] = value_of(a.y, c.y); plain[2] = value_of(a.z, c.z); plain[3] = value_of(a.w, c.w);
                plain[4] = value_of(b.x, e.x); plain[5] = value_of(b.y, e.y); plain[6] = value_of(b.z, e.z); plain[7] = value_of(b.w, e.w);
                if(f == n_block - 1) {
#pragma unroll
                    for(int k = 1; k < FOLD; ++k) { if(FOLD * f + k >= unique[s]) { plain[k] = 0.0f; } }
                }
#pragma unroll
                for(int k = 0; k < FOLD; ++k) { p[k] = plain[k]; }
                if(weigh) {
                    const float4 low = *reinterpret_cast<const float4*>(weights + FOLD * f);
                    const float4 high = *reinterpret_cast<const float4*>(weights + FOLD * f + 4);
                    p[0] *= low.x; p[1] *= low.y; p[2] *= low.z; p[3] *= low.w;
                    p[4] *= high.x; p[5] *= high.y; p[6] *= high.z; p[7] *= high.w;
                }
            };
            for(int f = 0; f < n_block; ++f) {
                float p[FOLD];
                float plain[FOLD];
                values_of_block(f, p, plain);
                if(WEIGHTED) {
                    plain_sum += ((plain[0] + plain[1]) + (plain[2] + plain[3])) + ((plain[4] + plain[5]) + (plain[6] + plain[7]));
                    if(weigh) {
#pragma unroll
                        for(int k = 0; k < FOLD; ++k) {
                            plain_second = __builtin_amdgcn_fmed3f(plain_top, plain[k], plain_second);      // second <= top: the median
                            plain_top = fmaxf(plain_top, plain[k]);
                        }
                    }
                }
                const float block_max = fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7])));
                const float block_sum = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
                const bool takes_over = block_max > best_value;                     // strict: the first block keeps a tie
                second = __builtin_amdgcn_fmed3f(best_value, block_max, second);    // second <= best_value: the median (pamld_lut.hpp)
                others_sum += takes_over ? held_sum : block_sum;
                held_sum = takes_over ? block_sum : held_sum;
                best_block = takes_over ? f : best_block;
                best_value = takes_over ? block_max : best_value;
            }
            // Inside the winning block (pamld_lut.hpp): the same instructions give the same eight values again, the
            // winner is the first member EQUAL to the maximum the scan found, the rest is the sum of the members that are
            // not.  Two members equal to the maximum: the block's sum less that rest is twice the maximum, not once; such a
            // read is handed on by raising the runner up to the winner.
            const float top = best_value;
            float top_plain = 0;
            float rest = others_sum;
            int best = -1;
            uint32_t flags_of_best = 0;
            {
                const int block = best_block < 0 ? 0 : best_block;
                float p[FOLD];
                float plain[FOLD];
                values_of_block(block, p, plain);
                int within = -1;
                float q[FOLD];
#pragma unroll
                for(int k = FOLD - 1; k >= 0; --k) {
                    const bool is_top = p[k] == best_value;
                    within = is_top ? k : within;
                    top_plain = is_top ? plain[k] : top_plain;
                    q[k] = is_top ? 0.0f : p[k];
                }
                const float inner_rest = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
                const float inner_second = fmaxf(fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3])), fmaxf(fmaxf(q[4], q[5]), fmaxf(q[6], q[7])));
                const bool crowded = held_sum - inner_rest >= 1.5f * best_value;   // the block's sum holds the maximum twice
                second = fmaxf(second, crowded ? best_value : inner_second);
                rest += crowded ? best_value : inner_rest;
                if(best_block >= 0 && best_value > 0.0f && within >= 0) { best = FOLD * block + within; }
                const int at = 16 * (best < 0 ? 0 : best / 8) + (best < 0 ? 0 : best % 8);
                flags_of_best = (observed_low ^ segment_words[at]) | (observed_high ^ segment_words[at + 8]);
            }

            if(best < 0) { certain = false; best = 0; }       // underflow: everything rounded to zero (flags_of_best is word 0's)
            pass_top[pass] = top;
            pass_plain[pass] = top_plain;
            pass_rest[pass] = rest;
            pass_second[pass] = second;
            pass_plain_sum[pass] = plain_sum;
            pass_plain_other[pass] = plain_top == top_plain ? plain_second : plain_top;      // two words at the top: the second equals it
            pass_best[pass] = best;
            if(pass < S) {
                if(!WEIGHTED) { certain = certain && top >= SMALLEST_CERTAIN && (second <= RUNNER_UP * top); }
                // the winner's mismatch flags, its distances and its probability again in fp64
                const uint32_t flags = (flags_of_best | outside) & full_mask[s];
                distance += __popc(flags);
                hq_distance += __popc(flags & high_quality);
#pragma unroll
                for(int j = 0; j < 4 * G; ++j) {
                    const uint32_t k = (row[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    best_conditional *= factor_pairs[2 * k + ((flags >> low_bit(j)) & 1u)];
                }
                // (winner + others) * (top + rest) - winner * top, every term positive
                others = others * (static_cast<double>(top) + static_cast<double>(rest)) + winner_product * static_cast<double>(rest);
                winner_product *= static_cast<double>(top);
                flat += best * radix[s];
            }
        }

        bool defer = !certain;
        {
            const int best = d.cross_ordinal[flat];
            const double prior = WEIGHTED ? d.concentration[best] : d.concentration[0];
            const double best_p = best_conditional * prior;
            bool elsewhere_clear = true;
            if(WEIGHTED) {
                // the winner (u0, u1): P0 = pass_top[0], P1 = pass_plain[1]; its row and its column are exact, the
                // rest is bounded.  `others` becomes the sum of prior_b P(r|b) over every other barcode, in units of p.
                const double p0 = pass_top[0];
                const double p1 = pass_plain[1];
                const double winner = static_cast<double>(pass_top[1]) * p0;                   // prior * P0 * P1, single precision
                const double row_rest = p0 * static_cast<double>(pass_rest[1]);                // u0 fixed, u1 elsewhere
                const double column_rest = p1 * static_cast<double>(pass_rest[2]);             // u1 fixed, u0 elsewhere
                // the plain rest of segment 1, bounded: the difference of two single precision sums plus their rounding
                const double plain_rest = fmax(static_cast<double>(pass_plain_sum[1]) - p1, 0.0) + 2e-7 * static_cast<double>(pass_plain_sum[1]);
                const double elsewhere = d.cross_prior_max * static_cast<double>(pass_rest[0]) * plain_rest;
                others = row_rest + column_rest + 0.5 * elsewhere;                             // the midpoint of [0, elsewhere]
                // the argmax: the column scan must name u0 again, and nothing in the row, the column or elsewhere
                // may come within single precision of the winner
                // (elsewhere: at most the largest prior, the runner up of segment 0 and the largest P of segment 1 off the winner's word)
                const double runner_up = fmax(fmax(p0 * static_cast<double>(pass_second[1]), p1 * static_cast<double>(pass_second[2])),
                                              d.cross_prior_max * static_cast<double>(pass_second[0]) * fmin(plain_rest, static_cast<double>(pass_plain_other[1]) * (1.0 + 1e-6)));
                if(pass_best[2] != pass_best[0] || !(winner >= 1e-30) || !(pass_top[0] >= SMALLEST_CERTAIN) || !(pass_plain[1] >= SMALLEST_CERTAIN) || !(runner_up <= static_cast<double>(RUNNER_UP) * winner)) { defer = true; }
                elsewhere_clear = elsewhere <= ELSEWHERE_WEIGHTED * winner;          // the posterior: only a read past the noise filter needs it
            }
            const double total = (best_p + (WEIGHTED ? others : others * prior)) + d.adjusted_noise_probability;   // pamld.cpp:61-86
            double confidence = best_p / total;                                       // pamld.cpp:92
            const double noise_margin = best_conditional - d.random_barcode_probability;
            const double confidence_margin = confidence - d.confidence_threshold;
            const bool noise_clear = fabs(noise_margin) > 1e-9 * d.random_barcode_probability;
            const bool is_noise = noise_margin <= 0;
            if(!noise_clear) { defer = true; }
            if(!is_noise) {
                // the posterior matters: a dominant winner and no decision within 1e-9 of the threshold
                if(!elsewhere_clear || !(others <= (WEIGHTED ? DOMINANCE_WEIGHTED * best_p : DOMINANCE * best_conditional)) || !(fabs(confidence_margin) > (WEIGHTED ? 1e-8 : 1e-9))) { defer = true; }
            }
            if(!defer && active) {
                int bin = best + 1;
                uint8_t status = STATUS_PASS;
                if(!is_noise) {                                                       // pamld.cpp:96
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

// Tier 1.5: the reads the single precision scan could not certify, decided exactly but still factorised.
//
// Consumes this workgroup's worklist segment and compacts what it cannot decide either back into the same
// segment for the literal kernel (in place: a barrier per tile keeps every wave's reads ahead of the writes).
//   * per segment, every distinct word's P in fp64 (products of the fp64 factor rows): the sum, the maximum and
//     the list (in LDS, one column per lane) of the words within 1e-10 of the maximum — the only ones the
//     reference's argmax can fall on: its own P(r|b) = pow(base, Kahan sum) is within ~1e-13 of the exact product;
//   * the candidates (combinations of near maximal words) are evaluated LITERALLY — the reference's Kahan
//     recurrence over the barcode's positions and the 2^x of the literal kernel (pamld_literal.hpp) — and
//     compared the way the reference's loop does: strict '>' in barcode order, so the lowest ordinal among
//     equal maxima wins;
//   * sum_b P(r|b) is the product of the per segment fp64 sums (relative error ~1e-15; the reference's Kahan sum
//     over all barcodes is not reproduced bit for bit, so a confidence within 1e-12 of its threshold is left to
//     the literal kernel — the only reads it still sees).
// UNIQUE = distinct words a segment may hold: CROSS_MAX_UNIQUE for the combinatorial sets; a single segment set (a
// codec that is not combinatorial, its barcode taken as one word of up to 16 nt) may hold EXACT_SINGLE_UNIQUE.
// PRIORS (single segment sets only): the barcodes' priors differ, every word's probability carries its prior through
// the sum, the maximum and the list, as the reference's loop does (pamld.cpp:66-79).
//
// A word's P is not multiplied up position by position but read from per read fp64 tables, one of 16 entries per group
// of 4 positions ([group][entry][lane] in the dynamic LDS block behind the histogram, built once per read and segment:
// 28 multiplications a group) — G table reads and G - 1 multiplications a word instead of 4 G of each (the products are
// associated by group: ~1e-16, far inside the 1e-10 window); and the list is a bit mask of UNIQUE bits in registers, the
// word ordinal being wave uniform.  (Round 1 multiplied the factors up and kept per lane lists of ordinals in LDS:
// 53 instructions per read and word against 27.)
constexpr int EXACT_TABLE_LANES = 128;        // lanes of a workgroup: 16 KiB of tables per group of 4 positions
// Round 3, three and four groups (9 to 16 nt words): the tables are by PAIR of positions — 4 entries a pair, 8 KiB a group of
// four positions for the workgroup instead of 16 KiB — as in pamld_wide_exact.hpp: twice the table reads and multiplications
// per word, and five workgroups on a CU where the 16 entry tables left two (48 KiB of them for 10 nt words).  The tier is
// latency bound: padded down from three workgroups a CU to two (PHQ_EXACT_PAD_LDS) the 8 nt shape takes 1.6 times as long.
// C3: 0.94 -> 0.63 ms per 100 M reads.  Up to 8 nt the 16 entry tables stay: with pairs C2's tier took 0.20 instead of 0.19 ms.
__host__ __device__ constexpr bool exact_pair_tables(int groups) { return groups >= 3; }
__host__ __device__ constexpr size_t exact_table_bytes(int groups) { return static_cast<size_t>(groups) * (exact_pair_tables(groups) ? 8u : 16u) * sizeof(double) * EXACT_TABLE_LANES; }

template <int G, int S, int UNIQUE, bool PRIORS>
__global__ __launch_bounds__(EXACT_TABLE_LANES) void pamld_cross_exact_kernel(StageArgs args) {
    static_assert(UNIQUE == CROSS_MAX_UNIQUE || S == 1, "long word lists are for single segment sets");
    static_assert(!PRIORS || S == 1, "priors by word are for single segment sets");
    static_assert(UNIQUE % 32 == 0, "whole mask words");
    constexpr int LANES = EXACT_TABLE_LANES;
    constexpr int MASK_WORDS = UNIQUE / 32;
    __shared__ double factor_double[FACTOR_ROWS][2];                          // probabilities [row][0 match | 1 mismatch]
    __shared__ double cost_double[FACTOR_ROWS][2];                            // Phred costs, same layout (phred.cpp:44-68)
    __shared__ uint32_t group_start[EXACT_GROUP_MAX + 1];                     // prefix sums of the group's segment counts
    __shared__ uint32_t group_kept[EXACT_GROUP_MAX];                          // reads left in each segment for the literal tier
    // what the candidate walk indexes per lane, out of LDS instead of chains of dependent global loads: the distinct words,
    // and for a single segment the barcode of each word
    __shared__ uint32_t lds_word[S == 1 ? UNIQUE : S * CROSS_MAX_UNIQUE];
    __shared__ int32_t lds_ordinal[S == 1 ? UNIQUE : 1];
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw);
    DecoderRef d = *constant_cast(args.decoder);

    for(int i = threadIdx.x; i < FACTOR_ROWS; i += LANES) {
        factor_double[i][0] = i < 128 ? d.match_probability[i] : d.uniform_probability;
        factor_double[i][1] = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        cost_double[i][0] = i < 128 ? d.true_positive_quality[i] : d.uniform_base_quality;
        cost_double[i][1] = i < 128 ? static_cast<double>(i) : d.uniform_base_quality;
    }
    for(int w = threadIdx.x; w < (S == 1 ? UNIQUE : S * CROSS_MAX_UNIQUE); w += LANES) { lds_word[w] = d.cross_words[w]; }
    if(S == 1) {
        for(int w = threadIdx.x; w < UNIQUE; w += LANES) { lds_ordinal[w] = w < d.cross_unique[0] ? d.cross_ordinal[w * d.cross_stride[0]] : 0; }
    }
    // the segments of this workgroup, as one list
    const int group = args.worklist_group < 1 ? 1 : (args.worklist_group > EXACT_GROUP_MAX ? EXACT_GROUP_MAX : args.worklist_group);
    const int first_segment = static_cast<int>(blockIdx.x) * group;
    const int n_segment_here = min(group, args.worklist_segments - first_segment);
    if(threadIdx.x == 0) {
        uint32_t total = 0;
        for(int s = 0; s < EXACT_GROUP_MAX; ++s) {
            group_start[s] = total;
            group_kept[s] = 0;
            if(s < n_segment_here) { total += args.worklist_count[first_segment + s]; }
        }
        group_start[EXACT_GROUP_MAX] = total;
    }
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    uint32_t* const segments = args.worklist + static_cast<int64_t>(first_segment) * args.worklist_stride;
    const int64_t n = static_cast<int64_t>(group_start[EXACT_GROUP_MAX]);
    const int64_t tiles = (n + LANES - 1) / LANES;

    int first_word[S];
    int length[S];
    uint32_t byte_shift[S];
    uint32_t byte_mask[S][G];
    uint32_t full_mask[S];
    int unique[S];
    int radix[S];
    const int row_words = d.plane_stride >> 2;
#pragma unroll
    for(int s = 0; s < S; ++s) {
        const int start = d.cross_start[s];
        length[s] = d.cross_length[s];
        first_word[s] = start >> 2;
        byte_shift[s] = static_cast<uint32_t>(start & 3);
        unique[s] = d.cross_unique[s];
        radix[s] = d.cross_stride[s];
        full_mask[s] = 0;
#pragma unroll
        for(int j = 0; j < 4 * G; ++j) { if(j < length[s]) { full_mask[s] |= 1u << low_bit(j); } }
#pragma unroll
        for(int g = 0; g < G; ++g) {
            const int valid = length[s] - 4 * g;
            byte_mask[s][g] = valid >= 4 ? 0xffffffffu : valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u);
        }
    }
    // this lane's column of the per read tables, behind the histogram in the dynamic block
    double* const group_table = reinterpret_cast<double*>(lds_raw + ((static_cast<size_t>(histogram_words(d)) * sizeof(unsigned long long) + 15u) & ~static_cast<size_t>(15u))) + threadIdx.x;
    const PHQ_CONSTANT uint32_t* const word_list = constant_cast(d.cross_words);
    const int64_t stride = d.plane_stride;
    constexpr double near = 1.0 - 1e-10;
    const double uniform_concentration = d.concentration[0];          // !PRIORS: every barcode's prior is this one
    const char* const factor_bytes = reinterpret_cast<const char*>(&factor_double[0][0]);
    const char* const cost_bytes = reinterpret_cast<const char*>(&cost_double[0][0]);

    // entry k of the list: in which segment, and where in it
    auto entry_of = [&](int64_t tile, int& home) {
        const int64_t k = tile * LANES + threadIdx.x;
        const uint32_t at = static_cast<uint32_t>(k < n ? k : n - 1);
        home = 0;
#pragma unroll
        for(int s = 1; s < EXACT_GROUP_MAX; ++s) { home += at >= group_start[s] ? 1 : 0; }       // starts are non decreasing
        return segments[static_cast<int64_t>(home) * args.worklist_stride + (at - group_start[home])];
    };
    int home_ahead = 0;
    uint32_t entry_ahead = tiles > 0 ? entry_of(0, home_ahead) : 0u;
    for(int64_t tile = 0; tile < tiles; ++tile) {
        const bool active = tile * LANES + threadIdx.x < n;
        const int home = home_ahead;
        uint32_t* const segment = segments + static_cast<int64_t>(home) * args.worklist_stride;
        const int64_t i = static_cast<int64_t>(entry_ahead);
        // the next tile's entries are read a tile ahead (their latency hides behind this tile's work); what this tile
        // compacts into a segment lands below the entries already read from it
        entry_ahead = entry_of(tile + 1 < tiles ? tile + 1 : tile, home_ahead);
        __syncthreads();                       // every entry of this tile is read before any wave compacts into it
        ReadState r = load_read_state(args, i);
        const uint32_t* const code_row = reinterpret_cast<const uint32_t*>(args.base + i * stride);
        const uint32_t* const quality_row = reinterpret_cast<const uint32_t*>(args.quality + i * stride);

        uint32_t row_offset[S][4 * G];         // byte offset of the position's factor / cost row
        uint32_t observed[S];
        uint32_t outside[S];
        uint32_t high_quality[S];
        int near_count[S];                     // listed words
        uint32_t list_mask[S][MASK_WORDS];     // the list: bit u of the segment's mask = word u is listed
        double sum_all = 1.0;                  // product of the per segment sums = sum over every barcode of P(r|b)
        bool defer = false;

#pragma unroll
        for(int s = 0; s < S; ++s) {
            uint32_t c[G + 1];
            uint32_t q[G + 1];
#pragma unroll
            for(int g = 0; g <= G; ++g) {
                const int w = min(first_word[s] + g, row_words - 1);
                c[g] = code_row[w];
                q[g] = quality_row[w];
            }
            observed[s] = 0;
            outside[s] = 0;
            high_quality[s] = 0;
#pragma unroll
            for(int j = 0; j < 4 * G; ++j) {
                const uint32_t code_word = __builtin_amdgcn_alignbyte(c[(j >> 2) + 1], c[j >> 2], byte_shift[s]) & byte_mask[s][j >> 2];
                const uint32_t quality_word = __builtin_amdgcn_alignbyte(q[(j >> 2) + 1], q[j >> 2], byte_shift[s]) & byte_mask[s][j >> 2];
                const uint32_t code = (code_word >> (8 * (j & 3))) & 0xffu;
                const uint32_t q8 = (quality_word >> (8 * (j & 3))) & 0xffu;
                const uint32_t quality = q8 & 0x7fu;
                const uint32_t row = strict_code(code) ? quality : (quality ? 128u : 0u);
                row_offset[s][j] = row * 16u;
                const uint32_t two = (0x30210u >> (2 * (code & 0xfu))) & 3u;
                observed[s] |= ((two & 1u) << low_bit(j)) | ((two >> 1) << high_bit(j));
                if(static_cast<int>(q8) >= d.high_quality_threshold) { high_quality[s] |= 1u << low_bit(j); }
                if(!strict_code(code)) { outside[s] |= 1u << low_bit(j); }
            }
            double top = 0;
            double sum = 0;
            int count = 0;
            {
                // this read's tables of the segment: entry m of group g = product over the group's positions of the
                // match (bit clear) or mismatch (bit set) factor
                if(exact_pair_tables(G)) {
#pragma unroll
                    for(int pr = 0; pr < 2 * G; ++pr) {
                        const double* const first = reinterpret_cast<const double*>(factor_bytes + row_offset[s][2 * pr]);
                        const double* const second = reinterpret_cast<const double*>(factor_bytes + row_offset[s][2 * pr + 1]);
                        const double m0 = first[0], e0 = first[1], m1 = second[0], e1 = second[1];
                        group_table[(pr * 4 + 0) * LANES] = m0 * m1;
                        group_table[(pr * 4 + 1) * LANES] = e0 * m1;
                        group_table[(pr * 4 + 2) * LANES] = m0 * e1;
                        group_table[(pr * 4 + 3) * LANES] = e0 * e1;
                    }
                } else {
#pragma unroll
                for(int g = 0; g < G; ++g) {
                    double entry[16];
#pragma unroll
                    for(int jj = 0; jj < 4; ++jj) {
                        const double* const pair = reinterpret_cast<const double*>(factor_bytes + row_offset[s][4 * g + jj]);
                        const double match = pair[0];
                        const double mismatch = pair[1];
                        if(jj == 0) { entry[0] = match; entry[1] = mismatch; }
                        else {
#pragma unroll
                            for(int m = (1 << jj) - 1; m >= 0; --m) {
                                entry[m + (1 << jj)] = entry[m] * mismatch;
                                entry[m] = entry[m] * match;
                            }
                        }
                    }
#pragma unroll
                    for(int m = 0; m < 16; ++m) { group_table[(g * 16 + m) * LANES] = entry[m]; }
                }
                }
                // flags of group g sit at bits 8 g + 2 .. 8 g + 5: entry * LANES * 8 bytes = (flags moved to bits 10..13) & 0x3c00
                static_assert(LANES * sizeof(double) == 1024, "an entry of the table form is 1 KiB");
                const char* const table_bytes = reinterpret_cast<const char*>(group_table);
                auto probability_of = [&](uint32_t word) {
                    const uint32_t flags = (mismatch_fields(observed[s] ^ word) | outside[s]) & full_mask[s];
                    double p = 1.0;
                    if(exact_pair_tables(G)) {
                        // the pair's two flags sit at bits 8 (pr / 2) + 2 + 2 (pr % 2) and the next: moved to bits 10 and 11 = entry * 1 KiB
#pragma unroll
                        for(int pr = 0; pr < 2 * G; ++pr) {
                            const int at_bit = 8 * (pr >> 1) + 2 + 2 * (pr & 1);
                            const uint32_t moved = at_bit < 10 ? flags << (10 - at_bit) : flags >> (at_bit - 10);
                            const double t = *reinterpret_cast<const double*>(table_bytes + pr * 4 * LANES * sizeof(double) + (moved & 0xc00u));
                            p = pr == 0 ? t : p * t;
                        }
                        return p;
                    }
#pragma unroll
                    for(int g = 0; g < G; ++g) {
                        const uint32_t moved = g == 0 ? flags << 8 : g == 1 ? flags : flags >> (8 * g - 8);
                        const double t = *reinterpret_cast<const double*>(table_bytes + g * 16 * LANES * sizeof(double) + (moved & 0x3c00u));
                        p = g == 0 ? t : p * t;
                    }
                    return p;
                };
                uint32_t chunk_mask[MASK_WORDS];
#pragma unroll
                for(int e = 0; e < MASK_WORDS; ++e) { chunk_mask[e] = 0u; }
                const PHQ_CONSTANT uint32_t* const words = word_list + s * CROSS_MAX_UNIQUE;
                auto scan_chunk = [&](auto which) {
                    constexpr int c = decltype(which)::value;
                    if(32 * c >= unique[s]) { return; }                                         // wave uniform
                    const int here = min(unique[s] - 32 * c, 32);
                    uint32_t current = 0;
                    auto visit = [&](int u, uint32_t bit) {
                        double p = probability_of(words[u]);                                    // scalar load
                        if(PRIORS) { p *= d.concentration[d.cross_ordinal[u]]; }
                        sum += p;
                        const bool clears = p * near > top;
#pragma unroll
                        for(int e = 0; e < c; ++e) { chunk_mask[e] = clears ? 0u : chunk_mask[e]; }
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
                    chunk_mask[c] = current;
                };
                for_each_chunk(scan_chunk, std::make_integer_sequence<int, MASK_WORDS>{});
#pragma unroll
                for(int e = 0; e < MASK_WORDS; ++e) { list_mask[s][e] = chunk_mask[e]; count += __popc(chunk_mask[e]); }
            }
            if(!(top > 0)) { defer = true; count = 0; }
            near_count[s] = count;
            sum_all *= sum;
        }

        // the listed word of segment s at or after ordinal `from` (table form), -1 when there is none
        auto next_listed = [&](int s, int from) {
            int found = -1;
#pragma unroll
            for(int e = MASK_WORDS - 1; e >= 0; --e) {
                const int low = from - 32 * e;
                uint32_t m = list_mask[s][e];
                m = low >= 32 ? 0u : low > 0 ? m & (~0u << low) : m;
                found = m ? 32 * e + __ffs(static_cast<int>(m)) - 1 : found;
            }
            return found;
        };
        // walk the combinations of listed words, the last segment fastest
        auto for_each_combination = [&](auto&& body) {
            int choice[S];                         // long lists: the entry; table form: the word's ordinal
            bool more = true;
#pragma unroll
            for(int s = 0; s < S; ++s) { choice[s] = next_listed(s, 0); more = more && near_count[s] > 0; }
            while(more) {
                int flat = 0;
                uint32_t flags[S];
#pragma unroll
                for(int s = 0; s < S; ++s) {
                    const int u = choice[s];
                    flat += u * radix[s];
                    const uint32_t word = lds_word[s * CROSS_MAX_UNIQUE + u];
                    flags[s] = (mismatch_fields(observed[s] ^ word) | outside[s]) & full_mask[s];
                }
                body(flat, flags);
                bool carry = true;
#pragma unroll
                for(int s = S - 1; s >= 0; --s) {
                    if(carry) {
                        const int next = next_listed(s, choice[s] + 1);
                        if(next < 0) { choice[s] = next_listed(s, 0); } else { choice[s] = next; carry = false; }
                    }
                }
                more = !carry;
            }
        };
        // P(r|b) of a combination literally: Kahan sum of the costs over the barcode's positions, then pow (barcode.h:131-164)
        auto literal_of = [&](const uint32_t (&flags)[S]) {
            double sigma = 0;
            double compensation = 0;
#pragma unroll
            for(int s = 0; s < S; ++s) {
#pragma unroll
                for(int j = 0; j < 4 * G; ++j) {
                    if(j < length[s]) {
                        const double x = *reinterpret_cast<const double*>(cost_bytes + (row_offset[s][j] + ((flags[s] >> low_bit(j)) & 1u) * 8u));
                        const double y = x - compensation;
                        const double t = sigma + y;
                        compensation = (t - sigma) - y;
                        sigma = t;
                    }
                }
            }
            return phred_probability(sigma, d.log2_base_hi, d.log2_base_lo);
        };

        int best = 0x7fffffff;
        double best_conditional = -1.0;
        double best_p = -1.0;                  // prior * P(r|b) of the winner, literal
        uint32_t best_flags[S];
#pragma unroll
        for(int s = 0; s < S; ++s) { best_flags[s] = 0; }
        bool any_candidate = true;
#pragma unroll
        for(int s = 0; s < S; ++s) { any_candidate = any_candidate && near_count[s] > 0; }
        for_each_combination([&](int flat, const uint32_t (&flags)[S]) {
            const int b = S == 1 ? lds_ordinal[S == 1 ? flat / radix[0] : 0] : d.cross_ordinal[flat];
            const double prior = PRIORS ? d.concentration[b] : uniform_concentration;
            const double conditional = literal_of(flags);
            const double p = conditional * prior;                                  // pamld.cpp:66
            if(p > best_p || (p == best_p && b < best)) {                          // strict '>' in barcode order
                best = b;
                best_p = p;
                best_conditional = conditional;
#pragma unroll
                for(int s = 0; s < S; ++s) { best_flags[s] = flags[s]; }
            }
        });
        if(!any_candidate || best == 0x7fffffff) { defer = true; best = 0; best_conditional = 0; best_p = 0; }

        int distance = 0;
        int hq_distance = 0;
#pragma unroll
        for(int s = 0; s < S; ++s) {
            distance += __popc(best_flags[s]);
            hq_distance += __popc(best_flags[s] & high_quality[s]);
        }
        const double total = (PRIORS ? sum_all : sum_all * d.concentration[0]) + d.adjusted_noise_probability;   // pamld.cpp:61-86
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
        // what is left goes back to the front of the read's own segment (few reads: one LDS atomic each); a segment
        // never holds more leftovers than entries already read from it, and this tile's entries were read before the
        // barrier at its top
        if(active && defer) {
            const uint32_t slot = atomicAdd(&group_kept[home], 1u);
            segment[slot] = static_cast<uint32_t>(i);
        }
        count_outgoing(args, histogram, active && !defer, r);
    }
    __syncthreads();
    if(static_cast<int>(threadIdx.x) < n_segment_here) { args.worklist_count[first_segment + threadIdx.x] = group_kept[threadIdx.x]; }
    histogram_flush(args, d, histogram);
}

// Tier 1.5 for two segment codecs whose priors differ between the barcodes: sum_b prior_b P(r|b) is the bilinear
// form  sum_u0 P0[u0] * (sum_u1 prior[u0][u1] P1[u1]),  in fp64, over the words that matter: every lane keeps
// in LDS a list of the words of each segment within 1e-15 of the segment's maximum (the others cannot move the
// sum by 1e-12 nor be the argmax; what they could add is bounded and checked), with the P of those of segment 1,
// and walks list against list — per lane lists, so a wave runs as many rounds as its longest lists.  The combinations within 1e-10 of the largest prior * P0 * P1 are
// evaluated literally and compared in barcode order, as in pamld_cross_exact_kernel.
constexpr int WEIGHTED_THREADS = 64;
constexpr int WEIGHTED_SHORT = 16;                // words of segment 1 a lane keeps, with their P
constexpr int WEIGHTED_LIST = 48;                 // words of segment 0 a lane keeps
constexpr int WEIGHTED_PAIRS = 48;                // pairs of listed words a lane walks — one prior from L2 each, one after the other, and the
                                                  // wave waits for its longest lane; reads with more go to pamld_cross_exact_dense_kernel

template <int G>
__global__ __launch_bounds__(WEIGHTED_THREADS) void pamld_cross_exact_weighted_kernel(StageArgs args) {
    constexpr int S = 2;
    __shared__ double listed_value[WEIGHTED_SHORT * WEIGHTED_THREADS];        // [entry][lane] P1 of the listed words of segment 1
    __shared__ uint8_t listed_word[WEIGHTED_SHORT * WEIGHTED_THREADS];        // [entry][lane] their ordinals
    __shared__ uint8_t significant[WEIGHTED_LIST * WEIGHTED_THREADS];         // [entry][lane] words of segment 0 that matter
    __shared__ uint32_t word_copy[2][CROSS_MAX_UNIQUE];                       // the segments' words: every loop below walks them
    __shared__ double factor_double[FACTOR_ROWS][2];
    __shared__ double cost_double[FACTOR_ROWS][2];
    __shared__ uint32_t deferred_total;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw);
    DecoderRef d = *constant_cast(args.decoder);
    for(int i = threadIdx.x; i < FACTOR_ROWS; i += WEIGHTED_THREADS) {
        factor_double[i][0] = i < 128 ? d.match_probability[i] : d.uniform_probability;
        factor_double[i][1] = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        cost_double[i][0] = i < 128 ? d.true_positive_quality[i] : d.uniform_base_quality;
        cost_double[i][1] = i < 128 ? static_cast<double>(i) : d.uniform_base_quality;
    }
    if(threadIdx.x == 0) { deferred_total = 0; }
    for(int i = threadIdx.x; i < 2 * CROSS_MAX_UNIQUE; i += WEIGHTED_THREADS) { word_copy[i / CROSS_MAX_UNIQUE][i % CROSS_MAX_UNIQUE] = d.cross_words[i]; }
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const int64_t n = static_cast<int64_t>(args.worklist_count[blockIdx.x]);
    const int64_t tiles = (n + WEIGHTED_THREADS - 1) / WEIGHTED_THREADS;
    int first_word[S];
    int length[S];
    uint32_t byte_shift[S];
    uint32_t byte_mask[S][G];
    uint32_t full_mask[S];
    int unique[S];
    int radix[S];
    const int row_words = d.plane_stride >> 2;
#pragma unroll
    for(int s = 0; s < S; ++s) {
        const int start = d.cross_start[s];
        length[s] = d.cross_length[s];
        first_word[s] = start >> 2;
        byte_shift[s] = static_cast<uint32_t>(start & 3);
        unique[s] = d.cross_unique[s];
        radix[s] = d.cross_stride[s];
        full_mask[s] = 0;
#pragma unroll
        for(int j = 0; j < 4 * G; ++j) { if(j < length[s]) { full_mask[s] |= 1u << low_bit(j); } }
#pragma unroll
        for(int g = 0; g < G; ++g) {
            const int valid = length[s] - 4 * g;
            byte_mask[s][g] = valid >= 4 ? 0xffffffffu : valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u);
        }
    }
    const int64_t stride = d.plane_stride;
    const char* const factor_bytes = reinterpret_cast<const char*>(&factor_double[0][0]);
    const char* const cost_bytes = reinterpret_cast<const char*>(&cost_double[0][0]);
    const uint32_t* const words0 = word_copy[0];
    const uint32_t* const words1 = word_copy[1];
    const PHQ_CONSTANT int32_t* const ordinal = constant_cast(d.cross_ordinal);

    for(int64_t tile = 0; tile < tiles; ++tile) {
        const int64_t k = tile * WEIGHTED_THREADS + threadIdx.x;
        const bool active = k < n;
        const int64_t i = static_cast<int64_t>(segment[active ? k : n - 1]);
        __syncthreads();                       // every entry of this tile is read before the wave compacts into it
        ReadState r = load_read_state(args, i);
        const uint32_t* const code_row = reinterpret_cast<const uint32_t*>(args.base + i * stride);
        const uint32_t* const quality_row = reinterpret_cast<const uint32_t*>(args.quality + i * stride);

        uint32_t row_offset[S][4 * G];
        uint32_t observed[S];
        uint32_t outside[S];
        uint32_t high_quality[S];
#pragma unroll
        for(int s = 0; s < S; ++s) {
            uint32_t c[G + 1];
            uint32_t q[G + 1];
#pragma unroll
            for(int g = 0; g <= G; ++g) {
                const int w = min(first_word[s] + g, row_words - 1);
                c[g] = code_row[w];
                q[g] = quality_row[w];
            }
            observed[s] = 0; outside[s] = 0; high_quality[s] = 0;
#pragma unroll
            for(int j = 0; j < 4 * G; ++j) {
                const uint32_t code_word = __builtin_amdgcn_alignbyte(c[(j >> 2) + 1], c[j >> 2], byte_shift[s]) & byte_mask[s][j >> 2];
                const uint32_t quality_word = __builtin_amdgcn_alignbyte(q[(j >> 2) + 1], q[j >> 2], byte_shift[s]) & byte_mask[s][j >> 2];
                const uint32_t code = (code_word >> (8 * (j & 3))) & 0xffu;
                const uint32_t q8 = (quality_word >> (8 * (j & 3))) & 0xffu;
                const uint32_t quality = q8 & 0x7fu;
                const uint32_t row = strict_code(code) ? quality : (quality ? 128u : 0u);
                row_offset[s][j] = row * 16u;
                const uint32_t two = (0x30210u >> (2 * (code & 0xfu))) & 3u;
                observed[s] |= ((two & 1u) << low_bit(j)) | ((two >> 1) << high_bit(j));
                if(static_cast<int>(q8) >= d.high_quality_threshold) { high_quality[s] |= 1u << low_bit(j); }
                if(!strict_code(code)) { outside[s] |= 1u << low_bit(j); }
            }
        }
        auto flags_of = [&](auto which, uint32_t word) {
            constexpr int s = decltype(which)::value;
            return (mismatch_fields(observed[s] ^ word) | outside[s]) & full_mask[s];
        };
        auto product_of = [&](auto which, uint32_t flags) {
            constexpr int s = decltype(which)::value;
            double p = 1.0;
#pragma unroll
            for(int j = 0; j < 4 * G; ++j) {
                p *= *reinterpret_cast<const double*>(factor_bytes + (row_offset[s][j] + ((flags >> low_bit(j)) & 1u) * 8u));
            }
            return p;
        };
        const std::integral_constant<int, 0> zero;
        const std::integral_constant<int, 1> one;

        // both segments: sum and maximum
        double sum1 = 0;
        double top1 = 0;
        int first1 = 0;
        for(int u1 = 0; u1 < unique[1]; ++u1) {
            const double p = product_of(one, flags_of(one, words1[u1]));
            sum1 += p;
            if(p > top1) { top1 = p; first1 = u1; }
        }
        double top0 = 0;
        double sum0 = 0;
        int first0 = 0;
        for(int u0 = 0; u0 < unique[0]; ++u0) {
            const double p = product_of(zero, flags_of(zero, words0[u0]));
            sum0 += p;
            if(p > top0) { top0 = p; first0 = u0; }
        }
        // A read no barcode of which can pass the noise filter (every P(r|b) <= top0 * top1 < the random barcode
        // probability, the literal values being within 1e-13 of these products) reports nothing but its argmax: then
        // only words that can still beat the pair of the two maxima are listed — prior P0 P1 >= prior_tt top0 top1
        // needs P0 >= top0 * prior_tt / largest prior, and the same for P1 — and no sum is taken.  Otherwise: every
        // word within 1e-15 of its segment's maximum, the only ones that can move the sum by 1e-12 or be the argmax.
        const double prior_of_tops = d.cross_prior_double[first0 * radix[0] + first1 * radix[1]];
        const bool only_argmax = top0 * top1 < d.random_barcode_probability * (1.0 - 1e-9) && prior_of_tops > 0;
        const double keep = only_argmax ? (prior_of_tops / d.cross_prior_max) * (1.0 - 1e-9) : 1e-15;
        bool defer = !(top0 > 0) || !(top1 > 0);
        double listed0 = 0;
        int count0 = 0;
        for(int u0 = 0; u0 < unique[0]; ++u0) {
            const double p = product_of(zero, flags_of(zero, words0[u0]));
            if(p >= top0 * keep && p > 0) {
                if(count0 < WEIGHTED_LIST) {
                    significant[count0 * WEIGHTED_THREADS + threadIdx.x] = static_cast<uint8_t>(u0);
                    ++count0;
                    listed0 += p;
                } else {
                    defer = true;
                }
            }
        }
        double listed1 = 0;
        int count1 = 0;
        for(int u1 = 0; u1 < unique[1]; ++u1) {
            const double p = product_of(one, flags_of(one, words1[u1]));
            if(p >= top1 * keep && p > 0) {
                if(count1 < WEIGHTED_SHORT) {
                    listed_value[count1 * WEIGHTED_THREADS + threadIdx.x] = p;
                    listed_word[count1 * WEIGHTED_THREADS + threadIdx.x] = static_cast<uint8_t>(u1);
                    ++count1;
                    listed1 += p;
                } else {
                    defer = true;
                }
            }
        }
        // the bilinear form over the listed words, and the largest prior * P0 * P1
        if(count0 * count1 > WEIGHTED_PAIRS) { defer = true; count0 = 0; }
        double weighted_sum = 0;
        double weighted_top = 0;
        for(int k0 = 0; k0 < count0; ++k0) {
            const int u0 = significant[k0 * WEIGHTED_THREADS + threadIdx.x];
            const double p0 = product_of(zero, flags_of(zero, words0[u0]));
            const double* const prior_row = d.cross_prior_double + u0 * radix[0];
            double row_sum = 0;
            double row_top = 0;
#pragma unroll 4
            for(int k1 = 0; k1 < count1; ++k1) {
                const int u1 = listed_word[k1 * WEIGHTED_THREADS + threadIdx.x];
                const double p = prior_row[u1 * radix[1]] * listed_value[k1 * WEIGHTED_THREADS + threadIdx.x];
                row_sum += p;
                row_top = fmax(row_top, p);
            }
            weighted_sum += p0 * row_sum;
            weighted_top = fmax(weighted_top, p0 * row_top);
        }
        // what the lists leave out, at most: the unlisted mass at the largest prior
        const double left_out = d.cross_prior_max * (fmax(sum0 * sum1 - listed0 * listed1, 0.0) + 1e-15 * sum0 * sum1);
        if(!(weighted_top > 0)) { defer = true; }
        if(!only_argmax && !(left_out <= 1e-12 * (weighted_sum + d.adjusted_noise_probability))) { defer = true; }

        // the combinations within 1e-10 of the largest, literally (barcode.h:131-164), in barcode order
        int best = 0x7fffffff;
        double best_conditional = -1.0;
        double best_p = -1.0;
        uint32_t best_flags[S] = {0, 0};
        if(!defer) {
            const double floor = weighted_top * (1.0 - 1e-10);
            for(int k0 = 0; k0 < count0; ++k0) {
                const int u0 = significant[k0 * WEIGHTED_THREADS + threadIdx.x];
                const uint32_t flags0 = flags_of(zero, words0[u0]);
                const double p0 = product_of(zero, flags0);
                const double* const prior_row = d.cross_prior_double + u0 * radix[0];
                for(int k1 = 0; k1 < count1; ++k1) {
                    const int u1 = listed_word[k1 * WEIGHTED_THREADS + threadIdx.x];
                    const double prior = prior_row[u1 * radix[1]];
                    if(!(prior * listed_value[k1 * WEIGHTED_THREADS + threadIdx.x] * p0 >= floor)) { continue; }
                    const int b = ordinal[u0 * radix[0] + u1 * radix[1]];
                    const uint32_t flags1 = flags_of(one, words1[u1]);
                    double sigma = 0;
                    double compensation = 0;
#pragma unroll
                    for(int s = 0; s < S; ++s) {
                        const uint32_t flags = s == 0 ? flags0 : flags1;
#pragma unroll
                        for(int j = 0; j < 4 * G; ++j) {
                            if(j < length[s]) {
                                const double x = *reinterpret_cast<const double*>(cost_bytes + (row_offset[s][j] + ((flags >> low_bit(j)) & 1u) * 8u));
                                const double y = x - compensation;
                                const double t = sigma + y;
                                compensation = (t - sigma) - y;
                                sigma = t;
                            }
                        }
                    }
                    const double conditional = phred_probability(sigma, d.log2_base_hi, d.log2_base_lo);
                    const double p = conditional * prior;                      // pamld.cpp:66
                    if(p > best_p || (p == best_p && b < best)) {              // strict '>' in barcode order
                        best = b; best_p = p; best_conditional = conditional;
                        best_flags[0] = flags0; best_flags[1] = flags1;
                    }
                }
            }
        }
        if(best == 0x7fffffff) { defer = true; best = 0; best_conditional = 0; best_p = 0; }

        int distance = __popc(best_flags[0]) + __popc(best_flags[1]);
        int hq_distance = __popc(best_flags[0] & high_quality[0]) + __popc(best_flags[1] & high_quality[1]);
        const double total = weighted_sum + d.adjusted_noise_probability;            // pamld.cpp:61-86
        double confidence = best_p / total;                                          // pamld.cpp:92
        const bool is_noise = !(best_conditional > d.random_barcode_probability);    // pamld.cpp:96, on the literal value
        if(!is_noise && !(fabs(confidence - d.confidence_threshold) > 1e-12)) { defer = true; }
        if(!defer && active) {
            int bin = best + 1;
            uint8_t status = STATUS_PASS;
            if(!is_noise) {
                if(confidence > d.confidence_threshold) {                            // pamld.cpp:99
                    if(d.high_quality_distance_threshold > 0 && hq_distance >= d.high_quality_distance_threshold) { r.qcfail = true; }
                    stat_add_confidence(d, sink, bin, r.qcfail ? 0 : 1, confidence);
                } else {
                    stat_add(d, sink, bin, 0, FIELD_LOW_CONFIDENCE, 1ull);
                    r.qcfail = true;
                    status = STATUS_LOW_CONFIDENCE;
                }
            } else {
                stat_add(d, sink, bin, 0, FIELD_LOW_CONDITIONAL_CONFIDENCE, 1ull);
                r.qcfail = true;
                bin = 0; distance = 0; hq_distance = 0; confidence = 0;
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

// What the kernel above leaves — reads with several low quality calls, more than WEIGHTED_SHORT words of segment 1 within
// 1e-15 of its maximum — takes the WHOLE bilinear form, one wave per read: the lanes share out the words (P of every word
// of both segments to LDS), then the rows of the prior matrix; nothing is left out, so nothing has to be bounded.  The
// argmax as above: the combinations within 1e-10 of the largest prior * P0 * P1 literally, in barcode order.  A pass of
// the literal kernel over a 96 x 96 codec takes a wave ~5 ms however few reads it holds; this takes ~10 us per read.
constexpr int DENSE_THREADS = 256;
constexpr int DENSE_WAVES = DENSE_THREADS / 64;   // a segment's reads go round the waves

// The walk reads the whole prior matrix from L2 for every read (72 KiB for 96 x 96: ~5 GB a step of 50 M C3 reads).  That is
// not what bounds the kernel: the matrix through LDS in tiles shared by the workgroup's four reads took the same 0.8 ms (the
// whole matrix in LDS, one workgroup a CU: 0.99 ms; four waves a SIMD by register limit, spills: 1.14 ms).  A read takes a
// wave ~30 us of dependent steps and two waves fit a SIMD: 51 K reads x 30 us over 2 048 waves in flight is the 0.75 ms.
template <int G>
__global__ __launch_bounds__(DENSE_THREADS) void pamld_cross_exact_dense_kernel(StageArgs args) {
    constexpr int S = 2;
    __shared__ double value_of_wave[DENSE_WAVES][S][CROSS_MAX_UNIQUE];        // P_s of every word, of the read a wave has in hand
    __shared__ uint32_t word_copy[S][CROSS_MAX_UNIQUE];
    __shared__ uint32_t deferred_total;
    __shared__ double factor_double[FACTOR_ROWS][2];
    __shared__ double cost_double[FACTOR_ROWS][2];
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw);
    DecoderRef d = *constant_cast(args.decoder);
    for(int i = threadIdx.x; i < FACTOR_ROWS; i += DENSE_THREADS) {
        factor_double[i][0] = i < 128 ? d.match_probability[i] : d.uniform_probability;
        factor_double[i][1] = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        cost_double[i][0] = i < 128 ? d.true_positive_quality[i] : d.uniform_base_quality;
        cost_double[i][1] = i < 128 ? static_cast<double>(i) : d.uniform_base_quality;
    }
    for(int i = threadIdx.x; i < S * CROSS_MAX_UNIQUE; i += DENSE_THREADS) { word_copy[i / CROSS_MAX_UNIQUE][i % CROSS_MAX_UNIQUE] = d.cross_words[i]; }
    if(threadIdx.x == 0) { deferred_total = 0; }
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const int64_t n = static_cast<int64_t>(args.worklist_count[blockIdx.x]);
    int first_word[S];
    int length[S];
    uint32_t byte_shift[S];
    uint32_t byte_mask[S][G];
    uint32_t full_mask[S];
    int unique[S];
    int radix[S];
    const int row_words = d.plane_stride >> 2;
#pragma unroll
    for(int s = 0; s < S; ++s) {
        const int start = d.cross_start[s];
        length[s] = d.cross_length[s];
        first_word[s] = start >> 2;
        byte_shift[s] = static_cast<uint32_t>(start & 3);
        unique[s] = d.cross_unique[s];
        radix[s] = d.cross_stride[s];
        full_mask[s] = 0;
#pragma unroll
        for(int j = 0; j < 4 * G; ++j) { if(j < length[s]) { full_mask[s] |= 1u << low_bit(j); } }
#pragma unroll
        for(int g = 0; g < G; ++g) {
            const int valid = length[s] - 4 * g;
            byte_mask[s][g] = valid >= 4 ? 0xffffffffu : valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u);
        }
    }
    const int64_t stride = d.plane_stride;
    const char* const factor_bytes = reinterpret_cast<const char*>(&factor_double[0][0]);
    const char* const cost_bytes = reinterpret_cast<const char*>(&cost_double[0][0]);
    const uint32_t* const words_of[S] = {word_copy[0], word_copy[1]};
    const PHQ_CONSTANT int32_t* const ordinal = constant_cast(d.cross_ordinal);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double (* const value)[CROSS_MAX_UNIQUE] = value_of_wave[wave];

    for(int64_t round = 0; round * DENSE_WAVES < n; ++round) {
        const int64_t k = round * DENSE_WAVES + wave;
        const bool active = k < n;                                             // a wave past the end takes the last read again, silently
        const int64_t i = static_cast<int64_t>(segment[active ? k : n - 1]);   // one read, the whole wave
        __syncthreads();                       // every entry of this round is read before any wave compacts into it
        ReadState r = load_read_state(args, i);
        const uint32_t* const code_row = reinterpret_cast<const uint32_t*>(args.base + i * stride);
        const uint32_t* const quality_row = reinterpret_cast<const uint32_t*>(args.quality + i * stride);
        uint32_t row_offset[S][4 * G];
        uint32_t observed[S];
        uint32_t outside[S];
        uint32_t high_quality[S];
#pragma unroll
        for(int s = 0; s < S; ++s) {
            uint32_t c[G + 1];
            uint32_t q[G + 1];
#pragma unroll
            for(int g = 0; g <= G; ++g) {
                const int w = min(first_word[s] + g, row_words - 1);
                c[g] = code_row[w];
                q[g] = quality_row[w];
            }
            observed[s] = 0; outside[s] = 0; high_quality[s] = 0;
#pragma unroll
            for(int j = 0; j < 4 * G; ++j) {
                const uint32_t code_word = __builtin_amdgcn_alignbyte(c[(j >> 2) + 1], c[j >> 2], byte_shift[s]) & byte_mask[s][j >> 2];
                const uint32_t quality_word = __builtin_amdgcn_alignbyte(q[(j >> 2) + 1], q[j >> 2], byte_shift[s]) & byte_mask[s][j >> 2];
                const uint32_t code = (code_word >> (8 * (j & 3))) & 0xffu;
                const uint32_t q8 = (quality_word >> (8 * (j & 3))) & 0xffu;
                const uint32_t quality = q8 & 0x7fu;
                const uint32_t row = strict_code(code) ? quality : (quality ? 128u : 0u);
                row_offset[s][j] = row * 16u;
                const uint32_t two = (0x30210u >> (2 * (code & 0xfu))) & 3u;
                observed[s] |= ((two & 1u) << low_bit(j)) | ((two >> 1) << high_bit(j));
                if(static_cast<int>(q8) >= d.high_quality_threshold) { high_quality[s] |= 1u << low_bit(j); }
                if(!strict_code(code)) { outside[s] |= 1u << low_bit(j); }
            }
        }
        auto flags_of = [&](int s, uint32_t word) { return (mismatch_fields(observed[s] ^ word) | outside[s]) & full_mask[s]; };
        auto product_of = [&](int s, uint32_t flags) {
            double p = 1.0;
#pragma unroll
            for(int j = 0; j < 4 * G; ++j) {
                p *= *reinterpret_cast<const double*>(factor_bytes + (row_offset[s][j] + ((flags >> low_bit(j)) & 1u) * 8u));
            }
            return p;
        };

        // P of every word of both segments; per segment the maximum (the first word that holds it)
        double top[S];
        int first[S];
#pragma unroll
        for(int s = 0; s < S; ++s) {
            double lane_top = 0;
            int lane_first = 0x7fffffff;
            for(int u = lane; u < unique[s]; u += 64) {
                const double p = product_of(s, flags_of(s, words_of[s][u]));
                value[s][u] = p;
                if(p > lane_top) { lane_top = p; lane_first = u; }
            }
#pragma unroll
            for(int offset = 32; offset > 0; offset >>= 1) {
                const double other_top = __shfl_xor(lane_top, offset);
                const int other_first = __shfl_xor(lane_first, offset);
                if(other_top > lane_top || (other_top == lane_top && other_first < lane_first)) { lane_top = other_top; lane_first = other_first; }
            }
            top[s] = lane_top;
            first[s] = lane_first;
        }
        __syncthreads();
        bool defer = !(top[0] > 0) || !(top[1] > 0);

        // the bilinear form, a lane per word of segment 1 — a column of the prior matrix, whose rows the lanes then read side by
        // side — and the largest prior * P0 * P1
        double weighted_sum = 0;
        double weighted_top = 0;
        double top_of_column[CROSS_MAX_UNIQUE / 64];           // P1 * the largest prior * P0 of this lane's columns
        {
            // both of a lane's columns in one walk down the rows: twice the loads in flight, half the trips
            static_assert(CROSS_MAX_UNIQUE == 128, "two columns per lane");
            const int u1_low = min(lane, unique[1] - 1);                      // a lane past the end walks the last column, unheard
            const int u1_high = min(lane + 64, unique[1] - 1);
            const double* const column_low = d.cross_prior_double + u1_low * radix[1];
            const double* const column_high = d.cross_prior_double + u1_high * radix[1];
            double sum_low = 0, sum_high = 0, top_low = 0, top_high = 0;
#pragma unroll 16
            for(int u0 = 0; u0 < unique[0]; ++u0) {
                const double p0 = value[0][u0];
                const double low = column_low[u0 * radix[0]] * p0;
                const double high = column_high[u0 * radix[0]] * p0;
                sum_low += low;
                sum_high += high;
                top_low = fmax(top_low, low);
                top_high = fmax(top_high, high);
            }
            const bool has_low = lane < unique[1], has_high = lane + 64 < unique[1];
            top_of_column[0] = has_low ? value[1][u1_low] * top_low : 0.0;
            top_of_column[1] = has_high ? value[1][u1_high] * top_high : 0.0;
            weighted_sum = (has_low ? value[1][u1_low] * sum_low : 0.0) + (has_high ? value[1][u1_high] * sum_high : 0.0);
            weighted_top = fmax(top_of_column[0], top_of_column[1]);
        }
#pragma unroll
        for(int offset = 32; offset > 0; offset >>= 1) {
            weighted_sum += __shfl_xor(weighted_sum, offset);
            weighted_top = fmax(weighted_top, __shfl_xor(weighted_top, offset));
        }
        if(!(weighted_top > 0)) { defer = true; }

        // the combinations within 1e-10 of the largest, literally (barcode.h:131-164); the first in barcode order wins
        int best = 0x7fffffff;
        double best_conditional = -1.0;
        double best_p = -1.0;
        uint32_t best_flags[S] = {0, 0};
        if(!defer) {
            const double floor = weighted_top * (1.0 - 1e-10);
            // only a column whose largest product comes that close holds any: those — one or two — are walked again, the lanes
            // now sharing out the words of segment 0
#pragma unroll
            for(int half = 0; half < CROSS_MAX_UNIQUE / 64; ++half) {
              unsigned long long columns = __ballot(top_of_column[half] >= floor);
              while(columns) {
                const int u1 = 64 * half + __ffsll(static_cast<long long>(columns)) - 1;
                columns &= columns - 1;
                const uint32_t flags1 = flags_of(1, words_of[1][u1]);
                const double p1 = value[1][u1];
                for(int u0 = lane; u0 < unique[0]; u0 += 64) {
                    const double prior = d.cross_prior_double[u0 * radix[0] + u1 * radix[1]];
                    if(!(prior * value[0][u0] * p1 >= floor)) { continue; }
                    {
                    const int b = ordinal[u0 * radix[0] + u1 * radix[1]];
                    const uint32_t flags0 = flags_of(0, words_of[0][u0]);
                    double sigma = 0;
                    double compensation = 0;
#pragma unroll
                    for(int s = 0; s < S; ++s) {
                        const uint32_t flags = s == 0 ? flags0 : flags1;
#pragma unroll
                        for(int j = 0; j < 4 * G; ++j) {
                            if(j < length[s]) {
                                const double x = *reinterpret_cast<const double*>(cost_bytes + (row_offset[s][j] + ((flags >> low_bit(j)) & 1u) * 8u));
                                const double y = x - compensation;
                                const double t = sigma + y;
                                compensation = (t - sigma) - y;
                                sigma = t;
                            }
                        }
                    }
                    const double conditional = phred_probability(sigma, d.log2_base_hi, d.log2_base_lo);
                    const double p = conditional * prior;                      // pamld.cpp:66
                    if(p > best_p || (p == best_p && b < best)) {              // strict '>' in barcode order
                        best = b; best_p = p; best_conditional = conditional;
                        best_flags[0] = flags0; best_flags[1] = flags1;
                    }
                    }
                }
              }
            }
            // across the lanes: the largest, the lowest barcode among equals; its lane hands over the rest
            double wave_p = best_p;
            int wave_best = best;
#pragma unroll
            for(int offset = 32; offset > 0; offset >>= 1) {
                const double other_p = __shfl_xor(wave_p, offset);
                const int other_best = __shfl_xor(wave_best, offset);
                if(other_p > wave_p || (other_p == wave_p && other_best < wave_best)) { wave_p = other_p; wave_best = other_best; }
            }
            const unsigned long long holders = __ballot(best == wave_best && wave_best != 0x7fffffff);
            if(holders) {
                const int holder = __ffsll(static_cast<long long>(holders)) - 1;
                best = wave_best;
                best_p = wave_p;
                best_conditional = __shfl(best_conditional, holder);
                best_flags[0] = __shfl(best_flags[0], holder);
                best_flags[1] = __shfl(best_flags[1], holder);
            } else {
                best = 0x7fffffff;
            }
        }
        if(best == 0x7fffffff) { defer = true; best = 0; best_conditional = 0; best_p = 0; }
        __syncthreads();                       // every lane is done with this read's P before the next one's are written

        int distance = __popc(best_flags[0]) + __popc(best_flags[1]);
        int hq_distance = __popc(best_flags[0] & high_quality[0]) + __popc(best_flags[1] & high_quality[1]);
        const double total = weighted_sum + d.adjusted_noise_probability;            // pamld.cpp:61-86
        double confidence = best_p / total;                                          // pamld.cpp:92
        const bool is_noise = !(best_conditional > d.random_barcode_probability);    // pamld.cpp:96, on the literal value
        if(!is_noise && !(fabs(confidence - d.confidence_threshold) > 1e-12)) { defer = true; }
        const bool writer = lane == 0 && active;
        if(!defer && writer) {
            int bin = best + 1;
            uint8_t status = STATUS_PASS;
            if(!is_noise) {
                if(confidence > d.confidence_threshold) {                            // pamld.cpp:99
                    if(d.high_quality_distance_threshold > 0 && hq_distance >= d.high_quality_distance_threshold) { r.qcfail = true; }
                    stat_add_confidence(d, sink, bin, r.qcfail ? 0 : 1, confidence);
                } else {
                    stat_add(d, sink, bin, 0, FIELD_LOW_CONFIDENCE, 1ull);
                    r.qcfail = true;
                    status = STATUS_LOW_CONFIDENCE;
                }
            } else {
                stat_add(d, sink, bin, 0, FIELD_LOW_CONDITIONAL_CONFIDENCE, 1ull);
                r.qcfail = true;
                bin = 0; distance = 0; hq_distance = 0; confidence = 0;
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
        if(defer && writer) { segment[atomicAdd(&deferred_total, 1u)] = static_cast<uint32_t>(i); }      // a place behind this round's entries
        count_outgoing(args, histogram, writer && !defer, r);
    }
    __syncthreads();
    if(threadIdx.x == 0) { args.worklist_count[blockIdx.x] = deferred_total; }
    histogram_flush(args, d, histogram);
}

inline int launch_exact_dense(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.cross_groups) {
        case 1: hipLaunchKernelGGL((pamld_cross_exact_dense_kernel<1>), grid, dim3(DENSE_THREADS), histogram_bytes, stream, args); break;
        case 2: hipLaunchKernelGGL((pamld_cross_exact_dense_kernel<2>), grid, dim3(DENSE_THREADS), histogram_bytes, stream, args); break;
        case 3: hipLaunchKernelGGL((pamld_cross_exact_dense_kernel<3>), grid, dim3(DENSE_THREADS), histogram_bytes, stream, args); break;
        case 4: hipLaunchKernelGGL((pamld_cross_exact_dense_kernel<4>), grid, dim3(DENSE_THREADS), histogram_bytes, stream, args); break;
        default: return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}

inline int launch_exact_weighted(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.cross_groups) {
        case 1: hipLaunchKernelGGL((pamld_cross_exact_weighted_kernel<1>), grid, dim3(WEIGHTED_THREADS), histogram_bytes, stream, args); break;
        case 2: hipLaunchKernelGGL((pamld_cross_exact_weighted_kernel<2>), grid, dim3(WEIGHTED_THREADS), histogram_bytes, stream, args); break;
        case 3: hipLaunchKernelGGL((pamld_cross_exact_weighted_kernel<3>), grid, dim3(WEIGHTED_THREADS), histogram_bytes, stream, args); break;
        case 4: hipLaunchKernelGGL((pamld_cross_exact_weighted_kernel<4>), grid, dim3(WEIGHTED_THREADS), histogram_bytes, stream, args); break;
        default: return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}

// a single segment set: more than CROSS_MAX_UNIQUE words and / or priors that differ
template <int G, int S, int UNIQUE, bool PRIORS>
int launch_exact_form(const StageArgs& args, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    static const size_t pad = std::getenv("PHQ_EXACT_PAD_LDS") ? static_cast<size_t>(std::atoi(std::getenv("PHQ_EXACT_PAD_LDS"))) : 0;      // occupancy experiments
    const size_t bytes = ((histogram_bytes + 15u) & ~static_cast<size_t>(15u)) + exact_table_bytes(G) + pad;
    if(bytes > 64u * 1024u) {
        static std::atomic<uint64_t> prepared{0};
        // at most 32 KiB of histogram + 64 KiB of tables; the kernel also has ~5 KiB of static LDS
        if(const int status = allow_large_lds(&pamld_cross_exact_kernel<G, S, UNIQUE, PRIORS>, prepared, 128 * 1024)) { return status; }
    }
    hipLaunchKernelGGL((pamld_cross_exact_kernel<G, S, UNIQUE, PRIORS>), grid, dim3(EXACT_TABLE_LANES), bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}

template <int UNIQUE, bool PRIORS>
int launch_exact_single(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.cross_groups) {
        case 1: return launch_exact_form<1, 1, UNIQUE, PRIORS>(args, grid, histogram_bytes, stream);
        case 2: return launch_exact_form<2, 1, UNIQUE, PRIORS>(args, grid, histogram_bytes, stream);
        case 3: return launch_exact_form<3, 1, UNIQUE, PRIORS>(args, grid, histogram_bytes, stream);
        case 4: return launch_exact_form<4, 1, UNIQUE, PRIORS>(args, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

template <int S>
int launch_exact_segments(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.cross_groups) {
        case 1: return launch_exact_form<1, S, CROSS_MAX_UNIQUE, false>(args, grid, histogram_bytes, stream);
        case 2: return launch_exact_form<2, S, CROSS_MAX_UNIQUE, false>(args, grid, histogram_bytes, stream);
        case 3: return launch_exact_form<3, S, CROSS_MAX_UNIQUE, false>(args, grid, histogram_bytes, stream);
        case 4: return launch_exact_form<4, S, CROSS_MAX_UNIQUE, false>(args, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

template <int G, int S, bool WEIGHTED>
int launch_scan(const StageArgs& args, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    static std::atomic<uint64_t> prepared{0};
    if(const int status = allow_large_lds(&pamld_cross_kernel<G, S, WEIGHTED>, prepared)) { return status; }
    hipLaunchKernelGGL((pamld_cross_kernel<G, S, WEIGHTED>), grid, dim3(THREADS), scan_fixed_bytes(G, S) + histogram_bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}

template <int S, bool WEIGHTED>
int launch_segments(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.cross_groups) {
        case 1: return launch_scan<1, S, WEIGHTED>(args, grid, histogram_bytes, stream);
        case 2: return launch_scan<2, S, WEIGHTED>(args, grid, histogram_bytes, stream);
        case 3: return launch_scan<3, S, WEIGHTED>(args, grid, histogram_bytes, stream);
        case 4: return launch_scan<4, S, WEIGHTED>(args, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace cross
}  // namespace phq
