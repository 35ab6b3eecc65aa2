// K0g (plates): PamlDecoder<T>::classify for dual index sets made of FEW words per segment — the rows and columns of a
// plate: 12 x 8 i7 / i5 words of 8 nt, all 96 combinations or most of them (the 94 pairs of docs/illumina_api_vignette.md's
// flowcell H7LT2DSXX are a 12 x 8 plate less two wells) — two segments of 4 to 8 nt, at most 16 distinct words each.
//
//   reference pamld.cpp:37-123   posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164  P(r|b) = pow(base, sum of Phred costs), distance, high quality distance
//
// Sets like these fell between the tiers: not every combination (pamld_cross*.hpp), words shared between barcodes
// (pamld_pair.hpp wants them all different), too few words a segment for neighbour lists — they ran through the product form
// over all barcodes (8.2 ms per 100 M reads).  With 16 words a segment nothing has to be bounded: EVERY word of both
// segments is scored, per read, through the per read tables of pamld_stream.hpp:
//   * a lookup per segment (4^length entries of one byte, the word at most two mismatches away) names c = (c0, c1); a word's
//     block in LDS lists its plane words first and then every other word of its segment, so the winner is slot 0 and the
//     others slots 1..15, with no select;
//   * P(r|b) factorises over the segments.  In units of the winner, the barcodes that share c1 weigh
//     sum_w0 prior(w0, c1) / prior(c) * P0(w0) / P0(c0), those that share c0 likewise, and every other barcode at most
//     (largest prior / prior(c)) * x0 * x1 with x_s the plain sum of the segment's other words over its winner.  The prior
//     ratios come per combination (32 floats in LDS: a well that is not in the set has ratio 0, a slot past the segment's
//     words too), so uniform priors, priors an estimation pass left and missing wells are one code path;
//   * the certificate is pamld_near.hpp's: everything but the winner at most 5e-4 of it (single precision sums: 3e-10 in
//     the posterior), the part that is only bounded at most 9e-9, no decision within 2e-8 of its threshold; bases outside
//     ACGT, a second mismatch in a segment, a combination that is not in the set, noise: the read goes to this workgroup's
//     worklist segment for the product form kernel (pamld_lut.hpp as consumer), the exact tier and the literal kernel.
#include <atomic>
#include <cstdlib>

#include "kernel_common.hpp"
#include "pamld_near.hpp"

namespace phq {
namespace plate {

constexpr int WAVES = 12;                                      // one workgroup a CU: a wave's tables are 8 KiB, the set's blocks and ratios are shared
constexpr int THREADS = 64 * WAVES;
constexpr int FACTOR_ROWS = 129;
constexpr int SLOTS = 16;                                      // words of a segment, at most
constexpr int BLOCK_WORDS = 3 * SLOTS;                         // low plane words, high plane words, 1.0 / 0.0 for a slot that holds a word
constexpr int RATIO_WORDS = 2 * SLOTS;                         // per combination: the row's prior ratios, the column's
constexpr int INFO_WORDS = 2;                                  // per combination: counter bin (0: not in the set), barcode index
constexpr uint32_t TABLE_BYTES = WAVES * 8192u;                // [wave][group][entry][lane]: 256 bytes an entry, 4 KiB a group
constexpr uint32_t FACTOR_AT = TABLE_BYTES;
constexpr uint32_t DEFERRED_AT = FACTOR_AT + 1040u;
constexpr uint32_t BLOCK_AT = DEFERRED_AT + 16u;
__host__ __device__ constexpr uint32_t ratio_at(int n0, int n1) { return BLOCK_AT + static_cast<uint32_t>(n0 + n1) * BLOCK_WORDS * 4u; }
__host__ __device__ constexpr uint32_t info_at(int n0, int n1) { return ratio_at(n0, n1) + static_cast<uint32_t>(n0 * n1) * RATIO_WORDS * 4u; }
__host__ __device__ constexpr uint32_t histogram_at(int n0, int n1) { return (info_at(n0, n1) + static_cast<uint32_t>(n0 * n1) * INFO_WORDS * 4u + 15u) & ~15u; }

typedef uint32_t Quad __attribute__((ext_vector_type(4)));
typedef float Pair __attribute__((ext_vector_type(2)));
typedef float Four __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const float* LdsFloats;
typedef __attribute__((address_space(3))) const Quad* LdsQuads;
typedef __attribute__((address_space(3))) const Four* LdsFours;
typedef __attribute__((address_space(3))) const Pair* LdsPairs;
using near::and_or;

// G = words of the plane row (4 positions each); STATE as in pamld_near.hpp
template <int G, int STATE>
__global__ __launch_bounds__(THREADS) void pamld_plate_kernel(StageArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    float2* const factor_single = reinterpret_cast<float2*>(lds_raw + FACTOR_AT);           // [row] {P(match), rho = P(mismatch) / P(match)}
    DecoderRef d = *constant_cast(args.decoder);
    const int n0 = d.plate_unique[0], n1 = d.plate_unique[1];
    unsigned long long* const histogram = reinterpret_cast<unsigned long long*>(lds_raw + histogram_at(n0, n1));

    for(int i = threadIdx.x; i < FACTOR_ROWS; i += THREADS) {
        const double m = i < 128 ? d.match_probability[i] : d.uniform_probability;
        const double e = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        factor_single[i] = make_float2(static_cast<float>(m), static_cast<float>(e / m));
    }
    {
        uint32_t* const blocks = reinterpret_cast<uint32_t*>(lds_raw + BLOCK_AT);
        for(int w = threadIdx.x; w < (n0 + n1) * BLOCK_WORDS; w += THREADS) { blocks[w] = d.plate_block[w]; }
        uint32_t* const ratios = reinterpret_cast<uint32_t*>(lds_raw + ratio_at(n0, n1));
        for(int w = threadIdx.x; w < n0 * n1 * RATIO_WORDS; w += THREADS) { ratios[w] = d.plate_ratio[w]; }
        uint32_t* const info = reinterpret_cast<uint32_t*>(lds_raw + info_at(n0, n1));
        for(int w = threadIdx.x; w < n0 * n1 * INFO_WORDS; w += THREADS) { info[w] = d.plate_info[w]; }
    }
    histogram_clear(d, histogram);
    // entry 0 of both tables (no mismatch in the group) is 1 in units of the all-match product: written once
    const uint32_t lane_const = (threadIdx.x >> 6) * 8192u + (threadIdx.x & 63u) * 4u;
    *reinterpret_cast<float*>(lds_raw + lane_const) = 1.0f;
    *reinterpret_cast<float*>(lds_raw + lane_const + 4096u) = 1.0f;
    __syncthreads();

    // a worklist segment per WAVE (the tiers behind take a segment per wave or workgroup: 12 per CU keep them busy)
    const int64_t own_segment = static_cast<int64_t>(blockIdx.x) * WAVES + (threadIdx.x >> 6);
    uint32_t* const segment = args.worklist + own_segment * args.worklist_stride;
    uint32_t segment_count = 0;
    const uint8_t* const nearest[2] = {d.plate_table[0], d.plate_table[1]};
    const uint32_t start[2] = {static_cast<uint32_t>(d.plate_start[0]), static_cast<uint32_t>(d.plate_start[1])};
    const uint32_t length[2] = {static_cast<uint32_t>(d.plate_length[0]), static_cast<uint32_t>(d.plate_length[1])};
    const uint32_t ratios_at = ratio_at(n0, n1), infos_at = info_at(n0, n1);
    const int quads[2] = {(n0 + 3) / 4, (n1 + 3) / 4};              // slots of a block that hold words, in fours
    const uint32_t blocks_at[2] = {BLOCK_AT, BLOCK_AT + static_cast<uint32_t>(n0) * BLOCK_WORDS * 4u};
    const int L = d.nucleotide_cardinality;
    const int64_t n = args.n_read;
    const int64_t tiles = (n + THREADS - 1) / THREADS;
    const int high_quality_threshold = min(max(d.high_quality_threshold, 0), 128);
    const uint32_t high_quality_addend = static_cast<uint32_t>(128 - high_quality_threshold) * 0x01010101u;
    const float noise_floor = static_cast<float>(d.random_barcode_probability) * near::SLOP;
    const double confidence_threshold = d.confidence_threshold;
    uint32_t valid[G];
#pragma unroll
    for(int w = 0; w < G; ++w) { valid[w] = L - 4 * w >= 4 ? 0xffffffffu : (L - 4 * w <= 0 ? 0u : ((1u << (8 * (L - 4 * w))) - 1u)); }

    struct Inputs { uint32_t code[G]; uint32_t quality[G]; uint32_t state; };
    auto fetch = [&](int64_t tile, Inputs& in) {
        const int64_t first = tile * THREADS;
        const uint32_t last = static_cast<uint32_t>(min(static_cast<int64_t>(THREADS), n - first)) - 1u;
        const uint32_t at = min(threadIdx.x, last);
        const uint32_t* const code_row = reinterpret_cast<const uint32_t*>(args.base + first * (4 * G) + at * (4u * G));
        const uint32_t* const quality_row = reinterpret_cast<const uint32_t*>(args.quality + first * (4 * G) + at * (4u * G));
#pragma unroll
        for(int w = 0; w < G; ++w) { in.code[w] = code_row[w]; in.quality[w] = quality_row[w]; }
        if(STATE == 1) { in.state = static_cast<uint32_t>(args.input_flag[first + at]); }      // transcode.h:207-211
        else if(STATE == 2) { in.state = args.word[first + at]; }
        else { in.state = 0; }
    };
    Inputs next;
    fetch(min(static_cast<int64_t>(blockIdx.x), tiles - 1), next);
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0) before the loop (pamld_near.hpp)
    uint32_t pending_slot = 0xffffffffu;
    uint32_t pending_read = 0;
    const int hq_threshold = d.high_quality_distance_threshold;
    const bool multiplexing = d.multiplexing != 0;

    for(int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const Inputs in = next;
        const int64_t first = tile * THREADS;
        const bool active = first + threadIdx.x < n;
        const uint32_t at = active ? threadIdx.x : static_cast<uint32_t>(n - 1 - first);
        const int64_t i = first + at;
        ReadState r;
        if(STATE != 2) { r.qcfail = (in.state & 1u) != 0; r.channel = 0; }
        else { r.qcfail = (in.state >> 31) != 0; r.channel = static_cast<int32_t>(in.state & 0x7fffffffu); }

        auto gather = [](uint32_t planes) {            // bits 8 j and 8 j + 4 -> bits j and 4 + j
            const uint32_t t = planes | (planes >> 7);
            return (t | (t >> 14)) & 0xffu;
        };
        uint32_t q7[G];
        uint32_t low = 0, high = 0, high_quality = 0;    // bit j = position j of the row
        uint32_t outside = 0;                            // bit j: the base is none of A, C, G, T
        uint32_t not_strict = 0;
#pragma unroll
        for(int w = 0; w < G; ++w) {
            const uint32_t x = in.code[w] & (0x0f0f0f0fu & valid[w]);
            const uint32_t quality7 = in.quality[w] & (0x7f7f7f7fu & valid[w]);
            const uint32_t z = x & ((x | 0x80808080u) - 0x01010101u);
            const uint32_t strict = (x + 0x7f7f7f7fu) & ~(z + 0x7f7f7f7fu) & 0x80808080u;
            not_strict |= in.quality[w] & 0x80808080u & valid[w];      // a quality past 127
            // Round 5: a base outside ACGT stays in the kernel (pamld_pair.hpp): its factor row is 128 — match and mismatch weigh the
            // same against every word, so it moves no word against another — or 0 at quality 0; it counts as a mismatch of the winner
            const uint32_t strict_bytes = (strict - (strict >> 7)) | strict;
            const uint32_t nonzero = (quality7 + 0x7f7f7f7fu) & 0x80808080u;
            q7[w] = (quality7 & strict_bytes) | (nonzero & ~strict_bytes & valid[w]);               // the factor ROW of each position
            outside |= (gather((~strict & 0x80808080u & valid[w]) >> 7) & 0xfu) << (4 * w);
            const uint32_t both = gather((((x >> 1) | (x >> 3)) & 0x01010101u) | (((x << 2) | (x << 1)) & 0x10101010u));
            low |= (both & 0xfu) << (4 * w);
            high |= (both >> 4) << (4 * w);
            const uint32_t is_high = ((quality7 + high_quality_addend) | in.quality[w]) & (0x80808080u & valid[w]);
            high_quality |= (gather(is_high >> 7) & 0xfu) << (4 * w);
        }
        const bool all_strict = not_strict == 0;
        uint32_t segment_low[2], segment_high[2], named[2];
#pragma unroll
        for(int s = 0; s < 2; ++s) {
            const uint32_t mask = (1u << length[s]) - 1u;
            segment_low[s] = (low >> start[s]) & mask;
            segment_high[s] = (high >> start[s]) & mask;
            named[s] = static_cast<uint32_t>(nearest[s][segment_low[s] | (segment_high[s] << length[s])]);
        }
        if(pending_slot != 0xffffffffu) { segment[pending_slot] = pending_read; }      // the previous tile's deferred reads
        fetch(min(tile + gridDim.x, tiles - 1), next);

        const bool known = named[0] < static_cast<uint32_t>(n0) && named[1] < static_cast<uint32_t>(n1);
        const uint32_t c0 = known ? named[0] : 0u, c1 = known ? named[1] : 0u;
        const uint32_t combination = c0 * static_cast<uint32_t>(n1) + c1;
        Four ratio[2][SLOTS / 4];              // [0]: the row (segment 1's slots), [1]: the column (segment 0's slots)
#pragma unroll
        for(int quad = 0; quad < SLOTS / 4; ++quad) {
            // (round 5: only the quads that hold words — the row runs over segment 1's slots, the column over segment 0's; the same for
            // every lane — 5 quads instead of 8 for a 12 x 8 plate: the kernel 2.75 -> 2.59 ms; the `holds` quads read the same way: no change)
            if(quad == 0 || quad < quads[1]) { ratio[0][quad] = *(LdsFours)static_cast<uintptr_t>(ratios_at + combination * (RATIO_WORDS * 4u) + 16u * quad); }
            if(quad == 0 || quad < quads[0]) { ratio[1][quad] = *(LdsFours)static_cast<uintptr_t>(ratios_at + combination * (RATIO_WORDS * 4u) + 64u + 16u * quad); }
        }
        float winner_of[2], plain_of[2], weighted_of[2], matches_of[2];
        uint32_t mismatches_of[2];
#pragma unroll
        for(int s = 0; s < 2; ++s) {
            // the segment's qualities, byte j = its position j: its rho row and all-match product
            float rho[8];
            float everything_matches = 1.0f;
            const uint32_t first_word = start[s] >> 2;
            const uint32_t byte_shift = start[s] & 3u;
            uint32_t quality_word[2];
#pragma unroll
            for(int w = 0; w < 2; ++w) {
                uint32_t a = 0, b = 0;
#pragma unroll
                for(int k = 0; k < G; ++k) { a = (first_word + w == static_cast<uint32_t>(k)) ? q7[k] : a; b = (first_word + w + 1 == static_cast<uint32_t>(k)) ? q7[k] : b; }
                const uint32_t shifted = byte_shift == 0 ? a : byte_shift == 1 ? __builtin_amdgcn_alignbyte(b, a, 1) : byte_shift == 2 ? __builtin_amdgcn_alignbyte(b, a, 2) : __builtin_amdgcn_alignbyte(b, a, 3);
                const uint32_t left = length[s] > 4u * w ? length[s] - 4u * w : 0u;
                quality_word[w] = shifted & (left >= 4 ? 0xffffffffu : ((1u << (8 * left)) - 1u));
            }
#pragma unroll
            for(int j = 0; j < 8; ++j) {
                const uint32_t word = quality_word[j >> 2];
                const int byte = j & 3;
                const uint32_t shifted = byte == 0 ? (word << 3) : (word >> (8 * byte - 3));
                const Pair pair = *(LdsPairs)static_cast<uintptr_t>(and_or(shifted, 0x7f8u, FACTOR_AT));       // row 0 (a position past the segment): {1, 1}; row 128: a base outside ACGT
                everything_matches = j == 0 ? pair.x : everything_matches * pair.x;
                rho[j] = static_cast<uint32_t>(j) < length[s] ? pair.y : 1.0f;
            }
            // the normalised tables (pamld_stream.hpp): entry m of group g = product of rho over the group's mismatching positions
#pragma unroll
            for(int g = 0; g < 2; ++g) {
                const float r0 = rho[4 * g], r1 = rho[4 * g + 1], r2 = rho[4 * g + 2], r3 = rho[4 * g + 3];
                const float lo[4] = {1.0f, r0, r1, r0 * r1};
                const float hi[4] = {1.0f, r2, r3, r2 * r3};
#pragma unroll
                for(int m = 1; m < 16; ++m) {
                    const float t = (m >> 2) == 0 ? lo[m & 3] : (m & 3) == 0 ? hi[m >> 2] : lo[m & 3] * hi[m >> 2];
                    *reinterpret_cast<float*>(lds_raw + lane_const + static_cast<uint32_t>(g) * 4096u + static_cast<uint32_t>(m) * 256u) = t;
                }
            }
            // every word of the segment through the tables: slot 0 the named word, then the others; a slot's flags for
            // position j sit at bit j (positions 0..3: the first table, 4..7: the second)
            const uint32_t block = blocks_at[s] + (s == 0 ? c0 : c1) * (BLOCK_WORDS * 4u);
            float plain = 0.0f;
            float weighted = 0.0f;
            float winner = 0.0f;
            uint32_t mismatches = 0;
#pragma unroll
            for(int quad = 0; quad < SLOTS / 4; ++quad) {
                if(quad >= quads[s]) { break; }                    // (the same for every lane) slots past the segment's words hold nothing
                const Quad word_low = *(LdsQuads)static_cast<uintptr_t>(block + 16u * quad);
                const Quad word_high = *(LdsQuads)static_cast<uintptr_t>(block + 64u + 16u * quad);
                const Four holds = *(LdsFours)static_cast<uintptr_t>(block + 128u + 16u * quad);
#pragma unroll
                for(int k = 0; k < 4; ++k) {
                    const uint32_t fields = (segment_low[s] ^ word_low[k]) | (segment_high[s] ^ word_high[k]);
                    const float first_group = *(LdsFloats)static_cast<uintptr_t>(and_or(fields << 8, 0xf00u, lane_const));
                    const float second_group = *(LdsFloats)static_cast<uintptr_t>(and_or(fields << 4, 0xf00u, lane_const) + 4096u);
                    const float p = first_group * second_group;
                    if(quad == 0 && k == 0) { winner = p; mismatches = fields; }
                    else {
                        plain = __builtin_fmaf(holds[k], p, plain);
                        weighted = __builtin_fmaf(ratio[1 - s][quad][k], p, weighted);      // segment 0's words run down the column, segment 1's along the row
                    }
                }
            }
            winner_of[s] = winner;
            plain_of[s] = plain;
            weighted_of[s] = weighted;
            matches_of[s] = everything_matches;
            mismatches_of[s] = mismatches | ((outside >> start[s]) & ((1u << length[s]) - 1u));       // a base outside ACGT mismatches every word (barcode.h:147-161)
        }
        const float inverse0 = __builtin_amdgcn_rcpf(winner_of[0]), inverse1 = __builtin_amdgcn_rcpf(winner_of[1]);
        const float winner_probability = (winner_of[0] * matches_of[0]) * (winner_of[1] * matches_of[1]);      // P(r|c)
        const float listed = weighted_of[0] * inverse0 + weighted_of[1] * inverse1;
        const float x_rest = ratio[1][0][0] * (plain_of[0] * inverse0) * (plain_of[1] * inverse1) * near::SLOP;
        const float x_low = __builtin_fmaf(ratio[0][0][0], __builtin_amdgcn_rcpf(winner_probability), listed);
        const float x_mid = __builtin_fmaf(0.25f, x_rest, x_low);              // a quarter of the way into what was only bounded (pamld_near.hpp)
        const float y = x_mid * __builtin_amdgcn_rcpf(1.0f + x_mid);           // 1 - posterior
        const double confidence = 1.0 - static_cast<double>(y);                // pamld.cpp:92
        const double confidence_margin = confidence - confidence_threshold;
        const uint32_t k0 = static_cast<uint32_t>(__popc(mismatches_of[0])), k1 = static_cast<uint32_t>(__popc(mismatches_of[1]));
        const Pair info = *(LdsPairs)static_cast<uintptr_t>(infos_at + combination * (INFO_WORDS * 4u));
        const uint32_t bin = __float_as_uint(info.x);
        const uint32_t index = __float_as_uint(info.y);
        const bool finished = all_strict && known && bin != 0 && k0 <= 2 && k1 <= 2 && winner_probability > noise_floor && listed <= near::LISTED_MAX &&
                              x_low + x_rest <= near::X_MAX && x_rest <= near::REST_MAX && fabs(confidence_margin) > 2e-8;
        const bool defer = !finished;

        const uint32_t k = k0 + k1;
        const uint32_t hq_distance = static_cast<uint32_t>(__popc(mismatches_of[0] & (high_quality >> start[0])) + __popc(mismatches_of[1] & (high_quality >> start[1])));
        const bool confident = confidence_margin > 0;                               // pamld.cpp:99
        const bool qcfail = r.qcfail || !confident || (hq_threshold > 0 && static_cast<int>(hq_distance) >= hq_threshold);
        const bool qcfail_out = finished ? qcfail : r.qcfail;
        {
            typedef uint32_t vector4 __attribute__((ext_vector_type(4)));
            vector4 out;
            out.x = finished ? index : 0u;
            out.y = finished ? (k | (hq_distance << 8) | ((confident ? STATUS_PASS : STATUS_LOW_CONFIDENCE) << 16) | (qcfail_out ? 1u << 24 : 0u))
                             : ((static_cast<uint32_t>(STATUS_UNCLASSIFIED) << 16) | (qcfail_out ? 1u << 24 : 0u));
            out.z = finished ? static_cast<uint32_t>(__double2loint(confidence)) : 0u;
            out.w = finished ? static_cast<uint32_t>(__double2hiint(confidence)) : 0u;
            __builtin_nontemporal_store(out, reinterpret_cast<vector4*>(reinterpret_cast<char*>(args.result) + first * 16 + at * 16u));
            const uint32_t channel = (finished && multiplexing) ? index : static_cast<uint32_t>(r.channel);
            __builtin_nontemporal_store((channel & 0x7fffffffu) | (qcfail_out ? 0x80000000u : 0u), reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(args.word) + first * 4 + at * 4u));
        }
        r.qcfail = qcfail_out;
        if(finished && active && args.defer_counters) {
            // count, distance and confidence come from the records (counters.hip); the rare count goes straight to the buffer
            if(!confident) { atomicAdd(&args.stats[d.stats_offset + bin * STAT_WORDS_PER_BIN + FIELD_LOW_CONFIDENCE], 1ull); }      // pamld.cpp:104-106
        } else if(finished && active) {
            const uint32_t word = STAT_JOB_WORDS + bin * STAT_WORDS_PER_BIN + (qcfail ? 0u : static_cast<uint32_t>(STAT_FIELDS));
            uint32_t* const low_words = reinterpret_cast<uint32_t*>(histogram);      // a workgroup's counts and distances stay below 2^32
            atomicAdd(&low_words[2 * (word + FIELD_COUNT)], 1u);
            if(k) { atomicAdd(&low_words[2 * (word + FIELD_DISTANCE)], k); }
            if(confident) {
                const double shifted = confidence + 1.0;                          // round(confidence * 2^52) in the mantissa (pamld_near.hpp)
                const uint32_t fixed_low = static_cast<uint32_t>(__double2loint(shifted));
                const uint32_t fixed_high = static_cast<uint32_t>(__double2hiint(shifted)) - 0x3ff00000u;
                atomicAdd(&histogram[word + FIELD_CONFIDENCE_LOW], static_cast<unsigned long long>(fixed_low & 0x3ffffffu));
                atomicAdd(&histogram[word + FIELD_CONFIDENCE_HIGH], static_cast<unsigned long long>((fixed_high << 6) | (fixed_low >> 26)));
            } else {
                atomicAdd(&low_words[2 * (STAT_JOB_WORDS + bin * STAT_WORDS_PER_BIN + FIELD_LOW_CONFIDENCE)], 1u);      // pamld.cpp:104-106
            }
        }
        const unsigned long long deferred = __ballot(active && defer);
        pending_slot = 0xffffffffu;
        if(deferred) {
            const int lane = threadIdx.x & 63;
            if(active && defer) {
                pending_slot = segment_count + __popcll(deferred & ((1ull << lane) - 1ull));
                pending_read = static_cast<uint32_t>(i);
            }
            segment_count += static_cast<uint32_t>(__popcll(deferred));
        }
        count_outgoing(args, histogram, active && !defer, r);
    }
    if(pending_slot != 0xffffffffu) { segment[pending_slot] = pending_read; }
    if((threadIdx.x & 63u) == 0) { args.worklist_count[own_segment] = segment_count; }
    __syncthreads();
    histogram_flush(args, d, histogram);
}

template <int G, int STATE>
int launch_state(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    const size_t lds_bytes = histogram_at(host_decoder.plate_unique[0], host_decoder.plate_unique[1]) + histogram_bytes;
    static std::atomic<uint64_t> prepared{0};
    if(const int status = allow_large_lds(&pamld_plate_kernel<G, STATE>, prepared)) { return status; }
    hipLaunchKernelGGL((pamld_plate_kernel<G, STATE>), grid, dim3(THREADS), lds_bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}

template <int G>
int launch_one(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(!args.first_stage) { return launch_state<G, 2>(args, host_decoder, grid, histogram_bytes, stream); }
    return args.input_flag ? launch_state<G, 1>(args, host_decoder, grid, histogram_bytes, stream)
                           : launch_state<G, 0>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace plate

// the plate tables are there (DecoderSet::build_device_images), the counters fit the workgroup's histogram, and the reads the
// kernel hands on have the product form kernel behind them
bool pamld_plate_supports(const DevDecoder& host_decoder) {
    static const bool off = std::getenv("PHQ_NO_PLATE") != nullptr;
    const int words = host_decoder.plane_stride / 4;
    if(off || host_decoder.algorithm != 1 || host_decoder.plate_ready == 0 || words < 2 || words > 4) { return false; }
    // (round 6: a plate whose bins do not fit the workgroup — 171 to 256 wells — leaves the classified reads' counters to
    // count_results_kernel, as the other tier 0 kernels do for such codecs)
    if(host_decoder.use_lds_histogram == 0 && !count_results_supports(host_decoder)) { return false; }
    const size_t lds_bytes = plate::histogram_at(host_decoder.plate_unique[0], host_decoder.plate_unique[1]) +
                             sizeof(unsigned long long) * static_cast<size_t>(STAT_JOB_WORDS + (host_decoder.use_lds_histogram ? host_decoder.n_barcode + 1 : 1) * STAT_WORDS_PER_BIN);
    return lds_bytes <= 160u * 1024u && host_decoder.high_quality_threshold <= 128 && pamld_lut_supports(host_decoder);
}

int pamld_plate_block() { return plate::THREADS; }
int pamld_plate_waves() { return plate::WAVES; }

int launch_pamld_plate(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.plane_stride / 4) {
        case 2: return plate::launch_one<2>(args, host_decoder, grid, histogram_bytes, stream);
        case 3: return plate::launch_one<3>(args, host_decoder, grid, histogram_bytes, stream);
        case 4: return plate::launch_one<4>(args, host_decoder, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace phq
