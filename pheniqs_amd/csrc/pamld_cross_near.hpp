// K0c (neighbour lists, combinatorial sets): PamlDecoder<T>::classify for dual index codecs that hold EVERY combination
// of two word sets of equal length (5 to 10 nt each; C3: 96 x 96 words of 10 nt).
//
//   reference pamld.cpp:37-123   posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164  P(r|b) = pow(base, sum of Phred costs), distance, high quality distance
//
// P(r|b) factorises over the segments (pamld_cross.hpp): with c_s the word a segment sits on, W_s = P_s(c_s) and
// x_s = (sum of P_s over every other word of segment s) / W_s,
//     sum over every other combination / (W_0 W_1)  =  x_0 + x_1 + x_0 x_1 .
// Each x_s is what pamld_near.hpp computes for a single index: the word is named by a table lookup, it and the 14 or 15
// words nearest to it are scored with the per read product tables (the two segments use the same LDS tables one after
// the other), every other word is bounded.  The certificate is the one of pamld_near.hpp on the combined x; a read it
// does not cover goes to pamld_cross_kernel, a consumer of this workgroup's worklist segment.
#pragma once

#include "pamld_near.hpp"

namespace phq {
namespace cross_near {

using near::Shape;
using near::Quad;
using near::Pair;
using near::LdsFloats;
using near::LdsQuads;
using near::LdsPairs;
using near::and_or;
constexpr int FACTOR_ROWS = near::FACTOR_ROWS;
constexpr float X_MAX = 3e-4f;                 // as near::X_MAX, for the up to 20 positions of two segments
constexpr int SEGMENTS = 2;

// L = nucleotides of EACH segment; the plane row holds 2 L positions in ROW_WORDS words
template <int L> struct CrossShape {
    typedef Shape<L> One;
    static constexpr int ROW_WORDS = (2 * L + 3) / 4;
    static constexpr uint32_t RECORD_AT = One::RECORD_AT;                      // segment 0's records, then segment 1's
    __host__ __device__ static constexpr uint32_t histogram_at(int words0, int words1) {
        return RECORD_AT + static_cast<uint32_t>(words0 + words1) * One::RECORD_WORDS * 4u;
    }
};

// PRIORS: the combinations' priors differ (estimated priors, phq_adjust_prior).  With rho_ij = prior_ij / prior of the named
// combination (c0, c1),
//     sum over every other combination / (prior W_0 W_1)  =  sum_j rho_c0j P_1(j)/W_1  +  sum_i rho_ic1 P_0(i)/W_0
//                                                           +  sum_{i != c0, j != c1} rho_ij P_0(i) P_1(j) / (W_0 W_1) :
// the first two sums are taken over the listed words with their own rho (DevDecoder::cross_near_weight, 2 x 64 B per read
// from a table that stays in L2), everything else is bounded with the largest rho.
template <int L, int STATE, bool PRIORS>
__global__ __launch_bounds__(Shape<L>::THREADS) void pamld_cross_near_kernel(StageArgs args) {
    typedef Shape<L> Layout;
    constexpr int G = Layout::G;
    constexpr int THREADS = Layout::THREADS;
    constexpr int FIELD = Layout::FIELD;
    constexpr int PER_WORD = Layout::PER_WORD;
    constexpr int LIST_WORDS = Layout::LIST_WORDS;
    constexpr int RECORD_WORDS = Layout::RECORD_WORDS;
    constexpr int ROW_WORDS = CrossShape<L>::ROW_WORDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    float2* const factor_single = reinterpret_cast<float2*>(lds_raw + Layout::FACTOR_AT);
    uint32_t& deferred_total = *reinterpret_cast<uint32_t*>(lds_raw + Layout::DEFERRED_AT);
    uint32_t* const records = reinterpret_cast<uint32_t*>(lds_raw + Layout::RECORD_AT);
    DecoderRef d = *constant_cast(args.decoder);
    const int unique0 = d.cross_unique[0], unique1 = d.cross_unique[1];
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw + CrossShape<L>::histogram_at(unique0, unique1));

    for(int i = threadIdx.x; i < FACTOR_ROWS; i += THREADS) {
        const double m = i < 128 ? d.match_probability[i] : d.uniform_probability;
        const double e = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        factor_single[i] = make_float2(static_cast<float>(m), static_cast<float>(e));
    }
    if(threadIdx.x == 0) { deferred_total = 0; }
    for(int w = threadIdx.x; w < unique0 * RECORD_WORDS; w += THREADS) { records[w] = d.cross_near_record[0][w]; }
    for(int w = threadIdx.x; w < unique1 * RECORD_WORDS; w += THREADS) { records[unique0 * RECORD_WORDS + w] = d.cross_near_record[1][w]; }
    histogram_clear(d, histogram);
    __syncthreads();

    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const uint16_t* const nearest[SEGMENTS] = {d.cross_near_table[0], d.cross_near_table[1]};
    const uint32_t record_base[SEGMENTS] = {Layout::RECORD_AT, Layout::RECORD_AT + static_cast<uint32_t>(unique0) * RECORD_WORDS * 4u};
    const uint32_t words_of[SEGMENTS] = {static_cast<uint32_t>(unique0), static_cast<uint32_t>(unique1)};
    const int64_t n = args.n_read;
    const int64_t tiles = (n + THREADS - 1) / THREADS;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t lane_const = ((threadIdx.x & 63u) << 2) | (wave << 12);
    const int high_quality_threshold = min(max(d.high_quality_threshold, 0), 128);
    const uint32_t high_quality_addend = static_cast<uint32_t>(128 - high_quality_threshold) * 0x01010101u;
    const float noise_over_prior = static_cast<float>(d.adjusted_noise_probability / d.concentration[0]);       // uniform priors
    const float noise_floor = static_cast<float>(d.random_barcode_probability) * near::SLOP;
    const double confidence_threshold = d.confidence_threshold;
    auto valid = [](int w) -> uint32_t { return L - 4 * w >= 4 ? 0xffffffffu : ((1u << (8 * (L - 4 * w))) - 1u); };

    struct Inputs { uint32_t code[ROW_WORDS]; uint32_t quality[ROW_WORDS]; uint32_t state; };
    auto fetch = [&](int64_t tile, Inputs& in) {
        const int64_t first = tile * THREADS;
        const uint32_t last = static_cast<uint32_t>(min(static_cast<int64_t>(THREADS), n - first)) - 1u;
        const uint32_t at = min(threadIdx.x, last);
        const uint32_t* const code_row = reinterpret_cast<const uint32_t*>(args.base + first * (4 * ROW_WORDS) + at * (4u * ROW_WORDS));
        const uint32_t* const quality_row = reinterpret_cast<const uint32_t*>(args.quality + first * (4 * ROW_WORDS) + at * (4u * ROW_WORDS));
#pragma unroll
        for(int w = 0; w < ROW_WORDS; ++w) { in.code[w] = code_row[w]; in.quality[w] = quality_row[w]; }
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
    const bool lds_counters = d.use_lds_histogram != 0 && args.defer_counters == 0;
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
        // the words of each segment, funnel shifted to its first byte (segment s starts at position s L)
        uint32_t x[SEGMENTS][G], q7[SEGMENTS][G], q8[SEGMENTS][G];
        uint32_t low[SEGMENTS], high[SEGMENTS], high_quality[SEGMENTS], key[SEGMENTS];
        uint32_t not_strict = 0;
#pragma unroll
        for(int s = 0; s < SEGMENTS; ++s) {
            const int first_word = (s * L) >> 2;
            const int byte_shift = (s * L) & 3;
#pragma unroll
            for(int w = 0; w < G; ++w) {
                const int a = first_word + w < ROW_WORDS ? first_word + w : ROW_WORDS - 1;
                const int b = first_word + w + 1 < ROW_WORDS ? first_word + w + 1 : ROW_WORDS - 1;
                const uint32_t code = byte_shift ? __builtin_amdgcn_alignbyte(in.code[b], in.code[a], byte_shift) : in.code[a];
                const uint32_t quality = byte_shift ? __builtin_amdgcn_alignbyte(in.quality[b], in.quality[a], byte_shift) : in.quality[a];
                x[s][w] = code & (0x0f0f0f0fu & valid(w));
                q8[s][w] = quality & valid(w);
                q7[s][w] = quality & (0x7f7f7f7fu & valid(w));
                const uint32_t z = x[s][w] & ((x[s][w] | 0x80808080u) - 0x01010101u);
                const uint32_t strict = (x[s][w] + 0x7f7f7f7fu) & ~(z + 0x7f7f7f7fu) & 0x80808080u;
                not_strict |= strict ^ (0x80808080u & valid(w));
            }
            low[s] = gather((((x[s][0] >> 1) | (x[s][0] >> 3)) & 0x01010101u) | (((x[s][1] << 3) | (x[s][1] << 1)) & 0x10101010u));
            high[s] = gather((((x[s][0] >> 2) | (x[s][0] >> 3)) & 0x01010101u) | (((x[s][1] << 2) | (x[s][1] << 1)) & 0x10101010u));
            high_quality[s] = gather(((((q7[s][0] + high_quality_addend) | q8[s][0]) & 0x80808080u) >> 7) |
                                     ((((q7[s][1] + high_quality_addend) | q8[s][1]) & (0x80808080u & valid(1))) >> 3));
            if(G == 3) {
                const uint32_t both = gather((((x[s][G - 1] >> 1) | (x[s][G - 1] >> 3)) & 0x01010101u) | (((x[s][G - 1] << 2) | (x[s][G - 1] << 1)) & 0x10101010u));
                low[s] |= (both & 0xfu) << 8;
                high[s] |= (both >> 4) << 8;
                high_quality[s] |= (gather((((q7[s][G - 1] + high_quality_addend) | q8[s][G - 1]) & (0x80808080u & valid(G - 1))) >> 7) & 0xfu) << 8;
            }
            key[s] = low[s] | (high[s] << L);
        }
        const bool all_strict = not_strict == 0;
        if(pending_slot != 0xffffffffu) { segment[pending_slot] = pending_read; }      // the previous tile's deferred reads
        uint32_t looked_up[SEGMENTS];
#pragma unroll
        for(int s = 0; s < SEGMENTS; ++s) { looked_up[s] = nearest[s][key[s]]; }
        fetch(min(tile + gridDim.x, tiles - 1), next);

        // per segment: x_s bounds, the winner's probability, mismatches
        float winner_of[SEGMENTS], x_listed_of[SEGMENTS], x_rest_of[SEGMENTS], x_weighted_of[SEGMENTS];
        typedef float Weights __attribute__((ext_vector_type(4)));
        Weights along_row[4], along_column[4];                 // PRIORS: cross_near_weight of this read's combination
        uint32_t mismatches_of[SEGMENTS], flat_of[SEGMENTS];
        bool usable = true;
#pragma unroll
        for(int s = 0; s < SEGMENTS; ++s) {
            const bool known = looked_up[s] < words_of[s];
            const uint32_t c = known ? looked_up[s] : 0u;
            constexpr uint32_t REPLICATE = PER_WORD == 4 ? 0x01010101u : 0x00100401u;
            const uint32_t observed_low = low[s] * REPLICATE;
            const uint32_t observed_high = high[s] * REPLICATE;
            float fm[4 * G];
            float fe[4 * G];
#pragma unroll
            for(int j = 0; j < L; ++j) {
                const uint32_t q = q7[s][j >> 2];
                const int byte = j & 3;
                const uint32_t shifted = byte == 0 ? (q << 3) : (q >> (8 * byte - 3));
                const Pair pair = *(LdsPairs)static_cast<uintptr_t>(and_or(shifted, 0x3f8u, static_cast<uint32_t>(Layout::FACTOR_AT)));
                fm[j] = pair.x;
                fe[j] = pair.y;
            }
#pragma unroll
            for(int j = L; j < 4 * G; ++j) { fm[j] = 1.0f; fe[j] = 1.0f; }
            float ge[G][5];
            float everything_matches = 1.0f;
#pragma unroll
            for(int g = 0; g < G; ++g) {
                const int p0 = 4 * g;
                float lo[4];
                float hi[4];
                lo[0] = fm[p0] * fm[p0 + 1]; lo[1] = fe[p0] * fm[p0 + 1]; lo[2] = fm[p0] * fe[p0 + 1]; lo[3] = fe[p0] * fe[p0 + 1];
                hi[0] = fm[p0 + 2] * fm[p0 + 3]; hi[1] = fe[p0 + 2] * fm[p0 + 3]; hi[2] = fm[p0 + 2] * fe[p0 + 3]; hi[3] = fe[p0 + 2] * fe[p0 + 3];
                const int positions = Layout::positions(g);
                const int entries = 1 << positions;
                float t[16];
                float by_count[5] = {0, 0, 0, 0, 0};
#pragma unroll
                for(int m = 0; m < 16; ++m) {
                    if(m < entries) {
                        t[m] = positions <= 2 ? lo[m & 3] : lo[m & 3] * hi[m >> 2];
                        *reinterpret_cast<float*>(lds_raw + Layout::TABLE_AT + static_cast<uint32_t>(g) * Layout::GROUP_BYTES + (lane_const | (static_cast<uint32_t>(m) << 8))) = t[m];
                        const int a = __builtin_popcount(static_cast<unsigned>(m));
                        by_count[a] = (m == (1 << a) - 1) ? t[m] : fmaxf(by_count[a], t[m]);
                    }
                }
                everything_matches = g == 0 ? t[0] : everything_matches * t[0];
#pragma unroll
                for(int a = 4; a >= 0; --a) {
                    ge[g][a] = a > positions ? 0.0f : a == positions ? by_count[a] : fmaxf(by_count[a], ge[g][a + 1 > 4 ? 4 : a + 1]);
                }
            }
            float h[7] = {0, 0, 0, 0, 0, 0, 0};
            {
                float front[9];
#pragma unroll
                for(int m = 0; m <= 8; ++m) {
                    front[m] = 0.0f;
                    bool any = false;
#pragma unroll
                    for(int a = 0; a <= 4; ++a) {
                        const int b = m - a;
                        if(b >= 0 && b <= 4 && a <= Layout::positions(0) && b <= Layout::positions(1)) {
                            const float product = ge[0][a] * ge[1][b];
                            front[m] = any ? fmaxf(front[m], product) : product;
                            any = true;
                        }
                    }
                }
#pragma unroll
                for(int count = 3; count <= 6; ++count) {
                    if(G == 2) { h[count] = front[count]; }
                    else {
                        bool any = false;
#pragma unroll
                        for(int third = 0; third <= 4; ++third) {
                            const int m = count - third;
                            if(third <= Layout::positions(G - 1) && m >= 0 && m <= 8) {
                                const float product = front[m] * ge[G - 1][third];
                                h[count] = any ? fmaxf(h[count], product) : product;
                                any = true;
                            }
                        }
                    }
                }
            }
            if(PRIORS && s == 0) {
                // behind segment 0's tables in program order: the lookups had those to arrive under, these have the sums below
                uint32_t flat = 0;
#pragma unroll
                for(int t = 0; t < SEGMENTS; ++t) {
                    const uint32_t word = looked_up[t] < words_of[t] ? looked_up[t] : 0u;
                    flat += *(__attribute__((address_space(3))) const uint32_t*)static_cast<uintptr_t>(record_base[t] + word * (RECORD_WORDS * 4u) + (2 * LIST_WORDS + 3) * 4u);
                }
                const Weights* const row = reinterpret_cast<const Weights*>(d.cross_near_weight[0]) + static_cast<size_t>(flat) * (CROSS_NEAR_WEIGHTS / 4);
                const Weights* const column = reinterpret_cast<const Weights*>(d.cross_near_weight[1]) + static_cast<size_t>(flat) * (CROSS_NEAR_WEIGHTS / 4);
#pragma unroll
                for(int quad = 0; quad < 4; ++quad) { along_column[quad] = column[quad]; }
#pragma unroll
                for(int quad = 0; quad < 4; ++quad) { along_row[quad] = row[quad]; }
            }
            const uint32_t record_at = record_base[s] + c * (RECORD_WORDS * 4u);
            Quad record[Layout::RECORD_QUADS];
#pragma unroll
            for(int quad = 0; quad < Layout::RECORD_QUADS; ++quad) { record[quad] = *(LdsQuads)static_cast<uintptr_t>(record_at + 16u * quad); }
            auto record_word = [&](int w) { return record[w >> 2][w & 3]; };
            float winner = 0;
            float others = 0;
            float weighted = 0;               // PRIORS: the listed words by their prior ratios
            uint32_t mismatches = 0;
#pragma unroll
            for(int w = 0; w < LIST_WORDS; ++w) {
                const uint32_t fields = (observed_low ^ record_word(2 * w)) | (observed_high ^ record_word(2 * w + 1));
                if(w == 0) { mismatches = fields & ((1u << L) - 1u); }
                float p[PER_WORD];
#pragma unroll
                for(int b = 0; b < PER_WORD; ++b) {
#pragma unroll
                    for(int g = 0; g < G; ++g) {
                        const int at_bit = FIELD * b + 4 * g;
                        const uint32_t mask = ((1u << Layout::positions(g)) - 1u) << 8;
                        const uint32_t address = and_or(at_bit < 8 ? (fields << (8 - at_bit)) : (fields >> (at_bit - 8)), mask, lane_const);
                        const float entry = ((LdsFloats)static_cast<uintptr_t>(address))[(Layout::TABLE_AT + g * Layout::GROUP_BYTES) / 4];
                        p[b] = g == 0 ? entry : p[b] * entry;
                    }
                }
                float sum = 0;
#pragma unroll
                for(int b = 0; b < PER_WORD; ++b) {
                    if(w == 0 && b == 0) { winner = p[0]; }
                    else {
                        sum = (w == 0 && b == 1) ? p[b] : sum + p[b];
                        if(PRIORS) {
                            const int entry = w * PER_WORD + b;               // the order of the record = the order of the weights
                            const float ratio = s == 0 ? along_column[entry >> 2][entry & 3] : along_row[entry >> 2][entry & 3];
                            weighted = fmaf(ratio, p[b], weighted);
                        }
                    }
                }
                others = w == 0 ? sum : others + sum;
            }
            const uint32_t radius = record_word(2 * LIST_WORDS);
            const float cnt_r = __uint_as_float(record_word(2 * LIST_WORDS + 1));
            const float cnt_rest = __uint_as_float(record_word(2 * LIST_WORDS + 2));
            flat_of[s] = record_word(2 * LIST_WORDS + 3);                        // this word's share of the barcode ordinal
            const int k = __popc(mismatches);
            const bool near5 = radius >= 5u;
            const bool near3 = radius == 3u;                 // pamld_near.hpp: only a read ON the word
            const float h_r = near5 ? h[5] : near3 ? h[3] : h[4];
            const float h_up = near5 ? h[6] : near3 ? h[4] : h[5];
            const float h_down = near5 ? h[4] : h[3];
            const float inverse_winner = __builtin_amdgcn_rcpf(winner);
            const float inverse_rho = everything_matches * inverse_winner;
            const float boost = fmaxf(winner * __builtin_amdgcn_rcpf(everything_matches), 1.0f);
            const float at_r = k == 0 ? h_r : boost * fminf(h_down, h_r * inverse_rho);
            const float beyond = k == 0 ? h_up : boost * fminf(h_r, h_up * inverse_rho);
            winner_of[s] = winner;
            x_listed_of[s] = others * inverse_winner;
            x_weighted_of[s] = weighted * inverse_winner;
            x_rest_of[s] = (cnt_r * at_r + cnt_rest * beyond) * near::SLOP * inverse_winner;
            mismatches_of[s] = mismatches;
            usable = usable && known && k <= 1 && (radius >= 4u || (near3 && k == 0));
        }

        // both segments together: others / winner = x_0 + x_1 + x_0 x_1, between its listed part and listed + bound
        const float winner = winner_of[0] * winner_of[1];
        const float upper0 = x_listed_of[0] + x_rest_of[0], upper1 = x_listed_of[1] + x_rest_of[1];
        // PRIORS: the winner's row and column by their own priors; the other combinations and what the lists leave out of the
        // row and the column by the largest prior over the winner's (along_column[0].x)
        const float listed = PRIORS ? x_weighted_of[0] + x_weighted_of[1] : x_listed_of[0] + x_listed_of[1] + x_listed_of[0] * x_listed_of[1];
        const float x_rest = PRIORS ? along_column[0].x * (upper0 * upper1 + x_rest_of[0] + x_rest_of[1]) * near::SLOP
                                    : (upper0 + upper1 + upper0 * upper1 - listed) * near::SLOP;
        const float inverse_winner = __builtin_amdgcn_rcpf(winner);
        const float x_low = listed + (PRIORS ? along_row[0].x : noise_over_prior) * inverse_winner;
        // the posterior is taken a QUARTER of the way into what was only bounded: those barcodes really weigh between nothing and
        // x_rest, typically a tenth of it (profiles/tools/sim/near_certificate.py), so the error is at most 0.75 x_rest —
        // 6.8e-9 at REST_MAX, with x's own 7e-10 on top 7.5e-9: a proven distance from the 1e-8 the tests assert (until round 5
        // it was taken without them: at most x_rest, 9.7e-9 with the rounding, and 8.8e-9 was seen) — and typically 0.15 of it
        const float x_mid = __builtin_fmaf(0.25f, x_rest, x_low);
        const float y = x_mid * __builtin_amdgcn_rcpf(1.0f + x_mid);       // 1 - posterior
        const double confidence = 1.0 - static_cast<double>(y);            // pamld.cpp:92
        const double confidence_margin = confidence - confidence_threshold;
        // a low confidence read is left to the next tier here: its counter lives in a bin this kernel may not hold
        const bool finished = all_strict && usable && winner > noise_floor && listed <= near::LISTED_MAX && x_low + x_rest <= X_MAX &&
                              x_rest <= near::REST_MAX && confidence_margin > 2e-8;
        const bool defer = !finished;

        const uint32_t ordinal = flat_of[0] + flat_of[1];                          // barcodes in lexicographic order of the words (host checked)
        const uint32_t index = ordinal + 1u;                                       // barcode_index is the identity (host checked)
        const int k = __popc(mismatches_of[0]) + __popc(mismatches_of[1]);
        const uint32_t hq_distance = static_cast<uint32_t>(__popc(mismatches_of[0] & high_quality[0]) + __popc(mismatches_of[1] & high_quality[1]));
        const bool qcfail = r.qcfail || (hq_threshold > 0 && static_cast<int>(hq_distance) >= hq_threshold);
        const bool qcfail_out = finished ? qcfail : r.qcfail;
        {
            typedef uint32_t vector4 __attribute__((ext_vector_type(4)));
            vector4 out;
            out.x = index;
            out.y = static_cast<uint32_t>(k) | (hq_distance << 8) | (static_cast<uint32_t>(STATUS_PASS) << 16) | (qcfail_out ? 1u << 24 : 0u);
            out.z = static_cast<uint32_t>(__double2loint(confidence));
            out.w = static_cast<uint32_t>(__double2hiint(confidence));
            __builtin_nontemporal_store(out, reinterpret_cast<vector4*>(reinterpret_cast<char*>(args.result) + first * 16 + at * 16u));
            const uint32_t channel = (finished && multiplexing) ? index : static_cast<uint32_t>(r.channel);
            __builtin_nontemporal_store((channel & 0x7fffffffu) | (qcfail_out ? 0x80000000u : 0u), reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(args.word) + first * 4 + at * 4u));
        }
        r.qcfail = qcfail_out;
        if(finished && active && lds_counters) {
            const uint32_t word = STAT_JOB_WORDS + (ordinal + 1u) * STAT_WORDS_PER_BIN + (qcfail ? 0u : static_cast<uint32_t>(STAT_FIELDS));
            uint32_t* const low_words = reinterpret_cast<uint32_t*>(histogram);
            atomicAdd(&low_words[2 * (word + FIELD_COUNT)], 1u);
            if(k) { atomicAdd(&low_words[2 * (word + FIELD_DISTANCE)], static_cast<uint32_t>(k)); }
            const double shifted = confidence + 1.0;                          // round(confidence * 2^52) in the mantissa (pamld_near.hpp)
            const uint32_t fixed_low = static_cast<uint32_t>(__double2loint(shifted));
            const uint32_t fixed_high = static_cast<uint32_t>(__double2hiint(shifted)) - 0x3ff00000u;
            atomicAdd(&histogram[word + FIELD_CONFIDENCE_LOW], static_cast<unsigned long long>(fixed_low & 0x3ffffffu));
            atomicAdd(&histogram[word + FIELD_CONFIDENCE_HIGH], static_cast<unsigned long long>((fixed_high << 6) | (fixed_low >> 26)));
        }
        const unsigned long long deferred = __ballot(active && defer);
        pending_slot = 0xffffffffu;
        if(deferred) {
            const int lane = threadIdx.x & 63;
            uint32_t run = 0;
            if(lane == 0) { run = atomicAdd(&deferred_total, static_cast<uint32_t>(__popcll(deferred))); }
            run = __shfl(run, 0);
            if(active && defer) {
                pending_slot = run + __popcll(deferred & ((1ull << lane) - 1ull));
                pending_read = static_cast<uint32_t>(i);
            }
        }
        count_outgoing(args, histogram, active && !defer, r);
    }
    if(pending_slot != 0xffffffffu) { segment[pending_slot] = pending_read; }
    __syncthreads();
    if(threadIdx.x == 0) { args.worklist_count[blockIdx.x] = deferred_total; }
    histogram_flush(args, d, histogram);
}

template <int L, int STATE, bool PRIORS>
int launch_state(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    const size_t lds_bytes = CrossShape<L>::histogram_at(host_decoder.cross_unique[0], host_decoder.cross_unique[1]) + histogram_bytes;
    static std::atomic<uint64_t> prepared{0};
    if(const int status = allow_large_lds(&pamld_cross_near_kernel<L, STATE, PRIORS>, prepared)) { return status; }
    hipLaunchKernelGGL((pamld_cross_near_kernel<L, STATE, PRIORS>), grid, dim3(Shape<L>::THREADS), lds_bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}

template <int L, bool PRIORS>
int launch_by_state(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(!args.first_stage) { return launch_state<L, 2, PRIORS>(args, host_decoder, grid, histogram_bytes, stream); }
    return args.input_flag ? launch_state<L, 1, PRIORS>(args, host_decoder, grid, histogram_bytes, stream)
                           : launch_state<L, 0, PRIORS>(args, host_decoder, grid, histogram_bytes, stream);
}

template <int L>
int launch_one(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return host_decoder.uniform_prior ? launch_by_state<L, false>(args, host_decoder, grid, histogram_bytes, stream)
                                      : launch_by_state<L, true>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace cross_near
}  // namespace phq
