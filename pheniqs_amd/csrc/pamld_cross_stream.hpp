// K0cs (streaming, combinatorial sets): PamlDecoder<T>::classify for dual index codecs that hold EVERY combination of two
// word sets of equal length (8 to 10 nt each; C3: 96 x 96 words of 10 nt), neighbour lists and factorised scan in ONE
// persistent kernel — pamld_stream.hpp's pipeline per wave over pamld_cross_near.hpp's arithmetic.
//
//   reference pamld.cpp:37-123   posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164  P(r|b) = pow(base, sum of Phred costs), distance, high quality distance
//
// P(r|b) factorises over the segments (pamld_cross.hpp): with c_s the word a segment sits on, W_s = P_s(c_s) and
// x_s = (sum of P_s over every other word of segment s) / W_s,
//     sum over every other combination / (W_0 W_1)  =  x_0 + x_1 + x_0 x_1 .
//   * pass A, every 64 read tile, each segment in turn through the same per read tables: the word is named by a table
//     lookup, it and the words nearest to it (15 for 8 nt words, 7 for 9 and 10 nt words, whose positions 8 and 9 are picked
//     by masks) are scored in units of the segment's all-match product, every other word is bounded by four distance
//     classes (pamld_stream.hpp); the certificate is pamld_cross_near.hpp's on the combined x.  Nothing is counted here:
//     a codec of this size has its counters rebuilt from the records (counters.hip);
//   * what pass A cannot certify (~10 % of C3's reads: either segment may fail) waits in the wave's 64 entry ring in LDS —
//     its ordinal and carried state; the input row (40 bytes of C3) is read again by pass C, from L2;
//   * pass C, when the ring would overflow: the factorised single precision scan of pamld_cross.hpp over both segments'
//     words with all lanes busy, the winner again in fp64, its certificate (dominant winner or a certain argmax below the
//     noise filter); what is left goes to the wave's segment of the HBM worklist for the exact tier and the literal kernel.
// Rounds 2 and 3 ran this as two launches with a worklist, carry-less gathers and placeholder records between them:
// 3.2 + 1.4 ms per 100 M reads of C3.
#pragma once

#include "kernel_common.hpp"
#include "pamld_lut.hpp"
#include "pamld_near.hpp"
#include "pamld_cross.hpp"

namespace phq {
namespace cross_stream {

constexpr int FACTOR_ROWS = 129;
constexpr int RING = 64;
constexpr int SEGMENTS = 2;
constexpr float X_MAX = 3e-4f;                 // as pamld_cross_near.hpp: the up to 20 positions of two segments

// L = nucleotides of EACH segment.  The LDS map: per read tables of the waves, factor rows, rings, the two segments' words
// (one barcode per plane word pair, blocks of 8: eight low plane words then eight high plane words), their blocks — a
// word's list as LISTED low + LISTED high plane words and the four weights of the bound —, a small histogram (the job
// words and bin 0).  The host puts block ADDRESSES into the lookup tables (cross_stream_block_at).
template <int L, int WAVES> struct Shape {
    static_assert(L >= 5 && L <= 10, "two groups of 4 positions, and up to two positions picked by masks");
    static_assert(WAVES >= 2 && WAVES <= 16 && WAVES % 2 == 0, "pairs of waves share a 16 KiB block of tables");
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int TABLED = L > 8 ? 8 : L;
    static constexpr int TAIL = L - TABLED;
    static constexpr int G = (L + 3) / 4;                          // words of a segment, funnel shifted to its first byte
    static constexpr int ROW_WORDS = (2 * L + 3) / 4;              // words of the whole input row
    static constexpr int LISTED = stream_list(L);
    static constexpr int BLOCK_WORDS = cross_stream_block_words(L);
    static constexpr uint32_t BLOCK_BYTES = BLOCK_WORDS * 4u;
    static constexpr int BLOCK_QUADS = (2 * LISTED + 4) / 4;       // what pass A reads of a block
    static constexpr int RING_PLANES = 2;                          // carried state, ordinal: pass C reads the input row again (it is in L2)
    static constexpr int RINGS = 3;                                // per wave: segment 0 to scan, segment 1 to scan, both
    static constexpr uint32_t TABLE_BYTES = WAVES * 8192u;
    static constexpr uint32_t FACTOR_AT = TABLE_BYTES;
    static constexpr uint32_t DOUBLE_AT = FACTOR_AT + 1040;
    static constexpr uint32_t RING_AT = DOUBLE_AT + FACTOR_ROWS * 16;
    static constexpr uint32_t RING_BYTES = RINGS * RING_PLANES * RING * 4;
    static constexpr uint32_t WORDS_AT = RING_AT + WAVES * RING_BYTES;
    __host__ __device__ static constexpr uint32_t words_bytes(int unique) { return static_cast<uint32_t>((unique + 7) & ~7) * 8u; }
    __host__ __device__ static constexpr uint32_t words_at(int unique0, int s) { return WORDS_AT + (s ? words_bytes(unique0) : 0u); }
    __host__ __device__ static constexpr uint32_t block_at(int unique0, int unique1, int s) {
        return WORDS_AT + words_bytes(unique0) + words_bytes(unique1) + (s ? static_cast<uint32_t>(unique0) * BLOCK_BYTES : 0u);
    }
    __host__ __device__ static constexpr uint32_t histogram_at(int unique0, int unique1) { return block_at(unique0, unique1, 1) + static_cast<uint32_t>(unique1) * BLOCK_BYTES; }
    __host__ __device__ static constexpr uint32_t lds_bytes(int unique0, int unique1) { return histogram_at(unique0, unique1) + 8u * (STAT_JOB_WORDS + STAT_WORDS_PER_BIN); }
    static_assert(block_at(96, 96, 1) == cross_stream_block_at(WAVES, L, 96, 96, 1), "the host puts block addresses into the lookup tables");
    __host__ __device__ static constexpr int positions(int g) { return TABLED - 4 * g >= 4 ? 4 : TABLED - 4 * g; }
    __host__ __device__ static constexpr uint32_t valid(int w) { return L - 4 * w >= 4 ? 0xffffffffu : L - 4 * w <= 0 ? 0u : ((1u << (8 * (L - 4 * w))) - 1u); }
    static constexpr uint32_t FULL_MASK = (L >= 8 ? 0x3c3cu : (0x3cu | (((1u << (L - 4)) - 1u) << 10))) | (TAIL ? (((1u << TAIL) - 1u) << 18) : 0u);
    static_assert(FACTOR_AT % 1024 == 0 && DOUBLE_AT % 16 == 0 && RING_AT % 16 == 0 && WORDS_AT % 16 == 0 && BLOCK_BYTES % 16 == 0, "alignment");
};

typedef uint32_t Quad __attribute__((ext_vector_type(4)));
typedef float Pair __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const float* LdsFloats;
typedef __attribute__((address_space(3))) const uint32_t* LdsWords;
typedef __attribute__((address_space(3))) const Quad* LdsQuads;
typedef __attribute__((address_space(3))) const Pair* LdsPairs;
using near::and_or;
using lut::low_bit;

// STATE: where a read's carried state comes from: 0 nothing to load, 1 the input flag bytes, 2 the status word of the
// stage before
// PRIORS (round 4): the combinations' priors differ (estimated priors, phq_adjust_prior).  Pass A then weighs the winner's row
// and column by their own priors — the listed words' prior ratios per combination, 2 x 64 bytes a read from L2
// (DevDecoder::cross_near_weight, as pamld_cross_near.hpp) —, bounds every other combination and what the lists leave out at
// the largest prior over the winner's; pass C (drain_weighted below) takes the reads ONE of whose segments pass A found good,
// scanning the other segment along a row of the prior matrix; a read with no good segment goes to the wave's worklist
// segment for the weighted scan kernel (pamld_cross.hpp, WEIGHTED, as consumer).
template <int L, int STATE, int WAVES, bool PRIORS = false>
__global__ __launch_bounds__((Shape<L, WAVES>::THREADS)) __attribute__((amdgpu_waves_per_eu(WAVES / 4, (WAVES + 3) / 4)))
void pamld_cross_stream_kernel(StageArgs args) {
    typedef Shape<L, WAVES> Layout;
    constexpr int THREADS = Layout::THREADS;
    constexpr int TAIL = Layout::TAIL;
    constexpr int G = Layout::G;
    constexpr int ROW_WORDS = Layout::ROW_WORDS;
    constexpr int LISTED = Layout::LISTED;
    constexpr int RHO = TAIL ? 10 : 8;
    constexpr uint32_t FULL_MASK = Layout::FULL_MASK;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    float2* const factor_single = reinterpret_cast<float2*>(lds_raw + Layout::FACTOR_AT);                 // [row] {P(match), rho}
    double (*const factor_double)[2] = reinterpret_cast<double (*)[2]>(lds_raw + Layout::DOUBLE_AT);      // [row][0 match | 1 mismatch]
    DecoderRef d = *constant_cast(args.decoder);
    const int unique[SEGMENTS] = {d.cross_unique[0], d.cross_unique[1]};
    const int radix[SEGMENTS] = {d.cross_stride[0], d.cross_stride[1]};
    uint32_t* const lds_words[SEGMENTS] = {reinterpret_cast<uint32_t*>(lds_raw + Layout::words_at(unique[0], 0)), reinterpret_cast<uint32_t*>(lds_raw + Layout::words_at(unique[0], 1))};
    unsigned long long* const histogram = reinterpret_cast<unsigned long long*>(lds_raw + Layout::histogram_at(unique[0], unique[1]));      // the two job words, then bin 0

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    // this lane's corner of the tables (pamld_stream.hpp): everything of an entry address but group (bit 8) and entry (bits 10..13)
    const uint32_t wave_byte = ((wave & 1u) << 1) | (((wave >> 1) & 3u) << 6);
    const uint32_t lane_low = (lane << 2) | ((wave >> 3) << 16);
    const uint32_t lane_const = lane_low | (wave_byte << 8);
    const uint32_t wave_bytes = wave_byte * 0x01010101u;

    for(int i = threadIdx.x; i < FACTOR_ROWS; i += THREADS) {
        const double m = i < 128 ? d.match_probability[i] : d.uniform_probability;
        const double e = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        factor_single[i] = make_float2(static_cast<float>(m), static_cast<float>(e / m));
        factor_double[i][0] = m;
        factor_double[i][1] = e;
    }
#pragma unroll
    for(int s = 0; s < SEGMENTS; ++s) {
        // per block of 8 words: their low plane words, then their high plane words; zero past the segment's words
        const int padded = (unique[s] + 7) & ~7;
        for(int w = threadIdx.x; w < padded; w += THREADS) {
            const uint32_t sliced = w < unique[s] ? d.cross_words[s * CROSS_MAX_UNIQUE + w] : 0u;
            const int at = 16 * (w / 8) + (w % 8);
            lds_words[s][at] = lut::low_plane(sliced);
            lds_words[s][at + 8] = lut::high_plane(sliced);
        }
        uint32_t* const blocks = reinterpret_cast<uint32_t*>(lds_raw + Layout::block_at(unique[0], unique[1], s));
        for(int w = threadIdx.x; w < unique[s] * Layout::BLOCK_WORDS; w += THREADS) { blocks[w] = d.cross_stream_record[s][w]; }
    }
    for(int i = threadIdx.x; i < STAT_JOB_WORDS + STAT_WORDS_PER_BIN; i += THREADS) { histogram[i] = 0; }
    // entry 0 of both tables (no mismatch in the group) is 1 in units of the all-match product: written once
    *reinterpret_cast<float*>(lds_raw + lane_const) = 1.0f;
    *reinterpret_cast<float*>(lds_raw + lane_const + 256u) = 1.0f;
    __syncthreads();

    const uint32_t histogram_at = Layout::histogram_at(unique[0], unique[1]);
    const int64_t n = args.n_read;
    const int64_t tiles = (n + 63) / 64;
    const int64_t tile_step = static_cast<int64_t>(gridDim.x) * WAVES;
    // the lookup tables hold a word's ORDINAL in its segment, one byte an entry (1 MB a segment of 10 nt: both stay in L2)
    const uint8_t* const nearest[SEGMENTS] = {static_cast<const uint8_t*>(d.cross_stream_table[0]), static_cast<const uint8_t*>(d.cross_stream_table[1])};
    const uint32_t blocks_at[SEGMENTS] = {Layout::block_at(unique[0], unique[1], 0), Layout::block_at(unique[0], unique[1], 1)};
    const uint32_t rings_at = Layout::RING_AT + wave * Layout::RING_BYTES;      // this wave's three rings: by what pass C has to scan
    const int64_t own_segment = static_cast<int64_t>(blockIdx.x) * WAVES + wave;
    uint32_t* const segment = args.worklist + own_segment * args.worklist_stride;
    uint32_t segment_count = 0;
    uint32_t ring_count[Layout::RINGS] = {0, 0, 0};

    const int high_quality_threshold = min(max(d.high_quality_threshold, 0), 128);
    const uint32_t high_quality_addend = static_cast<uint32_t>(128 - high_quality_threshold) * 0x01010101u;
    const float rho_at_threshold = high_quality_threshold <= 0 ? __builtin_inff() : high_quality_threshold >= 128 ? -1.0f : factor_single[high_quality_threshold].y;
    const bool zero_is_high = high_quality_threshold <= 0;
    const float noise_over_prior = static_cast<float>(d.adjusted_noise_probability / d.concentration[0]);       // uniform priors
    const float noise_floor = static_cast<float>(d.random_barcode_probability) * near::SLOP;
    const double confidence_threshold = d.confidence_threshold;
    const float y_threshold = static_cast<float>(1.0 - confidence_threshold);
    const float y_guard = static_cast<float>((2e-8 + fabs(static_cast<double>(y_threshold) - (1.0 - confidence_threshold))) * 1.001 + 1e-12);
    const double common_prior = d.concentration[0];
    uint32_t outgoing_pass = 0;
    uint32_t outgoing_fail = 0;
    const int hq_threshold = d.high_quality_distance_threshold;
    const bool multiplexing = d.multiplexing != 0;
    const bool last_stage = args.last_stage != 0;

    struct Inputs { uint32_t code[ROW_WORDS]; uint32_t quality[ROW_WORDS]; uint32_t state; };
    auto fetch = [&](int64_t tile, Inputs& in) {
        const int64_t first = tile * 64;
        const uint32_t last = static_cast<uint32_t>(min(static_cast<int64_t>(64), n - first)) - 1u;
        const uint32_t at = min(lane, last);
        const uint32_t* const code_row = reinterpret_cast<const uint32_t*>(args.base + first * (4 * ROW_WORDS) + at * (4u * ROW_WORDS));
        const uint32_t* const quality_row = reinterpret_cast<const uint32_t*>(args.quality + first * (4 * ROW_WORDS) + at * (4u * ROW_WORDS));
#pragma unroll
        for(int w = 0; w < ROW_WORDS; ++w) { in.code[w] = code_row[w]; in.quality[w] = quality_row[w]; }
        if(STATE == 1) { in.state = static_cast<uint32_t>(args.input_flag[first + at]); }      // transcode.h:207-211
        else if(STATE == 2) { in.state = args.word[first + at]; }
        else { in.state = 0; }
    };
    auto gather = [](uint32_t planes) {            // bits 8 j and 8 j + 4 -> bits j and 4 + j
        const uint32_t t = planes | (planes >> 7);
        return (t | (t >> 14)) & 0xffu;
    };
    // the words of segment s (it starts at position s L of the row), funnel shifted to its first byte
    auto segment_words = [&](const uint32_t (&row)[ROW_WORDS], int s, uint32_t (&out)[G]) {
        const int first_word = (s * L) >> 2;
        const int byte_shift = (s * L) & 3;
#pragma unroll
        for(int w = 0; w < G; ++w) {
            const int a = first_word + w < ROW_WORDS ? first_word + w : ROW_WORDS - 1;
            const int b = first_word + w + 1 < ROW_WORDS ? first_word + w + 1 : ROW_WORDS - 1;
            out[w] = (byte_shift ? __builtin_amdgcn_alignbyte(row[b], row[a], byte_shift) : row[a]) & Layout::valid(w);
        }
    };
    auto entry = [&](uint32_t fields, int byte, int g) -> float {
        const uint32_t address = __builtin_amdgcn_perm(fields, lane_low, 0x0c020400u | (static_cast<uint32_t>(byte) << 8));
        return ((LdsFloats)static_cast<uintptr_t>(address))[g * 64];
    };
    // the normalised tables of a segment from its rho row (pamld_stream.hpp); ge for the bound
    auto build_tables = [&](const float (&rho)[RHO], float (&ge)[3][5]) {
#pragma unroll
        for(int g = 0; g < 2; ++g) {
            const int positions = Layout::positions(g);
            const int entries = 1 << positions;
            const float r0 = rho[4 * g], r1 = positions > 1 ? rho[4 * g + 1] : 1.0f, r2 = positions > 2 ? rho[4 * g + 2] : 1.0f, r3 = positions > 3 ? rho[4 * g + 3] : 1.0f;
            const float lo[4] = {1.0f, r0, r1, r0 * r1};
            const float hi[4] = {1.0f, r2, r3, r2 * r3};
            float t[16];
            float by_count[5] = {1.0f, 0, 0, 0, 0};
#pragma unroll
            for(int m = 1; m < 16; ++m) {
                if(m < entries) {
                    t[m] = (m >> 2) == 0 ? lo[m & 3] : (m & 3) == 0 ? hi[m >> 2] : lo[m & 3] * hi[m >> 2];
                    *reinterpret_cast<float*>(lds_raw + lane_const + static_cast<uint32_t>(g) * 256u + static_cast<uint32_t>(m) * 1024u) = t[m];
                    const int a = __builtin_popcount(static_cast<unsigned>(m));
                    by_count[a] = (m == (1 << a) - 1) ? t[m] : fmaxf(by_count[a], t[m]);
                }
            }
#pragma unroll
            for(int a = 4; a >= 0; --a) {
                ge[g][a] = a > positions ? 0.0f : a == positions ? by_count[a] : fmaxf(by_count[a], ge[g][a + 1 > 4 ? 4 : a + 1]);
            }
        }
        ge[2][4] = 0.0f; ge[2][3] = 0.0f;
        if(TAIL == 0) { ge[2][0] = 1.0f; ge[2][1] = 0.0f; ge[2][2] = 0.0f; }
        else if(TAIL == 1) { ge[2][2] = 0.0f; ge[2][1] = rho[8]; ge[2][0] = fmaxf(1.0f, rho[8]); }
        else {
            ge[2][2] = rho[8] * rho[9];
            ge[2][1] = fmaxf(fmaxf(rho[8], rho[9]), ge[2][2]);
            ge[2][0] = fmaxf(1.0f, ge[2][1]);
        }
    };
    // a position past the tables: rho where its flag bit is set, else 1 (one v_bfe_i32, one v_bitop3_b32)
    auto tail_factor = [&](uint32_t word, const float (&rho)[RHO]) {
        float factor = 1.0f;
#pragma unroll
        for(int t = 0; t < TAIL; ++t) {
            const uint32_t mask = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(word), 18 + t, 1));
            const float picked = __uint_as_float(__builtin_amdgcn_bitop3_b32(mask, __float_as_uint(rho[8 + t]), 0x3f800000u, 0xCA));
            factor = t == 0 ? picked : factor * picked;
        }
        return factor;
    };

    // pass A of ONE segment (pamld_stream.hpp): the word named by the lookup and the words listed with it scored through the
    // per read tables, every other word bounded by four distance classes — x_s = (listed + rest) / winner, in units of the
    // segment's all-match product
    typedef float Weights __attribute__((ext_vector_type(4)));
    struct Score { float winner, matches, x_listed, x_rest, x_weighted; uint32_t k, high_mismatch, ordinal, mismatches; bool known; };
    auto score_segment = [&](const uint32_t (&q)[G], uint32_t sliced, uint32_t looked_up, uint32_t first_block, Score& out, const Weights* ratio = nullptr) {
        const uint32_t ordinal = looked_up;                                    // the word's ordinal in its segment
        const bool known = ordinal != CROSS_STREAM_UNKNOWN;
        const uint32_t block = first_block + (known ? ordinal : 0u) * Layout::BLOCK_BYTES;
        Quad record[Layout::BLOCK_QUADS];
#pragma unroll
        for(int quad = 0; quad < Layout::BLOCK_QUADS; ++quad) { record[quad] = *(LdsQuads)static_cast<uintptr_t>(block + 16u * quad); }
        auto record_word = [&](int w) { return record[w >> 2][w & 3]; };
        const uint32_t observed = sliced;
        const uint32_t observed_high = lut::high_plane(observed);
        const uint32_t observed_low = lut::low_plane(observed) | wave_bytes;
        float rho[RHO];
        float everything_matches = 1.0f;
#pragma unroll
        for(int j = 0; j < RHO; ++j) {
            if(j < L) {
                const uint32_t word = q[j >> 2] & 0x7f7f7f7fu;
                const int byte = j & 3;
                const uint32_t shifted = byte == 0 ? (word << 3) : (word >> (8 * byte - 3));
                const Pair pair = *(LdsPairs)static_cast<uintptr_t>(and_or(shifted, 0x3f8u, Layout::FACTOR_AT));
                everything_matches = j == 0 ? pair.x : everything_matches * pair.x;
                rho[j] = pair.y;
            } else { rho[j] = 1.0f; }
        }
        float ge[3][5];
        build_tables(rho, ge);
        uint32_t fields[LISTED];
#pragma unroll
        for(int o = 0; o < LISTED; ++o) { fields[o] = (observed_low ^ record_word(o)) | (observed_high ^ record_word(LISTED + o)); }
        float first_group[LISTED];
        float second_group[LISTED];
#pragma unroll
        for(int o = 0; o < LISTED; ++o) {
            first_group[o] = entry(fields[o], 0, 0);
            second_group[o] = entry(fields[o], 1, 1);
        }
        // H[n]: the largest product over n or more mismatching positions, n = 3 .. 6 (pamld_stream.hpp)
        float h[7];
        {
            float tabled[7];
#pragma unroll
            for(int m = 1; m <= 6; ++m) {
                tabled[m] = 0.0f;
                bool any = false;
#pragma unroll
                for(int a = 0; a <= 4; ++a) {
                    const int o = m - a;
                    if(o >= 0 && o <= 4 && a <= Layout::positions(0) && o <= Layout::positions(1) && (m >= 3 || TAIL)) {
                        const float product = ge[0][a] * ge[1][o];
                        tabled[m] = any ? fmaxf(tabled[m], product) : product;
                        any = true;
                    }
                }
            }
#pragma unroll
            for(int m = 3; m <= 6; ++m) {
                h[m] = tabled[m];
#pragma unroll
                for(int t = 1; t <= TAIL; ++t) { h[m] = fmaxf(h[m], tabled[m - t] * ge[2][t]); }
            }
        }
        const uint32_t mismatches = fields[0] & FULL_MASK;
        float winner = first_group[0] * second_group[0];
        if(TAIL) { winner *= tail_factor(fields[0], rho); }
        float others = 0.0f;
        float weighted = 0.0f;                 // PRIORS: the listed words by their prior over the combination's
#pragma unroll
        for(int o = 1; o < LISTED; ++o) {
            float p = first_group[o] * second_group[o];
            if(TAIL) { p *= tail_factor(fields[o], rho); }
            others = o == 1 ? p : others + p;
            if(PRIORS && ratio != nullptr) { weighted = __builtin_fmaf(ratio[o >> 2][o & 3], p, weighted); }
        }
        const float inverse_winner = __builtin_amdgcn_rcpf(winner);
        float rest = __uint_as_float(record_word(2 * LISTED)) * h[3];
        rest = __builtin_fmaf(__uint_as_float(record_word(2 * LISTED + 1)), h[4], rest);
        rest = __builtin_fmaf(__uint_as_float(record_word(2 * LISTED + 2)), h[5], rest);
        rest = __builtin_fmaf(__uint_as_float(record_word(2 * LISTED + 3)), h[6], rest);
        rest *= (fmaxf(winner, 1.0f) * near::SLOP) * inverse_winner;
        const uint32_t k = static_cast<uint32_t>(__popc(mismatches));
        const bool high_mismatch = k == 1 && winner <= rho_at_threshold && (zero_is_high || winner != 1.0f);
        out.winner = winner;
        out.matches = everything_matches;
        out.x_listed = others * inverse_winner;
        out.x_rest = rest * inverse_winner;
        out.x_weighted = weighted * inverse_winner;
        out.k = k;
        out.high_mismatch = high_mismatch ? 1u : 0u;
        out.ordinal = known ? ordinal : 0u;
        out.mismatches = mismatches;
        out.known = known;
    };
    // a segment pass A leaves alone: the word is known, at most one mismatch away, and x_s is small and tightly bounded (half of
    // pass A's allowance for the bounded part: the other half is the scanned segment's)
    auto segment_good = [&](const Score& score, bool strict) {
        return strict && score.known && score.k <= 1 && score.x_listed <= near::LISTED_MAX && score.x_rest <= 0.5f * near::REST_MAX;
    };

    // ---- pass C: the factorised scan (pamld_cross.hpp) over the wave's ring ------------------------------------------------
    // mode (wave uniform): 0 / 1 = only that segment is scanned, the other one is scored again as pass A scored it (it was
    // good: a lookup, the listed words and the bound instead of a scan over every word); 2 = both are scanned
    auto drain = [&](int mode, uint32_t count) {
        __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): pass A's placeholder stores of these reads have left
        const bool live = lane < count;
        const uint32_t slot = live ? lane : 0u;
        const uint32_t ring_at = rings_at + static_cast<uint32_t>(mode) * (Layout::RING_PLANES * RING * 4u) + slot * 4u;      // an LDS address: no generic pointer
        const uint32_t state = *(LdsWords)static_cast<uintptr_t>(ring_at);
        const int64_t i = static_cast<int64_t>(*(LdsWords)static_cast<uintptr_t>(ring_at + RING * 4u));
        uint32_t code_row[ROW_WORDS];
        uint32_t quality_row[ROW_WORDS];
        {
            const uint32_t* const code_at = reinterpret_cast<const uint32_t*>(args.base + i * (4 * ROW_WORDS));
            const uint32_t* const quality_at = reinterpret_cast<const uint32_t*>(args.quality + i * (4 * ROW_WORDS));
#pragma unroll
            for(int w = 0; w < ROW_WORDS; ++w) { code_row[w] = code_at[w]; quality_row[w] = quality_at[w]; }
        }
        ReadState r;
        if(STATE != 2) { r.qcfail = (state & 1u) != 0; r.channel = 0; }
        else { r.qcfail = (state >> 31) != 0; r.channel = static_cast<int32_t>(state & 0x7fffffffu); }

        double best_conditional = 1.0;         // the winner's P(r|b), fp64
        double everything_matches = 1.0;       // the all-match product of both segments, fp64
        double winner_product = 1.0;           // the winner over the all-match product, from the single precision scan
        double others = 0.0;                   // every other combination over the all-match product
        int flat = 0;
        int distance = 0;
        int hq_distance = 0;
        bool certain = true;
#pragma unroll
        for(int s = 0; s < SEGMENTS; ++s) {
            uint32_t code_words[G];
            uint32_t quality_words[G];
            segment_words(code_row, s, code_words);
            segment_words(quality_row, s, quality_words);
            uint32_t row[G];
            uint32_t observed = 0;
            uint32_t high_quality = 0;
            uint32_t outside = 0;
#pragma unroll
            for(int w = 0; w < G; ++w) {
                const uint32_t valid = Layout::valid(w);
                const uint32_t x = code_words[w] & 0x0f0f0f0fu;
                const uint32_t q8 = quality_words[w];
                const uint32_t q7 = q8 & 0x7f7f7f7fu;
                const uint32_t z = x & ((x | 0x80808080u) - 0x01010101u);
                const uint32_t strict = (x + 0x7f7f7f7fu) & ~(z + 0x7f7f7f7fu) & 0x80808080u;
                const uint32_t strict_bytes = (strict - (strict >> 7)) | strict;
                const uint32_t nonzero = (q7 + 0x7f7f7f7fu) & 0x80808080u;
                row[w] = (q7 & strict_bytes) | (nonzero & ~strict_bytes);
                const uint32_t codes = gather((((x >> 1) | (x >> 3)) & 0x01010101u) | (((x << 2) | (x << 1)) & 0x10101010u));
                observed |= codes << (8 * w + 2);
                const uint32_t high = ((q7 + high_quality_addend) | q8) & 0x80808080u & valid;
                const uint32_t flags = gather(((~strict & 0x80808080u & valid) >> 7) | (high >> 3));
                outside |= (flags & 0xfu) << (8 * w + 2);
                high_quality |= (flags >> 4) << (8 * w + 2);
            }
            float top = 0;
            float rest = 0;
            float second = 0;
            int best = -1;
            uint32_t flags_of_best = 0;
            if(mode == 2 || mode == s) {
                float rho[RHO];
    #pragma unroll
                for(int j = 0; j < RHO; ++j) {
                    if(j < L) { rho[j] = factor_single[(row[j >> 2] >> (8 * (j & 3))) & 0xffu].y; }
                    else { rho[j] = 1.0f; }
                }
                const uint32_t observed_high = lut::high_plane(observed);
                const uint32_t observed_low = lut::low_plane(observed) | wave_bytes;
                float unused[3][5];
                build_tables(rho, unused);

                // the segment's words eight at a time (pamld_lut.hpp): per block its maximum and its sum; across blocks the running
                // maximum, which block holds it, that block's sum set aside, the sum of every other block
                float others_sum = 0;
                float held_sum = 0;
                float best_value = 0;
                int best_block = -1;
                auto block_values = [&](const uint32_t (&w)[16], float (&p)[8]) {
    #pragma unroll
                    for(int k = 0; k < 8; ++k) {
                        const uint32_t fields = (observed_low ^ w[k]) | (observed_high ^ w[8 + k]);
                        p[k] = entry(fields, 0, 0) * entry(fields, 1, 1);
                        if(TAIL) { p[k] *= tail_factor(fields, rho); }
                    }
                };
                const uint32_t words_at = s == 0 ? Layout::WORDS_AT : Layout::WORDS_AT + Layout::words_bytes(unique[0]);      // LDS addresses: no generic pointers
                const LdsQuads blocks = (LdsQuads)static_cast<uintptr_t>(words_at);
                const LdsWords words = (LdsWords)static_cast<uintptr_t>(words_at);
                const int n_blocks = (unique[s] + 7) >> 3;
    #pragma unroll 2
                for(int f = 0; f < n_blocks; ++f) {
                    uint32_t w[16];
    #pragma unroll
                    for(int quad = 0; quad < 4; ++quad) {
                        const Quad four = blocks[4 * f + quad];
                        w[4 * quad] = four.x; w[4 * quad + 1] = four.y; w[4 * quad + 2] = four.z; w[4 * quad + 3] = four.w;
                    }
                    float p[8];
                    block_values(w, p);
                    if(8 * f + 8 > unique[s]) {
    #pragma unroll
                        for(int k = 1; k < 8; ++k) { if(8 * f + k >= unique[s]) { p[k] = 0.0f; } }
                    }
                    const float block_max = fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7])));
                    const float block_sum = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
                    const bool takes_over = block_max > best_value;                      // strict: the first block keeps a tie
                    second = __builtin_amdgcn_fmed3f(best_value, block_max, second);
                    others_sum += takes_over ? held_sum : block_sum;
                    held_sum = takes_over ? block_sum : held_sum;
                    best_block = takes_over ? f : best_block;
                    best_value = takes_over ? block_max : best_value;
                }
                // inside the winning block: the winner (first of equals) and the sum of the other seven, recomputed
                top = best_value;
                rest = others_sum;
                {
                    const int block = best_block < 0 ? 0 : best_block;
                    uint32_t w[16];
    #pragma unroll
                    for(int k = 0; k < 16; ++k) { w[k] = words[16 * block + k]; }
                    float p[8];
                    block_values(w, p);
                    {
                        const int limit = unique[s] - 8 * block;
    #pragma unroll
                        for(int k = 1; k < 8; ++k) { p[k] = k < limit ? p[k] : 0.0f; }
                    }
                    int within = -1;
                    float q[8];
    #pragma unroll
                    for(int k = 7; k >= 0; --k) {
                        const bool is_top = p[k] == best_value;
                        within = is_top ? k : within;
                        q[k] = is_top ? 0.0f : p[k];
                    }
                    const float inner_rest = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
                    const float inner_second = fmaxf(fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3])), fmaxf(fmaxf(q[4], q[5]), fmaxf(q[6], q[7])));
                    const bool crowded = held_sum - inner_rest >= 1.5f * best_value;         // the block's sum holds the maximum twice
                    second = fmaxf(second, crowded ? best_value : inner_second);
                    rest += crowded ? best_value : inner_rest;
                    if(best_block >= 0 && best_value > 0.0f && within >= 0) { best = 8 * block + within; }
                    const int at = 16 * (best < 0 ? 0 : best / 8) + (best < 0 ? 0 : best % 8);
                    flags_of_best = (observed_low ^ words[at]) | (observed_high ^ words[at + 8]);
                }
            } else {
                // the segment pass A found good: the same lookup, listed words and bound again
                uint32_t key = 0;
                bool strict = true;
#pragma unroll
                for(int w = 0; w < G; ++w) {
                    const uint32_t codes = (observed >> (8 * w + 2)) & 0xffu;
                    if(w < 2) { key |= codes << (8 * w); }
                    else { key |= ((codes & ((1u << TAIL) - 1u)) << 16) | (((codes >> 4) & ((1u << TAIL) - 1u)) << (16 + TAIL)); }
                    strict = strict && (quality_words[w] & 0x80808080u) == 0;
                }
                strict = strict && outside == 0;
                Score score;
                score_segment(quality_words, observed, static_cast<uint32_t>(nearest[s][key]), blocks_at[s], score);
                top = score.winner;
                rest = score.x_listed * score.winner;
                second = rest;
                best = static_cast<int>(score.ordinal);
                flags_of_best = score.mismatches;
                if(!segment_good(score, strict)) { certain = false; }
            }
            if(best < 0) { certain = false; best = 0; }       // underflow: everything rounded to zero
            certain = certain && top >= cross::SMALLEST_CERTAIN && (second <= cross::RUNNER_UP * top);
            // the winner's mismatch flags, its distances, its probability and the all-match product in fp64
            const uint32_t flags = (flags_of_best | outside) & FULL_MASK;
            distance += __popc(flags);
            hq_distance += __popc(flags & high_quality);
#pragma unroll
            for(int j = 0; j < L; ++j) {
                const uint32_t k = (row[j >> 2] >> (8 * (j & 3))) & 0xffu;
                const double match = factor_double[k][0];
                const double mismatch = factor_double[k][1];
                best_conditional *= ((flags >> low_bit(j)) & 1u) ? mismatch : match;
                everything_matches *= match;
            }
            // (winner + others) * (top + rest) - winner * top, every term positive (pamld_cross.hpp)
            others = others * (static_cast<double>(top) + static_cast<double>(rest)) + winner_product * static_cast<double>(rest);
            winner_product *= static_cast<double>(top);
            flat += best * radix[s];
        }

        bool defer = !certain;
        {
            const int best = flat;                                                        // cross_ordinal is the identity (host checked)
            const double best_p = best_conditional * common_prior;
            const double others_p = (others * everything_matches) * common_prior;
            const double total = (best_p + others_p) + d.adjusted_noise_probability;          // pamld.cpp:61-86
            double confidence = best_p / total;                                               // pamld.cpp:92
            const double noise_margin = best_conditional - d.random_barcode_probability;
            const double confidence_margin = confidence - confidence_threshold;
            const bool noise_clear = fabs(noise_margin) > 1e-9 * d.random_barcode_probability;
            const bool is_noise = noise_margin <= 0;
            if(!noise_clear) { defer = true; }
            if(!is_noise) {
                // the posterior matters: a dominant winner and no decision within 1e-9 of the threshold (pamld_cross.hpp)
                if(!(others * everything_matches <= cross::DOMINANCE * best_conditional) || !(fabs(confidence_margin) > 1e-9)) { defer = true; }
            }
            if(!defer && live) {
                // the counters of a codec this size: count, distance and confidence of a classified read are rebuilt from its
                // record (counters.hip); the two rare counts go straight to the counter buffer, bin 0 to the workgroup's copy
                int bin = best + 1;
                uint8_t status = STATUS_PASS;
                unsigned long long* const bin_words = args.stats + d.stats_offset + static_cast<int64_t>(bin) * STAT_WORDS_PER_BIN;
                if(!is_noise) {                                                       // pamld.cpp:96
                    if(confidence_margin > 0) {                                       // pamld.cpp:99
                        if(hq_threshold > 0 && hq_distance >= hq_threshold) { r.qcfail = true; }
                    } else {
                        atomicAdd(&bin_words[FIELD_LOW_CONFIDENCE], 1ull);
                        r.qcfail = true;
                        status = STATUS_LOW_CONFIDENCE;
                    }
                } else {
                    atomicAdd(&bin_words[FIELD_LOW_CONDITIONAL_CONFIDENCE], 1ull);
                    r.qcfail = true;
                    bin = 0;
                    distance = 0;
                    hq_distance = 0;
                    confidence = 0;
                    status = STATUS_NOISE;
                    // Decoder<T>::classify then Classifier<T>::classify of the unclassified bin (decoder.h:68-76, classifier.h:78-86)
                    __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned long long*)static_cast<uintptr_t>(histogram_at + 8u * (STAT_JOB_WORDS + FIELD_COUNT)), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if(multiplexing) { r.channel = bin > 0 ? d.barcode_index[bin - 1] : 0; }
                DevResult out;
                out.index = bin > 0 ? d.barcode_index[bin - 1] : 0;
                out.distance = static_cast<uint8_t>(distance > 255 ? 255 : distance);
                out.hq_distance = static_cast<uint8_t>(hq_distance > 255 ? 255 : hq_distance);
                out.status = status;
                out.qcfail = r.qcfail ? 1 : 0;
                out.confidence = confidence;
                stream_result(&args.result[i], out);
                stream_read_state(args, i, r);
            }
        }
        const unsigned long long handed_on = __ballot(live && defer);
        if(handed_on) {
            if(live && defer) { segment[segment_count + __popcll(handed_on & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(i); }
            segment_count += static_cast<uint32_t>(__popcll(handed_on));
        }
        if(last_stage) {
            outgoing_pass += static_cast<uint32_t>(__popcll(__ballot(live && !defer && !r.qcfail)));
            outgoing_fail += static_cast<uint32_t>(__popcll(__ballot(live && !defer && r.qcfail)));
        }
    };

    // ---- pass C with priors that differ: a read ONE of whose segments pass A found good ------------------------------------
    // mode = the segment that is scanned (0 or 1); the other one, g, named its word c_g by the lookup.  Every combination through
    // c_g is exact: the scan weighs the words of the scanned segment by prior(c_g, .), a row of the prior matrix (or of its
    // transpose) from L2.  Through the scan's winner w the listed words of segment g come by their prior ratios for the
    // combination (c_g, w); what the lists leave out and every combination through neither is bounded at the largest prior over
    // the winner's — pamld_cross.hpp's weighted certificate: that bound at most 4e-9 of the winner and taken at its midpoint,
    // everything else at most 1.5e-3 of it, the argmax certain, no decision within 1e-8 of its threshold.  A read neither of
    // whose segments is good never comes here: the tile loop hands it to the worklist for the three scan kernel.
    auto drain_weighted = [&](int mode, uint32_t count) {
        __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): pass A's placeholder stores of these reads have left
        const bool live = lane < count;
        const uint32_t slot = live ? lane : 0u;
        const uint32_t ring_at = rings_at + static_cast<uint32_t>(mode) * (Layout::RING_PLANES * RING * 4u) + slot * 4u;
        const uint32_t state = *(LdsWords)static_cast<uintptr_t>(ring_at);
        const int64_t i = static_cast<int64_t>(*(LdsWords)static_cast<uintptr_t>(ring_at + RING * 4u));
        uint32_t code_row[ROW_WORDS];
        uint32_t quality_row[ROW_WORDS];
        {
            const uint32_t* const code_at = reinterpret_cast<const uint32_t*>(args.base + i * (4 * ROW_WORDS));
            const uint32_t* const quality_at = reinterpret_cast<const uint32_t*>(args.quality + i * (4 * ROW_WORDS));
#pragma unroll
            for(int w = 0; w < ROW_WORDS; ++w) { code_row[w] = code_at[w]; quality_row[w] = quality_at[w]; }
        }
        ReadState r;
        if(STATE != 2) { r.qcfail = (state & 1u) != 0; r.channel = 0; }
        else { r.qcfail = (state >> 31) != 0; r.channel = static_cast<int32_t>(state & 0x7fffffffu); }

        // both segments: factor rows, bit planes, flags (as the uniform pass C)
        uint32_t row_of[SEGMENTS][G], observed_of[SEGMENTS], outside_of[SEGMENTS], high_quality_of[SEGMENTS], quality_words_of[SEGMENTS][G];
        bool strict_of[SEGMENTS];
#pragma unroll
        for(int s = 0; s < SEGMENTS; ++s) {
            uint32_t code_words[G];
            segment_words(code_row, s, code_words);
            segment_words(quality_row, s, quality_words_of[s]);
            observed_of[s] = 0; outside_of[s] = 0; high_quality_of[s] = 0;
            bool strict = true;
#pragma unroll
            for(int w = 0; w < G; ++w) {
                const uint32_t valid = Layout::valid(w);
                const uint32_t x = code_words[w] & 0x0f0f0f0fu;
                const uint32_t q8 = quality_words_of[s][w];
                const uint32_t q7 = q8 & 0x7f7f7f7fu;
                const uint32_t z = x & ((x | 0x80808080u) - 0x01010101u);
                const uint32_t strict_bits = (x + 0x7f7f7f7fu) & ~(z + 0x7f7f7f7fu) & 0x80808080u;
                const uint32_t strict_bytes = (strict_bits - (strict_bits >> 7)) | strict_bits;
                const uint32_t nonzero = (q7 + 0x7f7f7f7fu) & 0x80808080u;
                row_of[s][w] = (q7 & strict_bytes) | (nonzero & ~strict_bytes);
                const uint32_t codes = gather((((x >> 1) | (x >> 3)) & 0x01010101u) | (((x << 2) | (x << 1)) & 0x10101010u));
                observed_of[s] |= codes << (8 * w + 2);
                const uint32_t high = ((q7 + high_quality_addend) | q8) & 0x80808080u & valid;
                const uint32_t flags = gather(((~strict_bits & 0x80808080u & valid) >> 7) | (high >> 3));
                outside_of[s] |= (flags & 0xfu) << (8 * w + 2);
                high_quality_of[s] |= (flags >> 4) << (8 * w + 2);
                strict = strict && (q8 & 0x80808080u) == 0;
            }
            strict_of[s] = strict && outside_of[s] == 0;
        }
        auto key_of = [&](int s) {
            uint32_t key = 0;
#pragma unroll
            for(int w = 0; w < G; ++w) {
                const uint32_t codes = (observed_of[s] >> (8 * w + 2)) & 0xffu;
                if(w < 2) { key |= codes << (8 * w); }
                else { key |= ((codes & ((1u << TAIL) - 1u)) << 16) | (((codes >> 4) & ((1u << TAIL) - 1u)) << (16 + TAIL)); }
            }
            return key;
        };
        // the good segment's word, by its lookup alone: the row of the prior matrix the scan runs along
        const int scanned = mode, good = 1 - mode;
        const uint32_t looked_up_good = static_cast<uint32_t>(nearest[good][key_of(good)]);
        const uint32_t word_good = looked_up_good != CROSS_STREAM_UNKNOWN ? looked_up_good : 0u;
        const float* const weights = (good == 0 ? d.cross_prior : d.cross_prior_transposed) + static_cast<size_t>(word_good) * CROSS_MAX_UNIQUE;
        // the scan of the other segment, weighted: the block's maximum and sum, across blocks the running maximum and so on
        // (pamld_cross.hpp), and the plain sum and the plain value of the weighted winner for the bound
        float top = 0, rest = 0, second = 0, plain_sum = 0, top_plain = 0;
        int best = -1;
        uint32_t flags_of_best = 0;
        {
            float rho[RHO];
#pragma unroll
            for(int j = 0; j < RHO; ++j) {
                if(j < L) { rho[j] = factor_single[(row_of[scanned][j >> 2] >> (8 * (j & 3))) & 0xffu].y; }
                else { rho[j] = 1.0f; }
            }
            const uint32_t observed_high = lut::high_plane(observed_of[scanned]);
            const uint32_t observed_low = lut::low_plane(observed_of[scanned]) | wave_bytes;
            float unused[3][5];
            build_tables(rho, unused);
            const uint32_t words_at = scanned == 0 ? Layout::WORDS_AT : Layout::WORDS_AT + Layout::words_bytes(unique[0]);
            const LdsQuads blocks = (LdsQuads)static_cast<uintptr_t>(words_at);
            const LdsWords words = (LdsWords)static_cast<uintptr_t>(words_at);
            const int limit = scanned == 0 ? unique[0] : unique[1];
            const int n_blocks = (limit + 7) >> 3;
            auto block_values = [&](int f, float (&p)[8], float (&plain)[8]) {
                uint32_t w[16];
#pragma unroll
                for(int quad = 0; quad < 4; ++quad) {
                    const Quad four = blocks[4 * f + quad];
                    w[4 * quad] = four.x; w[4 * quad + 1] = four.y; w[4 * quad + 2] = four.z; w[4 * quad + 3] = four.w;
                }
                const Weights low_weights = *reinterpret_cast<const Weights*>(weights + 8 * f);
                const Weights high_weights = *reinterpret_cast<const Weights*>(weights + 8 * f + 4);
#pragma unroll
                for(int k = 0; k < 8; ++k) {
                    const uint32_t fields = (observed_low ^ w[k]) | (observed_high ^ w[8 + k]);
                    plain[k] = entry(fields, 0, 0) * entry(fields, 1, 1);
                    if(TAIL) { plain[k] *= tail_factor(fields, rho); }
                    if(8 * f + k >= limit) { plain[k] = 0.0f; }
                    p[k] = plain[k] * (k < 4 ? low_weights[k] : high_weights[k - 4]);       // the prior matrix is zero padded past the segment's words
                }
            };
            float others_sum = 0, held_sum = 0, best_value = 0;
            int best_block = -1;
#pragma unroll 1
            for(int f = 0; f < n_blocks; ++f) {
                float p[8], plain[8];
                block_values(f, p, plain);
                plain_sum += ((plain[0] + plain[1]) + (plain[2] + plain[3])) + ((plain[4] + plain[5]) + (plain[6] + plain[7]));
                const float block_max = fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7])));
                const float block_sum = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
                const bool takes_over = block_max > best_value;                      // strict: the first block keeps a tie
                second = __builtin_amdgcn_fmed3f(best_value, block_max, second);
                others_sum += takes_over ? held_sum : block_sum;
                held_sum = takes_over ? block_sum : held_sum;
                best_block = takes_over ? f : best_block;
                best_value = takes_over ? block_max : best_value;
            }
            top = best_value;
            rest = others_sum;
            {
                const int block = best_block < 0 ? 0 : best_block;
                float p[8], plain[8];
                block_values(block, p, plain);
                int within = -1;
                float q[8];
#pragma unroll
                for(int k = 7; k >= 0; --k) {
                    const bool is_top = p[k] == best_value;
                    within = is_top ? k : within;
                    top_plain = is_top ? plain[k] : top_plain;
                    q[k] = is_top ? 0.0f : p[k];
                }
                const float inner_rest = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
                const float inner_second = fmaxf(fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3])), fmaxf(fmaxf(q[4], q[5]), fmaxf(q[6], q[7])));
                const bool crowded = held_sum - inner_rest >= 1.5f * best_value;         // the block's sum holds the maximum twice
                second = fmaxf(second, crowded ? best_value : inner_second);
                rest += crowded ? best_value : inner_rest;
                if(best_block >= 0 && best_value > 0.0f && within >= 0) { best = 8 * block + within; }
                const int at = 16 * (best < 0 ? 0 : best / 8) + (best < 0 ? 0 : best % 8);
                flags_of_best = (observed_low ^ words[at]) | (observed_high ^ words[at + 8]);
            }
        }
        bool certain = best >= 0 && top >= cross::SMALLEST_CERTAIN && second <= cross::RUNNER_UP * top && top_plain >= cross::SMALLEST_CERTAIN;
        const uint32_t word_scanned = best < 0 ? 0u : static_cast<uint32_t>(best);
        // the good segment again, its listed words by the prior ratios of the combination the scan has named
        const uint32_t ordinal_of[SEGMENTS] = {good == 0 ? word_good : word_scanned, good == 0 ? word_scanned : word_good};
        const size_t combination = static_cast<size_t>(ordinal_of[0]) * static_cast<size_t>(radix[0]) + static_cast<size_t>(ordinal_of[1]) * static_cast<size_t>(radix[1]);
        Weights ratio[LISTED / 4];
        {
            const Weights* const from = reinterpret_cast<const Weights*>(d.cross_near_weight[good == 0 ? 1 : 0]) + combination * (CROSS_NEAR_WEIGHTS / 4);
#pragma unroll
            for(int quad = 0; quad < LISTED / 4; ++quad) { ratio[quad] = from[quad]; }
        }
        const float largest_over_own = reinterpret_cast<const float*>(d.cross_near_weight[1])[combination * CROSS_NEAR_WEIGHTS];      // its slot 0: the largest prior over the combination's
        Score score;
        score_segment(quality_words_of[good], observed_of[good], looked_up_good, blocks_at[good], score, ratio);
        if(!segment_good(score, strict_of[good]) || !(score.x_weighted <= near::LISTED_MAX)) { certain = false; }
        // everything in units of the winner prior(c) P_good(c_good) P_scanned(w)
        const float inverse_top = __builtin_amdgcn_rcpf(top);
        const float through_good = rest * inverse_top;                                            // exact: the weighted scan
        const float through_scanned = score.x_weighted;                                           // the listed words of the good segment
        const float plain_others = fmaxf(plain_sum - top_plain, 0.0f) * __builtin_amdgcn_rcpf(top_plain) * (1.0f + 1e-5f) + 2e-6f * plain_sum * __builtin_amdgcn_rcpf(top_plain);
        const float bounded = largest_over_own * (score.x_rest + (score.x_listed + score.x_rest) * plain_others) * near::SLOP;
        const float others_single = through_good + through_scanned + 0.5f * bounded;             // the bounded part at its midpoint

        // the winner again in fp64 over both segments, its prior in fp64
        double best_conditional = 1.0;
        int distance = 0;
        int hq_distance = 0;
#pragma unroll
        for(int s = 0; s < SEGMENTS; ++s) {
            const uint32_t flags = ((s == scanned ? flags_of_best : score.mismatches) | outside_of[s]) & FULL_MASK;
            distance += __popc(flags);
            hq_distance += __popc(flags & high_quality_of[s]);
#pragma unroll
            for(int j = 0; j < L; ++j) {
                const uint32_t k = (row_of[s][j >> 2] >> (8 * (j & 3))) & 0xffu;
                best_conditional *= ((flags >> low_bit(j)) & 1u) ? factor_double[k][1] : factor_double[k][0];
            }
        }
        bool defer = !certain;
        {
            const double prior = d.cross_prior_double[combination];
            const double best_p = best_conditional * prior;
            const double others_p = static_cast<double>(others_single) * best_p;
            const double total = (best_p + others_p) + d.adjusted_noise_probability;                // pamld.cpp:61-86
            double confidence = best_p / total;                                                     // pamld.cpp:92
            const double noise_margin = best_conditional - d.random_barcode_probability;
            const double confidence_margin = confidence - confidence_threshold;
            const bool noise_clear = fabs(noise_margin) > 1e-9 * d.random_barcode_probability;
            const bool is_noise = noise_margin <= 0;
            if(!noise_clear || !(prior > 0)) { defer = true; }
            if(!is_noise) {
                if(!(bounded <= static_cast<float>(cross::ELSEWHERE_WEIGHTED)) || !(others_single <= static_cast<float>(cross::DOMINANCE_WEIGHTED)) || !(fabs(confidence_margin) > 1e-8)) { defer = true; }
            } else if(!(others_single < 0.5f)) { defer = true; }                                   // the argmax: nothing else comes near the winner
            if(!defer && live) {
                int bin = static_cast<int>(combination) + 1;                                       // cross_ordinal and the index are the identity (host checked)
                uint8_t status = STATUS_PASS;
                unsigned long long* const bin_words = args.stats + d.stats_offset + static_cast<int64_t>(bin) * STAT_WORDS_PER_BIN;
                if(!is_noise) {                                                       // pamld.cpp:96
                    if(confidence_margin > 0) {                                       // pamld.cpp:99
                        if(hq_threshold > 0 && hq_distance >= hq_threshold) { r.qcfail = true; }
                    } else {
                        atomicAdd(&bin_words[FIELD_LOW_CONFIDENCE], 1ull);
                        r.qcfail = true;
                        status = STATUS_LOW_CONFIDENCE;
                    }
                } else {
                    atomicAdd(&bin_words[FIELD_LOW_CONDITIONAL_CONFIDENCE], 1ull);
                    r.qcfail = true;
                    bin = 0;
                    distance = 0;
                    hq_distance = 0;
                    confidence = 0;
                    status = STATUS_NOISE;
                    __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned long long*)static_cast<uintptr_t>(histogram_at + 8u * (STAT_JOB_WORDS + FIELD_COUNT)), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if(multiplexing) { r.channel = bin > 0 ? d.barcode_index[bin - 1] : 0; }
                DevResult out;
                out.index = bin > 0 ? d.barcode_index[bin - 1] : 0;
                out.distance = static_cast<uint8_t>(distance > 255 ? 255 : distance);
                out.hq_distance = static_cast<uint8_t>(hq_distance > 255 ? 255 : hq_distance);
                out.status = status;
                out.qcfail = r.qcfail ? 1 : 0;
                out.confidence = confidence;
                stream_result(&args.result[i], out);
                stream_read_state(args, i, r);
            }
        }
        const unsigned long long handed_on = __ballot(live && defer);
        if(handed_on) {
            if(live && defer) { segment[segment_count + __popcll(handed_on & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(i); }
            segment_count += static_cast<uint32_t>(__popcll(handed_on));
        }
        if(last_stage) {
            outgoing_pass += static_cast<uint32_t>(__popcll(__ballot(live && !defer && !r.qcfail)));
            outgoing_fail += static_cast<uint32_t>(__popcll(__ballot(live && !defer && r.qcfail)));
        }
    };

    // ---- the wave's loop over its tiles: pass A, then the ring (software pipeline of pamld_stream.hpp) ------------------
    struct Keyed { uint32_t quality[ROW_WORDS]; uint32_t state; uint32_t sliced[SEGMENTS]; uint32_t looked_up[SEGMENTS]; bool strict[SEGMENTS]; };
    auto key_and_look_up = [&](const Inputs& in, Keyed& to) {
#pragma unroll
        for(int s = 0; s < SEGMENTS; ++s) {
            uint32_t x[G];
            uint32_t q[G];
            segment_words(in.code, s, x);
            segment_words(in.quality, s, q);
            uint32_t every_byte = 0x80808080u;
            uint32_t beyond_rows = 0;
            int bits = 0;
            uint32_t sliced = 0;
            uint32_t key = 0;
#pragma unroll
            for(int w = 0; w < G; ++w) {
                const uint32_t valid = Layout::valid(w);
                x[w] &= 0x0f0f0f0fu;
                every_byte &= (x[w] + 0x7f7f7f7fu) | ~(0x80808080u & valid);
                beyond_rows |= q[w] & 0x80808080u;
                bits += __popc(x[w]);
                const uint32_t codes = gather((((x[w] >> 1) | (x[w] >> 3)) & 0x01010101u) | (((x[w] << 2) | (x[w] << 1)) & 0x10101010u));
                sliced |= codes << (8 * w + 2);
                if(w < 2) { key |= codes << (8 * w); }
                else { key |= ((codes & ((1u << TAIL) - 1u)) << 16) | (((codes >> 4) & ((1u << TAIL) - 1u)) << (16 + TAIL)); }
            }
            to.strict[s] = bits == L && every_byte == 0x80808080u && beyond_rows == 0;
            to.sliced[s] = sliced;
            to.looked_up[s] = static_cast<uint32_t>(nearest[s][key]);
        }
#pragma unroll
        for(int w = 0; w < ROW_WORDS; ++w) { to.quality[w] = in.quality[w]; }
        to.state = in.state;
    };
    int64_t tile = static_cast<int64_t>(blockIdx.x) * WAVES + wave;
    Keyed current;
    Inputs next;
    {
        Inputs first_inputs;
        fetch(min(tile, tiles - 1), first_inputs);
        fetch(min(tile + tile_step, tiles - 1), next);
        key_and_look_up(first_inputs, current);
    }
    for(bool more = true; more;) {
        unsigned long long deferred_to[Layout::RINGS] = {0, 0, 0};
        const uint32_t state_in = current.state;
        int64_t read_ordinal = 0;
        if(tile < tiles) {
            const int64_t first = tile * 64;
            const bool active = first + lane < n;
            const uint32_t at = active ? lane : static_cast<uint32_t>(n - 1 - first);
            read_ordinal = first + at;
            const uint32_t sliced_of[SEGMENTS] = {current.sliced[0], current.sliced[1]};
            const uint32_t looked_up_of[SEGMENTS] = {current.looked_up[0], current.looked_up[1]};
            const bool strict_of[SEGMENTS] = {current.strict[0], current.strict[1]};
            const bool all_strict = strict_of[0] && strict_of[1];
            uint32_t quality_in[ROW_WORDS];
#pragma unroll
            for(int w = 0; w < ROW_WORDS; ++w) { quality_in[w] = current.quality[w]; }
            Keyed following;
            key_and_look_up(next, following);
            fetch(min(tile + 2 * tile_step, tiles - 1), next);
            current = following;

            bool incoming_qcfail;
            uint32_t incoming_channel;
            if(STATE != 2) { incoming_qcfail = (state_in & 1u) != 0; incoming_channel = 0; }
            else { incoming_qcfail = (state_in >> 31) != 0; incoming_channel = state_in & 0x7fffffffu; }

            // per segment: the winner over the segment's all-match product, the listed and the bounded part of x_s
            float winner_of[SEGMENTS], x_listed_of[SEGMENTS], x_rest_of[SEGMENTS], matches_of[SEGMENTS], x_weighted_of[SEGMENTS];
            uint32_t k_of[SEGMENTS], hq_of[SEGMENTS], ordinal_of[SEGMENTS];
            bool good_of[SEGMENTS];
            bool usable = true;
            // PRIORS: the prior ratios of this read's combination, along its row (segment 1's listed words) and its column
            Weights along_row[LISTED / 4], along_column[LISTED / 4];
            if(PRIORS) {
                const uint32_t named0 = looked_up_of[0] != CROSS_STREAM_UNKNOWN ? looked_up_of[0] : 0u, named1 = looked_up_of[1] != CROSS_STREAM_UNKNOWN ? looked_up_of[1] : 0u;
                const size_t combination = static_cast<size_t>(named0) * static_cast<size_t>(radix[0]) + static_cast<size_t>(named1) * static_cast<size_t>(radix[1]);
                const Weights* const row = reinterpret_cast<const Weights*>(d.cross_near_weight[0]) + combination * (CROSS_NEAR_WEIGHTS / 4);
                const Weights* const column = reinterpret_cast<const Weights*>(d.cross_near_weight[1]) + combination * (CROSS_NEAR_WEIGHTS / 4);
#pragma unroll
                for(int quad = 0; quad < LISTED / 4; ++quad) { along_column[quad] = column[quad]; along_row[quad] = row[quad]; }
            }
#pragma unroll
            for(int s = 0; s < SEGMENTS; ++s) {
                uint32_t q[G];
                segment_words(quality_in, s, q);
                Score score;
                score_segment(q, sliced_of[s], looked_up_of[s], blocks_at[s], score, PRIORS ? (s == 0 ? along_column : along_row) : nullptr);
                x_weighted_of[s] = score.x_weighted;
                winner_of[s] = score.winner;
                matches_of[s] = score.matches;
                x_listed_of[s] = score.x_listed;
                x_rest_of[s] = score.x_rest;
                k_of[s] = score.k;
                hq_of[s] = score.high_mismatch;
                ordinal_of[s] = score.ordinal;
                good_of[s] = segment_good(score, strict_of[s]);
                usable = usable && score.known && score.k <= 1;
            }

            // both segments together: others / winner = x_0 + x_1 + x_0 x_1, between its listed part and listed + bound
            const float winner_probability = (winner_of[0] * matches_of[0]) * (winner_of[1] * matches_of[1]);      // P(r|c)
            const float upper0 = x_listed_of[0] + x_rest_of[0], upper1 = x_listed_of[1] + x_rest_of[1];
            // PRIORS: the winner's row and column by their own priors; the other combinations and what the lists leave out of the row
            // and the column by the largest prior over the winner's (along_column[0].x), as pamld_cross_near.hpp
            const float listed = PRIORS ? x_weighted_of[0] + x_weighted_of[1] : x_listed_of[0] + x_listed_of[1] + x_listed_of[0] * x_listed_of[1];
            const float x_rest = PRIORS ? along_column[0].x * (upper0 * upper1 + x_rest_of[0] + x_rest_of[1]) * near::SLOP
                                        : (upper0 + upper1 + upper0 * upper1 - listed) * near::SLOP;
            const float x_low = __builtin_fmaf(PRIORS ? along_row[0].x : noise_over_prior, __builtin_amdgcn_rcpf(winner_probability), listed);
            // the posterior is taken a quarter of the way into what was only bounded (pamld_near.hpp: at most 0.75 x_rest off):
            // 1 - y, y = x / (1 + x) = x - x^2 + x^3 (x at most 3e-4)
            const float x_mid = __builtin_fmaf(0.25f, x_rest, x_low);
            const float y = x_mid * __builtin_fmaf(-x_mid, 1.0f - x_mid, 1.0f);
            const double confidence = 1.0 - static_cast<double>(y);            // pamld.cpp:92
            // a low confidence read is left to pass C: its counter lives in a bin only that pass reaches
            const bool finished = all_strict && usable && winner_probability > noise_floor && listed <= near::LISTED_MAX && x_low + x_rest <= X_MAX &&
                                  x_rest <= near::REST_MAX && y_threshold - y > y_guard;

            const uint32_t flat_ordinal = ordinal_of[0] * static_cast<uint32_t>(radix[0]) + ordinal_of[1] * static_cast<uint32_t>(radix[1]);
            const uint32_t index = flat_ordinal + 1u;                                  // barcodes in lexicographic order of the words, identity index (host checked)
            const uint32_t k = k_of[0] + k_of[1];
            const uint32_t hq_distance = hq_of[0] + hq_of[1];
            const bool qcfail = incoming_qcfail || (hq_threshold > 0 && static_cast<int>(hq_distance) >= hq_threshold);
            const bool qcfail_out = finished ? qcfail : incoming_qcfail;
            {
                Quad out;
                out.x = finished ? index : 0u;
                out.y = finished ? (k | (hq_distance << 8) | (static_cast<uint32_t>(STATUS_PASS) << 16) | (qcfail_out ? 1u << 24 : 0u))
                                 : ((static_cast<uint32_t>(STATUS_UNCLASSIFIED) << 16) | (qcfail_out ? 1u << 24 : 0u));
                out.z = finished ? static_cast<uint32_t>(__double2loint(confidence)) : 0u;
                out.w = finished ? static_cast<uint32_t>(__double2hiint(confidence)) : 0u;
                __builtin_nontemporal_store(out, reinterpret_cast<Quad*>(reinterpret_cast<char*>(args.result) + first * 16 + at * 16u));
                const uint32_t channel = (finished && multiplexing) ? index : incoming_channel;
                __builtin_nontemporal_store((channel & 0x7fffffffu) | (qcfail_out ? 0x80000000u : 0u), reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(args.word) + first * 4 + at * 4u));
            }
            {
                // which ring: a segment pass A found good is not scanned again
                const bool left = active && !finished;
                const bool bad0 = !good_of[0], bad1 = !good_of[1];
                const unsigned long long both = __ballot(left && bad0 && bad1);
                deferred_to[1] = __ballot(left && !bad0 && bad1);
                deferred_to[0] = __ballot(left && !bad1);
                if(PRIORS) {
                    // priors that differ: pass C takes a read one of whose segments is good (and, of a read both of whose are,
                    // scans segment 0); one with neither goes to the worklist for the three scan kernel
                    if(both) {
                        if((both >> lane) & 1ull) { segment[segment_count + static_cast<uint32_t>(__popcll(both & ((1ull << lane) - 1ull)))] = static_cast<uint32_t>(read_ordinal); }
                        segment_count += static_cast<uint32_t>(__popcll(both));
                    }
                } else { deferred_to[2] = both; }
            }
            if(last_stage) {
                outgoing_pass += static_cast<uint32_t>(__popcll(__ballot(active && finished && !qcfail)));
                outgoing_fail += static_cast<uint32_t>(__popcll(__ballot(active && finished && qcfail)));
            }
            tile += tile_step;
        } else {
            more = false;
        }
        // a ring its arrivals would overflow (or, behind the last tile, one that holds anything) is scanned first — one call
        // site, the ring's number a scalar
        uint32_t arriving[Layout::RINGS];
#pragma unroll
        for(int m = 0; m < Layout::RINGS; ++m) { arriving[m] = static_cast<uint32_t>(__popcll(deferred_to[m])); }
        for(;;) {
            const bool due0 = ring_count[0] + arriving[0] > RING || (!more && ring_count[0] > 0);
            const bool due1 = ring_count[1] + arriving[1] > RING || (!more && ring_count[1] > 0);
            const bool due2 = ring_count[2] + arriving[2] > RING || (!more && ring_count[2] > 0);
            if(__builtin_expect(!(due0 || due1 || due2), 1)) { break; }      // the scan is the cold path: what it spills, it spills there
            const int m = due0 ? 0 : due1 ? 1 : 2;
            if(PRIORS) { drain_weighted(m, m == 0 ? ring_count[0] : ring_count[1]); }
            else { drain(m, m == 0 ? ring_count[0] : m == 1 ? ring_count[1] : ring_count[2]); }
            ring_count[0] = m == 0 ? 0u : ring_count[0];
            ring_count[1] = m == 1 ? 0u : ring_count[1];
            ring_count[2] = m == 2 ? 0u : ring_count[2];
        }
#pragma unroll
        for(int m = 0; m < Layout::RINGS; ++m) {
            const unsigned long long deferred = deferred_to[m];
            if(arriving[m]) {
                if((deferred >> lane) & 1ull) {
                    const uint32_t slot = ring_count[m] + static_cast<uint32_t>(__popcll(deferred & ((1ull << lane) - 1ull)));
                    const uint32_t ring_at = rings_at + static_cast<uint32_t>(m) * (Layout::RING_PLANES * RING * 4u) + slot * 4u;
                    *(__attribute__((address_space(3))) uint32_t*)static_cast<uintptr_t>(ring_at) = state_in;
                    *(__attribute__((address_space(3))) uint32_t*)static_cast<uintptr_t>(ring_at + RING * 4u) = static_cast<uint32_t>(read_ordinal);
                }
                ring_count[m] += arriving[m];
            }
        }
    }
    if(lane == 0) {
        args.worklist_count[own_segment] = segment_count;
        if(outgoing_fail) { atomicAdd(&histogram[0], static_cast<unsigned long long>(outgoing_fail)); }
        if(outgoing_pass) { atomicAdd(&histogram[1], static_cast<unsigned long long>(outgoing_pass)); }
    }
    __syncthreads();
    histogram_flush(args, d, histogram);
}

template <int L, int STATE, int WAVES, bool PRIORS = false>
int launch_state(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, hipStream_t stream) {
    static std::atomic<uint64_t> prepared{0};
    if(const int status = allow_large_lds(&pamld_cross_stream_kernel<L, STATE, WAVES, PRIORS>, prepared)) { return status; }
    typedef Shape<L, WAVES> Layout;
    hipLaunchKernelGGL((pamld_cross_stream_kernel<L, STATE, WAVES, PRIORS>), grid, dim3(Layout::THREADS), Layout::lds_bytes(host_decoder.cross_unique[0], host_decoder.cross_unique[1]), stream, args);
    return static_cast<int>(hipGetLastError());
}

template <int L, int WAVES, bool PRIORS = false>
int launch_one(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, hipStream_t stream) {
    if(!args.first_stage) { return launch_state<L, 2, WAVES, PRIORS>(args, host_decoder, grid, stream); }
    return args.input_flag ? launch_state<L, 1, WAVES, PRIORS>(args, host_decoder, grid, stream) : launch_state<L, 0, WAVES, PRIORS>(args, host_decoder, grid, stream);
}

}  // namespace cross_stream
}  // namespace phq
