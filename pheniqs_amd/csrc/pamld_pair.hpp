// K0p (paired barcodes): PamlDecoder<T>::classify for codecs whose barcodes are PAIRS of words that do not form every
// combination (unique dual indexes, sci-RNA-seq ligation + RT pairs): two segments of up to 10 nt, up to 32 nt together —
// and, since round 4, for single index sets of 11 to 20 nt, which the host hands over as two parts of one segment (nothing
// below depends on where a segment ends: a barcode is scored over all its positions and the others are bounded by its
// distance to the nearest of them).
//
//   reference pamld.cpp:37-123   posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164  P(r|b) = pow(base, sum of Phred costs), distance, high quality distance
//
// Two barcodes of such a codec differ in most of their ~20 positions (the host finds, per barcode c, the distance r(c)
// to the nearest other one: 8 or more for sets like these), so a read that sits on or next to c leaves every other
// barcode at least n = r(c) - k mismatches away, k = the read's own mismatches against c.  The kernel therefore scores
// ONE barcode per read and bounds all the others:
//   * c by two table lookups, one per segment (the tables of pamld_near.hpp: the word at most one mismatch from a
//     segment word — here up to TWO mismatches: the nearest word's barcode, lowest ordinal among equals —, 4^length entries
//     of 16 bits in global memory); both segments must name the same barcode;
//   * priors enter through two numbers per barcode (host): the priors of all the others over c's, which scales the bound,
//     and the noise term over c's prior;
//   * P(r|c) and the all-match product A as single precision products of the per position factors, the factor picked by
//     putting the position's mismatch flag into the LDS address (no select);
//   * every other barcode is at most  A * 4^wild * rho12^a * rho20^b * rho25^c * rho30^d * rho36^e  where the n mismatches are
//     charged to the positions in order of quality — those below Q12 for free, a of those below Q20, b below Q25, ... —
//     `wild` counts the positions below Q4 (a mismatch there is MORE likely than a match, by at most 4), and all the
//     counts come from byte parallel compares.  The others come in three classes (host): at the radius, one further away
//     (n + 1 mismatches), the rest (n + 2) — of 383 other pairs two or three sit at the radius, and charging all of them
//     there left the bound 100 times what it is (384 pairs of 8 + 8 nt: 21.8 % of the reads deferred).
// A read is finished under the certificate of pamld_near.hpp (1e-8 in the posterior); everything else — a third
// mismatch in a segment, a base outside ACGT, many low quality positions, noise — goes to the product form kernel (pamld_wide.hpp
// for 17 to 32 nt, pamld_lut.hpp up to 16) as a consumer of this workgroup's worklist segment.
// No per read tables: 12 KiB of LDS per workgroup, eight waves per SIMD.
#pragma once

#include "kernel_common.hpp"

namespace phq {
namespace pair {

constexpr int THREADS = 256;
constexpr int FACTOR_ROWS = 129;
constexpr float X_MAX = 3e-4f;                 // noise / winner: single precision is enough below this (pamld_near.hpp; up to 20 positions here)
constexpr float REST_MAX = 9e-9f;              // the bound on what was not scored, as a share of the winner (pamld_near.hpp: a tenth under the 1e-8 of the tests)
constexpr float SLOP = 1.0f + 1e-5f;
// The bound sorts the positions by quality into six classes; n mismatches are charged to the cheapest classes first:
// below Q12 for free, then rho(Q12) each for those below Q20, rho(Q20) below Q25, rho(Q25) below Q30, rho(Q30) below Q36,
// rho(Q36) for the rest (rho = P(mismatch) / P(match) falls with the quality; the logarithms are rounded towards zero).
// Measured on the model of the sci-RNA-seq shape (profiles/tools/sim/pair_certificate.py): three marks {20, 30, 36}
// finish 87.3 % of the reads (82.3 % with 196 pairs), these five 91.0 % (89.1 %), and with tables that name a barcode
// up to two mismatches per segment away 93.6 % (90.8 %) — at the 1e-9 bound of then; at 1e-8 94.4 %, and 95.4 % with the
// other barcodes in three distance classes (the kernel: 95.3 %).
constexpr int MARKS = 5;
constexpr int MARK[MARKS] = {12, 20, 25, 30, 36};
constexpr float LOG2_RHO[MARKS] = {-3.8922f, -6.6293f, -8.3002f, -9.9643f, -11.9584f};
constexpr int MOST_MISMATCHES = 4;            // of a finished read against its barcode: two per segment (the tables' reach)
constexpr int WILD_QUALITY = 4;                // below it P(mismatch) > P(match), by at most 4 (Q1: 3.86)

// LISTS (round 5): the second look, inside the same launch.  What defers a read of the first look is almost never a barcode
// that weighs something: it is a low quality call that COULD hold the mismatch of one — a read with a Q2 call outside its own
// mismatches has that position absorb a mismatch of every other barcode for free.  Scoring the four nearest barcodes of c (all
// those at the radius and most of those one further, for sets like these) moves the bound's reach out by one or two positions,
// each worth 1e-2 to 1e-4 (profiles/tools/sim/pair_bound_own.py: 384 pairs of 8 + 8 nt: 93.8 % of the reads finished by the
// first look, 96.7 % with four listed — 97.3 % is everything but the reads with a base outside ACGT and the noise reads; 96
// indexes of 12 nt: 86.3 -> 96.7 %).  As pamld_stream.hpp does it: a read the first look cannot certify — but whose barcode it
// knows and whose bases are all ACGT — waits, inputs and all, in its WAVE's ring in LDS (64 entries, placed by a ballot and a
// popcount); when the next tile could overflow the ring the wave turns to it with all lanes busy: the same arithmetic with the
// listed barcodes' products beside the winner's (lists from L2: 64 bytes a barcode), the others bounded from the nearest barcode
// that is NOT listed on.  No HBM round trip for those reads (as a consumer launch over the worklist the same second look cost
// more than it saved: 13.7 M scattered rows and records of 16 bytes are 5 GB of line traffic); the product form over ALL
// barcodes behind this kernel sees half to a quarter of the reads it saw.
constexpr int LISTED = PAIR_LISTED;
constexpr int RING = 64;
template <int G, bool LISTS> struct LayoutOf {
    static constexpr uint32_t FACTOR_AT = 0;                                   // FACTOR_ROWS x {match, mismatch} floats
    static constexpr uint32_t DEFERRED_AT = FACTOR_AT + 1040;
    static constexpr uint32_t RING_AT = DEFERRED_AT + 16;                      // LISTS: per wave [2 G + 3 planes][RING]: code words, quality words, state, ordinal, c
    static constexpr uint32_t RING_PLANES = 2 * G + 3;
    static constexpr uint32_t RING_BYTES = LISTS ? RING_PLANES * RING * 4u : 0u;
    static constexpr uint32_t RECORD_AT = RING_AT + (THREADS / 64) * RING_BYTES;      // n_barcode x {low, high, radius, index, others, noise, -, -}
    __host__ __device__ static constexpr uint32_t histogram_at(int n_barcode) { return RECORD_AT + static_cast<uint32_t>(n_barcode) * (PAIR_RECORD_WORDS * 4u); }
};

typedef uint32_t Quad __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const float* LdsFloats;
typedef __attribute__((address_space(3))) const Quad* LdsQuads;

__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, mask, c, 0xEA); }

// G = words of the plane row (4 positions each); STATE as in pamld_near.hpp; LISTS: the second look on the wave's ring
#ifndef PHQ_PAIR_WAVES
#define PHQ_PAIR_WAVES 5
#endif
template <int G, int STATE, bool LISTS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(LISTS ? PHQ_PAIR_WAVES : 7, 8))) void pamld_pair_kernel(StageArgs args) {
    typedef LayoutOf<G, LISTS> Layout;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    float* const factor = reinterpret_cast<float*>(lds_raw + Layout::FACTOR_AT);           // [row][0 match | 1 mismatch]
    uint32_t& deferred_total = *reinterpret_cast<uint32_t*>(lds_raw + Layout::DEFERRED_AT);
    uint32_t* const records = reinterpret_cast<uint32_t*>(lds_raw + Layout::RECORD_AT);
    DecoderRef d = *constant_cast(args.decoder);
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw + Layout::histogram_at(d.n_barcode));

    for(int i = threadIdx.x; i < FACTOR_ROWS; i += THREADS) {
        factor[2 * i] = static_cast<float>(i < 128 ? d.match_probability[i] : d.uniform_probability);
        factor[2 * i + 1] = static_cast<float>(i < 128 ? d.mismatch_probability[i] : d.uniform_probability);
    }
    if(threadIdx.x == 0) { deferred_total = 0; }
    for(int w = threadIdx.x; w < d.n_barcode * PAIR_RECORD_WORDS; w += THREADS) { records[w] = d.pair_record[w]; }
    histogram_clear(d, histogram);
    __syncthreads();

    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const uint16_t* const nearest0 = d.pair_table[0];
    const uint16_t* const nearest1 = d.pair_table[1];
    const Quad* const lists = reinterpret_cast<const Quad*>(d.pair_list);                  // LISTS: four quads a barcode
    const uint32_t start0 = static_cast<uint32_t>(d.pair_start[0]), start1 = static_cast<uint32_t>(d.pair_start[1]);
    const uint32_t length0 = static_cast<uint32_t>(d.pair_length[0]), length1 = static_cast<uint32_t>(d.pair_length[1]);
    const uint32_t mask0 = (1u << length0) - 1u, mask1 = (1u << length1) - 1u;
    const int L = d.nucleotide_cardinality;
    const uint32_t full_mask = L >= 32 ? 0xffffffffu : ((1u << L) - 1u);
    const int64_t n = args.n_read;
    const int64_t tiles = (n + THREADS - 1) / THREADS;
    const int high_quality_threshold = min(max(d.high_quality_threshold, 0), 128);
    const uint32_t high_quality_addend = static_cast<uint32_t>(128 - high_quality_threshold) * 0x01010101u;
    const float noise_floor = static_cast<float>(d.random_barcode_probability) * SLOP;
    const double confidence_threshold = d.confidence_threshold;
    const uint32_t n_barcode = static_cast<uint32_t>(d.n_barcode);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* const ring = reinterpret_cast<uint32_t*>(lds_raw + Layout::RING_AT + (threadIdx.x >> 6) * Layout::RING_BYTES);
    uint32_t ring_count = 0;                       // wave uniform
    // the bytes of the positions a word holds (word w: positions 4 w .. 4 w + 3)
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
    const int hq_threshold = d.high_quality_distance_threshold;
    const bool multiplexing = d.multiplexing != 0;
    const bool lds_bins = d.use_lds_histogram != 0;          // else the launch defers the classified reads' counters (StageArgs::defer_counters)

    // What a look at a read comes to.  `candidate`: everything but the bound stands — the barcode is known, the bases are all
    // ACGT, the mismatches within the tables' reach, the noise filter passed: the read is worth the second look.
    // `retry`: the two parts do not agree on a barcode (an error in the shorter part of a 12 nt index sends its lookup to another
    // barcode's word: 9 % of such reads) but one of them names one: the second look takes that one — its certificate holds for
    // whatever barcode it is given (the listed barcodes are scored, the others bounded by their distance from THAT barcode), a
    // wrong guess only fails it.
    struct Look { bool finished, candidate, retry, confident, qcfail; uint32_t c, k, hq_distance, index; double confidence; };
    // One read against its barcode (and, second look, against the barcodes listed with it).  first look: `named` comes from the
    // two lookups; second look: from the ring.
    auto look = [&](auto second_look, const uint32_t (&code)[G], const uint32_t (&quality)[G], bool incoming_qcfail, bool known, uint32_t named) {
        constexpr bool SECOND = decltype(second_look)::value;
        auto gather = [](uint32_t planes) {            // bits 8 j and 8 j + 4 -> bits j and 4 + j
            const uint32_t t = planes | (planes >> 7);
            return (t | (t >> 14)) & 0xffu;
        };
        uint32_t q7[G];
        uint32_t row_of[G];                              // per position the row of its factor pair: its quality, 128 for a base outside ACGT, 0 at quality 0
        uint32_t low = 0, high = 0, high_quality = 0;    // bit j = position j
        uint32_t outside = 0;                            // bit j: the base is none of A, C, G, T
#pragma unroll
        for(int w = 0; w < G; ++w) {
            const uint32_t x = code[w] & (0x0f0f0f0fu & valid[w]);
            q7[w] = quality[w] & (0x7f7f7f7fu & valid[w]);
            // A, C, G or T = exactly one bit of the nibble (iupac.h:51-56)
            const uint32_t z = x & ((x | 0x80808080u) - 0x01010101u);
            const uint32_t strict = (x + 0x7f7f7f7fu) & ~(z + 0x7f7f7f7fu) & 0x80808080u;
            // Round 5: a base outside ACGT no longer sends the read on.  Against every barcode base it costs the uniform quality
            // (row 128: match and mismatch weigh the same; nothing at quality 0: row 0) whatever the lookup makes of its bits
            // (phred.cpp:62-64), it counts as a mismatch (barcode.h:147-161), and as one of the read's OWN mismatches it is outside
            // what the bound charges: another barcode differs from c there for free, which is what `n = r(c) - k` says with k
            // counting it.  1.6 % of a 16 nt index's reads hold an N; they were the product form's for that alone.
            const uint32_t strict_bytes = (strict - (strict >> 7)) | strict;                       // 0xff where A, C, G or T
            const uint32_t nonzero = (q7[w] + 0x7f7f7f7fu) & 0x80808080u;
            row_of[w] = (q7[w] & strict_bytes) | (nonzero & ~strict_bytes & valid[w]);
            outside |= (gather((~strict & 0x80808080u & valid[w]) >> 7) & 0xfu) << (4 * w);
            // 2-bit codes: the low code bit is bit 1 | bit 3 of the nibble, the high one bit 2 | bit 3
            const uint32_t both = gather((((x >> 1) | (x >> 3)) & 0x01010101u) | (((x << 2) | (x << 1)) & 0x10101010u));
            low |= (both & 0xfu) << (4 * w);
            high |= (both >> 4) << (4 * w);
            // quality >= high quality threshold
            const uint32_t is_high = ((q7[w] + high_quality_addend) | quality[w]) & (0x80808080u & valid[w]);
            high_quality |= (gather(is_high >> 7) & 0xfu) << (4 * w);
        }
        uint32_t either = n_barcode;                   // first look: the barcode one part names when the two do not agree
        if(!SECOND) {
            const uint32_t key0 = ((low >> start0) & mask0) | (((high >> start0) & mask0) << length0);
            const uint32_t key1 = ((low >> start1) & mask1) | (((high >> start1) & mask1) << length1);
            const uint32_t named0 = nearest0[key0];
            const uint32_t named1 = nearest1[key1];
            // both parts name the same barcode — or one of them names no one of its own (PAIR_ANY_BARCODE: a single index of 11 to 20 nt
            // taken as two parts, whose shorter part's words are shared between barcodes) and the other part decides
            named = named0 == PAIR_ANY_BARCODE ? named1 : named0;
            known = (named0 == named1 || named0 == PAIR_ANY_BARCODE || named1 == PAIR_ANY_BARCODE) && named < n_barcode;
            either = named0 < n_barcode ? named0 : named1;
        }
        const uint32_t c = known ? named : 0u;
        const Quad record = *(LdsQuads)static_cast<uintptr_t>(Layout::RECORD_AT + c * (PAIR_RECORD_WORDS * 4u));
        // the priors of all other barcodes over c's (N - 1 when they are equal) and the noise term over c's prior (pamld.cpp:29,66)
        const Quad weights = *(LdsQuads)static_cast<uintptr_t>(Layout::RECORD_AT + c * (PAIR_RECORD_WORDS * 4u) + 16u);
        float others_at_radius = __uint_as_float(weights.x);          // ... at the radius, one further away, all the rest
        const float noise_over_prior = __uint_as_float(weights.y);
        float others_next = __uint_as_float(weights.z);
        float others_beyond = __uint_as_float(weights.w);
        uint32_t radius = record.z;
        // second look: the LISTED nearest barcodes of c as plane words, the distance of the nearest barcode that is not listed with
        // the others' priors over c's in three distance classes from there on, the prior of each listed barcode over c's
        uint32_t listed_flags[LISTED];
        float listed_ratio[LISTED];
        if(SECOND) {
            static_assert(LISTED == 4 && PAIR_LIST_WORDS == 16, "four quads: low planes, high planes, {distance, three weights}, ratios");
            const Quad list_low = lists[4u * c], list_high = lists[4u * c + 1u], beyond_list = lists[4u * c + 2u], ratios = lists[4u * c + 3u];
#pragma unroll
            for(int o = 0; o < LISTED; ++o) {
                listed_flags[o] = (((low ^ list_low[o]) | (high ^ list_high[o])) & full_mask) | outside;
                listed_ratio[o] = __uint_as_float(ratios[o]);
            }
            radius = beyond_list.x;
            others_at_radius = __uint_as_float(beyond_list.y); others_next = __uint_as_float(beyond_list.z); others_beyond = __uint_as_float(beyond_list.w);
        }
        const uint32_t mismatches = (((low ^ record.x) | (high ^ record.y)) & full_mask) | outside;
        const int k = __popc(mismatches);
        // The positions by quality, for the bound — WITHOUT the read's own mismatches (round 5).  Another barcode differs from c in
        // r(c) or more positions, at most k of them the read's own: it mismatches the read in n = r(c) - k or more positions
        // where the read agrees with c, and those are what the bound charges to the cheapest positions.  At the read's own
        // mismatches that barcode weighs at most max(1, 1 / rho) of what c does — the division by the winner below, and the
        // factor 4 of a wild position.  (Until round 4 the counts took every position: a read whose one error sits on its one
        // Q2 call had that call absorb a mismatch it cannot hold, and five orders of magnitude went with it — 384 pairs of 8 + 8
        // nt: 13.4 % of the reads deferred, 6.2 % now; 96 indexes of 12 nt 22.6 -> 13.7 %; profiles/tools/sim/pair_bound_own.py.)
        uint32_t below[MARKS] = {0, 0, 0, 0, 0};         // positions below each mark
        uint32_t wild = 0;
#pragma unroll
        for(int w = 0; w < G; ++w) {
            const uint32_t own = (((mismatches >> (4 * w)) & 0xfu) * 0x00204081u & 0x01010101u) << 7;          // bit j of the nibble -> 0x80 of byte j
            const uint32_t counted = 0x80808080u & valid[w] & ~own;
#pragma unroll
            for(int m = 0; m < MARKS; ++m) { below[m] = __builtin_popcount(~(q7[w] + (128u - MARK[m]) * 0x01010101u) & counted) + below[m]; }
            wild = __builtin_popcount(~(q7[w] + (128u - WILD_QUALITY) * 0x01010101u) & (0x80808080u & valid[w])) + wild;
        }

        // P(r|c) and the all match product: the factor row of the position's quality, the mismatch flag in the address
        float winner = 1.0f;
        float everything_matches = 1.0f;
        float listed_product[LISTED];
#pragma unroll
        for(int j = 0; j < 4 * G; ++j) {
            const uint32_t q = row_of[j >> 2];
            const int byte = j & 3;
            const uint32_t row = and_or(byte == 0 ? (q << 3) : (q >> (8 * byte - 3)), 0x7f8u, static_cast<uint32_t>(Layout::FACTOR_AT));
            const uint32_t picked = and_or(j >= 2 ? (mismatches >> (j - 2)) : (mismatches << (2 - j)), 4u, row);
            const float matching = *(LdsFloats)static_cast<uintptr_t>(row);
            const float own = *(LdsFloats)static_cast<uintptr_t>(picked);
            everything_matches = j == 0 ? matching : everything_matches * matching;
            winner = j == 0 ? own : winner * own;
            if(SECOND) {
#pragma unroll
                for(int o = 0; o < LISTED; ++o) {
                    const float factor_of = *(LdsFloats)static_cast<uintptr_t>(and_or(j >= 2 ? (listed_flags[o] >> (j - 2)) : (listed_flags[o] << (2 - j)), 4u, row));
                    listed_product[o] = j == 0 ? factor_of : listed_product[o] * factor_of;
                }
            }
        }

        // the other barcodes: those at the radius at least n = r(c) - k mismatches outside the read's own, the next n + 1, the rest n + 2
        auto log2_bound_of = [&](int mismatches_outside) {
            int needed = max(mismatches_outside - static_cast<int>(below[0]), 0);
            float log2_bound = 2.0f * static_cast<float>(wild);
#pragma unroll
            for(int m = 0; m + 1 < MARKS; ++m) {
                const int charged = min(needed, static_cast<int>(below[m + 1] - below[m]));
                needed -= charged;
                log2_bound += LOG2_RHO[m] * static_cast<float>(charged);
            }
            return log2_bound + LOG2_RHO[MARKS - 1] * static_cast<float>(needed);
        };
        const int beyond_own = static_cast<int>(radius) - k;
        const float inverse_winner = __builtin_amdgcn_rcpf(winner);
        const float x_rest = (others_at_radius * __builtin_amdgcn_exp2f(log2_bound_of(beyond_own)) + others_next * __builtin_amdgcn_exp2f(log2_bound_of(beyond_own + 1)) +
                              others_beyond * __builtin_amdgcn_exp2f(log2_bound_of(beyond_own + 2))) * everything_matches * inverse_winner * SLOP;
        // second look: what the listed barcodes weigh, in units of the winner (a slot past the set's barcodes has ratio 0)
        float x_listed = 0.0f;
        if(SECOND) {
#pragma unroll
            for(int o = 0; o < LISTED; ++o) { x_listed = __builtin_fmaf(listed_ratio[o], listed_product[o], x_listed); }
            x_listed *= inverse_winner;
        }
        const float x_low = __builtin_fmaf(noise_over_prior, inverse_winner, x_listed);
        // the posterior is taken a QUARTER of the way into what was only bounded: those barcodes really weigh between nothing and
        // x_rest, typically a tenth of it (profiles/tools/sim/near_certificate.py), so the error is at most 0.75 x_rest —
        // 6.8e-9 at REST_MAX, with x's own 7e-10 on top 7.5e-9: a proven distance from the 1e-8 the tests assert (until round 5
        // it was taken without them: at most x_rest, 9.7e-9 with the rounding, and 8.8e-9 was seen) — and typically 0.15 of it
        const float x_mid = __builtin_fmaf(0.25f, x_rest, x_low);
        const float y = x_mid * __builtin_amdgcn_rcpf(1.0f + x_mid);       // 1 - posterior
        Look out;
        out.confidence = 1.0 - static_cast<double>(y);                     // pamld.cpp:92
        const double confidence_margin = out.confidence - confidence_threshold;
        out.candidate = known && k <= MOST_MISMATCHES && winner > noise_floor;
        out.finished = out.candidate && x_low + x_rest <= X_MAX && x_rest <= REST_MAX && x_listed <= X_MAX && fabs(confidence_margin) > 2e-8;
        out.retry = !SECOND && !known && either < n_barcode;
        out.c = out.retry ? either : c;
        out.k = static_cast<uint32_t>(k);
        out.index = record.w;                                              // barcode_index[c]
        out.hq_distance = static_cast<uint32_t>(__popc(mismatches & high_quality));
        out.confident = confidence_margin > 0;                             // pamld.cpp:99
        out.qcfail = incoming_qcfail || !out.confident || (hq_threshold > 0 && static_cast<int>(out.hq_distance) >= hq_threshold);
        return out;
    };
    // Decoder<T>::classify then Classifier<T>::classify for a finished read (decoder.h:68-76, classifier.h:78-86)
    auto count = [&](const Look& o) {
        const uint32_t word = STAT_JOB_WORDS + (o.c + 1u) * STAT_WORDS_PER_BIN + (o.qcfail ? 0u : static_cast<uint32_t>(STAT_FIELDS));
        uint32_t* const low_words = reinterpret_cast<uint32_t*>(histogram);      // a workgroup's counts and distances stay below 2^32
        if(o.confident) {
            if(!args.defer_counters) {
                atomicAdd(&low_words[2 * (word + FIELD_COUNT)], 1u);
                if(o.k) { atomicAdd(&low_words[2 * (word + FIELD_DISTANCE)], o.k); }
                const double shifted = o.confidence + 1.0;                        // round(confidence * 2^52) in the mantissa (pamld_near.hpp)
                const uint32_t fixed_low = static_cast<uint32_t>(__double2loint(shifted));
                const uint32_t fixed_high = static_cast<uint32_t>(__double2hiint(shifted)) - 0x3ff00000u;
                atomicAdd(&histogram[word + FIELD_CONFIDENCE_LOW], static_cast<unsigned long long>(fixed_low & 0x3ffffffu));
                atomicAdd(&histogram[word + FIELD_CONFIDENCE_HIGH], static_cast<unsigned long long>((fixed_high << 6) | (fixed_low >> 26)));
            }
        } else {
            // qc-fail half (pamld.cpp:104-106); a codec too large for a workgroup histogram counts it in global memory (rare)
            if(lds_bins) { atomicAdd(&low_words[2 * (STAT_JOB_WORDS + (o.c + 1u) * STAT_WORDS_PER_BIN + FIELD_LOW_CONFIDENCE)], 1u); }
            else { atomicAdd(&args.stats[d.stats_offset + (o.c + 1u) * STAT_WORDS_PER_BIN + FIELD_LOW_CONFIDENCE], 1ull); }
            if(!args.defer_counters) {
                atomicAdd(&low_words[2 * (word + FIELD_COUNT)], 1u);
                if(o.k) { atomicAdd(&low_words[2 * (word + FIELD_DISTANCE)], o.k); }
            }
        }
    };
    typedef uint32_t vector4 __attribute__((ext_vector_type(4)));
    auto record_of = [&](const Look& o, bool finished, bool qcfail_out) {
        vector4 out;
        out.x = o.index;
        out.y = o.k | (o.hq_distance << 8) | ((o.confident ? STATUS_PASS : STATUS_LOW_CONFIDENCE) << 16) | (qcfail_out ? 1u << 24 : 0u);
        out.z = static_cast<uint32_t>(__double2loint(o.confidence));
        out.w = static_cast<uint32_t>(__double2hiint(o.confidence));
        return out;
    };
    // a run of slots in this workgroup's worklist segment for the reads of a wave that go on to the next tier
    auto reserve = [&](unsigned long long handed_on) {
        uint32_t run = 0;
        if(lane == 0) { run = atomicAdd(&deferred_total, static_cast<uint32_t>(__popcll(handed_on))); }
        return static_cast<uint32_t>(__shfl(static_cast<int>(run), 0)) + static_cast<uint32_t>(__popcll(handed_on & ((1ull << lane) - 1ull)));
    };

    // ---- the second look: the wave's ring, all lanes busy -------------------------------------------------------------------
    auto drain = [&](uint32_t waiting) {
        // the records and status words the first look stored for these reads (placeholders, by other lanes) must have left
        // before the real ones go to the same places
        __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0)
        const bool live = lane < waiting;
        const uint32_t slot = live ? lane : 0u;
        uint32_t code[G], quality[G];
#pragma unroll
        for(int w = 0; w < G; ++w) { code[w] = ring[w * RING + slot]; quality[w] = ring[(G + w) * RING + slot]; }
        const uint32_t state = ring[2 * G * RING + slot];
        const int64_t i = static_cast<int64_t>(ring[(2 * G + 1) * RING + slot]);
        const uint32_t named = ring[(2 * G + 2) * RING + slot];
        ReadState r;
        if(STATE != 2) { r.qcfail = (state & 1u) != 0; r.channel = 0; }
        else { r.qcfail = (state >> 31) != 0; r.channel = static_cast<int32_t>(state & 0x7fffffffu); }
        const Look o = look(std::integral_constant<bool, true>(), code, quality, r.qcfail, true, named);
        const bool finished = live && o.finished;
        if(finished) {
            *reinterpret_cast<vector4*>(reinterpret_cast<char*>(args.result) + i * 16) = record_of(o, true, o.qcfail);
            const uint32_t channel = multiplexing ? o.index : static_cast<uint32_t>(r.channel);
            args.word[i] = (channel & 0x7fffffffu) | (o.qcfail ? 0x80000000u : 0u);
            r.qcfail = o.qcfail;
            count(o);
        }
        const unsigned long long handed_on = __ballot(live && !o.finished);
        if(handed_on) {
            const uint32_t at = reserve(handed_on);
            if(live && !o.finished) { segment[at] = static_cast<uint32_t>(i); }
        }
        count_outgoing(args, histogram, finished, r);
    };

    Inputs next;
    fetch(min(static_cast<int64_t>(blockIdx.x), tiles - 1), next);
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0) before the loop (pamld_near.hpp)
    uint32_t pending_slot = 0xffffffffu;
    uint32_t pending_read = 0;
    for(int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const Inputs in = next;
        const int64_t first = tile * THREADS;
        const bool active = first + threadIdx.x < n;
        const uint32_t at = active ? threadIdx.x : static_cast<uint32_t>(n - 1 - first);
        const int64_t i = first + at;
        ReadState r;
        if(STATE != 2) { r.qcfail = (in.state & 1u) != 0; r.channel = 0; }
        else { r.qcfail = (in.state >> 31) != 0; r.channel = static_cast<int32_t>(in.state & 0x7fffffffu); }
        if(pending_slot != 0xffffffffu) { segment[pending_slot] = pending_read; }      // the previous tile's deferred reads
        const Look o = look(std::integral_constant<bool, false>(), in.code, in.quality, r.qcfail, false, 0u);
        fetch(min(tile + gridDim.x, tiles - 1), next);

        // outcome: straight line stores (pamld_near.hpp) — the real record of a finished read, a placeholder and the state it came
        // with for every other (the second look or the next tier writes both again)
        const bool qcfail_out = o.finished ? o.qcfail : r.qcfail;
        {
            __builtin_nontemporal_store(record_of(o, o.finished, qcfail_out), reinterpret_cast<vector4*>(reinterpret_cast<char*>(args.result) + first * 16 + at * 16u));
            const uint32_t channel = (o.finished && multiplexing) ? o.index : static_cast<uint32_t>(r.channel);
            __builtin_nontemporal_store((channel & 0x7fffffffu) | (qcfail_out ? 0x80000000u : 0u), reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(args.word) + first * 4 + at * 4u));
        }
        r.qcfail = qcfail_out;
        if(o.finished && active) { count(o); }
        const bool waits = LISTS && active && !o.finished && (o.candidate || o.retry);      // to the wave's ring: the second look
        const unsigned long long deferred = __ballot(active && !o.finished && !waits);  // to the next tier
        pending_slot = 0xffffffffu;
        if(deferred) {
            const uint32_t slot = reserve(deferred);
            if(active && !o.finished && !waits) { pending_slot = slot; pending_read = static_cast<uint32_t>(i); }
        }
        count_outgoing(args, histogram, active && o.finished, r);
        if(LISTS) {
            const unsigned long long waiting = __ballot(waits);
            const uint32_t arriving = static_cast<uint32_t>(__popcll(waiting));
            if(ring_count + arriving > static_cast<uint32_t>(RING)) { drain(ring_count); ring_count = 0; }
            if(arriving) {
                if(waits) {
                    const uint32_t slot = ring_count + static_cast<uint32_t>(__popcll(waiting & ((1ull << lane) - 1ull)));
#pragma unroll
                    for(int w = 0; w < G; ++w) { ring[w * RING + slot] = in.code[w]; ring[(G + w) * RING + slot] = in.quality[w]; }
                    ring[2 * G * RING + slot] = in.state;
                    ring[(2 * G + 1) * RING + slot] = static_cast<uint32_t>(i);
                    ring[(2 * G + 2) * RING + slot] = o.c;
                }
                ring_count += arriving;
            }
        }
    }
    if(pending_slot != 0xffffffffu) { segment[pending_slot] = pending_read; }
    if(LISTS && ring_count > 0) { drain(ring_count); }
    __syncthreads();
    if(threadIdx.x == 0) { args.worklist_count[blockIdx.x] = deferred_total; }
    histogram_flush(args, d, histogram);
}

template <int G, int STATE>
int launch_state(const StageArgs& args, int n_barcode, bool lists, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    static std::atomic<uint64_t> prepared{0}, prepared_lists{0};
    if(lists) {
        const size_t lds_bytes = LayoutOf<G, true>::histogram_at(n_barcode) + histogram_bytes;
        if(const int status = allow_large_lds(&pamld_pair_kernel<G, STATE, true>, prepared_lists)) { return status; }
        hipLaunchKernelGGL((pamld_pair_kernel<G, STATE, true>), grid, dim3(THREADS), lds_bytes, stream, args);
    } else {
        const size_t lds_bytes = LayoutOf<G, false>::histogram_at(n_barcode) + histogram_bytes;
        if(const int status = allow_large_lds(&pamld_pair_kernel<G, STATE, false>, prepared)) { return status; }
        hipLaunchKernelGGL((pamld_pair_kernel<G, STATE, false>), grid, dim3(THREADS), lds_bytes, stream, args);
    }
    return static_cast<int>(hipGetLastError());
}

template <int G>
int launch_one(const StageArgs& args, int n_barcode, bool lists, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(!args.first_stage) { return launch_state<G, 2>(args, n_barcode, lists, grid, histogram_bytes, stream); }
    return args.input_flag ? launch_state<G, 1>(args, n_barcode, lists, grid, histogram_bytes, stream)
                           : launch_state<G, 0>(args, n_barcode, lists, grid, histogram_bytes, stream);
}

}  // namespace pair
}  // namespace phq
