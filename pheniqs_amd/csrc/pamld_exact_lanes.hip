// K1x, the exact tier with SEVERAL LANES PER READ (round 4): what pamld_cross_exact_kernel decides, for the few reads
// (0.2 - 1 % of a batch) the single precision tiers hand on, laid out for a worklist that short.
//
//   reference pamld.cpp:37-123   posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164  P(r|b) = pow(base, Kahan sum of the Phred costs), distance, high quality distance
//
// The tier it replaces gives a read to ONE lane: 96 (x 2) fp64 products per lane out of 16 KiB of per read tables a
// group of four positions — two wave workgroups, three of them on a CU, the lanes of a tile waiting for its slowest
// read, every segment of the worklist a dependent chain of global loads; it ran at a tenth of what its arithmetic costs
// (C2: 0.19 ms for 0.58 % of 100 M reads, C3: 0.65 ms for 0.66 %).  Here R = 8 lanes share a read:
//   * the read's tables (16 fp64 entries a group of four positions, as before) are built ONCE by the eight lanes — two
//     entries each — into 128 bytes a group of the wave's LDS: 8 reads a wave, 1 - 4 KiB a wave, sixteen waves a CU;
//   * the words of a segment are dealt to the lanes (word u to lane u % 8): a table read a group and a multiplication
//     less per word, the sum, the maximum and the list of words within 1e-10 of it per lane, folded over the eight
//     lanes by three butterfly steps;
//   * the combinations of listed words are dealt to the lanes again and evaluated LITERALLY — the reference's Kahan
//     recurrence over the barcode's positions and the 2^x of the literal kernel — and compared as the reference's loop
//     compares them: strict '>' in barcode order, so the lowest ordinal among equal maxima wins;
//   * every wave owns whole segments of the worklist (a group of them when the batch is small), reads its entries two
//     iterations ahead and their rows one ahead, and compacts what it cannot decide either — a confidence within 1e-12
//     of its threshold, more listed words than the list holds — to the front of the read's own segment for the literal
//     kernel: no barrier after the set up, nothing shared between waves but the counter histogram.
// Serves single segment sets (up to EXACT_SINGLE_UNIQUE words of up to 16 nt — an image of EXACT_SINGLE_MEDIUM where that holds them —, priors that differ or not) and two segment
// combinatorial sets, with uniform priors or (WEIGHTED, below) priors that differ; three segment sets keep pamld_cross.hpp's kernel.
#include <algorithm>
#include <cstdlib>

#include "kernel_common.hpp"

#ifndef PHQ_WEIGHTED_LANES
#define PHQ_WEIGHTED_LANES 8
#endif
#ifndef PHQ_WEIGHTED_CAP
#define PHQ_WEIGHTED_CAP 48
#endif
#ifndef PHQ_WEIGHTED_PAIRS
#define PHQ_WEIGHTED_PAIRS 256
#endif
#ifndef PHQ_LANES_TRIP
#define PHQ_LANES_TRIP 4
#endif

namespace phq {
namespace lanes {

constexpr int WAVES = 4;
constexpr int THREADS = 64 * WAVES;
constexpr int WEIGHTED_CAP = PHQ_WEIGHTED_CAP;           // listed words of a segment in the weighted form
constexpr int CAP = 32;                    // listed words of a segment a read may bring to the literal walk
constexpr int GROUP_MAX = EXACT_GROUP_MAX;
// Round 5, tied noise reads (pamld_lut.hpp, NEAR_TOP): an entry with bit 31 set is a read handed on for nothing but a tied argmax,
// clearly under the noise filter, whose placeholder record names the two blocks of eight barcodes that hold every candidate.  A
// third to a half of the noise reads tie — 52 % of what reached this tier on C2, 97 % on H94 — and the scan below took every word
// of the codec through its fp64 tables for each.  A first pass settles them instead: a wave goes through its segments once, the
// entries without the bit compacted to the segment's front (in order), the flagged ones queued in LDS; whenever a wave's worth is
// queued every LANE takes one: the Kahan recurrence and the careful 2^x of the sixteen members, TIE_PACE of them side by side,
// compared as the reference compares them.  (As a kernel of its own between the product form and this tier the pass cost 0.06 ms
// with next to nothing to do; a workgroup per segment 0.11; re-scanning the codec in single precision for the candidates, with
// nothing kept by the scan in front: as much as this tier.  Behind the STREAMING kernel — whose second pass can flag the same
// reads — it gains nothing: C2 0.164 -> 0.170 ms, C5 0.088 -> 0.092 a decoder: not wired there.)
constexpr int TIE_PACE = 4;                // members walked side by side: a walk is one chain of dependent fp64 operations
constexpr int TIE_AHEAD = 2;               // entries a lane reads of a tile: their loads are in flight together
constexpr int TIE_FOLD = 8;                // pamld_lut.hpp's blocks

// pamld_cross.hpp's conventions: a position's low code bit at 8 (j / 4) + 2 + j % 4 of a word, its high bit four places up
constexpr int FACTOR_ROWS = 129;           // 0..127 = quality of an ACGT base, 128 = a base outside ACGT with q >= 1
__host__ __device__ constexpr uint32_t low_bit(int position) { return 8u * (position >> 2) + 2u + (position & 3); }
__device__ __forceinline__ uint32_t mismatch_fields(uint32_t x) { return x | __builtin_amdgcn_alignbit(x, x, 4); }      // x | rotr(x, 4)

__device__ __forceinline__ double shuffle_xor(double v, int mask) {
    const int lo = __shfl_xor(__double2loint(v), mask, 64);
    const int hi = __shfl_xor(__double2hiint(v), mask, 64);
    return __hiloint2double(hi, lo);
}

// WORDS = distinct words a segment may hold (CROSS_MAX_UNIQUE; a single segment set may hold EXACT_SINGLE_UNIQUE), R = lanes of
// a read: 4 where a lane's share of the words fits one mask word, 8 for the long lists
// WEIGHTED: a two segment combinatorial set whose priors differ between the barcodes (what pamld_cross_exact_weighted_kernel
// and, behind it, pamld_cross_exact_dense_kernel decided): sum_b prior_b P(r|b) is the bilinear form
// sum_u0 P0[u0] (sum_u1 prior[u0][u1] P1[u1]) over the words that matter — a first scan of both segments for their sums and
// maxima, a second one that lists (word, P) of every word within 1e-15 of its segment's maximum (or, for a read no barcode of
// which can pass the noise filter, the words that can still hold the argmax), the pairs of listed words dealt to the read's
// lanes with their priors from L2, what the lists leave out bounded at the largest prior, the pairs within 1e-10 of the
// largest prior P0 P1 literally.  A read with more listed words than WEIGHTED_CAP stays in the worklist for the dense kernel.
template <int G, int S, bool PRIORS, int WORDS, int R, bool WEIGHTED = false>
// (four word groups of two segments — 13 to 16 nt each — keep 2 waves a SIMD whatever is asked for: their tables take the registers)
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu((WEIGHTED || (G == 4 && S == 2)) ? 2 : 3, 4))) void pamld_exact_lanes_kernel(StageArgs args) {
    static_assert(!WEIGHTED || (S == 2 && !PRIORS), "the weighted form is the two segment one; PRIORS is priors by word of a single segment set");
    static_assert(S == 1 || S == 2, "single segment sets and two segment combinatorial sets");
    static_assert(!PRIORS || S == 1, "priors by word are for single segment sets");
    static_assert(WORDS == CROSS_MAX_UNIQUE || S == 1, "long word lists are for single segment sets");
    static_assert(R == 4 || R == 8, "a read's lanes build its 16 entry tables: four or two entries each");
    constexpr int READS = 64 / R;                                               // reads of a wave's iteration
    constexpr int MASKS = (WORDS / R + 31) / 32;                                // words of a lane's list mask
    constexpr int TRIP = PHQ_LANES_TRIP;                                        // words of a lane in flight in the scan
    constexpr int ROW_WORDS = 2 * S * (G + 1) + 1;                              // what a read brings: code and quality words, carried state
    __shared__ double factor_double[FACTOR_ROWS][2];                            // probabilities [row][0 match | 1 mismatch]
    __shared__ double cost_double[FACTOR_ROWS][2];                              // Phred costs, same layout (phred.cpp:44-68)
    __shared__ uint32_t lds_word[S * WORDS];
    __shared__ int32_t lds_ordinal[S == 1 ? WORDS : 1];
    __shared__ __attribute__((aligned(128))) double tables[WAVES][READS][G * 16];                             // this iteration's reads: entry m of group g
    __shared__ uint16_t listed[WAVES][READS][S][CAP];
    __shared__ uint32_t listed_count[WAVES][READS][S];
    __shared__ uint32_t group_start[WAVES][GROUP_MAX + 1];                      // prefix sums of the wave's segment counts
    __shared__ uint32_t group_kept[WAVES][GROUP_MAX];                           // reads left in each segment for the literal tier
    __shared__ uint32_t staged[WAVES][READS][ROW_WORDS + (ROW_WORDS % 2 ? 0 : 1)];   // the rows of the iteration in hand (odd stride)
    // what a read keeps from one phase of an iteration to the next goes through LDS, not registers (the kernel is short of
    // those: 128 a lane buy the fourth wave of a SIMD): a segment's factor rows and flags between the scan and the literal
    // walk, and what the iteration decided until the next one stores it
    __shared__ uint32_t kept[WAVES][READS][S][G + 3];
    __shared__ uint32_t decided[WAVES][READS][11];
    constexpr bool TIES = S == 1 && !PRIORS && !WEIGHTED;                      // the variants behind kernels that flag tied noise reads
    __shared__ uint32_t tie_queue[TIES ? WAVES : 1][TIES ? (TIE_AHEAD + 1) * 64 : 1];
    __shared__ double weighted_value[WEIGHTED ? WAVES : 1][WEIGHTED ? READS : 1][2][WEIGHTED ? WEIGHTED_CAP : 1];      // P of the listed words
    __shared__ uint16_t weighted_word[WEIGHTED ? WAVES : 1][WEIGHTED ? READS : 1][2][WEIGHTED ? WEIGHTED_CAP : 1];     // their ordinals
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* const histogram = reinterpret_cast<unsigned long long*>(lds_raw);
    DecoderRef d = *constant_cast(args.decoder);

    for(int i = threadIdx.x; i < FACTOR_ROWS; i += THREADS) {
        factor_double[i][0] = i < 128 ? d.match_probability[i] : d.uniform_probability;
        factor_double[i][1] = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        cost_double[i][0] = i < 128 ? d.true_positive_quality[i] : d.uniform_base_quality;
        cost_double[i][1] = i < 128 ? static_cast<double>(i) : d.uniform_base_quality;
    }
    // (the words of a single segment set may be a list of 384 or of 512: nothing is read past the segment's own)
    for(int w = threadIdx.x; w < S * WORDS; w += THREADS) { lds_word[w] = (w % WORDS) < d.cross_unique[w / WORDS] ? d.cross_words[(w / WORDS) * CROSS_MAX_UNIQUE + (w % WORDS)] : 0u; }
    if(S == 1) {
        for(int w = threadIdx.x; w < WORDS; w += THREADS) { lds_ordinal[w] = w < d.cross_unique[0] ? d.cross_ordinal[w * d.cross_stride[0]] : 0; }
    }
    const int wave = static_cast<int>(threadIdx.x >> 6);
    const int lane = static_cast<int>(threadIdx.x & 63u);
    const int slot = lane / R;                 // which read of the iteration
    const int sub = lane % R;                  // which of its lanes
    // the segments of this wave, as one list
    const int group = args.worklist_group < 1 ? 1 : (args.worklist_group > GROUP_MAX ? GROUP_MAX : args.worklist_group);
    const int first_segment = (static_cast<int>(blockIdx.x) * WAVES + wave) * group;
    const int n_segment_here = max(0, min(group, args.worklist_segments - first_segment));
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    uint32_t* const segments = args.worklist + static_cast<int64_t>(first_segment) * args.worklist_stride;
    int64_t n = 0;                             // entries of the wave's segments, as one list (known behind the first pass)

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
    constexpr double near = 1.0 - 1e-10;
    const double uniform_concentration = d.concentration[0];          // !PRIORS: every barcode's prior is this one
    const char* const factor_bytes = reinterpret_cast<const char*>(&factor_double[0][0]);
    const char* const cost_bytes = reinterpret_cast<const char*>(&cost_double[0][0]);
    const int high_quality_threshold = min(max(d.high_quality_threshold, 0), 128);
    const uint32_t high_quality_addend = static_cast<uint32_t>(128 - high_quality_threshold) * 0x01010101u;
    double* const my_tables = &tables[wave][slot][0];
    typedef __attribute__((address_space(3))) const double* LdsDouble;
    // the read's tables start on 128 bytes (checked): an entry's address is (flags moved to bits 3..6) | table_at
    const uint32_t table_at = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) double*)my_tables));
    if((table_at & 127u) != 0) { __builtin_trap(); }
    const uint32_t rotation = static_cast<uint32_t>(slot) << 3;

    // entry k of the wave's list: in which segment, and which read
    auto entry_of = [&](int64_t iteration, int& home) {
        const int64_t k = iteration * READS + slot;
        const uint32_t at = static_cast<uint32_t>(k < n ? k : n - 1);
        home = 0;
#pragma unroll
        for(int s = 1; s < GROUP_MAX; ++s) { home += at >= group_start[wave][s] ? 1 : 0; }       // starts are non decreasing
        return segments[static_cast<int64_t>(home) * args.worklist_stride + (at - group_start[wave][home])];
    };
    // a read's row set — S x (G + 1) code words, as many quality words, the carried state — is fetched by its lanes in shares
    // (word k by lane k % R) and handed round through LDS: ROW_WORDS / R registers of prefetch instead of ROW_WORDS
    constexpr int SHARE = (ROW_WORDS + R - 1) / R;
    struct Share { uint32_t w[SHARE]; };
    auto share_of = [&](int64_t i, Share& to) {
#pragma unroll
        for(int t = 0; t < SHARE; ++t) {
            const int k = sub + R * t;
            uint32_t v = 0;
            if(k < 2 * S * (G + 1)) {
                const int plane = k / (S * (G + 1));
                const int within = k % (S * (G + 1));
                const int s = within / (G + 1);
                const int w = min((S > 1 && s == 1 ? first_word[S - 1] : first_word[0]) + within % (G + 1), row_words - 1);
                v = reinterpret_cast<const uint32_t*>((plane ? args.quality : args.base) + i * stride)[w];
            } else if(k == ROW_WORDS - 1) {
                v = args.first_stage ? (args.input_flag ? static_cast<uint32_t>(args.input_flag[i]) : 0u) : args.word[i];
            }
            to.w[t] = v;
        }
    };
    struct Rows { uint32_t c[S][G + 1]; uint32_t q[S][G + 1]; uint32_t state; };

    // What an iteration decides is stored and counted at the top of the NEXT one, behind that iteration's wait for its rows:
    // the stores and atomics then have a whole iteration to leave before anything waits on the memory counter again (it
    // counts loads, stores and atomics in one order: stores issued at the end of an iteration were waited for, every
    // iteration, by the next one's first use of its prefetched rows).
    struct Pending { bool writer, defer, is_noise; int best, distance, hq_distance, home; double confidence; ReadState r; int64_t i; };
    auto put_off = [&](const Pending& p) {
        if(sub == 0) {
            uint32_t* const to = &decided[wave][slot][0];
            to[0] = (p.writer ? 1u : 0u) | (p.defer ? 2u : 0u) | (p.is_noise ? 4u : 0u) | (p.r.qcfail ? 8u : 0u);
            to[1] = static_cast<uint32_t>(p.best);
            to[2] = static_cast<uint32_t>(p.distance);
            to[3] = static_cast<uint32_t>(p.hq_distance);
            to[4] = static_cast<uint32_t>(p.home);
            to[5] = static_cast<uint32_t>(__double2loint(p.confidence));
            to[6] = static_cast<uint32_t>(__double2hiint(p.confidence));
            to[7] = static_cast<uint32_t>(p.r.channel);
            to[8] = static_cast<uint32_t>(p.i);
            to[9] = static_cast<uint32_t>(p.i >> 32);
        }
    };
    auto taken_up = [&]() {
        const uint32_t* const from = &decided[wave][slot][0];
        Pending p;
        const uint32_t bits = from[0];
        p.writer = (bits & 1u) != 0 && sub == 0; p.defer = (bits & 2u) != 0; p.is_noise = (bits & 4u) != 0; p.r.qcfail = (bits & 8u) != 0;
        p.best = static_cast<int>(from[1]); p.distance = static_cast<int>(from[2]); p.hq_distance = static_cast<int>(from[3]); p.home = static_cast<int>(from[4]);
        p.confidence = __hiloint2double(static_cast<int>(from[6]), static_cast<int>(from[5]));
        p.r.channel = static_cast<int32_t>(from[7]);
        p.i = static_cast<int64_t>(from[8]) | (static_cast<int64_t>(from[9]) << 32);
        return p;
    };
    auto commit = [&](const Pending& decided) {
        const int64_t i = decided.i;
        ReadState r = decided.r;
        if(!decided.defer && decided.writer) {
            int bin = decided.best + 1;
            int distance = decided.distance;
            int hq_distance = decided.hq_distance;
            double confidence = decided.confidence;
            uint8_t status = STATUS_PASS;
            if(!decided.is_noise) {
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
            // Decoder<T>::classify then Classifier<T>::classify (decoder.h:68-76, classifier.h:78-86): decoder_tail with the
            // barcode's index taken from the bin where the compiler numbered the barcodes (no load to wait for)
            const int index = bin > 0 ? (d.identity_index ? bin : constant_cast(d.barcode_index)[bin - 1]) : 0;
            {
                const int pf = r.qcfail ? 0 : 1;
                if(!(sink.defer && bin > 0)) {
                    if(bin > 0 && distance) { stat_add(d, sink, bin, pf, FIELD_DISTANCE, static_cast<unsigned long long>(distance)); }
                    stat_add(d, sink, bin, pf, FIELD_COUNT, 1ull);
                }
                if(d.multiplexing) { r.channel = index; }
            }
            if(args.result) {
                DevResult out;
                out.index = index;
                out.distance = static_cast<uint8_t>(distance > 255 ? 255 : distance);
                out.hq_distance = static_cast<uint8_t>(hq_distance > 255 ? 255 : hq_distance);
                out.status = status;
                out.qcfail = r.qcfail ? 1 : 0;
                out.confidence = confidence;
                args.result[i] = out;
            }
            store_read_state(args, i, r);
        }
        // what is left goes back to the front of the read's own segment: a segment never holds more leftovers than entries
        // already read from it, and the entries of the iterations in hand are in registers
        if(decided.writer && decided.defer) {
            const uint32_t at = atomicAdd(&group_kept[wave][decided.home], 1u);
            segments[static_cast<int64_t>(decided.home) * args.worklist_stride + at] = static_cast<uint32_t>(i);
        }
        count_outgoing(args, histogram, decided.writer && !decided.defer, r);
    };

    // the first pass: tied noise reads (TIE_PACE, above)
    if constexpr(TIES) {
        if(args.flag_ties != 0) {
            uint32_t* const queue = tie_queue[wave];
            const double common_prior = uniform_concentration;
            uint32_t queued = 0;                   // wave uniform, as is
            for(int home = 0; home < n_segment_here; ++home) {
                uint32_t* const segment = segments + static_cast<int64_t>(home) * args.worklist_stride;
                const int64_t entries = static_cast<int64_t>(args.worklist_count[first_segment + home]);
                uint32_t kept_total = 0;           // what the wave leaves in this segment
                auto keep = [&](bool stays, uint32_t entry) {
                    const unsigned long long staying = __ballot(stays);
                    if(stays) { segment[kept_total + __popcll(staying & ((1ull << lane) - 1ull))] = entry & 0x7fffffffu; }
                    kept_total += static_cast<uint32_t>(__popcll(staying));
                };
                // queue[first .. first + waiting): a read a lane
                auto settle = [&](uint32_t first, uint32_t waiting) {
                    const bool active = static_cast<uint32_t>(lane) < waiting;
                    const uint32_t entry = queue[first + (active ? static_cast<uint32_t>(lane) : 0u)];
                    const int64_t i = static_cast<int64_t>(entry);        // (a lane past the end takes the first read again, unheard)
                    ReadState r = load_read_state(args, i);
                    const uint32_t blocks = static_cast<uint32_t>(__double2loint(args.result[i].confidence));
                    const uint32_t* const code_row = reinterpret_cast<const uint32_t*>(args.base + i * stride);
                    const uint32_t* const quality_row = reinterpret_cast<const uint32_t*>(args.quality + i * stride);
                    uint32_t c[G + 1], q[G + 1];
#pragma unroll
                    for(int g = 0; g <= G; ++g) {
                        const int w = min(first_word[0] + g, row_words - 1);
                        c[g] = code_row[w];
                        q[g] = quality_row[w];
                    }
                    // per position a byte: the row of its cost pair; its 2 bit code at the words' places; the positions outside ACGT
                    uint32_t row[G];
                    uint32_t observed = 0, outside = 0;
#pragma unroll
                    for(int g = 0; g < G; ++g) {
                        const uint32_t code_word = __builtin_amdgcn_alignbyte(c[g + 1], c[g], byte_shift[0]) & byte_mask[0][g];
                        const uint32_t quality_word = __builtin_amdgcn_alignbyte(q[g + 1], q[g], byte_shift[0]) & byte_mask[0][g];
                        row[g] = 0;
#pragma unroll
                        for(int k = 0; k < 4; ++k) {
                            const uint32_t code = (code_word >> (8 * k)) & 0xfu;
                            const uint32_t quality = (quality_word >> (8 * k)) & 0x7fu;          // the table has 128 rows
                            const bool strict = strict_code(code);
                            const bool in_word = 4 * g + k < length[0];
                            const uint32_t two = (0x30210u >> (2 * code)) & 3u;                  // A, C, G, T (1, 2, 4, 8) -> 0 .. 3
                            observed |= ((two & 1u) << low_bit(4 * g + k)) | ((two >> 1) << ((low_bit(4 * g + k) + 4u) & 31u));
                            outside |= (in_word && !strict) ? 1u << low_bit(4 * g + k) : 0u;
                            row[g] |= (strict ? quality : (quality ? 128u : 0u)) << (8 * k);
                        }
                    }
                    const int block_of[2] = {static_cast<int>(blocks & 0x1fffu), static_cast<int>((blocks >> 13) & 0x1fffu)};
                    const int members = ((blocks >> 26) & 1u) ? 2 * TIE_FOLD : TIE_FOLD;
                    int best = 0x7fffffff;
                    double best_p = -1.0;
                    double best_conditional = 0;
#pragma nounroll
                    for(int m = 0; m < members; m += TIE_PACE) {
                        int b[TIE_PACE];
                        uint32_t flags[TIE_PACE];
                        double sigma_q[TIE_PACE];
                        double compensation[TIE_PACE];
#pragma unroll
                        for(int u = 0; u < TIE_PACE; ++u) {
                            b[u] = TIE_FOLD * block_of[(m + u) / TIE_FOLD] + ((m + u) % TIE_FOLD);
                            flags[u] = (mismatch_fields(observed ^ lds_word[b[u] < unique[0] ? b[u] : 0]) | outside) & full_mask[0];
                            sigma_q[u] = 0;
                            compensation[u] = 0;
                        }
                        // barcode.h:131-164: the Kahan sum of the costs over the barcode's positions, then pow
#pragma unroll
                        for(int j = 0; j < 4 * G; ++j) {
                            if(j < length[0]) {
                                const uint32_t at = ((row[j >> 2] >> (8 * (j & 3))) & 0xffu) * 16u;
                                const double match = *reinterpret_cast<const double*>(cost_bytes + at);
                                const double mismatch = *reinterpret_cast<const double*>(cost_bytes + at + 8u);
#pragma unroll
                                for(int u = 0; u < TIE_PACE; ++u) {
                                    const double x = ((flags[u] >> low_bit(j)) & 1u) ? mismatch : match;
                                    const double y = x - compensation[u];
                                    const double t = sigma_q[u] + y;
                                    compensation[u] = (t - sigma_q[u]) - y;
                                    sigma_q[u] = t;
                                }
                            }
                        }
#pragma unroll
                        for(int u = 0; u < TIE_PACE; ++u) {
                            const double conditional = phred_probability(sigma_q[u], d.log2_base_hi, d.log2_base_lo);
                            const double p = conditional * common_prior;                                   // pamld.cpp:66
                            if(b[u] < unique[0] && (p > best_p || (p == best_p && b[u] < best))) {           // strict '>' in barcode order: the lowest of equals
                                best = b[u]; best_p = p; best_conditional = conditional;
                            }
                        }
                    }
                    // a noise read on the literal value as well (pamld.cpp:96): charged to its argmax, it leaves unclassified
                    const bool settled = active && best != 0x7fffffff && !(best_conditional > d.random_barcode_probability);
                    if(settled) {
                        stat_add(d, sink, best + 1, 0, FIELD_LOW_CONDITIONAL_CONFIDENCE, 1ull);
                        r.qcfail = true;
                        decoder_tail(d, sink, r, 0, 0);
                        DevResult out;
                        out.index = 0; out.distance = 0; out.hq_distance = 0; out.status = STATUS_NOISE; out.qcfail = 1; out.confidence = 0;
                        args.result[i] = out;
                        store_read_state(args, i, r);
                    }
                    keep(active && !settled, entry);
                    count_outgoing(args, histogram, active && settled, r);
                };
                for(int64_t tile = 0; tile * (TIE_AHEAD * 64) < entries; ++tile) {
                    uint32_t entry[TIE_AHEAD];
                    bool active[TIE_AHEAD];
#pragma unroll
                    for(int e = 0; e < TIE_AHEAD; ++e) {
                        const int64_t k = (tile * TIE_AHEAD + e) * 64 + lane;
                        active[e] = k < entries;
                        entry[e] = active[e] ? segment[k] : 0u;
                    }
                    __builtin_amdgcn_s_waitcnt(0x0f70);        // every entry of the tile is read before anything is written to the segment's front
#pragma unroll
                    for(int e = 0; e < TIE_AHEAD; ++e) {
                        const bool flagged = active[e] && (entry[e] >> 31) != 0;
                        keep(active[e] && !flagged, entry[e]);
                        const unsigned long long queueing = __ballot(flagged);
                        if(flagged) { queue[queued + __popcll(queueing & ((1ull << lane) - 1ull))] = entry[e] & 0x7fffffffu; }
                        queued += static_cast<uint32_t>(__popcll(queueing));
                    }
                    __builtin_amdgcn_wave_barrier();           // (a wave's LDS operations complete in order)
                    while(queued >= 64u) {                     // (under a wave's worth was waiting, a tile brings TIE_AHEAD of them at most)
                        queued -= 64u;
                        settle(queued, 64u);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                // what is still queued goes on into the wave's next segment; behind the last one it is settled
                if(home == n_segment_here - 1 && queued > 0) { settle(0, queued); queued = 0; }
                if(lane == 0) { args.worklist_count[first_segment + home] = kept_total; }
            }
            // the lists as the pass left them, before the wave reads them again: a wave's own stores, through the CU's own L1 — a
            // fence of the workgroup's scope (the device's, __threadfence, writes the L2 back: 0.08 ms a launch, measured)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
    }
    if(lane == 0) {
        uint32_t total = 0;
        for(int s = 0; s < GROUP_MAX; ++s) {
            group_start[wave][s] = total;
            group_kept[wave][s] = 0;
            if(s < n_segment_here) { total += args.worklist_count[first_segment + s]; }
        }
        group_start[wave][GROUP_MAX] = total;
    }
    __builtin_amdgcn_wave_barrier();
    n = static_cast<int64_t>(group_start[wave][GROUP_MAX]);
    const int64_t iterations = (n + READS - 1) / READS;

    if(iterations > 0) {
        int home_now = 0, home_next = 0;
        uint32_t entry_now = entry_of(0, home_now);
        uint32_t entry_next = entry_of(iterations > 1 ? 1 : 0, home_next);
        Share share_now;
        share_of(static_cast<int64_t>(entry_now), share_now);
        if(sub == 0) { decided[wave][slot][0] = 0; }          // nothing decided yet
        __builtin_amdgcn_wave_barrier();
        for(int64_t iteration = 0; iteration < iterations; ++iteration) {
            const bool active = iteration * READS + slot < n;
            const int home = home_now;
            const int64_t i = static_cast<int64_t>(entry_now);
            Rows rows;
            {
                uint32_t* const mine = &staged[wave][slot][0];
#pragma unroll
                for(int t = 0; t < SHARE; ++t) { if(sub + R * t < ROW_WORDS) { mine[sub + R * t] = share_now.w[t]; } }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for(int s = 0; s < S; ++s) {
#pragma unroll
                    for(int g = 0; g <= G; ++g) {
                        rows.c[s][g] = mine[s * (G + 1) + g];
                        rows.q[s][g] = mine[S * (G + 1) + s * (G + 1) + g];
                    }
                }
                rows.state = mine[ROW_WORDS - 1];
                __builtin_amdgcn_wave_barrier();
            }
            commit(taken_up());                    // the iteration before: its stores have this iteration to leave
            // the entries two iterations ahead, the rows one ahead: their latency hides behind this iteration's work
            int home_after = 0;
            const uint32_t entry_after = entry_of(iteration + 2 < iterations ? iteration + 2 : iterations - 1, home_after);
            share_of(static_cast<int64_t>(entry_next), share_now);
            entry_now = entry_next; home_now = home_next;
            entry_next = entry_after; home_next = home_after;

            ReadState r;
            if(args.first_stage) { r.qcfail = (rows.state & 1u) != 0; r.channel = 0; }       // transcode.h:207-211
            else { r.qcfail = (rows.state >> 31) != 0; r.channel = static_cast<int32_t>(rows.state & 0x7fffffffu); }

            uint32_t row[S][G];                    // per position a byte: the row of its factor / cost pair
            uint32_t observed[S];
            uint32_t outside[S];
            uint32_t high_quality[S];
            double sum_all = 1.0;                  // sum over every barcode of (prior) P(r|b): uniform priors, the product of the per segment sums
            bool defer = false;
            if(sub < S) { listed_count[wave][slot][sub < S ? sub : 0] = 0; }
            __builtin_amdgcn_wave_barrier();       // the lanes of a read talk through LDS: a wave's LDS operations complete in order

#pragma unroll
            for(int s = 0; s < S; ++s) {
                observed[s] = 0;
                outside[s] = 0;
                high_quality[s] = 0;
                auto gather = [](uint32_t planes) {            // bits 8 j and 8 j + 4 -> bits j and 4 + j
                    const uint32_t t = planes | (planes >> 7);
                    return (t | (t >> 14)) & 0xffu;
                };
#pragma unroll
                for(int g = 0; g < G; ++g) {
                    // per position, four at a time and byte parallel (pamld_cross.hpp): factor row, 2-bit base, flags
                    const uint32_t valid = byte_mask[s][g];
                    const uint32_t code_word = __builtin_amdgcn_alignbyte(rows.c[s][g + 1], rows.c[s][g], byte_shift[s]) & valid;
                    const uint32_t quality_word = __builtin_amdgcn_alignbyte(rows.q[s][g + 1], rows.q[s][g], byte_shift[s]) & valid;
                    const uint32_t x = code_word & 0x0f0f0f0fu;
                    const uint32_t q8 = quality_word;
                    const uint32_t q7 = q8 & 0x7f7f7f7fu;
                    const uint32_t z = x & ((x | 0x80808080u) - 0x01010101u);
                    const uint32_t strict = (x + 0x7f7f7f7fu) & ~(z + 0x7f7f7f7fu) & 0x80808080u;
                    const uint32_t strict_bytes = (strict - (strict >> 7)) | strict;
                    // outside ACGT: the uniform quality against every barcode base (row 128), nothing at quality 0 or for an
                    // absent position (row 0 holds 1 and 1)   phred.cpp:62-64
                    const uint32_t nonzero = (q7 + 0x7f7f7f7fu) & 0x80808080u;
                    row[s][g] = (q7 & strict_bytes) | (nonzero & ~strict_bytes);
                    const uint32_t codes = gather((((x >> 1) | (x >> 3)) & 0x01010101u) | (((x << 2) | (x << 1)) & 0x10101010u));
                    observed[s] |= __builtin_amdgcn_alignbit(codes, codes, (32 - (8 * g + 2)) & 31);
                    const uint32_t high = ((q7 + high_quality_addend) | q8) & 0x80808080u & valid;
                    const uint32_t flags = gather(((~strict & 0x80808080u & valid) >> 7) | (high >> 3));
                    outside[s] |= (flags & 0xfu) << (8 * g + 2);
                    high_quality[s] |= (flags >> 4) << (8 * g + 2);
                }
                // what the literal walk reads again: nothing of the scans stays in registers for it
                if(sub == 0) {
#pragma unroll
                    for(int g = 0; g < G; ++g) { kept[wave][slot][s][g] = row[s][g]; }
                    kept[wave][slot][s][G] = observed[s];
                    kept[wave][slot][s][G + 1] = outside[s];
                    kept[wave][slot][s][G + 2] = high_quality[s];
                }
            }
            // this read's tables of segment s, 16 / R entries a lane: entry m of group g = product over the group's positions of the
            // match (bit clear) or mismatch (bit set) factor.  A lane's entries m = sub + R half share their two low positions:
            // one product for those, one for the two high positions of each entry ((p0 p1) (p2 p3): associated by pair)
            auto build_tables = [&](auto which) {
                constexpr int s = decltype(which)::value;
                __builtin_amdgcn_wave_barrier();   // the tables in place are read no more
#pragma unroll
                for(int g = 0; g < G; ++g) {
                    auto factor = [&](int jj, uint32_t bit) {
                        return *reinterpret_cast<const double*>(factor_bytes + (((row[s][g] >> (8 * jj)) & 0xffu) * 16u + bit * 8u));
                    };
                    const double low_pair = factor(0, static_cast<uint32_t>(sub) & 1u) * factor(1, (static_cast<uint32_t>(sub) >> 1) & 1u);
#pragma unroll
                    for(int half = 0; half < 16 / R; ++half) {
                        const uint32_t m = static_cast<uint32_t>(sub + R * half);
                        const double entry = low_pair * (factor(2, (m >> 2) & 1u) * factor(3, m >> 3));
                        my_tables[g * 16 + static_cast<int>((m + static_cast<uint32_t>(slot)) & 15u)] = entry;      // rotated by the slot: see probability_of
                    }
                }
                __builtin_amdgcn_wave_barrier();
            };
            // flags of group g sit at bits 8 g + 2 .. 8 g + 5: entry * 8 bytes = (flags moved to bits 3..6) & 0x78
            auto probability_of = [&](auto which, uint32_t word) {
                constexpr int s = decltype(which)::value;
                const uint32_t flags = (mismatch_fields(observed[s] ^ word) | outside[s]) & full_mask[s];
                double p = 1.0;
#pragma unroll
                for(int g = 0; g < G; ++g) {
                    // (the reads of a wave keep their tables 128 G bytes apart — the same banks: entry m sits at (m + slot) % 16)
                    const uint32_t moved = (g == 0 ? flags << 1 : flags >> (8 * g - 1)) + rotation;
                    const double t = *(LdsDouble)static_cast<uintptr_t>(__builtin_amdgcn_bitop3_b32(moved, 0x78u, table_at, 0xEA) + static_cast<uint32_t>(g) * 128u);      // (moved & 0x78) | table_at
                    p = g == 0 ? t : p * t;
                }
                return p;
            };
            // the words of segment s dealt to the read's lanes, several a trip — their table reads are in flight together (the
            // loop is a chain of LDS round trips otherwise): visit(word ordinal, ordinal / R, P) in ascending order of the lane's words
            auto for_each_word = [&](auto which, auto&& visit) {
                constexpr int s = decltype(which)::value;
                for(int first = 0, t = 0; first < unique[s]; first += TRIP * R, t += TRIP) {
                    uint32_t word[TRIP];
                    bool valid[TRIP];
                    double p[TRIP];
#pragma unroll
                    for(int h = 0; h < TRIP; ++h) {
                        const int u = first + h * R + sub;
                        valid[h] = u < unique[s];
                        word[h] = lds_word[s * WORDS + (valid[h] ? u : 0)];
                    }
#pragma unroll
                    for(int h = 0; h < TRIP; ++h) {
                        p[h] = probability_of(which, word[h]);
                        if(PRIORS) { p[h] *= d.concentration[lds_ordinal[S == 1 && valid[h] ? first + h * R + sub : 0]]; }
                        p[h] = valid[h] ? p[h] : 0.0;
                    }
#pragma unroll
                    for(int h = 0; h < TRIP; ++h) { visit(first + h * R + sub, t + h, p[h], valid[h]); }
                }
            };
            // P(r|b) of a combination LITERALLY: Kahan sum of the costs over the barcode's positions, then pow (barcode.h:131-164).
            // Position by position, a loop and not unrolled: unrolled, the scheduler hoists every position's cost read in front of
            // the recurrence and holds 2 x 4 G S registers for them; the factor rows come back from LDS byte by byte
            auto literal_of = [&](const uint32_t (&flags)[S]) {
                double sigma = 0;
                double compensation = 0;
#pragma unroll
                for(int s = 0; s < S; ++s) {
                    const uint8_t* const row_bytes = reinterpret_cast<const uint8_t*>(&kept[wave][slot][s][0]);
#pragma nounroll
                    for(int j = 0; j < length[s]; ++j) {
                        const uint32_t bit = (flags[s] >> (8 * (j >> 2) + 2 + (j & 3))) & 1u;
                        const double x = *reinterpret_cast<const double*>(cost_bytes + (static_cast<uint32_t>(row_bytes[j]) * 16u + bit * 8u));
                        const double y = x - compensation;
                        const double t = sigma + y;
                        compensation = (t - sigma) - y;
                        sigma = t;
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
            auto candidate = [&](int b, double prior, const uint32_t (&flags)[S]) {
                const double conditional = literal_of(flags);
                const double p = conditional * prior;                                  // pamld.cpp:66
                if(p > best_p || (p == best_p && b < best)) {                          // strict '>' in barcode order
                    best = b;
                    best_p = p;
                    best_conditional = conditional;
#pragma unroll
                    for(int s = 0; s < S; ++s) { best_flags[s] = flags[s]; }
                }
            };
            const std::integral_constant<int, 0> zero;
            const std::integral_constant<int, S - 1> last;

            if constexpr(!WEIGHTED) {
                auto scan = [&](auto which) {
                    constexpr int s = decltype(which)::value;
                    build_tables(which);
                    double top = 0;
                    double sum = 0;
                    uint32_t mask[MASKS];
#pragma unroll
                    for(int e = 0; e < MASKS; ++e) { mask[e] = 0u; }
                    for_each_word(which, [&](int, int t, double p, bool valid) {
                        sum += p;
                        const bool clears = p * near > top;
                        const bool lists = valid && p >= top * near;
#pragma unroll
                        for(int e = 0; e < MASKS; ++e) {
                            mask[e] = clears ? 0u : mask[e];
                            if((t >> 5) == e) { mask[e] |= lists ? 1u << (t & 31) : 0u; }
                        }
                        top = fmax(top, p);
                    });
                    // over the read's lanes: the sum, the maximum; a lane whose own maximum is outside the window of the read's lists nothing
                    double read_top = top;
#pragma unroll
                    for(int step = 1; step < R; step <<= 1) {
                        sum += shuffle_xor(sum, step);
                        read_top = fmax(read_top, shuffle_xor(read_top, step));
                    }
                    if(!(read_top > 0)) { defer = true; }
                    const bool mine = top >= read_top * near && read_top > 0;
#pragma unroll
                    for(int e = 0; e < MASKS; ++e) {
                        uint32_t m = mine ? mask[e] : 0u;
                        while(m) {
                            const int t = 32 * e + __ffs(static_cast<int>(m)) - 1;
                            m &= m - 1u;
                            const uint32_t at = atomicAdd(&listed_count[wave][slot][s], 1u);
                            if(at < static_cast<uint32_t>(CAP)) { listed[wave][slot][s][at] = static_cast<uint16_t>(sub + R * t); }
                        }
                    }
                    sum_all *= sum;
                };
                scan(zero);
                if constexpr(S > 1) { scan(last); }
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
                // the combinations of listed words, dealt to the read's lanes
                uint32_t count[S];
                uint32_t combinations = 1;
#pragma unroll
                for(int s = 0; s < S; ++s) {
                    count[s] = listed_count[wave][slot][s];
                    if(count[s] == 0 || count[s] > static_cast<uint32_t>(CAP)) { defer = true; count[s] = 0; }
                    combinations *= count[s];
                }
                for(uint32_t c = static_cast<uint32_t>(sub); c < combinations; c += R) {
                    uint32_t flags[S];
                    int flat = 0;
                    uint32_t rest = c;
#pragma unroll
                    for(int s = S - 1; s >= 0; --s) {
                        const uint32_t pick = S == 1 ? rest : rest % count[s];
                        rest = S == 1 ? 0u : rest / count[s];
                        const int u = listed[wave][slot][s][pick];
                        flat += u * radix[s];
                        flags[s] = (mismatch_fields(observed[s] ^ lds_word[s * WORDS + u]) | outside[s]) & full_mask[s];
                    }
                    const int b = S == 1 ? lds_ordinal[S == 1 ? flat / radix[0] : 0] : (d.cross_identity ? flat : d.cross_ordinal[flat]);
                    candidate(b, PRIORS ? d.concentration[b] : uniform_concentration, flags);
                }
            } else {
                // both segments: sum, maximum and where it is (the first word of the largest P)
                double top_of[2], sum_of[2];
                int first_of[2];
                auto survey = [&](auto which) {
                    constexpr int s = decltype(which)::value;
                    build_tables(which);
                    double top = 0;
                    double sum = 0;
                    int first = 0;
                    for_each_word(which, [&](int u, int, double p, bool) {
                        sum += p;
                        const bool larger = p > top;                     // a lane's words come in ascending order: the first of equals stays
                        first = larger ? u : first;
                        top = larger ? p : top;
                    });
#pragma unroll
                    for(int step = 1; step < R; step <<= 1) {
                        sum += shuffle_xor(sum, step);
                        const double other_top = shuffle_xor(top, step);
                        const int other_first = __shfl_xor(first, step, 64);
                        const bool takes = other_top > top || (other_top == top && other_first < first);
                        top = takes ? other_top : top;
                        first = takes ? other_first : first;
                    }
                    top_of[s] = top; sum_of[s] = sum; first_of[s] = first;
                };
                survey(zero);
                survey(last);
                if(!(top_of[0] > 0) || !(top_of[1] > 0)) { defer = true; }
                // A read no barcode of which can pass the noise filter (every P(r|b) <= top0 * top1 < the random barcode probability,
                // the literal values being within 1e-13 of these products) reports nothing but its argmax: then only words that can
                // still beat the pair of the two maxima are listed — prior P0 P1 >= prior_tt top0 top1 needs P0 >= top0 * prior_tt /
                // largest prior, and the same for P1 — and no sum is taken.  Otherwise: every word within 1e-15 of its segment's
                // maximum, the only ones that can move the sum by 1e-12 or be the argmax (pamld_cross_exact_weighted_kernel).
                const double prior_of_tops = d.cross_prior_double[first_of[0] * radix[0] + first_of[1] * radix[1]];
                const bool only_argmax = top_of[0] * top_of[1] < d.random_barcode_probability * (1.0 - 1e-9) && prior_of_tops > 0;
                const double keep = only_argmax ? (prior_of_tops / d.cross_prior_max) * (1.0 - 1e-9) : 1e-15;
                double listed_sum[2];
                auto list = [&](auto which) {
                    constexpr int s = decltype(which)::value;
                    build_tables(which);
                    double total = 0;
                    const double floor = top_of[s] * keep;
                    for_each_word(which, [&](int u, int, double p, bool valid) {
                        if(valid && p >= floor && p > 0) {
                            const uint32_t at = atomicAdd(&listed_count[wave][slot][s], 1u);
                            if(at < static_cast<uint32_t>(WEIGHTED_CAP)) {
                                weighted_value[wave][slot][s][at] = p;
                                weighted_word[wave][slot][s][at] = static_cast<uint16_t>(u);
                                total += p;
                            }
                        }
                    });
#pragma unroll
                    for(int step = 1; step < R; step <<= 1) { total += shuffle_xor(total, step); }
                    listed_sum[s] = total;
                };
                list(zero);
                list(last);
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
                uint32_t count[2];
#pragma unroll
                for(int s = 0; s < 2; ++s) {
                    count[s] = listed_count[wave][slot][s];
                    if(count[s] == 0 || count[s] > static_cast<uint32_t>(WEIGHTED_CAP)) { defer = true; count[s] = 0; }
                }
                // (a read with more pairs than this goes to the dense kernel as well: a wave would walk them for tens of microseconds
                // with one read's lanes busy)
                if(count[0] * count[1] > static_cast<uint32_t>(PHQ_WEIGHTED_PAIRS)) { defer = true; count[0] = 0; }
                const uint32_t pairs = count[0] * count[1];
                // the bilinear form over the listed words, and the largest prior * P0 * P1: the pairs dealt to the lanes
                double weighted_sum = 0;
                double weighted_top = 0;
                // four pairs a trip: their priors come from L2, one round trip a trip instead of one a pair
                for(uint32_t c = static_cast<uint32_t>(sub); c < pairs; c += 4 * R) {
                    double prior[4], value[4];
#pragma unroll
                    for(int h = 0; h < 4; ++h) {
                        const uint32_t at = c + static_cast<uint32_t>(h * R);
                        const bool valid = at < pairs;
                        const uint32_t k0 = valid ? at / count[1] : 0u, k1 = valid ? at % count[1] : 0u;
                        const int u0 = weighted_word[wave][slot][0][k0], u1 = weighted_word[wave][slot][1][k1];
                        prior[h] = d.cross_prior_double[u0 * radix[0] + u1 * radix[1]];
                        value[h] = valid ? weighted_value[wave][slot][1][k1] * weighted_value[wave][slot][0][k0] : 0.0;
                    }
#pragma unroll
                    for(int h = 0; h < 4; ++h) {
                        const double p = prior[h] * value[h];
                        weighted_sum += p;
                        weighted_top = fmax(weighted_top, p);
                    }
                }
#pragma unroll
                for(int step = 1; step < R; step <<= 1) {
                    weighted_sum += shuffle_xor(weighted_sum, step);
                    weighted_top = fmax(weighted_top, shuffle_xor(weighted_top, step));
                }
                // what the lists leave out, at most: the unlisted mass at the largest prior
                const double left_out = d.cross_prior_max * (fmax(sum_of[0] * sum_of[1] - listed_sum[0] * listed_sum[1], 0.0) + 1e-15 * sum_of[0] * sum_of[1]);
                if(!(weighted_top > 0)) { defer = true; }
                if(!only_argmax && !(left_out <= 1e-12 * (weighted_sum + d.adjusted_noise_probability))) { defer = true; }
                sum_all = weighted_sum;
                // the pairs within 1e-10 of the largest, literally, in barcode order
                const double floor = weighted_top * (1.0 - 1e-10);
                for(uint32_t c = static_cast<uint32_t>(sub); c < (defer ? 0u : pairs); c += 4 * R) {
                    double prior[4], value[4];
                    int word0[4], word1[4];
#pragma unroll
                    for(int h = 0; h < 4; ++h) {
                        const uint32_t at = c + static_cast<uint32_t>(h * R);
                        const bool valid = at < pairs;
                        const uint32_t k0 = valid ? at / count[1] : 0u, k1 = valid ? at % count[1] : 0u;
                        word0[h] = weighted_word[wave][slot][0][k0]; word1[h] = weighted_word[wave][slot][1][k1];
                        prior[h] = d.cross_prior_double[word0[h] * radix[0] + word1[h] * radix[1]];
                        value[h] = valid ? weighted_value[wave][slot][1][k1] * weighted_value[wave][slot][0][k0] : -1.0;
                    }
#pragma unroll
                    for(int h = 0; h < 4; ++h) {
                        if(!(value[h] >= 0 && prior[h] * value[h] >= floor)) { continue; }
                        const int flat = word0[h] * radix[0] + word1[h] * radix[1];
                        uint32_t flags[S];
                        flags[0] = (mismatch_fields(observed[0] ^ lds_word[word0[h]]) | outside[0]) & full_mask[0];
                        flags[S - 1] = (mismatch_fields(observed[S - 1] ^ lds_word[(S - 1) * WORDS + word1[h]]) | outside[S - 1]) & full_mask[S - 1];
                        candidate(d.cross_identity ? flat : d.cross_ordinal[flat], prior[h], flags);
                    }
                }
            }
            // ... folded over the read's lanes the same way
#pragma unroll
            for(int step = 1; step < R; step <<= 1) {
                const double other_p = shuffle_xor(best_p, step);
                const int other = __shfl_xor(best, step, 64);
                const double other_conditional = shuffle_xor(best_conditional, step);
                uint32_t other_flags[S];
#pragma unroll
                for(int s = 0; s < S; ++s) { other_flags[s] = static_cast<uint32_t>(__shfl_xor(static_cast<int>(best_flags[s]), step, 64)); }
                if(other_p > best_p || (other_p == best_p && other < best)) {
                    best = other;
                    best_p = other_p;
                    best_conditional = other_conditional;
#pragma unroll
                    for(int s = 0; s < S; ++s) { best_flags[s] = other_flags[s]; }
                }
            }
            if(best == 0x7fffffff) { defer = true; best = 0; best_conditional = 0; best_p = 0; }

            int distance = 0;
            int hq_distance = 0;
#pragma unroll
            for(int s = 0; s < S; ++s) {
                distance += __popc(best_flags[s]);
                hq_distance += __popc(best_flags[s] & high_quality[s]);
            }
            const double total = ((PRIORS || WEIGHTED) ? sum_all : sum_all * uniform_concentration) + d.adjusted_noise_probability;   // pamld.cpp:61-86
            const double confidence = best_p / total;                                    // pamld.cpp:92
            const bool is_noise = !(best_conditional > d.random_barcode_probability);    // pamld.cpp:96, on the literal value
            if(!is_noise && !(fabs(confidence - d.confidence_threshold) > 1e-12)) { defer = true; }
            Pending pending;
            pending.writer = active; pending.defer = defer; pending.is_noise = is_noise; pending.best = best; pending.distance = distance;
            pending.hq_distance = hq_distance; pending.confidence = confidence; pending.r = r; pending.i = i; pending.home = home;
            asm volatile("" ::: "memory");
            put_off(pending);
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
        }
        commit(taken_up());
    }
    if(lane < n_segment_here) { args.worklist_count[first_segment + lane] = group_kept[wave][lane]; }
    __syncthreads();
    histogram_flush(args, d, histogram);
}

template <int G, int S, bool PRIORS, int WORDS, int R>
int launch_form(const StageArgs& args, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    hipLaunchKernelGGL((pamld_exact_lanes_kernel<G, S, PRIORS, WORDS, R>), grid, dim3(THREADS), histogram_bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}

template <int S, bool PRIORS, int WORDS, int R>
int launch_groups(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.cross_groups) {
        case 1: return launch_form<1, S, PRIORS, WORDS, R>(args, grid, histogram_bytes, stream);
        case 2: return launch_form<2, S, PRIORS, WORDS, R>(args, grid, histogram_bytes, stream);
        case 3: return launch_form<3, S, PRIORS, WORDS, R>(args, grid, histogram_bytes, stream);
        case 4: return launch_form<4, S, PRIORS, WORDS, R>(args, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

template <int G>
int launch_weighted_form(const StageArgs& args, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    hipLaunchKernelGGL((pamld_exact_lanes_kernel<G, 2, false, CROSS_MAX_UNIQUE, PHQ_WEIGHTED_LANES, true>), grid, dim3(THREADS), histogram_bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}
inline int launch_weighted(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.cross_groups) {
        case 1: return launch_weighted_form<1>(args, grid, histogram_bytes, stream);
        case 2: return launch_weighted_form<2>(args, grid, histogram_bytes, stream);
        case 3: return launch_weighted_form<3>(args, grid, histogram_bytes, stream);
        case 4: return launch_weighted_form<4>(args, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

constexpr int SHORT_LANES = 4;             // lanes of a read of a set of at most CROSS_MAX_UNIQUE words a segment
constexpr int LONG_LANES = 8;              // ... of a single segment set of up to EXACT_SINGLE_UNIQUE words

}  // namespace lanes

// what pamld_exact_supports takes, less the three segment sets and the weighted combinatorial sets; the byte parallel
// quality compare wants a threshold of at most 128 and the histogram must leave the kernel's static LDS its room
bool pamld_exact_lanes_supports(const DevDecoder& host_decoder, size_t histogram_bytes) {
    static const bool off = std::getenv("PHQ_NO_EXACT_LANES") != nullptr;
    static const bool no_weighted = std::getenv("PHQ_NO_EXACT_LANES_WEIGHTED") != nullptr;
    if(off || !pamld_exact_supports(host_decoder)) { return false; }
    if(host_decoder.cross_segments > 2) { return false; }
    if(host_decoder.cross_segments == 2 && !host_decoder.uniform_prior && (host_decoder.cross_prior_double == nullptr || no_weighted)) { return false; }
    return host_decoder.high_quality_threshold <= 128 && histogram_bytes <= 24u * 1024u;
}

// The tier's first pass settles tied noise reads (lanes::TIE_PACE) for a single segment codec of 5 to 16 nt with uniform priors, all
// ACGT, numbered in barcode order, whose reads' ordinals leave bit 31 of a worklist entry free and whose blocks' numbers fit their
// thirteen bits: the kernels in front of it then flag such reads (StageArgs::flag_ties)
bool pamld_exact_lanes_ties(const StageArgs& args, const DevDecoder& host_decoder, size_t histogram_bytes) {
    static const bool off = std::getenv("PHQ_NO_TIES") != nullptr;
    return !off && pamld_exact_lanes_supports(host_decoder, histogram_bytes) && host_decoder.algorithm == 1 && host_decoder.all_strict && host_decoder.uniform_prior &&
           host_decoder.cross_segments == 1 && host_decoder.cross_identity != 0 && host_decoder.cross_unique[0] == host_decoder.n_barcode &&
           host_decoder.nucleotide_cardinality >= 5 && host_decoder.nucleotide_cardinality <= 16 && host_decoder.n_barcode <= 8 * 8192 &&
           args.n_read < (int64_t(1) << 31) && args.result != nullptr;
}

// `grid` is the producer's grid = the number of worklist segments; a wave takes `group` of them as one list
int launch_pamld_exact_lanes(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    StageArgs grouped = args;
    grouped.worklist_segments = static_cast<int32_t>(grid.x);
    // about 1 % of the reads reach this tier: segments are grouped until a wave can expect a few iterations' worth of them
    const bool many = host_decoder.cross_segments == 1 && host_decoder.cross_unique[0] > CROSS_MAX_UNIQUE;
    const bool weighted = host_decoder.cross_segments == 2 && !host_decoder.uniform_prior;
    const int reads = 64 / (many ? lanes::LONG_LANES : weighted ? PHQ_WEIGHTED_LANES : lanes::SHORT_LANES);
    // (round 5, swept with PHQ_EXACT_GROUP on C2: at 1 Mi reads a call — 3.4 reads expected a segment — groups of 1 / 2 / 4 / 8 / 16 took
    // 31.9 / 23.0 / 18.7 / 24.1 / 32.8 us, at 4 Mi 36.0 / 30.3 / 34.6 / 48.5 / 76.7, at 256 Ki 26.8 / 21.4 / 17.7 / 17.7 / 24.3: a wave wants
    // an iteration or two, not a list — every iteration is a chain of dependent loads — and sixteen segments a wave, the rule until
    // now, was the slowest choice but none)
    const double expected_per_segment = 0.01 * static_cast<double>(args.n_read) / static_cast<double>(std::max(1u, grid.x));
    const int group = expected_per_segment >= 1.7 * reads ? 1 : expected_per_segment >= 0.55 * reads ? 2 : 4;
    grouped.worklist_group = std::min(lanes::GROUP_MAX, std::max(1, group));
    static const int forced = std::getenv("PHQ_EXACT_GROUP") ? std::atoi(std::getenv("PHQ_EXACT_GROUP")) : 0;       // sweeps
    if(forced > 0) { grouped.worklist_group = std::min(lanes::GROUP_MAX, forced); }
    const unsigned per_workgroup = static_cast<unsigned>(lanes::WAVES * grouped.worklist_group);
    const dim3 fewer((grid.x + per_workgroup - 1) / per_workgroup);
    if(weighted) {
        // the listed form with eight lanes a read, then the whole form — a wave per read — on what its lists could not hold
        if(const int status = lanes::launch_weighted(grouped, host_decoder, fewer, histogram_bytes, stream)) { return status; }
        trace_worklist("the exact tier's lists", args, grid.x, stream);
        return launch_pamld_cross_exact_dense(args, host_decoder, grid, histogram_bytes, stream);
    }
    if(host_decoder.cross_segments == 1) {
        if(many && host_decoder.cross_unique[0] > EXACT_SINGLE_MEDIUM) {
            return host_decoder.uniform_prior ? lanes::launch_groups<1, false, EXACT_SINGLE_UNIQUE, lanes::LONG_LANES>(grouped, host_decoder, fewer, histogram_bytes, stream)
                                              : lanes::launch_groups<1, true, EXACT_SINGLE_UNIQUE, lanes::LONG_LANES>(grouped, host_decoder, fewer, histogram_bytes, stream);
        }
        if(many) {
            return host_decoder.uniform_prior ? lanes::launch_groups<1, false, EXACT_SINGLE_MEDIUM, lanes::LONG_LANES>(grouped, host_decoder, fewer, histogram_bytes, stream)
                                              : lanes::launch_groups<1, true, EXACT_SINGLE_MEDIUM, lanes::LONG_LANES>(grouped, host_decoder, fewer, histogram_bytes, stream);
        }
        return host_decoder.uniform_prior ? lanes::launch_groups<1, false, CROSS_MAX_UNIQUE, lanes::SHORT_LANES>(grouped, host_decoder, fewer, histogram_bytes, stream)
                                          : lanes::launch_groups<1, true, CROSS_MAX_UNIQUE, lanes::SHORT_LANES>(grouped, host_decoder, fewer, histogram_bytes, stream);
    }
    return lanes::launch_groups<2, false, CROSS_MAX_UNIQUE, lanes::SHORT_LANES>(grouped, host_decoder, fewer, histogram_bytes, stream);
}

}  // namespace phq
