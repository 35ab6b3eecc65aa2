// K0 (neighbour lists): PamlDecoder<T>::classify for the reads a short list of barcodes settles.
//
//   reference pamld.cpp:37-123   posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164  P(r|b) = pow(base, sum of Phred costs), distance, high quality distance
//
// The product form kernel (pamld_lut.hpp) scores every barcode for every read.  Almost all of that work adds
// nothing: a read sits on or next to ONE barcode c, and a barcode b' that differs from c in d positions mismatches a
// read at distance k from c in at least d - k positions, each of which costs it a factor rho = P(mismatch) / P(match)
// of that position's quality.  So per read this kernel
//   * names c with one lookup: the read's 2-bit codes are a key into a host built table (4^L entries of 16 bits, in
//     global memory; the few thousand keys real reads carry stay in L1/L2) that holds, for every word at most ONE
//     mismatch away from a barcode, that barcode — a read further away is not finished here anyway;
//   * scores c and the 15 (14 for 9 and 10 nt) barcodes nearest to c with the per read product tables of
//     pamld_lut.hpp (one 16 entry table of single precision products per group of 4 positions, in LDS); the list of
//     c — the bit sliced words of those barcodes, four (three) barcodes per pair of plane words — is read per lane
//     from LDS (three or four ds_read_b128);
//   * BOUNDS every barcode it did not score: with r = the distance from c of the nearest one not listed (host: 4 or
//     5 for sets like Illumina's), such a barcode mismatches the read in n >= r - k positions outside the k positions
//     where the read differs from c, so its probability is at most the largest product the tables hold over n (or
//     more) mismatches.  That maximum is formed from the tables' own entries: per group of 4 positions the largest
//     entry by number of mismatches (G[0..4], made monotone so that qualities below 4, where a mismatch is MORE likely
//     than a match, stay covered), then H[n] = max over a + b = n of G1[a] * G2[b].  The barcodes at distance exactly
//     r (cnt_r of them, host) are charged H[r - k], the others H[r + 1 - k]; the k = 1 case leaves the read's own
//     mismatch position out through  top_n(without j) <= min(top_n, top_{n+1} / rho_j).
// A read is FINISHED here when all of this is certain to 1e-8 in the posterior: it passes the noise filter clearly,
// everything else sums to at most 5e-4 of the winner (so single precision, relative error below ~1e-6, moves the
// posterior by less than 5e-10: the posterior is formed as 1 - x / (1 + x), x = (others + noise) / winner, and only
// that last subtraction is in double precision), the bound on what was not scored moves it by at most 1e-8, and
// the confidence is further than 2e-8 from its threshold.  About 94 % of the reads of the 96 x 8 nt benchmark.
// Every other read — more than one mismatch against c, a base outside ACGT, several low quality positions, noise
// reads — is appended to this workgroup's worklist segment; the product form kernel, now a CONSUMER of that list,
// scores all barcodes for those, then the exact tier and the literal kernel as before.
//
// Instruction mix: on gfx950 the logical / add / shift right / fp32 mul, add, fma instructions issue at twice the
// rate of everything else (v_perm, v_cndmask, v_max, v_lshlrev, packed and fp64 arithmetic; measured,
// profiles/tools/micro).  A table address is therefore formed with one shift right and one v_bitop3_b32
// ((fields >> s) & 0xf00 | lane constant): the tables are laid out [wave][group][entry][lane] with the entry index at
// address bits 8..11.
#pragma once

#include "kernel_common.hpp"

namespace phq {
namespace near {

constexpr int FACTOR_ROWS = 129;
constexpr float LISTED_MAX = 5e-4f;            // scored barcodes other than c sum to at most this share of c's probability
// (others + noise) / winner = x: single precision is enough below this.  x carries the roundings of ~2 L + 3 single
// precision operations (the factors themselves, their products, the reciprocal): at most ~1.4e-6 relative for 10 nt if
// all of them align, which moves the posterior by x * 1.4e-6 < 7e-10
constexpr float X_MAX = 5e-4f;
// the bound on what was not scored, as a share of the winner = the largest error leaving it out of the posterior can cause:
// two orders inside the 1e-6 the posterior has to hold.  (1e-9 until the
// end of round 2: the model — profiles/tools/sim/near_certificate.py — and the GPU agree that ten times the room finishes
// 94.8 % of the 96 x 8 nt reads here instead of 91.4 %, and every deferred read costs two scattered line writes.)
// Round 3: 9e-9, not 1e-8 — the bound can be tight (a crowded 76 x 6 nt set, a Q7 call beside five Q37: what was left out weighed
// 1.0001 of its bound's 1e-8) and x's own 7e-10 comes on top: seed 12982 of the random jobs, 1.0008e-8 from the oracle
constexpr float REST_MAX = 9e-9f;
constexpr float SLOP = 1.0f + 1e-5f;           // single precision rounding of the bound itself

// Shapes by barcode length.  5 .. 8 nt: two groups of 4 positions, 8 flag bits per barcode, four barcodes per plane
// word, 16 listed, 512 lane workgroups (two per CU).  9 and 10 nt: three groups (the last one of 1 or 2 positions),
// 10 flag bits per barcode, three per word, 15 listed, 256 lane workgroups.
// PRIORS: the barcodes' priors differ (estimated priors, phq_adjust_prior): a record also holds the prior of each listed
// barcode over c's, 64 bytes more per barcode.  Beside two 512 lane workgroups' tables (64 KiB each) that no longer fits
// a CU, so the two group shapes then run ONE workgroup of 1024 lanes per CU (measured with uniform priors: 4 % slower
// than two of 512), its tables laid out [wave][group] so that a group is still reached through the offset field.
// BIG: a codec whose records alone push two 512 lane workgroups out of a CU (more than ~120 barcodes of 5 to 8 nt) takes the
// same 1024 lane shape.
// SLIM (round 6): prior ratios in the 512 lane layout, ONE such workgroup a CU — 64 KiB of tables instead of 128 leave the records
// of up to ~850 barcodes their room (the 1 024 lane shape holds 190 with ratios: a 384-plex after its prior estimation pass had
// the product form over all barcodes as first kernel).
template <int L, bool PRIORS = false, bool BIG = false, bool SLIM = false> struct Shape {
    static_assert(L >= 5 && L <= 10, "two or three groups of 4 positions");
    static constexpr int G = (L + 3) / 4;
    static constexpr bool WIDE = (PRIORS || BIG) && G == 2 && !SLIM;
    static_assert(!SLIM || (PRIORS && !BIG && G == 2), "the slim shape is the two group one with prior ratios");
    static constexpr int FIELD = near_field_bits(L);
    static constexpr int PER_WORD = 32 / FIELD;
    static constexpr int LIST = near_list(L);
    static constexpr int LIST_WORDS = LIST / PER_WORD;
    static constexpr int RECORD_WORDS = near_record_words(L, PRIORS);
    static constexpr int RECORD_QUADS = RECORD_WORDS / 4;
    static constexpr int WEIGHT_AT = near_record_words(L);                     // PRIORS: the first of LIST prior ratios (slot 0: the noise term)
    static constexpr int THREADS = G == 2 ? (WIDE ? 1024 : 512) : 256;
    static constexpr int WAVES = THREADS / 64;
    static constexpr uint32_t WAVE_TABLE = 16u * 64u * 4u;                     // one group's table of one wave
    static constexpr uint32_t GROUP_BYTES = WIDE ? WAVE_TABLE : WAVES * WAVE_TABLE;      // from a group's table to the next one's
    static constexpr uint32_t WAVE_BYTES = WIDE ? G * WAVE_TABLE : WAVE_TABLE;           // from a wave's tables to the next one's
    static constexpr uint32_t TABLE_AT = 0;                                    // [group][wave][entry][lane] floats; WIDE: [wave][group][entry][lane]
    static constexpr uint32_t TABLE_BYTES = G * WAVES * WAVE_TABLE;
    static constexpr uint32_t FACTOR_AT = TABLE_AT + TABLE_BYTES;              // FACTOR_ROWS x {match, mismatch} floats
    static constexpr uint32_t DEFERRED_AT = FACTOR_AT + 1040;
    static constexpr uint32_t RECORD_AT = DEFERRED_AT + 16;                    // n_barcode x RECORD_WORDS words
    __host__ __device__ static constexpr uint32_t histogram_at(int n_barcode) {
        return RECORD_AT + static_cast<uint32_t>(n_barcode) * RECORD_WORDS * 4u;
    }
    __host__ __device__ static constexpr int positions(int g) { return L - 4 * g >= 4 ? 4 : L - 4 * g; }
    static_assert(LIST % PER_WORD == 0 && RECORD_WORDS % 4 == 0 && RECORD_WORDS >= 2 * LIST_WORDS + 4, "record layout");
    static_assert((G - 1) * GROUP_BYTES <= 65535u, "a group's table is reached through the instruction's offset field");
};

typedef uint32_t Quad __attribute__((ext_vector_type(4)));
typedef float Pair __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const float* LdsFloats;
typedef __attribute__((address_space(3))) const Quad* LdsQuads;
typedef __attribute__((address_space(3))) const Pair* LdsPairs;

// (a & mask) | c as v_bitop3_b32 (full rate; hipcc's own choice, v_and_or_b32, issues at half rate): truth table 0xEA
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, mask, c, 0xEA);
}

// STATE: where a read's carried state comes from — 0 the first stage of a job without input flags (nothing to load),
// 1 the first stage with flag bytes, 2 a later stage (the status word).  A template parameter so that the loads of
// the prefetch sit in straight line code: behind a branch the compiler can no longer count them and waits for all.
template <int L, int STATE, bool PRIORS, bool BIG, bool SLIM = false>
__global__ __launch_bounds__((Shape<L, PRIORS, BIG, SLIM>::THREADS)) void pamld_near_kernel(StageArgs args) {
    typedef Shape<L, PRIORS, BIG, SLIM> Layout;
    constexpr int G = Layout::G;
    constexpr int THREADS = Layout::THREADS;
    constexpr int FIELD = Layout::FIELD;
    constexpr int PER_WORD = Layout::PER_WORD;
    constexpr int LIST_WORDS = Layout::LIST_WORDS;
    constexpr int RECORD_WORDS = Layout::RECORD_WORDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    float2* const factor_single = reinterpret_cast<float2*>(lds_raw + Layout::FACTOR_AT);
    uint32_t& deferred_total = *reinterpret_cast<uint32_t*>(lds_raw + Layout::DEFERRED_AT);
    uint32_t* const records = reinterpret_cast<uint32_t*>(lds_raw + Layout::RECORD_AT);
    DecoderRef d = *constant_cast(args.decoder);
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw + Layout::histogram_at(d.n_barcode));

    for(int i = threadIdx.x; i < FACTOR_ROWS; i += THREADS) {
        const double m = i < 128 ? d.match_probability[i] : d.uniform_probability;
        const double e = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        factor_single[i] = make_float2(static_cast<float>(m), static_cast<float>(e));
    }
    if(threadIdx.x == 0) { deferred_total = 0; }
    for(int w = threadIdx.x; w < d.n_barcode * RECORD_WORDS; w += THREADS) { records[w] = d.near_record[w]; }
    histogram_clear(d, histogram);
    __syncthreads();

    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const uint16_t* const nearest = d.near_table;
    const int64_t n = args.n_read;
    const int64_t tiles = (n + THREADS - 1) / THREADS;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t lane_const = ((threadIdx.x & 63u) << 2) + wave * Layout::WAVE_BYTES;     // everything of a table address but group and entry (bits 8..11 clear)
    const int high_quality_threshold = min(max(d.high_quality_threshold, 0), 128);
    const uint32_t high_quality_addend = static_cast<uint32_t>(128 - high_quality_threshold) * 0x01010101u;
    const float uniform_noise_over_prior = static_cast<float>(d.adjusted_noise_probability / d.concentration[0]);
    const float noise_floor = static_cast<float>(d.random_barcode_probability) * SLOP;
    const double confidence_threshold = d.confidence_threshold;
    const uint32_t n_barcode = static_cast<uint32_t>(d.n_barcode);
    // the bytes of the positions a word holds (word w: positions 4 w .. 4 w + 3)
    auto valid = [](int w) -> uint32_t { return L - 4 * w >= 4 ? 0xffffffffu : ((1u << (8 * (L - 4 * w))) - 1u); };

    // A tile's inputs are fetched one tile ahead — behind this tile's table lookup, so that waiting for the lookup does
    // not wait for them (vector memory operations complete in issue order) — through a scalar tile base and a 32-bit
    // lane offset.  Lanes past the end of the batch recompute its last read and are kept from every counter.
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
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): the loop is entered with nothing in flight, so that its first wait is the one of its back edge
    uint32_t pending_slot = 0xffffffffu;       // this lane's worklist entry of the previous tile, not stored yet
    uint32_t pending_read = 0;
    uint32_t pending_input[CARRY_PLANES] = {0, 0, 0, 0, 0, 0, 0};
    const uint32_t carry_stride = args.carry ? static_cast<uint32_t>(args.carry_stride) : 0u;
    uint32_t* const carry = args.carry + static_cast<int64_t>(blockIdx.x) * CARRY_PLANES * carry_stride;
    const int hq_threshold = d.high_quality_distance_threshold;
    const bool multiplexing = d.multiplexing != 0;
    const bool lds_bins = d.use_lds_histogram != 0;          // else the launch defers the classified reads' counters (StageArgs::defer_counters)
    for(int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const Inputs in = next;
        const int64_t first = tile * THREADS;
        const bool active = first + threadIdx.x < n;
        const uint32_t at = active ? threadIdx.x : static_cast<uint32_t>(n - 1 - first);
        const int64_t i = first + at;
        ReadState r;
        if(STATE != 2) { r.qcfail = (in.state & 1u) != 0; r.channel = 0; }
        else { r.qcfail = (in.state >> 31) != 0; r.channel = static_cast<int32_t>(in.state & 0x7fffffffu); }
        // BAM codes, four positions per word.  A, C, G or T = exactly one bit of the nibble (iupac.h:51-56); a read
        // with anything else goes to the next tier.
        uint32_t x[G];
        uint32_t q7[G];
        uint32_t not_strict = 0;
#pragma unroll
        for(int w = 0; w < G; ++w) {
            x[w] = in.code[w] & (0x0f0f0f0fu & valid(w));
            q7[w] = in.quality[w] & (0x7f7f7f7fu & valid(w));                  // the table has 128 rows
            const uint32_t z = x[w] & ((x[w] | 0x80808080u) - 0x01010101u);
            const uint32_t strict = (x[w] + 0x7f7f7f7fu) & ~(z + 0x7f7f7f7fu) & 0x80808080u;
            not_strict |= strict ^ (0x80808080u & valid(w));
        }
        const bool all_strict = not_strict == 0;
        // 2-bit codes A, C, G, T -> 0..3: the low code bit is bit 1 | bit 3 of the nibble, the high one bit 2 | bit 3.
        // Bits 0 / 4 of the bytes of two flag planes gathered into one byte; bit j of low / high = position j.
        auto gather = [](uint32_t planes) {            // bits 8 j and 8 j + 4 -> bits j and 4 + j
            const uint32_t t = planes | (planes >> 7);
            return (t | (t >> 14)) & 0xffu;
        };
        uint32_t low = gather((((x[0] >> 1) | (x[0] >> 3)) & 0x01010101u) | (((x[1] << 3) | (x[1] << 1)) & 0x10101010u));
        uint32_t high = gather((((x[0] >> 2) | (x[0] >> 3)) & 0x01010101u) | (((x[1] << 2) | (x[1] << 1)) & 0x10101010u));
        if(G == 3) {
            const uint32_t both = gather((((x[G - 1] >> 1) | (x[G - 1] >> 3)) & 0x01010101u) | (((x[G - 1] << 2) | (x[G - 1] << 1)) & 0x10101010u));
            low |= (both & 0xfu) << 8;
            high |= (both >> 4) << 8;
        }
        const uint32_t key = low | (high << L);
        if(pending_slot != 0xffffffffu) {                                         // the previous tile's deferred reads
            segment[pending_slot] = pending_read;
            if(pending_slot < carry_stride) {
#pragma unroll
                for(int plane = 0; plane < CARRY_PLANES; ++plane) {
                    if(plane < G || (plane >= 3 && plane < 3 + G) || plane == 6) { carry[plane * carry_stride + pending_slot] = pending_input[plane]; }
                }
            }
        }
        const uint32_t looked_up = nearest[key];                                  // the barcode this word is (next to), if any
        fetch(min(tile + gridDim.x, tiles - 1), next);                           // past the last tile: that tile again, unused
        const bool known = looked_up < n_barcode;
        const uint32_t c = known ? looked_up : 0u;
        constexpr uint32_t REPLICATE = PER_WORD == 4 ? 0x01010101u : 0x00100401u;  // the read against PER_WORD barcodes at a time
        const uint32_t observed_low = low * REPLICATE;
        const uint32_t observed_high = high * REPLICATE;
        // quality >= high quality threshold (thresholds of 0..128), bit 7 of each byte -> bit j = position j
        uint32_t high_quality = gather(((((q7[0] + high_quality_addend) | in.quality[0]) & 0x80808080u) >> 7) |
                                       ((((q7[1] + high_quality_addend) | in.quality[1]) & (0x80808080u & valid(1))) >> 3));
        if(G == 3) { high_quality |= (gather((((q7[G - 1] + high_quality_addend) | in.quality[G - 1]) & (0x80808080u & valid(G - 1))) >> 7) & 0xfu) << 8; }

        // per position {P(match), P(mismatch)} of its quality, single precision
        float fm[4 * G];
        float fe[4 * G];
        {
            const uint32_t factor_base = Layout::FACTOR_AT;
#pragma unroll
            for(int j = 0; j < L; ++j) {
                const uint32_t q = q7[j >> 2];
                const int byte = j & 3;
                const uint32_t shifted = byte == 0 ? (q << 3) : (q >> (8 * byte - 3));
                const uint32_t address = and_or(shifted, 0x3f8u, factor_base);
                const Pair pair = *(LdsPairs)static_cast<uintptr_t>(address);
                fm[j] = pair.x;
                fe[j] = pair.y;
            }
#pragma unroll
            for(int j = L; j < 4 * G; ++j) { fm[j] = 1.0f; fe[j] = 1.0f; }
        }
        // tables: entry m of group g = product over the group's positions of (bit ? mismatch : match); and per group the
        // largest entry by number of mismatches, made monotone: ge[g][a] = the largest entry with a OR MORE mismatches
        // (none past the group's positions)
        float ge[G][5];
        float everything_matches = 1.0f;
#pragma unroll
        for(int g = 0; g < G; ++g) {
            constexpr int base = 0;
            const int p0 = 4 * g + base;
            float lo[4];
            float hi[4];
            lo[0] = fm[p0] * fm[p0 + 1]; lo[1] = fe[p0] * fm[p0 + 1]; lo[2] = fm[p0] * fe[p0 + 1]; lo[3] = fe[p0] * fe[p0 + 1];
            hi[0] = fm[p0 + 2] * fm[p0 + 3]; hi[1] = fe[p0 + 2] * fm[p0 + 3]; hi[2] = fm[p0 + 2] * fe[p0 + 3]; hi[3] = fe[p0 + 2] * fe[p0 + 3];
            const int positions = Layout::positions(g);
            const int entries = 1 << positions;
            float t[16];
            float by_count[5] = {0, 0, 0, 0, 0};        // the largest entry with exactly a mismatches
#pragma unroll
            for(int m = 0; m < 16; ++m) {
                if(m < entries) {
                    t[m] = positions <= 2 ? lo[m & 3] : lo[m & 3] * hi[m >> 2];
                    *reinterpret_cast<float*>(lds_raw + Layout::TABLE_AT + static_cast<uint32_t>(g) * Layout::GROUP_BYTES + (lane_const | (static_cast<uint32_t>(m) << 8))) = t[m];
                    const int a = __builtin_popcount(static_cast<unsigned>(m));
                    by_count[a] = (m == (1 << a) - 1) ? t[m] : fmaxf(by_count[a], t[m]);      // the first entry of a count is 2^a - 1
                }
            }
            everything_matches = g == 0 ? t[0] : everything_matches * t[0];
#pragma unroll
            for(int a = 4; a >= 0; --a) {
                ge[g][a] = a > positions ? 0.0f : a == positions ? by_count[a] : fmaxf(by_count[a], ge[g][a + 1 > 4 ? 4 : a + 1]);
            }
        }
        // H[n]: the largest product over n or more mismatching positions, n = 3 .. 6.  Two groups at a time.
        float h[7] = {0, 0, 0, 0, 0, 0, 0};
        {
            float front[9];                        // groups 0 and 1 together, by number of mismatches
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

        // the list of c: its barcodes as LIST_WORDS x {low plane word, high plane word}, then {radius, cnt_r, cnt_rest, index}
        const uint32_t record_at = Layout::RECORD_AT + c * (RECORD_WORDS * 4u);
        Quad record[Layout::RECORD_QUADS];
#pragma unroll
        for(int quad = 0; quad < Layout::RECORD_QUADS; ++quad) { record[quad] = *(LdsQuads)static_cast<uintptr_t>(record_at + 16u * quad); }
        auto record_word = [&](int w) { return record[w >> 2][w & 3]; };
        float winner = 0;                        // P(r|c), single precision
        float others = 0;                        // the other listed barcodes
        uint32_t mismatches = 0;                 // of c: bit j = position j
#pragma unroll
        for(int w = 0; w < LIST_WORDS; ++w) {
            // FIELD bits of `fields` per barcode: its mismatch flags, four per group
            const uint32_t fields = (observed_low ^ record_word(2 * w)) | (observed_high ^ record_word(2 * w + 1));
            if(w == 0) { mismatches = fields & ((1u << L) - 1u); }
            float p[PER_WORD];
#pragma unroll
            for(int b = 0; b < PER_WORD; ++b) {
#pragma unroll
                for(int g = 0; g < G; ++g) {
                    const int at_bit = FIELD * b + 4 * g;     // where the group's flags start; the entry index belongs at bits 8..
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
                else if(PRIORS) {
                    const float weight = __uint_as_float(record_word(Layout::WEIGHT_AT + PER_WORD * w + b));     // prior of this barcode over c's
                    sum = (w == 0 && b == 1) ? p[b] * weight : __builtin_fmaf(p[b], weight, sum);
                }
                else { sum = (w == 0 && b == 1) ? p[b] : sum + p[b]; }
            }
            others = w == 0 ? sum : others + sum;
        }

        // what was not scored: cnt_r barcodes at distance r from c, cnt_rest further away
        const uint32_t radius = record_word(2 * LIST_WORDS);
        const float cnt_r = __uint_as_float(record_word(2 * LIST_WORDS + 1));
        const float cnt_rest = __uint_as_float(record_word(2 * LIST_WORDS + 2));
        const uint32_t index = record_word(2 * LIST_WORDS + 3);                  // barcode_index[c]
        const float noise_over_prior = PRIORS ? __uint_as_float(record_word(Layout::WEIGHT_AT)) : uniform_noise_over_prior;
        const int k = __popc(mismatches);
        const bool near5 = radius >= 5u;                                    // the host clamps the radius to 5
        const bool near3 = radius == 3u;                                    // a crowded set (384 words of 8 nt): only a read ON c is finished here
        const float h_r = near5 ? h[5] : near3 ? h[3] : h[4];               // H[r]
        const float h_up = near5 ? h[6] : near3 ? h[4] : h[5];              // H[r + 1]
        const float h_down = near5 ? h[4] : h[3];                           // H[r - 1] (radius 3: not used)
        const float inverse_winner = __builtin_amdgcn_rcpf(winner);
        // k = 1: rho of the read's own mismatch position is winner / everything_matches; without that position
        // top_n <= min(top_n, top_{n+1} / rho), and the position itself costs at most max(rho, 1)
        const float inverse_rho = everything_matches * inverse_winner;
        const float boost = fmaxf(winner * __builtin_amdgcn_rcpf(everything_matches), 1.0f);
        const float at_r = k == 0 ? h_r : boost * fminf(h_down, h_r * inverse_rho);
        const float beyond = k == 0 ? h_up : boost * fminf(h_r, h_up * inverse_rho);
        const float rest = (cnt_r * at_r + cnt_rest * beyond) * SLOP;
        const float x_listed = others * inverse_winner;
        const float x_rest = rest * inverse_winner;
        const float x_low = x_listed + noise_over_prior * inverse_winner;
        // the posterior is taken a QUARTER of the way into what was only bounded: those barcodes really weigh between nothing and
        // x_rest, typically a tenth of it (profiles/tools/sim/near_certificate.py), so the error is at most 0.75 x_rest —
        // 6.8e-9 at REST_MAX, with x's own 7e-10 on top 7.5e-9: a proven distance from the 1e-8 the tests assert (until round 5
        // it was taken without them: at most x_rest, 9.7e-9 with the rounding, and 8.8e-9 was seen) — and typically 0.15 of it
        const float x_mid = __builtin_fmaf(0.25f, x_rest, x_low);
        const float y = x_mid * __builtin_amdgcn_rcpf(1.0f + x_mid);            // 1 - posterior
        const double confidence = 1.0 - static_cast<double>(y);            // pamld.cpp:92
        const double confidence_margin = confidence - confidence_threshold;
        const bool finished = all_strict && known && k <= 1 && (radius >= 4u || (near3 && k == 0)) && winner > noise_floor &&
                              x_listed <= LISTED_MAX && x_low + x_rest <= X_MAX && x_rest <= REST_MAX &&
                              fabs(confidence_margin) > 2e-8;
        const bool defer = !finished;

        // Outcome.  Every lane stores a record and a status word in straight line code — the real ones when the read is
        // finished; a deferred read's record is whatever was computed (the next tier writes it again), its status word
        // the state it came with; a lane past the end of the batch stores its (last) read's again — so that the compiler
        // can count these two stores at the wait for the prefetched inputs.
        const uint32_t hq_distance = static_cast<uint32_t>(__popc(mismatches & high_quality));
        const bool confident = confidence_margin > 0;                               // pamld.cpp:99
        const bool qcfail = r.qcfail || !confident || (hq_threshold > 0 && static_cast<int>(hq_distance) >= hq_threshold);
        const bool qcfail_out = finished ? qcfail : r.qcfail;
        {
            typedef uint32_t vector4 __attribute__((ext_vector_type(4)));
            vector4 out;
            out.x = index;
            out.y = static_cast<uint32_t>(k) | (hq_distance << 8) | ((confident ? STATUS_PASS : STATUS_LOW_CONFIDENCE) << 16) | (qcfail_out ? 1u << 24 : 0u);
            out.z = static_cast<uint32_t>(__double2loint(confidence));
            out.w = static_cast<uint32_t>(__double2hiint(confidence));
            __builtin_nontemporal_store(out, reinterpret_cast<vector4*>(reinterpret_cast<char*>(args.result) + first * 16 + at * 16u));
            const uint32_t channel = (finished && multiplexing) ? index : static_cast<uint32_t>(r.channel);
            __builtin_nontemporal_store((channel & 0x7fffffffu) | (qcfail_out ? 0x80000000u : 0u), reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(args.word) + first * 4 + at * 4u));
        }
        r.qcfail = qcfail_out;
        if(finished && active) {
            // Decoder<T>::classify then Classifier<T>::classify (decoder.h:68-76, classifier.h:78-86), workgroup histogram
            const uint32_t word = STAT_JOB_WORDS + (c + 1u) * STAT_WORDS_PER_BIN + (qcfail ? 0u : static_cast<uint32_t>(STAT_FIELDS));
            uint32_t* const low_words = reinterpret_cast<uint32_t*>(histogram);      // a workgroup's counts and distances stay below 2^32
            if(confident) {
                if(!args.defer_counters) {
                    atomicAdd(&low_words[2 * (word + FIELD_COUNT)], 1u);
                    if(k) { atomicAdd(&low_words[2 * (word + FIELD_DISTANCE)], static_cast<uint32_t>(k)); }
                    // round(confidence * 2^52) as the mantissa of 1 + confidence (a confidence of exactly 1 carries into
                    // the exponent, which the subtraction turns into bit 52), split in two 26 bit limbs
                    const double shifted = confidence + 1.0;
                    const uint32_t fixed_low = static_cast<uint32_t>(__double2loint(shifted));
                    const uint32_t fixed_high = static_cast<uint32_t>(__double2hiint(shifted)) - 0x3ff00000u;
                    atomicAdd(&histogram[word + FIELD_CONFIDENCE_LOW], static_cast<unsigned long long>(fixed_low & 0x3ffffffu));
                    atomicAdd(&histogram[word + FIELD_CONFIDENCE_HIGH], static_cast<unsigned long long>((fixed_high << 6) | (fixed_low >> 26)));
                }
            } else {
                // qc-fail half (pamld.cpp:104-106); a codec too large for a workgroup histogram counts it in global memory (rare)
                if(lds_bins) { atomicAdd(&low_words[2 * (STAT_JOB_WORDS + (c + 1u) * STAT_WORDS_PER_BIN + FIELD_LOW_CONFIDENCE)], 1u); }
                else { atomicAdd(&args.stats[d.stats_offset + (c + 1u) * STAT_WORDS_PER_BIN + FIELD_LOW_CONFIDENCE], 1ull); }
                if(!args.defer_counters) {
                    atomicAdd(&low_words[2 * (word + FIELD_COUNT)], 1u);
                    if(k) { atomicAdd(&low_words[2 * (word + FIELD_DISTANCE)], static_cast<uint32_t>(k)); }
                }
            }
        }
        // deferred reads: one LDS add per wave reserves a run of slots in this workgroup's worklist segment; the store
        // itself waits for the next tile (in front of its prefetch: see above)
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
#pragma unroll
                for(int w = 0; w < G; ++w) { pending_input[w] = in.code[w]; pending_input[3 + w] = in.quality[w]; }
                pending_input[6] = in.state;
            }
        }
        count_outgoing(args, histogram, active && !defer, r);
    }
    if(pending_slot != 0xffffffffu) {
        segment[pending_slot] = pending_read;
        if(pending_slot < carry_stride) {
#pragma unroll
            for(int plane = 0; plane < CARRY_PLANES; ++plane) {
                if(plane < G || (plane >= 3 && plane < 3 + G) || plane == 6) { carry[plane * carry_stride + pending_slot] = pending_input[plane]; }
            }
        }
    }
    __syncthreads();
    if(threadIdx.x == 0) { args.worklist_count[blockIdx.x] = deferred_total; }
    histogram_flush(args, d, histogram);
}

template <int L, int STATE, bool PRIORS, bool BIG, bool SLIM = false>
int launch_state(const StageArgs& args, int n_barcode, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    const size_t lds_bytes = Shape<L, PRIORS, BIG, SLIM>::histogram_at(n_barcode) + histogram_bytes;
    static std::atomic<uint64_t> prepared{0};
    if(const int status = allow_large_lds(&pamld_near_kernel<L, STATE, PRIORS, BIG, SLIM>, prepared)) { return status; }
    hipLaunchKernelGGL((pamld_near_kernel<L, STATE, PRIORS, BIG, SLIM>), grid, dim3(Shape<L, PRIORS, BIG, SLIM>::THREADS), lds_bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}

template <int L, bool PRIORS, bool BIG, bool SLIM = false>
int launch_one(const StageArgs& args, int n_barcode, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(!args.first_stage) { return launch_state<L, 2, PRIORS, BIG, SLIM>(args, n_barcode, grid, histogram_bytes, stream); }
    return args.input_flag ? launch_state<L, 1, PRIORS, BIG, SLIM>(args, n_barcode, grid, histogram_bytes, stream)
                           : launch_state<L, 0, PRIORS, BIG, SLIM>(args, n_barcode, grid, histogram_bytes, stream);
}

// the 1024 lane shape for the two group lengths when two 512 lane workgroups with this codec's records (and histogram)
// no longer fit a CU
template <int L>
inline bool wants_big(int n_barcode, size_t histogram_bytes) {
    return Shape<L, false, false>::G == 2 && 2 * (Shape<L, false, false>::histogram_at(n_barcode) + histogram_bytes) > 160u * 1024u;
}

}  // namespace near
}  // namespace phq
