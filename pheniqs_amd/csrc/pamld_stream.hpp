// K0s (streaming): PamlDecoder<T>::classify, the neighbour list tier and the product form scan in ONE persistent kernel.
//
//   reference pamld.cpp:37-123   posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164  P(r|b) = pow(base, sum of Phred costs), distance, high quality distance
//
// Rounds 1 and 2 ran a stage as a chain of launches: the neighbour list kernel (pamld_near.hpp) finished ~93 % of the
// reads and wrote the others to a worklist in HBM, the product form kernel (pamld_lut.hpp) came back for them — their
// inputs through carry planes, their 16 + 4 byte results as scattered partial line stores, short lists, a memory round
// trip per tile — and cost a third of the step for 7 % of the reads.  Here every WAVE is its own pipeline:
//   * pass A, every 64 read tile: name the barcode c the read sits on or next to (one table lookup), score c and the
//     15 barcodes nearest to c exactly from per read tables in LDS, bound everything else, finish the read under the
//     certificate of pamld_near.hpp (posterior to 1e-8, every integer exact);
//   * what pass A cannot certify waits in the wave's own 64 entry ring in LDS — code words, quality words, carried
//     state, ordinal: 24 bytes — no atomics, no barrier: the wave's ballot and a popcount place the entries;
//   * pass C, whenever the next tile's deferred reads would overflow the ring (every ~14 tiles on the 96 x 8 nt
//     benchmark): the wave turns to its ring with all lanes busy — the product form scan over every barcode with the
//     certificate of pamld_lut.hpp (dominant winner, winner and posterior again in fp64) — and stores those reads'
//     records and status words.  Only what still cannot be decided (a second plausible barcode, ~0.9 %) goes to the
//     wave's segment of the HBM worklist for the exact tier (pamld_cross.hpp) and the literal kernel.
// No workgroup barrier after the set up, no HBM round trip for the deferred reads, one launch instead of two.
//
// Arithmetic.  Both passes work in units of the read's all-match product A = prod_j P(match | q_j): the per read tables
// hold products of rho_j = P(mismatch | q_j) / P(match | q_j) over a group's mismatching positions (entry 0 = 1, written
// once per kernel), so a barcode's probability over A is T0[m0] * T1[m1], the winner of a read ON its barcode is exactly
// 1, and a table costs 11 multiplications instead of 24.  A itself only enters the noise term (single precision in
// pass A; pass C forms it in fp64 next to the winner's probability).  The tables are laid out as pamld_lut.hpp's permuted
// ones — lane << 2 | group << 8 | wave bit << 9 | entry << 10 | wave bits << 14 — so one v_perm_b32 makes an address from
// a byte of the mismatch field word in both passes; the neighbour records hold their 16 barcodes as eight pairs in
// that kernel's plane words.
//
// 9 and 10 nt.  The tables stay two groups of four positions (8 KiB a wave); positions 8 and 9 are picked by masks — the flag
// bit spread to a word (v_bfe_i32), rho or 1.0 taken by it (v_bitop3_b32) — and, words that long sitting further apart, 8
// barcodes are listed instead of 16, one per plane word (6.2 % of the reads deferred instead of 5.4 %:
// profiles/tools/sim/list_size.py).  The lookup table has 4^L entries of 16 bits, the barcode's ordinal.
#pragma once

#include "kernel_common.hpp"
#include "pamld_lut.hpp"
#include "pamld_near.hpp"

namespace phq {
namespace stream {

constexpr int FACTOR_ROWS = 129;
constexpr int LIST = 16;                       // barcodes scored in pass A: c and the 15 nearest to it (5 to 8 nt)
// 9 and 10 nt: words that long sit further apart — 8 listed barcodes defer 6.2 % where 16 defer 5.4 %
// (profiles/tools/sim/list_size.py) — and their positions 8 and 9 have no table: the factor of each is picked by a mask
__host__ __device__ constexpr int list_of(int length) { return stream_list(length); }
constexpr int PLANE_WORDS = STREAM_PLANE_WORDS; // a block's list: 8 low + 8 high plane words — 16 barcodes two per word, or 8 one per word
// a barcode's block in LDS: its list — 8 low plane words, 8 high plane words (two listed barcodes per word) —, the weights
// of the bound (how many barcodes that are not listed sit at distance 3, 4, 5 and 6 or more from it), then its counter
// bin, then padding to a stride of 11 quads: an odd number, so the blocks start in all 16 quad places of the banks
// BINS = false (round 4): a codec whose counters do not fit the workgroup — 171 to ~500 barcodes — keeps no bin in its
// blocks (80 bytes a barcode, 144 with prior ratios: 5 and 9 quads, odd strides again); the classified reads' count, distance
// and confidence are then rebuilt from the records (counters.hip), the two rare counts go straight to the counter buffer,
// bin 0 stays in the workgroup's histogram — as pamld_cross_stream.hpp does for its sets
template <bool PRIORS, bool BINS = true> struct Block {
    static constexpr int WORDS = stream_record_words(PRIORS, BINS);
    static constexpr uint32_t BYTES = WORDS * 4u;
    static constexpr uint32_t WEIGHT_AT = 64;      // bytes into the block: the four weights of the bound
    static constexpr uint32_t RATIO_AT = 80;       // PRIORS: the noise term over c's prior, then the prior of each listed barcode over c's
    static constexpr uint32_t BIN_AT = PRIORS ? 144 : 80;
    static constexpr int QUADS = PRIORS ? 9 : 5;   // what pass A reads of it
};
constexpr int RING = 64;                       // entries of a wave's ring
// a ring entry: the code words, the quality words, the carried state, the ordinal — 6 planes up to 8 nt, 8 beyond
// the workgroup histogram of this kernel, per bin and [qc fail | pass filter] half: {count | distance << 32,
// confidence low limb, confidence high limb, low conditional confidence count | low confidence count << 32}; a
// workgroup sees fewer than 2^31 reads of at most 8 nt per launch, so no half word overflows
constexpr int BIN_WORDS = 8;                   // words 0..3 the qc fail half, 4..7 the pass filter half: {count | distance << 32, confidence low limb, confidence high limb, low conditional confidence count | low confidence count << 32}
// the posterior of a read pass A finishes is 1 - y with y a float of at most 5e-4: it is counted as the integer
// 2^38 - round(y 2^38), formed from 32-bit integers and split into the buffer's two limbs
constexpr int PASS_A_FRACTION_BITS = 38;

template <int L, int WAVES, bool PRIORS = false, bool BINS = true> struct Shape {
    static_assert(L >= 5 && L <= 10, "two groups of 4 positions, and up to two positions picked by masks");
    static_assert(WAVES >= 2 && WAVES <= 16 && WAVES % 2 == 0, "pairs of waves share a 16 KiB block of tables");
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int TABLED = L > 8 ? 8 : L;                  // positions that go through the tables
    static constexpr int TAIL = L - TABLED;                       // positions 8 .. L - 1: a mask picks rho or 1
    static constexpr int ROW_WORDS = stream_row_words(L);
    static constexpr int LISTED = list_of(L);
    static constexpr int RING_PLANES = 2 * ROW_WORDS + 2;
    static constexpr uint32_t TABLE_BYTES = WAVES * 8192u;        // wave w: block w >> 1 of 16 KiB, bit 9 = w & 1
    static constexpr uint32_t FACTOR_AT = TABLE_BYTES;            // FACTOR_ROWS x {P(match), rho} floats; a multiple of 1024: rows are reached by OR
    static constexpr uint32_t DOUBLE_AT = FACTOR_AT + 1040;       // FACTOR_ROWS x {P(match), P(mismatch)} doubles
    static constexpr uint32_t RING_AT = DOUBLE_AT + FACTOR_ROWS * 16;
    static constexpr uint32_t RING_BYTES = stream_ring_bytes(L);
    static_assert(RING_BYTES == RING_PLANES * RING * 4, "the host's map of the block");
    static constexpr uint32_t WORDS_AT = RING_AT + WAVES * RING_BYTES;        // pamld_lut.hpp's plane words of the codec
    __host__ __device__ static constexpr uint32_t words_bytes(int n_barcode) { return stream_words_bytes(n_barcode, PRIORS, L); }
    __host__ __device__ static constexpr uint32_t prior_at(int n_barcode) { return WORDS_AT + stream_words_bytes(n_barcode, false, L); }        // PRIORS: the codec's priors, single precision, padded to blocks of 8
    __host__ __device__ static constexpr uint32_t block_at(int n_barcode) { return WORDS_AT + words_bytes(n_barcode); }      // the barcodes' blocks
    static_assert(WORDS_AT + stream_words_bytes(96, PRIORS, L) == stream_block_at(WAVES, 96, PRIORS, L), "the host puts block addresses into the lookup table");
    // bin 0 (unclassified) and the two job words in front of it, behind the blocks
    __host__ __device__ static constexpr uint32_t histogram_at(int n_barcode) { return block_at(n_barcode) + static_cast<uint32_t>(n_barcode) * Block<PRIORS, BINS>::BYTES; }
    __host__ __device__ static constexpr uint32_t lds_bytes(int n_barcode) { return histogram_at(n_barcode) + 8u * (STAT_JOB_WORDS + BIN_WORDS); }
    __host__ __device__ static constexpr int positions(int g) { return TABLED - 4 * g >= 4 ? 4 : TABLED - 4 * g; }
    static constexpr uint32_t VALID1 = L >= 8 ? 0xffffffffu : ((1u << (8 * (L - 4))) - 1u);      // the bytes of positions 4 .. 7 in the second word
    static constexpr uint32_t VALID2 = TAIL == 0 ? 0u : ((1u << (8 * TAIL)) - 1u);               // ... of positions 8 .. L - 1 in the third
    __host__ __device__ static constexpr uint32_t valid(int w) { return w == 0 ? 0xffffffffu : w == 1 ? VALID1 : VALID2; }
    static_assert(FACTOR_AT % 1024 == 0 && DOUBLE_AT % 16 == 0 && RING_AT % 16 == 0 && WORDS_AT % 16 == 0, "alignment");
};

typedef uint32_t Quad __attribute__((ext_vector_type(4)));
typedef float Pair __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const float* LdsFloats;
typedef __attribute__((address_space(3))) const Quad* LdsQuads;
typedef __attribute__((address_space(3))) const Pair* LdsPairs;
using near::and_or;
using lut::low_bit;

// a bin's words by their LDS address (bin 0 behind the job words, a barcode's bin inside its block)
__device__ __forceinline__ void bin_add(unsigned char* lds, uint32_t bin_address, uint32_t pass_filter, uint32_t word, unsigned long long value) {
#ifdef PHQ_DIAG_NO_ATOMICS
    return;
#endif
    atomicAdd(reinterpret_cast<unsigned long long*>(lds + bin_address + pass_filter * 32u + word * 8u), value);
}

// STATE: where a read's carried state comes from (pamld_near.hpp): 0 nothing to load, 1 the input flag bytes, 2 the
// status word of the stage before
template <int L, int STATE, int WAVES, bool PRIORS, bool BINS = true>
// one workgroup per CU: the LDS it takes fixes the occupancy at WAVES / 4 waves per SIMD, and the compiler is told so — left
// to assume that fewer registers would buy more waves, it serialises the table reads into pairs, each waited for
__global__ __launch_bounds__((Shape<L, WAVES, PRIORS, BINS>::THREADS)) __attribute__((amdgpu_waves_per_eu(WAVES / 4, (WAVES + 3) / 4)))
void pamld_stream_kernel(StageArgs args) {
    typedef Shape<L, WAVES, PRIORS, BINS> Layout;
    constexpr int BLOCK_WORDS = Block<PRIORS, BINS>::WORDS;
    constexpr uint32_t BLOCK_BYTES = Block<PRIORS, BINS>::BYTES;
    constexpr uint32_t BIN_AT = Block<PRIORS, BINS>::BIN_AT;
    constexpr int THREADS = Layout::THREADS;
    constexpr uint32_t VALID1 = Layout::VALID1;
    constexpr uint32_t VALID2 = Layout::VALID2;
    constexpr int TAIL = Layout::TAIL;
    constexpr int ROW_WORDS = Layout::ROW_WORDS;
    constexpr int LISTED = Layout::LISTED;                   // barcodes scored in pass A
    constexpr int RHO = TAIL ? 10 : 8;                       // slots of a read's rho row
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    float2* const factor_single = reinterpret_cast<float2*>(lds_raw + Layout::FACTOR_AT);                 // [row] {P(match), rho}
    double (*const factor_double)[2] = reinterpret_cast<double (*)[2]>(lds_raw + Layout::DOUBLE_AT);      // [row][0 match | 1 mismatch]
    uint32_t* const lds_words = reinterpret_cast<uint32_t*>(lds_raw + Layout::WORDS_AT);
    DecoderRef d = *constant_cast(args.decoder);
    const int n_barcode = d.n_barcode;
    uint32_t* const blocks_of_barcodes = reinterpret_cast<uint32_t*>(lds_raw + Layout::block_at(n_barcode));
    const uint32_t block_base = Layout::block_at(n_barcode);
    const uint32_t histogram_base = Layout::histogram_at(n_barcode);                    // the two job words, then bin 0
    unsigned long long* const histogram = reinterpret_cast<unsigned long long*>(lds_raw + histogram_base);

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // wave uniform: tiles, segments and bases stay in scalar registers
    const uint32_t lane = threadIdx.x & 63u;
    // this lane's corner of the tables: everything of an entry address but group (bit 8) and entry (bits 10..13)
    const uint32_t wave_byte = ((wave & 1u) << 1) | (((wave >> 1) & 3u) << 6);        // the wave's share of byte 1 of an address
    const uint32_t lane_low = (lane << 2) | ((wave >> 3) << 16);                      // bytes 0 and 2 of an address (v_perm_b32 picks them)
    const uint32_t lane_const = lane_low | (wave_byte << 8);
    const uint32_t wave_bytes = wave_byte * 0x01010101u;

    for(int i = threadIdx.x; i < FACTOR_ROWS; i += THREADS) {
        const double m = i < 128 ? d.match_probability[i] : d.uniform_probability;
        const double e = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        factor_single[i] = make_float2(static_cast<float>(m), static_cast<float>(e / m));
        factor_double[i][0] = m;
        factor_double[i][1] = e;
    }
    {
        // per block of 8 barcodes: the low plane words of the block, then its high plane words (pamld_lut.hpp)
        const int padded = (n_barcode + 7) & ~7;
        if(TAIL == 0) {
            for(int w = threadIdx.x; w < padded >> 1; w += THREADS) {
                const uint32_t sliced = d.barcode_packed2[w];
                const int at = 8 * (w / 4) + (w % 4);
                lds_words[at] = lut::low_plane(sliced);
                lds_words[at + 4] = lut::high_plane(sliced);
            }
        } else {
            // past 8 nt a word holds one barcode: per block of 8 barcodes eight low plane words, then eight high plane words
            for(int w = threadIdx.x; w < padded; w += THREADS) {
                const uint32_t sliced = d.barcode_packed2[w];
                const int at = 16 * (w / 8) + (w % 8);
                lds_words[at] = lut::low_plane(sliced);
                lds_words[at + 8] = lut::high_plane(sliced);
            }
        }
    }
    float* const lds_prior = reinterpret_cast<float*>(lds_raw + Layout::prior_at(n_barcode));
    if(PRIORS) { for(int b = threadIdx.x; b < ((n_barcode + 7) & ~7); b += THREADS) { lds_prior[b] = d.concentration_single[b]; } }      // padded with zeros by the host
    for(int w = threadIdx.x; w < n_barcode * BLOCK_WORDS; w += THREADS) { blocks_of_barcodes[w] = d.stream_record[w]; }      // lists and weights; the bins come as zeros
    for(int i = threadIdx.x; i < STAT_JOB_WORDS + BIN_WORDS; i += THREADS) { histogram[i] = 0; }
    // entry 0 of both tables (no mismatch in the group) is 1 in units of the all-match product: written once
    *reinterpret_cast<float*>(lds_raw + lane_const) = 1.0f;
    *reinterpret_cast<float*>(lds_raw + lane_const + 256u) = 1.0f;
    __syncthreads();

    const int64_t n = args.n_read;
    const int64_t tiles = (n + 63) / 64;
    const int64_t tile_step = static_cast<int64_t>(gridDim.x) * WAVES;
    const void* const nearest = d.stream_table;      // 32 bit entries (block address | index) up to 8 nt, 16 bit ordinals past it
    uint32_t* const ring = reinterpret_cast<uint32_t*>(lds_raw + Layout::RING_AT + wave * Layout::RING_BYTES);
    // this wave's segment of the HBM worklist (what pass C cannot decide either)
    const int64_t own_segment = static_cast<int64_t>(blockIdx.x) * WAVES + wave;
    uint32_t* const segment = args.worklist + own_segment * args.worklist_stride;
    uint32_t segment_count = 0;
    uint32_t ring_count = 0;

    const int high_quality_threshold = min(max(d.high_quality_threshold, 0), 128);
    const uint32_t high_quality_addend = static_cast<uint32_t>(128 - high_quality_threshold) * 0x01010101u;
    // pass A tells a high quality mismatch of a read one mismatch from c by its rho: rho falls as the quality rises from
    // 1 on; quality 0 (rho = 1, a value no other row holds) is below every threshold but 0
    const float rho_at_threshold = high_quality_threshold <= 0 ? __builtin_inff() : high_quality_threshold >= 128 ? -1.0f : factor_single[high_quality_threshold].y;
    const bool zero_is_high = high_quality_threshold <= 0;
    const float uniform_noise_over_prior = static_cast<float>(d.adjusted_noise_probability / d.concentration[0]);
    const float noise_floor = static_cast<float>(d.random_barcode_probability) * near::SLOP;
    const double confidence_threshold = d.confidence_threshold;
    // pass A decides the confidence filter on y = 1 - posterior, a float: y against 1 - threshold as a float, further apart
    // than 2e-8 plus what that float is off by — then the fp64 comparison of the posterior it reports says the same
    const float y_threshold = static_cast<float>(1.0 - confidence_threshold);
    const float y_guard = static_cast<float>((2e-8 + fabs(static_cast<double>(y_threshold) - (1.0 - confidence_threshold))) * 1.001 + 1e-12);
    const double common_prior = d.concentration[0];
    uint32_t outgoing_pass = 0;                       // TranscodingDecoder::count / pf_count (transcode.h:61-64): per wave, added once
    uint32_t outgoing_fail = 0;
    const int hq_threshold = d.high_quality_distance_threshold;
    const bool multiplexing = d.multiplexing != 0;
    const bool identity_index = d.identity_index != 0;
    const bool last_stage = args.last_stage != 0;
    constexpr uint32_t FULL_MASK = (L >= 8 ? 0x3c3cu : (0x3cu | (((1u << (L - 4)) - 1u) << 10))) | (TAIL ? (((1u << TAIL) - 1u) << 18) : 0u);       // the flag bit of each position, one barcode

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
    // table entry address of (barcode h of a word pair, group g): byte 2 h + g of the fields word to byte 1, the lane's
    // bytes 0 and 2; the group is the load's immediate offset
    auto entry = [&](uint32_t fields, int byte, int g) -> float {
        const uint32_t address = __builtin_amdgcn_perm(fields, lane_low, 0x0c020400u | (static_cast<uint32_t>(byte) << 8));
        return ((LdsFloats)static_cast<uintptr_t>(address))[g * 64];
    };
    // the normalised tables of a read from its rho row
    auto build_tables = [&](const float (&rho)[RHO], float (&ge)[3][5]) {
#pragma unroll
        for(int g = 0; g < 2; ++g) {
            const int positions = Layout::positions(g);
            const int entries = 1 << positions;
            const float r0 = rho[4 * g], r1 = positions > 1 ? rho[4 * g + 1] : 1.0f, r2 = positions > 2 ? rho[4 * g + 2] : 1.0f, r3 = positions > 3 ? rho[4 * g + 3] : 1.0f;
            const float lo[4] = {1.0f, r0, r1, r0 * r1};
            const float hi[4] = {1.0f, r2, r3, r2 * r3};
            float t[16];
            float by_count[5] = {1.0f, 0, 0, 0, 0};        // the largest entry with exactly a mismatches
#pragma unroll
            for(int m = 1; m < 16; ++m) {
                if(m < entries) {
                    t[m] = (m >> 2) == 0 ? lo[m & 3] : (m & 3) == 0 ? hi[m >> 2] : lo[m & 3] * hi[m >> 2];
                    *reinterpret_cast<float*>(lds_raw + lane_const + static_cast<uint32_t>(g) * 256u + static_cast<uint32_t>(m) * 1024u) = t[m];
                    const int a = __builtin_popcount(static_cast<unsigned>(m));
                    by_count[a] = (m == (1 << a) - 1) ? t[m] : fmaxf(by_count[a], t[m]);      // the first entry of a count is 2^a - 1
                }
            }
            // ge[g][a] = the largest entry with a OR MORE mismatches (none past the group's positions): qualities below 4,
            // where a mismatch is more likely than a match, stay covered
#pragma unroll
            for(int a = 4; a >= 0; --a) {
                ge[g][a] = a > positions ? 0.0f : a == positions ? by_count[a] : fmaxf(by_count[a], ge[g][a + 1 > 4 ? 4 : a + 1]);
            }
        }
        // the positions past the tables, for the bound: the largest product over a or more of them
        ge[2][4] = 0.0f; ge[2][3] = 0.0f;
        if(TAIL == 0) { ge[2][0] = 1.0f; ge[2][1] = 0.0f; ge[2][2] = 0.0f; }
        else if(TAIL == 1) { ge[2][2] = 0.0f; ge[2][1] = rho[8]; ge[2][0] = fmaxf(1.0f, rho[8]); }
        else {
            ge[2][2] = rho[8] * rho[9];
            ge[2][1] = fmaxf(fmaxf(rho[8], rho[9]), ge[2][2]);
            ge[2][0] = fmaxf(1.0f, ge[2][1]);
        }
    };

    // ---- pass C: the product form scan (pamld_lut.hpp) over the wave's ring --------------------------------------------
    auto drain = [&](uint32_t count) {
        // the records and status words pass A stored for these reads (placeholders, by other lanes) must have left
        // before this pass stores the real ones to the same places: wait for every store of the wave (and its loads)
        __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0)
        const bool live = lane < count;
        const uint32_t slot = live ? lane : 0u;
        uint32_t code_words[ROW_WORDS];
        uint32_t quality_words[ROW_WORDS];
#pragma unroll
        for(int w = 0; w < ROW_WORDS; ++w) { code_words[w] = ring[w * RING + slot]; quality_words[w] = ring[(ROW_WORDS + w) * RING + slot]; }
        const uint32_t state = ring[2 * ROW_WORDS * RING + slot];
        const int64_t i = static_cast<int64_t>(ring[(2 * ROW_WORDS + 1) * RING + slot]);
        ReadState r;
        if(STATE != 2) { r.qcfail = (state & 1u) != 0; r.channel = 0; }
        else { r.qcfail = (state >> 31) != 0; r.channel = static_cast<int32_t>(state & 0x7fffffffu); }
        bool defer = false;

        uint32_t row[ROW_WORDS];                       // factor rows, 8 bits per position
        uint32_t observed = 0;                         // bit-sliced 2-bit codes; anything outside ACGT arbitrary
        uint32_t high_quality = 0;                     // flag bits (low_bit of the position)
        uint32_t outside = 0;                          // flag bits: not A, C, G or T (mismatches every barcode base)
#pragma unroll
        for(int w = 0; w < ROW_WORDS; ++w) {
            const uint32_t valid = Layout::valid(w);
            const uint32_t x = code_words[w] & (0x0f0f0f0fu & valid);        // BAM codes
            const uint32_t q8 = quality_words[w] & valid;
            const uint32_t q7 = q8 & 0x7f7f7f7fu;                             // the table has 128 rows
            // A, C, G or T = exactly one bit of the nibble: x != 0 and x & (x - 1) == 0   (iupac.h:51-56)
            const uint32_t z = x & ((x | 0x80808080u) - 0x01010101u);
            const uint32_t strict = (x + 0x7f7f7f7fu) & ~(z + 0x7f7f7f7fu) & 0x80808080u;
            const uint32_t strict_bytes = (strict - (strict >> 7)) | strict;                       // 0xff where strict
            // a base outside ACGT costs the uniform quality against every barcode base (row 128), nothing at
            // quality 0 (row 0): both factors equal, whatever its 2-bit code compares to (phred.cpp:62-64)
            const uint32_t nonzero = (q7 + 0x7f7f7f7fu) & 0x80808080u;
            row[w] = (q7 & strict_bytes) | (nonzero & ~strict_bytes);
            const uint32_t codes = gather((((x >> 1) | (x >> 3)) & 0x01010101u) | (((x << 2) | (x << 1)) & 0x10101010u));
            observed |= codes << (8 * w + 2);
            const uint32_t high = ((q7 + high_quality_addend) | q8) & 0x80808080u & valid;
            const uint32_t flags = gather(((~strict & 0x80808080u & valid) >> 7) | (high >> 3));
            outside |= (flags & 0xfu) << (8 * w + 2);
            high_quality |= (flags >> 4) << (8 * w + 2);
        }
        float rho[RHO];
#pragma unroll
        for(int j = 0; j < RHO; ++j) {
            if(j < L) { rho[j] = factor_single[(row[j >> 2] >> (8 * (j & 3))) & 0xffu].y; }
            else { rho[j] = 1.0f; }
        }
        if(TAIL == 0) { observed |= __builtin_amdgcn_alignbit(observed, observed, 16); }        // the same read against both halves of a word
        const uint32_t observed_high = lut::high_plane(observed);
        const uint32_t observed_low = lut::low_plane(observed) | wave_bytes;   // + the wave's address bits
        float unused[3][5];
        build_tables(rho, unused);
        // the factor of a position past the tables: rho where the flag bit is set, else 1 (one v_bfe_i32, one v_bitop3_b32)
        auto tail_factor = [&](uint32_t fields) {
            float factor = 1.0f;
#pragma unroll
            for(int t = 0; t < TAIL; ++t) {
                const uint32_t mask = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(fields), 18 + t, 1));
                const float picked = __uint_as_float(__builtin_amdgcn_bitop3_b32(mask, __float_as_uint(rho[8 + t]), 0x3f800000u, 0xCA));
                factor = t == 0 ? picked : factor * picked;
            }
            return factor;
        };

        // Single precision scan, eight barcodes at a time (pamld_lut.hpp): per block its maximum and its sum; across
        // blocks the running maximum, which block holds it, that block's sum set aside, the sum of every other block
        float others_sum = 0;
        float held_sum = 0;
        float best_value = 0;
        float runner_up = 0;
        int best_fold = -1;
        constexpr int BLOCK_PLANE_WORDS = TAIL ? 16 : 8;           // plane words of a block of 8 barcodes
        auto block_values = [&](const uint32_t (&w)[BLOCK_PLANE_WORDS], float (&p)[8]) {
            if(TAIL == 0) {
#pragma unroll
                for(int k = 0; k < 4; ++k) {
                    const uint32_t fields = (observed_low ^ w[k]) | (observed_high ^ w[4 + k]);
#pragma unroll
                    for(int h = 0; h < 2; ++h) { p[2 * k + h] = entry(fields, 2 * h, 0) * entry(fields, 2 * h + 1, 1); }
                }
            } else {
#pragma unroll
                for(int k = 0; k < 8; ++k) {
                    const uint32_t fields = (observed_low ^ w[k]) | (observed_high ^ w[(BLOCK_PLANE_WORDS / 2) + k]);
                    p[k] = (entry(fields, 0, 0) * entry(fields, 1, 1)) * tail_factor(fields);
                }
            }
        };
        const uint4* const blocks = reinterpret_cast<const uint4*>(lds_words);
        const int n_blocks = (n_barcode + 7) >> 3;
#pragma unroll 2
        for(int f = 0; f < n_blocks; ++f) {
            uint32_t w[BLOCK_PLANE_WORDS];
#pragma unroll
            for(int quad = 0; quad < BLOCK_PLANE_WORDS / 4; ++quad) {
                const uint4 four = blocks[(BLOCK_PLANE_WORDS / 4) * f + quad];
                w[4 * quad] = four.x; w[4 * quad + 1] = four.y; w[4 * quad + 2] = four.z; w[4 * quad + 3] = four.w;
            }
            float p[8];
            block_values(w, p);
            if(PRIORS) {
                const float4 lower = reinterpret_cast<const float4*>(lds_prior)[2 * f];
                const float4 upper = reinterpret_cast<const float4*>(lds_prior)[2 * f + 1];
                p[0] *= lower.x; p[1] *= lower.y; p[2] *= lower.z; p[3] *= lower.w; p[4] *= upper.x; p[5] *= upper.y; p[6] *= upper.z; p[7] *= upper.w;
            }
            if(8 * f + 8 > n_barcode) {                 // the padding of a partial last block
#pragma unroll
                for(int k = 1; k < 8; ++k) { if(8 * f + k >= n_barcode) { p[k] = 0.0f; } }
            }
            const float block_max = fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7])));
            const float block_sum = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
            const bool takes_over = block_max > best_value;                      // strict: the first block keeps a tie
            runner_up = __builtin_amdgcn_fmed3f(best_value, block_max, runner_up);
            others_sum += takes_over ? held_sum : block_sum;
            held_sum = takes_over ? block_sum : held_sum;
            best_fold = takes_over ? f : best_fold;
            best_value = takes_over ? block_max : best_value;
        }
        // inside the winning block: the winner (first of equals) and the sum of the other seven, recomputed
        int best = -1;
        uint32_t best_flags = 0;
        float sigma = others_sum;
        const float winner_single = best_value;
        {
            const int block = best_fold < 0 ? 0 : best_fold;
            uint32_t w[BLOCK_PLANE_WORDS];
#pragma unroll
            for(int k = 0; k < BLOCK_PLANE_WORDS; ++k) { w[k] = lds_words[BLOCK_PLANE_WORDS * block + k]; }
            float p[8];
            block_values(w, p);
            if(PRIORS) {
#pragma unroll
                for(int k = 0; k < 8; ++k) { p[k] *= lds_prior[8 * block + k]; }
            }
            {
                const int limit = n_barcode - 8 * block;
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
            const float rest = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
            const float second = fmaxf(fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3])), fmaxf(fmaxf(q[4], q[5]), fmaxf(q[6], q[7])));
            const bool crowded = held_sum - rest >= 1.5f * best_value;         // the block's sum holds the maximum twice
            runner_up = fmaxf(runner_up, crowded ? best_value : second);
            sigma += crowded ? best_value : rest;
            if(best_fold >= 0 && best_value > 0.0f && within >= 0) {
                best = 8 * block + within;
                if(TAIL == 0) {
                    const int at = 8 * block + within / 2;
                    const uint32_t fields = (observed_low ^ lds_words[at]) | (observed_high ^ lds_words[at + 4]);
                    best_flags = ((fields >> (16 * (best & 1))) | outside) & FULL_MASK;
                } else {
                    const int at = 16 * block + within;
                    const uint32_t fields = (observed_low ^ lds_words[at]) | (observed_high ^ lds_words[at + 8]);
                    best_flags = (fields | outside) & FULL_MASK;
                }
            }
        }

        DevResult out;
        out.index = 0; out.distance = 0; out.hq_distance = 0; out.status = STATUS_UNCLASSIFIED; out.qcfail = 0; out.confidence = 0;
        if(best < 0) {
            defer = true;
        } else {
            // the winner's distances, its probability and the all-match product in fp64 from the fp64 factor rows
            int distance = __popc(best_flags);
            int hq_distance = __popc(best_flags & high_quality);
            double best_conditional = 1.0;
            double everything_matches = 1.0;
#pragma unroll
            for(int j = 0; j < L; ++j) {
                const uint32_t k = (row[j >> 2] >> (8 * (j & 3))) & 0xffu;
                const double match = factor_double[k][0];
                const double mismatch = factor_double[k][1];
                best_conditional *= ((best_flags >> low_bit(j)) & 1u) ? mismatch : match;
                everything_matches *= match;
            }
            const double prior_of_best = PRIORS ? d.concentration[best] : common_prior;
            const double best_p = best_conditional * prior_of_best;
            // everything but the winner: a single precision sum in units of the all-match product; only ever compared
            // against and added to much larger terms
            const double others = PRIORS ? static_cast<double>(sigma) * everything_matches : (static_cast<double>(sigma) * everything_matches) * common_prior;
            const double total = (best_p + others) + d.adjusted_noise_probability;   // pamld.cpp:61-86
            double confidence = best_p / total;                                       // pamld.cpp:92
            const double noise_margin = best_conditional - d.random_barcode_probability;
            const double confidence_margin = confidence - confidence_threshold;
            // the certificate of pamld_lut.hpp: a read that clearly fails the noise filter only needs its argmax; otherwise a
            // dominant winner and no decision within 2e-8 of its threshold
            const bool noise_clear = fabs(noise_margin) > 1e-9 * d.random_barcode_probability;
            defer = !noise_clear
                 || (noise_margin <= 0 ? !(winner_single >= lut::SMALLEST_CERTAIN && runner_up <= lut::RUNNER_UP * winner_single)
                                       : (!(others <= lut::dominance(L) * best_p) || !(fabs(confidence_margin) > 2e-8)));
            if(!defer) {
                uint32_t bin = static_cast<uint32_t>(best) + 1u;
                uint32_t bin_address = block_base + static_cast<uint32_t>(best) * BLOCK_BYTES + BIN_AT;
                uint8_t status = STATUS_PASS;
                if(noise_margin > 0) {                                                // pamld.cpp:96
                    if(confidence_margin > 0) {                                       // pamld.cpp:99
                        if(hq_threshold > 0 && hq_distance >= hq_threshold) { r.qcfail = true; }
                        if(live && BINS) {
                            const unsigned long long fixed = static_cast<unsigned long long>(__double2ll_rn(confidence * 4503599627370496.0));      // 2^-52 fixed point, two limbs
                            bin_add(lds_raw, bin_address, r.qcfail ? 0u : 1u, 1, fixed & ((1ull << CONFIDENCE_LIMB_BITS) - 1));
                            bin_add(lds_raw, bin_address, r.qcfail ? 0u : 1u, 2, fixed >> CONFIDENCE_LIMB_BITS);
                        }
                    } else {
                        if(live) {                                                                // low confidence count
                            if(BINS) { bin_add(lds_raw, bin_address, 0, 3, 1ull << 32); }
                            else { atomicAdd(&args.stats[d.stats_offset + static_cast<int64_t>(bin) * STAT_WORDS_PER_BIN + FIELD_LOW_CONFIDENCE], 1ull); }
                        }
                        r.qcfail = true;
                        status = STATUS_LOW_CONFIDENCE;
                    }
                } else {
                    if(live) {                                                                    // low conditional confidence count
                        if(BINS) { bin_add(lds_raw, bin_address, 0, 3, 1ull); }
                        else { atomicAdd(&args.stats[d.stats_offset + static_cast<int64_t>(bin) * STAT_WORDS_PER_BIN + FIELD_LOW_CONDITIONAL_CONFIDENCE], 1ull); }
                    }
                    r.qcfail = true;
                    bin = 0;
                    bin_address = histogram_base + 8u * STAT_JOB_WORDS;
                    distance = 0;
                    hq_distance = 0;
                    confidence = 0;
                    status = STATUS_NOISE;
                }
                const int32_t index_of_bin = bin == 0 ? 0 : identity_index ? static_cast<int32_t>(bin) : d.barcode_index[bin - 1];
                // Decoder<T>::classify then Classifier<T>::classify (decoder.h:68-76, classifier.h:78-86)
                // (BINS = false: a classified read's count and distance are rebuilt from its record; bin 0 is counted here)
                if(live && (BINS || bin == 0)) { bin_add(lds_raw, bin_address, r.qcfail ? 0u : 1u, 0, 1ull | (static_cast<unsigned long long>(bin > 0 ? distance : 0) << 32)); }
                if(multiplexing) { r.channel = index_of_bin; }
                out.index = index_of_bin;
                out.distance = static_cast<uint8_t>(distance);
                out.hq_distance = static_cast<uint8_t>(hq_distance);
                out.status = status;
                out.confidence = confidence;
            }
        }
        out.qcfail = r.qcfail ? 1 : 0;
        if(live) {
            // a read handed on keeps its placeholder record and the state it came with (the next tier writes both)
            if(!defer) {
                stream_result(&args.result[i], out);
                __builtin_nontemporal_store((static_cast<uint32_t>(r.channel) & 0x7fffffffu) | (r.qcfail ? 0x80000000u : 0u), &args.word[i]);
            }
        }
        const unsigned long long handed_on = __ballot(live && defer);
        if(handed_on) {
            if(live && defer) { segment[segment_count + __popcll(handed_on & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(i); }
            segment_count += static_cast<uint32_t>(__popcll(handed_on));
        }
        if(last_stage) {                                  // TranscodingDecoder::count / pf_count (transcode.h:61-64)
            outgoing_pass += static_cast<uint32_t>(__popcll(__ballot(live && !defer && !r.qcfail)));
            outgoing_fail += static_cast<uint32_t>(__popcll(__ballot(live && !defer && r.qcfail)));
        }
    };

    // ---- the wave's loop over its tiles: pass A, then the ring --------------------------------------------------------
    // Software pipeline, so that no wait of a tile is for something asked in that tile: a tile's inputs are fetched TWO
    // tiles ahead, its key is formed and looked up ONE tile ahead (first thing in the tile before — in front of that tile's
    // stores, so that waiting for the lookup never waits for them: vector memory operations complete in issue order).
    struct Keyed { Inputs in; uint32_t sliced; uint32_t looked_up; bool all_strict; };
    auto key_and_look_up = [&](const Inputs& in, Keyed& to) {
        // BAM codes, four positions per word.  A, C, G or T = exactly one bit of the nibble (iupac.h:51-56): every valid
        // byte is not zero and the words hold L bits in all; a read with anything else goes to the ring
        const uint32_t x0 = in.code[0] & 0x0f0f0f0fu;
        const uint32_t x1 = in.code[1] & (0x0f0f0f0fu & VALID1);
        const uint32_t x2 = TAIL ? in.code[ROW_WORDS - 1] & (0x0f0f0f0fu & VALID2) : 0u;
        const uint32_t every_byte = (x0 + 0x7f7f7f7fu) & ((x1 + 0x7f7f7f7fu) | ~(0x80808080u & VALID1)) & ((x2 + 0x7f7f7f7fu) | ~(0x80808080u & VALID2)) & 0x80808080u;
        // ... and so does one with a quality byte of 128 or more: it counts as high quality whatever row its low bits
        // name, which the rho test of pass A cannot see
        const uint32_t beyond_rows = (in.quality[0] | (in.quality[1] & VALID1) | (TAIL ? in.quality[ROW_WORDS - 1] & VALID2 : 0u)) & 0x80808080u;
        to.all_strict = (__popc(x0) + __popc(x1) + __popc(x2)) == L && every_byte == 0x80808080u && beyond_rows == 0;
        // 2-bit codes A, C, G, T -> 0..3, bit-sliced as pamld_lut.hpp has them: position 4 f + k with its low bit at
        // 8 f + 2 + k and its high bit 4 places up.  Bits 2..17 of that word (and the bits of positions 8 and 9) are the key of the lookup.
        const uint32_t codes0 = gather((((x0 >> 1) | (x0 >> 3)) & 0x01010101u) | (((x0 << 2) | (x0 << 1)) & 0x10101010u));
        const uint32_t codes1 = gather((((x1 >> 1) | (x1 >> 3)) & 0x01010101u) | (((x1 << 2) | (x1 << 1)) & 0x10101010u));
        to.sliced = (codes0 << 2) | (codes1 << 10);
        uint32_t key = to.sliced >> 2;
        if(TAIL) {
            const uint32_t codes2 = gather((((x2 >> 1) | (x2 >> 3)) & 0x01010101u) | (((x2 << 2) | (x2 << 1)) & 0x10101010u));
            to.sliced |= codes2 << 18;
            key |= ((codes2 & ((1u << TAIL) - 1u)) << 16) | (((codes2 >> 4) & ((1u << TAIL) - 1u)) << (16 + TAIL));
        }
        // the barcode this word is (next to), if any.  Past 8 nt the table has 4^L entries and holds the barcode's ORDINAL in 16
        // bits (2 MB for 10 nt, where 32 bit entries were 4 MB and missed in L2 for a third of the reads)
        to.looked_up = TAIL ? static_cast<uint32_t>(static_cast<const uint16_t*>(nearest)[key]) : static_cast<const uint32_t*>(nearest)[key];
        to.in = in;
    };
#ifdef PHQ_DIAG_STAGGER
    // the waves of a SIMD (w, w + 4, w + 8) and the SIMDs among themselves start out of phase
    for(uint32_t k = 0; k < ((wave >> 2) * 4u + (wave & 3u)) * PHQ_DIAG_STAGGER; ++k) { __builtin_amdgcn_s_sleep(8); }
#endif
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
        unsigned long long deferred = 0;
        const Inputs in = current.in;
        int64_t read_ordinal = 0;
        if(tile < tiles) {
            const int64_t first = tile * 64;
            const bool active = first + lane < n;
            const uint32_t at = active ? lane : static_cast<uint32_t>(n - 1 - first);
            read_ordinal = first + at;
            const uint32_t sliced = current.sliced;
            const uint32_t looked_up = current.looked_up;
            const bool all_strict = current.all_strict;
            // the tile after this one: its key and lookup, and the inputs of the one after that (past the last tile: that
            // tile again, unused)
            Keyed following;
            key_and_look_up(next, following);
            fetch(min(tile + 2 * tile_step, tiles - 1), next);
            current = following;

            bool incoming_qcfail;
            uint32_t incoming_channel;
            if(STATE != 2) { incoming_qcfail = (in.state & 1u) != 0; incoming_channel = 0; }
            else { incoming_qcfail = (in.state >> 31) != 0; incoming_channel = in.state & 0x7fffffffu; }
            const uint32_t q0 = in.quality[0] & 0x7f7f7f7fu;                  // the table has 128 rows
            const uint32_t q1 = in.quality[1] & (0x7f7f7f7fu & VALID1);
            const uint32_t q2 = TAIL ? in.quality[ROW_WORDS - 1] & (0x7f7f7f7fu & VALID2) : 0u;
#ifdef PHQ_DIAG_EXTRA_VALU
            { uint32_t sink = lane;
#pragma unroll
              for(int k = 0; k < 64; ++k) { asm volatile("v_add_u32 %0, %0, 1" : "+v"(sink)); }
              asm volatile("" :: "v"(sink)); }
#endif
#ifdef PHQ_DIAG_EXTRA_SLOW
            { uint32_t sink = lane;
#pragma unroll
              for(int k = 0; k < 64; ++k) { asm volatile("v_perm_b32 %0, %0, %0, %0" : "+v"(sink)); }
              asm volatile("" :: "v"(sink)); }
#endif
            // the lookup names the barcode by the LDS address of its block (low 18 bits) and its index (above them); a word
            // more than one mismatch from every barcode comes back with the index field all ones and the first block's address
            const bool known = TAIL ? looked_up != 0xffffu : (looked_up >> 18) != 0x3fffu;
            const uint32_t block = TAIL ? block_base + (known ? looked_up : 0u) * BLOCK_BYTES : looked_up & 0x3ffffu;
            const uint32_t index = TAIL ? looked_up + 1u : looked_up >> 18;          // barcode_index[c] (past 8 nt: the identity, host checked)
            // the list of c — eight pairs of barcodes as plane words, then the weights of the bound — asked for together
            // with the factor rows: one wait for both
            Quad record[Block<PRIORS, BINS>::QUADS];
#pragma unroll
            for(int quad = 0; quad < Block<PRIORS, BINS>::QUADS; ++quad) { record[quad] = *(LdsQuads)static_cast<uintptr_t>(block + 16u * quad); }
            auto record_word = [&](int w) { return record[w >> 2][w & 3]; };
            const uint32_t observed = TAIL ? sliced : (sliced | __builtin_amdgcn_alignbit(sliced, sliced, 16));      // up to 8 nt: against both halves of a word
            const uint32_t observed_high = lut::high_plane(observed);
            const uint32_t observed_low = lut::low_plane(observed) | wave_bytes;

            // per position {P(match), rho} of its quality
            float rho[RHO];
            float everything_matches = 1.0f;
#pragma unroll
            for(int j = 0; j < RHO; ++j) {
                if(j < L) {
                    const uint32_t q = j < 4 ? q0 : j < 8 ? q1 : q2;
                    const int byte = j & 3;
                    const uint32_t shifted = byte == 0 ? (q << 3) : (q >> (8 * byte - 3));
                    const Pair pair = *(LdsPairs)static_cast<uintptr_t>(and_or(shifted, 0x3f8u, Layout::FACTOR_AT));
                    everything_matches = j == 0 ? pair.x : everything_matches * pair.x;
                    rho[j] = pair.y;
                } else { rho[j] = 1.0f; }
            }
            float ge[3][5];
            build_tables(rho, ge);

            // all the table reads are asked for before the first is used: the wave has three to a SIMD for company, so a
            // dependent chain of short read groups would expose the LDS latency eight times over
            constexpr int FIELD_WORDS = TAIL ? LISTED : LISTED / 2;        // up to 8 nt a word holds the flags of two listed barcodes
            uint32_t fields[FIELD_WORDS];
#pragma unroll
            for(int w = 0; w < FIELD_WORDS; ++w) { fields[w] = (observed_low ^ record_word(w)) | (observed_high ^ record_word(PLANE_WORDS / 2 + w)); }
            float first_group[LISTED];
            float second_group[LISTED];
#pragma unroll
            for(int o = 0; o < LISTED; ++o) {
                first_group[o] = TAIL ? entry(fields[o], 0, 0) : entry(fields[o >> 1], 2 * (o & 1), 0);
                second_group[o] = TAIL ? entry(fields[o], 1, 1) : entry(fields[o >> 1], 2 * (o & 1) + 1, 1);
            }
            // H[n]: the largest product over n or more mismatching positions, n = 3 .. 6, in units of the all-match product
            // (while the reads are under way): over the two tabled groups, then with the positions past them
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
            // a position past the tables: rho where its flag bit is set, else 1 (one v_bfe_i32, one v_bitop3_b32)
            auto tail_factor = [&](uint32_t word) {
                float factor = 1.0f;
#pragma unroll
                for(int t = 0; t < TAIL; ++t) {
                    const uint32_t mask = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(word), 18 + t, 1));
                    const float picked = __uint_as_float(__builtin_amdgcn_bitop3_b32(mask, __float_as_uint(rho[8 + t]), 0x3f800000u, 0xCA));
                    factor = t == 0 ? picked : factor * picked;
                }
                return factor;
            };
            const uint32_t mismatches = fields[0] & FULL_MASK;      // of c: the flag bits of its mismatching positions
            const float winner = TAIL ? (first_group[0] * second_group[0]) * tail_factor(fields[0]) : first_group[0] * second_group[0];  // P(r|c) over the all-match product: 1 for a read ON c
            float others;                                           // the other listed barcodes
            {
                float p[LIST];
#pragma unroll
                for(int o = 1; o < LIST; ++o) {
                    if(o < LISTED) {
                        p[o] = first_group[o] * second_group[o];
                        if(TAIL) { p[o] *= tail_factor(fields[o]); }
                        if(PRIORS) { p[o] *= __uint_as_float(record_word(PLANE_WORDS + 4 + o)); }            // the barcode's prior over c's
                    } else { p[o] = 0.0f; }
                }
                others = LISTED == LIST ? (((p[1] + p[2]) + (p[3] + p[4])) + ((p[5] + p[6]) + (p[7] + p[8]))) + (((p[9] + p[10]) + (p[11] + p[12])) + ((p[13] + p[14]) + p[15]))
                                        : ((p[1] + p[2]) + (p[3] + p[4])) + ((p[5] + p[6]) + p[7]);
            }

            // What was not scored.  A barcode n mismatches from c mismatches a read one mismatch from c (at position p) in at
            // least n - 1 positions besides p, and p itself costs it at most max(rho_p, 1); rho_p is the winner itself, and a
            // product over n - 1 or more positions besides p is at most H[n] / rho_p (with p it is one over n or more).  For
            // a read ON c the same expression is H[n]: the winner is exactly 1.  So, with w_n barcodes n mismatches from c
            // (w_6: six or more; the host's counts), everything not scored is at most
            //     max(winner, 1) * sum_n w_n * H[n] / winner
            // (rounds 2 and 3 also took the smaller of that and H[n - 1]: no read of the model is finished by it,
            // profiles/tools/sim/list_size.py).  One reciprocal serves it, the noise term and the posterior.
            const float inverse_both = __builtin_amdgcn_rcpf(winner * everything_matches);
            const float inverse_winner = inverse_both * everything_matches;
            float rest = __uint_as_float(record_word(PLANE_WORDS)) * h[3];
            rest = __builtin_fmaf(__uint_as_float(record_word(PLANE_WORDS + 1)), h[4], rest);
            rest = __builtin_fmaf(__uint_as_float(record_word(PLANE_WORDS + 2)), h[5], rest);
            rest = __builtin_fmaf(__uint_as_float(record_word(PLANE_WORDS + 3)), h[6], rest);
            rest *= (fmaxf(winner, 1.0f) * near::SLOP) * inverse_winner;
            const float x_listed = others * inverse_winner;
            const float x_rest = rest * inverse_winner;
            const float noise_over_prior = PRIORS ? __uint_as_float(record_word(PLANE_WORDS + 4)) : uniform_noise_over_prior;
            const float x_low = __builtin_fmaf(noise_over_prior, inverse_both, x_listed);
            // the posterior is taken a quarter of the way into what was only bounded (pamld_near.hpp: at most 0.75 x_rest off,
            // 7.5e-9 with the rounding): 1 - y, y = x / (1 + x) = x - x^2 + x^3 up to x^4 < 7e-14 (x is at most 5e-4 where it counts)
            const float x_mid = __builtin_fmaf(0.25f, x_rest, x_low);
            const float y = x_mid * __builtin_fmaf(-x_mid, 1.0f - x_mid, 1.0f);
            const double confidence = 1.0 - static_cast<double>(y);            // pamld.cpp:92
            const int k = __popc(mismatches);
            const bool finished = all_strict && known && k <= 1 && winner * everything_matches > noise_floor &&
                                  x_listed <= near::LISTED_MAX && x_low + x_rest <= near::X_MAX && x_rest <= near::REST_MAX &&
                                  fabsf(y - y_threshold) > y_guard;

            // Outcome.  Every lane stores a record and a status word in straight line code — the real ones when the read
            // is finished; a deferred read's record is whatever was computed (pass C or the tiers behind it write it again),
            // its status word the state it came with; a lane past the end of the batch stores its (last) read's again
            const bool high_mismatch = k == 1 && winner <= rho_at_threshold && (zero_is_high || winner != 1.0f);
            const uint32_t hq_distance = high_mismatch ? 1u : 0u;
            const bool confident = y < y_threshold;                                     // pamld.cpp:99
            const bool qcfail = incoming_qcfail || !confident || (hq_threshold > 0 && static_cast<int>(hq_distance) >= hq_threshold);
            const bool qcfail_out = finished ? qcfail : incoming_qcfail;
            {
                Quad out;
                out.x = index;
                out.y = static_cast<uint32_t>(k) | (hq_distance << 8) | ((confident ? STATUS_PASS : STATUS_LOW_CONFIDENCE) << 16) | (qcfail_out ? 1u << 24 : 0u);
                out.z = static_cast<uint32_t>(__double2loint(confidence));
                out.w = static_cast<uint32_t>(__double2hiint(confidence));
                __builtin_nontemporal_store(out, reinterpret_cast<Quad*>(reinterpret_cast<char*>(args.result) + first * 16 + at * 16u));
                const uint32_t channel = (finished && multiplexing) ? index : incoming_channel;
                __builtin_nontemporal_store((channel & 0x7fffffffu) | (qcfail_out ? 0x80000000u : 0u), reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(args.word) + first * 4 + at * 4u));
            }
            if(!BINS) {
                // the counters of a codec this size are rebuilt from the records; only the rare count leaves from here
                if(finished && active && !confident) { atomicAdd(&args.stats[d.stats_offset + static_cast<int64_t>(index) * STAT_WORDS_PER_BIN + FIELD_LOW_CONFIDENCE], 1ull); }      // identity numbering: the index is the bin (count_results_supports)
            } else if(finished && active) {
                // Decoder<T>::classify then Classifier<T>::classify (decoder.h:68-76, classifier.h:78-86): the barcode's bin, in its block
                const uint32_t half = qcfail ? 0u : 1u;
                bin_add(lds_raw, block + BIN_AT, half, 0, 1ull | (static_cast<unsigned long long>(k) << 32));
                if(confident) {
                    // the posterior as the integer 2^38 - round(y 2^38) (y is at most 5e-4 here: its multiple fits 32 bits), as
                    // the buffer's two limbs of 2^-52: every read adds the same two numbers however the reads are batched
                    const uint32_t t = __float2uint_rn(y * static_cast<float>(1ull << PASS_A_FRACTION_BITS));
                    constexpr int SHIFT = CONFIDENCE_FRACTION_BITS - PASS_A_FRACTION_BITS;                 // 14
                    constexpr uint32_t BELOW = (1u << (CONFIDENCE_LIMB_BITS - SHIFT)) - 1u;                 // the 12 bits under the high limb
                    bin_add(lds_raw, block + BIN_AT, half, 1, static_cast<unsigned long long>(((0u - t) & BELOW) << SHIFT));
                    bin_add(lds_raw, block + BIN_AT, half, 2, static_cast<unsigned long long>((1u << CONFIDENCE_LIMB_BITS) - ((t + BELOW) >> (CONFIDENCE_LIMB_BITS - SHIFT))));
                } else {
                    bin_add(lds_raw, block + BIN_AT, 0, 3, 1ull << 32);                            // low confidence count (pamld.cpp:104-106)
                }
            }
            deferred = __ballot(active && !finished);
            if(last_stage) {
                outgoing_pass += static_cast<uint32_t>(__popcll(__ballot(active && finished && !qcfail)));
                outgoing_fail += static_cast<uint32_t>(__popcll(__ballot(active && finished && qcfail)));
            }
            tile += tile_step;
        } else {
            more = false;
        }
        const uint32_t arriving = static_cast<uint32_t>(__popcll(deferred));
        if(ring_count + arriving > RING || (!more && ring_count > 0)) {
            drain(ring_count);
            ring_count = 0;
        }
        if(arriving) {
            if((deferred >> lane) & 1ull) {
                const uint32_t slot = ring_count + static_cast<uint32_t>(__popcll(deferred & ((1ull << lane) - 1ull)));
#pragma unroll
                for(int w = 0; w < ROW_WORDS; ++w) { ring[w * RING + slot] = in.code[w]; ring[(ROW_WORDS + w) * RING + slot] = in.quality[w]; }
                ring[2 * ROW_WORDS * RING + slot] = in.state;
                ring[(2 * ROW_WORDS + 1) * RING + slot] = static_cast<uint32_t>(read_ordinal);
            }
            ring_count += arriving;
        }
    }
    if(lane == 0) {
        args.worklist_count[own_segment] = segment_count;
        if(outgoing_fail) { atomicAdd(&histogram[0], static_cast<unsigned long long>(outgoing_fail)); }
        if(outgoing_pass) { atomicAdd(&histogram[1], static_cast<unsigned long long>(outgoing_pass)); }
    }
    __syncthreads();
    // the workgroup's bins into the counter buffer: one global atomic per field that is not zero
    outgoing_flush(args, histogram);
    for(int i = threadIdx.x; i < (BINS ? n_barcode + 1 : 1) * BIN_WORDS; i += THREADS) {
        const int bin = i / BIN_WORDS;
        const int pass_filter = (i % BIN_WORDS) / 4;
        const int word = i % 4;
        const unsigned long long v = bin == 0 ? histogram[STAT_JOB_WORDS + i]
                                              : *reinterpret_cast<const unsigned long long*>(lds_raw + block_base + static_cast<uint32_t>(bin - 1) * BLOCK_BYTES + BIN_AT + 8u * (i % BIN_WORDS));
        if(v == 0) { continue; }
        unsigned long long* const to = args.stats + d.stats_offset + bin * STAT_WORDS_PER_BIN + pass_filter * STAT_FIELDS;
        if(word == 0) {
            atomicAdd(&to[FIELD_COUNT], v & 0xffffffffull);
            if(v >> 32) { atomicAdd(&to[FIELD_DISTANCE], v >> 32); }
        } else if(word == 1) { atomicAdd(&to[FIELD_CONFIDENCE_LOW], v); }
        else if(word == 2) { atomicAdd(&to[FIELD_CONFIDENCE_HIGH], v); }
        else if(word == 3) {
            if(v & 0xffffffffull) { atomicAdd(&to[FIELD_LOW_CONDITIONAL_CONFIDENCE], v & 0xffffffffull); }
            if(v >> 32) { atomicAdd(&to[FIELD_LOW_CONFIDENCE], v >> 32); }
        }
    }
}

template <int L, int STATE, int WAVES, bool PRIORS, bool BINS = true>
int launch_state(const StageArgs& args, int n_barcode, dim3 grid, hipStream_t stream) {
    static std::atomic<uint64_t> prepared{0};
    if(const int status = allow_large_lds(&pamld_stream_kernel<L, STATE, WAVES, PRIORS, BINS>, prepared)) { return status; }
    typedef Shape<L, WAVES, PRIORS, BINS> Layout;
    hipLaunchKernelGGL((pamld_stream_kernel<L, STATE, WAVES, PRIORS, BINS>), grid, dim3(Layout::THREADS), Layout::lds_bytes(n_barcode), stream, args);
    return static_cast<int>(hipGetLastError());
}

template <int L, int WAVES, bool PRIORS, bool BINS = true>
int launch_one(const StageArgs& args, int n_barcode, dim3 grid, hipStream_t stream) {
    if(!args.first_stage) { return launch_state<L, 2, WAVES, PRIORS, BINS>(args, n_barcode, grid, stream); }
    return args.input_flag ? launch_state<L, 1, WAVES, PRIORS, BINS>(args, n_barcode, grid, stream) : launch_state<L, 0, WAVES, PRIORS, BINS>(args, n_barcode, grid, stream);
}

}  // namespace stream
}  // namespace phq
