// K1 (product form): PamlDecoder<T>::classify for reads whose winner is unambiguous.
//
// pow(base, sum of costs) = product of pow(base, cost): per read and position there are only two factors,
// P(match) = pow(base, tpq[q]) and P(mismatch) = pow(base, q) (or pow(base, 10 log10 4) for a base outside
// ACGT; 1 for q = 0) — reference phred.cpp:44-68, barcode.h:151-163.  So P(r|b) is a product of L factors
// selected by the mismatch mask of (read, barcode).  The kernel
//   * builds, per read, one 16 entry table per group of 4 positions (every product of 4 factors), in LDS;
//   * gets mismatch masks with XOR + rotate + OR on bit-sliced 2-bit bases (for barcodes of up to 8 nt one
//     32-bit word carries TWO barcodes), which leaves each group's 4-bit table index in its own byte, already
//     scaled; one byte permute makes the LDS address;
//   * multiplies the G table entries, and folds eight barcodes at a time into max / sum / "which block".
// No exp, no Kahan chain: ~7 VALU instructions per (read, barcode) instead of ~75.  Every VALU instruction
// costs one quad-cycle on this part whatever its type, so instruction count is the metric.
//
// The reference's argmax between NEAR-TIED barcodes is decided by the last bits of its own Kahan sums
// (pamld_literal.hpp), and the posterior must hold 1e-6 (held here to 1e-8, like the tier in front of it).  So a read
// is only FINISHED in this kernel when its winner dominates: the summed probability of every other barcode is at most
// dominance(L) of the winner's — 3e-3 to 5e-3.  Then (i) the argmax cannot be a tie, (ii) the tables and the running sum
// can be single precision — their relative error, at most ~4e-7 per table of a barcode's product plus ~1e-6 for the sum
// over the barcodes if every rounding aligns, moves the posterior by at most dominance(L) times that: 9e-9 — while the
// winner's own probability and the posterior are recomputed in fp64 from the fp64 factor tables, and (iii) the noise and
// confidence filters are decided in fp64, unless they fall within 2e-8 of their thresholds.  (5e-4 and 1e-9 until the end
// of round 2: the looser bound halves what reaches the exact tier on the 96 x 8 nt benchmark.)  A read that clearly
// fails the noise filter reports nothing but its argmax (pamld.cpp:113-119): it is finished as soon as the runner
// up is further than single precision can blur (RUNNER_UP).  Every other read (a second plausible barcode, tied
// noise reads: ~1.4 % of C2) is appended to a worklist that the exact tier (pamld_cross.hpp) and then the literal
// kernel decode right after, on the same stream.
#pragma once

#include "kernel_common.hpp"

namespace phq {
namespace lut {

// others <= dominance(L) * winner, see the header: 9e-9 over the worst relative error of the single precision sum
__host__ __device__ constexpr double dominance(int length) { return 9e-9 / (((length + 3) / 4) * 4.3e-7 + 1e-6); }
constexpr float RUNNER_UP = 0.9999f;          // runner up <= RUNNER_UP * winner: the argmax is certain in single precision
constexpr float SMALLEST_CERTAIN = 1e-30f;    // ... for a winner well inside the normal range (a subnormal one has lost its digits)
constexpr int FACTOR_ROWS = 129;              // 0..127 = quality of an ACGT base, 128 = a base outside ACGT with q >= 1
// Round 5, tied noise reads: a read every barcode of which is clearly under the noise filter reports nothing but its argmax
// (pamld.cpp:113-119), and a third to a half of such reads TIE at the top — two barcodes at the same distance with the same
// qualities under their mismatches are the same product to the last bit of a float, and which of them the reference's own Kahan
// sums prefer is a matter of their last bits (52 % of what reached C2's exact tier, 97 % of H94's: 2.1 % of all its reads, a
// quarter of its step).  The scan keeps, beside the block of eight that holds the maximum, WHICH block holds the runner up and the
// third largest block maximum (four instructions a block); where the third is under NEAR_TOP of the maximum, every barcode outside
// those two blocks is further from the argmax than the floats can blur (RUNNER_UP), and the read is handed on with bit 31 of its
// worklist entry set and the two blocks' numbers in its placeholder record: the exact tier's first pass (pamld_exact_lanes.hip) walks those sixteen
// barcodes literally instead of the exact tier scanning the codec.  Only where that kernel follows (StageArgs::flag_ties).
constexpr float NEAR_TOP = 0.9998f;

// Bit-sliced 2-bit bases (A, C, G, T = 0..3).  Position 4f + k lives in byte f of the word: its low code bit at
// bit 8f + 2 + k, its high code bit 4 places further up, wrapping (bits 6, 7 of the byte and bits 0, 1 of the next
// byte; the last byte wraps to bits 0, 1 of the word).  Then for X = observed ^ expected,
//     X | rotr(X, 4)
// has, in bits 2..5 of byte f, the four mismatch flags of group f = the table index times 4; byte 1 of that word,
// masked, is directly index * 1024 = the byte offset of an entry in an [index][256 lanes] table of floats.
__host__ __device__ constexpr uint32_t low_bit(int position) { return 8u * (position >> 2) + 2u + (position & 3); }
__host__ __device__ constexpr uint32_t high_bit(int position) { return (low_bit(position) + 4u) & 31u; }
constexpr uint32_t FIELD_MASK = 0x3c3c3c3cu;

// In LDS and in the registers of a read the two code bits live in two PLANE words instead: the low bits of all
// positions in one word, the high bits — moved 4 places down, to the same bits 2..5 of each byte — in another, every
// other bit zero.  The mismatch fields of a word are then
//     (observed_low ^ expected_low) | (observed_high ^ expected_high)
// with nothing to mask, and what is OR-ed into the unused bits of observed_low (the wave's share of the table address)
// comes through unchanged: two instructions (v_xor_b32, v_bitop3_b32) per word instead of four.
__device__ __forceinline__ uint32_t low_plane(uint32_t sliced) { return sliced & FIELD_MASK; }
__device__ __forceinline__ uint32_t high_plane(uint32_t sliced) { return __builtin_amdgcn_alignbit(sliced, sliced, 4) & FIELD_MASK; }

// Workgroup size: the 16 entry tables cost 64 bytes of LDS per lane and group.  Up to 8 nt (2 groups) a 512
// lane workgroup = two 256 lane half tables of 32 KiB leaves room for two workgroups per CU = 4 waves per SIMD;
// longer barcodes keep 256 lanes.
//
// LDS is ONE dynamic block laid out by hand, and the kernel declares no static LDS: the block then starts at LDS
// address 0 (checked once per workgroup), the tables sit at its front, and a table address is just
// (field & mask) | lane offset — one v_and_or_b32 — instead of a mask and an add to a base.
template <int L> struct LutGeometry {
    static constexpr int G = (L + 3) / 4;
    static constexpr int THREADS = L <= 8 ? 512 : 256;
    // Table entry (wave w, lane l, group g, index m) lives at byte address
    //     l << 2  |  g << 8  |  m << 10  |  wave bits in what is left of bits 8, 9 and from bit 14 up
    // so that byte 1 of the address is a byte of the (masked) mismatch field word OR-ed with a per lane constant:
    // one v_perm_b32 builds an address, whichever byte of the word the field sits in.  Three groups (9-12 nt)
    // would leave a quarter of that span unused and cost a workgroup of occupancy: they keep the plain
    // [group][index][lane] layout, where a field is shifted to byte 1 first.
    static constexpr bool PERMUTED = G != 3;
    static constexpr uint32_t GROUP_STEP = PERMUTED ? 256u : 16u * 256u * sizeof(float);
    static constexpr uint32_t ENTRY_STEP = 256u * sizeof(float);
    static constexpr uint32_t TABLE_BYTES = (THREADS / 64) * G * 16u * 64u * sizeof(float);
    static constexpr uint32_t TABLE_AT = 0;
    static constexpr uint32_t MATCH_AT = TABLE_AT + TABLE_BYTES;           // FACTOR_ROWS x {match, mismatch} floats
    static constexpr uint32_t FACTOR_AT = MATCH_AT + 1056;                 // FACTOR_ROWS x [match | mismatch] doubles
    static constexpr uint32_t DEFERRED_AT = FACTOR_AT + 129 * 16;          // one counter
    static constexpr uint32_t WORDS_AT = DEFERRED_AT + 16;                 // the barcode words, as two planes: words_bytes()
    // ... then the counter histogram (kernel_common.hpp).  The words region is as long as the codec needs: two
    // workgroups of the 8 nt shape fit a CU with a few KiB to spare, which a fixed 4 KiB region would use up.
    static constexpr int PER_WORD = L <= 8 ? 2 : 1;
    __host__ __device__ static constexpr uint32_t words_bytes(int n_barcode) {
        return static_cast<uint32_t>(((n_barcode + 7) & ~7) / PER_WORD) * 2u * 4u;          // a multiple of 32
    }
    __host__ __device__ static constexpr uint32_t histogram_at(int n_barcode) { return WORDS_AT + words_bytes(n_barcode); }
};

typedef __attribute__((address_space(3))) const float* LdsFloats;         // an LDS address as the hardware takes it

// CONSUMER: instead of every read of the batch, the kernel visits the reads an earlier tier (pamld_near.hpp) left in
// this workgroup's worklist segment, and compacts what it defers itself back into the same segment (in place: a
// barrier per tile keeps every wave's reads of the list ahead of the writes).
template <int L, bool UNIFORM_PRIOR, bool CONSUMER>
__global__ __launch_bounds__(LutGeometry<L>::THREADS) void pamld_lut_kernel(StageArgs args) {
    constexpr int G = (L + 3) / 4;            // groups of 4 positions = bytes of the field word
    constexpr int THREADS = LutGeometry<L>::THREADS;
    constexpr bool PAIRED = L <= 8;           // two barcodes per 32-bit word of 2-bit codes
    constexpr int PER_WORD = PAIRED ? 2 : 1;
    static_assert(L <= 16, "2-bit codes of one barcode fit one word");
    // Round 5: past 8 nt the tables are scaled by 2^(96 / G) a group, every barcode's probability by 2^96: nine Q37 mismatches of
    // a 16 nt barcode are 5e-34 — under SMALLEST_CERTAIN, and a noise read (whose argmax is all that matters) sits that far from
    // every barcode: the floats had lost its winner and the exact tier took it (H94: 2.3 % of all reads).  Scaled back in fp64.
    constexpr int GROUP_SCALE_LOG2 = G >= 3 ? 96 / G : 0;
    constexpr float GROUP_SCALE = static_cast<float>(1ull << GROUP_SCALE_LOG2);
    constexpr double PRODUCT_UNSCALE = G >= 3 ? 1.0 / (static_cast<double>(1ull << 48) * static_cast<double>(1ull << 48)) : 1.0;
    static_assert(G < 3 || GROUP_SCALE_LOG2 * G == 96, "2^96 a product");
    typedef LutGeometry<L> Layout;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    float* const table = reinterpret_cast<float*>(lds_raw + Layout::TABLE_AT);
    float2* const factor_single = reinterpret_cast<float2*>(lds_raw + Layout::MATCH_AT);       // [row] {match, mismatch}
    double (*const factor_double)[2] = reinterpret_cast<double (*)[2]>(lds_raw + Layout::FACTOR_AT);   // [row][0 match | 1 mismatch]
    uint32_t& deferred_total = *reinterpret_cast<uint32_t*>(lds_raw + Layout::DEFERRED_AT);
    DecoderRef d = *constant_cast(args.decoder);
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw + Layout::histogram_at(d.n_barcode));

    for(int i = threadIdx.x; i < FACTOR_ROWS; i += THREADS) {
        const double m = i < 128 ? d.match_probability[i] : d.uniform_probability;
        const double e = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        factor_single[i] = make_float2(static_cast<float>(m), static_cast<float>(e));
        factor_double[i][0] = m;
        factor_double[i][1] = e;
    }
    if(threadIdx.x == 0) { deferred_total = 0; }
    // per block of 8 barcodes: the low plane words of the block, then its high plane words
    uint32_t* const lds_words = reinterpret_cast<uint32_t*>(lds_raw + Layout::WORDS_AT);
    constexpr int BLOCK_WORDS = 8 / PER_WORD;
    {
        const int padded = (d.n_barcode + 7) & ~7;
        const int words = PAIRED ? padded >> 1 : padded;
        for(int w = threadIdx.x; w < words; w += THREADS) {
            const uint32_t sliced = d.barcode_packed2[w];
            const int at = 2 * BLOCK_WORDS * (w / BLOCK_WORDS) + (w % BLOCK_WORDS);
            lds_words[at] = low_plane(sliced);
            lds_words[at + BLOCK_WORDS] = high_plane(sliced);
        }
    }
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    // A consumer takes worklist_group segments one after the other (fewer, longer lived workgroups: the set up above and
    // the flush below are paid once for all of them); a producer owns the segment of its number.
    const int group = CONSUMER ? (args.worklist_group < 1 ? 1 : args.worklist_group) : 1;
    for(int member = 0; member < group; ++member) {
    const int64_t own = CONSUMER ? static_cast<int64_t>(blockIdx.x) * group + member : static_cast<int64_t>(blockIdx.x);
    if(CONSUMER && own >= args.worklist_segments) { break; }
    if(CONSUMER && member > 0) {
        __syncthreads();
        if(threadIdx.x == 0) { deferred_total = 0; }
        __syncthreads();
    }
    uint32_t* const segment = args.worklist + own * args.worklist_stride;
    // The packed barcode words sit in LDS: the scan reads a block's words at a wave uniform address (scalar loads
    // would share a counter with the LDS reads of the scan and serialise with them), the winner's block is
    // read back per lane.
    const PHQ_CONSTANT float* prior = constant_cast(d.concentration_single);
    const int n_barcode = d.n_barcode;
    const int64_t n = CONSUMER ? static_cast<int64_t>(args.worklist_count[own]) : args.n_read;
    const uint32_t carry_stride = (CONSUMER && args.carry) ? static_cast<uint32_t>(args.carry_stride) : 0u;
    const uint32_t* const carry = args.carry + own * CARRY_PLANES * carry_stride;
    const int64_t tiles = (n + THREADS - 1) / THREADS;
    constexpr bool PERMUTED = Layout::PERMUTED;
    constexpr uint32_t GROUP_STEP = Layout::GROUP_STEP;
    constexpr uint32_t ENTRY_STEP = Layout::ENTRY_STEP;
    constexpr uint32_t INDEX_MASK = 0x3c00u;              // index * 1024 at byte 1 of a field word
    // this lane's corner of the tables: everything of an entry address but group and index
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t wave_byte = !PERMUTED ? 0u                                  // bits of byte 1 of the address that name the wave
                             : G == 1 ? ((wave & 3u) | ((wave >> 2) << 6))
                             : G == 2 ? (((wave & 1u) << 1) | ((wave >> 1) << 6))
                             : (wave << 6);
    const uint32_t lane_low = (threadIdx.x & 63u) << 2;
    const uint32_t lane_offset = PERMUTED ? (lane_low | (wave_byte << 8)) : threadIdx.x * static_cast<uint32_t>(sizeof(float));
    const uint32_t wave_bytes = wave_byte * 0x01010101u;
    // quality >= high quality threshold, byte parallel: (quality & 0x7f) + (128 - threshold) reaches bit 7
    const int high_quality_threshold = min(max(d.high_quality_threshold, 0), 128);
    const uint32_t high_quality_addend = static_cast<uint32_t>(128 - high_quality_threshold) * 0x01010101u;
    uint32_t full_mask = 0;                                // the flag bit of each of the L positions
#pragma unroll
    for(int j = 0; j < L; ++j) { full_mask |= 1u << low_bit(j); }

    // A consumer's loop is laid out so that no load waits behind a scattered store (vector memory operations complete in
    // issue order, and the 16 + 4 byte stores of a read's record and status word at its own place are slow to complete):
    //   * a tile's entries and the inputs the producer left beside them are asked for one tile ahead, first thing in a tile;
    //   * a tile's records and status words are stored at its end by every lane in straight line code (a read deferred
    //     again stores its placeholder once more, a lane past the end its last read's record again), so that the compiler can
    //     count them at the next tile's wait for its entries, which were asked for before them;
    //   * no per lane global load in between: the prior of the winner is the common one when the priors are equal, the barcode
    //     index is the ordinal + 1 when the codec is numbered that way (else two cached loads, each waiting for the stores);
    //   * what the tile defers again is written BEHIND the segment's entries when twice their number fits the segment — then
    //     no wave can overwrite an entry another has not read yet, and there is no barrier in the loop — and moved to the
    //     front once at the end; else it is compacted in place behind a barrier per tile.
    constexpr int CARRIED_WORDS = (L + 3) / 4 <= 3 ? (L + 3) / 4 : 0;        // rows of four words are gathered (no carry planes for them)
    const double common_prior = d.concentration[0];
    const bool identity_index = d.identity_index != 0;
    const uint32_t entries_rounded = static_cast<uint32_t>(tiles) * THREADS;
    const bool apart = CONSUMER && 2u * entries_rounded <= static_cast<uint32_t>(args.worklist_stride);
    const uint32_t gathered_at = apart ? entries_rounded : 0u;
    const bool has_carry = CONSUMER && CARRIED_WORDS > 0 && carry_stride > 0;
    struct Ahead { uint32_t read; uint32_t carried[CARRY_PLANES]; };
    auto fetch_entries = [&](int64_t tile, Ahead& to) {
        const int64_t at = tile * THREADS + threadIdx.x;
        const uint32_t entry = static_cast<uint32_t>(at < n ? at : n - 1);
        to.read = segment[entry];
        if(has_carry) {
            const uint32_t beside = entry < carry_stride ? entry : carry_stride - 1u;      // past the carried entries: unused
#pragma unroll
            for(int plane = 0; plane < CARRY_PLANES; ++plane) {
                if(plane < CARRIED_WORDS || (plane >= 3 && plane < 3 + CARRIED_WORDS) || plane == 6) { to.carried[plane] = carry[plane * carry_stride + beside]; }
            }
        }
    };
    Ahead ahead;
    if(CONSUMER && tiles > 0) { fetch_entries(0, ahead); }
    if(CONSUMER) { __builtin_amdgcn_s_waitcnt(0x0f70); }      // vmcnt(0): the loop is entered with nothing in flight, its first wait is the one of its back edge (pamld_near.hpp)
    for(int64_t tile = CONSUMER ? 0 : blockIdx.x; tile < tiles; tile += CONSUMER ? 1 : gridDim.x) {
        // every lane stays enabled (one code path, whole wave LDS traffic): lanes past the end
        // of the batch recompute its last read and are kept from every store and counter
        const bool active = tile * THREADS + threadIdx.x < n;
        int64_t i = active ? tile * THREADS + threadIdx.x : n - 1;
        uint32_t carried[CARRY_PLANES];
        bool is_carried = false;               // the producer left this entry's inputs beside it (StageArgs::carry)
        if(CONSUMER) {
            i = static_cast<int64_t>(ahead.read);
            is_carried = has_carry && (tile + 1) * THREADS <= static_cast<int64_t>(carry_stride);        // whole tiles, wave uniform
#pragma unroll
            for(int plane = 0; plane < CARRY_PLANES; ++plane) { carried[plane] = ahead.carried[plane]; }
            if(!apart) { __syncthreads(); }    // in place: every entry of this tile is read before any wave compacts into it
            fetch_entries(tile + 1 < tiles ? tile + 1 : tile, ahead);
        }
        ReadState r{false, 0};
        bool defer = false;
        bool tie_only = false;
        uint32_t tie_blocks = 0;
        DevResult finished_out;                // a consumer's record of this tile: the placeholder unless the read is finished
        finished_out.index = 0; finished_out.distance = 0; finished_out.hq_distance = 0; finished_out.status = STATUS_UNCLASSIFIED; finished_out.qcfail = 0; finished_out.confidence = 0;
        {
            constexpr int W = (L + 3) / 4;
            uint32_t code_words[W];
            uint32_t quality_words[W];
            if(CONSUMER && is_carried) {
                if(args.first_stage) { r.qcfail = (carried[6] & 1u) != 0; r.channel = 0; }
                else { r.qcfail = (carried[6] >> 31) != 0; r.channel = static_cast<int32_t>(carried[6] & 0x7fffffffu); }
#pragma unroll
                for(int w = 0; w < W && w < 3; ++w) { code_words[w] = carried[w]; quality_words[w] = carried[3 + w]; }
            } else {
                r = load_read_state(args, i);
                const uint32_t* code_row = reinterpret_cast<const uint32_t*>(args.base + i * (4 * W));
                const uint32_t* quality_row = reinterpret_cast<const uint32_t*>(args.quality + i * (4 * W));
#pragma unroll
                for(int w = 0; w < W; ++w) { code_words[w] = code_row[w]; quality_words[w] = quality_row[w]; }
                // a consumer's gathered tile (past the carried entries: rare) waits here, so that the wait for it does not end up
                // behind the join, where it would also wait for the carried tiles' next entries
                if(CONSUMER) { __builtin_amdgcn_s_waitcnt(0x0f70); }
            }

            // Per position: factor row, 2-bit base, high quality and "outside ACGT" flags — four positions (one plane
            // word) at a time, byte parallel.  Bit 7 of a byte is its position's flag until two flag planes at a time
            // (bits 0 and 4 of each byte) are gathered into one byte [plane 1 nibble | plane 0 nibble].
            float fm[L];
            float fe[L];
            uint32_t row[W];                               // factor rows, 8 bits per position
            uint32_t observed = 0;                         // bit-sliced 2-bit codes; anything outside ACGT arbitrary
            uint32_t high_quality = 0;                     // flag bits (low_bit of the position)
            uint32_t outside = 0;                          // flag bits: not A, C, G or T (mismatches every barcode base)
            auto gather = [](uint32_t planes) {            // bits 8 j and 8 j + 4 -> bits j and 4 + j
                const uint32_t t = planes | (planes >> 7);
                return (t | (t >> 14)) & 0xffu;
            };
#pragma unroll
            for(int w = 0; w < W; ++w) {
                constexpr uint32_t ALL = 0xffffffffu;
                const uint32_t valid = (L - 4 * w) >= 4 ? ALL : ((1u << (8 * (L - 4 * w))) - 1u);      // the bytes of positions < L
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
                // A, C, G, T (1, 2, 4, 8) -> 0..3: the low code bit is bit 1 | bit 3 (plane 0), the high one bit 2 | bit 3 (plane 1)
                const uint32_t codes = gather((((x >> 1) | (x >> 3)) & 0x01010101u) | (((x << 2) | (x << 1)) & 0x10101010u));
                observed |= __builtin_amdgcn_alignbit(codes, codes, (32 - (8 * w + 2)) & 31);           // rotl: the high nibble of byte 3 wraps
                // quality >= threshold, for thresholds of 0..128 (pamld_lut_supports): no byte carries into the next
                const uint32_t high = ((q7 + high_quality_addend) | q8) & 0x80808080u & valid;
                const uint32_t flags = gather(((~strict & 0x80808080u & valid) >> 7) | (high >> 3));
                outside |= (flags & 0xfu) << (8 * w + 2);
                high_quality |= (flags >> 4) << (8 * w + 2);
            }
#pragma unroll
            for(int j = 0; j < L; ++j) {
                const uint32_t k = (row[j >> 2] >> (8 * (j & 3))) & 0xffu;
                const float2 pair = factor_single[k];
                fm[j] = pair.x;
                fe[j] = pair.y;
            }
            if(PAIRED) { observed |= __builtin_amdgcn_alignbit(observed, observed, 16); }     // the same read against both halves
            const uint32_t observed_high = high_plane(observed);
            const uint32_t observed_low = low_plane(observed) | (PERMUTED ? wave_bytes : 0u);   // + the wave's address bits
            // tables: entry m of group g = product over the group's positions of (bit ? mismatch : match)
#pragma unroll
            for(int g = 0; g < G; ++g) {
                const int base = 4 * g;
                float lo[4];
                float hi[4];
                lo[0] = fm[base];
                lo[1] = fe[base];
                lo[2] = lo[0];
                lo[3] = lo[1];
                if(base + 1 < L) { lo[0] *= fm[base + 1]; lo[1] *= fm[base + 1]; lo[2] *= fe[base + 1]; lo[3] *= fe[base + 1]; }
                hi[0] = hi[1] = hi[2] = hi[3] = GROUP_SCALE;       // (1 up to 8 nt; round 5: 2^(96 / G) for 9 to 16 nt, see GROUP_SCALE)
                if(base + 2 < L) { hi[0] = fm[base + 2] * GROUP_SCALE; hi[1] = fe[base + 2] * GROUP_SCALE; hi[2] = hi[0]; hi[3] = hi[1]; }
                if(base + 3 < L) { hi[0] *= fm[base + 3]; hi[1] *= fm[base + 3]; hi[2] *= fe[base + 3]; hi[3] *= fe[base + 3]; }
#pragma unroll
                for(int m = 0; m < 16; ++m) {
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(table) + g * GROUP_STEP + m * ENTRY_STEP + lane_offset) = lo[m & 3] * hi[m >> 2];
                }
            }

            // conditional probabilities of the PER_WORD barcodes packed in one word
            auto probabilities_of = [&](uint32_t low_word, uint32_t high_word, float (&p)[PER_WORD]) {
                // byte k of `fields` carries index * 4 of group (k mod G) of barcode (k / G)
                const uint32_t fields = (observed_low ^ low_word) | (observed_high ^ high_word);
                const uint32_t cleaned = fields;                                         // permuted layout: byte 1 of an address
                const uint32_t up = fields << 8;                                         // plain layout: the field to byte 1
                const uint32_t down = fields >> 8;
                const uint32_t far = fields >> 16;
#pragma unroll
                for(int h = 0; h < PER_WORD; ++h) {
                    float value = 1.0f;
#pragma unroll
                    for(int g = 0; g < G; ++g) {
                        const int byte = (PAIRED ? 2 * h : 0) + g;
                        uint32_t address;
                        if(PERMUTED) {
                            // { 0, 0, cleaned.byte[byte], lane_low.byte[0] }
                            address = __builtin_amdgcn_perm(cleaned, lane_low, 0x0c0c0400u | (static_cast<uint32_t>(byte) << 8));
                        } else {
                            const uint32_t source = byte == 0 ? up : byte == 1 ? fields : byte == 2 ? down : far;
                            address = (source & INDEX_MASK) | lane_offset;
                        }
                        const float entry = ((LdsFloats)static_cast<uintptr_t>(address))[g * (GROUP_STEP / sizeof(float))];
                        value = g == 0 ? entry : value * entry;
                    }
                    p[h] = value;
                }
            };

            // Single precision scan, eight barcodes at a time.  Per block of eight: its maximum and its sum.  Across
            // blocks: the running maximum, WHICH block holds it, that block's sum set aside, and the sum of every
            // other block.  The winning block's own sum is never used — subtracting the winner from it would cancel
            // — its members are recomputed after the scan, which also names the winner inside the block.
            constexpr int FOLD = 8;
            constexpr int WORDS_PER_FOLD = FOLD / PER_WORD;
            float others_sum = 0;          // every block but the one holding the running maximum
            float held_sum = 0;            // the sum of the block holding the running maximum
            float best_value = 0;
            float runner_up = 0;           // the largest block maximum but the running one
            int best_fold = -1;
            int runner_fold = -1;          // the block that holds the runner up (tied noise reads, NEAR_TOP above) ...
            float third_value = 0;         // ... and the third largest block maximum
            auto fold_block = [&](int f, const float (&p)[FOLD]) {
                const float block_max = fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7])));
                const float block_sum = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
                const bool takes_over = block_max > best_value;                      // strict: the first block keeps a tie
                if(UNIFORM_PRIOR) {
                    // third <= runner up always holds: the third of (runner up, this block's maximum, third) is their median
                    third_value = __builtin_amdgcn_fmed3f(runner_up, block_max, third_value);
                    runner_fold = takes_over ? best_fold : (block_max > runner_up ? f : runner_fold);
                }
                // the larger of the old runner up and whichever of (old maximum, this block's maximum) does not lead
                // now: runner_up <= best_value always holds, so that is the median of the three (one v_med3_f32)
                runner_up = __builtin_amdgcn_fmed3f(best_value, block_max, runner_up);
                others_sum += takes_over ? held_sum : block_sum;
                held_sum = takes_over ? block_sum : held_sum;
                best_fold = takes_over ? f : best_fold;
                best_value = takes_over ? block_max : best_value;
            };
            auto scan_block = [&](int f, const uint32_t (&w)[2 * WORDS_PER_FOLD], bool partial) {
                // barcodes 8f .. 8f+7; the table is padded to whole blocks, the padding's probabilities are zeroed
                float p[FOLD];
#pragma unroll
                for(int k = 0; k < WORDS_PER_FOLD; ++k) {
                    float a[PER_WORD];
                    probabilities_of(w[k], w[WORDS_PER_FOLD + k], a);
#pragma unroll
                    for(int h = 0; h < PER_WORD; ++h) { p[PER_WORD * k + h] = a[h]; }
                }
                if(!UNIFORM_PRIOR) {
#pragma unroll
                    for(int k = 0; k < FOLD; ++k) { p[k] *= prior[FOLD * f + k]; }     // padded with zeros
                }
                if(partial) {
#pragma unroll
                    for(int k = 1; k < FOLD; ++k) { if(FOLD * f + k >= n_barcode) { p[k] = 0.0f; } }
                }
                fold_block(f, p);
            };
            const int n_full = n_barcode / FOLD;
            {
                // the block's words from LDS at a wave uniform address (a broadcast: one ds_read_b128 brings four
                // words to every lane; the other waves of the SIMD cover its latency)
                const uint4* const blocks = reinterpret_cast<const uint4*>(lds_words);
                constexpr int QUADS = 2 * WORDS_PER_FOLD / 4;
                auto words_of = [&](int f, uint32_t (&w)[2 * WORDS_PER_FOLD]) {
#pragma unroll
                    for(int q = 0; q < QUADS; ++q) {
                        const uint4 quad = blocks[QUADS * f + q];
                        w[4 * q] = quad.x; w[4 * q + 1] = quad.y; w[4 * q + 2] = quad.z; w[4 * q + 3] = quad.w;
                    }
                };
#pragma unroll 2
                for(int f = 0; f < n_full; ++f) {
                    uint32_t w[2 * WORDS_PER_FOLD];
                    words_of(f, w);
                    scan_block(f, w, false);
                }
                if(n_barcode % FOLD) {                                                   // the last, partial block
                    uint32_t w[2 * WORDS_PER_FOLD];
                    words_of(n_full, w);
                    scan_block(n_full, w, true);
                }
            }

            // Inside the winning block: the winner (first of equals) and the sum of the other seven.  The scan left the
            // block's maximum (best_value) and its sum (held_sum); the same instructions give the same eight values
            // again, so the winner is the first member EQUAL to best_value, the rest is the sum of the members that are
            // not (no subtraction from the block sum, which would cancel).  Two members equal to the maximum: the block's
            // sum less that rest is then twice the maximum, not once (the rest is at most seven times the maximum, so
            // single precision tells 1 from 2 easily); such a read is handed on by raising the runner up to the winner.
            int best = -1;
            uint32_t best_flags = 0;
            const float runner_block = runner_up;          // the largest maximum of another block (the runner up below counts the winner's block in)
            float sigma = others_sum;
            const float winner_single = best_value;
            {
                const int block = best_fold < 0 ? 0 : best_fold;
                float p[FOLD];
#pragma unroll
                for(int k = 0; k < WORDS_PER_FOLD; ++k) {
                    float a[PER_WORD];
                    probabilities_of(lds_words[2 * WORDS_PER_FOLD * block + k], lds_words[2 * WORDS_PER_FOLD * block + WORDS_PER_FOLD + k], a);
#pragma unroll
                    for(int h = 0; h < PER_WORD; ++h) {
                        const int b = FOLD * block + PER_WORD * k + h;
                        float value = a[h];
                        if(!UNIFORM_PRIOR) { value *= d.concentration_single[b]; }
                        p[PER_WORD * k + h] = value;
                    }
                }
                if(n_barcode % FOLD) {                           // the padding of a partial last block, as the scan zeroes it
                    const int limit = n_barcode - FOLD * block;
#pragma unroll
                    for(int k = 1; k < FOLD; ++k) { p[k] = k < limit ? p[k] : 0.0f; }
                }
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
                const bool crowded = held_sum - rest >= 1.5f * best_value;         // the block's sum holds the maximum twice
                runner_up = fmaxf(runner_up, crowded ? best_value : second);
                sigma += crowded ? best_value : rest;
                if(best_fold >= 0 && best_value > 0.0f && within >= 0) {
                    best = FOLD * block + within;
                    const int at = 2 * WORDS_PER_FOLD * block + within / PER_WORD;
                    const uint32_t fields = (observed_low ^ lds_words[at]) | (observed_high ^ lds_words[at + WORDS_PER_FOLD]);
                    best_flags = ((PAIRED ? (fields >> (16 * (best & 1))) : fields) | outside) & full_mask;
                }
            }

            if(best < 0) {
                defer = true;
            } else {
                // the winner's distances and its probability again in fp64 from the fp64 factor rows
                int distance = __popc(best_flags);
                int hq_distance = __popc(best_flags & high_quality);
                double best_conditional = 1.0;
#pragma unroll
                for(int j = 0; j < L; ++j) {
                    const uint32_t k = (row[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    best_conditional *= factor_double[k][(best_flags >> low_bit(j)) & 1u];
                }
                const double prior_of_best = UNIFORM_PRIOR ? common_prior : d.concentration[best];
                const double best_p = best_conditional * prior_of_best;
                // everything but the winner, in units of p: only ever compared against and added to much larger terms
                const double others = (UNIFORM_PRIOR ? static_cast<double>(sigma) * prior_of_best : static_cast<double>(sigma)) * PRODUCT_UNSCALE;
                const double total = (best_p + others) + d.adjusted_noise_probability;   // pamld.cpp:61-86
                double confidence = best_p / total;                                       // pamld.cpp:92
                const double noise_margin = best_conditional - d.random_barcode_probability;
                const double confidence_margin = confidence - d.confidence_threshold;
                // certificate: a dominant winner and no decision within 1e-9 (relative) of its threshold
                // a read that clearly fails the noise filter only needs its argmax (it is charged to it and leaves
                // unclassified, pamld.cpp:96,113-119): the runner up further than single precision can blur;
                // otherwise the posterior matters: a dominant winner, no decision within 1e-9 of its threshold
                const bool noise_clear = fabs(noise_margin) > 1e-9 * d.random_barcode_probability;
                defer = !noise_clear
                     || (noise_margin <= 0 ? !(winner_single >= SMALLEST_CERTAIN && runner_up <= RUNNER_UP * winner_single)
                                           : (!(others <= dominance(L) * best_p) || !(fabs(confidence_margin) > 2e-8)));
                // handed on for a tied argmax alone (NEAR_TOP, above): every candidate a thousandth under the noise filter, in two blocks
                if(UNIFORM_PRIOR) {
                    tie_only = args.flag_ties != 0 && defer && noise_clear && noise_margin <= 0 && winner_single >= SMALLEST_CERTAIN &&
                               third_value < NEAR_TOP * winner_single && best_conditional * 1.001 < d.random_barcode_probability;
                    // bits 0..12 the winner's block, 13..25 the runner up's, 26: that one comes within NEAR_TOP (else the winner's block alone)
                    tie_blocks = static_cast<uint32_t>(best_fold < 0 ? 0 : best_fold) | (static_cast<uint32_t>(runner_fold < 0 ? 0 : runner_fold) << 13) |
                                 ((runner_fold >= 0 && runner_block >= NEAR_TOP * winner_single) ? 1u << 26 : 0u);
                    if(CONSUMER && tie_only) { finished_out.confidence = __hiloint2double(0x7ff80000, static_cast<int>(tie_blocks)); }
                }
                // a consumer's lanes past the end of the list recompute its last read and store its record again (the counters
                // are kept from them): every lane has a record to store
                if(!defer && (active || CONSUMER)) {
                    int bin = best + 1;
                    uint8_t status = STATUS_PASS;
                    if(noise_margin > 0) {                                                // pamld.cpp:96
                        if(confidence_margin > 0) {                                       // pamld.cpp:99
                            if(d.high_quality_distance_threshold > 0 && hq_distance >= d.high_quality_distance_threshold) {
                                r.qcfail = true;
                            }
                            if(active) { stat_add_confidence(d, sink, bin, r.qcfail ? 0 : 1, confidence); }
                        } else {
                            if(active) { stat_add(d, sink, bin, 0, FIELD_LOW_CONFIDENCE, 1ull); }
                            r.qcfail = true;
                            status = STATUS_LOW_CONFIDENCE;
                        }
                    } else {
                        if(active) { stat_add(d, sink, bin, 0, FIELD_LOW_CONDITIONAL_CONFIDENCE, 1ull); }
                        r.qcfail = true;
                        bin = 0;
                        distance = 0;
                        hq_distance = 0;
                        confidence = 0;
                        status = STATUS_NOISE;
                    }
                    const int32_t index_of_bin = bin == 0 ? 0 : identity_index ? bin : d.barcode_index[bin - 1];
                    if(active && !(sink.defer && bin > 0)) {                              // decoder_tail (kernel_common.hpp)
                        const int pf = r.qcfail ? 0 : 1;
                        if(bin > 0 && distance) { stat_add(d, sink, bin, pf, FIELD_DISTANCE, static_cast<unsigned long long>(distance)); }
                        stat_add(d, sink, bin, pf, FIELD_COUNT, 1ull);
                    }
                    if(d.multiplexing) { r.channel = index_of_bin; }
                    DevResult out;
                    out.index = index_of_bin;
                    out.distance = static_cast<uint8_t>(distance > 255 ? 255 : distance);
                    out.hq_distance = static_cast<uint8_t>(hq_distance > 255 ? 255 : hq_distance);
                    out.status = status;
                    out.qcfail = r.qcfail ? 1 : 0;
                    out.confidence = confidence;
                    if(CONSUMER) { finished_out = out; }
                    else {
                        if(args.result) { stream_result(&args.result[i], out); }
                        stream_read_state(args, i, r);
                    }
                } else if(active && !CONSUMER) {
                    // the next tier overwrites it (a tied noise read's names its two blocks in the bits of its confidence)
                    DevResult out;
                    out.index = 0; out.distance = 0; out.hq_distance = 0; out.status = STATUS_UNCLASSIFIED; out.qcfail = r.qcfail ? 1 : 0;
                    out.confidence = tie_only ? __hiloint2double(0x7ff80000, static_cast<int>(tie_blocks)) : 0.0;
                    if(args.result) { stream_result(&args.result[i], out); }
                    stream_read_state(args, i, r);
                }
            }
        }
        if(CONSUMER) {
            // every lane stores a record and a status word, in straight line code: the placeholder again for a read deferred
            // once more, the last read's again for a lane past the end.  The compiler can count these two stores, so the wait
            // for the next tile's entries (asked for before them) does not wait for them.
            finished_out.qcfail = defer ? (r.qcfail ? 1 : 0) : finished_out.qcfail;
            stream_result(&args.result[i], finished_out);
            __builtin_nontemporal_store((static_cast<uint32_t>(r.channel) & 0x7fffffffu) | (r.qcfail ? 0x80000000u : 0u), &args.word[i]);
        }
        // deferred reads: one LDS add per wave reserves a run of slots in this workgroup's worklist segment
        const unsigned long long deferred = __ballot(active && defer);
        if(deferred) {
            const int lane = threadIdx.x & 63;
            uint32_t first = 0;
            if(lane == 0) { first = atomicAdd(&deferred_total, static_cast<uint32_t>(__popcll(deferred))); }
            first = __shfl(first, 0);
            if(active && defer) {
                segment[gathered_at + first + __popcll(deferred & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(i) | (tie_only ? 0x80000000u : 0u);
            }
        }
        count_outgoing(args, histogram, active && !defer, r);
    }
    __syncthreads();
    if(apart) {
        // the list was gathered behind the entries: down to the front (it is no longer than the entries were: no overlap)
        const uint32_t left = deferred_total;
        for(uint32_t k = threadIdx.x; k < left; k += THREADS) { segment[k] = segment[gathered_at + k]; }
    }
    if(threadIdx.x == 0) { args.worklist_count[own] = deferred_total; }
    }
    histogram_flush(args, d, histogram);
}

template <int L, bool UNIFORM_PRIOR, bool CONSUMER>
int launch_one(const StageArgs& args, int n_barcode, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    const size_t lds_bytes = LutGeometry<L>::histogram_at(n_barcode) + histogram_bytes;
    static std::atomic<uint64_t> prepared{0};
    if(const int status = allow_large_lds(&pamld_lut_kernel<L, UNIFORM_PRIOR, CONSUMER>, prepared)) { return status; }
    hipLaunchKernelGGL((pamld_lut_kernel<L, UNIFORM_PRIOR, CONSUMER>), grid, dim3(LutGeometry<L>::THREADS), lds_bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}

// instantiate every nucleotide cardinality of [LO, HI] and dispatch on the decoder
template <int LO, int HI>
int launch_range(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if constexpr(LO > HI) {
        (void)args; (void)host_decoder; (void)grid; (void)histogram_bytes; (void)stream;
        return static_cast<int>(hipErrorInvalidValue);
    } else {
        if(host_decoder.nucleotide_cardinality == LO) {
            // behind the neighbour list kernels (pamld_near.hpp, pamld_pair.hpp): as a consumer of their lists; priors that
            // differ from 5 nt up (what those two kernels take)
            if(args.worklist_role == 2) {
                if(host_decoder.uniform_prior) { return launch_one<LO, true, true>(args, host_decoder.n_barcode, grid, histogram_bytes, stream); }
                if constexpr(LO >= NEAR_MIN_LENGTH) { return launch_one<LO, false, true>(args, host_decoder.n_barcode, grid, histogram_bytes, stream); }
                else { return static_cast<int>(hipErrorInvalidValue); }
            }
            return host_decoder.uniform_prior ? launch_one<LO, true, false>(args, host_decoder.n_barcode, grid, histogram_bytes, stream)
                                              : launch_one<LO, false, false>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        }
        return launch_range<LO + 1, HI>(args, host_decoder, grid, histogram_bytes, stream);
    }
}


}  // namespace lut
}  // namespace phq
