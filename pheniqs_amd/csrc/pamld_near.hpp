// K0 (neighbour lists): PamlDecoder<T>::classify for the reads a short list of barcodes settles.
//
//   reference pamld.cpp:37-123   posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164  P(r|b) = pow(base, sum of Phred costs), distance, high quality distance
//
// The product form kernel (pamld_lut.hpp) scores every barcode for every read.  Almost all of that work adds
// nothing: a read sits on or next to ONE barcode c, and a barcode b' that differs from c in d positions mismatches a
// read at distance k from c in at least d - k positions, each of which costs it a factor rho = P(mismatch) / P(match)
// of that position's quality.  So per read this kernel
//   * names c with one lookup: the read's 2-bit codes are a key into a host built table of the nearest barcode
//     (Hamming distance, first on ties) of every possible word — 4^L entries of 16 bits, in global memory, the few
//     hundred words real reads carry stay in L1/L2;
//   * scores c and the LIST - 1 = 15 barcodes nearest to c with the per read product tables of pamld_lut.hpp (two
//     16 entry tables of single precision products of 4 factors in LDS); the list of c — the bit sliced words of those
//     16 barcodes, four barcodes per pair of plane words — is read per lane from LDS (48 bytes: three ds_read_b128);
//   * BOUNDS every barcode it did not score: with r = the distance from c of the nearest one not listed (host: 4 or
//     5 for sets like Illumina's), such a barcode mismatches the read in n >= r - k positions outside the k positions
//     where the read differs from c, so its probability is at most the largest product the tables hold over n (or
//     more) mismatches.  That maximum is formed from the tables' own entries: per group of 4 positions the largest
//     entry by number of mismatches (G[0..4], made monotone so that qualities below 4, where a mismatch is MORE likely
//     than a match, stay covered), then H[n] = max over a + b = n of G1[a] * G2[b].  The barcodes at distance exactly
//     r (cnt_r of them, host) are charged H[r - k], the others H[r + 1 - k]; the k = 1 case leaves the read's own
//     mismatch position out through  top_n(without j) <= min(top_n, top_{n+1} / rho_j).
// A read is FINISHED here when all of this is certain to 1e-9 in the posterior: it passes the noise filter clearly,
// everything else sums to at most 2e-3 of the winner (so single precision, relative error ~2e-7, moves the
// posterior by less than 5e-10: the posterior is formed as 1 - x / (1 + x), x = (others + noise) / winner, and only
// that last subtraction is in double precision), the bound on what was not scored moves it by at most 5e-10, and
// the confidence is further than 2e-9 from its threshold.  About 90 % of the reads of the 96 x 8 nt benchmark.
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

constexpr int LIST = NEAR_LIST;                // barcodes scored per read: c and the 15 nearest to it
constexpr int RECORD_WORDS = NEAR_RECORD_WORDS;    // per barcode: 4 x {low plane word, high plane word}, radius, cnt_r, cnt_rest, pad
static_assert(LIST == 16 && RECORD_WORDS == 12, "the kernel reads a record as three quads");
constexpr int THREADS = 512;
constexpr int FACTOR_ROWS = 129;
constexpr float LISTED_MAX = 5e-4f;            // scored barcodes other than c sum to at most this share of c's probability
constexpr float X_MAX = 2e-3f;                 // (others + noise) / winner: single precision is enough below this
constexpr float REST_MAX = 1e-9f;              // the bound on what was not scored, as a share of the winner
constexpr float SLOP = 1.0f + 1e-5f;           // single precision rounding of the bound itself

struct Layout {
    static constexpr uint32_t TABLE_AT = 0;                                    // [wave][group][entry][lane] floats: 64 KiB
    static constexpr uint32_t TABLE_BYTES = (THREADS / 64) * 2 * 16 * 64 * 4;
    static constexpr uint32_t FACTOR_AT = TABLE_AT + TABLE_BYTES;              // FACTOR_ROWS x {match, mismatch} floats
    static constexpr uint32_t DEFERRED_AT = FACTOR_AT + 1040;
    static constexpr uint32_t RECORD_AT = DEFERRED_AT + 16;                    // n_barcode x RECORD_WORDS words
    __host__ __device__ static constexpr uint32_t histogram_at(int n_barcode) {
        return RECORD_AT + static_cast<uint32_t>(n_barcode) * RECORD_WORDS * 4u;
    }
};

typedef __attribute__((address_space(3))) const float* LdsFloats;
typedef uint32_t Quad __attribute__((ext_vector_type(4)));
typedef float Pair __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const Quad* LdsQuads;
typedef __attribute__((address_space(3))) const Pair* LdsPairs;

// (a & mask) | c as v_bitop3_b32 (full rate; hipcc's own choice, v_and_or_b32, issues at half rate): truth table 0xEA
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, mask, c, 0xEA);
}

// STATE: where a read's carried state comes from — 0 the first stage of a job without input flags (nothing to load),
// 1 the first stage with flag bytes, 2 a later stage (the status word).  A template parameter so that the loads of
// the prefetch sit in straight line code: behind a branch the compiler can no longer count them and waits for all.
template <int L, int STATE>
__global__ __launch_bounds__(THREADS) void pamld_near_kernel(StageArgs args) {
    static_assert(L >= 5 && L <= 8, "two groups of 4 positions");
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
    const uint32_t lane_const = ((threadIdx.x & 63u) << 2) | (wave << 13);          // everything of a table address but group and entry
    const int high_quality_threshold = min(max(d.high_quality_threshold, 0), 128);
    const uint32_t high_quality_addend = static_cast<uint32_t>(128 - high_quality_threshold) * 0x01010101u;
    const float noise_over_prior = static_cast<float>(d.adjusted_noise_probability / d.concentration[0]);       // uniform priors
    const float noise_floor = static_cast<float>(d.random_barcode_probability) * SLOP;
    const double confidence_threshold = d.confidence_threshold;
    constexpr uint32_t VALID1 = L >= 8 ? 0xffffffffu : ((1u << (8 * (L - 4))) - 1u);     // the bytes of positions 4 .. L - 1

    // A tile's inputs are fetched one tile ahead — behind this tile's table lookup, so that waiting for the lookup does
    // not wait for them (vector memory operations complete in issue order) — through a scalar tile base and a 32-bit
    // lane offset.  Lanes past the end of the batch recompute its last read and are kept from every store and counter.
    struct Inputs { uint32_t code0, code1, quality0, quality1, state; };
    auto fetch = [&](int64_t tile, Inputs& in) {
        const int64_t first = tile * THREADS;
        const uint32_t last = static_cast<uint32_t>(min(static_cast<int64_t>(THREADS), n - first)) - 1u;
        const uint32_t at = min(threadIdx.x, last);
        const uint2 code = *reinterpret_cast<const uint2*>(args.base + first * 8 + at * 8u);
        const uint2 quality = *reinterpret_cast<const uint2*>(args.quality + first * 8 + at * 8u);
        in.code0 = code.x; in.code1 = code.y; in.quality0 = quality.x; in.quality1 = quality.y;
        if(STATE == 1) { in.state = static_cast<uint32_t>(args.input_flag[first + at]); }      // transcode.h:207-211
        else if(STATE == 2) { in.state = args.word[first + at]; }
        else { in.state = 0; }
    };
    Inputs next;
    fetch(min(static_cast<int64_t>(blockIdx.x), tiles - 1), next);
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): the loop is entered with nothing in flight, so that its first wait is the one of its back edge
    uint32_t pending_slot = 0xffffffffu;       // this lane's worklist entry of the previous tile, not stored yet
    uint32_t pending_read = 0;
    uint32_t pending_input[CARRY_PLANES] = {0, 0, 0, 0, 0};
    const uint32_t carry_stride = args.carry ? static_cast<uint32_t>(args.carry_stride) : 0u;
    uint32_t* const carry = args.carry + static_cast<int64_t>(blockIdx.x) * CARRY_PLANES * carry_stride;
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
        const uint32_t code0 = in.code0, code1 = in.code1, quality0 = in.quality0, quality1 = in.quality1;
        // BAM codes, four positions per word.  A, C, G or T = exactly one bit of the nibble (iupac.h:51-56); a read
        // with anything else goes to the next tier.
        const uint32_t x0 = code0 & 0x0f0f0f0fu;
        const uint32_t x1 = code1 & (0x0f0f0f0fu & VALID1);
        const uint32_t z0 = x0 & ((x0 | 0x80808080u) - 0x01010101u);
        const uint32_t z1 = x1 & ((x1 | 0x80808080u) - 0x01010101u);
        const uint32_t strict0 = (x0 + 0x7f7f7f7fu) & ~(z0 + 0x7f7f7f7fu) & 0x80808080u;
        const uint32_t strict1 = (x1 + 0x7f7f7f7fu) & ~(z1 + 0x7f7f7f7fu) & 0x80808080u;
        const bool all_strict = ((strict0 ^ 0x80808080u) | (strict1 ^ (0x80808080u & VALID1))) == 0;
        // 2-bit codes A, C, G, T -> 0..3: the low code bit is bit 1 | bit 3 of the nibble, the high one bit 2 | bit 3.
        // Bits 0 / 4 of the bytes of word 0 / word 1 gathered into one byte: bit j = position j.
        auto gather = [](uint32_t planes) {            // bits 8 j and 8 j + 4 -> bits j and 4 + j
            const uint32_t t = planes | (planes >> 7);
            return (t | (t >> 14)) & 0xffu;
        };
        const uint32_t low8 = gather((((x0 >> 1) | (x0 >> 3)) & 0x01010101u) | (((x1 << 3) | (x1 << 1)) & 0x10101010u));
        const uint32_t high8 = gather((((x0 >> 2) | (x0 >> 3)) & 0x01010101u) | (((x1 << 2) | (x1 << 1)) & 0x10101010u));
        const uint32_t key = low8 | (high8 << 8);
        if(pending_slot != 0xffffffffu) {                                         // the previous tile's deferred reads
            segment[pending_slot] = pending_read;
            if(pending_slot < carry_stride) {
#pragma unroll
                for(int plane = 0; plane < CARRY_PLANES; ++plane) { carry[plane * carry_stride + pending_slot] = pending_input[plane]; }
            }
        }
        const uint32_t c = nearest[key];                                          // the barcode nearest to this word
        fetch(min(tile + gridDim.x, tiles - 1), next);                           // past the last tile: that tile again, unused
        const uint32_t observed_low = low8 * 0x01010101u;                        // the read against four barcodes at a time
        const uint32_t observed_high = high8 * 0x01010101u;
        const uint32_t q0 = quality0 & 0x7f7f7f7fu;                              // the table has 128 rows
        const uint32_t q1 = quality1 & (0x7f7f7f7fu & VALID1);
        // quality >= high quality threshold (thresholds of 0..128), bit 7 of each byte -> bit j = position j
        const uint32_t high_quality = gather(((((q0 + high_quality_addend) | quality0) & 0x80808080u) >> 7) |
                                             ((((q1 + high_quality_addend) | quality1) & (0x80808080u & VALID1)) >> 3));

        // per position {P(match), P(mismatch)} of its quality, single precision
        float fm[8];
        float fe[8];
        {
            const uint32_t factor_base = Layout::FACTOR_AT;
#pragma unroll
            for(int j = 0; j < L; ++j) {
                const uint32_t q = j < 4 ? q0 : q1;
                const int byte = j & 3;
                const uint32_t shifted = byte == 0 ? (q << 3) : (q >> (8 * byte - 3));
                const uint32_t address = and_or(shifted, 0x3f8u, factor_base);
                const Pair pair = *(LdsPairs)static_cast<uintptr_t>(address);
                fm[j] = pair.x;
                fe[j] = pair.y;
            }
#pragma unroll
            for(int j = L; j < 8; ++j) { fm[j] = 1.0f; fe[j] = 1.0f; }
        }
        // tables: entry m of group g = product over the group's positions of (bit ? mismatch : match); and per group the
        // largest entry by number of mismatches, made monotone: ge[a] = the largest entry with a OR MORE mismatches
        float ge[2][5];
        float all_match[2];
#pragma unroll
        for(int g = 0; g < 2; ++g) {
            const int base = 4 * g;
            float lo[4];
            float hi[4];
            lo[0] = fm[base] * fm[base + 1]; lo[1] = fe[base] * fm[base + 1]; lo[2] = fm[base] * fe[base + 1]; lo[3] = fe[base] * fe[base + 1];
            hi[0] = fm[base + 2] * fm[base + 3]; hi[1] = fe[base + 2] * fm[base + 3]; hi[2] = fm[base + 2] * fe[base + 3]; hi[3] = fe[base + 2] * fe[base + 3];
            float t[16];
#pragma unroll
            for(int m = 0; m < 16; ++m) {
                t[m] = lo[m & 3] * hi[m >> 2];
                *reinterpret_cast<float*>(lds_raw + (lane_const | (static_cast<uint32_t>(g) << 12) | (static_cast<uint32_t>(m) << 8))) = t[m];
            }
            all_match[g] = t[0];
            const float g1 = fmaxf(__builtin_fmaxf(__builtin_fmaxf(t[1], t[2]), t[4]), t[8]);
            const float g2 = fmaxf(fmaxf(fmaxf(t[3], t[5]), fmaxf(t[6], t[9])), fmaxf(t[10], t[12]));
            const float g3 = fmaxf(fmaxf(t[7], t[11]), fmaxf(t[13], t[14]));
            ge[g][4] = t[15];
            ge[g][3] = fmaxf(g3, ge[g][4]);
            ge[g][2] = fmaxf(g2, ge[g][3]);
            ge[g][1] = fmaxf(g1, ge[g][2]);
            ge[g][0] = fmaxf(t[0], ge[g][1]);
        }
        // H[n]: the largest product over n or more mismatching positions, n = 3 .. 6
        const float h3 = fmaxf(fmaxf(ge[0][0] * ge[1][3], ge[0][1] * ge[1][2]), fmaxf(ge[0][2] * ge[1][1], ge[0][3] * ge[1][0]));
        const float h4 = fmaxf(fmaxf(fmaxf(ge[0][0] * ge[1][4], ge[0][1] * ge[1][3]), fmaxf(ge[0][2] * ge[1][2], ge[0][3] * ge[1][1])), ge[0][4] * ge[1][0]);
        const float h5 = fmaxf(fmaxf(ge[0][1] * ge[1][4], ge[0][2] * ge[1][3]), fmaxf(ge[0][3] * ge[1][2], ge[0][4] * ge[1][1]));
        const float h6 = fmaxf(fmaxf(ge[0][2] * ge[1][4], ge[0][3] * ge[1][3]), ge[0][4] * ge[1][2]);

        // the list of c: 16 barcodes as 4 x {low plane word, high plane word}, then {radius, cnt_r, cnt_rest}
        const uint32_t record_at = Layout::RECORD_AT + c * (RECORD_WORDS * 4u);
        const Quad list0 = *(LdsQuads)static_cast<uintptr_t>(record_at);
        const Quad list1 = *(LdsQuads)static_cast<uintptr_t>(record_at + 16u);
        const Quad meta = *(LdsQuads)static_cast<uintptr_t>(record_at + 32u);
        const uint32_t low_words[4] = {list0.x, list0.z, list1.x, list1.z};
        const uint32_t high_words[4] = {list0.y, list0.w, list1.y, list1.w};
        float winner = 0;                        // P(r|c), single precision
        float others = 0;                        // the other 15
        uint32_t mismatches = 0;                 // of c: bit j = position j
#pragma unroll
        for(int g4 = 0; g4 < 4; ++g4) {
            // byte b of fields = the 8 mismatch flags of barcode 4 g4 + b: the low nibble indexes table 0, the high one table 1
            const uint32_t fields = (observed_low ^ low_words[g4]) | (observed_high ^ high_words[g4]);
            if(g4 == 0) { mismatches = fields & 0xffu; }
            float p[4];
#pragma unroll
            for(int b = 0; b < 4; ++b) {
                const int low_at = 8 * b;            // bit of the low nibble; the entry index belongs at bits 8..11
                const uint32_t a0 = and_or(low_at < 8 ? (fields << (8 - low_at)) : (fields >> (low_at - 8)), 0xf00u, lane_const);
                const int high_at = 8 * b + 4;
                const uint32_t a1 = and_or(high_at < 8 ? (fields << (8 - high_at)) : (fields >> (high_at - 8)), 0xf00u, lane_const);
                const float e0 = *(LdsFloats)static_cast<uintptr_t>(a0);
                const float e1 = ((LdsFloats)static_cast<uintptr_t>(a1))[4096 / 4];
                p[b] = e0 * e1;
            }
            if(g4 == 0) {
                winner = p[0];
                others = (p[1] + p[2]) + p[3];
            } else {
                others += (p[0] + p[1]) + (p[2] + p[3]);
            }
        }

        // what was not scored: cnt_r barcodes at distance r from c, cnt_rest further away
        const uint32_t radius = meta.x;
        const float cnt_r = __uint_as_float(meta.y);
        const float cnt_rest = __uint_as_float(meta.z);
        const int k = __popc(mismatches);
        const bool near5 = radius >= 5u;                                    // the host clamps the radius to 5; 4 otherwise (less: not listed here)
        const float h_r = near5 ? h5 : h4;                                  // H[r]
        const float h_up = near5 ? h6 : h5;                                 // H[r + 1]
        const float h_down = near5 ? h4 : h3;                               // H[r - 1]
        const float inverse_winner = __builtin_amdgcn_rcpf(winner);
        const float everything_matches = all_match[0] * all_match[1];
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
        const float x = x_low + 0.5f * x_rest;
        const float y = x * __builtin_amdgcn_rcpf(1.0f + x);              // 1 - posterior
        double confidence = 1.0 - static_cast<double>(y);                  // pamld.cpp:92
        const double confidence_margin = confidence - confidence_threshold;
        const bool finished = all_strict && k <= 1 && radius >= 4u && winner > noise_floor &&
                              x_listed <= LISTED_MAX && x_low + x_rest <= X_MAX && x_rest <= REST_MAX &&
                              fabs(confidence_margin) > 2e-9;
        const bool defer = !finished;

        // Outcome.  Every lane stores a record and a status word in straight line code — the real ones when the read is
        // finished, the placeholder the next tier overwrites when it is not; a lane past the end of the batch stores its
        // (last) read's again — so that the compiler can count these two stores at the wait for the prefetched inputs.
        const uint32_t index = meta.w;                                              // barcode_index[c]
        const uint32_t hq_distance = static_cast<uint32_t>(__popc(mismatches & high_quality));
        const bool confident = confidence_margin > 0;                               // pamld.cpp:99
        const bool qcfail = r.qcfail || !confident || (hq_threshold > 0 && static_cast<int>(hq_distance) >= hq_threshold);
        const bool qcfail_out = finished ? qcfail : r.qcfail;
        {
            typedef uint32_t vector4 __attribute__((ext_vector_type(4)));
            const uint32_t detail = static_cast<uint32_t>(k) | (hq_distance << 8) | ((confident ? STATUS_PASS : STATUS_LOW_CONFIDENCE) << 16);
            vector4 out;
            out.x = finished ? index : 0u;
            out.y = (finished ? detail : (static_cast<uint32_t>(STATUS_UNCLASSIFIED) << 16)) | (qcfail_out ? 1u << 24 : 0u);
            out.z = finished ? static_cast<uint32_t>(__double2loint(confidence)) : 0u;
            out.w = finished ? static_cast<uint32_t>(__double2hiint(confidence)) : 0u;
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
                atomicAdd(&low_words[2 * (STAT_JOB_WORDS + (c + 1u) * STAT_WORDS_PER_BIN + FIELD_LOW_CONFIDENCE)], 1u);      // qc-fail half (pamld.cpp:104-106)
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
                pending_input[0] = code0; pending_input[1] = code1; pending_input[2] = quality0; pending_input[3] = quality1; pending_input[4] = in.state;
            }
        }
        count_outgoing(args, histogram, active && !defer, r);
    }
    if(pending_slot != 0xffffffffu) {
        segment[pending_slot] = pending_read;
        if(pending_slot < carry_stride) {
#pragma unroll
            for(int plane = 0; plane < CARRY_PLANES; ++plane) { carry[plane * carry_stride + pending_slot] = pending_input[plane]; }
        }
    }
    __syncthreads();
    if(threadIdx.x == 0) { args.worklist_count[blockIdx.x] = deferred_total; }
    histogram_flush(args, d, histogram);
}

template <int L, int STATE>
int launch_state(const StageArgs& args, int n_barcode, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    const size_t lds_bytes = Layout::histogram_at(n_barcode) + histogram_bytes;
    static std::atomic<uint64_t> prepared{0};
    if(const int status = allow_large_lds(&pamld_near_kernel<L, STATE>, prepared)) { return status; }
    hipLaunchKernelGGL((pamld_near_kernel<L, STATE>), grid, dim3(THREADS), lds_bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}

template <int L>
int launch_one(const StageArgs& args, int n_barcode, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(!args.first_stage) { return launch_state<L, 2>(args, n_barcode, grid, histogram_bytes, stream); }
    return args.input_flag ? launch_state<L, 1>(args, n_barcode, grid, histogram_bytes, stream)
                           : launch_state<L, 0>(args, n_barcode, grid, histogram_bytes, stream);
}

}  // namespace near
}  // namespace phq
