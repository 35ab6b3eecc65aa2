// K1g: PamlDecoder<T>::classify for codecs of a FEW barcodes (at most 8 ACGT barcodes of up to 8 nt), one persistent
// kernel with a pipeline per wave — the codecs pamld_tiny.hip serves, its arithmetic, laid out for this part.
//
//   reference pamld.cpp:37-123   posterior over all barcodes, noise / confidence / high quality filters
//   reference barcode.h:131-164  P(r|b) = pow(base, sum of Phred costs), distance, high quality distance
//
// A SPLiT-seq round of 4 barcodes (C5's fourth decoder), a multiplex of a handful of samples: no lookup, no lists, no
// per read tables — every barcode is scored for every read, in fp64, as pamld_tiny.hip does.  What differs:
//   * the products are taken in units of the read's all-match product A: a barcode's probability over A is the product
//     of rho_j = P(mismatch | q_j) / P(match | q_j) over its mismatching positions, the barcodes' bit-sliced words in
//     scalar registers, a position's factor picked by a MASK (one v_bfe_i32, two v_bitop3_b32) instead of a compare and two
//     selects per position and barcode; the largest term and the sum of the others are kept apart;
//   * the posterior is top / total by a reciprocal refined twice (v_rcp_f64 + two Newton steps: within ~2 ulp of the
//     quotient) instead of the full division sequence;
//   * every wave is its own pipeline over tiles of 64 reads (inputs fetched a tile ahead), whole line stores, and the
//     counters go to LDS bins kept in 16 copies (lane & 15 picks one), 18 banks apart — with 4 bins for 64 lanes the
//     adds of a wave would otherwise queue on the same words.
// The certificate is pamld_tiny.hip's: the products are within ~1e-15 of the reference's pow of a Kahan sum, so a read is
// finished here when nothing hangs on less than that — everything but the winner at most half of it (no near tie for
// the argmax), the noise filter and the confidence filter both further than 1e-9 from their thresholds.  Everything else
// (~0.3 % of C5's fourth round) goes to the wave's segment of the HBM worklist for the exact tier and the literal kernel.
#include "kernel_common.hpp"

namespace phq {
namespace few {

constexpr int WAVES = 8;
constexpr int THREADS = 64 * WAVES;
constexpr int MAX_BARCODES = 8;
constexpr int COPIES = 16;
constexpr int BIN_WORDS = 8;                   // words 0..3 the qc fail half, 4..7 the pass filter half: {count | distance << 32, confidence low limb, confidence high limb, low conditional confidence count | low confidence count << 32}
constexpr uint32_t COPY_BYTES = 72;            // 8 words and 8 bytes of padding: the copies of a bin start 18 banks apart
constexpr uint32_t BIN_BYTES = COPIES * COPY_BYTES;
constexpr int FACTOR_ROWS = 129;
// the LDS map (dynamic block at address 0)
constexpr uint32_t FACTOR_AT = 0;              // FACTOR_ROWS x {P(match), rho = P(mismatch) / P(match)} doubles: a row's address is its number << 4
constexpr uint32_t BINS_AT = FACTOR_AT + FACTOR_ROWS * 16;            // bin 0 .. MAX_BARCODES, then the two job words
constexpr uint32_t JOB_AT = BINS_AT + (MAX_BARCODES + 1) * BIN_BYTES;
constexpr uint32_t WORDS_AT = JOB_AT + 8 * STAT_JOB_WORDS;            // the barcodes' bit-sliced words, then their indexes by bin
constexpr uint32_t INDEX_AT = WORDS_AT + 4 * MAX_BARCODES;
constexpr uint32_t LDS_BYTES = INDEX_AT + 4 * (MAX_BARCODES + 1);
static_assert(BINS_AT % 8 == 0 && JOB_AT % 8 == 0, "alignment");

typedef uint32_t Quad __attribute__((ext_vector_type(4)));
typedef double Factors __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const Factors* LdsFactors;

// position 4 f + k: its flag (and low code) bit at 8 f + 2 + k, its high code bit 4 places up (pamld_lut.hpp)
__host__ __device__ constexpr uint32_t low_bit(int position) { return 8u * (position >> 2) + 2u + (position & 3); }

__device__ __forceinline__ void bin_add(unsigned char* lds, uint32_t bin, uint32_t copy, uint32_t pass_filter, uint32_t word, unsigned long long value) {
    atomicAdd(reinterpret_cast<unsigned long long*>(lds + BINS_AT + bin * BIN_BYTES + copy * COPY_BYTES + pass_filter * 32u + word * 8u), value);
}

// mask ? value : 1.0, mask all ones or all zeros: two v_bitop3_b32 (truth table: a ? b : c)
__device__ __forceinline__ double one_unless(uint32_t mask, double value) {
    const uint32_t low = __builtin_amdgcn_bitop3_b32(mask, static_cast<uint32_t>(__double2loint(value)), 0u, 0xCA);
    const uint32_t high = __builtin_amdgcn_bitop3_b32(mask, static_cast<uint32_t>(__double2hiint(value)), 0x3ff00000u, 0xCA);
    return __hiloint2double(static_cast<int>(high), static_cast<int>(low));
}

// STATE: where a read's carried state comes from: 0 nothing to load, 1 the input flag bytes, 2 the status word of the
// stage before
template <int L, int STATE, bool PRIORS>
__global__ __launch_bounds__(THREADS) void pamld_few_kernel(StageArgs args) {
    static_assert(L >= 1 && L <= 8, "one or two words of four positions");
    constexpr int ROW_WORDS = (L + 3) / 4;
    constexpr uint32_t VALID0 = L >= 4 ? 0xffffffffu : ((1u << (8 * L)) - 1u);                               // the bytes of positions 0 .. 3
    constexpr uint32_t VALID1 = L <= 4 ? 0u : L >= 8 ? 0xffffffffu : ((1u << (8 * (L - 4))) - 1u);           // ... of positions 4 .. L - 1
    constexpr uint32_t FULL_MASK = (L >= 4 ? 0x3cu : (((1u << L) - 1u) << 2)) | (L <= 4 ? 0u : (((1u << (L - 4)) - 1u) << 10));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    double (*const factor)[2] = reinterpret_cast<double (*)[2]>(lds_raw + FACTOR_AT);             // [row][0 match | 1 mismatch over match]
    unsigned long long* const job = reinterpret_cast<unsigned long long*>(lds_raw + JOB_AT);
    uint32_t* const word_of = reinterpret_cast<uint32_t*>(lds_raw + WORDS_AT);
    int32_t* const index_of = reinterpret_cast<int32_t*>(lds_raw + INDEX_AT);                     // by bin: 0 for bin 0
    DecoderRef d = *constant_cast(args.decoder);
    const int n_barcode = d.n_barcode;

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t copy = lane & (COPIES - 1);

    for(int i = threadIdx.x; i < FACTOR_ROWS; i += THREADS) {
        const double m = i < 128 ? d.match_probability[i] : d.uniform_probability;
        const double e = i < 128 ? d.mismatch_probability[i] : d.uniform_probability;
        factor[i][0] = m;
        factor[i][1] = e / m;
    }
    for(uint32_t w = threadIdx.x; w < ((MAX_BARCODES + 1) * BIN_BYTES + 8 * STAT_JOB_WORDS) / 8; w += THREADS) { reinterpret_cast<unsigned long long*>(lds_raw + BINS_AT)[w] = 0; }
    if(threadIdx.x < MAX_BARCODES) {
        // the barcode's 2-bit codes A, C, G, T = 0 .. 3, bit-sliced like the reads'
        const int b = static_cast<int>(threadIdx.x);
        uint32_t sliced = 0;
        if(b < n_barcode) {
            for(int j = 0; j < L; ++j) {
                const uint32_t e = (d.barcode_packed[b * d.n_word + (j >> 3)] >> (4 * (j & 7))) & 0xfu;
                const uint32_t two = (0x30210u >> (2 * e)) & 3u;
                sliced |= ((two & 1u) << low_bit(j)) | ((two >> 1) << (low_bit(j) + 4u));
            }
        }
        word_of[b] = sliced;
    }
    if(threadIdx.x <= MAX_BARCODES) {
        const int bin = static_cast<int>(threadIdx.x);
        index_of[bin] = (bin >= 1 && bin <= n_barcode) ? d.barcode_index[bin - 1] : 0;
    }
    __syncthreads();
    // wave uniform: the barcodes' words in scalar registers
    uint32_t barcode_word[MAX_BARCODES];
#pragma unroll
    for(int b = 0; b < MAX_BARCODES; ++b) { barcode_word[b] = __builtin_amdgcn_readfirstlane(word_of[b]); }

    const int64_t n = args.n_read;
    const int64_t tiles = (n + 63) / 64;
    const int64_t tile_step = static_cast<int64_t>(gridDim.x) * WAVES;
    // this wave's segment of the HBM worklist
    const int64_t own_segment = static_cast<int64_t>(blockIdx.x) * WAVES + wave;
    uint32_t* const segment = args.worklist + own_segment * args.worklist_stride;
    uint32_t segment_count = 0;

    const int high_quality_threshold = min(max(d.high_quality_threshold, 0), 128);
    const uint32_t high_quality_addend = static_cast<uint32_t>(128 - high_quality_threshold) * 0x01010101u;
    const double confidence_threshold = d.confidence_threshold;
    const double random_barcode_probability = d.random_barcode_probability;
    const double adjusted_noise_probability = d.adjusted_noise_probability;
    const double common_prior = d.concentration[0];
    uint32_t outgoing_pass = 0;                       // TranscodingDecoder::count / pf_count (transcode.h:61-64): per wave, added once
    uint32_t outgoing_fail = 0;
    const int hq_threshold = d.high_quality_distance_threshold;
    const bool multiplexing = d.multiplexing != 0;
    const bool last_stage = args.last_stage != 0;

    struct Inputs { uint32_t code[2]; uint32_t quality[2]; uint32_t state; };
    auto fetch = [&](int64_t tile, Inputs& in) {
        const int64_t first = tile * 64;
        const uint32_t last = static_cast<uint32_t>(min(static_cast<int64_t>(64), n - first)) - 1u;
        const uint32_t at = min(lane, last);
        const uint32_t* const code_row = reinterpret_cast<const uint32_t*>(args.base + first * (4 * ROW_WORDS) + at * (4u * ROW_WORDS));
        const uint32_t* const quality_row = reinterpret_cast<const uint32_t*>(args.quality + first * (4 * ROW_WORDS) + at * (4u * ROW_WORDS));
        in.code[0] = code_row[0]; in.quality[0] = quality_row[0];
        if(ROW_WORDS > 1) { in.code[1] = code_row[1]; in.quality[1] = quality_row[1]; } else { in.code[1] = 0; in.quality[1] = 0; }
        if(STATE == 1) { in.state = static_cast<uint32_t>(args.input_flag[first + at]); }      // transcode.h:207-211
        else if(STATE == 2) { in.state = args.word[first + at]; }
        else { in.state = 0; }
    };
    auto gather = [](uint32_t planes) {            // bits 8 j and 8 j + 4 -> bits j and 4 + j
        const uint32_t t = planes | (planes >> 7);
        return (t | (t >> 14)) & 0xffu;
    };

    // ---- the wave's loop over its tiles -------------------------------------------------------------------------------
    int64_t tile = static_cast<int64_t>(blockIdx.x) * WAVES + wave;
    Inputs next;
    fetch(min(tile, tiles - 1), next);
    for(; tile < tiles; tile += tile_step) {
        const Inputs in = next;
        const int64_t first = tile * 64;
        const bool active = first + lane < n;
        const uint32_t at = active ? lane : static_cast<uint32_t>(n - 1 - first);
        const int64_t read_ordinal = first + at;
        fetch(min(tile + tile_step, tiles - 1), next);        // the inputs of the tile after this one

        bool incoming_qcfail;
        uint32_t incoming_channel;
        if(STATE != 2) { incoming_qcfail = (in.state & 1u) != 0; incoming_channel = 0; }
        else { incoming_qcfail = (in.state >> 31) != 0; incoming_channel = in.state & 0x7fffffffu; }

        uint32_t row[2] = {0, 0};                      // factor rows, 8 bits per position
        uint32_t observed = 0;                         // bit-sliced 2-bit codes; anything outside ACGT arbitrary
        uint32_t high_quality = 0;                     // flag bits (low_bit of the position)
        uint32_t outside = 0;                          // flag bits: not A, C, G or T (mismatches every barcode base)
#pragma unroll
        for(int w = 0; w < ROW_WORDS; ++w) {
            const uint32_t valid = w == 0 ? VALID0 : VALID1;
            const uint32_t x = in.code[w] & (0x0f0f0f0fu & valid);           // BAM codes
            const uint32_t q8 = in.quality[w] & valid;
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
        // per position P(match) and rho of its row; the all-match product
        double rho[L];
        double everything_matches = 1.0;
#pragma unroll
        for(int j = 0; j < L; ++j) {
            const uint32_t word = row[j >> 2];
            const int byte = j & 3;
            const uint32_t address = (byte == 0 ? (word << 4) : byte == 1 ? (word >> 4) : (word >> (8 * byte - 4))) & 0xff0u;      // FACTOR_AT is 0
            const Factors pair = *(LdsFactors)static_cast<uintptr_t>(address);
            everything_matches = j == 0 ? pair.x : everything_matches * pair.x;
            rho[j] = pair.y;
        }

        // every barcode: its probability over the all-match product (times its prior), the largest and the sum of the
        // others kept apart
        double top = 0.0;                              // the largest term
        double top_product = 0.0;                      // ... without its prior
        double others = 0.0;
        int best = -1;
        uint32_t best_flags = 0;
#pragma unroll
        for(int b = 0; b < MAX_BARCODES; ++b) {
            if(b < n_barcode) {                                                               // wave uniform
                const uint32_t difference = observed ^ barcode_word[b];
                const uint32_t flags = ((difference | (difference >> 4)) | outside) & FULL_MASK;
                double product = 1.0;
#pragma unroll
                for(int j = 0; j < L; ++j) {
                    const uint32_t mask = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(flags), low_bit(j), 1));     // all ones where position j differs
                    const double picked = one_unless(mask, rho[j]);
                    product = j == 0 ? picked : product * picked;
                }
                const double term = PRIORS ? product * d.concentration[b] : product;          // pamld.cpp:66, over A (and over the common prior)
                const bool takes_over = term > top;                                           // strict: the first keeps a tie
                others += takes_over ? top : term;
                best = takes_over ? b : best;
                best_flags = takes_over ? flags : best_flags;
                if(PRIORS) { top_product = takes_over ? product : top_product; }
                top = takes_over ? term : top;
            }
        }
        if(!PRIORS) { top_product = top; }
        // P(r|c), the adjusted probabilities of the winner and of everything (pamld.cpp:61-86), the posterior as winner * 1 / total
        const double best_conditional = top_product * everything_matches;
        const double best_p = PRIORS ? top * everything_matches : (top * everything_matches) * common_prior;
        const double others_p = PRIORS ? others * everything_matches : (others * everything_matches) * common_prior;
        const double total = (best_p + others_p) + adjusted_noise_probability;
        double inverse = __builtin_amdgcn_rcp(total);
        inverse = __builtin_fma(__builtin_fma(-total, inverse, 1.0), inverse, inverse);
        inverse = __builtin_fma(__builtin_fma(-total, inverse, 1.0), inverse, inverse);
        double confidence = best_p * inverse;                                             // pamld.cpp:92
        const double noise_margin = best_conditional - random_barcode_probability;
        const double confidence_margin = confidence - confidence_threshold;
        // the certificate of pamld_tiny.hip; "others at most half of the winner" stands for its runner up test
        const bool noise_clear = fabs(noise_margin) > 1e-9 * random_barcode_probability;
        const bool finished = best >= 0 && others <= 0.5 * top && noise_clear && (noise_margin <= 0 || fabs(confidence_margin) > 1e-9);

        // Outcome.  Every lane stores a record and a status word in straight line code — the real ones when the read
        // is finished; a deferred read's record is a placeholder (the tiers behind write it again), its status word the
        // state it came with; a lane past the end of the batch stores its (last) read's again
        const bool noise = !(noise_margin > 0);                                          // pamld.cpp:96
        const bool confident = confidence_margin > 0;                                    // pamld.cpp:99
        uint32_t distance = static_cast<uint32_t>(__popc(best_flags));
        uint32_t hq_distance = static_cast<uint32_t>(__popc(best_flags & high_quality));
        const uint32_t argmax_bin = static_cast<uint32_t>(best < 0 ? 0 : best) + 1u;
        const bool qcfail = incoming_qcfail || noise || !confident || (hq_threshold > 0 && static_cast<int>(hq_distance) >= hq_threshold);
        const uint32_t bin = noise ? 0u : argmax_bin;
        if(noise) { distance = 0; hq_distance = 0; confidence = 0; }
        const uint32_t status = noise ? STATUS_NOISE : confident ? STATUS_PASS : STATUS_LOW_CONFIDENCE;
        const uint32_t index = static_cast<uint32_t>(index_of[bin]);
        const bool qcfail_out = finished ? qcfail : incoming_qcfail;
        {
            Quad out;
            out.x = finished ? index : 0u;
            out.y = finished ? (distance | (hq_distance << 8) | (status << 16) | (qcfail_out ? 1u << 24 : 0u)) : ((STATUS_UNCLASSIFIED << 16) | (qcfail_out ? 1u << 24 : 0u));
            out.z = finished ? static_cast<uint32_t>(__double2loint(confidence)) : 0u;
            out.w = finished ? static_cast<uint32_t>(__double2hiint(confidence)) : 0u;
            __builtin_nontemporal_store(out, reinterpret_cast<Quad*>(reinterpret_cast<char*>(args.result) + first * 16 + at * 16u));
            const uint32_t channel = (finished && multiplexing) ? index : incoming_channel;
            __builtin_nontemporal_store((channel & 0x7fffffffu) | (qcfail_out ? 0x80000000u : 0u), reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(args.word) + first * 4 + at * 4u));
        }
        if(finished && active) {
            const uint32_t half = qcfail ? 0u : 1u;
            if(noise) {
                bin_add(lds_raw, argmax_bin, copy, 0, 3, 1ull);                                 // low conditional confidence count, charged to the argmax
            } else if(confident) {
                const unsigned long long fixed = static_cast<unsigned long long>(__double2ll_rn(confidence * 4503599627370496.0));      // 2^-52 fixed point, two limbs
                bin_add(lds_raw, bin, copy, half, 1, fixed & ((1ull << CONFIDENCE_LIMB_BITS) - 1));
                bin_add(lds_raw, bin, copy, half, 2, fixed >> CONFIDENCE_LIMB_BITS);
            } else {
                bin_add(lds_raw, bin, copy, 0, 3, 1ull << 32);                                  // low confidence count (pamld.cpp:104-106)
            }
            // Decoder<T>::classify then Classifier<T>::classify (decoder.h:68-76, classifier.h:78-86)
            bin_add(lds_raw, bin, copy, half, 0, 1ull | (static_cast<unsigned long long>(distance) << 32));
        }
        const unsigned long long handed_on = __ballot(active && !finished);
        if(handed_on) {
            if(active && !finished) { segment[segment_count + __popcll(handed_on & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(read_ordinal); }
            segment_count += static_cast<uint32_t>(__popcll(handed_on));
        }
        if(last_stage) {
            outgoing_pass += static_cast<uint32_t>(__popcll(__ballot(active && finished && !qcfail)));
            outgoing_fail += static_cast<uint32_t>(__popcll(__ballot(active && finished && qcfail)));
        }
    }
    if(lane == 0) {
        args.worklist_count[own_segment] = segment_count;
        if(outgoing_fail) { atomicAdd(&job[0], static_cast<unsigned long long>(outgoing_fail)); }
        if(outgoing_pass) { atomicAdd(&job[1], static_cast<unsigned long long>(outgoing_pass)); }
    }
    __syncthreads();
    // the workgroup's bins into the counter buffer: the copies of a word summed, one global atomic per field that is not zero
    outgoing_flush(args, job);
    for(int i = threadIdx.x; i < (n_barcode + 1) * BIN_WORDS; i += THREADS) {
        const int bin = i / BIN_WORDS;
        const int pass_filter = (i % BIN_WORDS) / 4;
        const int word = i % 4;
        unsigned long long v = 0;
        for(int c = 0; c < COPIES; ++c) { v += *reinterpret_cast<const unsigned long long*>(lds_raw + BINS_AT + static_cast<uint32_t>(bin) * BIN_BYTES + static_cast<uint32_t>(c) * COPY_BYTES + 8u * (i % BIN_WORDS)); }
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

template <int L, bool PRIORS>
int launch_one(const StageArgs& args, dim3 grid, hipStream_t stream) {
    if(!args.first_stage) { hipLaunchKernelGGL((pamld_few_kernel<L, 2, PRIORS>), grid, dim3(THREADS), LDS_BYTES, stream, args); }
    else if(args.input_flag) { hipLaunchKernelGGL((pamld_few_kernel<L, 1, PRIORS>), grid, dim3(THREADS), LDS_BYTES, stream, args); }
    else { hipLaunchKernelGGL((pamld_few_kernel<L, 0, PRIORS>), grid, dim3(THREADS), LDS_BYTES, stream, args); }
    return static_cast<int>(hipGetLastError());
}

template <int LO, int HI, bool PRIORS>
int launch_range(const StageArgs& args, int length, dim3 grid, hipStream_t stream) {
    if constexpr(LO > HI) { (void)args; (void)length; (void)grid; (void)stream; return static_cast<int>(hipErrorInvalidValue); }
    else {
        if(length == LO) { return launch_one<LO, PRIORS>(args, grid, stream); }
        return launch_range<LO + 1, HI, PRIORS>(args, length, grid, stream);
    }
}

}  // namespace few

int pamld_few_waves() { return few::WAVES; }

// what pamld_tiny.hip takes, up to 8 nt, with a workgroup histogram
bool pamld_few_supports(const DevDecoder& host_decoder) {
    const int L = host_decoder.nucleotide_cardinality;
    return pamld_tiny_supports(host_decoder) && host_decoder.n_barcode <= few::MAX_BARCODES && L >= 1 && L <= 8 &&
           host_decoder.plane_stride == 4 * ((L + 3) / 4) && host_decoder.use_lds_histogram != 0;
}

int launch_pamld_few(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, hipStream_t stream) {
    const int L = host_decoder.nucleotide_cardinality;
    return host_decoder.uniform_prior ? few::launch_range<1, 8, false>(args, L, grid, stream) : few::launch_range<1, 8, true>(args, L, grid, stream);
}

}  // namespace phq
