// K1 (product form): PamlDecoder<T>::classify for reads whose winner is unambiguous.
//
// pow(base, sum of costs) = product of pow(base, cost): per read and position there are only two factors,
// P(match) = pow(base, tpq[q]) and P(mismatch) = pow(base, q) (or pow(base, 10 log10 4) for a base outside
// ACGT; 1 for q = 0) — reference phred.cpp:44-68, barcode.h:151-163.  So P(r|b) is a product of L factors
// selected by the mismatch mask of (read, barcode).  The kernel
//   * builds, per read, one 16 entry table per group of 4 positions (every product of 4 factors), in LDS;
//   * gets a barcode's mismatch mask with XOR + nibble fold on 4-bit packed bases (scalar barcode operand),
//     turns each 4 flag group into a table index with one 24-bit multiply, and multiplies the G table entries;
//   * accumulates sum_b P(r|b) concentration_b and the argmax in fp64.
// No exp, no Kahan chain: ~16 VALU instructions per (read, barcode) instead of ~75.
//
// The reference's argmax between NEAR-TIED barcodes is decided by the last bits of its own Kahan sums
// (pamld_literal.hpp), and the posterior must hold 1e-6 (held here to ~1e-12).  So a read is only FINISHED in
// this kernel when its winner dominates: the summed probability of every other barcode is at most 2e-6 of the
// winner's.  Then (i) the argmax cannot be a tie, (ii) the tables and the running sum can be single precision
// — their ~1e-6 relative error moves the posterior by at most 2e-6 * 1e-6 — while the winner's own probability
// and the posterior are recomputed in fp64 from the fp64 factor tables, and (iii) the noise and confidence
// filters are decided in fp64, unless they fall within 1e-9 of their thresholds.  Every other read (noise
// reads, reads with a second plausible barcode: a few %) is appended to a worklist that the literal kernel
// decodes right after, on the same stream.  Single precision tables halve the LDS per workgroup (33 KiB for
// 8 nt) which doubles the resident waves; the kernel is bound by VALU issue, not by LDS or HBM.
#include "kernel_common.hpp"

namespace phq {
namespace {

// bit 4k of the result is set when nibble k of x is non zero
__device__ __forceinline__ uint32_t nibble_flags(uint32_t x) {
    x |= x >> 1;
    x |= x >> 2;
    return x & 0x11111111u;
}

// Four flags at bits 0,4,8,12 of a 16-bit half word -> a 4-bit index at bits 11..14, by one multiply:
// 0x924 = 2^11 + 2^8 + 2^5 + 2^2 moves flag k (bit 4k) to bit 11 + k; no two partial products meet.
// Four flags at bits 0,4,8,12 -> a 4-bit index at bits 10..13 (table entry * 1024 bytes = 256 lanes * 4 bytes)
constexpr uint32_t GATHER = 0x492u;           // 2^10 + 2^7 + 2^4 + 2^1
constexpr uint32_t INDEX_MASK = 0x3c00u;
constexpr double DOMINANCE = 1e-4;            // others <= DOMINANCE * winner, see the header

template <int L, bool UNIFORM_PRIOR>
__global__ __launch_bounds__(BLOCK) void pamld_lut_kernel(StageArgs args) {
    constexpr int G = (L + 3) / 4;            // groups of 4 positions
    constexpr int LP = (L + 3) / 4 * 4;
    static_assert(L <= 16, "one or two packed words");
    // static LDS (addresses are compile time constants, so table reads need no base add): the tables and the
    // two fp64 factor columns; dynamic LDS: the counter histogram, sized per decoder
    __shared__ float table[G * 16 * BLOCK];                                   // [G * 16][BLOCK]
    __shared__ double match_probability[128];
    __shared__ double mismatch_probability[128];
    __shared__ uint32_t deferred_total;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw);
    DecoderRef d = *constant_cast(args.decoder);

    for(int i = threadIdx.x; i < 128; i += BLOCK) {
        match_probability[i] = d.match_probability[i];
        mismatch_probability[i] = d.mismatch_probability[i];
    }
    if(threadIdx.x == 0) { deferred_total = 0; }
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats};
    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const PHQ_CONSTANT uint32_t* packed = constant_cast(d.barcode_packed);
    const PHQ_CONSTANT double* concentration = constant_cast(d.concentration);
    const int n_barcode = d.n_barcode;
    constexpr int W = (G + 1) / 2;            // packed words per barcode
    const double uniform_probability = d.uniform_probability;
    const int64_t n = args.n_read;
    const int64_t tiles = (n + BLOCK - 1) / BLOCK;
    constexpr uint32_t GROUP_BYTES = 16 * BLOCK * sizeof(float);
    constexpr uint32_t ENTRY_BYTES = BLOCK * sizeof(float);
    const uint32_t lane_offset = threadIdx.x * static_cast<uint32_t>(sizeof(float));
    const float prior = UNIFORM_PRIOR ? static_cast<float>(concentration[0]) : 1.0f;

    for(int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t i = tile * BLOCK + threadIdx.x;
        const bool active = i < n;
        ReadState r{false, 0};
        bool defer = false;
        if(active) {
            r = load_read_state(args, i);
            uint8_t code[LP];
            uint8_t quality[LP];
            load_plane_row<L>(args.base, i, code);
            load_plane_row<L>(args.quality, i, quality);

            // per position factors, the packed observation and the high quality flags
            float fm[L];
            float fe[L];
            uint32_t observed[W];
            uint32_t high_quality[W];
#pragma unroll
            for(int w = 0; w < W; ++w) { observed[w] = 0; high_quality[w] = 0; }
#pragma unroll
            for(int j = 0; j < L; ++j) {
                const uint32_t c = code[j];
                const uint32_t q = quality[j] & 0x7fu;
                const double u = q ? uniform_probability : 1.0;
                const bool strict = strict_code(c);
                const double m = match_probability[q];
                const double e = mismatch_probability[q];
                fm[j] = static_cast<float>(strict ? m : u);
                fe[j] = static_cast<float>(strict ? e : u);
                observed[j >> 3] |= (c & 0xfu) << (4 * (j & 7));
                if(static_cast<int>(quality[j]) >= d.high_quality_threshold) { high_quality[j >> 3] |= 1u << (4 * (j & 7)); }
            }
            // tables: entry m of group g = product over the group's positions of (bit ? mismatch : match)
#pragma unroll
            for(int g = 0; g < G; ++g) {
                constexpr int FULL = 4;
                const int base = 4 * g;
                float lo[4];
                float hi[4];
                lo[0] = fm[base];
                lo[1] = fe[base];
                lo[2] = lo[0];
                lo[3] = lo[1];
                if(base + 1 < L) { lo[0] *= fm[base + 1]; lo[1] *= fm[base + 1]; lo[2] *= fe[base + 1]; lo[3] *= fe[base + 1]; }
                hi[0] = hi[1] = hi[2] = hi[3] = 1.0f;
                if(base + 2 < L) { hi[0] = fm[base + 2]; hi[1] = fe[base + 2]; hi[2] = hi[0]; hi[3] = hi[1]; }
                if(base + 3 < L) { hi[0] *= fm[base + 3]; hi[1] *= fm[base + 3]; hi[2] *= fe[base + 3]; hi[3] *= fe[base + 3]; }
                (void)FULL;
#pragma unroll
                for(int m = 0; m < 16; ++m) {
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(table) + g * GROUP_BYTES + m * ENTRY_BYTES + lane_offset) = lo[m & 3] * hi[m >> 2];
                }
            }

            // single precision scan: the maximum, where it is, and the sum of everything BUT the maximum (summing
            // it apart keeps the small remainder from being rounded away against the winner)
            float sigma = 0;
            float best_value = 0;
            int best = -1;
#pragma unroll 4
            for(int b = 0; b < n_barcode; ++b) {
                float conditional = 1.0f;
#pragma unroll
                for(int w = 0; w < W; ++w) {
                    const uint32_t flags = nibble_flags(observed[w] ^ packed[b * W + w]);
                    const uint32_t low = __umul24(flags, GATHER) & INDEX_MASK;        // flags 4 and 5 land above bit 13
                    const float a = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(table) + (2 * w) * GROUP_BYTES + (low | lane_offset));
                    conditional = (w == 0) ? a : conditional * a;
                    if(2 * w + 1 < G) {
                        const uint32_t high = __umul24(flags >> 16, GATHER) & INDEX_MASK;
                        conditional *= *reinterpret_cast<const float*>(reinterpret_cast<const char*>(table) + (2 * w + 1) * GROUP_BYTES + (high | lane_offset));
                    }
                }
                const float p = UNIFORM_PRIOR ? conditional : conditional * static_cast<float>(concentration[b]);
                sigma += fminf(p, best_value);
                if(p > best_value) { best = b; }
                best_value = fmaxf(p, best_value);
            }

            if(best < 0) {
                defer = true;
            } else {
                // the winner's mask, its distances, and its probability again in fp64 from the fp64 factors
                uint32_t flags[W];
                int distance = 0;
                int hq_distance = 0;
#pragma unroll
                for(int w = 0; w < W; ++w) {
                    flags[w] = nibble_flags(observed[w] ^ d.barcode_packed[static_cast<int64_t>(best) * W + w]) & d.full_mask[w];
                    distance += __popc(flags[w]);
                    hq_distance += __popc(flags[w] & high_quality[w]);
                }
                double best_conditional = 1.0;
#pragma unroll
                for(int j = 0; j < L; ++j) {
                    const uint32_t c = code[j];
                    const uint32_t q = quality[j] & 0x7fu;
                    const double u = q ? uniform_probability : 1.0;
                    const bool mismatch = ((flags[j >> 3] >> (4 * (j & 7))) & 1u) != 0;
                    const double m = match_probability[q];
                    const double e = mismatch_probability[q];
                    best_conditional *= strict_code(c) ? (mismatch ? e : m) : u;
                }
                const double best_p = best_conditional * concentration[best];
                // everything but the winner, in units of p: only ever compared against and added to much larger terms
                const double others = static_cast<double>(sigma) * static_cast<double>(prior);
                const double total = (best_p + others) + d.adjusted_noise_probability;   // pamld.cpp:61-86
                double confidence = best_p / total;                                       // pamld.cpp:92
                const double noise_margin = best_conditional - d.random_barcode_probability;
                const double confidence_margin = confidence - d.confidence_threshold;
                // certificate: a dominant winner and no decision within 1e-9 (relative) of its threshold
                defer = !(others <= DOMINANCE * best_p)
                     || !(fabs(noise_margin) > 1e-9 * d.random_barcode_probability)
                     || !(fabs(confidence_margin) > 1e-9);
                if(!defer) {
                    int bin = best + 1;
                    uint8_t status = STATUS_PASS;
                    if(noise_margin > 0) {                                                // pamld.cpp:96
                        if(confidence_margin > 0) {                                       // pamld.cpp:99
                            if(d.high_quality_distance_threshold > 0 && hq_distance >= d.high_quality_distance_threshold) {
                                r.qcfail = true;
                            }
                            stat_add_confidence(d, sink, bin, r.qcfail ? 0 : 1, confidence);
                        } else {
                            stat_add(d, sink, bin, 0, FIELD_LOW_CONFIDENCE, 1ull);
                            r.qcfail = true;
                            status = STATUS_LOW_CONFIDENCE;
                        }
                    } else {
                        stat_add(d, sink, bin, 0, FIELD_LOW_CONDITIONAL_CONFIDENCE, 1ull);
                        r.qcfail = true;
                        bin = 0;
                        distance = 0;
                        hq_distance = 0;
                        confidence = 0;
                        status = STATUS_NOISE;
                    }
                    decoder_tail(d, sink, r, bin, distance);
                    if(args.result) {
                        DevResult out;
                        out.index = bin > 0 ? d.barcode_index[bin - 1] : 0;
                        out.distance = static_cast<uint8_t>(distance > 255 ? 255 : distance);
                        out.hq_distance = static_cast<uint8_t>(hq_distance > 255 ? 255 : hq_distance);
                        out.status = status;
                        out.qcfail = r.qcfail ? 1 : 0;
                        out.confidence = confidence;
                        args.result[i] = out;
                    }
                    store_read_state(args, i, r);
                }
            }
        }
        // deferred reads: one LDS add per wave reserves a run of slots in this workgroup's worklist segment
        const unsigned long long deferred = __ballot(active && defer);
        if(deferred) {
            const int lane = threadIdx.x & 63;
            uint32_t first = 0;
            if(lane == 0) { first = atomicAdd(&deferred_total, static_cast<uint32_t>(__popcll(deferred))); }
            first = __shfl(first, 0);
            if(active && defer) {
                segment[first + __popcll(deferred & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(i);
            }
        }
        count_outgoing(args, histogram, active && !defer, r);
    }
    __syncthreads();
    if(threadIdx.x == 0) { args.worklist_count[blockIdx.x] = deferred_total; }
    histogram_flush(args, d, histogram);
}

template <int L, bool UNIFORM_PRIOR>
int launch_one(const StageArgs& args, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    hipLaunchKernelGGL((pamld_lut_kernel<L, UNIFORM_PRIOR>), grid, dim3(BLOCK), histogram_bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}

template <int LO, int HI>
int launch_range(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if constexpr(LO > HI) {
        (void)args; (void)host_decoder; (void)grid; (void)histogram_bytes; (void)stream;
        return static_cast<int>(hipErrorInvalidValue);
    } else {
        if(host_decoder.nucleotide_cardinality == LO) {
            return host_decoder.uniform_prior ? launch_one<LO, true>(args, grid, histogram_bytes, stream)
                                              : launch_one<LO, false>(args, grid, histogram_bytes, stream);
        }
        return launch_range<LO + 1, HI>(args, host_decoder, grid, histogram_bytes, stream);
    }
}

}  // namespace

bool pamld_lut_supports(const DevDecoder& host_decoder) {
    return host_decoder.all_strict && host_decoder.nucleotide_cardinality >= 1 && host_decoder.nucleotide_cardinality <= 16 &&
           host_decoder.n_barcode > 0;
}

int launch_pamld_lut(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return launch_range<1, 16>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
