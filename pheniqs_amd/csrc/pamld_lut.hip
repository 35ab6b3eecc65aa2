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
// The result differs from the reference's arithmetic by ~L ulp, far inside the 1e-6 posterior contract, but
// the reference's argmax between NEAR-TIED barcodes is decided by the last bits of its own Kahan sums
// (pamld_literal.hpp).  A read is therefore only finished here when its winner leads every other barcode by
// 4x or more and no filter decision is within 1e-9 of its threshold; every other read (noise reads, ambiguous
// reads: ~1-2 %) is appended to a worklist that the literal kernel decodes right after, on the same stream.
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
constexpr uint32_t GATHER = 0x924u;
constexpr uint32_t INDEX_MASK = 0x7800u;      // bits 11..14 = table entry * 2048 bytes (256 lanes * 8 bytes)

template <int L>
__global__ __launch_bounds__(BLOCK) void pamld_lut_kernel(StageArgs args) {
    constexpr int G = (L + 3) / 4;            // groups of 4 positions
    constexpr int LP = (L + 3) / 4 * 4;
    static_assert(L <= 16, "one or two packed words");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* match_probability = reinterpret_cast<double*>(lds_raw);
    double* mismatch_probability = match_probability + 128;
    double* table = mismatch_probability + 128;                               // [G * 16][BLOCK]
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(table + G * 16 * BLOCK);
    DecoderRef d = *constant_cast(args.decoder);

    for(int i = threadIdx.x; i < 128; i += BLOCK) {
        match_probability[i] = d.match_probability[i];
        mismatch_probability[i] = d.mismatch_probability[i];
    }
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats};
    const PHQ_CONSTANT uint32_t* packed = constant_cast(d.barcode_packed);
    const PHQ_CONSTANT double* concentration = constant_cast(d.concentration);
    const int n_barcode = d.n_barcode;
    const int n_word = d.n_word;
    const double uniform_probability = d.uniform_probability;
    const int64_t n = args.n_read;
    const int64_t tiles = (n + BLOCK - 1) / BLOCK;
    // byte offsets from the start of LDS, so that the compiler folds the constant part into the DS offset field
    constexpr uint32_t TABLE_BASE = 256 * sizeof(double);
    constexpr uint32_t GROUP_BYTES = 16 * BLOCK * sizeof(double);
    constexpr uint32_t ENTRY_BYTES = BLOCK * sizeof(double);
    const uint32_t lane_offset = threadIdx.x * static_cast<uint32_t>(sizeof(double));

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
            double fm[L];
            double fe[L];
            uint32_t observed[2] = {0, 0};
            uint32_t high_quality[2] = {0, 0};
#pragma unroll
            for(int j = 0; j < L; ++j) {
                const uint32_t c = code[j];
                const uint32_t q = quality[j] & 0x7fu;
                const double u = q ? uniform_probability : 1.0;
                const bool strict = strict_code(c);
                fm[j] = strict ? match_probability[q] : u;
                fe[j] = strict ? mismatch_probability[q] : u;
                observed[j >> 3] |= (c & 0xfu) << (4 * (j & 7));
                if(static_cast<int>(quality[j]) >= d.high_quality_threshold) { high_quality[j >> 3] |= 1u << (4 * (j & 7)); }
            }
            // tables: entry m of group g = product over the group's positions of (bit ? mismatch : match)
#pragma unroll
            for(int g = 0; g < G; ++g) {
                constexpr int FULL = 4;
                const int base = 4 * g;
                double lo[4];
                double hi[4];
                lo[0] = fm[base];
                lo[1] = fe[base];
                lo[2] = lo[0];
                lo[3] = lo[1];
                if(base + 1 < L) { lo[0] *= fm[base + 1]; lo[1] *= fm[base + 1]; lo[2] *= fe[base + 1]; lo[3] *= fe[base + 1]; }
                hi[0] = hi[1] = hi[2] = hi[3] = 1.0;
                if(base + 2 < L) { hi[0] = fm[base + 2]; hi[1] = fe[base + 2]; hi[2] = hi[0]; hi[3] = hi[1]; }
                if(base + 3 < L) { hi[0] *= fm[base + 3]; hi[1] *= fm[base + 3]; hi[2] *= fe[base + 3]; hi[3] *= fe[base + 3]; }
                (void)FULL;
#pragma unroll
                for(int m = 0; m < 16; ++m) {
                    *reinterpret_cast<double*>(lds_raw + TABLE_BASE + g * GROUP_BYTES + m * ENTRY_BYTES + lane_offset) = lo[m & 3] * hi[m >> 2];
                }
            }

            double sigma_p = 0;
            double best_p = 0;
            int best = -1;
#pragma unroll 4
            for(int b = 0; b < n_barcode; ++b) {
                double conditional = 1.0;
#pragma unroll
                for(int w = 0; w < (G + 1) / 2; ++w) {
                    const uint32_t flags = nibble_flags(observed[w] ^ packed[b * n_word + w]);
                    const uint32_t low = __umul24(flags, GATHER) & INDEX_MASK;        // flags 4 and 5 land above bit 14
                    const double a = *reinterpret_cast<const double*>(lds_raw + TABLE_BASE + (2 * w) * GROUP_BYTES + (low | lane_offset));
                    conditional = (w == 0) ? a : conditional * a;
                    if(2 * w + 1 < G) {
                        const uint32_t high = __umul24(flags >> 16, GATHER) & INDEX_MASK;
                        conditional *= *reinterpret_cast<const double*>(lds_raw + TABLE_BASE + (2 * w + 1) * GROUP_BYTES + (high | lane_offset));
                    }
                }
                const double p = conditional * concentration[b];
                sigma_p += p;
                if(p > best_p) {
                    best = b;
                    best_p = p;
                }
            }

            if(best < 0) {
                defer = true;
            } else {
                // the winner's mask, distances and conditional probability (fp64 product of its own factors)
                uint32_t flags[2] = {0, 0};
                int distance = 0;
                int hq_distance = 0;
#pragma unroll
                for(int w = 0; w < (G + 1) / 2; ++w) {
                    flags[w] = nibble_flags(observed[w] ^ d.barcode_packed[static_cast<int64_t>(best) * n_word + w]) & d.full_mask[w];
                    distance += __popc(flags[w]);
                    hq_distance += __popc(flags[w] & high_quality[w]);
                }
                double best_conditional = 1.0;
#pragma unroll
                for(int j = 0; j < L; ++j) {
                    best_conditional *= ((flags[j >> 3] >> (4 * (j & 7))) & 1u) ? fe[j] : fm[j];
                }
                const double total = sigma_p + d.adjusted_noise_probability;          // pamld.cpp:83-86
                double confidence = best_p / total;                                       // pamld.cpp:92
                const double others = sigma_p - best_p;
                const double noise_margin = best_conditional - d.random_barcode_probability;
                const double confidence_margin = confidence - d.confidence_threshold;
                // certificate: a clear winner and no decision within 1e-9 (relative) of its threshold
                defer = !(others < 0.25 * best_p)
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
        // deferred reads: one atomic per wave reserves a run of worklist slots
        const unsigned long long deferred = __ballot(active && defer);
        if(deferred) {
            const int lane = threadIdx.x & 63;
            uint32_t first = 0;
            if(lane == 0) { first = atomicAdd(args.worklist_count, static_cast<uint32_t>(__popcll(deferred))); }
            first = __shfl(first, 0);
            if(active && defer) {
                args.worklist[first + __popcll(deferred & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(i);
            }
        }
        count_outgoing(args, histogram, active && !defer, r);
    }
    __syncthreads();
    histogram_flush(args, d, histogram);
}

template <int LO, int HI>
int launch_range(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if constexpr(LO > HI) {
        (void)args; (void)host_decoder; (void)grid; (void)histogram_bytes; (void)stream;
        return static_cast<int>(hipErrorInvalidValue);
    } else {
        if(host_decoder.nucleotide_cardinality == LO) {
            constexpr int G = (LO + 3) / 4;
            const size_t lds_bytes = (256 + static_cast<size_t>(G) * 16 * BLOCK) * sizeof(double) + histogram_bytes;
            static bool configured = false;
            if(!configured) {
                // more than the default 64 KiB of dynamic LDS for the longer barcodes
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pamld_lut_kernel<LO>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                configured = true;
            }
            hipLaunchKernelGGL((pamld_lut_kernel<LO>), grid, dim3(BLOCK), lds_bytes, stream, args);
            return static_cast<int>(hipGetLastError());
        }
        return launch_range<LO + 1, HI>(args, host_decoder, grid, histogram_bytes, stream);
    }
}

}  // namespace

bool pamld_lut_supports(const DevDecoder& host_decoder) {
    return host_decoder.all_strict && host_decoder.nucleotide_cardinality >= 1 && host_decoder.nucleotide_cardinality <= 12 &&
           host_decoder.n_barcode > 0;
}

int launch_pamld_lut(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return launch_range<1, 12>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
