// K3b: the counters of a large codec from the result records of the stage.
//
// A decoder with more bins than fit a workgroup histogram (C3: 9 217) used to send ~3.3 scattered global atomics
// per read; global atomics execute at the memory side on this part, one 64 byte request per lane, and cost as
// much as the classification itself.  Instead the stage kernels skip count / distance / confidence of every
// classified read (StageArgs::defer_counters) and this kernel rebuilds them from the 16 byte result records:
//   reference decoder.h:68-76 (accumulated distance), classifier.h:78-86 (count, pf count), pamld.cpp:99-105
//   (accumulated confidence of the reads that pass the confidence filter)
// One workgroup per CU keeps the pass filter half of every bin in LDS — 16 bytes per bin: count | distance in one
// word, the confidence as 2^-44 fixed point in the other (at most 2^19 reads per workgroup, so neither can
// overflow) — and flushes it once, bin after bin (contiguous words).  Reads that failed QC (a few %) go to
// global atomics directly.  Rounding a confidence to 2^-44 moves a sum by < 3e-14 relative; sums stay
// independent of the order reads are visited in.
#include "kernel_common.hpp"

namespace phq {
namespace {

constexpr int COUNT_THREADS = 1024;
constexpr int MAX_DEFERRED_BINS = 9984;                  // 16 bytes each: 156 KiB of the 160 KiB of LDS
constexpr int64_t READS_PER_WORKGROUP = 1 << 19;
constexpr int CONFIDENCE_SHIFT = CONFIDENCE_FRACTION_BITS - 44;

__global__ __launch_bounds__(COUNT_THREADS) void count_results_kernel(StageArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* const bins = reinterpret_cast<unsigned long long*>(lds_raw);       // [bin][count | distance << 32, confidence]
    DecoderRef d = *constant_cast(args.decoder);
    const int n_bin = d.n_barcode + 1;
    for(int w = threadIdx.x; w < 2 * n_bin; w += COUNT_THREADS) { bins[w] = 0; }
    __syncthreads();

    const int64_t n = args.n_read;
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t first = per_block * blockIdx.x;
    const int64_t last = first + per_block < n ? first + per_block : n;
    for(int64_t i = first + threadIdx.x; i < last; i += COUNT_THREADS) {
        const DevResult record = args.result[i];
        if(record.index <= 0) { continue; }              // unclassified: bin 0 was counted by the stage kernel
        const int bin = record.index;                    // identity numbering (count_results_supports)
        const unsigned long long fixed = record.status == STATUS_PASS
            ? static_cast<unsigned long long>(__double2ll_rn(record.confidence * 17592186044416.0)) : 0ull;      // 2^44
        if(record.qcfail == 0) {
            atomicAdd(&bins[2 * bin], 1ull | (static_cast<unsigned long long>(record.distance) << 32));
            if(fixed) { atomicAdd(&bins[2 * bin + 1], fixed); }
        } else {
            unsigned long long* const words = args.stats + d.stats_offset + static_cast<int64_t>(bin) * STAT_WORDS_PER_BIN;
            atomicAdd(&words[FIELD_COUNT], 1ull);
            if(record.distance) { atomicAdd(&words[FIELD_DISTANCE], static_cast<unsigned long long>(record.distance)); }
            if(fixed) {
                const unsigned long long exact = fixed << CONFIDENCE_SHIFT;
                atomicAdd(&words[FIELD_CONFIDENCE_LOW], exact & ((1ull << CONFIDENCE_LIMB_BITS) - 1));
                atomicAdd(&words[FIELD_CONFIDENCE_HIGH], exact >> CONFIDENCE_LIMB_BITS);
            }
        }
    }
    __syncthreads();
    for(int bin = 1 + threadIdx.x; bin < n_bin; bin += COUNT_THREADS) {
        const unsigned long long packed = bins[2 * bin];
        if(packed == 0) { continue; }
        unsigned long long* const words = args.stats + d.stats_offset + static_cast<int64_t>(bin) * STAT_WORDS_PER_BIN + STAT_FIELDS;   // pass filter half
        atomicAdd(&words[FIELD_COUNT], packed & 0xffffffffull);
        if(packed >> 32) { atomicAdd(&words[FIELD_DISTANCE], packed >> 32); }
        const unsigned long long confidence = bins[2 * bin + 1];
        if(confidence) {
            // 2^-44 units back to the two 26 bit limbs of the 2^-52 fixed point sum (the high limb takes the carry)
            const unsigned long long low = (confidence & ((1ull << (CONFIDENCE_LIMB_BITS - CONFIDENCE_SHIFT)) - 1)) << CONFIDENCE_SHIFT;
            atomicAdd(&words[FIELD_CONFIDENCE_LOW], low);
            atomicAdd(&words[FIELD_CONFIDENCE_HIGH], confidence >> (CONFIDENCE_LIMB_BITS - CONFIDENCE_SHIFT));
        }
    }
}

}  // namespace

// bins that fit the kernel's LDS, numbered the way the compiler numbers them, and no workgroup histogram
bool count_results_supports(const DevDecoder& host_decoder) {
    return !host_decoder.use_lds_histogram && host_decoder.identity_index && host_decoder.n_barcode + 1 <= MAX_DEFERRED_BINS;
}

int launch_count_results(const StageArgs& args, const DevDecoder& host_decoder, int32_t compute_units, hipStream_t stream) {
    static std::atomic<uint64_t> prepared{0};
    if(const int status = allow_large_lds(&count_results_kernel, prepared)) { return status; }
    const int64_t by_reads = (args.n_read + READS_PER_WORKGROUP - 1) / READS_PER_WORKGROUP;
    const int64_t blocks = std::max<int64_t>(std::min<int64_t>(compute_units, (args.n_read + COUNT_THREADS - 1) / COUNT_THREADS), by_reads);
    const size_t lds_bytes = static_cast<size_t>(host_decoder.n_barcode + 1) * 16;
    hipLaunchKernelGGL(count_results_kernel, dim3(static_cast<unsigned>(std::max<int64_t>(blocks, 1))), dim3(COUNT_THREADS), lds_bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}

}  // namespace phq
