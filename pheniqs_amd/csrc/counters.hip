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
// overflow) — and flushes it once, bin after bin (contiguous words).  Reads that failed QC (a few %) are queued in
// LDS, a queue per wave, and leave as dense global atomics — four lanes to an entry: its count, distance and the two limbs of
// its confidence are 32 contiguous bytes of the counter buffer — whenever the next tile could overflow the queue: sent
// directly, almost every wave would issue four global atomic instructions for its one or two failed lanes, and those
// instructions (one memory side request per lane whatever the lane count, ~50 ns each per CU) were three quarters of the
// kernel's time.  (The records a lane holds in flight do not matter: 2, 4 and 8 tiles measured the same, 0.37 ms for C3's
// 1.6 GB of records = 4.3 TB/s.)
// Rounding a confidence to 2^-44 moves a sum by < 3e-14 relative; sums stay independent of the order reads are
// visited in.
#include "kernel_common.hpp"

namespace phq {
namespace {

constexpr int COUNT_THREADS = 1024;
constexpr int QUEUE = 768;                               // failed reads waiting for their global atomics: 12 bytes each
constexpr int64_t READS_PER_WORKGROUP = 1 << 19;
constexpr int CONFIDENCE_SHIFT = CONFIDENCE_FRACTION_BITS - 44;

__global__ __launch_bounds__(COUNT_THREADS) void count_results_kernel(StageArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* const bins = reinterpret_cast<unsigned long long*>(lds_raw);       // [bin][count | distance << 32, confidence]
    DecoderRef d = *constant_cast(args.decoder);
    const int n_bin = d.n_barcode + 1;
    unsigned long long* const queue_fixed = bins + 2 * n_bin;                               // [QUEUE] confidence, 2^-44 units
    uint32_t* const queue_key = reinterpret_cast<uint32_t*>(queue_fixed + QUEUE);           // [QUEUE] bin | distance << 16
    for(int w = threadIdx.x; w < 2 * n_bin; w += COUNT_THREADS) { bins[w] = 0; }
    __syncthreads();
    auto send = [&](int bin, uint32_t distance, unsigned long long fixed) {                  // the QC fail half of a bin, in global memory
        unsigned long long* const words = args.stats + d.stats_offset + static_cast<int64_t>(bin) * STAT_WORDS_PER_BIN;
        atomicAdd(&words[FIELD_COUNT], 1ull);
        if(distance) { atomicAdd(&words[FIELD_DISTANCE], static_cast<unsigned long long>(distance)); }
        if(fixed) {
            const unsigned long long exact = fixed << CONFIDENCE_SHIFT;
            atomicAdd(&words[FIELD_CONFIDENCE_LOW], exact & ((1ull << CONFIDENCE_LIMB_BITS) - 1));
            atomicAdd(&words[FIELD_CONFIDENCE_HIGH], exact >> CONFIDENCE_LIMB_BITS);
        }
    };
    const int64_t n = args.n_read;
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t first = per_block * blockIdx.x;
    const int64_t last = first + per_block < n ? first + per_block : n;
    // Round 5: every WAVE a stream of its own — no workgroup barrier between the set up and the flush.  A wave takes tiles of
    // IN_FLIGHT x 64 records of the workgroup's range in turn (a load instruction = 1 KiB of consecutive records), asks for its
    // next tile before it counts the one it holds, and keeps the failed reads of its own lanes in its own queue (placed by a
    // ballot and a popcount: no atomic on a shared count), which it sends itself — four lanes to an entry — whenever the next
    // tile could overflow it.  (Until round 4 the workgroup shared one queue and settled it between barriers every fourth pass:
    // 16 waves waiting for the slowest four times per 16 Ki records; C3 0.53 ms, 3 TB/s for a pure stream of 16 byte records.)
    constexpr int IN_FLIGHT = 4;
    constexpr int WAVES = COUNT_THREADS / 64;
    constexpr uint32_t WAVE_QUEUE = QUEUE / WAVES;                    // 48 entries a wave
    static_assert(WAVE_QUEUE * WAVES == QUEUE && WAVE_QUEUE >= 32, "the queue is dealt to the waves");
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long* const my_fixed = queue_fixed + wave * WAVE_QUEUE;
    uint32_t* const my_key = queue_key + wave * WAVE_QUEUE;
    uint32_t queued = 0;                                               // wave uniform
    auto drain_wave = [&]() {
        for(uint32_t at = lane; at < 4u * queued; at += 64u) {
            const uint32_t e = at >> 2, field = at & 3u;
            const uint32_t key = my_key[e];
            const unsigned long long exact = my_fixed[e] << CONFIDENCE_SHIFT;
            const unsigned long long value = field == 0 ? 1ull : field == 1 ? static_cast<unsigned long long>(key >> 16)
                                           : field == 2 ? (exact & ((1ull << CONFIDENCE_LIMB_BITS) - 1)) : (exact >> CONFIDENCE_LIMB_BITS);
            if(value) { atomicAdd(&args.stats[d.stats_offset + static_cast<int64_t>(key & 0xffffu) * STAT_WORDS_PER_BIN + field], value); }
        }
        queued = 0;
    };
    typedef uint32_t Quad __attribute__((ext_vector_type(4)));
    const Quad* const records = reinterpret_cast<const Quad*>(args.result);
    constexpr int64_t TILE = IN_FLIGHT * 64;
    auto fetch = [&](int64_t tile_first, Quad (&to)[IN_FLIGHT]) {
#pragma unroll
        for(int k = 0; k < IN_FLIGHT; ++k) {
            const int64_t i = tile_first + k * 64 + lane;
            to[k] = __builtin_nontemporal_load(&records[i < last ? i : last - 1]);       // past the range: its last record again, not counted
        }
    };
    if(first < last) {
        int64_t tile = first + static_cast<int64_t>(wave) * TILE;
        Quad held[IN_FLIGHT];
        if(tile < last) { fetch(tile, held); }
        while(tile < last) {
            const int64_t following = tile + static_cast<int64_t>(WAVES) * TILE;
            Quad next[IN_FLIGHT];
            if(following < last) { fetch(following, next); }
#pragma unroll
            for(int k = 0; k < IN_FLIGHT; ++k) {
                const Quad r = held[k];
                const int32_t index = static_cast<int32_t>(r.x);
                const bool present = tile + k * 64 + lane < last && index > 0;          // unclassified: bin 0 was counted by the stage kernel
                const uint32_t distance = r.y & 0xffu, status = (r.y >> 16) & 0xffu, qcfail = r.y >> 24;
                const double confidence = __hiloint2double(static_cast<int>(r.w), static_cast<int>(r.z));
                const unsigned long long fixed = status == STATUS_PASS ? static_cast<unsigned long long>(__double2ll_rn(confidence * 17592186044416.0)) : 0ull;      // 2^44
                if(present && qcfail == 0) {
                    atomicAdd(&bins[2 * index], 1ull | (static_cast<unsigned long long>(distance) << 32));       // identity numbering (count_results_supports)
                    if(fixed) { atomicAdd(&bins[2 * index + 1], fixed); }
                }
                const unsigned long long failed = __ballot(present && qcfail != 0);
                if(failed) {
                    const uint32_t arriving = static_cast<uint32_t>(__popcll(failed));
                    if(queued + arriving > WAVE_QUEUE) { drain_wave(); }
                    if(arriving > WAVE_QUEUE) {                                                  // more than a queue holds in one instruction: straight to memory
                        if(present && qcfail != 0) { send(index, distance, fixed); }
                    } else {
                        if(present && qcfail != 0) {
                            const uint32_t slot = queued + static_cast<uint32_t>(__popcll(failed & ((1ull << lane) - 1ull)));
                            my_key[slot] = static_cast<uint32_t>(index) | (distance << 16);
                            my_fixed[slot] = fixed;
                        }
                        queued += arriving;
                    }
                }
            }
#pragma unroll
            for(int k = 0; k < IN_FLIGHT; ++k) { held[k] = next[k]; }
            tile = following;
        }
        drain_wave();
    }
    __syncthreads();
    // The bins leave as DENSE atomics into the staging array: 64 consecutive words per wave instruction.  Added straight
    // into the counter buffer (96 bytes from bin to bin) every lane's request went to a row of its own — memory side
    // atomics run 17 times slower that way — and every workgroup sends every bin: that flush was most of the kernel.
    unsigned long long* const staging = d.count_staging;
    for(int w = threadIdx.x + 4; w < 4 * n_bin; w += COUNT_THREADS) {                       // bin 0 is not kept here
        const int bin = w >> 2;
        const int field = w & 3;
        const unsigned long long packed = bins[2 * bin];
        const unsigned long long confidence = bins[2 * bin + 1];
        // 2^-44 units back to the two 26 bit limbs of the 2^-52 fixed point sum (the high limb takes the carry)
        const unsigned long long value = field == 0 ? (packed & 0xffffffffull) : field == 1 ? (packed >> 32)
                                       : field == 2 ? ((confidence & ((1ull << (CONFIDENCE_LIMB_BITS - CONFIDENCE_SHIFT)) - 1)) << CONFIDENCE_SHIFT)
                                                    : (confidence >> (CONFIDENCE_LIMB_BITS - CONFIDENCE_SHIFT));
        if(value) { atomicAdd(&staging[w], value); }
    }
}

// staging -> the pass filter halves of the counter buffer, and the staging array back to zero
__global__ __launch_bounds__(256) void count_results_finish_kernel(StageArgs args) {
    DecoderRef d = *constant_cast(args.decoder);
    const int w = blockIdx.x * 256 + threadIdx.x;
    if(w < 4 || w >= 4 * (d.n_barcode + 1)) { return; }
    const unsigned long long value = d.count_staging[w];
    if(value == 0) { return; }
    d.count_staging[w] = 0;
    static_assert(FIELD_COUNT == 0 && FIELD_DISTANCE == 1 && FIELD_CONFIDENCE_LOW == 2 && FIELD_CONFIDENCE_HIGH == 3, "staging field order");
    atomicAdd(&args.stats[d.stats_offset + static_cast<int64_t>(w >> 2) * STAT_WORDS_PER_BIN + STAT_FIELDS + (w & 3)], value);
}

}  // namespace

// bins that fit the kernel's LDS, numbered the way the compiler numbers them, and no workgroup histogram
bool count_results_supports(const DevDecoder& host_decoder) {
    return !host_decoder.use_lds_histogram && host_decoder.identity_index && host_decoder.n_barcode + 1 <= MAX_DEFERRED_BINS && host_decoder.count_staging != nullptr;
}

int launch_count_results(const StageArgs& args, const DevDecoder& host_decoder, int32_t compute_units, hipStream_t stream) {
    static std::atomic<uint64_t> prepared{0};
    if(const int status = allow_large_lds(&count_results_kernel, prepared)) { return status; }
    const int64_t by_reads = (args.n_read + READS_PER_WORKGROUP - 1) / READS_PER_WORKGROUP;
    const int64_t blocks = std::max<int64_t>(std::min<int64_t>(compute_units, (args.n_read + COUNT_THREADS - 1) / COUNT_THREADS), by_reads);
    const size_t lds_bytes = static_cast<size_t>(host_decoder.n_barcode + 1) * 16 + QUEUE * 12 + 16;
    hipLaunchKernelGGL(count_results_kernel, dim3(static_cast<unsigned>(std::max<int64_t>(blocks, 1))), dim3(COUNT_THREADS), lds_bytes, stream, args);
    if(const hipError_t status = hipGetLastError()) { return static_cast<int>(status); }
    hipLaunchKernelGGL(count_results_finish_kernel, dim3(static_cast<unsigned>((4 * (host_decoder.n_barcode + 1) + 255) / 256)), dim3(256), 0, stream, args);
    return static_cast<int>(hipGetLastError());
}

}  // namespace phq
