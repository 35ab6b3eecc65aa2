// What the kernels of fastq_device.hip and record_device.hip share: where a read's segments sit in the chunks of a round.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>

#include "fastq_device.hpp"
#include "job.hpp"

namespace phq {

constexpr int TILE_THREADS = 256;             // 16 bytes per thread: a workgroup indexes 4 KiB of text

struct DecodeArgs {
    const FastqTables* tables;      // device
    int32_t slot;
    uint32_t n;
    int64_t capacity;
    uint8_t* base_planes;
    uint8_t* quality_planes;
    uint8_t* flag;
    uint32_t* irregular;
    unsigned long long* flagged;
    uint8_t* keep;                  // a plan that filters: 1 = the read is decoded
    uint32_t* keep_block;           // ... kept reads per workgroup's reads, then their exclusive prefix sums
    uint32_t* longest;              // NULL, or per input segment the longest one decoded so far (a running maximum)
};

// The text a wave works on, in LDS.  A wave takes `reads_per_wave` consecutive reads; their records are consecutive bytes of
// every input file, so the wave copies that span of each feed's chunk into its part of the workgroup's LDS with 16 byte
// loads (1 KiB a wave instruction) and every later look at a byte — the name up to its first blank, the names of the other
// segments against it, the bases and qualities a token takes — is an LDS read.  Read where they lie, those bytes came one
// lane at a time, 64 lines a wave instruction, through an L1 the waves of a CU overflow many times over: every byte pulled
// its line from L2 again (decode 1.08 ms, record sizes 1.17 ms, record text 4.2 ms per 1 M reads of 50 + 8 nt).
// A feed whose span does not fit the wave's room stays where it is: bytes(f, x) looks in LDS first.
extern __shared__ __attribute__((aligned(16))) uint8_t staged[];

struct FeedBytes {                      // one feed's text as a wave sees it: byte x of the chunk
    uint32_t base, size, at;
    const uint8_t* raw;
    // (the LDS read is volatile so that the compiler keeps two loads: merged into one load through a selected pointer, the
    // LDS address is cast to a generic one, and this compiler's null check for that cast is an illegal instruction)
    __device__ __forceinline__ uint8_t operator()(uint32_t x) const {
        const uint32_t within = x - base;
        if(within < size) { return const_cast<const volatile uint8_t&>(staged[at + within]); }
        return raw[x];
    }
};
struct WaveInput {
    uint32_t base[FASTQ_MAX_FEED];      // the chunk offset the copy of feed f starts at, a multiple of 16
    uint32_t size[FASTQ_MAX_FEED];      // bytes copied; 0: the feed is read where it lies
    uint32_t at[FASTQ_MAX_FEED];        // where in `staged`
    const FastqFeedChunk* chunk;
    // taken once in front of a loop over a feed's bytes: f is a run time index into three arrays held in registers
    __device__ __forceinline__ FeedBytes feed(int f) const { return FeedBytes{base[f], size[f], at[f], chunk[f].raw}; }
};

// Every lane of the workgroup calls this (it ends in a barrier); the wave's reads are [first_read, first_read + count).
__device__ __forceinline__ void stage_wave_input(const FastqPlan& plan, const FastqFeedChunk* chunk, uint32_t first_read, uint32_t count, uint32_t lds_at, WaveInput& in) {
    in.chunk = chunk;
    uint32_t used = 0;
    const uint32_t lane = threadIdx.x & 63u;
    for(int f = 0; f < plan.n_feed; ++f) {
        in.base[f] = 0; in.size[f] = 0; in.at[f] = 0;
        if(count == 0) { continue; }
        const uint32_t resolution = static_cast<uint32_t>(plan.feed_resolution[f]);
        const uint32_t first_record = first_read * resolution;
        const uint32_t last_record = (first_read + count) * resolution - 1u;
        const uint32_t begin = first_record == 0 ? 0u : chunk[f].newline[4u * first_record - 1u] + 1u;
        const uint32_t end = chunk[f].newline[4u * last_record + 3u] + 1u;
        const uint32_t aligned = begin & ~15u;
        const uint32_t bytes = (end - aligned + 15u) & ~15u;                   // the chunk is zero padded to whole units of 16 bytes
        if(end < begin || used + bytes > static_cast<uint32_t>(plan.input_room)) { continue; }
        for(uint32_t offset = lane * 16u; offset < bytes; offset += 1024u) {
            *reinterpret_cast<uint4*>(&staged[lds_at + used + offset]) = *reinterpret_cast<const uint4*>(chunk[f].raw + aligned + offset);
        }
        in.base[f] = aligned; in.size[f] = bytes; in.at[f] = lds_at + used;
        used += bytes;
    }
    __syncthreads();
}

// which read a lane works on: wave w of workgroup b takes reads [(b WAVES + w) reads_per_wave, + reads_per_wave)
constexpr int TILE_WAVES = TILE_THREADS / 64;
__device__ __forceinline__ uint32_t read_of_lane(const FastqPlan& plan, uint32_t n, uint32_t& wave_first, uint32_t& wave_count, bool& present) {
    const uint32_t per_wave = static_cast<uint32_t>(plan.reads_per_wave);
    const uint32_t lane = threadIdx.x & 63u;
    wave_first = (blockIdx.x * TILE_WAVES + (threadIdx.x >> 6)) * per_wave;
    wave_count = wave_first < n ? min(per_wave, n - wave_first) : 0u;
    present = lane < wave_count;
    return wave_first + lane;
}
// more than 64 KiB of dynamic LDS has to be asked for, once per kernel and device
template <typename Kernel>
inline int allow_lds(Kernel kernel, std::atomic<uint64_t>& done, size_t bytes) {
    if(bytes <= 64u * 1024u) { return 0; }
    int device = 0;
    if(hipGetDevice(&device) != hipSuccess) { return static_cast<int>(hipErrorInvalidDevice); }
    const uint64_t bit = 1ull << (static_cast<unsigned>(device) & 63u);
    if((done.load(std::memory_order_acquire) & bit) == 0) {
        const hipError_t status = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);      // the kernels keep a few words of static LDS
        if(status != hipSuccess) { return static_cast<int>(status); }
        done.fetch_or(bit, std::memory_order_release);
    }
    return 0;
}
inline uint32_t fastq_groups(const FastqPlan& plan, uint32_t n) {
    const uint32_t per_group = static_cast<uint32_t>(TILE_WAVES * plan.reads_per_wave);
    return (n + per_group - 1u) / per_group;
}

// where read r's segments sit in their files' chunks, and what the pull loop would make of it
struct Located {
    uint32_t sequence_at[FASTQ_MAX_FEED];
    uint32_t quality_at[FASTQ_MAX_FEED];
    uint32_t length[FASTQ_MAX_FEED];
    uint32_t name_at;               // the identifier of the first segment (every segment's is the same), in its feed's chunk
    uint32_t name_size;
    bool odd;                       // framing or name sync out of the ordinary: the host reader takes the run
    bool filtered;                  // the Illumina filter flag of the leading segment
    bool dropped;                   // the pre-filters of the pull loop drop the read (transcode.h:193-205)
};

__device__ __forceinline__ void locate(const FastqPlan& plan, const FastqFeedChunk* chunk, const WaveInput& bytes, uint32_t r, Located& to) {
    to.odd = false;
    to.filtered = false;
    to.dropped = false;
    uint32_t first_name_at = 0;
    uint32_t first_name_size = 0;
    FeedBytes first_text{0, 0, 0, nullptr};
    for(int s = 0; s < plan.n_segment; ++s) {
        const int f = plan.segment_feed[s];
        const FeedBytes text = bytes.feed(f);
        const uint32_t record = r * static_cast<uint32_t>(plan.feed_resolution[f]) + static_cast<uint32_t>(plan.segment_slot[s]);
        const uint32_t* const newline = chunk[f].newline + 4u * record;
        const uint32_t header_at = record == 0 ? 0u : newline[-1] + 1u;
        const uint32_t header_end = newline[0];
        const uint32_t sequence_end = newline[1];
        const uint32_t separator_end = newline[2];
        const uint32_t quality_end = newline[3];
        to.sequence_at[s] = header_end + 1u;
        to.quality_at[s] = separator_end + 1u;
        to.length[s] = sequence_end - to.sequence_at[s];
        // the framing of a record (fastq.h:427-440), nothing the host reader would have to tolerate or report
        to.odd = to.odd || header_end == header_at || text(header_at) != '@' || separator_end == sequence_end + 1u || text(sequence_end + 1u) != '+' ||
                 quality_end - to.quality_at[s] != to.length[s] || text(header_end - 1u) == '\r' || (to.length[s] > 0 && text(sequence_end - 1u) == '\r') ||
                 text(separator_end - 1u) == '\r' || (to.length[s] > 0 && text(quality_end - 1u) == '\r');
        // the name up to the first blank (kseq), the same in every segment (read.h:253-264)
        uint32_t name_end = header_at + 1u;
        while(name_end < header_end) {
            const uint8_t c = text(name_end);
            if(c == ' ' || c == '\t') { break; }
            ++name_end;
        }
        if(s == 0) { first_name_at = header_at + 1u; first_name_size = name_end - header_at - 1u; first_text = text; }
        else {
            bool same = name_end - header_at - 1u == first_name_size;
            for(uint32_t k = 0; same && k < first_name_size; ++k) { same = text(header_at + 1u + k) == first_text(first_name_at + k); }
            to.odd = to.odd || !same;
        }
        // the Illumina comment `segment:filtered:control:barcode` of the leading segment (fastq.h:253-285)
        if(s == plan.leader_segment && plan.illumina_comment) {
            uint32_t p = name_end < header_end ? name_end + 1u : header_end;
            while(p < header_end && text(p) != ':') { ++p; }
            if(p + 1u < header_end) { to.filtered = text(p + 1u) == 'Y'; }
        }
        if(s >= 1 && static_cast<int32_t>(to.length[s]) < plan.min_length[s]) { to.dropped = true; }      // filter_input, transcode.h:193-200
    }
    to.name_at = first_name_at;
    to.name_size = first_name_size;
    if(plan.filter_incoming_qc_fail && to.filtered) { to.dropped = true; }                                  // transcode.h:205
}

// FILTER: the row of read r is its rank among the kept reads (keep_block holds the exclusive prefix sums of the kept reads
// per workgroup's reads); else row r.  Every lane of the workgroup calls this; `wave_kept` is TILE_WAVES words of LDS.
template <bool FILTER>
__device__ __forceinline__ uint32_t row_of(uint32_t r, bool present, const uint8_t* keep, const uint32_t* keep_block, uint32_t* wave_kept, bool& wanted) {
    wanted = present;
    if(!FILTER) { return r; }
    wanted = present && keep[r] != 0;
    const unsigned long long kept = __ballot(wanted);
    if((threadIdx.x & 63) == 0) { wave_kept[threadIdx.x >> 6] = static_cast<uint32_t>(__popcll(kept)); }
    __syncthreads();
    uint32_t row = keep_block[blockIdx.x] + static_cast<uint32_t>(__popcll(kept & ((1ull << (threadIdx.x & 63)) - 1ull)));
    for(uint32_t w = 0; w < (threadIdx.x >> 6); ++w) { row += wave_kept[w]; }
    return row;
}

// Token::absolute_start / absolute_end (transform.h:65-80)
__device__ __forceinline__ void token_range(int32_t token_start, int32_t token_end, int32_t end_terminated, int32_t length, int32_t& start, int32_t& end) {
    start = token_start;
    if(start < 0) { start = length + start; start = start < 0 ? 0 : start; }
    else if(start > length) { start = 0; }
    end = length;
    if(end_terminated) {
        if(token_end < 0) { end = length + token_end; end = end < 0 ? 0 : end; }
        else { end = token_end > length ? length : token_end; }
    }
}

}  // namespace phq
