// What the kernels of fastq_device.hip and record_device.hip share: where a read's segments sit in the chunks of a round.
#pragma once

#include <hip/hip_runtime.h>

#include "fastq_device.hpp"
#include "job.hpp"

namespace phq {

constexpr int TILE_THREADS = 256;             // 16 bytes per thread: a workgroup indexes 4 KiB of text

struct DecodeArgs {
    FastqPlan plan;
    FastqFeedChunk chunk[FASTQ_MAX_FEED];
    uint32_t n;
    int64_t capacity;
    uint8_t* base_planes;
    uint8_t* quality_planes;
    uint8_t* flag;
    uint32_t* irregular;
    unsigned long long* flagged;
    uint8_t* keep;                  // a plan that filters: 1 = the read is decoded
    uint32_t* keep_block;           // ... kept reads per workgroup of 256 reads, then their exclusive prefix sums
    uint8_t bam_of_ascii[256];
    uint8_t complement_of_bam[16];
};

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

__device__ __forceinline__ void locate(const FastqPlan& plan, const FastqFeedChunk* chunk, uint32_t r, Located& to) {
    to.odd = false;
    to.filtered = false;
    to.dropped = false;
    uint32_t first_name_at = 0;
    uint32_t first_name_size = 0;
    const uint8_t* first_raw = nullptr;
    for(int s = 0; s < plan.n_segment; ++s) {
        const int f = plan.segment_feed[s];
        const uint32_t record = r * static_cast<uint32_t>(plan.feed_resolution[f]) + static_cast<uint32_t>(plan.segment_slot[s]);
        const uint8_t* const raw = chunk[f].raw;
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
        to.odd = to.odd || header_end == header_at || raw[header_at] != '@' || separator_end == sequence_end + 1u || raw[sequence_end + 1u] != '+' ||
                 quality_end - to.quality_at[s] != to.length[s] || raw[header_end - 1u] == '\r' || (to.length[s] > 0 && raw[sequence_end - 1u] == '\r') ||
                 raw[separator_end - 1u] == '\r' || (to.length[s] > 0 && raw[quality_end - 1u] == '\r');
        // the name up to the first blank (kseq), the same in every segment (read.h:253-264)
        uint32_t name_end = header_at + 1u;
        while(name_end < header_end && raw[name_end] != ' ' && raw[name_end] != '\t') { ++name_end; }
        if(s == 0) { first_name_at = header_at + 1u; first_name_size = name_end - header_at - 1u; first_raw = raw; }
        else {
            bool same = name_end - header_at - 1u == first_name_size;
            for(uint32_t k = 0; same && k < first_name_size; ++k) { same = raw[header_at + 1u + k] == first_raw[first_name_at + k]; }
            to.odd = to.odd || !same;
        }
        // the Illumina comment `segment:filtered:control:barcode` of the leading segment (fastq.h:253-285)
        if(s == plan.leader_segment && plan.illumina_comment) {
            uint32_t p = name_end < header_end ? name_end + 1u : header_end;
            while(p < header_end && raw[p] != ':') { ++p; }
            if(p + 1u < header_end) { to.filtered = raw[p + 1u] == 'Y'; }
        }
        if(s >= 1 && static_cast<int32_t>(to.length[s]) < plan.min_length[s]) { to.dropped = true; }      // filter_input, transcode.h:193-200
    }
    to.name_at = first_name_at;
    to.name_size = first_name_size;
    if(plan.filter_incoming_qc_fail && to.filtered) { to.dropped = true; }                                  // transcode.h:205
}

// FILTER: the row of read r is its rank among the kept reads (keep_block holds the exclusive prefix sums of the kept reads
// per workgroup); else row r.  Every lane of the workgroup calls this; `wave_kept` is TILE_THREADS / 64 words of LDS.
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
