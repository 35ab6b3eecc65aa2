// FASTQ text -> decoder planes on the device: see fastq_device.hpp.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "fastq_device_inner.hpp"

namespace phq {

namespace {

// bit 7 of every byte of x that is '\n' (exact per byte: no borrow runs from one byte into the next)
__device__ __forceinline__ uint32_t newline_bytes(uint32_t x) {
    const uint32_t y = x ^ 0x0a0a0a0au;
    return ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y) & 0x80808080u;
}

__device__ __forceinline__ uint4 unit_of(const uint8_t* raw, uint32_t bytes, uint32_t unit) {
    // the buffer is zero padded to whole units of 16 bytes: a zero is no newline
    return static_cast<uint64_t>(unit) * 16u < bytes ? reinterpret_cast<const uint4*>(raw)[unit] : make_uint4(0, 0, 0, 0);
}

__global__ __launch_bounds__(TILE_THREADS) void fastq_count_kernel(const uint8_t* raw, uint32_t bytes, uint32_t* block_count) {
    __shared__ uint32_t sum;
    if(threadIdx.x == 0) { sum = 0; }
    __syncthreads();
    const uint4 v = unit_of(raw, bytes, blockIdx.x * TILE_THREADS + threadIdx.x);
    uint32_t mine = __popc(newline_bytes(v.x)) + __popc(newline_bytes(v.y)) + __popc(newline_bytes(v.z)) + __popc(newline_bytes(v.w));
    for(int step = 32; step > 0; step >>= 1) { mine += __shfl_down(mine, step); }
    if((threadIdx.x & 63) == 0 && mine) { atomicAdd(&sum, mine); }
    __syncthreads();
    if(threadIdx.x == 0) { block_count[blockIdx.x] = sum; }
}

// exclusive prefix sums of the workgroups' counts, in place; block_count[n_block] = *total = the number of newlines
__global__ __launch_bounds__(1024) void fastq_scan_kernel(uint32_t* block_count, uint32_t n_block, uint32_t* total) {
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t carried;
    if(threadIdx.x == 0) { carried = 0; }
    __syncthreads();
    for(uint32_t first = 0; first < n_block; first += 1024) {
        const uint32_t at = first + threadIdx.x;
        const uint32_t own = at < n_block ? block_count[at] : 0u;
        uint32_t inclusive = own;
        for(int step = 1; step < 64; step <<= 1) {
            const uint32_t below = __shfl_up(inclusive, step);
            if(static_cast<int>(threadIdx.x & 63) >= step) { inclusive += below; }
        }
        if((threadIdx.x & 63) == 63) { wave_sum[threadIdx.x >> 6] = inclusive; }
        __syncthreads();
        uint32_t before = carried;
        for(uint32_t w = 0; w < (threadIdx.x >> 6); ++w) { before += wave_sum[w]; }
        if(at < n_block) { block_count[at] = before + inclusive - own; }
        __syncthreads();
        if(threadIdx.x == 1023) { carried = before + inclusive; }
        __syncthreads();
    }
    if(threadIdx.x == 0) { block_count[n_block] = carried; *total = carried; }
}

__global__ __launch_bounds__(TILE_THREADS) void fastq_scatter_kernel(const uint8_t* raw, uint32_t bytes, const uint32_t* block_offset, uint32_t* newline, uint32_t capacity) {
    __shared__ uint32_t wave_sum[TILE_THREADS / 64];
    const uint32_t unit = blockIdx.x * TILE_THREADS + threadIdx.x;
    const uint4 v = unit_of(raw, bytes, unit);
    const uint32_t word[4] = {newline_bytes(v.x), newline_bytes(v.y), newline_bytes(v.z), newline_bytes(v.w)};
    const uint32_t own = __popc(word[0]) + __popc(word[1]) + __popc(word[2]) + __popc(word[3]);
    uint32_t inclusive = own;
    for(int step = 1; step < 64; step <<= 1) {
        const uint32_t below = __shfl_up(inclusive, step);
        if(static_cast<int>(threadIdx.x & 63) >= step) { inclusive += below; }
    }
    if((threadIdx.x & 63) == 63) { wave_sum[threadIdx.x >> 6] = inclusive; }
    __syncthreads();
    uint32_t at = block_offset[blockIdx.x] + inclusive - own;
    for(uint32_t w = 0; w < (threadIdx.x >> 6); ++w) { at += wave_sum[w]; }
    if(own == 0) { return; }
#pragma unroll
    for(int k = 0; k < 4; ++k) {
        uint32_t bits = word[k];
        while(bits) {
            const uint32_t bit = static_cast<uint32_t>(__ffs(static_cast<int>(bits))) - 1u;      // 7, 15, 23 or 31
            bits &= bits - 1u;
            if(at < capacity) { newline[at] = unit * 16u + static_cast<uint32_t>(k) * 4u + (bit >> 3); }
            ++at;
        }
    }
}

struct PlanArgs {
    FastqFeedChunk chunk[FASTQ_MAX_FEED];
    int32_t resolution[FASTQ_MAX_FEED];
    int32_t n_feed;
    uint32_t capacity;
    FastqRound* round;
};

__global__ void fastq_plan_kernel(PlanArgs args) {
    if(threadIdx.x != 0 || blockIdx.x != 0) { return; }
    uint32_t records = args.capacity;
    uint32_t irregular = 0;
    for(int f = 0; f < args.n_feed; ++f) {
        const uint32_t lines = *args.chunk[f].total;
        args.round->lines[f] = lines;
        if(lines > args.chunk[f].newline_capacity) { irregular = 1; }
        records = min(records, lines / (4u * static_cast<uint32_t>(args.resolution[f])));      // whole reads: every segment of an interleaved feed
    }
    if(irregular) { records = 0; }
    for(int f = 0; f < args.n_feed; ++f) {
        args.round->consumed[f] = records > 0 ? args.chunk[f].newline[4u * records * static_cast<uint32_t>(args.resolution[f]) - 1u] + 1u : 0u;
    }
    args.round->records = records;
    args.round->irregular = irregular;
}

__device__ __forceinline__ void count_flagged(const DecodeArgs& args, bool filtered) {
    const unsigned long long marked = __ballot(filtered);
    if(marked && (threadIdx.x & 63) == static_cast<unsigned>(__ffsll(static_cast<long long>(marked)) - 1)) { atomicAdd(args.flagged, static_cast<unsigned long long>(__popcll(marked))); }
}

// a plan that filters, first pass: which reads are kept, how many per workgroup; framing and the flag count as the decode does
__global__ __launch_bounds__(TILE_THREADS) void fastq_keep_kernel(DecodeArgs args) {
    __shared__ uint32_t kept_here;
    if(threadIdx.x == 0) { kept_here = 0; }
    const FastqPlan& plan = args.tables->plan;
    const FastqFeedChunk* const chunk = args.tables->chunk[args.slot];
    uint32_t wave_first, wave_count;
    bool present;
    const uint32_t r = read_of_lane(plan, args.n, wave_first, wave_count, present);
    WaveInput bytes;
    stage_wave_input(plan, chunk, wave_first, wave_count, (threadIdx.x >> 6) * static_cast<uint32_t>(plan.input_room), bytes);
    bool keep = false;
    bool filtered = false;
    if(present) {
        Located at;
        locate(plan, chunk, bytes, r, at);
        if(at.odd) { atomicOr(args.irregular, 1u); }
        keep = !at.dropped;
        filtered = at.filtered;
        args.keep[r] = keep ? 1 : 0;
    }
    count_flagged(args, filtered);
    const unsigned long long kept = __ballot(keep);
    if((threadIdx.x & 63) == 0 && kept) { atomicAdd(&kept_here, static_cast<uint32_t>(__popcll(kept))); }
    __syncthreads();
    if(threadIdx.x == 0) { args.keep_block[blockIdx.x] = kept_here; }
}

// FILTER: rows by rank among the kept reads (keep_block holds the exclusive prefix sums by now); else row r for read r
template <bool FILTER>
__global__ __launch_bounds__(TILE_THREADS) void fastq_decode_kernel(DecodeArgs args) {
    const FastqPlan& plan = args.tables->plan;
    const FastqFeedChunk* const chunk = args.tables->chunk[args.slot];
    __shared__ uint32_t wave_kept[TILE_WAVES];
    uint32_t wave_first, wave_count;
    bool present;
    const uint32_t r = read_of_lane(plan, args.n, wave_first, wave_count, present);
    WaveInput bytes;
    stage_wave_input(plan, chunk, wave_first, wave_count, (threadIdx.x >> 6) * static_cast<uint32_t>(plan.input_room), bytes);
    bool wanted;
    const uint32_t row = row_of<FILTER>(r, present, args.keep, args.keep_block, wave_kept, wanted);
    bool filtered = false;
    Located at;
    if(wanted) { locate(plan, chunk, bytes, r, at); }
    if(args.longest != nullptr) {
        // what sizes the quality control accumulators' cycles (record_device.hpp): the longest segment so far
        for(int s = 0; s < plan.n_segment; ++s) {
            uint32_t longest = wanted && !at.odd ? at.length[s] : 0u;
            for(int step = 32; step > 0; step >>= 1) { longest = max(longest, static_cast<uint32_t>(__shfl_xor(static_cast<int>(longest), step))); }
            if((threadIdx.x & 63) == 0 && longest > 0) { atomicMax(&args.longest[s], longest); }
        }
    }
    if(wanted) {
        filtered = at.filtered;
        if(!FILTER && at.odd) { atomicOr(args.irregular, 1u); }
        // Rule::apply of every decoder into its plane (transform.h:142-169), zero filled past what the tokens bring
        for(int k = 0; k < plan.n_decoder; ++k) {
            const FastqDecoderPlan& d = plan.decoder[k];
            if(d.plane_stride == 0) { continue; }
            uint8_t* const row_code = args.base_planes + args.capacity * d.plane_prefix + static_cast<int64_t>(row) * d.plane_stride;
            uint8_t* const row_quality = args.quality_planes + args.capacity * d.plane_prefix + static_cast<int64_t>(row) * d.plane_stride;
            for(int w = 0; w < d.plane_stride / 4; ++w) {
                reinterpret_cast<uint32_t*>(row_code)[w] = 0u;
                reinterpret_cast<uint32_t*>(row_quality)[w] = 0u;
            }
            int32_t filled[DEV_MAX_SEGMENT] = {0, 0, 0, 0, 0, 0, 0, 0};
            for(int t = d.first_token; t < d.first_token + d.n_token; ++t) {
                const FastqToken& token = plan.token[t];
                const int32_t from_length = static_cast<int32_t>(at.length[token.segment]);
                int32_t start, end;
                token_range(token.start, token.end, token.end_terminated, from_length, start, end);
                const int32_t size = end - start;
                if(size <= 0 || at.odd) { continue; }
                int32_t& done = filled[token.output_segment];
                const int32_t room = token.output_room - done;
                const int32_t take = size < room ? size : room;
                const FeedBytes text = bytes.feed(plan.segment_feed[token.segment]);
                for(int32_t j = 0; j < take; ++j) {
                    const int32_t from = token.reverse_complement ? end - j - 1 : start + j;
                    uint8_t code = args.tables->bam_of_ascii[text(at.sequence_at[token.segment] + static_cast<uint32_t>(from))];
                    if(token.reverse_complement) { code = args.tables->complement_of_bam[code & 15u]; }
                    row_code[token.output_offset + done + j] = code;
                    row_quality[token.output_offset + done + j] = static_cast<uint8_t>(text(at.quality_at[token.segment] + static_cast<uint32_t>(from)) - plan.phred_offset);
                }
                done += take > 0 ? take : 0;
            }
        }
        args.flag[row] = filtered ? 1 : 0;
    }
    if(!FILTER) { count_flagged(args, filtered); }
}

}  // namespace

// A wave's reads by what their text takes in LDS: 64 reads of 190 bytes are 12 KiB; reads of 2 x 150 nt (700 bytes) go 16 to a wave
void fastq_plan_waves(FastqPlan& plan, double bytes_per_read) {
    constexpr double ROOM = 20.0 * 1024.0;
    int32_t reads = 64;
    if(const char* text = std::getenv("PHQ_FASTQ_READS_PER_WAVE")) {      // sweeps; a power of two from 8 to 64: the per workgroup scratch is sized for at least 32 reads a workgroup
        const int asked = std::max(8, std::min(64, std::atoi(text)));
        reads = 8;
        while(reads * 2 <= asked) { reads *= 2; }
    }
    while(reads > 8 && static_cast<double>(reads) * bytes_per_read * 1.25 + 32.0 * plan.n_feed > ROOM) { reads /= 2; }
    plan.reads_per_wave = reads;
    const double room = std::min(ROOM, static_cast<double>(reads) * bytes_per_read * 1.25 + 32.0 * plan.n_feed);
    plan.input_room = (static_cast<int32_t>(room) + 15) & ~15;
}

int fastq_index(const FastqFeedChunk& chunk, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t units = (chunk.bytes + 15u) / 16u;
    const uint32_t blocks = (units + TILE_THREADS - 1) / TILE_THREADS;
    if(blocks > 0) { hipLaunchKernelGGL(fastq_count_kernel, dim3(blocks), dim3(TILE_THREADS), 0, s, chunk.raw, chunk.bytes, chunk.block_count); }
    hipLaunchKernelGGL(fastq_scan_kernel, dim3(1), dim3(1024), 0, s, chunk.block_count, blocks, chunk.total);
    if(blocks > 0) { hipLaunchKernelGGL(fastq_scatter_kernel, dim3(blocks), dim3(TILE_THREADS), 0, s, chunk.raw, chunk.bytes, chunk.block_count, chunk.newline, chunk.newline_capacity); }
    return static_cast<int>(hipGetLastError());
}

int fastq_plan_round(const FastqFeedChunk* chunk, int32_t n_feed, const int32_t* resolution, uint32_t capacity, FastqRound* round_on_device, void* stream) {
    PlanArgs args;
    for(int f = 0; f < n_feed && f < FASTQ_MAX_FEED; ++f) { args.chunk[f] = chunk[f]; args.resolution[f] = resolution[f] > 0 ? resolution[f] : 1; }
    args.n_feed = n_feed;
    args.capacity = capacity;
    args.round = round_on_device;
    hipLaunchKernelGGL(fastq_plan_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), args);
    return static_cast<int>(hipGetLastError());
}

void fastq_fill_tables(FastqTables& tables) {
    for(int c = 0; c < 256; ++c) { tables.bam_of_ascii[c] = ascii_to_bam(static_cast<uint8_t>(c)); }
    for(int c = 0; c < 16; ++c) { tables.complement_of_bam[c] = reverse_complement_bam(static_cast<uint8_t>(c)); }
}

int fastq_decode(const FastqTables* on_device, const FastqPlan& plan, int32_t slot, uint32_t n, int64_t capacity, uint8_t* base_planes, uint8_t* quality_planes,
                 uint8_t* flag, uint32_t* irregular, unsigned long long* flagged, uint8_t* keep, uint32_t* keep_block, uint32_t* kept, uint32_t* longest, void* stream) {
    if(n == 0) { return 0; }
    DecodeArgs call;
    call.longest = longest;
    call.tables = on_device;
    call.slot = slot;
    call.n = n;
    call.capacity = capacity;
    call.base_planes = base_planes;
    call.quality_planes = quality_planes;
    call.flag = flag;
    call.irregular = irregular;
    call.flagged = flagged;
    call.keep = keep;
    call.keep_block = keep_block;
    // 8, 16, 32 or 64 reads a wave: the callers size keep_block and the record writers' per workgroup arrays for 32 reads a workgroup at least
    if(plan.reads_per_wave < 8 || plan.reads_per_wave > 64 || (plan.reads_per_wave & (plan.reads_per_wave - 1)) != 0 || plan.input_room < 0 || plan.input_room % 16 != 0) { return static_cast<int>(hipErrorInvalidValue); }
    const uint32_t blocks = fastq_groups(plan, n);
    const size_t lds = static_cast<size_t>(TILE_WAVES) * static_cast<size_t>(plan.input_room);
    hipStream_t s = static_cast<hipStream_t>(stream);
    static std::atomic<uint64_t> prepared[3];
    if(const int status = allow_lds(&fastq_keep_kernel, prepared[0], lds)) { return status; }
    if(const int status = allow_lds(&fastq_decode_kernel<true>, prepared[1], lds)) { return status; }
    if(const int status = allow_lds(&fastq_decode_kernel<false>, prepared[2], lds)) { return status; }
    if(plan.filtering) {
        if(keep == nullptr || keep_block == nullptr || kept == nullptr) { return static_cast<int>(hipErrorInvalidValue); }
        hipLaunchKernelGGL(fastq_keep_kernel, dim3(blocks), dim3(TILE_THREADS), lds, s, call);
        hipLaunchKernelGGL(fastq_scan_kernel, dim3(1), dim3(1024), 0, s, keep_block, blocks, kept);
        hipLaunchKernelGGL(fastq_decode_kernel<true>, dim3(blocks), dim3(TILE_THREADS), lds, s, call);
    } else {
        hipLaunchKernelGGL(fastq_decode_kernel<false>, dim3(blocks), dim3(TILE_THREADS), lds, s, call);
    }
    return static_cast<int>(hipGetLastError());
}

}  // namespace phq
