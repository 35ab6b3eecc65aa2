// K2: MdDecoder<T>::classify for one decoder, one read per lane, integer only.
//
//   reference mdd.cpp:37-86         exact match first, then the FIRST barcode (index order) whose per segment
//                                   distance is within tolerance; none -> unclassified + QC fail
//   reference sequence.h:90-98      distance_from: walks the OBSERVED length
//   reference sequence.h:321-332    masked_distance_from: quality below the masking threshold is a miss
//
// Observed bases and barcodes are 4-bit BAM codes packed 8 per word; a mismatch mask is XOR + nibble
// fold, a segment distance is a popcount under the segment's nibble mask.
#include "kernel_common.hpp"

namespace phq {
namespace {

// bit 4k of the result is set when nibble k of x is non zero
__device__ __forceinline__ uint32_t nibble_nonzero(uint32_t x) {
    x |= x >> 1;
    x |= x >> 2;
    return x & 0x11111111u;
}

// SEGMENTS: the number of barcode segments when it is 1 or 2 (their masks and tolerances then live in registers and
// the per segment loop unrolls; with a run time segment count every barcode costs indexed scalar loads — measured on
// the 10 + 10 nt decoder of C4: 5.6 ms per 50 M reads), 0 = any.
template <int W, int SEGMENTS>
__global__ __launch_bounds__(BLOCK) void mdd_kernel(StageArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw);
    DecoderRef d = *constant_cast(args.decoder);
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    const PHQ_CONSTANT uint32_t* packed = constant_cast(d.barcode_packed);
    const int n_barcode = d.n_barcode;
    const int n_segment = SEGMENTS > 0 ? SEGMENTS : d.n_segment;
    const int n_word = d.n_word;
    const int L = d.nucleotide_cardinality;
    const int stride = d.plane_stride;
    uint32_t segment_mask[SEGMENTS > 0 ? SEGMENTS : 1][W];
    int tolerance[SEGMENTS > 0 ? SEGMENTS : 1];
    if(SEGMENTS > 0) {
#pragma unroll
        for(int s = 0; s < SEGMENTS; ++s) {
            tolerance[s] = d.distance_tolerance[s];
#pragma unroll
            for(int w = 0; w < W; ++w) { segment_mask[s][w] = d.segment_mask[s][w]; }
        }
    }
    // as the second tier of PHQ_MODE_FAST workgroup g visits only the reads the exact match kernel deferred
    const bool consumer = args.worklist_role == 2;
    const uint32_t* const segment = consumer ? args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride : nullptr;
    const int64_t n = consumer ? static_cast<int64_t>(args.worklist_count[blockIdx.x]) : args.n_read;
    const int64_t tiles = (n + BLOCK - 1) / BLOCK;
    const int64_t tile_first = consumer ? 0 : blockIdx.x;
    const int64_t tile_step = consumer ? 1 : gridDim.x;

    for(int64_t tile = tile_first; tile < tiles; tile += tile_step) {
        const int64_t k = tile * BLOCK + threadIdx.x;
        const bool active = k < n;
        ReadState r{false, 0};
        if(active) {
            const int64_t i = consumer ? static_cast<int64_t>(segment[k]) : k;
            r = load_read_state(args, i);
            const uint32_t* code_row = reinterpret_cast<const uint32_t*>(args.base + i * stride);
            const uint32_t* quality_row = reinterpret_cast<const uint32_t*>(args.quality + i * stride);
            // all loads first (the rows are at most 2 W words; words past the plane stride are not touched)
            uint32_t code_words[2 * W];
            uint32_t quality_words[2 * W];
#pragma unroll
            for(int g = 0; g < 2 * W; ++g) {
                const bool inside = 4 * g < stride;
                code_words[g] = inside ? code_row[inside ? g : 0] : 0u;
                quality_words[g] = inside ? quality_row[inside ? g : 0] : 0u;
            }
            uint32_t observed[W];
            uint32_t masked[W];      // quality below the masking threshold (sequence.h:324-326)
            uint32_t counted[W];     // positions inside the observed length of their segment
#pragma unroll
            for(int w = 0; w < W; ++w) { observed[w] = 0; masked[w] = 0; counted[w] = 0; }
            bool seen = false;
#pragma unroll
            for(int g = 2 * W - 1; g >= 0; --g) {          // groups of 4 positions = one 32-bit load per plane
                if(4 * g < L) {
                    const uint32_t cw = code_words[g];
                    const uint32_t qw = quality_words[g];
#pragma unroll
                    for(int k = 3; k >= 0; --k) {
                        const int j = 4 * g + k;
                        if(j < L) {
                            const uint32_t c = (cw >> (8 * k)) & 0xffu;
                            const uint32_t q = (qw >> (8 * k)) & 0xffu;
                            if((d.segment_end_bits[j >> 5] >> (j & 31)) & 1u) { seen = false; }
                            seen = seen || (c != 0) || (q != 0);
                            const uint32_t flag = 1u << (4 * (j & 7));
                            observed[j >> 3] |= (c & 0xfu) << (4 * (j & 7));
                            if(seen) { counted[j >> 3] |= flag; }
                            if(static_cast<int>(q) < d.quality_masking_threshold) { masked[j >> 3] |= flag; }
                        }
                    }
                }
            }
            bool complete = true;
#pragma unroll
            for(int w = 0; w < W; ++w) { complete = complete && (counted[w] == d.full_mask[w]); }

            int exact = -1;
            int fit = -1;
            int fit_distance = 0;
            for(int b = 0; b < n_barcode; ++b) {
                uint32_t mismatch[W];
                uint32_t any = 0;
#pragma unroll
                for(int w = 0; w < W; ++w) {
                    mismatch[w] = (w < n_word) ? nibble_nonzero(observed[w] ^ packed[b * n_word + w]) : 0u;
                    any |= mismatch[w];
                }
                // mdd.cpp:44-46: the whole observation equals a barcode (keys are full length, barcode.h:46-56)
                if(complete && any == 0 && exact < 0) { exact = b; }
                if(fit < 0) {
                    int distance = 0;
                    bool successful = true;
                    if(SEGMENTS > 0) {
#pragma unroll
                        for(int s = 0; s < SEGMENTS; ++s) {
                            int error = 0;
#pragma unroll
                            for(int w = 0; w < W; ++w) { error += __popc((mismatch[w] | masked[w]) & counted[w] & segment_mask[s][w]); }
                            if(error > tolerance[s]) { successful = false; }
                            distance += error;
                        }
                    } else {
                        for(int s = 0; s < n_segment; ++s) {
                            int error = 0;
#pragma unroll
                            for(int w = 0; w < W; ++w) {
                                error += __popc((mismatch[w] | masked[w]) & counted[w] & d.segment_mask[s][w]);
                            }
                            if(error > d.distance_tolerance[s]) { successful = false; }
                            distance += error;
                        }
                    }
                    if(successful) { fit = b; fit_distance = distance; }
                }
            }
            int bin = 0;
            int distance = 0;
            if(exact >= 0) { bin = exact + 1; }
            else if(fit >= 0) { bin = fit + 1; distance = fit_distance; }
            uint8_t status = STATUS_PASS;
            if(bin == 0) { r.qcfail = true; status = STATUS_UNCLASSIFIED; }      // mdd.cpp:82-84
            decoder_tail(d, sink, r, bin, distance);
            if(args.result) {
                DevResult out;
                out.index = bin > 0 ? d.barcode_index[bin - 1] : 0;
                out.distance = static_cast<uint8_t>(distance > 255 ? 255 : distance);
                out.hq_distance = 0;
                out.status = status;
                out.qcfail = r.qcfail ? 1 : 0;
                out.confidence = 0;
                args.result[i] = out;
            }
            store_read_state(args, i, r);
        }
        count_outgoing(args, histogram, active, r);
    }
    __syncthreads();
    histogram_flush(args, d, histogram);
}

// Tier 1 of PHQ_MODE_FAST for MDD: the reference's own first step, "try a perfect match to the full barcode
// sequence" (mdd.cpp:43-46, an unordered_map keyed by the BAM coded concatenation, mdd.cpp:29-31).  Here: the
// observation packed 2 bits per base is looked up in an open addressing hash table of the barcodes (built by
// the host, load factor <= 1/4).  A hit finishes the read — index, distance 0, counters — having read only its
// bases; everything else (a base outside ACGT, a short observation, no exact match: ~18 % of the C2 reads) goes
// to the workgroup's worklist segment for the scan kernel.  This kernel is a pure stream: ~30 B per read.
constexpr int EXACT_LDS_SLOTS = 1024;                // 16 KiB: codecs of up to 256 barcodes (load factor <= 1/4)

template <int ROW_WORDS>
__global__ __launch_bounds__(BLOCK) void mdd_exact_kernel(StageArgs args) {
    constexpr int READS = 4;                         // reads per lane and pass: independent loads in flight
    __shared__ uint32_t deferred_total;
    __shared__ uint4 lds_table[EXACT_LDS_SLOTS];      // the hash table, when it fits: a probe is an LDS read, not a trip to L2
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw);
    DecoderRef d = *constant_cast(args.decoder);
    if(threadIdx.x == 0) { deferred_total = 0; }
    const bool table_in_lds = d.exact_slots <= static_cast<uint32_t>(EXACT_LDS_SLOTS);
    if(table_in_lds) {
        for(uint32_t slot = threadIdx.x; slot < d.exact_slots; slot += BLOCK) { lds_table[slot] = reinterpret_cast<const uint4*>(d.exact_table)[slot]; }
    }
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const int L = d.nucleotide_cardinality;
    constexpr int stride = 4 * ROW_WORDS;            // = d.plane_stride
    const uint32_t slot_mask = d.exact_slots - 1u;
    const uint4* const table = reinterpret_cast<const uint4*>(d.exact_table);
    const int64_t n = args.n_read;
    const int64_t tiles = (n + BLOCK - 1) / BLOCK;

    for(int64_t tile = static_cast<int64_t>(blockIdx.x) * READS; tile < tiles; tile += static_cast<int64_t>(gridDim.x) * READS) {
        int64_t index[READS];
        bool active[READS];
        ReadState state[READS];
        uint32_t row[READS][ROW_WORDS];
        // every load of the pass is issued before the first use: lanes past the end re-read the last read
        int64_t safe[READS];
#pragma unroll
        for(int u = 0; u < READS; ++u) {
            index[u] = (tile + u) * BLOCK + threadIdx.x;
            active[u] = index[u] < n;
            safe[u] = active[u] ? index[u] : n - 1;
            const uint32_t* code_row = reinterpret_cast<const uint32_t*>(args.base + safe[u] * stride);
#pragma unroll
            for(int g = 0; g < ROW_WORDS; ++g) { row[u][g] = code_row[g]; }
        }
        // the carried state of the four reads under ONE uniform branch (a branch per read would wait per read)
        uint32_t carried[READS];
        if(args.first_stage) {
            if(args.input_flag) {
#pragma unroll
                for(int u = 0; u < READS; ++u) { carried[u] = (args.input_flag[safe[u]] & 1u) << 31; }
            } else {
#pragma unroll
                for(int u = 0; u < READS; ++u) { carried[u] = 0; }
            }
        } else {
#pragma unroll
            for(int u = 0; u < READS; ++u) { carried[u] = args.word[safe[u]]; }
        }
#pragma unroll
        for(int u = 0; u < READS; ++u) {
            state[u].qcfail = (carried[u] >> 31) != 0;
            state[u].channel = static_cast<int32_t>(carried[u] & 0x7fffffffu);
        }
        uint32_t key[READS][2];
        bool valid[READS];
#pragma unroll
        for(int u = 0; u < READS; ++u) {
            key[u][0] = 0; key[u][1] = 0;
            valid[u] = active[u];
#pragma unroll
            for(int j = 0; j < 4 * ROW_WORDS; ++j) {
                if(j < L) {
                    const uint32_t c = (row[u][j >> 2] >> (8 * (j & 3))) & 0xffu;
                    valid[u] = valid[u] && strict_code(c) && c < 16u;
                    key[u][(j >> 4) & 1] |= ((0x30210u >> (2 * (c & 0xfu))) & 3u) << (2 * (j & 15));
                }
            }
        }
        int bin[READS];
        int32_t barcode_index[READS];                     // the slot's fourth word: Classifier index of the barcode
#pragma unroll
        for(int u = 0; u < READS; ++u) {
            bin[u] = 0;
            barcode_index[u] = 0;
            if(valid[u]) {
                uint32_t h = (key[u][0] * 0x9e3779b1u) ^ (key[u][1] * 0x85ebca77u);
                h ^= h >> 15;
                for(int probe = 0; probe < d.exact_probes; ++probe) {
                    const uint32_t slot = (h + probe) & slot_mask;
                    const uint4 entry = table_in_lds ? lds_table[slot] : table[slot];
                    if(entry.z == 0u) { break; }                       // empty slot: not a barcode
                    if(entry.x == key[u][0] && entry.y == key[u][1]) { bin[u] = static_cast<int>(entry.z); barcode_index[u] = static_cast<int32_t>(entry.w); break; }
                }
            }
        }
#pragma unroll
        for(int u = 0; u < READS; ++u) {
            const bool defer = active[u] && bin[u] == 0;
            if(active[u] && !defer) { decoder_tail(d, sink, state[u], bin[u], 0); }
            if(active[u] && !defer) {
                if(args.result) {
                    DevResult out;
                    out.index = barcode_index[u];
                    out.distance = 0;
                    out.hq_distance = 0;
                    out.status = STATUS_PASS;
                    out.qcfail = state[u].qcfail ? 1 : 0;
                    out.confidence = 0;
                    stream_result(&args.result[index[u]], out);
                }
                stream_read_state(args, index[u], state[u]);
            } else if(active[u]) {
                stream_placeholder(args, index[u], state[u]);                      // the scan kernel overwrites it
            }
            const unsigned long long deferred = __ballot(defer);
            if(deferred) {
                const int lane = threadIdx.x & 63;
                uint32_t first = 0;
                if(lane == 0) { first = atomicAdd(&deferred_total, static_cast<uint32_t>(__popcll(deferred))); }
                first = __shfl(first, 0);
                if(defer) {
                    segment[first + __popcll(deferred & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(index[u]);
                }
            }
            count_outgoing(args, histogram, active[u] && !defer, state[u]);
        }
    }
    __syncthreads();
    if(threadIdx.x == 0) { args.worklist_count[blockIdx.x] = deferred_total; }
    histogram_flush(args, d, histogram);
}

// NaiveMolecularDecoder (naive.h:40-45) and the bare Classifier<Barcode> a passthrough decoder is
// (transcode.cpp:77-80): decoded never leaves `unclassified`, so the stage only counts.
__global__ __launch_bounds__(BLOCK) void count_kernel(StageArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw);
    DecoderRef d = *constant_cast(args.decoder);
    histogram_clear(d, histogram);
    __syncthreads();
    const int64_t n = args.n_read;
    unsigned long long pass_total = 0;
    unsigned long long fail_total = 0;
    int64_t done = 0;
    if(args.first_stage && (reinterpret_cast<uintptr_t>(args.word) & 15u) == 0 && (args.input_flag == nullptr || (reinterpret_cast<uintptr_t>(args.input_flag) & 3u) == 0)) {
        // first in the chain (the naive UMI decoder of C4 and C5 is): four reads per lane — their flag bytes as one word, their
        // state words as one 16 byte store — instead of a byte load and a word store per read (round 3: 2.8 -> TB/s of 5 B/read)
        typedef uint32_t Quad __attribute__((ext_vector_type(4)));
        const int64_t quads = n / 4;
        const uint32_t* const flags = reinterpret_cast<const uint32_t*>(args.input_flag);
        Quad* const words = reinterpret_cast<Quad*>(args.word);
        unsigned int fails = 0;
        unsigned int reads = 0;
        for(int64_t q = static_cast<int64_t>(blockIdx.x) * BLOCK + threadIdx.x; q < quads; q += static_cast<int64_t>(gridDim.x) * BLOCK) {
            const uint32_t four = flags ? (flags[q] & 0x01010101u) : 0u;      // transcode.h:207-211: bit 0 = incoming QC fail
            Quad state;
            state.x = (four & 1u) << 31; state.y = ((four >> 8) & 1u) << 31; state.z = ((four >> 16) & 1u) << 31; state.w = ((four >> 24) & 1u) << 31;
            __builtin_nontemporal_store(state, &words[q]);
            fails += static_cast<unsigned int>(__popc(four));
            reads += 4;
        }
        for(int offset = 32; offset > 0; offset >>= 1) {
            fails += __shfl_down(fails, offset);
            reads += __shfl_down(reads, offset);
        }
        if((threadIdx.x & 63) == 0) {
            fail_total += fails;
            pass_total += reads - fails;
            if(args.last_stage) {
                if(fails) { atomicAdd(&histogram[0], static_cast<unsigned long long>(fails)); }
                if(reads - fails) { atomicAdd(&histogram[1], static_cast<unsigned long long>(reads - fails)); }
            }
        }
        done = quads * 4;
    }
    const int64_t tiles = (n - done + BLOCK - 1) / BLOCK;
    for(int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t i = done + tile * BLOCK + threadIdx.x;
        const bool active = i < n;
        ReadState r{false, 0};
        if(active) {
            r = load_read_state(args, i);
            if(args.first_stage) { store_read_state(args, i, r); }
        }
        pass_total += __popcll(__ballot(active && !r.qcfail));
        fail_total += __popcll(__ballot(active && r.qcfail));
        count_outgoing(args, histogram, active, r);
    }
    if((threadIdx.x & 63) == 0) {
        // bin 0, [qc-fail | pass-filter], FIELD_COUNT
        if(d.use_lds_histogram) {
            if(fail_total) { atomicAdd(&histogram[STAT_JOB_WORDS + FIELD_COUNT], fail_total); }
            if(pass_total) { atomicAdd(&histogram[STAT_JOB_WORDS + STAT_FIELDS + FIELD_COUNT], pass_total); }
        } else {
            if(fail_total) { atomicAdd(&args.stats[d.stats_offset + FIELD_COUNT], fail_total); }
            if(pass_total) { atomicAdd(&args.stats[d.stats_offset + STAT_FIELDS + FIELD_COUNT], pass_total); }
        }
    }
    __syncthreads();
    histogram_flush(args, d, histogram);
}

}  // namespace

int launch_mdd(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    const int segments = host_decoder.n_segment == 1 || host_decoder.n_segment == 2 ? host_decoder.n_segment : 0;
#define PHQ_LAUNCH_MDD(W) \
    if(segments == 1) { hipLaunchKernelGGL((mdd_kernel<W, 1>), grid, dim3(BLOCK), lds_bytes, stream, args); } \
    else if(segments == 2) { hipLaunchKernelGGL((mdd_kernel<W, 2>), grid, dim3(BLOCK), lds_bytes, stream, args); } \
    else { hipLaunchKernelGGL((mdd_kernel<W, 0>), grid, dim3(BLOCK), lds_bytes, stream, args); }
    switch(host_decoder.n_word) {
        case 1: PHQ_LAUNCH_MDD(1) break;
        case 2: PHQ_LAUNCH_MDD(2) break;
        case 3: PHQ_LAUNCH_MDD(3) break;
        case 4: PHQ_LAUNCH_MDD(4) break;
        default:
            // barcodes of 33 to 128 nt (the reference's loops take any length, sequence.h:90-98): the same kernel over 8 or 16
            // packed words, the segments at run time
            if(host_decoder.n_word <= 8) { hipLaunchKernelGGL((mdd_kernel<8, 0>), grid, dim3(BLOCK), lds_bytes, stream, args); }
            else if(host_decoder.n_word <= DEV_MAX_WORD) { hipLaunchKernelGGL((mdd_kernel<16, 0>), grid, dim3(BLOCK), lds_bytes, stream, args); }
            else { return static_cast<int>(hipErrorInvalidValue); }
    }
#undef PHQ_LAUNCH_MDD
    return static_cast<int>(hipGetLastError());
}

bool mdd_exact_supports(const DevDecoder& host_decoder) {
    return host_decoder.exact_table != nullptr && host_decoder.exact_slots > 0;
}

int launch_mdd_exact(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    switch(host_decoder.plane_stride / 4) {
        case 1: hipLaunchKernelGGL((mdd_exact_kernel<1>), grid, dim3(BLOCK), lds_bytes, stream, args); break;
        case 2: hipLaunchKernelGGL((mdd_exact_kernel<2>), grid, dim3(BLOCK), lds_bytes, stream, args); break;
        case 3: hipLaunchKernelGGL((mdd_exact_kernel<3>), grid, dim3(BLOCK), lds_bytes, stream, args); break;
        case 4: hipLaunchKernelGGL((mdd_exact_kernel<4>), grid, dim3(BLOCK), lds_bytes, stream, args); break;
        case 5: hipLaunchKernelGGL((mdd_exact_kernel<5>), grid, dim3(BLOCK), lds_bytes, stream, args); break;
        case 6: hipLaunchKernelGGL((mdd_exact_kernel<6>), grid, dim3(BLOCK), lds_bytes, stream, args); break;
        case 7: hipLaunchKernelGGL((mdd_exact_kernel<7>), grid, dim3(BLOCK), lds_bytes, stream, args); break;
        case 8: hipLaunchKernelGGL((mdd_exact_kernel<8>), grid, dim3(BLOCK), lds_bytes, stream, args); break;
        default: return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}

int launch_count(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    (void)host_decoder;
    hipLaunchKernelGGL(count_kernel, grid, dim3(BLOCK), lds_bytes, stream, args);
    return static_cast<int>(hipGetLastError());
}

}  // namespace phq
