// K2, both tiers in ONE kernel: MdDecoder<T>::classify (reference mdd.cpp:37-86) for ACGT codecs of up to 16 nt.
//
// The two kernel form (mdd_exact_kernel, then mdd_scan_kernel over a worklist in HBM) moves every deferred read
// three times: its index out, its rows gathered back in 8 byte pieces (145 B of HBM traffic per deferred read
// instead of 37), its result out.  Here a workgroup keeps the reads its exact match lookup cannot finish in an
// LDS ring — ordinal, plane rows, carried status word: 24 bytes — and whenever the ring holds a full 256 it runs
// the error correcting scan over them with every lane busy.  Nothing but the planes is read, nothing but the
// results written.
//
//   exact match   mdd.cpp:43-46     the observation packed 2 bits per base looked up in an open addressing hash of the
//                                   barcodes, in LDS (mdd.hip has the table's layout)
//   scan          mdd.cpp:47-80     barcodes in index order, the FIRST within every segment's tolerance wins;
//                                   sequence.h:90-98 (distance over the OBSERVED length), :321-332 (quality masking);
//                                   bit-sliced 2-bit bases, two barcodes per word up to 8 nt (mdd_scan.hip)
//
// Every read is written exactly once by the phase that finishes it.  The exact phase writes whole lines with non
// temporal stores (a placeholder for the reads it queues); the scan phase overwrites those placeholders after a
// workgroup barrier, which orders the two stores.
#include "kernel_common.hpp"

namespace phq {
namespace {

// Reads per lane and pass of the exact phase.  The ring's size follows from it, and the ring is most of the kernel's
// LDS, which decides how many workgroups share a CU; this kernel lives on loads and waves in flight.  Per 100 M C2
// reads: 4 reads per lane with the 16 byte hash slots and the 12 word histogram bins of the other kernels (48 KiB,
// 3 workgroups per CU) 2.24 ms; 2 reads (4 workgroups) 1.95 ms; with 8 byte slots and the compact histogram below
// 2 reads (26 KiB, 6 workgroups) 1.92 ms, 3 reads (5 workgroups) 1.67-1.85 ms, 4 reads (38 KiB, 4 workgroups) 1.68 ms.
constexpr int READS = 4;
constexpr int RING = 1280;                           // ring capacity: at most 255 left over + READS x 256 pushed per pass
constexpr int FUSED_LDS_SLOTS = 2048;                // hash table slots kept in LDS (8 bytes each): codecs of up to 512 barcodes
constexpr int FUSED_LDS_WORDS = 512;                 // packed barcode words kept in LDS

__host__ __device__ constexpr uint32_t low_bit(int position) { return 8u * (position >> 2) + 2u + (position & 3); }
__host__ __device__ constexpr uint32_t high_bit(int position) { return (low_bit(position) + 4u) & 31u; }

template <int ROW_WORDS, int SEGMENTS>
__global__ __launch_bounds__(BLOCK) void mdd_fused_kernel(StageArgs args) {
    constexpr int LMAX = 4 * ROW_WORDS;
    constexpr bool PAIRED = ROW_WORDS <= 2;          // up to 8 nt: two barcodes per packed word
    constexpr int stride = 4 * ROW_WORDS;
    __shared__ uint32_t ring_index[RING];
    __shared__ uint32_t ring_state[RING];
    __shared__ uint32_t ring_code[RING][ROW_WORDS];
    __shared__ uint32_t ring_quality[RING][ROW_WORDS];
    __shared__ uint32_t ring_tail;                   // entries ever pushed
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    DecoderRef d = *constant_cast(args.decoder);
    // The dynamic block, sized by the codec (fused_lds_bytes): everything here is kept small because the kernel's LDS
    // decides how many workgroups share a CU, and this kernel lives on waves in flight.
    //   hash table   exact_slots x {key, ordinal + 1}: 8 bytes, a key of up to 16 nt is one word
    //   index        the Classifier index of every barcode (what the 16 byte slots of mdd_exact_kernel carry)
    //   packed       the bit-sliced barcode words of the scan
    //   histogram    count and distance only, per bin and [QC fail | pass filter]: 4 words per bin instead of 12
    const int L = d.nucleotide_cardinality;
    const int n_barcode = d.n_barcode;
    const int n_segment = SEGMENTS > 0 ? SEGMENTS : d.n_segment;
    const int padded_barcodes = (n_barcode + 7) & ~7;
    const int n_word = PAIRED ? padded_barcodes >> 1 : padded_barcodes;
    uint2* const table = reinterpret_cast<uint2*>(lds_raw);
    int32_t* const index_of = reinterpret_cast<int32_t*>(lds_raw + static_cast<size_t>(d.exact_slots) * sizeof(uint2));
    uint32_t* const packed = reinterpret_cast<uint32_t*>(index_of + padded_barcodes);                     // 16 byte aligned: slots and padding are multiples of 8
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(packed + n_word);
    const int compact_bins = d.use_lds_histogram ? n_barcode + 1 : 1;                                      // bin 0 always (kernel_common.hpp)
    for(uint32_t slot = threadIdx.x; slot < d.exact_slots; slot += BLOCK) {
        const uint4 entry = reinterpret_cast<const uint4*>(d.exact_table)[slot];
        table[slot] = make_uint2(entry.x, entry.z);
    }
    for(int b = threadIdx.x; b < padded_barcodes; b += BLOCK) { index_of[b] = b < n_barcode ? d.barcode_index[b] : 0; }
    for(int w = threadIdx.x; w < n_word; w += BLOCK) { packed[w] = d.barcode_packed2[w]; }
    if(threadIdx.x == 0) { ring_tail = 0; }
    for(int w = threadIdx.x; w < STAT_JOB_WORDS + 4 * compact_bins; w += BLOCK) { histogram[w] = 0; }
    __syncthreads();

    // Decoder<T>::classify then Classifier<T>::classify (decoder.h:68-76, classifier.h:78-86) on the compact histogram
    const bool defer_counters = args.defer_counters != 0;
    auto add = [&](int bin, int pf, int field, unsigned long long value) {                                 // field 0 = count, 1 = distance
        if(d.use_lds_histogram || bin == 0) { atomicAdd(&histogram[STAT_JOB_WORDS + 4 * bin + 2 * pf + field], value); }
        else { atomicAdd(&args.stats[d.stats_offset + bin * STAT_WORDS_PER_BIN + pf * STAT_FIELDS + (field ? FIELD_DISTANCE : FIELD_COUNT)], value); }
    };
    auto tail_of = [&](ReadState& r, int bin, int distance) {
        const int pf = r.qcfail ? 0 : 1;
        if(!(defer_counters && bin > 0)) {
            if(bin > 0 && distance) { add(bin, pf, 1, static_cast<unsigned long long>(distance)); }
            add(bin, pf, 0, 1ull);
        }
        if(d.multiplexing) { r.channel = bin > 0 ? index_of[bin - 1] : 0; }
    };
    const uint32_t slot_mask = d.exact_slots - 1u;
    int tolerance[DEV_MAX_SEGMENT];
#pragma unroll
    for(int k = 0; k < DEV_MAX_SEGMENT; ++k) { tolerance[k] = d.distance_tolerance[k]; }
    // segment masks in flag layout (one flag per position at low_bit), wave uniform
    uint32_t segment_flags[DEV_MAX_SEGMENT];
    {
        int s = 0;
#pragma unroll
        for(int k = 0; k < DEV_MAX_SEGMENT; ++k) { segment_flags[k] = 0; }
#pragma unroll
        for(int j = 0; j < LMAX; ++j) {
            if(j < L) {
#pragma unroll
                for(int k = 0; k < DEV_MAX_SEGMENT; ++k) { if(k == s) { segment_flags[k] |= 1u << low_bit(j); } }
                if((d.segment_end_bits[j >> 5] >> (j & 31)) & 1u) { ++s; }
            }
        }
    }

    // the scan over the ring entries [head, head + count): one entry per lane
    auto scan_ring = [&](uint32_t head, uint32_t count) {
        const bool active = threadIdx.x < count;
        const uint32_t e = (head + (active ? threadIdx.x : 0u)) % RING;
        const int64_t i = static_cast<int64_t>(ring_index[e]);
        ReadState r;
        r.qcfail = (ring_state[e] >> 31) != 0;
        r.channel = static_cast<int32_t>(ring_state[e] & 0x7fffffffu);
        uint32_t observed = 0;           // bit-sliced 2-bit codes
        uint32_t always = 0;             // flags that count whatever the barcode: a base outside ACGT, a masked quality
        uint32_t counted = 0;            // flags of the positions inside the observed length of their segment
        bool seen = false;
#pragma unroll
        for(int j = LMAX - 1; j >= 0; --j) {
            if(j < L) {
                const uint32_t c = (ring_code[e][j >> 2] >> (8 * (j & 3))) & 0xffu;
                const uint32_t q = (ring_quality[e][j >> 2] >> (8 * (j & 3))) & 0xffu;
                if((d.segment_end_bits[j >> 5] >> (j & 31)) & 1u) { seen = false; }
                seen = seen || (c != 0) || (q != 0);
                const uint32_t two = (0x30210u >> (2 * (c & 0xfu))) & 3u;
                observed |= ((two & 1u) << low_bit(j)) | ((two >> 1) << high_bit(j));
                if(seen) { counted |= 1u << low_bit(j); }
                if(!strict_code(c) || c > 15u || static_cast<int>(q) < d.quality_masking_threshold) { always |= 1u << low_bit(j); }
            }
        }
        if(PAIRED) {
            observed |= __builtin_amdgcn_alignbit(observed, observed, 16);
            always |= always << 16;
            counted |= counted << 16;
        }
        uint32_t fit = 0xffffffffu;      // (ordinal << 8 | distance) of the first barcode within tolerance
        auto visit = [&](uint32_t word, int first_ordinal) {
            const uint32_t x = observed ^ word;
            const uint32_t flags = ((x | __builtin_amdgcn_alignbit(x, x, 4)) | always) & counted;
#pragma unroll
            for(int h = 0; h < (PAIRED ? 2 : 1); ++h) {
                int distance = 0;
                bool successful = true;
#pragma unroll
                for(int s = 0; s < (SEGMENTS > 0 ? SEGMENTS : DEV_MAX_SEGMENT); ++s) {
                    if(SEGMENTS > 0 || s < n_segment) {
                        const int error = __popc(flags & (segment_flags[s] << (16 * h)));
                        successful = successful && error <= tolerance[s];
                        distance += error;
                    }
                }
                const int b = first_ordinal + h;
                const uint32_t key = (successful && b < n_barcode) ? ((static_cast<uint32_t>(b) << 8) | static_cast<uint32_t>(distance)) : 0xffffffffu;
                fit = key < fit ? key : fit;
            }
        };
        const uint4* const quads = reinterpret_cast<const uint4*>(packed);
        for(int w4 = 0; w4 < n_word / 4; ++w4) {                       // n_word is a multiple of 4: barcodes padded to 8
            const uint4 quad = quads[w4];                              // wave uniform address: a broadcast
            const int first = PAIRED ? 8 * w4 : 4 * w4;
            visit(quad.x, first);
            visit(quad.y, first + (PAIRED ? 2 : 1));
            visit(quad.z, first + (PAIRED ? 4 : 2));
            visit(quad.w, first + (PAIRED ? 6 : 3));
        }
        int bin = 0;
        int distance = 0;
        if(fit != 0xffffffffu) { bin = static_cast<int>(fit >> 8) + 1; distance = static_cast<int>(fit & 0xffu); }
        uint8_t status = STATUS_PASS;
        if(bin == 0) { r.qcfail = true; status = STATUS_UNCLASSIFIED; }          // mdd.cpp:82-84
        if(active) {
            tail_of(r, bin, distance);
            if(args.result) {
                DevResult out;
                out.index = bin > 0 ? index_of[bin - 1] : 0;
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
    };

    const int64_t n = args.n_read;
    const int64_t tiles = (n + BLOCK - 1) / BLOCK;
    uint32_t head = 0;                               // entries consumed; the same in every lane
    for(int64_t tile = static_cast<int64_t>(blockIdx.x) * READS; tile < tiles; tile += static_cast<int64_t>(gridDim.x) * READS) {
        // ---- exact phase: READS tiles ------------------------------------------------------------------------
        int64_t index[READS];
        bool active[READS];
        uint32_t row[READS][ROW_WORDS];
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
            ReadState state;
            state.qcfail = (carried[u] >> 31) != 0;
            state.channel = static_cast<int32_t>(carried[u] & 0x7fffffffu);
            uint32_t key[2] = {0, 0};
            bool valid = active[u];
#pragma unroll
            for(int j = 0; j < 4 * ROW_WORDS; ++j) {
                if(j < L) {
                    const uint32_t c = (row[u][j >> 2] >> (8 * (j & 3))) & 0xffu;
                    valid = valid && strict_code(c) && c < 16u;
                    key[(j >> 4) & 1] |= ((0x30210u >> (2 * (c & 0xfu))) & 3u) << (2 * (j & 15));
                }
            }
            int bin = 0;
            int32_t barcode_index = 0;
            if(valid) {
                uint32_t h = (key[0] * 0x9e3779b1u) ^ (key[1] * 0x85ebca77u);
                h ^= h >> 15;
                for(int probe = 0; probe < d.exact_probes; ++probe) {
                    const uint2 entry = table[(h + probe) & slot_mask];
                    if(entry.y == 0u) { break; }                       // empty slot: not a barcode
                    if(entry.x == key[0]) { bin = static_cast<int>(entry.y); barcode_index = index_of[bin - 1]; break; }     // key[1] == 0 up to 16 nt
                }
            }
            const bool defer = active[u] && bin == 0;
            if(active[u] && !defer) {
                tail_of(state, bin, 0);
                if(args.result) {
                    DevResult out;
                    out.index = barcode_index;
                    out.distance = 0;
                    out.hq_distance = 0;
                    out.status = STATUS_PASS;
                    out.qcfail = state.qcfail ? 1 : 0;
                    out.confidence = 0;
                    stream_result(&args.result[index[u]], out);
                }
                stream_read_state(args, index[u], state);
            } else if(active[u]) {
                stream_placeholder(args, index[u], state);             // overwritten by the scan phase, after a barrier
            }
            count_outgoing(args, histogram, active[u] && !defer, state);
            // queue the deferred reads: a run of ring slots per wave, the quality row read for these lanes only
            const unsigned long long deferred = __ballot(defer);
            if(deferred) {
                const int lane = threadIdx.x & 63;
                uint32_t first = 0;
                if(lane == 0) { first = atomicAdd(&ring_tail, static_cast<uint32_t>(__popcll(deferred))); }
                first = __shfl(first, 0);
                if(defer) {
                    const uint32_t e = (first + static_cast<uint32_t>(__popcll(deferred & ((1ull << lane) - 1ull)))) % RING;
                    const uint32_t* quality_row = reinterpret_cast<const uint32_t*>(args.quality + index[u] * stride);
                    ring_index[e] = static_cast<uint32_t>(index[u]);
                    ring_state[e] = carried[u];
#pragma unroll
                    for(int g = 0; g < ROW_WORDS; ++g) { ring_code[e][g] = row[u][g]; ring_quality[e][g] = quality_row[g]; }
                }
            }
        }
        // ---- scan phase: while the ring holds a full workgroup of reads -----------------------------------------
        __syncthreads();
        uint32_t tail = ring_tail;
        while(tail - head >= BLOCK) {
            scan_ring(head, BLOCK);
            head += BLOCK;
        }
        __syncthreads();                                 // the ring slots just consumed may be pushed into again
    }
    __syncthreads();
    {
        const uint32_t tail = ring_tail;
        if(tail - head > 0) { scan_ring(head, tail - head); }
    }
    __syncthreads();
    // one global atomic per non zero word per workgroup, back in the counter buffer's layout
    outgoing_flush(args, histogram);
    for(int w = threadIdx.x; w < 4 * compact_bins; w += BLOCK) {
        const unsigned long long v = histogram[STAT_JOB_WORDS + w];
        const int bin = w >> 2;
        const int pf = (w >> 1) & 1;
        if(v) { atomicAdd(&args.stats[d.stats_offset + bin * STAT_WORDS_PER_BIN + pf * STAT_FIELDS + ((w & 1) ? FIELD_DISTANCE : FIELD_COUNT)], v); }
    }
}

// bytes of the kernel's dynamic LDS block (the layout at the top of the kernel)
size_t fused_lds_bytes(const DevDecoder& host_decoder) {
    const size_t padded = static_cast<size_t>((host_decoder.n_barcode + 7) & ~7);
    const size_t words = host_decoder.nucleotide_cardinality <= 8 ? padded / 2 : padded;
    const size_t bins = host_decoder.use_lds_histogram ? static_cast<size_t>(host_decoder.n_barcode) + 1 : 1;
    return static_cast<size_t>(host_decoder.exact_slots) * 8 + padded * 4 + words * 4 + (STAT_JOB_WORDS + 4 * bins) * 8;
}

template <int ROW_WORDS>
int launch_rows(const StageArgs& args, int segments, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    if(segments == 1) { hipLaunchKernelGGL((mdd_fused_kernel<ROW_WORDS, 1>), grid, dim3(BLOCK), lds_bytes, stream, args); }
    else if(segments == 2) { hipLaunchKernelGGL((mdd_fused_kernel<ROW_WORDS, 2>), grid, dim3(BLOCK), lds_bytes, stream, args); }
    else { hipLaunchKernelGGL((mdd_fused_kernel<ROW_WORDS, 0>), grid, dim3(BLOCK), lds_bytes, stream, args); }
    return static_cast<int>(hipGetLastError());
}

}  // namespace

// what both tiers of the two kernel form take, with the hash table and the packed words small enough for LDS
bool mdd_fused_supports(const DevDecoder& host_decoder) {
    const int padded = (host_decoder.n_barcode + 7) & ~7;
    const int words = host_decoder.nucleotide_cardinality <= 8 ? padded / 2 : padded;
    return host_decoder.algorithm == 2 && mdd_exact_supports(host_decoder) && host_decoder.all_strict && host_decoder.barcode_packed2 != nullptr &&
           host_decoder.nucleotide_cardinality >= 1 && host_decoder.nucleotide_cardinality <= 16 &&
           host_decoder.exact_slots <= static_cast<uint32_t>(FUSED_LDS_SLOTS) && words <= FUSED_LDS_WORDS;
}

int launch_mdd_fused(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    (void)lds_bytes;                                     // the generic histogram's size: this kernel keeps a compact one
    switch(host_decoder.plane_stride / 4) {
        case 1: return launch_rows<1>(args, host_decoder.n_segment, grid, fused_lds_bytes(host_decoder), stream);
        case 2: return launch_rows<2>(args, host_decoder.n_segment, grid, fused_lds_bytes(host_decoder), stream);
        case 3: return launch_rows<3>(args, host_decoder.n_segment, grid, fused_lds_bytes(host_decoder), stream);
        case 4: return launch_rows<4>(args, host_decoder.n_segment, grid, fused_lds_bytes(host_decoder), stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace phq
