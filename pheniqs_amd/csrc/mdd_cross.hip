// K2, tier 2 for COMBINATORIAL barcode sets: the error correcting scan of MdDecoder<T>::classify (reference
// mdd.cpp:47-80) when the codec holds every combination of its per segment sets (dual / triple index) in
// lexicographic order of the per segment ordinals — what a compiled codec, ordered by key, is.
//
// The reference walks the barcodes in index order and takes the FIRST whose every segment is within its
// tolerance (mdd.cpp:75-79).  With b = (u1, .., uS) in lexicographic order and the test a conjunction of per
// segment tests, that first barcode is (first u1 within tolerance, .., first uS within tolerance): the kernel
// scans U1 + .. + US distinct segment words instead of U1 * .. * US barcodes (192 instead of 9 216 for a
// 96 x 96 dual index; the per barcode scan ran at 0.12 Greads/s there).
//
//   reference sequence.h:90-98      distance_from walks the OBSERVED length of the segment
//   reference sequence.h:321-332    masked_distance_from: quality below the masking threshold always counts
//
// Segment extraction as in pamld_cross.hpp (funnel shifted plane words), mismatch flags as in mdd_scan.hip.
#include "kernel_common.hpp"

namespace phq {
namespace {

constexpr int TABLE_REGISTERS = CROSS_MAX_UNIQUE / 64;

__host__ __device__ constexpr uint32_t low_bit(int position) { return 8u * (position >> 2) + 2u + (position & 3); }
__host__ __device__ constexpr uint32_t high_bit(int position) { return (low_bit(position) + 4u) & 31u; }

template <int G, int S>
__global__ __launch_bounds__(BLOCK) void mdd_cross_kernel(StageArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw);
    DecoderRef d = *constant_cast(args.decoder);
    histogram_clear(d, histogram);
    __syncthreads();
    const StatSink sink{histogram, args.stats, args.defer_counters != 0};

    uint32_t word_table[S][TABLE_REGISTERS];               // lane l of register v holds distinct word 64 v + l
    int first_word[S];
    int length[S];
    uint32_t byte_shift[S];
    uint32_t byte_mask[S][G];
    int unique[S];
    int radix[S];
    int tolerance[S];
    const int row_words = d.plane_stride >> 2;
#pragma unroll
    for(int s = 0; s < S; ++s) {
        const int start = d.cross_start[s];
        length[s] = d.cross_length[s];
        first_word[s] = start >> 2;
        byte_shift[s] = static_cast<uint32_t>(start & 3);
        unique[s] = d.cross_unique[s];
        radix[s] = d.cross_stride[s];
        tolerance[s] = d.distance_tolerance[s];
#pragma unroll
        for(int g = 0; g < G; ++g) {
            const int valid = length[s] - 4 * g;
            byte_mask[s][g] = valid >= 4 ? 0xffffffffu : valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u);
        }
#pragma unroll
        for(int v = 0; v < TABLE_REGISTERS; ++v) {
            const int w = 64 * v + static_cast<int>(threadIdx.x & 63);
            word_table[s][v] = w < unique[s] ? d.cross_words[s * CROSS_MAX_UNIQUE + w] : 0u;
        }
    }

    const uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const int64_t n = static_cast<int64_t>(args.worklist_count[blockIdx.x]);
    const int64_t tiles = (n + BLOCK - 1) / BLOCK;
    const int64_t stride = d.plane_stride;

    for(int64_t tile = 0; tile < tiles; ++tile) {
        const int64_t k = tile * BLOCK + threadIdx.x;
        const bool active = k < n;
        const int64_t i = static_cast<int64_t>(segment[active ? k : n - 1]);       // every lane stays enabled
        ReadState r = load_read_state(args, i);
        const uint32_t* const code_row = reinterpret_cast<const uint32_t*>(args.base + i * stride);
        const uint32_t* const quality_row = reinterpret_cast<const uint32_t*>(args.quality + i * stride);

        int flat = 0;
        int distance = 0;
        bool successful = true;
#pragma unroll
        for(int s = 0; s < S; ++s) {
            uint32_t c[G + 1];
            uint32_t q[G + 1];
#pragma unroll
            for(int g = 0; g <= G; ++g) {
                const int w = min(first_word[s] + g, row_words - 1);              // never past the row
                c[g] = code_row[w];
                q[g] = quality_row[w];
            }
            uint32_t observed = 0;       // bit-sliced 2-bit codes
            uint32_t always = 0;         // flags that count whatever the word: a base outside ACGT, a masked quality
            uint32_t counted = 0;        // flags of the positions inside the observed length of the segment
            bool seen = false;
#pragma unroll
            for(int j = 4 * G - 1; j >= 0; --j) {
                const uint32_t code_word = __builtin_amdgcn_alignbyte(c[(j >> 2) + 1], c[j >> 2], byte_shift[s]) & byte_mask[s][j >> 2];
                const uint32_t quality_word = __builtin_amdgcn_alignbyte(q[(j >> 2) + 1], q[j >> 2], byte_shift[s]) & byte_mask[s][j >> 2];
                const uint32_t code = (code_word >> (8 * (j & 3))) & 0xffu;
                const uint32_t quality = (quality_word >> (8 * (j & 3))) & 0xffu;
                const bool inside = j < length[s];
                seen = seen || (inside && (code != 0 || quality != 0));
                const uint32_t two = (0x30210u >> (2 * (code & 0xfu))) & 3u;
                observed |= ((two & 1u) << low_bit(j)) | ((two >> 1) << high_bit(j));
                if(seen && inside) { counted |= 1u << low_bit(j); }
                if(!strict_code(code) || code > 15u || static_cast<int>(quality) < d.quality_masking_threshold) { always |= 1u << low_bit(j); }
            }
            // first fit over the segment's distinct words = unsigned minimum of (ordinal << 8 | distance)
            uint32_t fit = 0xffffffffu;
#pragma unroll
            for(int v = 0; v < TABLE_REGISTERS; ++v) {
                const int last = min(unique[s], 64 * (v + 1));
                for(int u = 64 * v; u < last; ++u) {
                    const uint32_t x = observed ^ __builtin_amdgcn_readlane(word_table[s][v], u - 64 * v);
                    const uint32_t flags = ((x | __builtin_amdgcn_alignbit(x, x, 4)) | always) & counted;
                    const int error = __popc(flags);
                    const uint32_t key = error <= tolerance[s] ? ((static_cast<uint32_t>(u) << 8) | static_cast<uint32_t>(error)) : 0xffffffffu;
                    fit = key < fit ? key : fit;
                }
            }
            successful = successful && fit != 0xffffffffu;
            flat += static_cast<int>((fit >> 8) & 0xffffffu) * radix[s];
            distance += static_cast<int>(fit & 0xffu);
        }

        int bin = 0;
        if(successful) { bin = flat + 1; } else { distance = 0; }                 // lexicographic numbering (mdd_cross_supports)
        uint8_t status = STATUS_PASS;
        if(bin == 0) { r.qcfail = true; status = STATUS_UNCLASSIFIED; }          // mdd.cpp:82-84
        if(active) {
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

template <int S>
int launch_segments(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    switch(host_decoder.cross_groups) {
        case 1: hipLaunchKernelGGL((mdd_cross_kernel<1, S>), grid, dim3(BLOCK), lds_bytes, stream, args); break;
        case 2: hipLaunchKernelGGL((mdd_cross_kernel<2, S>), grid, dim3(BLOCK), lds_bytes, stream, args); break;
        case 3: hipLaunchKernelGGL((mdd_cross_kernel<3, S>), grid, dim3(BLOCK), lds_bytes, stream, args); break;
        case 4: hipLaunchKernelGGL((mdd_cross_kernel<4, S>), grid, dim3(BLOCK), lds_bytes, stream, args); break;
        default: return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}

}  // namespace

// DecoderSet::build_device_images fills the cross_* members for an ACGT codec that is every combination of its per
// segment sets; cross_identity adds that the barcodes are numbered in lexicographic order of the segment ordinals
bool mdd_cross_supports(const DevDecoder& host_decoder) {
    return host_decoder.algorithm == 2 && host_decoder.cross_segments >= 2 && host_decoder.cross_segments <= CROSS_MAX_SEGMENT &&
           host_decoder.cross_identity && host_decoder.cross_groups >= 1 && host_decoder.cross_groups <= 4 &&
           host_decoder.cross_words != nullptr && host_decoder.nucleotide_cardinality < 256;
}

int launch_mdd_cross(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    if(host_decoder.cross_segments == 2) { return launch_segments<2>(args, host_decoder, grid, lds_bytes, stream); }
    return launch_segments<3>(args, host_decoder, grid, lds_bytes, stream);
}

}  // namespace phq
