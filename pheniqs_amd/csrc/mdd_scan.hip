// K2, tier 2 of PHQ_MODE_FAST: the error correcting scan of MdDecoder<T>::classify (reference mdd.cpp:47-80) over
// the reads the exact match kernel deferred, for ACGT barcodes of up to 16 nt.
//
//   reference mdd.cpp:49-79         barcodes in index order; per segment distance_from / masked_distance_from;
//                                   the FIRST barcode with every segment within tolerance wins (not the closest)
//   reference sequence.h:90-98      distance_from walks the OBSERVED length
//   reference sequence.h:321-332    masked_distance_from: quality below the masking threshold always counts
//
// Same machinery as the product form PAMLD kernel (pamld_lut.hpp): bit-sliced 2-bit bases, two barcodes per word
// up to 8 nt, the packed barcode words in a wave wide register read with v_readlane.  X | rotr(X, 4) leaves one
// mismatch flag per position; a segment's distance is a popcount under three masks (inside the observed length,
// in the segment, in this half of the word); "first fit" is an unsigned minimum over (ordinal << 8 | distance).
// An exact match cannot reach this kernel (tier 1 took it), so there is no exact match logic here.
#include "kernel_common.hpp"

namespace phq {
namespace {

__host__ __device__ constexpr uint32_t low_bit(int position) { return 8u * (position >> 2) + 2u + (position & 3); }
__host__ __device__ constexpr uint32_t high_bit(int position) { return (low_bit(position) + 4u) & 31u; }

// SEGMENTS = barcode segments when 1 or 2 (the per segment tolerance test is then straight line code, no scalar
// loop over DEV_MAX_SEGMENT per barcode: that loop made the kernel scalar-ALU bound), 0 = any number
template <int ROW_WORDS, int TABLE_REGISTERS, int SEGMENTS>
__global__ __launch_bounds__(BLOCK) void mdd_scan_kernel(StageArgs args) {
    constexpr int LMAX = 4 * ROW_WORDS;
    constexpr bool PAIRED = ROW_WORDS <= 2;          // plane stride <= 8 bytes <=> up to 8 nt <=> two barcodes per word
    constexpr int WORDS_PER_GROUP = PAIRED ? 2 : 4;  // packed words per four barcodes (the table is padded to fours)
    constexpr int stride = 4 * ROW_WORDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw);
    DecoderRef d = *constant_cast(args.decoder);
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    const int L = d.nucleotide_cardinality;
    const int n_barcode = d.n_barcode;
    const int n_segment = SEGMENTS > 0 ? SEGMENTS : d.n_segment;
    int tolerance[DEV_MAX_SEGMENT];
#pragma unroll
    for(int k = 0; k < DEV_MAX_SEGMENT; ++k) { tolerance[k] = d.distance_tolerance[k]; }
    const int padded_barcodes = (n_barcode + 3) & ~3;
    const int n_word = PAIRED ? padded_barcodes >> 1 : padded_barcodes;
    uint32_t word_table[TABLE_REGISTERS];            // lane l of register v holds packed word 64 v + l
#pragma unroll
    for(int v = 0; v < TABLE_REGISTERS; ++v) {
        const int w = 64 * v + static_cast<int>(threadIdx.x & 63);
        word_table[v] = w < n_word ? d.barcode_packed2[w] : 0u;
    }
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

    const uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const int64_t n = static_cast<int64_t>(args.worklist_count[blockIdx.x]);
    const int64_t tiles = (n + BLOCK - 1) / BLOCK;

    for(int64_t tile = 0; tile < tiles; ++tile) {
        const int64_t k = tile * BLOCK + threadIdx.x;
        const bool active = k < n;
        const int64_t i = static_cast<int64_t>(segment[active ? k : n - 1]);       // every lane stays enabled
        ReadState r = load_read_state(args, i);
        uint32_t code_words[ROW_WORDS];
        uint32_t quality_words[ROW_WORDS];
        {
            const uint32_t* code_row = reinterpret_cast<const uint32_t*>(args.base + i * stride);
            const uint32_t* quality_row = reinterpret_cast<const uint32_t*>(args.quality + i * stride);
#pragma unroll
            for(int g = 0; g < ROW_WORDS; ++g) { code_words[g] = code_row[g]; quality_words[g] = quality_row[g]; }
        }
        uint32_t observed = 0;           // bit-sliced 2-bit codes
        uint32_t always = 0;             // flags that count whatever the barcode: a base outside ACGT, a masked quality
        uint32_t counted = 0;            // flags of the positions inside the observed length of their segment
        bool seen = false;
#pragma unroll
        for(int j = LMAX - 1; j >= 0; --j) {
            if(j < L) {
                const uint32_t c = (code_words[j >> 2] >> (8 * (j & 3))) & 0xffu;
                const uint32_t q = (quality_words[j >> 2] >> (8 * (j & 3))) & 0xffu;
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
                for(int s = 0; s < (SEGMENTS > 0 ? SEGMENTS : DEV_MAX_SEGMENT); ++s) {     // unrolled: segment_flags stays in registers
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
        constexpr int WORDS_PER_REGISTER = 64;
#pragma unroll
        for(int v = 0; v < TABLE_REGISTERS; ++v) {
            const int last = min(n_word, WORDS_PER_REGISTER * (v + 1));
            for(int w = WORDS_PER_REGISTER * v; w < last; ++w) {
                visit(__builtin_amdgcn_readlane(word_table[v], w - WORDS_PER_REGISTER * v), PAIRED ? 2 * w : w);
            }
        }
        (void)WORDS_PER_GROUP;

        int bin = 0;
        int distance = 0;
        if(fit != 0xffffffffu) { bin = static_cast<int>(fit >> 8) + 1; distance = static_cast<int>(fit & 0xffu); }
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

template <int ROW_WORDS, int SEGMENTS>
int launch_segments(const StageArgs& args, int words, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    if(words <= 64) { hipLaunchKernelGGL((mdd_scan_kernel<ROW_WORDS, 1, SEGMENTS>), grid, dim3(BLOCK), lds_bytes, stream, args); }
    else if(words <= 128) { hipLaunchKernelGGL((mdd_scan_kernel<ROW_WORDS, 2, SEGMENTS>), grid, dim3(BLOCK), lds_bytes, stream, args); }
    else { hipLaunchKernelGGL((mdd_scan_kernel<ROW_WORDS, 4, SEGMENTS>), grid, dim3(BLOCK), lds_bytes, stream, args); }
    return static_cast<int>(hipGetLastError());
}

template <int ROW_WORDS>
int launch_rows(const StageArgs& args, int words, int segments, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    if(segments == 1) { return launch_segments<ROW_WORDS, 1>(args, words, grid, lds_bytes, stream); }
    if(segments == 2) { return launch_segments<ROW_WORDS, 2>(args, words, grid, lds_bytes, stream); }
    return launch_segments<ROW_WORDS, 0>(args, words, grid, lds_bytes, stream);
}

int packed_words(const DevDecoder& host_decoder) {
    const int padded = (host_decoder.n_barcode + 3) & ~3;
    return host_decoder.nucleotide_cardinality <= 8 ? padded / 2 : padded;
}

}  // namespace

bool mdd_scan_supports(const DevDecoder& host_decoder) {
    return host_decoder.all_strict && host_decoder.barcode_packed2 != nullptr && host_decoder.nucleotide_cardinality >= 1 &&
           host_decoder.nucleotide_cardinality <= 16 && packed_words(host_decoder) <= 256 && host_decoder.nucleotide_cardinality < 256;
}

int launch_mdd_scan(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    const int words = packed_words(host_decoder);
    switch(host_decoder.plane_stride / 4) {
        case 1: return launch_rows<1>(args, words, host_decoder.n_segment, grid, lds_bytes, stream);
        case 2: return launch_rows<2>(args, words, host_decoder.n_segment, grid, lds_bytes, stream);
        case 3: return launch_rows<3>(args, words, host_decoder.n_segment, grid, lds_bytes, stream);
        case 4: return launch_rows<4>(args, words, host_decoder.n_segment, grid, lds_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace phq
