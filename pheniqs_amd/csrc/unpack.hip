// One byte per nucleotide across the link (phq_classify_host_packed): the planes a host batch is uploaded as, and the
// kernels that turn them back into the observation planes every stage kernel reads.
//
//   reference fastq.h:55-78   FastqRecord::decode: the BAM code and the Phred value of a nucleotide, a byte each
//
// A nucleotide is A, C, G or T with a quality of at most 62 in all but a few cases: byte = ordinal << 6 | quality.  The
// others — N and the other ambiguity codes, qualities of 63 and more — are marked (0x3F) and travel beside the planes as a
// list of (read, column, code, quality); (0, 0), what lies past the observed length of a short read and in the columns a
// plane's stride adds to its barcode, is 0x7F.
#include <hip/hip_runtime.h>

#include "unpack.hpp"

namespace phq {

namespace {

constexpr int UNPACK_THREADS = 256;

// 16 packed bytes a lane in, 16 codes and 16 qualities out
__global__ __launch_bounds__(UNPACK_THREADS) void unpack_planes_kernel(const uint4* packed, uint4* base, uint4* quality, size_t units) {
    const size_t unit = static_cast<size_t>(blockIdx.x) * UNPACK_THREADS + threadIdx.x;
    if(unit >= units) { return; }
    const uint4 in = packed[unit];
    auto code_of = [](uint32_t word) {
        const uint32_t ordinal = (word >> 6) & 0x03030303u;
        const uint32_t low = ordinal & 0x01010101u, high = (ordinal >> 1) & 0x01010101u;
        const uint32_t first = 0x01010101u + low;                 // 1 or 2 per byte
        return first + 3u * (first & (high * 0xffu));              // x 4 where the high bit is set: 1, 2, 4, 8
    };
    auto quality_of = [](uint32_t word) { return word & 0x3f3f3f3fu; };
    // 0x00 in every byte that is PACKED_NOTHING, 0xff elsewhere (exact per byte: no borrow runs from one byte into the next)
    auto something = [](uint32_t word) {
        const uint32_t z = word ^ 0x7f7f7f7fu;
        const uint32_t hit = ~(((z & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z) & 0x80808080u;
        return ~((hit >> 7) * 0xffu);
    };
    const uint32_t keep[4] = {something(in.x), something(in.y), something(in.z), something(in.w)};
    base[unit] = make_uint4(code_of(in.x) & keep[0], code_of(in.y) & keep[1], code_of(in.z) & keep[2], code_of(in.w) & keep[3]);
    quality[unit] = make_uint4(quality_of(in.x) & keep[0], quality_of(in.y) & keep[1], quality_of(in.z) & keep[2], quality_of(in.w) & keep[3]);
}

struct PatchArgs {
    const uint64_t* escapes;
    int64_t n;
    int64_t first_read;
    int64_t chunk_capacity;
    uint8_t* base;
    uint8_t* quality;
    int32_t n_decoder;
    int32_t prefix[UNPACK_MAX_DECODER];
    int32_t stride[UNPACK_MAX_DECODER];
};

__global__ __launch_bounds__(UNPACK_THREADS) void patch_escapes_kernel(PatchArgs args) {
    const int64_t at = static_cast<int64_t>(blockIdx.x) * UNPACK_THREADS + threadIdx.x;
    if(at >= args.n) { return; }
    const uint64_t entry = args.escapes[at];
    const int64_t k = static_cast<int64_t>(entry >> 32) - args.first_read;
    const int32_t column = static_cast<int32_t>((entry >> 16) & 0xffffu);
    for(int d = 0; d < args.n_decoder; ++d) {
        if(column >= args.prefix[d] && column < args.prefix[d] + args.stride[d]) {
            const int64_t offset = args.chunk_capacity * args.prefix[d] + k * args.stride[d] + (column - args.prefix[d]);
            args.base[offset] = static_cast<uint8_t>((entry >> 8) & 0xffu);
            args.quality[offset] = static_cast<uint8_t>(entry & 0xffu);
        }
    }
}

}  // namespace

int unpack_planes(const uint8_t* packed, uint8_t* base, uint8_t* quality, size_t bytes, void* stream) {
    const size_t units = (bytes + 15) / 16;                        // the planes are allocated in whole units
    if(units == 0) { return 0; }
    const unsigned blocks = static_cast<unsigned>((units + UNPACK_THREADS - 1) / UNPACK_THREADS);
    hipLaunchKernelGGL(unpack_planes_kernel, dim3(blocks), dim3(UNPACK_THREADS), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint4*>(packed), reinterpret_cast<uint4*>(base), reinterpret_cast<uint4*>(quality), units);
    return static_cast<int>(hipGetLastError());
}

int patch_escapes(const uint64_t* escapes, int64_t n, int64_t first_read, int64_t chunk_capacity, int32_t n_decoder, const int32_t* prefix, const int32_t* stride,
                  uint8_t* base, uint8_t* quality, void* stream) {
    if(n <= 0) { return 0; }
    if(n_decoder > UNPACK_MAX_DECODER) { return static_cast<int>(hipErrorInvalidValue); }
    PatchArgs args;
    args.escapes = escapes;
    args.n = n;
    args.first_read = first_read;
    args.chunk_capacity = chunk_capacity;
    args.base = base;
    args.quality = quality;
    args.n_decoder = n_decoder;
    for(int d = 0; d < UNPACK_MAX_DECODER; ++d) { args.prefix[d] = d < n_decoder ? prefix[d] : 0; args.stride[d] = d < n_decoder ? stride[d] : 0; }
    const unsigned blocks = static_cast<unsigned>((n + UNPACK_THREADS - 1) / UNPACK_THREADS);
    hipLaunchKernelGGL(patch_escapes_kernel, dim3(blocks), dim3(UNPACK_THREADS), 0, static_cast<hipStream_t>(stream), args);
    return static_cast<int>(hipGetLastError());
}

}  // namespace phq
