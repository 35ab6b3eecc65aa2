// Packed observation planes (phq_pack_planes / phq_classify_host_packed): see unpack.hip.
#pragma once

#include <cstddef>
#include <cstdint>

namespace phq {

constexpr int UNPACK_MAX_DECODER = 16;
constexpr uint8_t PACKED_ESCAPE = 0x3f;       // a byte whose nucleotide is in the list beside the planes
constexpr uint8_t PACKED_NOTHING = 0x7f;      // (0, 0): past the observed length, and the columns a plane's stride adds to its barcode

// ordinal << 6 | quality of a nucleotide that packs, PACKED_ESCAPE of one that does not
inline uint8_t pack_nucleotide(uint8_t code, uint8_t quality) {
    if(code == 0 && quality == 0) { return PACKED_NOTHING; }
    if(quality > 62) { return PACKED_ESCAPE; }
    switch(code) {
        case 1: return static_cast<uint8_t>(quality);
        case 2: return static_cast<uint8_t>(0x40 | quality);
        case 4: return static_cast<uint8_t>(0x80 | quality);
        case 8: return static_cast<uint8_t>(0xc0 | quality);
        default: return PACKED_ESCAPE;
    }
}
// read << 32 | column << 16 | code << 8 | quality: column = the byte's place among the plane_width bytes of a read
inline uint64_t escape_entry(int64_t read, int32_t column, uint8_t code, uint8_t quality) {
    return (static_cast<uint64_t>(read) << 32) | (static_cast<uint64_t>(column & 0xffff) << 16) | (static_cast<uint64_t>(code) << 8) | quality;
}

// every function returns a hipError_t value; all work goes to `stream`; `bytes` is rounded up to whole units of 16
int unpack_planes(const uint8_t* packed, uint8_t* base, uint8_t* quality, size_t bytes, void* stream);
int patch_escapes(const uint64_t* escapes, int64_t n, int64_t first_read, int64_t chunk_capacity, int32_t n_decoder, const int32_t* prefix, const int32_t* stride,
                  uint8_t* base, uint8_t* quality, void* stream);

}  // namespace phq
