// Exact fp64 tier (pamld_cross.hpp) for single segment barcode sets: the second tier behind pamld_lut.hpp.
#include "pamld_cross.hpp"

namespace phq {

int launch_pamld_cross_exact_s1(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(host_decoder.cross_unique[0] > CROSS_MAX_UNIQUE) { return cross::launch_exact_single_long(args, host_decoder, grid, histogram_bytes, stream); }
    return cross::launch_exact_segments<1>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
