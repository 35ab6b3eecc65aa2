// Product form PAMLD kernels whose barcode table fits 1 wave wide register (64 packed words).
#include "pamld_lut.hpp"

namespace phq {
int launch_pamld_lut_t1(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return lut::launch_range<1, 16, 1>(args, host_decoder, grid, histogram_bytes, stream);
}
}  // namespace phq
