// Product form PAMLD kernels whose barcode table fits 2 wave wide registers (128 packed words).
#include "pamld_lut.hpp"

namespace phq {
int launch_pamld_lut_t2(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return lut::launch_range<1, 16, 2>(args, host_decoder, grid, histogram_bytes, stream);
}
}  // namespace phq
