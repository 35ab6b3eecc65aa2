// Product form PAMLD kernels for barcodes of up to 8 nt (two barcodes per packed word).
#include "pamld_lut.hpp"

namespace phq {
int launch_pamld_lut_l8(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return lut::launch_range<1, 8>(args, host_decoder, grid, histogram_bytes, stream);
}
}  // namespace phq
