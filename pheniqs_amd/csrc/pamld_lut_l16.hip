// Product form PAMLD kernels for barcodes of 9 to 16 nt.
#include "pamld_lut.hpp"

namespace phq {
int launch_pamld_lut_l16(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return lut::launch_range<9, 16>(args, host_decoder, grid, histogram_bytes, stream);
}
}  // namespace phq
