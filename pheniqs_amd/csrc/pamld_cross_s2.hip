// Factorised PAMLD kernels for two segment combinatorial barcode sets (dual index).
#include "pamld_cross.hpp"

namespace phq {

int launch_pamld_cross_s2(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return cross::launch_segments<2, false>(args, host_decoder, grid, histogram_bytes, stream);
}

int launch_pamld_cross_exact_s2(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return cross::launch_exact_segments<2>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
