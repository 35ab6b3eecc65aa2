// Factorised PAMLD kernels for three segment combinatorial barcode sets.
#include "pamld_cross.hpp"

namespace phq {

int launch_pamld_cross_s3(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return cross::launch_segments<3, false>(args, host_decoder, grid, histogram_bytes, stream);
}

int launch_pamld_cross_exact_s3(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return cross::launch_exact_segments<3>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
