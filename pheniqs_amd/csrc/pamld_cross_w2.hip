// Factorised PAMLD scan for two segment combinatorial codecs whose priors differ between the barcodes.
#include "pamld_cross.hpp"

namespace phq {

int launch_pamld_cross_weighted_s2(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return cross::launch_segments<2, true>(args, host_decoder, grid, histogram_bytes, stream);
}

int launch_pamld_cross_exact_weighted_s2(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    // the list kernel, then the whole form — a wave per read — on what its lists could not hold
    if(const int status = cross::launch_exact_weighted(args, host_decoder, grid, histogram_bytes, stream)) { return status; }
    trace_worklist("the exact tier's lists", args, grid.x, stream);
    return cross::launch_exact_dense(args, host_decoder, grid, histogram_bytes, stream);
}

// the whole bilinear form alone: behind the lane parallel list kernel (pamld_exact_lanes.hip)
int launch_pamld_cross_exact_dense(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return cross::launch_exact_dense(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
