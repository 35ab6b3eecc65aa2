// Exact tier behind the two word product form (pamld_wide_exact.hpp): ACGT barcodes of 17 to 32 nt, up to 512 of them.
#include "pamld_wide_exact.hpp"

namespace phq {

bool pamld_wide_exact_supports(const DevDecoder& host_decoder) {
    return pamld_wide_supports(host_decoder) && host_decoder.n_barcode <= wide::EXACT_MAX_BARCODES;
}

// `grid` is the producer's grid = the number of worklist segments; one workgroup of one wave per segment
int launch_pamld_wide_exact_big(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream);      // pamld_wide_exact_big.hip

int launch_pamld_wide_exact(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(host_decoder.n_barcode > 1024) { return launch_pamld_wide_exact_big(args, host_decoder, grid, histogram_bytes, stream); }
    switch(host_decoder.plane_stride / 4) {
        case 5: return wide::launch_exact_groups<5>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 6: return wide::launch_exact_groups<6>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 7: return wide::launch_exact_groups<7>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 8: return wide::launch_exact_groups<8>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace phq
