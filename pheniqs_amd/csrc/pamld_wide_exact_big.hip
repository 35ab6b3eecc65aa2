// pamld_wide_exact.hpp's kernels with list masks of 64 words: two word barcodes (17 to 32 nt) in sets of 1 025 to 2 048 barcodes.
// A unit of its own so that it compiles beside pamld_wide_exact.hip instead of behind it.
#include "pamld_wide_exact.hpp"

namespace phq {

int launch_pamld_wide_exact_big(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    constexpr int MASK_WORDS = wide::EXACT_MAX_BARCODES / 32;
    switch(host_decoder.plane_stride / 4) {
        case 5: return wide::launch_exact_masked<5, MASK_WORDS>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 6: return wide::launch_exact_masked<6, MASK_WORDS>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 7: return wide::launch_exact_masked<7, MASK_WORDS>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 8: return wide::launch_exact_masked<8, MASK_WORDS>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace phq
