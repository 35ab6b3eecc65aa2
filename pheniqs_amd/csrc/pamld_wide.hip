// Product form PAMLD kernels for ACGT barcodes of 17 to 32 nt (pamld_wide.hpp), by groups of 4 positions.
#include "pamld_wide.hpp"

namespace phq {

bool pamld_wide_supports(const DevDecoder& host_decoder) {
    return host_decoder.algorithm == 1 && host_decoder.all_strict && host_decoder.barcode_wide != nullptr &&
           host_decoder.nucleotide_cardinality > 16 && host_decoder.nucleotide_cardinality <= 32 &&
           host_decoder.n_barcode > 0 && host_decoder.n_barcode <= 2048 &&     // 16 bytes of plane words + a prior per barcode in LDS: 40 KiB beside 64 KiB of tables (512 until round 5: past them the literal kernel took every read)
           (host_decoder.use_lds_histogram == 0 || host_decoder.n_barcode <= 512);
}

int pamld_wide_block(const DevDecoder& host_decoder) { return wide::threads_of(host_decoder.plane_stride / 4); }

int launch_pamld_wide(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.plane_stride / 4) {
        case 5: return wide::launch_groups<5>(args, host_decoder, grid, histogram_bytes, stream);
        case 6: return wide::launch_groups<6>(args, host_decoder, grid, histogram_bytes, stream);
        case 7: return wide::launch_groups<7>(args, host_decoder, grid, histogram_bytes, stream);
        case 8: return wide::launch_groups<8>(args, host_decoder, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace phq
