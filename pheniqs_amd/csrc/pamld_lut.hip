// Dispatcher of the product form PAMLD kernels (pamld_lut.hpp) by barcode length.
#include "kernel_common.hpp"

namespace phq {

int launch_pamld_lut_l8(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);
int launch_pamld_lut_l16(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);

static int packed_words(const DevDecoder& host_decoder) {
    const int padded = (host_decoder.n_barcode + 7) & ~7;     // whole blocks of 8 barcodes
    return host_decoder.nucleotide_cardinality <= 8 ? padded / 2 : padded;
}

// ACGT barcodes of up to 16 nt and at most 512 packed words (LUT_MAX_WORDS: their two planes take 4 KiB of LDS); the byte parallel quality
// compare of the kernel covers thresholds up to 128
bool pamld_lut_supports(const DevDecoder& host_decoder) {
    return host_decoder.all_strict && host_decoder.high_quality_threshold <= 128 && host_decoder.nucleotide_cardinality >= 1 && host_decoder.nucleotide_cardinality <= 16 &&
           host_decoder.n_barcode > 0 && packed_words(host_decoder) <= 512 && host_decoder.barcode_packed2 != nullptr;
}

int pamld_lut_block(const DevDecoder& host_decoder) {
    return host_decoder.nucleotide_cardinality <= 8 ? 512 : 256;
}

int launch_pamld_lut(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(host_decoder.nucleotide_cardinality <= 8) { return launch_pamld_lut_l8(args, host_decoder, grid, histogram_bytes, stream); }
    return launch_pamld_lut_l16(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
