// Dispatcher of the product form PAMLD kernels (pamld_lut.hpp), by how many wave wide registers the packed
// barcode table needs.
#include "kernel_common.hpp"

namespace phq {

int launch_pamld_lut_t1(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);
int launch_pamld_lut_t2(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);
int launch_pamld_lut_t4(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);

static int packed_words(const DevDecoder& host_decoder) {
    const int padded = (host_decoder.n_barcode + 7) & ~7;     // whole blocks of 8 barcodes
    return host_decoder.nucleotide_cardinality <= 8 ? padded / 2 : padded;
}

bool pamld_lut_supports(const DevDecoder& host_decoder) {
    return host_decoder.all_strict && host_decoder.nucleotide_cardinality >= 1 && host_decoder.nucleotide_cardinality <= 16 &&
           host_decoder.n_barcode > 0 && packed_words(host_decoder) <= 256 && host_decoder.barcode_packed2 != nullptr;
}

int pamld_lut_block(const DevDecoder& host_decoder) {
    return host_decoder.nucleotide_cardinality <= 8 ? 512 : 256;
}

int launch_pamld_lut(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    const int words = packed_words(host_decoder);
    if(words <= 64) { return launch_pamld_lut_t1(args, host_decoder, grid, histogram_bytes, stream); }
    if(words <= 128) { return launch_pamld_lut_t2(args, host_decoder, grid, histogram_bytes, stream); }
    return launch_pamld_lut_t4(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
