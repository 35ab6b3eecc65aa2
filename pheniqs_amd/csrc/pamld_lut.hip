// Dispatcher of the product form PAMLD kernels (pamld_lut.hpp) by barcode length.
#include "kernel_common.hpp"

namespace phq {

int launch_pamld_lut_l8(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);
int launch_pamld_lut_l16(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);

static int packed_words(const DevDecoder& host_decoder) {
    const int padded = (host_decoder.n_barcode + 7) & ~7;     // whole blocks of 8 barcodes
    return host_decoder.nucleotide_cardinality <= 8 ? padded / 2 : padded;
}

// ACGT barcodes of up to 16 nt and at most LUT_MOST_WORDS packed words: their two planes take 8 bytes of LDS each beside the per
// read tables (64 KiB) — 4 096 barcodes of 9 to 16 nt are 32 KiB, one workgroup a CU instead of two.  (512 words until round 5:
// a codec past them had NO kernel but the literal one, for every read: 768 indexes of 10 nt 0.4 Greads/s, the cliff
// profiles/r05_shape_grid.md found.)  The byte parallel quality compare of the kernel covers thresholds up to 128
constexpr int LUT_MOST_WORDS = 4096;
bool pamld_lut_supports(const DevDecoder& host_decoder) {
    return host_decoder.all_strict && host_decoder.high_quality_threshold <= 128 && host_decoder.nucleotide_cardinality >= 1 && host_decoder.nucleotide_cardinality <= 16 &&
           host_decoder.n_barcode > 0 && packed_words(host_decoder) <= LUT_MOST_WORDS && host_decoder.barcode_packed2 != nullptr &&
           (host_decoder.use_lds_histogram == 0 || host_decoder.n_barcode <= 1024);
}

// ... and it is the FIRST choice up to 512 packed words only: a combinatorial set past them takes the factorised kernels
// (pamld_cross*.hpp: 784 combinations of 8 nt words 25 Greads/s through the streaming kernel of sets, 2.8 through this one)
bool pamld_lut_preferred(const DevDecoder& host_decoder) {
    return pamld_lut_supports(host_decoder) && packed_words(host_decoder) <= 512;
}

int pamld_lut_block(const DevDecoder& host_decoder) {
    return host_decoder.nucleotide_cardinality <= 8 ? 512 : 256;
}

int launch_pamld_lut(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(host_decoder.nucleotide_cardinality <= 8) { return launch_pamld_lut_l8(args, host_decoder, grid, histogram_bytes, stream); }
    return launch_pamld_lut_l16(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
