// Neighbour list tier of the PAMLD path (pamld_near.hpp): ACGT barcodes of 5 to 8 nt, uniform priors.
#include "pamld_near.hpp"

namespace phq {

bool pamld_near_supports(const DevDecoder& host_decoder) {
    return pamld_lut_supports(host_decoder) && host_decoder.near_ready != 0 && host_decoder.uniform_prior != 0 &&
           host_decoder.nucleotide_cardinality >= 5 && host_decoder.nucleotide_cardinality <= 8 &&
           host_decoder.plane_stride == 8 && host_decoder.n_barcode > NEAR_LIST && host_decoder.use_lds_histogram != 0 &&
           near::Layout::histogram_at(host_decoder.n_barcode) + 8u * (STAT_JOB_WORDS + (host_decoder.use_lds_histogram ? (host_decoder.n_barcode + 1) * STAT_WORDS_PER_BIN : STAT_WORDS_PER_BIN)) <= 160u * 1024u;
}

int launch_pamld_near(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.nucleotide_cardinality) {
        case 5: return near::launch_one<5>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 6: return near::launch_one<6>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 7: return near::launch_one<7>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 8: return near::launch_one<8>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace phq
