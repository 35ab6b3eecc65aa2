// Neighbour list tier of the PAMLD path (pamld_near.hpp): ACGT barcodes of 5 to 10 nt, uniform priors.
#include "pamld_near.hpp"

namespace phq {

bool pamld_near_supports(const DevDecoder& host_decoder) {
    const int L = host_decoder.nucleotide_cardinality;
    if(!(pamld_lut_supports(host_decoder) && host_decoder.near_ready != 0 && host_decoder.uniform_prior != 0 &&
         L >= NEAR_MIN_LENGTH && L <= NEAR_MAX_LENGTH && host_decoder.plane_stride == 4 * ((L + 3) / 4) &&
         host_decoder.n_barcode > near_list(L) && host_decoder.use_lds_histogram != 0)) { return false; }
    const size_t histogram = 8u * (STAT_JOB_WORDS + static_cast<size_t>(host_decoder.n_barcode + 1) * STAT_WORDS_PER_BIN);
    const size_t front = L <= 8 ? near::Shape<8>::histogram_at(host_decoder.n_barcode) : near::Shape<10>::histogram_at(host_decoder.n_barcode);
    return front + histogram <= 160u * 1024u;
}

int launch_pamld_near(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.nucleotide_cardinality) {
        case 5: return near::launch_one<5>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 6: return near::launch_one<6>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 7: return near::launch_one<7>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 8: return near::launch_one<8>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 9: return near::launch_one<9>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 10: return near::launch_one<10>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace phq
