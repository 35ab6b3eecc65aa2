// Neighbour list tier of the PAMLD path (pamld_near.hpp): ACGT barcodes of 5 to 10 nt.
#include "pamld_near.hpp"

namespace phq {

namespace {
template <bool PRIORS> size_t near_front_bytes(int L, int n_barcode) {
    return L <= 8 ? near::Shape<8, PRIORS>::histogram_at(n_barcode) : near::Shape<10, PRIORS>::histogram_at(n_barcode);
}
}

bool pamld_near_supports(const DevDecoder& host_decoder) {
    const int L = host_decoder.nucleotide_cardinality;
    if(!(pamld_lut_supports(host_decoder) && host_decoder.near_ready != 0 &&
         L >= NEAR_MIN_LENGTH && L <= NEAR_MAX_LENGTH && host_decoder.plane_stride == 4 * ((L + 3) / 4) &&
         host_decoder.n_barcode > near_list(L) && host_decoder.use_lds_histogram != 0)) { return false; }
    const size_t histogram = 8u * (STAT_JOB_WORDS + static_cast<size_t>(host_decoder.n_barcode + 1) * STAT_WORDS_PER_BIN);
    const size_t front = host_decoder.uniform_prior ? near_front_bytes<false>(L, host_decoder.n_barcode) : near_front_bytes<true>(L, host_decoder.n_barcode);
    return front + histogram <= 160u * 1024u;
}

int pamld_near_block(const DevDecoder& host_decoder) {
    const bool two_groups = host_decoder.nucleotide_cardinality <= 8;
    if(host_decoder.uniform_prior) { return two_groups ? near::Shape<8, false>::THREADS : near::Shape<10, false>::THREADS; }
    return two_groups ? near::Shape<8, true>::THREADS : near::Shape<10, true>::THREADS;
}

namespace {
template <bool PRIORS>
int launch_by_length(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.nucleotide_cardinality) {
        case 5: return near::launch_one<5, PRIORS>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 6: return near::launch_one<6, PRIORS>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 7: return near::launch_one<7, PRIORS>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 8: return near::launch_one<8, PRIORS>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 9: return near::launch_one<9, PRIORS>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 10: return near::launch_one<10, PRIORS>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}
}

int launch_pamld_near(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return host_decoder.uniform_prior ? launch_by_length<false>(args, host_decoder, grid, histogram_bytes, stream)
                                      : launch_by_length<true>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
