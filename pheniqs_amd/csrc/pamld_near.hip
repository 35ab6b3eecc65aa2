// Neighbour list tier of the PAMLD path (pamld_near.hpp): ACGT barcodes of 5 to 10 nt.
#include "pamld_near.hpp"

namespace phq {

namespace {
// bytes of this launch's workgroup histogram: the decoder's bins when they are kept in LDS, else bin 0 alone
size_t histogram_bytes_of(const DevDecoder& host_decoder) {
    return 8u * (STAT_JOB_WORDS + (host_decoder.use_lds_histogram ? static_cast<size_t>(host_decoder.n_barcode + 1) : 1u) * STAT_WORDS_PER_BIN);
}
bool big_shape(const DevDecoder& host_decoder) {
    return host_decoder.nucleotide_cardinality <= 8 && near::wants_big<8>(host_decoder.n_barcode, histogram_bytes_of(host_decoder));
}
// priors that differ, 5 to 8 nt: the 1 024 lane shape while its records fit beside its tables, the slim one past that
bool slim_shape(const DevDecoder& host_decoder) {
    return host_decoder.nucleotide_cardinality <= 8 && !host_decoder.uniform_prior &&
           near::Shape<8, true>::histogram_at(host_decoder.n_barcode) + histogram_bytes_of(host_decoder) > 160u * 1024u;
}
size_t front_bytes(const DevDecoder& host_decoder) {
    const int n = host_decoder.n_barcode;
    if(host_decoder.nucleotide_cardinality > 8) { return host_decoder.uniform_prior ? near::Shape<10, false>::histogram_at(n) : near::Shape<10, true>::histogram_at(n); }
    if(!host_decoder.uniform_prior) { return slim_shape(host_decoder) ? near::Shape<8, true, false, true>::histogram_at(n) : near::Shape<8, true>::histogram_at(n); }
    return big_shape(host_decoder) ? near::Shape<8, false, true>::histogram_at(n) : near::Shape<8, false, false>::histogram_at(n);
}
}

// a codec without a workgroup histogram (more than ~340 barcodes, or room made for the product form kernel) needs the
// launch to defer the classified reads' counters (count_results_kernel): stage.hip checks that
bool pamld_near_supports(const DevDecoder& host_decoder) {
    const int L = host_decoder.nucleotide_cardinality;
    if(!(pamld_lut_supports(host_decoder) && host_decoder.near_ready != 0 &&
         L >= NEAR_MIN_LENGTH && L <= NEAR_MAX_LENGTH && host_decoder.plane_stride == 4 * ((L + 3) / 4) &&
         host_decoder.n_barcode > near_list(L) && (host_decoder.use_lds_histogram != 0 || count_results_supports(host_decoder)))) { return false; }
    return front_bytes(host_decoder) + histogram_bytes_of(host_decoder) <= 160u * 1024u;
}

int pamld_near_block(const DevDecoder& host_decoder) {
    const bool two_groups = host_decoder.nucleotide_cardinality <= 8;
    if(!two_groups) { return near::Shape<10, false>::THREADS; }
    if(slim_shape(host_decoder)) { return near::Shape<8, true, false, true>::THREADS; }
    return (!host_decoder.uniform_prior || big_shape(host_decoder)) ? near::Shape<8, true>::THREADS : near::Shape<8, false>::THREADS;
}

namespace {
template <bool PRIORS, bool BIG>
int launch_by_length(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.nucleotide_cardinality) {
        case 5: return near::launch_one<5, PRIORS, BIG>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 6: return near::launch_one<6, PRIORS, BIG>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 7: return near::launch_one<7, PRIORS, BIG>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 8: return near::launch_one<8, PRIORS, BIG>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
        case 9: if constexpr(!BIG) { return near::launch_one<9, PRIORS, false>(args, host_decoder.n_barcode, grid, histogram_bytes, stream); } else { break; }
        case 10: if constexpr(!BIG) { return near::launch_one<10, PRIORS, false>(args, host_decoder.n_barcode, grid, histogram_bytes, stream); } else { break; }
        default: break;
    }
    return static_cast<int>(hipErrorInvalidValue);
}
}

int launch_pamld_near(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(slim_shape(host_decoder)) {
        switch(host_decoder.nucleotide_cardinality) {
            case 5: return near::launch_one<5, true, false, true>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
            case 6: return near::launch_one<6, true, false, true>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
            case 7: return near::launch_one<7, true, false, true>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
            case 8: return near::launch_one<8, true, false, true>(args, host_decoder.n_barcode, grid, histogram_bytes, stream);
            default: return static_cast<int>(hipErrorInvalidValue);
        }
    }
    if(!host_decoder.uniform_prior) { return launch_by_length<true, false>(args, host_decoder, grid, histogram_bytes, stream); }
    return big_shape(host_decoder) ? launch_by_length<false, true>(args, host_decoder, grid, histogram_bytes, stream)
                                   : launch_by_length<false, false>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
