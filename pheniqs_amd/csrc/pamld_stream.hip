// Streaming PAMLD kernel (pamld_stream.hpp): ACGT barcodes of 5 to 10 nt, a workgroup histogram; priors uniform or not.
#include "pamld_stream.hpp"

namespace phq {

namespace {
uint32_t bytes_needed(const DevDecoder& host_decoder) {
    if(host_decoder.nucleotide_cardinality > 8) {
        return host_decoder.uniform_prior ? stream::Shape<10, STREAM_WAVES, false>::lds_bytes(host_decoder.n_barcode) : stream::Shape<10, STREAM_WAVES, true>::lds_bytes(host_decoder.n_barcode);
    }
    return host_decoder.uniform_prior ? stream::Shape<8, STREAM_WAVES, false>::lds_bytes(host_decoder.n_barcode) : stream::Shape<8, STREAM_WAVES, true>::lds_bytes(host_decoder.n_barcode);
}
template <bool PRIORS>
int launch_by_length(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, hipStream_t stream) {
    switch(host_decoder.nucleotide_cardinality) {
        case 5: return stream::launch_one<5, STREAM_WAVES, PRIORS>(args, host_decoder.n_barcode, grid, stream);
        case 6: return stream::launch_one<6, STREAM_WAVES, PRIORS>(args, host_decoder.n_barcode, grid, stream);
        case 7: return stream::launch_one<7, STREAM_WAVES, PRIORS>(args, host_decoder.n_barcode, grid, stream);
        case 8: return stream::launch_one<8, STREAM_WAVES, PRIORS>(args, host_decoder.n_barcode, grid, stream);
        case 9: return stream::launch_one<9, STREAM_WAVES, PRIORS>(args, host_decoder.n_barcode, grid, stream);
        case 10: return stream::launch_one<10, STREAM_WAVES, PRIORS>(args, host_decoder.n_barcode, grid, stream);
        default: break;
    }
    return static_cast<int>(hipErrorInvalidValue);
}
}

int pamld_stream_waves() { return STREAM_WAVES; }

bool pamld_stream_supports(const DevDecoder& host_decoder) {
    const int L = host_decoder.nucleotide_cardinality;
    return pamld_lut_supports(host_decoder) && host_decoder.stream_ready != 0 && host_decoder.stream_table != nullptr &&
           L >= NEAR_MIN_LENGTH && L <= 10 && host_decoder.plane_stride == 4 * stream_row_words(L) && host_decoder.n_barcode > stream::LIST &&
           (L <= 8 || host_decoder.identity_index != 0) &&          // past 8 nt the lookup table holds ordinals
           (host_decoder.uniform_prior != 0 || host_decoder.concentration_single != nullptr) &&
           host_decoder.use_lds_histogram != 0 && bytes_needed(host_decoder) <= 160u * 1024u;
}

int launch_pamld_stream(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, hipStream_t stream) {
    return host_decoder.uniform_prior ? launch_by_length<false>(args, host_decoder, grid, stream) : launch_by_length<true>(args, host_decoder, grid, stream);
}

}  // namespace phq
