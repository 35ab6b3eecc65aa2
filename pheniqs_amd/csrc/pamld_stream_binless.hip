// The streaming PAMLD kernel (pamld_stream.hpp) for codecs whose counters do not fit a workgroup — 171 to ~500 barcodes of 5
// to 10 nt: blocks without bins, the counters of the classified reads rebuilt from the records (counters.hip).  A unit of its
// own so that the instances compile beside pamld_stream.hip's.
#include "pamld_stream.hpp"

namespace phq {

namespace {
template <bool PRIORS>
int launch_by_length(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, hipStream_t stream) {
    switch(host_decoder.nucleotide_cardinality) {
        case 5: return stream::launch_one<5, STREAM_WAVES, PRIORS, false>(args, host_decoder.n_barcode, grid, stream);
        case 6: return stream::launch_one<6, STREAM_WAVES, PRIORS, false>(args, host_decoder.n_barcode, grid, stream);
        case 7: return stream::launch_one<7, STREAM_WAVES, PRIORS, false>(args, host_decoder.n_barcode, grid, stream);
        case 8: return stream::launch_one<8, STREAM_WAVES, PRIORS, false>(args, host_decoder.n_barcode, grid, stream);
        case 9: return stream::launch_one<9, STREAM_WAVES, PRIORS, false>(args, host_decoder.n_barcode, grid, stream);
        case 10: return stream::launch_one<10, STREAM_WAVES, PRIORS, false>(args, host_decoder.n_barcode, grid, stream);
        default: break;
    }
    return static_cast<int>(hipErrorInvalidValue);
}
}

int launch_pamld_stream_binless(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, hipStream_t stream) {
    return host_decoder.uniform_prior ? launch_by_length<false>(args, host_decoder, grid, stream) : launch_by_length<true>(args, host_decoder, grid, stream);
}

}  // namespace phq
