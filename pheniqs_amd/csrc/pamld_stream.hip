// Streaming PAMLD kernel (pamld_stream.hpp): ACGT barcodes of 5 to 10 nt, priors uniform or not; the counters in the blocks'
// bins (a workgroup histogram) or, for a codec past that, rebuilt from the records (pamld_stream_binless.hip).
#include "pamld_stream.hpp"

namespace phq {

int launch_pamld_stream_binless(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, hipStream_t stream);      // pamld_stream_binless.hip

namespace {
constexpr int BINLESS_MOST = 288;
uint32_t bytes_needed(const DevDecoder& host_decoder) {
    const int n = host_decoder.n_barcode;
    if(host_decoder.stream_binless) {
        if(host_decoder.nucleotide_cardinality > 8) {
            return host_decoder.uniform_prior ? stream::Shape<10, STREAM_WAVES, false, false>::lds_bytes(n) : stream::Shape<10, STREAM_WAVES, true, false>::lds_bytes(n);
        }
        return host_decoder.uniform_prior ? stream::Shape<8, STREAM_WAVES, false, false>::lds_bytes(n) : stream::Shape<8, STREAM_WAVES, true, false>::lds_bytes(n);
    }
    if(host_decoder.nucleotide_cardinality > 8) {
        return host_decoder.uniform_prior ? stream::Shape<10, STREAM_WAVES, false>::lds_bytes(n) : stream::Shape<10, STREAM_WAVES, true>::lds_bytes(n);
    }
    return host_decoder.uniform_prior ? stream::Shape<8, STREAM_WAVES, false>::lds_bytes(n) : stream::Shape<8, STREAM_WAVES, true>::lds_bytes(n);
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
           // its counters: in the blocks' bins, or — a decoder without a workgroup histogram — rebuilt from the records
           // (pass C scans every barcode for the ~15 % such a codec's lists defer: measured, 192 barcodes of 8 nt 4.85 -> 3.46 ms per
           // 100 M reads against the chain of launches, 384 barcodes 4.88 -> 5.55 — so up to BINLESS_MOST barcodes)
           (host_decoder.stream_binless ? (host_decoder.use_lds_histogram == 0 && host_decoder.n_barcode <= BINLESS_MOST && count_results_supports(host_decoder)) : host_decoder.use_lds_histogram != 0) &&
           bytes_needed(host_decoder) <= 160u * 1024u;
}

int launch_pamld_stream(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, hipStream_t stream) {
    if(host_decoder.stream_binless) { return launch_pamld_stream_binless(args, host_decoder, grid, stream); }
    return host_decoder.uniform_prior ? launch_by_length<false>(args, host_decoder, grid, stream) : launch_by_length<true>(args, host_decoder, grid, stream);
}

}  // namespace phq
