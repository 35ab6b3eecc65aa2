// Streaming kernel of two segment combinatorial barcode sets (pamld_cross_stream.hpp): equal segment lengths of 8 to 10 nt,
// uniform priors (lists and scan in one launch) or priors that differ (pass A alone, the weighted scan kernel behind it),
// counters rebuilt from the records.
#include <cstdlib>

#include "pamld_cross_stream.hpp"

namespace phq {

namespace {
uint32_t bytes_needed(const DevDecoder& host_decoder) {
    const int u0 = host_decoder.cross_unique[0], u1 = host_decoder.cross_unique[1];
    switch(host_decoder.cross_length[0]) {
        case 8: return cross_stream::Shape<8, CROSS_STREAM_WAVES>::lds_bytes(u0, u1);
        case 9: return cross_stream::Shape<9, CROSS_STREAM_WAVES>::lds_bytes(u0, u1);
        default: return cross_stream::Shape<10, CROSS_STREAM_WAVES>::lds_bytes(u0, u1);
    }
}
}

int pamld_cross_stream_waves() { return CROSS_STREAM_WAVES; }

bool pamld_cross_stream_supports(const DevDecoder& host_decoder) {
    static const bool no_priors = std::getenv("PHQ_NO_CROSS_STREAM_PRIORS") != nullptr;
    if(!pamld_cross_near_supports(host_decoder)) { return false; }
    const int L = host_decoder.cross_length[0];
    // priors that differ: pass A with the prior ratios of pamld_cross_near.hpp, the weighted scan kernel behind it
    return host_decoder.cross_stream_ready != 0 && (host_decoder.uniform_prior != 0 || (host_decoder.cross_near_weight[0] != nullptr && host_decoder.cross_near_weight[1] != nullptr && !no_priors)) &&
           L >= 8 && L <= 10 && host_decoder.use_lds_histogram == 0 &&
           host_decoder.cross_identity != 0 && host_decoder.identity_index != 0 && bytes_needed(host_decoder) <= 160u * 1024u;
}

int launch_pamld_cross_stream(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, hipStream_t stream) {
    if(!host_decoder.uniform_prior) {
        switch(host_decoder.cross_length[0]) {
            case 8: return cross_stream::launch_one<8, CROSS_STREAM_WAVES, true>(args, host_decoder, grid, stream);
            case 9: return cross_stream::launch_one<9, CROSS_STREAM_WAVES, true>(args, host_decoder, grid, stream);
            case 10: return cross_stream::launch_one<10, CROSS_STREAM_WAVES, true>(args, host_decoder, grid, stream);
            default: return static_cast<int>(hipErrorInvalidValue);
        }
    }
    switch(host_decoder.cross_length[0]) {
        case 8: return cross_stream::launch_one<8, CROSS_STREAM_WAVES>(args, host_decoder, grid, stream);
        case 9: return cross_stream::launch_one<9, CROSS_STREAM_WAVES>(args, host_decoder, grid, stream);
        case 10: return cross_stream::launch_one<10, CROSS_STREAM_WAVES>(args, host_decoder, grid, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace phq
