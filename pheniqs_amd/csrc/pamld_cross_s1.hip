// Exact fp64 tier (pamld_cross.hpp) for single segment barcode sets: the second tier behind pamld_lut.hpp.  (Sets of up to
// EXACT_SINGLE_MEDIUM words: the larger ones are pamld_exact_lanes.hip's alone — pamld_exact_legacy_fits.)
#include "pamld_cross.hpp"

namespace phq {

int launch_pamld_cross_exact_s1(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    const bool many = host_decoder.cross_unique[0] > CROSS_MAX_UNIQUE;
    if(!host_decoder.uniform_prior) {
        return many ? cross::launch_exact_single<EXACT_SINGLE_MEDIUM, true>(args, host_decoder, grid, histogram_bytes, stream)
                    : cross::launch_exact_single<CROSS_MAX_UNIQUE, true>(args, host_decoder, grid, histogram_bytes, stream);
    }
    if(many) { return cross::launch_exact_single<EXACT_SINGLE_MEDIUM, false>(args, host_decoder, grid, histogram_bytes, stream); }
    return cross::launch_exact_segments<1>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
