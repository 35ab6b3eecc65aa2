// Neighbour list tier of two segment combinatorial barcode sets (pamld_cross_near.hpp): equal segment lengths of 5 to 10 nt.
#include "pamld_cross_near.hpp"

namespace phq {

bool pamld_cross_near_supports(const DevDecoder& host_decoder) {
    if(!(host_decoder.algorithm == 1 && host_decoder.cross_near_ready != 0 && (host_decoder.uniform_prior != 0 || host_decoder.cross_near_weight[0] != nullptr) && host_decoder.cross_segments == 2 &&
         host_decoder.cross_length[0] == host_decoder.cross_length[1] && pamld_cross_supports(host_decoder))) { return false; }
    const int L = host_decoder.cross_length[0];
    if(L < NEAR_MIN_LENGTH || L > NEAR_MAX_LENGTH || host_decoder.plane_stride != 4 * ((2 * L + 3) / 4)) { return false; }
    const size_t histogram = 8u * (STAT_JOB_WORDS + (host_decoder.use_lds_histogram ? static_cast<size_t>(host_decoder.n_barcode + 1) : 1u) * STAT_WORDS_PER_BIN);
    const int words0 = host_decoder.cross_unique[0], words1 = host_decoder.cross_unique[1];
    const size_t front = L <= 8 ? cross_near::CrossShape<8>::histogram_at(words0, words1) : cross_near::CrossShape<10>::histogram_at(words0, words1);
    return front + histogram <= 160u * 1024u;
}

int launch_pamld_cross_near(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.cross_length[0]) {
        case 5: return cross_near::launch_one<5>(args, host_decoder, grid, histogram_bytes, stream);
        case 6: return cross_near::launch_one<6>(args, host_decoder, grid, histogram_bytes, stream);
        case 7: return cross_near::launch_one<7>(args, host_decoder, grid, histogram_bytes, stream);
        case 8: return cross_near::launch_one<8>(args, host_decoder, grid, histogram_bytes, stream);
        case 9: return cross_near::launch_one<9>(args, host_decoder, grid, histogram_bytes, stream);
        case 10: return cross_near::launch_one<10>(args, host_decoder, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

int pamld_cross_near_block(const DevDecoder& host_decoder) { return host_decoder.cross_length[0] <= 8 ? 512 : 256; }

}  // namespace phq
