// One lookup MDD (mdd_near.hpp): single segment ACGT codecs of 5 to 10 nt, distance tolerance 0 or 1.
#include "mdd_near.hpp"

namespace phq {

bool mdd_near_supports(const DevDecoder& host_decoder) {
    const int L = host_decoder.nucleotide_cardinality;
    return host_decoder.algorithm == 2 && host_decoder.mdd_near_ready != 0 && host_decoder.near_table != nullptr && host_decoder.n_segment == 1 &&
           L >= NEAR_MIN_LENGTH && L <= NEAR_MAX_LENGTH && host_decoder.plane_stride == 4 * ((L + 3) / 4) && host_decoder.distance_tolerance[0] <= 1;
}

int launch_mdd_near(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.nucleotide_cardinality) {
        case 5: return mdd_near::launch_one<5>(args, host_decoder, grid, histogram_bytes, stream);
        case 6: return mdd_near::launch_one<6>(args, host_decoder, grid, histogram_bytes, stream);
        case 7: return mdd_near::launch_one<7>(args, host_decoder, grid, histogram_bytes, stream);
        case 8: return mdd_near::launch_one<8>(args, host_decoder, grid, histogram_bytes, stream);
        case 9: return mdd_near::launch_one<9>(args, host_decoder, grid, histogram_bytes, stream);
        case 10: return mdd_near::launch_one<10>(args, host_decoder, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

// two segment codecs whose words belong to one barcode each (pamld_pair.hpp's tables), tolerances of at most 2
bool mdd_pair_supports(const DevDecoder& host_decoder) {
    const int words = host_decoder.plane_stride / 4;
    return host_decoder.algorithm == 2 && host_decoder.mdd_pair_ready != 0 && host_decoder.n_segment == 2 && words >= 2 && words <= 8 &&
           host_decoder.nucleotide_cardinality <= 32 && host_decoder.distance_tolerance[0] <= 2 && host_decoder.distance_tolerance[1] <= 2;
}

int launch_mdd_pair(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.plane_stride / 4) {
        case 2: return mdd_near::launch_pair<2>(args, host_decoder, grid, histogram_bytes, stream);
        case 3: return mdd_near::launch_pair<3>(args, host_decoder, grid, histogram_bytes, stream);
        case 4: return mdd_near::launch_pair<4>(args, host_decoder, grid, histogram_bytes, stream);
        case 5: return mdd_near::launch_pair<5>(args, host_decoder, grid, histogram_bytes, stream);
        case 6: return mdd_near::launch_pair<6>(args, host_decoder, grid, histogram_bytes, stream);
        case 7: return mdd_near::launch_pair<7>(args, host_decoder, grid, histogram_bytes, stream);
        case 8: return mdd_near::launch_pair<8>(args, host_decoder, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

// two segment combinatorial codecs numbered in lexicographic order, segments of 5 to 10 nt, tolerances of at most 1
bool mdd_cross2_supports(const DevDecoder& host_decoder) {
    const int words = host_decoder.plane_stride / 4;
    return host_decoder.algorithm == 2 && host_decoder.mdd_cross_ready != 0 && host_decoder.n_segment == 2 && host_decoder.cross_segments == 2 &&
           host_decoder.cross_identity != 0 && words >= 2 && words <= 8 && host_decoder.nucleotide_cardinality <= 32 &&
           host_decoder.distance_tolerance[0] <= 1 && host_decoder.distance_tolerance[1] <= 1;
}

int launch_mdd_cross2(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.plane_stride / 4) {
        case 2: return mdd_near::launch_cross<2>(args, host_decoder, grid, histogram_bytes, stream);
        case 3: return mdd_near::launch_cross<3>(args, host_decoder, grid, histogram_bytes, stream);
        case 4: return mdd_near::launch_cross<4>(args, host_decoder, grid, histogram_bytes, stream);
        case 5: return mdd_near::launch_cross<5>(args, host_decoder, grid, histogram_bytes, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace phq
