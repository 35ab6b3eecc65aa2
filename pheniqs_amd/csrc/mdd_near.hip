// MDD by one table lookup per barcode segment (mdd_near.hpp): which codecs take it, and the dispatch by plane row width.
#include "mdd_near.hpp"

namespace phq {

namespace {
// the kernel's LDS: the words the distances are taken against and the workgroup histogram (no large LDS request: 64 KiB)
bool fits(const DevDecoder& host_decoder, int kind) {
    const size_t histogram = 8u * (STAT_JOB_WORDS + (host_decoder.use_lds_histogram ? static_cast<size_t>(host_decoder.n_barcode + 1) : 1u) * STAT_WORDS_PER_BIN);
    return mdd_near::Layout::histogram_at(mdd_near::record_bytes_of(host_decoder, kind)) + histogram <= 64u * 1024u;
}

template <int KIND>
int launch_by_width(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    switch(host_decoder.plane_stride / 4) {
        case 2: return mdd_near::launch_one<2, KIND>(args, host_decoder, grid, histogram_bytes, stream);
        case 3: return mdd_near::launch_one<3, KIND>(args, host_decoder, grid, histogram_bytes, stream);
        case 4: if constexpr(KIND != mdd_near::SINGLE) { return mdd_near::launch_one<4, KIND>(args, host_decoder, grid, histogram_bytes, stream); } else { break; }
        case 5: if constexpr(KIND != mdd_near::SINGLE) { return mdd_near::launch_one<5, KIND>(args, host_decoder, grid, histogram_bytes, stream); } else { break; }
        case 6: if constexpr(KIND == mdd_near::PAIRED) { return mdd_near::launch_one<6, KIND>(args, host_decoder, grid, histogram_bytes, stream); } else { break; }
        case 7: if constexpr(KIND == mdd_near::PAIRED) { return mdd_near::launch_one<7, KIND>(args, host_decoder, grid, histogram_bytes, stream); } else { break; }
        case 8: if constexpr(KIND == mdd_near::PAIRED) { return mdd_near::launch_one<8, KIND>(args, host_decoder, grid, histogram_bytes, stream); } else { break; }
        default: break;
    }
    return static_cast<int>(hipErrorInvalidValue);
}
}

// one segment of 5 to 10 nt, tolerance 0 or 1
bool mdd_near_supports(const DevDecoder& host_decoder) {
    const int L = host_decoder.nucleotide_cardinality;
    return host_decoder.algorithm == 2 && host_decoder.mdd_near_ready != 0 && host_decoder.near_table != nullptr && host_decoder.n_segment == 1 &&
           L >= NEAR_MIN_LENGTH && L <= NEAR_MAX_LENGTH && host_decoder.plane_stride == 4 * ((L + 3) / 4) && host_decoder.distance_tolerance[0] <= 1 &&
           fits(host_decoder, mdd_near::SINGLE);
}

// two segments whose words belong to one barcode each (pamld_pair.hpp's tables), tolerances of at most 2
bool mdd_pair_supports(const DevDecoder& host_decoder) {
    const int words = host_decoder.plane_stride / 4;
    const bool parts = host_decoder.mdd_pair_parts != 0;            // one segment of 11 to 20 nt as two parts (decoder_set.cpp)
    return host_decoder.algorithm == 2 && host_decoder.mdd_pair_ready != 0 && (parts ? host_decoder.n_segment == 1 : host_decoder.n_segment == 2) && words >= 2 && words <= 8 &&
           host_decoder.nucleotide_cardinality <= 32 && host_decoder.distance_tolerance[0] <= 2 && (parts || host_decoder.distance_tolerance[1] <= 2) &&
           fits(host_decoder, mdd_near::PAIRED);
}

// two segment combinatorial codecs numbered in lexicographic order, segments of 5 to 10 nt, tolerances of at most 1
bool mdd_cross2_supports(const DevDecoder& host_decoder) {
    const int words = host_decoder.plane_stride / 4;
    return host_decoder.algorithm == 2 && host_decoder.mdd_cross_ready != 0 && host_decoder.n_segment == 2 && host_decoder.cross_segments == 2 &&
           host_decoder.cross_identity != 0 && words >= 2 && words <= 5 && host_decoder.distance_tolerance[0] <= 1 && host_decoder.distance_tolerance[1] <= 1 &&
           fits(host_decoder, mdd_near::COMBINATORIAL);
}

int launch_mdd_near(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return launch_by_width<mdd_near::SINGLE>(args, host_decoder, grid, histogram_bytes, stream);
}

int launch_mdd_pair(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return launch_by_width<mdd_near::PAIRED>(args, host_decoder, grid, histogram_bytes, stream);
}

int launch_mdd_cross2(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    return launch_by_width<mdd_near::COMBINATORIAL>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
