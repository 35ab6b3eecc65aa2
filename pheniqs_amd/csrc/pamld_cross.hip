// Dispatcher of the factorised PAMLD kernels (pamld_cross.hpp) by number of segments.
#include "kernel_common.hpp"

namespace phq {

int launch_pamld_cross_s2(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);
int launch_pamld_cross_s3(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);
int launch_pamld_cross_weighted_s2(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);
int launch_pamld_cross_exact_weighted_s2(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);
int launch_pamld_cross_exact_s1(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);
int launch_pamld_cross_exact_s2(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);
int launch_pamld_cross_exact_s3(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);

// DecoderSet::build_device_images only fills cross_segments for a supported decoder: PAMLD, ACGT barcodes, uniform
// prior, 2 or 3 segments of up to 16 nt, at most CROSS_MAX_UNIQUE distinct words each, every combination present once
bool pamld_cross_supports(const DevDecoder& host_decoder) {
    return host_decoder.cross_segments >= 2 && host_decoder.cross_segments <= CROSS_MAX_SEGMENT &&
           (host_decoder.uniform_prior || (host_decoder.cross_segments == 2 && host_decoder.cross_prior != nullptr)) &&
           host_decoder.high_quality_threshold <= 128 &&               // the byte parallel quality compare of the scan kernel
           host_decoder.cross_groups >= 1 && host_decoder.cross_groups <= 4 &&
           host_decoder.cross_words != nullptr && host_decoder.cross_ordinal != nullptr;
}

int launch_pamld_cross(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(!host_decoder.uniform_prior) { return launch_pamld_cross_weighted_s2(args, host_decoder, grid, histogram_bytes, stream); }
    if(host_decoder.cross_segments == 2) { return launch_pamld_cross_s2(args, host_decoder, grid, histogram_bytes, stream); }
    return launch_pamld_cross_s3(args, host_decoder, grid, histogram_bytes, stream);
}

// the exact tier also serves single segment sets behind the per barcode product form (256 lane workgroups there too)
bool pamld_exact_supports(const DevDecoder& host_decoder) {
    return host_decoder.cross_segments >= 1 && host_decoder.cross_segments <= CROSS_MAX_SEGMENT &&
           (host_decoder.uniform_prior || (host_decoder.cross_segments == 2 && host_decoder.cross_prior != nullptr)) &&
           host_decoder.cross_groups >= 1 && host_decoder.cross_groups <= 4 &&
           host_decoder.cross_words != nullptr && host_decoder.cross_ordinal != nullptr;
}

// tier 1.5: same grid as the producer, consumes and compacts the worklist in place
int launch_pamld_cross_exact(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(!host_decoder.uniform_prior) { return launch_pamld_cross_exact_weighted_s2(args, host_decoder, grid, histogram_bytes, stream); }
    if(host_decoder.cross_segments == 1) { return launch_pamld_cross_exact_s1(args, host_decoder, grid, histogram_bytes, stream); }
    if(host_decoder.cross_segments == 2) { return launch_pamld_cross_exact_s2(args, host_decoder, grid, histogram_bytes, stream); }
    return launch_pamld_cross_exact_s3(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace phq
