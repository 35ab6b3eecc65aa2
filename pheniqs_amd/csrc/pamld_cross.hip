// Dispatcher of the factorised PAMLD kernels (pamld_cross.hpp) by number of segments.
#include <algorithm>
#include <cstdlib>

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
           (host_decoder.uniform_prior || host_decoder.cross_segments == 1 || (host_decoder.cross_segments == 2 && host_decoder.cross_prior != nullptr)) &&
           host_decoder.cross_groups >= 1 && host_decoder.cross_groups <= 4 &&
           host_decoder.cross_words != nullptr && host_decoder.cross_ordinal != nullptr;
}

// ... and the kernels of this file hold a single segment set of up to EXACT_SINGLE_MEDIUM words (their list is a bit a word in
// registers); a larger one — up to EXACT_SINGLE_UNIQUE — has pamld_exact_lanes.hip for its exact tier, or none
bool pamld_exact_legacy_fits(const DevDecoder& host_decoder) {
    return host_decoder.cross_segments != 1 || host_decoder.cross_unique[0] <= EXACT_SINGLE_MEDIUM;
}

// tier 1.5: consumes and compacts the producer's worklist in place.  `grid` is the producer's grid = the number of
// segments.  A segment holds ~1.4 % of a workgroup's reads, a fraction of a tile when the batch is small, and every
// workgroup pays its set up (factor and cost rows, histogram): for small batches the kernel takes groups of segments
// per workgroup.  Per 1 M / 4 M / 16 M C2 reads this tier took 0.099 / 0.147 / 0.149 ms with one segment per workgroup,
// 0.046 / 0.048 / 0.081 ms grouped.
int launch_pamld_cross_exact(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(!host_decoder.uniform_prior && host_decoder.cross_segments != 1) { return launch_pamld_cross_exact_weighted_s2(args, host_decoder, grid, histogram_bytes, stream); }
    StageArgs grouped = args;
    grouped.worklist_segments = static_cast<int32_t>(grid.x);
    // about 1.5 % of the reads reach this tier: group segments until a workgroup can expect a tile's worth of them; a
    // large batch keeps one segment per workgroup (the tier's long dependent chains want many workgroups in flight:
    // grouping by two cost C3 0.44 -> 0.59 ms)
    const double expected_per_segment = 0.015 * static_cast<double>(args.n_read) / static_cast<double>(std::max(1u, grid.x));
    const int group = expected_per_segment >= 128.0 ? 1 : static_cast<int>(256.0 / std::max(expected_per_segment, 1.0)) + 1;
    grouped.worklist_group = std::min(EXACT_GROUP_MAX, std::max(1, group));
    // a single segment set (behind the streaming or the product form kernel: 12 segments per CU): four segments per
    // workgroup, so that every workgroup is resident from the start — measured on C2, 100 M reads: 0.227 ms with one
    // segment per workgroup, 0.215 / 0.232 / 0.184 / 0.249 / 0.337 ms with 2 / 3 / 4 / 6 / 8
    if(host_decoder.cross_segments == 1) { grouped.worklist_group = std::max(grouped.worklist_group, 4); }
    static const int forced = std::getenv("PHQ_EXACT_GROUP") ? std::atoi(std::getenv("PHQ_EXACT_GROUP")) : 0;       // sweeps
    if(forced > 0) { grouped.worklist_group = std::min(EXACT_GROUP_MAX, forced); }
    const dim3 fewer(static_cast<unsigned>((grid.x + grouped.worklist_group - 1) / grouped.worklist_group));
    if(host_decoder.cross_segments == 1) { return launch_pamld_cross_exact_s1(grouped, host_decoder, fewer, histogram_bytes, stream); }
    if(host_decoder.cross_segments == 2) { return launch_pamld_cross_exact_s2(grouped, host_decoder, fewer, histogram_bytes, stream); }
    return launch_pamld_cross_exact_s3(grouped, host_decoder, fewer, histogram_bytes, stream);
}

}  // namespace phq
