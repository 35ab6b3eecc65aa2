// Stage dispatcher: picks the kernel for one decoder and sizes the launch.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernel_common.hpp"

namespace phq {

namespace {
// Plan only (DecoderSet::plan_chain, phq_plan_chain): the dispatcher walks its predicates and names the kernels of a stage to
// the timer as it always does, and launches none — the chain a decoder WOULD take is read off the one place that decides it,
// without a device.  Per thread: a plan on one thread does not silence another thread's launches.
thread_local bool plan_only = false;
template <typename Launch> int launched(Launch&& launch) { return plan_only ? 0 : launch(); }
}  // namespace
void stage_plan_only(bool on) { plan_only = on; }

int launch_pamld_literal_l8(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);
int launch_pamld_literal_l16(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);
int launch_pamld_literal_l24(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);
int launch_pamld_literal_l32(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);

bool pamld_exact_lanes_ties(const StageArgs&, const DevDecoder&, size_t);       // pamld_exact_lanes.hip: its first pass settles tied noise reads

int launch_pamld_literal_long(const StageArgs&, const DevDecoder&, dim3, size_t, hipStream_t);

int launch_pamld_literal(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    const int L = host_decoder.nucleotide_cardinality;
    if(L > DEV_TIER_NUCLEOTIDE) { return launch_pamld_literal_long(args, host_decoder, grid, lds_bytes, stream); }
    if(L <= 8) { return launch_pamld_literal_l8(args, host_decoder, grid, lds_bytes, stream); }
    if(L <= 16) { return launch_pamld_literal_l16(args, host_decoder, grid, lds_bytes, stream); }
    if(L <= 24) { return launch_pamld_literal_l24(args, host_decoder, grid, lds_bytes, stream); }
    return launch_pamld_literal_l32(args, host_decoder, grid, lds_bytes, stream);
}

bool stage_uses_worklist(const DevDecoder& host_decoder, int32_t mode) {
    // barcodes of 33 to 128 nt: the general kernels on every read (the reference is length agnostic, barcode.h:147-161; no tier is)
    if(mode != MODE_FAST || host_decoder.nucleotide_cardinality > DEV_TIER_NUCLEOTIDE) { return false; }
    return (host_decoder.algorithm == 1 && (pamld_tiny_supports(host_decoder) || pamld_lut_supports(host_decoder) || pamld_cross_supports(host_decoder) || pamld_wide_supports(host_decoder))) ||
           (host_decoder.algorithm == 2 && mdd_exact_supports(host_decoder));
}

// PHQ_TRACE_WORKLIST=1: after a tier, how many reads it left in the worklist (synchronises the stream: diagnosis only)
void trace_worklist(const char* tier, const StageArgs& args, unsigned segments, hipStream_t stream) {
    static const bool wanted = std::getenv("PHQ_TRACE_WORKLIST") != nullptr;
    if(!wanted || plan_only || args.worklist_count == nullptr) { return; }
    std::vector<uint32_t> count(segments);
    if(hipStreamSynchronize(stream) != hipSuccess) { return; }
    if(hipMemcpy(count.data(), args.worklist_count, segments * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) { return; }
    uint64_t total = 0;
    uint32_t largest = 0;
    for(uint32_t c : count) { total += c; largest = std::max(largest, c); }
    std::fprintf(stderr, "worklist after %s: %llu of %lld reads in %u segments (largest %u)\n", tier, static_cast<unsigned long long>(total),
                 static_cast<long long>(args.n_read), segments, largest);
}

namespace {
// a launch (or the launches of one tier) between two events of the timer, under the name of the kernel taken: the names
// are given where the dispatcher's predicates have just chosen, so phq_profile_chain reports the chain that really ran
struct Timed {
    LaunchTimer* timer;
    void* stream;
    Timed(LaunchTimer* timer, int32_t kind, void* stream, const char* name) : timer(timer), stream(stream) { if(timer) { timer->begin(kind, stream, name); } }
    ~Timed() { if(timer) { timer->end(stream); } }
};
const char* exact_name(const DevDecoder& host_decoder, size_t histogram_bytes) {
    if(pamld_exact_lanes_supports(host_decoder, histogram_bytes)) {
        return (host_decoder.cross_segments == 2 && !host_decoder.uniform_prior) ? "pamld_exact_lanes_kernel (weighted) + pamld_cross_exact_dense_kernel" : "pamld_exact_lanes_kernel";
    }
    return (host_decoder.cross_segments == 2 && !host_decoder.uniform_prior) ? "pamld_cross_exact_weighted_kernel + pamld_cross_exact_dense_kernel" : "pamld_cross_exact_kernel";
}
}  // namespace

int launch_classifier(const StageArgs& args, const DevDecoder& host_decoder, int32_t mode, int32_t max_blocks, void* stream, LaunchTimer* timer);

int launch_stage(const StageArgs& caller_args, const DevDecoder& host_decoder, int32_t mode, int32_t max_blocks, void* stream, LaunchTimer* timer) {
    if(caller_args.n_read <= 0) { return static_cast<int>(hipSuccess); }
    // a large codec: the classifying kernels leave the classified reads' counters to count_results_kernel
    StageArgs args = caller_args;
    args.defer_counters = (host_decoder.algorithm == 1 || host_decoder.algorithm == 2) && args.result != nullptr &&
                          count_results_supports(host_decoder) ? 1 : 0;
    const int status = launch_classifier(args, host_decoder, mode, max_blocks, stream, timer);
    if(status != 0 || !args.defer_counters) { return status; }
    Timed timed(timer, KERNEL_COUNT, stream, "count_results_kernel");
    return launched([&]() { return launch_count_results(args, host_decoder, max_blocks / GRID_BLOCKS_PER_CU, static_cast<hipStream_t>(stream)); });      // one workgroup per CU
}

int launch_classifier(const StageArgs& args, const DevDecoder& host_decoder, int32_t mode, int32_t max_blocks, void* stream, LaunchTimer* timer) {
    if(args.n_read <= 0) { return static_cast<int>(hipSuccess); }
    const int64_t tiles = (args.n_read + BLOCK - 1) / BLOCK;
    const int64_t cus = max_blocks / GRID_BLOCKS_PER_CU > 0 ? max_blocks / GRID_BLOCKS_PER_CU : 1;
    auto share = [&](int blocks_per_cu) { return std::min<int64_t>(max_blocks, cus * blocks_per_cu); };
    const int64_t default_blocks = share(GRID_SHARE_DEFAULT);
    const dim3 grid(static_cast<unsigned>(tiles < default_blocks ? tiles : default_blocks));
    const size_t histogram_bytes = sizeof(unsigned long long) *
        static_cast<size_t>(STAT_JOB_WORDS + (host_decoder.use_lds_histogram ? (host_decoder.n_barcode + 1) * STAT_WORDS_PER_BIN : STAT_WORDS_PER_BIN));
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch(host_decoder.algorithm) {
        case 1:
            if(stage_uses_worklist(host_decoder, mode)) {
                // tier 1: product form over every read, deferring the ambiguous ones; tier 2: the literal kernel on those
                // both launches use the same grid: literal workgroup g consumes what product form workgroup g deferred
                // a combinatorial barcode set too large for the per barcode product form goes through the factorised one
                // ... and long barcodes that are not combinatorial through the two word product form
                // ... and a codec of a handful of barcodes through the kernel without tables (pamld_tiny.hip)
                const bool tiny = pamld_tiny_supports(host_decoder);
                const bool listed = !tiny && args.result != nullptr && (host_decoder.use_lds_histogram || args.defer_counters) && pamld_near_supports(host_decoder);      // tier 0: neighbour lists, then the product form on what it leaves
                // tier 0 of plates: dual index sets of few words a segment, every word of both segments scored (pamld_plate.hip)
                // (... with its bins in the workgroup, or — 171 to 256 wells — with the counters left to count_results_kernel; a set the
                // streaming kernel of sets takes goes there instead)
                const bool plated = !tiny && !listed && args.result != nullptr && (host_decoder.use_lds_histogram ? !args.defer_counters : args.defer_counters != 0) && pamld_plate_supports(host_decoder) &&
                                    !(args.defer_counters && pamld_cross_near_supports(host_decoder) && pamld_cross_stream_supports(host_decoder));
                const bool paired = !tiny && !listed && !plated && args.result != nullptr && (host_decoder.use_lds_histogram || args.defer_counters) && pamld_pair_supports(host_decoder);      // tier 0 of paired barcodes: one barcode scored
                // ... and where the streaming kernel takes the codec (5 to 10 nt), lists and scan are ONE launch, every
                // wave a pipeline of its own with the deferred reads in an LDS ring (pamld_stream.hpp)
                static const bool no_stream = std::getenv("PHQ_NO_STREAM") != nullptr;
                // (round 4: also a codec past the workgroup histogram — blocks without bins, counters from the records)
                const bool streamed = listed && !no_stream && pamld_stream_supports(host_decoder) &&
                                      (host_decoder.stream_binless ? args.defer_counters != 0 : (host_decoder.use_lds_histogram && !args.defer_counters));
                // ... and a codec of a few barcodes of up to 8 nt has its own streaming kernel: every barcode scored in single
                // precision, the fp64 products of the tiny kernel on the wave's ring (pamld_few.hip)
                static const bool no_few = std::getenv("PHQ_NO_FEW") != nullptr;
                const bool few = tiny && !no_few && args.result != nullptr && !args.defer_counters && pamld_few_supports(host_decoder);
                // (round 6: a two segment set the streaming kernel of sets takes goes to it whatever its size — 400 combinations of 8 nt
                // words 3.3 ms through the product form over all of them, 0.44 through that kernel)
                static const bool no_set_stream = std::getenv("PHQ_NO_STREAM") != nullptr || std::getenv("PHQ_NO_CROSS_STREAM") != nullptr;
                const bool set_streamed = !tiny && !no_set_stream && !plated && args.result != nullptr && args.defer_counters && pamld_cross_supports(host_decoder) &&
                                          pamld_cross_near_supports(host_decoder) && pamld_cross_stream_supports(host_decoder);
                const bool factorised = !tiny && (set_streamed || !pamld_lut_preferred(host_decoder)) && pamld_cross_supports(host_decoder);
                const bool long_words = !tiny && !pamld_lut_supports(host_decoder) && !factorised;
                // tier 0 of a two segment combinatorial set: neighbour lists per segment, the factorised scan on what they leave
                const bool crossed = !plated && factorised && args.result != nullptr && (host_decoder.use_lds_histogram || args.defer_counters) && pamld_cross_near_supports(host_decoder);
                // ... and where the streaming kernel of such sets takes the codec, lists and factorised scan are one launch too
                static const bool no_cross_stream = std::getenv("PHQ_NO_CROSS_STREAM") != nullptr;
                const bool cross_streamed = crossed && !no_stream && !no_cross_stream && args.defer_counters && pamld_cross_stream_supports(host_decoder);
                const int64_t threads = listed ? pamld_near_block(host_decoder) : plated ? pamld_plate_block() : crossed ? pamld_cross_near_block(host_decoder) : (tiny || factorised || paired) ? BLOCK : long_words ? pamld_wide_block(host_decoder) : pamld_lut_block(host_decoder);
                const int64_t wide_tiles = (args.n_read + threads - 1) / threads;
                static const int near_share = std::getenv("PHQ_GRID_SHARE_NEAR") ? std::atoi(std::getenv("PHQ_GRID_SHARE_NEAR")) : GRID_SHARE_NEAR;
                const int64_t product_blocks = plated ? cus : (paired || crossed) ? share(near_share) : (factorised || long_words) ? default_blocks : listed ? share(near_share) : share(GRID_SHARE_LUT);
                const dim3 wide_grid(static_cast<unsigned>(wide_tiles < product_blocks ? wide_tiles : product_blocks));
                const int64_t tiles_per_block = (wide_tiles + wide_grid.x - 1) / wide_grid.x;
                StageArgs producer = args;
                producer.worklist_role = 1;
                producer.worklist_stride = static_cast<int32_t>(tiles_per_block * threads);
                // the plate kernel keeps a segment per WAVE: 12 per CU for the tiers behind it
                const dim3 segment_grid(plated ? wide_grid.x * static_cast<unsigned>(pamld_plate_waves()) : wide_grid.x);
                if(plated) { producer.worklist_stride = static_cast<int32_t>(tiles_per_block * 64); }
                // the deferred reads' inputs travel beside the first quarter of a segment (device_types.hpp, StageArgs::carry)
                producer.carry_stride = (listed && !paired && producer.carry) ? std::max<int32_t>(512, producer.worklist_stride / 4 / 512 * 512) : 0;
                if(producer.carry_stride == 0) { producer.carry = nullptr; }
                int status;
                // the product form kernel (as a consumer or as the first kernel) flags tied noise reads where the ties kernel follows it
                const bool lut_ties = !(streamed || few || cross_streamed) && !tiny && !factorised && !long_words && !crossed && pamld_exact_lanes_ties(args, host_decoder, histogram_bytes);
                if(streamed || few || cross_streamed) {
                    // one workgroup per CU (two of the smaller ones of the few barcode kernel), a worklist segment per WAVE: what
                    // the second pass cannot decide either goes there
                    const int waves = few ? pamld_few_waves() : cross_streamed ? pamld_cross_stream_waves() : pamld_stream_waves();
                    const int64_t wave_tiles = (args.n_read + 63) / 64;
                    const int64_t wanted = std::max<int64_t>(1, std::min<int64_t>(few ? std::min<int64_t>(2 * cus, max_blocks / waves) : cus, (wave_tiles + waves - 1) / waves));
                    const dim3 stream_grid(static_cast<unsigned>(wanted));
                    const int64_t segments = wanted * waves;
                    producer.worklist_stride = static_cast<int32_t>((wave_tiles + segments - 1) / segments * 64);
                    producer.carry = nullptr;
                    producer.carry_stride = 0;
                    {
                        Timed timed(timer, few ? KERNEL_PAMLD_PRODUCT : KERNEL_PAMLD_NEAR, stream, few ? "pamld_few_kernel" : cross_streamed ? "pamld_cross_stream_kernel" : "pamld_stream_kernel");
                        status = launched([&]() { return few ? launch_pamld_few(producer, host_decoder, stream_grid, s)
                               : cross_streamed ? launch_pamld_cross_stream(producer, host_decoder, stream_grid, s) : launch_pamld_stream(producer, host_decoder, stream_grid, s); });
                    }
                    if(status != 0) { return status; }
                    const dim3 segment_grid(static_cast<unsigned>(segments));
                    trace_worklist(few ? "the few barcode kernel" : cross_streamed ? "the streaming kernel of the set" : "the streaming kernel", producer, segment_grid.x, s);
                    StageArgs consumer = producer;
                    consumer.worklist_role = 2;
                    if(cross_streamed && !host_decoder.uniform_prior) {
                        // priors that differ: the streaming kernel ran pass A alone; the weighted scans take the wave's segments
                        StageArgs middle = consumer;
                        middle.worklist_segments = static_cast<int32_t>(segment_grid.x);
                        Timed timed(timer, KERNEL_PAMLD_PRODUCT, stream, "pamld_cross_kernel (weighted, consumer)");
                        status = launched([&]() { return launch_pamld_cross(middle, host_decoder, segment_grid, histogram_bytes, s); });
                        if(status != 0) { return status; }
                        trace_worklist("the product form", producer, segment_grid.x, s);
                    }
                    if(pamld_exact_lanes_supports(host_decoder, histogram_bytes) || (pamld_exact_supports(host_decoder) && pamld_exact_legacy_fits(host_decoder))) {
                        // several lanes per read where the set is one the lane parallel tier takes (pamld_exact_lanes.hip)
                        Timed timed(timer, KERNEL_PAMLD_EXACT, stream, exact_name(host_decoder, histogram_bytes));
                        status = launched([&]() { return pamld_exact_lanes_supports(host_decoder, histogram_bytes) ? launch_pamld_exact_lanes(consumer, host_decoder, segment_grid, histogram_bytes, s)
                                                                                          : launch_pamld_cross_exact(consumer, host_decoder, segment_grid, histogram_bytes, s); });
                        if(status != 0) { return status; }
                        trace_worklist("the exact tier", producer, segment_grid.x, s);
                    }
                    Timed timed(timer, KERNEL_PAMLD_LITERAL, stream, "pamld_literal_kernel");
                    return launched([&]() { return launch_pamld_literal(consumer, host_decoder, segment_grid, 128 * sizeof(double) + histogram_bytes, s); });
                }
                if(listed || paired || crossed || plated) {
                    {
                        Timed timed(timer, KERNEL_PAMLD_NEAR, stream, listed ? "pamld_near_kernel" : plated ? "pamld_plate_kernel" : paired ? "pamld_pair_kernel" : "pamld_cross_near_kernel");
                        status = launched([&]() { return listed ? launch_pamld_near(producer, host_decoder, wide_grid, histogram_bytes, s)
                               : plated ? launch_pamld_plate(producer, host_decoder, wide_grid, histogram_bytes, s)
                               : paired ? launch_pamld_pair(producer, host_decoder, wide_grid, histogram_bytes, s)
                                        : launch_pamld_cross_near(producer, host_decoder, wide_grid, histogram_bytes, s); });
                    }
                    if(status != 0) { return status; }
                    trace_worklist("the neighbour lists", producer, segment_grid.x, s);
                    StageArgs middle = producer;
                    middle.worklist_role = 2;
                    middle.worklist_segments = static_cast<int32_t>(segment_grid.x);
                    middle.flag_ties = lut_ties ? 1 : 0;
                    Timed timed(timer, KERNEL_PAMLD_PRODUCT, stream, (!crossed && !long_words) ? "pamld_lut_kernel (consumer)" : crossed ? (host_decoder.uniform_prior ? "pamld_cross_kernel (consumer)" : "pamld_cross_kernel (weighted, consumer)") : "pamld_wide_kernel (consumer)");
                    if(!crossed && !long_words) {
                        // the product form kernel takes the segments in groups: as many workgroups as stay resident
                        static const int consumer_share = std::getenv("PHQ_GRID_SHARE_CONSUMER") ? std::atoi(std::getenv("PHQ_GRID_SHARE_CONSUMER")) : GRID_SHARE_CONSUMER;
                        const int64_t wanted = std::max<int64_t>(1, share(consumer_share));
                        middle.worklist_group = static_cast<int32_t>((segment_grid.x + wanted - 1) / wanted);
                        const dim3 fewer(static_cast<unsigned>((segment_grid.x + middle.worklist_group - 1) / middle.worklist_group));
                        status = launched([&]() { return launch_pamld_lut(middle, host_decoder, fewer, histogram_bytes, s); });
                    } else {
                        status = launched([&]() { return crossed ? launch_pamld_cross(middle, host_decoder, wide_grid, histogram_bytes, s)
                                         : launch_pamld_wide(middle, host_decoder, wide_grid, histogram_bytes, s); });
                    }
                } else {
                    producer.flag_ties = lut_ties ? 1 : 0;
                    Timed timed(timer, KERNEL_PAMLD_PRODUCT, stream, tiny ? "pamld_tiny_kernel" : factorised ? (host_decoder.uniform_prior ? "pamld_cross_kernel" : "pamld_cross_kernel (weighted)") : long_words ? "pamld_wide_kernel" : "pamld_lut_kernel");
                    status = launched([&]() { return tiny ? launch_pamld_tiny(producer, host_decoder, wide_grid, histogram_bytes, s)
                           : factorised ? launch_pamld_cross(producer, host_decoder, wide_grid, histogram_bytes, s)
                           : long_words ? launch_pamld_wide(producer, host_decoder, wide_grid, histogram_bytes, s)
                                        : launch_pamld_lut(producer, host_decoder, wide_grid, histogram_bytes, s); });
                }
                if(status != 0) { return status; }
                trace_worklist("the product form", producer, segment_grid.x, s);
                StageArgs consumer = producer;
                consumer.worklist_role = 2;
                consumer.flag_ties = lut_ties ? 1 : 0;             // (the lane parallel exact tier follows: pamld_exact_lanes_ties)
                if(factorised || (!long_words && (pamld_exact_lanes_supports(host_decoder, histogram_bytes) || (pamld_exact_supports(host_decoder) && pamld_exact_legacy_fits(host_decoder))))) {
                    // tier 1.5: exact but still factorised; leaves in the worklist only what needs the reference's own sums
                    Timed timed(timer, KERNEL_PAMLD_EXACT, stream, exact_name(host_decoder, histogram_bytes));
                    status = launched([&]() { return pamld_exact_lanes_supports(host_decoder, histogram_bytes) ? launch_pamld_exact_lanes(consumer, host_decoder, segment_grid, histogram_bytes, s)
                                                                                      : launch_pamld_cross_exact(consumer, host_decoder, segment_grid, histogram_bytes, s); });
                    if(status != 0) { return status; }
                } else if(long_words && pamld_wide_exact_supports(host_decoder)) {
                    // the same for two word barcodes (pamld_wide_exact.hpp)
                    Timed timed(timer, KERNEL_PAMLD_EXACT, stream, "pamld_wide_exact_kernel");
                    status = launched([&]() { return launch_pamld_wide_exact(consumer, host_decoder, segment_grid, histogram_bytes, s); });
                    if(status != 0) { return status; }
                }
                trace_worklist("the exact tier", producer, segment_grid.x, s);
                Timed timed(timer, KERNEL_PAMLD_LITERAL, stream, "pamld_literal_kernel");
                return launched([&]() { return launch_pamld_literal(consumer, host_decoder, segment_grid, 128 * sizeof(double) + histogram_bytes, s); });
            }
            {
                Timed timed(timer, KERNEL_PAMLD_LITERAL, stream, host_decoder.nucleotide_cardinality > DEV_TIER_NUCLEOTIDE ? "pamld_literal_long_kernel (every read)" : "pamld_literal_kernel (every read)");
                return launched([&]() { return launch_pamld_literal(args, host_decoder, grid, 128 * sizeof(double) + histogram_bytes, s); });
            }
        case 2: {
            const bool general = host_decoder.nucleotide_cardinality > DEV_TIER_NUCLEOTIDE;
            const bool paired = !general && mdd_pair_supports(host_decoder);
            const bool crossed = !general && !paired && mdd_cross2_supports(host_decoder);
            if(mode == MODE_FAST && args.result != nullptr && (paired || crossed || mdd_near_supports(host_decoder)) && stage_uses_worklist(host_decoder, mode)) {
                // one table lookup per read — per segment for a paired codec — (mdd_near.hpp); the scan kernel on the few reads it
                // cannot finish (same grid)
                const int64_t near_tiles = (args.n_read + 255) / 256;
                const int64_t near_blocks = share(GRID_SHARE_NEAR);
                const dim3 near_grid(static_cast<unsigned>(near_tiles < near_blocks ? near_tiles : near_blocks));
                StageArgs producer = args;
                producer.worklist_role = 1;
                producer.worklist_stride = static_cast<int32_t>((near_tiles + near_grid.x - 1) / near_grid.x * 256);
                int status;
                {
                    Timed timed(timer, KERNEL_MDD_EXACT, stream, paired ? "mdd_lookup_kernel (paired)" : crossed ? "mdd_lookup_kernel (combinatorial)" : "mdd_lookup_kernel");
                    status = launched([&]() { return paired ? launch_mdd_pair(producer, host_decoder, near_grid, histogram_bytes, s)
                           : crossed ? launch_mdd_cross2(producer, host_decoder, near_grid, histogram_bytes, s)
                                     : launch_mdd_near(producer, host_decoder, near_grid, histogram_bytes, s); });
                }
                if(status != 0) { return status; }
                StageArgs consumer = producer;
                consumer.worklist_role = 2;
                Timed timed(timer, KERNEL_MDD, stream, (crossed && mdd_cross_supports(host_decoder)) ? "mdd_cross_kernel" : mdd_scan_supports(host_decoder) ? "mdd_scan_kernel" : "mdd_kernel");
                if(crossed && mdd_cross_supports(host_decoder)) { return launched([&]() { return launch_mdd_cross(consumer, host_decoder, near_grid, histogram_bytes, s); }); }
                if(mdd_scan_supports(host_decoder)) { return launched([&]() { return launch_mdd_scan(consumer, host_decoder, near_grid, histogram_bytes, s); }); }
                return launched([&]() { return launch_mdd(consumer, host_decoder, near_grid, histogram_bytes, s); });
            }
            if(mode == MODE_FAST && !general && mdd_fused_supports(host_decoder)) {
                // both tiers in one kernel: the deferred reads wait in an LDS ring, nothing goes through HBM twice
                Timed timed(timer, KERNEL_MDD, stream, "mdd_fused_kernel");
                const int64_t fused_blocks = share(GRID_SHARE_MDD_FUSED);
                const dim3 fused_grid(static_cast<unsigned>(tiles < fused_blocks ? tiles : fused_blocks));
                return launched([&]() { return launch_mdd_fused(args, host_decoder, fused_grid, histogram_bytes, s); });
            }
            if(stage_uses_worklist(host_decoder, mode)) {
                // tier 1: exact match lookup over every read; tier 2: the scan kernel on what it deferred (same grid)
                // the exact match kernel takes 4 tiles per pass, so a workgroup may own up to 3 tiles more than its share
                const int64_t tiles_per_block = (tiles + 4 * grid.x - 1) / (4 * grid.x) * 4;
                StageArgs producer = args;
                producer.worklist_role = 1;
                producer.worklist_stride = static_cast<int32_t>(tiles_per_block * BLOCK);
                int status;
                {
                    Timed timed(timer, KERNEL_MDD_EXACT, stream, "mdd_exact_kernel");
                    status = launched([&]() { return launch_mdd_exact(producer, host_decoder, grid, histogram_bytes, s); });
                }
                if(status != 0) { return status; }
                StageArgs consumer = producer;
                consumer.worklist_role = 2;
                Timed timed(timer, KERNEL_MDD, stream, mdd_scan_supports(host_decoder) ? "mdd_scan_kernel" : mdd_cross_supports(host_decoder) ? "mdd_cross_kernel" : "mdd_kernel");
                if(mdd_scan_supports(host_decoder)) { return launched([&]() { return launch_mdd_scan(consumer, host_decoder, grid, histogram_bytes, s); }); }
                if(mdd_cross_supports(host_decoder)) { return launched([&]() { return launch_mdd_cross(consumer, host_decoder, grid, histogram_bytes, s); }); }
                return launched([&]() { return launch_mdd(consumer, host_decoder, grid, histogram_bytes, s); });
            }
            Timed timed(timer, KERNEL_MDD, stream, "mdd_kernel (every read)");
            return launched([&]() { return launch_mdd(args, host_decoder, grid, histogram_bytes, s); });
        }
        default: {
            Timed timed(timer, KERNEL_COUNT, stream, "count_kernel");
            return launched([&]() { return launch_count(args, host_decoder, grid, histogram_bytes, s); });
        }
    }
}

}  // namespace phq
