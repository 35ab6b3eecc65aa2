// K1t, tied noise reads (round 5): the argmax of a read every barcode of which is clearly under the noise filter, from the
// literal sums of the sixteen barcodes that can hold it.
//
//   reference pamld.cpp:61-76     the argmax: strict '>' over prior * P(r|b) in barcode order
//   reference pamld.cpp:96,113-119  a read under the noise filter is charged to its argmax and leaves unclassified
//   reference barcode.h:131-164   P(r|b) = pow(base, Kahan sum of the Phred costs)
//
// The product form kernel (pamld_lut.hpp, NEAR_TOP) hands such a read on with bit 31 of its worklist entry set and, in the bits
// of its placeholder record's confidence, the block of eight barcodes that holds the single precision maximum and the block that
// holds the runner up: every other barcode is under 0.9998 of the maximum in single precision, further than the floats blur.  A
// third to a half of the noise reads tie (two barcodes at the same distance with the same qualities under their mismatches are
// the same product to a float's last bit): 2.1 % of all of H94's reads, which the exact tier behind this kernel scanned the
// whole codec for (pamld_exact_lanes.hip: 1.04 ms of H94's 4.3).
//
// Flagged entries are few in most tiles of a segment and a segment holds a few hundred entries, so the kernel is laid out for
// latency, not for throughput: a WAVE owns a segment (a workgroup per segment was measured first: 3 072 workgroups of one tile
// and one barrier chain each, four rounds of them at the three a CU holds, 0.11 ms with next to nothing to do).  Tile by tile the
// wave reads its segment, compacts the entries WITHOUT the bit to the segment's front (in order: the exact tier's, unchanged) and
// queues the flagged ones in its corner of LDS; whenever a wave's worth is queued, every lane takes one: the Kahan recurrence and
// the careful 2^x (kernel_common.hpp) of the members of the read's two blocks, four members side by side, compared as the
// reference compares them.  A read that turns out over the noise filter on the literal value goes back to the segment without
// the bit.  Nothing is shared between the waves of a workgroup but the counter histogram.
#include <cstdlib>

#include "kernel_common.hpp"

namespace phq {
namespace ties {

constexpr int WAVES = 4;                   // a segment each
constexpr int THREADS = 64 * WAVES;
constexpr int FOLD = 8;                    // pamld_lut.hpp's blocks
constexpr int MOST_BARCODES = 2048;        // the codec's 4 bit words in LDS: 16 KiB at most
constexpr int PACE = 4;                    // members of a read's blocks walked side by side
constexpr int AHEAD = 4;                   // entries a lane reads of a tile: their loads are in flight together

// G = words of a read's row (four positions each): 5 .. 8 nt, 9 .. 12, 13 .. 16
template <int G>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(3, 4))) void pamld_ties_kernel(StageArgs args) {
    constexpr int W = (G + 1) / 2;           // 4 bit codes: eight positions a word (device_types.hpp, barcode_packed)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* const tpq = reinterpret_cast<double*>(lds_raw);
    uint32_t* const packed = reinterpret_cast<uint32_t*>(lds_raw + 128 * sizeof(double));
    DecoderRef d = *constant_cast(args.decoder);
    const int n_barcode = d.n_barcode;
    unsigned long long* const histogram = reinterpret_cast<unsigned long long*>(lds_raw + 128 * sizeof(double) + ((static_cast<size_t>(n_barcode) * W * 4 + 15) & ~static_cast<size_t>(15)));
    __shared__ uint32_t queue_of[WAVES][(AHEAD + 1) * 64];

    // PhredScale in LDS as the literal kernel keeps it (phred.cpp:44-68): 128 true positive qualities; a mismatch costs the
    // Phred value itself, a base outside ACGT the constant 10 log10 4, nothing at quality 0
    for(int i = threadIdx.x; i < 128; i += THREADS) { tpq[i] = d.true_positive_quality[i]; }
    for(int i = threadIdx.x; i < n_barcode * W; i += THREADS) { packed[i] = d.barcode_packed[i]; }
    histogram_clear(d, histogram);
    __syncthreads();

    const StatSink sink{histogram, args.stats, args.defer_counters != 0};
    const int L = d.nucleotide_cardinality;
    const double uniform = d.uniform_base_quality;
    const double common_prior = d.concentration[0];
    const int wave = static_cast<int>(threadIdx.x >> 6);
    const int lane = static_cast<int>(threadIdx.x & 63u);
    const int64_t own = static_cast<int64_t>(blockIdx.x) * WAVES + wave;             // this wave's segment
    const bool present = own < args.worklist_segments;
    uint32_t* const segment = args.worklist + (present ? own : 0) * args.worklist_stride;
    const int64_t n = present ? static_cast<int64_t>(args.worklist_count[own]) : 0;
    const int64_t tiles = (n + AHEAD * 64 - 1) / (AHEAD * 64);
    uint32_t* const queue = queue_of[wave];
    uint32_t queued = 0;                       // wave uniform, as is
    uint32_t kept_total = 0;                   // what the wave has left in its segment

    // what stays in the worklist, to the front, without the bit
    auto keep = [&](bool stays, uint32_t entry) {
        const unsigned long long staying = __ballot(stays);
        if(stays) { segment[kept_total + __popcll(staying & ((1ull << lane) - 1ull))] = entry & 0x7fffffffu; }
        kept_total += static_cast<uint32_t>(__popcll(staying));
    };
    // queue[first .. first + count): a read a lane
    auto settle = [&](uint32_t first, uint32_t count) {
        const bool active = static_cast<uint32_t>(lane) < count;
        const uint32_t entry = queue[first + (active ? static_cast<uint32_t>(lane) : 0u)];
        const int64_t i = static_cast<int64_t>(entry);
        ReadState r{false, 0};
        bool settled = false;
        if(active) {
            r = load_read_state(args, i);
            const uint32_t blocks = static_cast<uint32_t>(__double2loint(args.result[i].confidence));
            const uint32_t* const code_row = reinterpret_cast<const uint32_t*>(args.base + i * (4 * G));
            const uint32_t* const quality_row = reinterpret_cast<const uint32_t*>(args.quality + i * (4 * G));
            uint32_t observed[4 * G];
            double match_cost[4 * G];
            double mismatch_cost[4 * G];
#pragma unroll
            for(int w = 0; w < G; ++w) {
                const uint32_t c4 = code_row[w];
                const uint32_t q4 = quality_row[w];
#pragma unroll
                for(int k = 0; k < 4; ++k) {
                    const uint32_t c = (c4 >> (8 * k)) & 0xffu;
                    const uint32_t q = (q4 >> (8 * k)) & 0x7fu;          // the table has 128 rows
                    observed[4 * w + k] = c;
                    match_cost[4 * w + k] = tpq[q];
                    mismatch_cost[4 * w + k] = strict_code(static_cast<uint8_t>(c)) ? static_cast<double>(q) : (q ? uniform : 0.0);
                }
            }
            const int block_of[2] = {static_cast<int>(blocks & 0x1fffu), static_cast<int>((blocks >> 13) & 0x1fffu)};
            const int members = ((blocks >> 26) & 1u) ? 2 * FOLD : FOLD;
            int best = 0x7fffffff;
            double best_p = -1.0;
            double best_conditional = 0;
            // four members at a time: a member's walk is one chain of dependent fp64 operations, four chains fill the pipe
#pragma nounroll
            for(int m = 0; m < members; m += PACE) {
                int b[PACE];
                uint32_t expected[PACE][W];
                double sigma_q[PACE];
                double compensation[PACE];
#pragma unroll
                for(int u = 0; u < PACE; ++u) {
                    b[u] = FOLD * block_of[(m + u) / FOLD] + ((m + u) % FOLD);
                    const int from = b[u] < n_barcode ? b[u] : 0;
#pragma unroll
                    for(int w = 0; w < W; ++w) { expected[u][w] = packed[from * W + w]; }
                    sigma_q[u] = 0;
                    compensation[u] = 0;
                }
                // barcode.h:131-164: the Kahan sum of the costs over the barcode's positions, then pow
#pragma unroll
                for(int j = 0; j < 4 * G; ++j) {
                    if(j < L) {
#pragma unroll
                        for(int u = 0; u < PACE; ++u) {
                            const uint32_t e = (expected[u][j >> 3] >> (4 * (j & 7))) & 0xfu;
                            const double x = (observed[j] == e) ? match_cost[j] : mismatch_cost[j];
                            const double y = x - compensation[u];
                            const double t = sigma_q[u] + y;
                            compensation[u] = (t - sigma_q[u]) - y;
                            sigma_q[u] = t;
                        }
                    }
                }
#pragma unroll
                for(int u = 0; u < PACE; ++u) {
                    const double conditional = phred_probability(sigma_q[u], d.log2_base_hi, d.log2_base_lo);
                    const double p = conditional * common_prior;                           // pamld.cpp:66
                    if(b[u] < n_barcode && (p > best_p || (p == best_p && b[u] < best))) {  // strict '>' in barcode order: the lowest of equals
                        best = b[u]; best_p = p; best_conditional = conditional;
                    }
                }
            }
            // a noise read on the literal value as well (pamld.cpp:96): charged to its argmax, it leaves unclassified
            settled = best != 0x7fffffff && !(best_conditional > d.random_barcode_probability);
            if(settled) {
                stat_add(d, sink, best + 1, 0, FIELD_LOW_CONDITIONAL_CONFIDENCE, 1ull);
                r.qcfail = true;
                decoder_tail(d, sink, r, 0, 0);
                DevResult out;
                out.index = 0; out.distance = 0; out.hq_distance = 0; out.status = STATUS_NOISE; out.qcfail = 1; out.confidence = 0;
                args.result[i] = out;
                store_read_state(args, i, r);
            }
        }
        keep(active && !settled, entry);
        count_outgoing(args, histogram, active && settled, r);
    };

    for(int64_t tile = 0; tile < tiles; ++tile) {
        uint32_t entry[AHEAD];
        bool active[AHEAD];
#pragma unroll
        for(int e = 0; e < AHEAD; ++e) {
            const int64_t k = (tile * AHEAD + e) * 64 + lane;
            active[e] = k < n;
            entry[e] = active[e] ? segment[k] : 0u;
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);    // every entry of the tile is read before anything is written to the segment's front
#pragma unroll
        for(int e = 0; e < AHEAD; ++e) {
            const bool flagged = active[e] && (entry[e] >> 31) != 0;
            keep(active[e] && !flagged, entry[e]);
            const unsigned long long queueing = __ballot(flagged);
            if(flagged) { queue[queued + __popcll(queueing & ((1ull << lane) - 1ull))] = entry[e] & 0x7fffffffu; }
            queued += static_cast<uint32_t>(__popcll(queueing));
        }
        __builtin_amdgcn_wave_barrier();       // (a wave's LDS operations complete in order)
        // (under a wave's worth was waiting, a tile brings AHEAD of them at most)
        while(queued >= 64u) {
            queued -= 64u;
            settle(queued, 64u);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if(queued > 0) { settle(0, queued); }
    if(present && lane == 0) { args.worklist_count[own] = kept_total; }
    __syncthreads();
    histogram_flush(args, d, histogram);
}

}  // namespace ties

// behind the product form kernel of a single segment codec of 5 to 16 nt with uniform priors, all ACGT, whose reads' ordinals
// leave bit 31 of a worklist entry free and whose blocks' numbers fit their thirteen bits
bool pamld_ties_supports(const StageArgs& args, const DevDecoder& host_decoder) {
    static const bool off = std::getenv("PHQ_NO_TIES") != nullptr;
    return !off && host_decoder.algorithm == 1 && host_decoder.all_strict && host_decoder.uniform_prior && host_decoder.nucleotide_cardinality >= 5 &&
           host_decoder.nucleotide_cardinality <= 16 && host_decoder.n_barcode <= ties::MOST_BARCODES && args.n_read < (int64_t(1) << 31) && args.result != nullptr;
}

// `grid` = the number of worklist segments, a wave each
int launch_pamld_ties(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    const int G = (host_decoder.nucleotide_cardinality + 3) / 4;
    const size_t words = static_cast<size_t>(host_decoder.n_barcode) * static_cast<size_t>((G + 1) / 2) * 4;
    const size_t lds_bytes = 128 * sizeof(double) + ((words + 15) & ~static_cast<size_t>(15)) + histogram_bytes;
    StageArgs by_wave = args;
    by_wave.worklist_segments = static_cast<int32_t>(grid.x);
    const dim3 fewer((grid.x + ties::WAVES - 1) / ties::WAVES);
    switch(G) {
        case 2: hipLaunchKernelGGL((ties::pamld_ties_kernel<2>), fewer, dim3(ties::THREADS), lds_bytes, stream, by_wave); break;
        case 3: hipLaunchKernelGGL((ties::pamld_ties_kernel<3>), fewer, dim3(ties::THREADS), lds_bytes, stream, by_wave); break;
        case 4: hipLaunchKernelGGL((ties::pamld_ties_kernel<4>), fewer, dim3(ties::THREADS), lds_bytes, stream, by_wave); break;
        default: return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}

}  // namespace phq
