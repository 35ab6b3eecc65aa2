// K2n (lookups): MdDecoder<T>::classify as one table lookup per barcode segment.
//
//   reference mdd.cpp:37-86   exact match first (the observation as a key of element_by_sequence), else the barcodes in
//                             index order, the FIRST within every segment's tolerance wins; nothing within: unclassified, QC fail
//   reference sequence.h:90-98, 321-332   distance_from / masked_distance_from
//
// A distance tolerance never exceeds the Shannon bound of its segment's words (the configuration compiler refuses more), so
// at most ONE word of a segment is within the tolerance of what was observed there, and the whole loop is a function of the
// observed words: per segment the word itself if it is one of the segment's words, else the one within the tolerance (the
// lowest ordinal the loop meets), else nothing.  That function is the table the PAMLD tiers already have — pamld_near.hpp's
// (one mismatch) for single segment and combinatorial sets, pamld_pair.hpp's (two mismatches) for paired ones: 4^length
// entries of 16 bits in global memory, the few thousand keys real reads carry stay in L1/L2.  Three kinds of codec:
//   SINGLE         one segment of 5 to 10 nt, tolerance 0 or 1: the table names the barcode;
//   PAIRED         two segments of 4 to 10 nt whose words belong to ONE barcode each (unique dual indexes, ligation + RT
//                  pairs), tolerances up to 2: a barcode is within both tolerances only if both segments name it;
//   COMBINATORIAL  two segments of 5 to 10 nt, every combination of their words a barcode, numbered in lexicographic
//                  order of the word ordinals (DevDecoder::cross_identity), tolerances 0 or 1: each segment names its word,
//                  the barcode is their combination.
// A read costs its code words (the quality plane is only read when a masking threshold is set), the lookups, the named
// words' own bits from LDS for the distances, and one record.  No hash probe, no scan, no ring: the kernel streams, with the
// prefetch of pamld_near.hpp.  What is not a function of the words goes on to the scan kernels (mdd_scan.hip,
// mdd_cross.hip, mdd.hip) through the workgroup's worklist segment: a base outside ACGT, a read shorter than the barcode, a
// quality below the masking threshold in a read that is not a barcode word for word (the exact match comes before the
// masking, mdd.cpp:43-46).  A table miss IS final: nothing within the tolerance.
#pragma once

#include "kernel_common.hpp"

namespace phq {
namespace mdd_near {

constexpr int THREADS = 256;
enum Kind : int { SINGLE = 0, PAIRED = 1, COMBINATORIAL = 2 };

// LDS: the worklist counter, then what the distances are taken against — SINGLE, PAIRED: one record of 16 bytes per
// barcode {low plane, high plane, -, barcode index} (bit j = code bit of position j of the whole barcode);
// COMBINATORIAL: {low plane, high plane} of every word of segment 0, then of segment 1 — then the histogram
struct Layout {
    static constexpr uint32_t DEFERRED_AT = 0;
    static constexpr uint32_t RECORD_AT = 16;
    __host__ __device__ static constexpr uint32_t histogram_at(uint32_t record_bytes) { return (RECORD_AT + record_bytes + 15u) & ~15u; }
};
inline uint32_t record_bytes_of(const DevDecoder& d, int kind) {
    return kind == COMBINATORIAL ? static_cast<uint32_t>(d.cross_unique[0] + d.cross_unique[1]) * 8u : static_cast<uint32_t>(d.n_barcode) * 16u;
}

typedef uint32_t Quad __attribute__((ext_vector_type(4)));
typedef uint32_t Duo __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const Quad* LdsQuads;
typedef __attribute__((address_space(3))) const Duo* LdsDuos;

// G = words of the plane row (4 positions each); STATE as in pamld_near.hpp (where the carried state comes from);
// MASKING: a quality masking threshold is set
template <int G, int STATE, bool MASKING, int KIND>
__global__ __launch_bounds__(THREADS) void mdd_lookup_kernel(StageArgs args) {
    constexpr bool TWO = KIND != SINGLE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    uint32_t& deferred_total = *reinterpret_cast<uint32_t*>(lds_raw + Layout::DEFERRED_AT);
    uint32_t* const records = reinterpret_cast<uint32_t*>(lds_raw + Layout::RECORD_AT);
    DecoderRef d = *constant_cast(args.decoder);
    const uint32_t record_bytes = KIND == COMBINATORIAL ? static_cast<uint32_t>(d.cross_unique[0] + d.cross_unique[1]) * 8u : static_cast<uint32_t>(d.n_barcode) * 16u;
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw + Layout::histogram_at(record_bytes));

    if(threadIdx.x == 0) { deferred_total = 0; }
    {
        const uint32_t* const source = KIND == SINGLE ? d.mdd_near_record : KIND == PAIRED ? d.pair_record : d.mdd_cross_word;
        const uint32_t words = record_bytes / 4u;
        for(uint32_t w = threadIdx.x; w < words; w += THREADS) {
            // a paired codec's records are pamld_pair.hpp's (PAIR_RECORD_WORDS each): their first four words
            records[w] = KIND == PAIRED ? source[(w >> 2) * PAIR_RECORD_WORDS + (w & 3u)] : source[w];
        }
    }
    histogram_clear(d, histogram);
    __syncthreads();

    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const int L = d.nucleotide_cardinality;
    // where the segments sit in the barcode, their tables, their tolerances
    const uint16_t* const nearest0 = KIND == SINGLE ? d.near_table : KIND == PAIRED ? d.pair_table[0] : d.cross_near_table[0];
    const uint16_t* const nearest1 = KIND == SINGLE ? d.near_table : KIND == PAIRED ? d.pair_table[1] : d.cross_near_table[1];
    const uint32_t start0 = KIND == SINGLE ? 0u : static_cast<uint32_t>(KIND == PAIRED ? d.pair_start[0] : d.cross_start[0]);
    const uint32_t start1 = KIND == SINGLE ? 0u : static_cast<uint32_t>(KIND == PAIRED ? d.pair_start[1] : d.cross_start[1]);
    const uint32_t length0 = KIND == SINGLE ? static_cast<uint32_t>(L) : static_cast<uint32_t>(KIND == PAIRED ? d.pair_length[0] : d.cross_length[0]);
    const uint32_t length1 = KIND == SINGLE ? 0u : static_cast<uint32_t>(KIND == PAIRED ? d.pair_length[1] : d.cross_length[1]);
    const uint32_t mask0 = (1u << length0) - 1u, mask1 = (1u << length1) - 1u;
    const bool parts = KIND == PAIRED && d.mdd_pair_parts != 0;
    const int tolerance0 = d.distance_tolerance[0], tolerance1 = (TWO && !parts) ? d.distance_tolerance[1] : 0;
    const uint32_t unique0 = KIND == COMBINATORIAL ? static_cast<uint32_t>(d.cross_unique[0]) : static_cast<uint32_t>(d.n_barcode);
    const uint32_t unique1 = KIND == COMBINATORIAL ? static_cast<uint32_t>(d.cross_unique[1]) : static_cast<uint32_t>(d.n_barcode);
    const uint32_t stride0 = KIND == COMBINATORIAL ? static_cast<uint32_t>(d.cross_stride[0]) : 0u;
    const uint32_t stride1 = KIND == COMBINATORIAL ? static_cast<uint32_t>(d.cross_stride[1]) : 0u;
    const bool identity_index = d.identity_index != 0;
    const int64_t n = args.n_read;
    const int64_t tiles = (n + THREADS - 1) / THREADS;
    const uint32_t masking_addend = static_cast<uint32_t>(128 - min(max(d.quality_masking_threshold, 0), 128)) * 0x01010101u;
    const bool multiplexing = d.multiplexing != 0;
    const bool lds_bins = d.use_lds_histogram != 0;
    const bool defer_counters = args.defer_counters != 0;
    // the bytes of the positions a word holds (word w: positions 4 w .. 4 w + 3)
    uint32_t valid[G];
#pragma unroll
    for(int w = 0; w < G; ++w) { valid[w] = L - 4 * w >= 4 ? 0xffffffffu : (L - 4 * w <= 0 ? 0u : ((1u << (8 * (L - 4 * w))) - 1u)); }

    // a tile's inputs one tile ahead, issued behind the table lookups (pamld_near.hpp)
    struct Inputs { uint32_t code[G]; uint32_t quality[G]; uint32_t state; };
    auto fetch = [&](int64_t tile, Inputs& in) {
        const int64_t first = tile * THREADS;
        const uint32_t last = static_cast<uint32_t>(min(static_cast<int64_t>(THREADS), n - first)) - 1u;
        const uint32_t at = min(threadIdx.x, last);
        const uint32_t* const code_row = reinterpret_cast<const uint32_t*>(args.base + first * (4 * G) + at * (4u * G));
#pragma unroll
        for(int w = 0; w < G; ++w) { in.code[w] = code_row[w]; }
        if(MASKING) {
            const uint32_t* const quality_row = reinterpret_cast<const uint32_t*>(args.quality + first * (4 * G) + at * (4u * G));
#pragma unroll
            for(int w = 0; w < G; ++w) { in.quality[w] = quality_row[w]; }
        }
        if(STATE == 1) { in.state = static_cast<uint32_t>(args.input_flag[first + at]); }      // transcode.h:207-211
        else if(STATE == 2) { in.state = args.word[first + at]; }
        else { in.state = 0; }
    };
    Inputs next;
    fetch(min(static_cast<int64_t>(blockIdx.x), tiles - 1), next);
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0) before the loop (pamld_near.hpp)
    uint32_t pending_slot = 0xffffffffu;
    uint32_t pending_read = 0;
    for(int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const Inputs in = next;
        const int64_t first = tile * THREADS;
        const bool active = first + threadIdx.x < n;
        const uint32_t at = active ? threadIdx.x : static_cast<uint32_t>(n - 1 - first);
        const int64_t i = first + at;
        ReadState r;
        if(STATE != 2) { r.qcfail = (in.state & 1u) != 0; r.channel = 0; }
        else { r.qcfail = (in.state >> 31) != 0; r.channel = static_cast<int32_t>(in.state & 0x7fffffffu); }

        // A, C, G or T = exactly one bit of the nibble (iupac.h:51-56); anything else — an absent position too — and the
        // read goes to the scan.  2-bit codes: the low code bit is bit 1 | bit 3 of the nibble, the high one bit 2 | bit 3.
        auto gather = [](uint32_t planes) {            // bits 8 j and 8 j + 4 -> bits j and 4 + j
            const uint32_t t = planes | (planes >> 7);
            return (t | (t >> 14)) & 0xffu;
        };
        uint32_t low = 0, high = 0;            // bit j = position j
        uint32_t unfinished = 0;
        uint32_t masked = 0;
#pragma unroll
        for(int w = 0; w < G; ++w) {
            const uint32_t x = in.code[w] & (0x0f0f0f0fu & valid[w]);
            const uint32_t z = x & ((x | 0x80808080u) - 0x01010101u);
            const uint32_t strict = (x + 0x7f7f7f7fu) & ~(z + 0x7f7f7f7fu) & 0x80808080u;
            unfinished |= (strict ^ (0x80808080u & valid[w])) | (in.code[w] & (0xf0f0f0f0u & valid[w]));      // a code above 15 is not a base either
            const uint32_t both = gather((((x >> 1) | (x >> 3)) & 0x01010101u) | (((x << 2) | (x << 1)) & 0x10101010u));
            low |= (both & 0xfu) << (4 * w);
            high |= (both >> 4) << (4 * w);
            if(MASKING) {
                // quality below the threshold (0..128): bit 7 stays clear after adding 128 - threshold; Q >= 128 is never below
                const uint32_t q7 = in.quality[w] & 0x7f7f7f7fu;
                masked |= ~((q7 + masking_addend) | in.quality[w]) & (0x80808080u & valid[w]);
            }
        }
        const uint32_t low0 = (low >> start0) & mask0, high0 = (high >> start0) & mask0;
        const uint32_t low1 = (low >> start1) & mask1, high1 = (high >> start1) & mask1;
        if(pending_slot != 0xffffffffu) { segment[pending_slot] = pending_read; }      // the previous tile's deferred reads
        const uint32_t named0 = nearest0[low0 | (high0 << length0)];                   // what this word is (next to), if anything
        const uint32_t named1 = TWO ? static_cast<uint32_t>(nearest1[low1 | (high1 << length1)]) : named0;
        fetch(min(tile + gridDim.x, tiles - 1), next);
        // (parts: a single index of 11 to 20 nt as two parts of its one segment — either part's word may name the barcode, a word
        // several barcodes share names no one, PAIR_ANY_BARCODE; the tolerance is the whole barcode's)
        const bool known = KIND == COMBINATORIAL ? (named0 < unique0 && named1 < unique1)
                         : parts ? (named0 < unique0 || named1 < unique0) : (named0 == named1 && named0 < unique0);
        const uint32_t u0 = known ? ((parts && !(named0 < unique0)) ? named1 : named0) : 0u, u1 = known ? named1 : 0u;
        int k0, k1 = 0;
        uint32_t c;                            // the barcode's ordinal
        uint32_t barcode_index;
        if(KIND == COMBINATORIAL) {
            const Duo word0 = *(LdsDuos)static_cast<uintptr_t>(Layout::RECORD_AT + u0 * 8u);
            const Duo word1 = *(LdsDuos)static_cast<uintptr_t>(Layout::RECORD_AT + (unique0 + u1) * 8u);
            k0 = __popc((low0 ^ word0.x) | (high0 ^ word0.y));
            k1 = __popc((low1 ^ word1.x) | (high1 ^ word1.y));
            c = u0 * stride0 + u1 * stride1;                                          // lexicographic numbering
            barcode_index = identity_index ? c + 1u : static_cast<uint32_t>(d.barcode_index[c]);
        } else {
            const Quad record = *(LdsQuads)static_cast<uintptr_t>(Layout::RECORD_AT + u0 * 16u);
            const uint32_t mismatches = (low ^ record.x) | (high ^ record.y);
            k0 = __popc(mismatches & (mask0 << start0));
            if(TWO) { k1 = __popc(mismatches & (mask1 << start1)); }
            c = u0;
            barcode_index = record.w;
        }
        const bool classified = known && (parts ? k0 + k1 <= tolerance0 : (k0 <= tolerance0 && k1 <= tolerance1));          // mdd.cpp:43-46 (no mismatch), 47-80
        // the exact match comes before any masking (mdd.cpp:43-46): a masked position only matters to a read that is not a barcode.
        // Parts: the tables name the barcode of the nearest word of a part, which need not be the one barcode within the tolerance
        // of the whole read — only a read that IS classified is finished here, the scan takes the others.
        const bool defer = unfinished != 0 || (MASKING && masked != 0 && !(known && k0 + k1 == 0)) || (parts && !classified);
        const uint32_t index = classified ? barcode_index : 0u;
        const uint32_t distance = classified ? static_cast<uint32_t>(k0 + k1) : 0u;
        const bool qcfail = r.qcfail || !classified;                   // mdd.cpp:82-84
        const bool qcfail_out = defer ? r.qcfail : qcfail;
        {
            // every lane stores a record and a status word in straight line code (pamld_near.hpp): a deferred read's are
            // written again by the scan
            typedef uint32_t vector4 __attribute__((ext_vector_type(4)));
            vector4 out;
            out.x = index;
            out.y = distance | ((classified ? STATUS_PASS : STATUS_UNCLASSIFIED) << 16) | (qcfail_out ? 1u << 24 : 0u);
            out.z = 0u;
            out.w = 0u;
            if(args.result) { __builtin_nontemporal_store(out, reinterpret_cast<vector4*>(reinterpret_cast<char*>(args.result) + first * 16 + at * 16u)); }
            const uint32_t channel = (!defer && multiplexing) ? index : static_cast<uint32_t>(r.channel);
            __builtin_nontemporal_store((channel & 0x7fffffffu) | (qcfail_out ? 0x80000000u : 0u), reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(args.word) + first * 4 + at * 4u));
        }
        r.qcfail = qcfail_out;
        if(!defer && active) {
            // Decoder<T>::classify then Classifier<T>::classify (decoder.h:68-76, classifier.h:78-86)
            const uint32_t bin = classified ? c + 1u : 0u;
            const uint32_t word = bin * STAT_WORDS_PER_BIN + (qcfail ? 0u : static_cast<uint32_t>(STAT_FIELDS));
            if(bin == 0u || lds_bins) {
                if(!(defer_counters && bin > 0u)) {
                    uint32_t* const low_words = reinterpret_cast<uint32_t*>(histogram);      // a workgroup's counts stay below 2^32
                    atomicAdd(&low_words[2 * (STAT_JOB_WORDS + word + FIELD_COUNT)], 1u);
                    if(distance) { atomicAdd(&low_words[2 * (STAT_JOB_WORDS + word + FIELD_DISTANCE)], distance); }
                }
            } else if(!defer_counters) {
                atomicAdd(&args.stats[d.stats_offset + word + FIELD_COUNT], 1ull);
                if(distance) { atomicAdd(&args.stats[d.stats_offset + word + FIELD_DISTANCE], static_cast<unsigned long long>(distance)); }
            }
        }
        const unsigned long long deferred = __ballot(active && defer);
        pending_slot = 0xffffffffu;
        if(deferred) {
            const int lane = threadIdx.x & 63;
            uint32_t run = 0;
            if(lane == 0) { run = atomicAdd(&deferred_total, static_cast<uint32_t>(__popcll(deferred))); }
            run = __shfl(run, 0);
            if(active && defer) {
                pending_slot = run + __popcll(deferred & ((1ull << lane) - 1ull));
                pending_read = static_cast<uint32_t>(i);
            }
        }
        count_outgoing(args, histogram, active && !defer, r);
    }
    if(pending_slot != 0xffffffffu) { segment[pending_slot] = pending_read; }
    __syncthreads();
    if(threadIdx.x == 0) { args.worklist_count[blockIdx.x] = deferred_total; }
    histogram_flush(args, d, histogram);
}

template <int G, int STATE, int KIND>
int launch_state(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    const size_t lds_bytes = Layout::histogram_at(record_bytes_of(host_decoder, KIND)) + histogram_bytes;
    if(lds_bytes > 64u * 1024u) { return static_cast<int>(hipErrorInvalidValue); }       // *_supports checked the size
    if(host_decoder.quality_masking_threshold > 0) { hipLaunchKernelGGL((mdd_lookup_kernel<G, STATE, true, KIND>), grid, dim3(THREADS), lds_bytes, stream, args); }
    else { hipLaunchKernelGGL((mdd_lookup_kernel<G, STATE, false, KIND>), grid, dim3(THREADS), lds_bytes, stream, args); }
    return static_cast<int>(hipGetLastError());
}

template <int G, int KIND>
int launch_one(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(!args.first_stage) { return launch_state<G, 2, KIND>(args, host_decoder, grid, histogram_bytes, stream); }
    return args.input_flag ? launch_state<G, 1, KIND>(args, host_decoder, grid, histogram_bytes, stream)
                           : launch_state<G, 0, KIND>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace mdd_near
}  // namespace phq
