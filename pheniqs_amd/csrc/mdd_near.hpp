// K2n (one lookup): MdDecoder<T>::classify for single segment ACGT codecs of 5 to 10 nt whose distance tolerance is 0 or 1.
//
//   reference mdd.cpp:37-86   exact match first (the observation as a key of element_by_sequence), else the barcodes in
//                             index order, the FIRST within the tolerance wins; nothing within it: unclassified, QC fail
//   reference sequence.h:90-98, 321-332   distance_from / masked_distance_from
//
// With a tolerance of at most one mismatch the whole loop is a function of the observed word alone: the word itself if
// it is a barcode, else the lowest ordinal among the barcodes one mismatch away (the first the reference's loop meets),
// else nothing.  That function is the table the PAMLD neighbour list tier already uses (pamld_near.hpp,
// build_near_tables: 4^L entries of 16 bits in global memory, the few thousand keys real reads carry stay in L1/L2), so a
// read costs: its code words (the quality plane is only read when a masking threshold is set), one lookup, the barcode's
// own bits from LDS for the distance, one record.  No hash probe, no scan, no ring: the kernel streams.
// What it cannot finish is appended to the workgroup's worklist segment for the scan kernel (mdd_scan.hip) or the
// general one (mdd.hip): a base outside ACGT, a read shorter than the barcode, a quality below the masking threshold in
// a read that is not a barcode word for word (the masked distance is then no longer a function of the word).  A table miss IS final: no barcode within one mismatch.
#pragma once

#include "kernel_common.hpp"

namespace phq {
namespace mdd_near {

constexpr int THREADS = 256;

struct Layout {
    static constexpr uint32_t DEFERRED_AT = 0;
    static constexpr uint32_t RECORD_AT = 16;                                  // n_barcode x {low plane, high plane, barcode index, -}
    __host__ __device__ static constexpr uint32_t histogram_at(int n_barcode) { return RECORD_AT + static_cast<uint32_t>(n_barcode) * 16u; }
};

typedef uint32_t Quad __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const Quad* LdsQuads;
// the combinatorial kernel keeps {low, high} of every segment word instead of barcode records
__host__ __device__ constexpr uint32_t cross_histogram_at(int words) { return (Layout::RECORD_AT + static_cast<uint32_t>(words) * 8u + 15u) & ~15u; }

// STATE as in pamld_near.hpp (where the carried state comes from); MASKING: a quality masking threshold is set
template <int L, int STATE, bool MASKING>
__global__ __launch_bounds__(THREADS) void mdd_near_kernel(StageArgs args) {
    static_assert(L >= 5 && L <= 10, "the lookup table has 4^L entries");
    constexpr int G = (L + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    uint32_t& deferred_total = *reinterpret_cast<uint32_t*>(lds_raw + Layout::DEFERRED_AT);
    uint32_t* const records = reinterpret_cast<uint32_t*>(lds_raw + Layout::RECORD_AT);
    DecoderRef d = *constant_cast(args.decoder);
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw + Layout::histogram_at(d.n_barcode));

    if(threadIdx.x == 0) { deferred_total = 0; }
    for(int w = threadIdx.x; w < d.n_barcode * 4; w += THREADS) { records[w] = d.mdd_near_record[w]; }
    histogram_clear(d, histogram);
    __syncthreads();

    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const uint16_t* const nearest = d.near_table;
    const int64_t n = args.n_read;
    const int64_t tiles = (n + THREADS - 1) / THREADS;
    const uint32_t n_barcode = static_cast<uint32_t>(d.n_barcode);
    const int tolerance = d.distance_tolerance[0];
    const uint32_t masking_addend = static_cast<uint32_t>(128 - min(max(d.quality_masking_threshold, 0), 128)) * 0x01010101u;
    const bool multiplexing = d.multiplexing != 0;
    const bool lds_bins = d.use_lds_histogram != 0;
    const bool defer_counters = args.defer_counters != 0;
    auto valid = [](int w) -> uint32_t { return L - 4 * w >= 4 ? 0xffffffffu : ((1u << (8 * (L - 4 * w))) - 1u); };

    // a tile's inputs one tile ahead, issued behind the table lookup (pamld_near.hpp)
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
        uint32_t x[G];
        uint32_t unfinished = 0;
        uint32_t masked = 0;
#pragma unroll
        for(int w = 0; w < G; ++w) {
            x[w] = in.code[w] & (0x0f0f0f0fu & valid(w));
            const uint32_t z = x[w] & ((x[w] | 0x80808080u) - 0x01010101u);
            const uint32_t strict = (x[w] + 0x7f7f7f7fu) & ~(z + 0x7f7f7f7fu) & 0x80808080u;
            unfinished |= (strict ^ (0x80808080u & valid(w))) | (in.code[w] & (0xf0f0f0f0u & valid(w)));      // a code above 15 is not a base either
            if(MASKING) {
                // quality below the threshold (0..128): bit 7 stays clear after adding 128 - threshold; Q >= 128 is never below
                const uint32_t q7 = in.quality[w] & 0x7f7f7f7fu;
                masked |= ~((q7 + masking_addend) | in.quality[w]) & (0x80808080u & valid(w));
            }
        }
        auto gather = [](uint32_t planes) {            // bits 8 j and 8 j + 4 -> bits j and 4 + j
            const uint32_t t = planes | (planes >> 7);
            return (t | (t >> 14)) & 0xffu;
        };
        uint32_t low = gather((((x[0] >> 1) | (x[0] >> 3)) & 0x01010101u) | (((x[1] << 3) | (x[1] << 1)) & 0x10101010u));
        uint32_t high = gather((((x[0] >> 2) | (x[0] >> 3)) & 0x01010101u) | (((x[1] << 2) | (x[1] << 1)) & 0x10101010u));
        if(G == 3) {
            const uint32_t both = gather((((x[G - 1] >> 1) | (x[G - 1] >> 3)) & 0x01010101u) | (((x[G - 1] << 2) | (x[G - 1] << 1)) & 0x10101010u));
            low |= (both & 0xfu) << 8;
            high |= (both >> 4) << 8;
        }
        const uint32_t key = (low | (high << L)) & ((1u << (2 * L)) - 1u);
        if(pending_slot != 0xffffffffu) { segment[pending_slot] = pending_read; }      // the previous tile's deferred reads
        const uint32_t looked_up = nearest[key];                                       // the barcode this word is (next to), if any
        fetch(min(tile + gridDim.x, tiles - 1), next);
        const bool known = looked_up < n_barcode;
        const uint32_t c = known ? looked_up : 0u;
        const Quad record = *(LdsQuads)static_cast<uintptr_t>(Layout::RECORD_AT + c * 16u);
        const int k = __popc(((low ^ record.x) | (high ^ record.y)) & ((1u << L) - 1u));
        // the exact match comes before any masking (mdd.cpp:43-46): a masked position only matters to a read that is not a barcode
        const bool defer = unfinished != 0 || (MASKING && masked != 0 && !(known && k == 0));
        const bool classified = known && k <= tolerance;               // mdd.cpp:43-46 (k = 0), 47-80 (k = 1 within the tolerance)
        const uint32_t index = classified ? record.z : 0u;
        const uint32_t distance = classified ? static_cast<uint32_t>(k) : 0u;
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

// Two segment codecs that are NOT every combination of their words (unique dual indexes, ligation + RT pairs), every
// word of a segment belonging to ONE barcode, tolerances of at most 2: the same idea with one lookup per segment (the
// tables of pamld_pair.hpp: the barcode of the segment word up to two mismatches away, the nearest first).  A barcode is
// within both tolerances only if it is the one each segment names — the tolerances never exceed the Shannon bound of their
// segment's words, so at most one word of a segment is that close — hence: both segments name the same barcode and both
// distances are within tolerance, or the read is unclassified.  G = words of the plane row.
template <int G, int STATE, bool MASKING>
__global__ __launch_bounds__(THREADS) void mdd_pair_kernel(StageArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    uint32_t& deferred_total = *reinterpret_cast<uint32_t*>(lds_raw + Layout::DEFERRED_AT);
    uint32_t* const records = reinterpret_cast<uint32_t*>(lds_raw + Layout::RECORD_AT);
    DecoderRef d = *constant_cast(args.decoder);
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw + Layout::histogram_at(d.n_barcode));

    if(threadIdx.x == 0) { deferred_total = 0; }
    for(int w = threadIdx.x; w < d.n_barcode * 4; w += THREADS) { records[w] = d.pair_record[(w >> 2) * PAIR_RECORD_WORDS + (w & 3)]; }      // {low, high, -, index}
    histogram_clear(d, histogram);
    __syncthreads();

    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const uint16_t* const nearest0 = d.pair_table[0];
    const uint16_t* const nearest1 = d.pair_table[1];
    const uint32_t start0 = static_cast<uint32_t>(d.pair_start[0]), start1 = static_cast<uint32_t>(d.pair_start[1]);
    const uint32_t length0 = static_cast<uint32_t>(d.pair_length[0]), length1 = static_cast<uint32_t>(d.pair_length[1]);
    const uint32_t mask0 = (1u << length0) - 1u, mask1 = (1u << length1) - 1u;
    const int L = d.nucleotide_cardinality;
    const int64_t n = args.n_read;
    const int64_t tiles = (n + THREADS - 1) / THREADS;
    const uint32_t n_barcode = static_cast<uint32_t>(d.n_barcode);
    const int tolerance0 = d.distance_tolerance[0], tolerance1 = d.distance_tolerance[1];
    const uint32_t masking_addend = static_cast<uint32_t>(128 - min(max(d.quality_masking_threshold, 0), 128)) * 0x01010101u;
    const bool multiplexing = d.multiplexing != 0;
    const bool lds_bins = d.use_lds_histogram != 0;
    const bool defer_counters = args.defer_counters != 0;
    uint32_t valid[G];
#pragma unroll
    for(int w = 0; w < G; ++w) { valid[w] = L - 4 * w >= 4 ? 0xffffffffu : (L - 4 * w <= 0 ? 0u : ((1u << (8 * (L - 4 * w))) - 1u)); }

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
            unfinished |= (strict ^ (0x80808080u & valid[w])) | (in.code[w] & (0xf0f0f0f0u & valid[w]));
            const uint32_t both = gather((((x >> 1) | (x >> 3)) & 0x01010101u) | (((x << 2) | (x << 1)) & 0x10101010u));
            low |= (both & 0xfu) << (4 * w);
            high |= (both >> 4) << (4 * w);
            if(MASKING) {
                const uint32_t q7 = in.quality[w] & 0x7f7f7f7fu;
                masked |= ~((q7 + masking_addend) | in.quality[w]) & (0x80808080u & valid[w]);
            }
        }
        const uint32_t key0 = ((low >> start0) & mask0) | (((high >> start0) & mask0) << length0);
        const uint32_t key1 = ((low >> start1) & mask1) | (((high >> start1) & mask1) << length1);
        if(pending_slot != 0xffffffffu) { segment[pending_slot] = pending_read; }      // the previous tile's deferred reads
        const uint32_t named0 = nearest0[key0];
        const uint32_t named1 = nearest1[key1];
        fetch(min(tile + gridDim.x, tiles - 1), next);
        const bool known = named0 == named1 && named0 < n_barcode;
        const uint32_t c = known ? named0 : 0u;
        const Quad record = *(LdsQuads)static_cast<uintptr_t>(Layout::RECORD_AT + c * 16u);
        const uint32_t mismatches = (low ^ record.x) | (high ^ record.y);
        const int k0 = __popc(mismatches & (mask0 << start0));
        const int k1 = __popc(mismatches & (mask1 << start1));
        const bool classified = known && k0 <= tolerance0 && k1 <= tolerance1;            // mdd.cpp:43-80
        // the exact match comes before any masking (mdd.cpp:43-46)
        const bool defer = unfinished != 0 || (MASKING && masked != 0 && !(known && k0 + k1 == 0));
        const uint32_t index = classified ? record.w : 0u;
        const uint32_t distance = classified ? static_cast<uint32_t>(k0 + k1) : 0u;
        const bool qcfail = r.qcfail || !classified;                   // mdd.cpp:82-84
        const bool qcfail_out = defer ? r.qcfail : qcfail;
        {
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
            const uint32_t bin = classified ? c + 1u : 0u;
            const uint32_t word = bin * STAT_WORDS_PER_BIN + (qcfail ? 0u : static_cast<uint32_t>(STAT_FIELDS));
            if(bin == 0u || lds_bins) {
                if(!(defer_counters && bin > 0u)) {
                    uint32_t* const low_words = reinterpret_cast<uint32_t*>(histogram);
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

// Two segment COMBINATORIAL codecs (every combination of the segments' words, numbered in lexicographic order of the word
// ordinals: DevDecoder::cross_identity), segments of 5 to 10 nt, tolerances of at most 1: one lookup per segment names the
// segment's word (pamld_near.hpp's table over the segment's distinct words), and the first barcode in index order whose
// every segment is within tolerance (mdd.cpp:47-80) is the combination of the named words — at most one word of a segment is
// within a tolerance that does not exceed the Shannon bound.  G = words of the plane row.
template <int G, int STATE, bool MASKING>
__global__ __launch_bounds__(THREADS) void mdd_cross_kernel2(StageArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds_raw) != 0) { __builtin_trap(); }
    uint32_t& deferred_total = *reinterpret_cast<uint32_t*>(lds_raw + Layout::DEFERRED_AT);
    uint32_t* const records = reinterpret_cast<uint32_t*>(lds_raw + Layout::RECORD_AT);
    DecoderRef d = *constant_cast(args.decoder);
    unsigned long long* histogram = reinterpret_cast<unsigned long long*>(lds_raw + cross_histogram_at(d.cross_unique[0] + d.cross_unique[1]));

    if(threadIdx.x == 0) { deferred_total = 0; }
    const uint32_t unique0 = static_cast<uint32_t>(d.cross_unique[0]), unique1 = static_cast<uint32_t>(d.cross_unique[1]);
    for(uint32_t w = threadIdx.x; w < 2u * (unique0 + unique1); w += THREADS) { records[w] = d.mdd_cross_word[w]; }      // {low, high} of segment 0's words, then segment 1's
    histogram_clear(d, histogram);
    __syncthreads();

    uint32_t* const segment = args.worklist + static_cast<int64_t>(blockIdx.x) * args.worklist_stride;
    const uint16_t* const nearest0 = d.cross_near_table[0];
    const uint16_t* const nearest1 = d.cross_near_table[1];
    const uint32_t start0 = static_cast<uint32_t>(d.cross_start[0]), start1 = static_cast<uint32_t>(d.cross_start[1]);
    const uint32_t length0 = static_cast<uint32_t>(d.cross_length[0]), length1 = static_cast<uint32_t>(d.cross_length[1]);
    const uint32_t stride0 = static_cast<uint32_t>(d.cross_stride[0]), stride1 = static_cast<uint32_t>(d.cross_stride[1]);
    const bool identity_index = d.identity_index != 0;
    const uint32_t mask0 = (1u << length0) - 1u, mask1 = (1u << length1) - 1u;
    const int L = d.nucleotide_cardinality;
    const int64_t n = args.n_read;
    const int64_t tiles = (n + THREADS - 1) / THREADS;
    const int tolerance0 = d.distance_tolerance[0], tolerance1 = d.distance_tolerance[1];
    const uint32_t masking_addend = static_cast<uint32_t>(128 - min(max(d.quality_masking_threshold, 0), 128)) * 0x01010101u;
    const bool multiplexing = d.multiplexing != 0;
    const bool lds_bins = d.use_lds_histogram != 0;
    const bool defer_counters = args.defer_counters != 0;
    uint32_t valid[G];
#pragma unroll
    for(int w = 0; w < G; ++w) { valid[w] = L - 4 * w >= 4 ? 0xffffffffu : (L - 4 * w <= 0 ? 0u : ((1u << (8 * (L - 4 * w))) - 1u)); }

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
            unfinished |= (strict ^ (0x80808080u & valid[w])) | (in.code[w] & (0xf0f0f0f0u & valid[w]));
            const uint32_t both = gather((((x >> 1) | (x >> 3)) & 0x01010101u) | (((x << 2) | (x << 1)) & 0x10101010u));
            low |= (both & 0xfu) << (4 * w);
            high |= (both >> 4) << (4 * w);
            if(MASKING) {
                const uint32_t q7 = in.quality[w] & 0x7f7f7f7fu;
                masked |= ~((q7 + masking_addend) | in.quality[w]) & (0x80808080u & valid[w]);
            }
        }
        const uint32_t key0 = ((low >> start0) & mask0) | (((high >> start0) & mask0) << length0);
        const uint32_t key1 = ((low >> start1) & mask1) | (((high >> start1) & mask1) << length1);
        if(pending_slot != 0xffffffffu) { segment[pending_slot] = pending_read; }      // the previous tile's deferred reads
        const uint32_t named0 = nearest0[key0];
        const uint32_t named1 = nearest1[key1];
        fetch(min(tile + gridDim.x, tiles - 1), next);
        const bool known = named0 < unique0 && named1 < unique1;
        const uint32_t u0 = known ? named0 : 0u, u1 = known ? named1 : 0u;
        typedef uint32_t Pair2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) const Pair2* LdsPairs2;
        const Pair2 word0 = *(LdsPairs2)static_cast<uintptr_t>(Layout::RECORD_AT + u0 * 8u);
        const Pair2 word1 = *(LdsPairs2)static_cast<uintptr_t>(Layout::RECORD_AT + (unique0 + u1) * 8u);
        const int k0 = __popc((((low >> start0) ^ word0.x) | ((high >> start0) ^ word0.y)) & mask0);
        const int k1 = __popc((((low >> start1) ^ word1.x) | ((high >> start1) ^ word1.y)) & mask1);
        const bool classified = known && k0 <= tolerance0 && k1 <= tolerance1;            // mdd.cpp:43-80
        const uint32_t c = u0 * stride0 + u1 * stride1;                                   // the barcode's ordinal (lexicographic numbering)
        // the exact match comes before any masking (mdd.cpp:43-46)
        const bool defer = unfinished != 0 || (MASKING && masked != 0 && !(known && k0 + k1 == 0));
        const uint32_t index = !classified ? 0u : identity_index ? c + 1u : static_cast<uint32_t>(d.barcode_index[c]);
        const uint32_t distance = classified ? static_cast<uint32_t>(k0 + k1) : 0u;
        const bool qcfail = r.qcfail || !classified;                   // mdd.cpp:82-84
        const bool qcfail_out = defer ? r.qcfail : qcfail;
        {
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
            const uint32_t bin = classified ? c + 1u : 0u;
            const uint32_t word = bin * STAT_WORDS_PER_BIN + (qcfail ? 0u : static_cast<uint32_t>(STAT_FIELDS));
            if(bin == 0u || lds_bins) {
                if(!(defer_counters && bin > 0u)) {
                    uint32_t* const low_words = reinterpret_cast<uint32_t*>(histogram);
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

template <int G, int STATE>
int launch_pair_state(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    const size_t lds_bytes = Layout::histogram_at(host_decoder.n_barcode) + histogram_bytes;
    if(host_decoder.quality_masking_threshold > 0) { hipLaunchKernelGGL((mdd_pair_kernel<G, STATE, true>), grid, dim3(THREADS), lds_bytes, stream, args); }
    else { hipLaunchKernelGGL((mdd_pair_kernel<G, STATE, false>), grid, dim3(THREADS), lds_bytes, stream, args); }
    return static_cast<int>(hipGetLastError());
}

template <int G, int STATE>
int launch_cross_state(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    const size_t lds_bytes = cross_histogram_at(host_decoder.cross_unique[0] + host_decoder.cross_unique[1]) + histogram_bytes;
    if(host_decoder.quality_masking_threshold > 0) { hipLaunchKernelGGL((mdd_cross_kernel2<G, STATE, true>), grid, dim3(THREADS), lds_bytes, stream, args); }
    else { hipLaunchKernelGGL((mdd_cross_kernel2<G, STATE, false>), grid, dim3(THREADS), lds_bytes, stream, args); }
    return static_cast<int>(hipGetLastError());
}

template <int G>
int launch_cross(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(!args.first_stage) { return launch_cross_state<G, 2>(args, host_decoder, grid, histogram_bytes, stream); }
    return args.input_flag ? launch_cross_state<G, 1>(args, host_decoder, grid, histogram_bytes, stream)
                           : launch_cross_state<G, 0>(args, host_decoder, grid, histogram_bytes, stream);
}

template <int G>
int launch_pair(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(!args.first_stage) { return launch_pair_state<G, 2>(args, host_decoder, grid, histogram_bytes, stream); }
    return args.input_flag ? launch_pair_state<G, 1>(args, host_decoder, grid, histogram_bytes, stream)
                           : launch_pair_state<G, 0>(args, host_decoder, grid, histogram_bytes, stream);
}

template <int L, int STATE>
int launch_state(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    const size_t lds_bytes = Layout::histogram_at(host_decoder.n_barcode) + histogram_bytes;
    if(host_decoder.quality_masking_threshold > 0) { hipLaunchKernelGGL((mdd_near_kernel<L, STATE, true>), grid, dim3(THREADS), lds_bytes, stream, args); }
    else { hipLaunchKernelGGL((mdd_near_kernel<L, STATE, false>), grid, dim3(THREADS), lds_bytes, stream, args); }
    return static_cast<int>(hipGetLastError());
}

template <int L>
int launch_one(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream) {
    if(!args.first_stage) { return launch_state<L, 2>(args, host_decoder, grid, histogram_bytes, stream); }
    return args.input_flag ? launch_state<L, 1>(args, host_decoder, grid, histogram_bytes, stream)
                           : launch_state<L, 0>(args, host_decoder, grid, histogram_bytes, stream);
}

}  // namespace mdd_near
}  // namespace phq
