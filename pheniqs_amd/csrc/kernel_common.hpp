// Device-side pieces shared by every decoder stage kernel (gfx950).
//
// A job runs as one launch per decoder, in the order TranscodingDecoder::classify calls its classifiers
// (reference transcode.h:51-60).  The only state a read carries from one decoder to the next is what
// the reference keeps on the output Read: the sticky QC-fail flag and channel_index
// (classifier.h:83-85).  Both live in one 32-bit word per read that every stage reads and rewrites.
#pragma once

#include <atomic>
#include <type_traits>
#include <utility>

#include <hip/hip_runtime.h>

#include "device_types.hpp"

namespace phq {

// Decoder tables are immutable during a launch and every access to them is wave uniform.  Reading
// them through the constant address space makes hipcc emit scalar loads (s_load_*): the expected
// barcode bases, priors and thresholds travel in SGPRs and cost no vector instruction.
#define PHQ_CONSTANT __attribute__((address_space(4)))
typedef const PHQ_CONSTANT DevDecoder& DecoderRef;
template <typename T> __device__ __forceinline__ const PHQ_CONSTANT T* constant_cast(const T* p) {
    return (const PHQ_CONSTANT T*)p;
}

constexpr int BLOCK = 256;

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): the 32 bit chunks of a list mask kept in registers, by
// their static number (the exact tiers, pamld_cross.hpp and pamld_wide_exact.hpp)
template <typename F, int... C> __device__ __forceinline__ void for_each_chunk(F&& f, std::integer_sequence<int, C...>) { (f(std::integral_constant<int, C>{}), ...); }

struct ReadState {
    bool qcfail;
    int32_t channel;
};

// Counter sink: the workgroup histogram in LDS when the decoder's bins fit, global atomics otherwise — except
// bin 0 (unclassified), which always has its workgroup copy in LDS: every noise or unmatched read of the batch
// lands on it, and same address global atomics serialise (~88 per microsecond on this part).
struct StatSink {
    unsigned long long* lds;       // histogram base: [0..1] job words, then the decoder's bins
    unsigned long long* global;    // counter buffer
    bool defer;                    // count / distance / confidence of classified reads are left to count_results_kernel
};

__device__ __forceinline__ void stat_add(DecoderRef d, const StatSink& sink, int bin, int pf, int field, unsigned long long value) {
    const int word = bin * STAT_WORDS_PER_BIN + pf * STAT_FIELDS + field;
    if(d.use_lds_histogram || bin == 0) {
        atomicAdd(&sink.lds[STAT_JOB_WORDS + word], value);
    } else {
        atomicAdd(&sink.global[d.stats_offset + word], value);
    }
}

__device__ __forceinline__ void stat_add_confidence(DecoderRef d, const StatSink& sink, int bin, int pf, double confidence) {
    if(sink.defer) { return; }
    // exact 2^-52 fixed point in two 26 bit limbs: no word can overflow in a workgroup or in a whole run,
    // and the sum does not depend on the order reads are visited in
    const unsigned long long fixed = static_cast<unsigned long long>(__double2ll_rn(confidence * 4503599627370496.0));
    stat_add(d, sink, bin, pf, FIELD_CONFIDENCE_LOW, fixed & ((1ull << CONFIDENCE_LIMB_BITS) - 1));
    stat_add(d, sink, bin, pf, FIELD_CONFIDENCE_HIGH, fixed >> CONFIDENCE_LIMB_BITS);
}

// Decoder<T>::classify then Classifier<T>::classify: reference decoder.h:68-76, classifier.h:78-86
__device__ __forceinline__ void decoder_tail(DecoderRef d, const StatSink& sink, ReadState& r, int bin, int distance) {
    const int pf = r.qcfail ? 0 : 1;
    if(!(sink.defer && bin > 0)) {
        if(bin > 0 && distance) {
            stat_add(d, sink, bin, pf, FIELD_DISTANCE, static_cast<unsigned long long>(distance));
        }
        stat_add(d, sink, bin, pf, FIELD_COUNT, 1ull);
    }
    if(d.multiplexing) {
        r.channel = bin > 0 ? constant_cast(d.barcode_index)[bin - 1] : 0;
    }
}

__device__ __forceinline__ ReadState load_read_state(const StageArgs& args, int64_t i) {
    ReadState r;
    if(args.first_stage) {
        r.qcfail = args.input_flag ? (args.input_flag[i] & 1) != 0 : false;     // transcode.h:207-211
        r.channel = 0;
    } else {
        const uint32_t w = args.word[i];
        r.qcfail = (w >> 31) != 0;
        r.channel = static_cast<int32_t>(w & 0x7fffffffu);
    }
    return r;
}

__device__ __forceinline__ void store_read_state(const StageArgs& args, int64_t i, const ReadState& r) {
    args.word[i] = (static_cast<uint32_t>(r.channel) & 0x7fffffffu) | (r.qcfail ? 0x80000000u : 0u);
}

// The first tier kernels visit every read of the batch once and never read their own output back: they write the
// result record and the status word of EVERY read — a placeholder for the reads they defer, which the next tier
// overwrites — with non temporal stores.  Whole lines leave each wave (no masked partial stores) and do not
// displace the planes in L2; measured 1.26 -> 1.01 ms on the MDD exact match kernel.
__device__ __forceinline__ void stream_result(DevResult* to, const DevResult& out) {
    typedef uint32_t vector4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(*reinterpret_cast<const vector4*>(&out), reinterpret_cast<vector4*>(to));
}
__device__ __forceinline__ void stream_read_state(const StageArgs& args, int64_t i, const ReadState& r) {
    __builtin_nontemporal_store((static_cast<uint32_t>(r.channel) & 0x7fffffffu) | (r.qcfail ? 0x80000000u : 0u), &args.word[i]);
}
__device__ __forceinline__ void stream_placeholder(const StageArgs& args, int64_t i, const ReadState& r) {
    if(args.result) {
        DevResult out;
        out.index = 0; out.distance = 0; out.hq_distance = 0; out.status = STATUS_UNCLASSIFIED; out.qcfail = r.qcfail ? 1 : 0; out.confidence = 0;
        stream_result(&args.result[i], out);
    }
    stream_read_state(args, i, r);
}

__device__ __forceinline__ int histogram_words(DecoderRef d) {
    return STAT_JOB_WORDS + (d.use_lds_histogram ? (d.n_barcode + 1) * STAT_WORDS_PER_BIN : STAT_WORDS_PER_BIN);
}

__device__ __forceinline__ void histogram_clear(DecoderRef d, unsigned long long* histogram) {
    const int words = histogram_words(d);
    for(int i = threadIdx.x; i < words; i += blockDim.x) { histogram[i] = 0; }
}

// TranscodingDecoder::count / pf_count (transcode.h:61-64), one LDS add per wave
__device__ __forceinline__ void count_outgoing(const StageArgs& args, unsigned long long* histogram, bool active, const ReadState& r) {
    if(args.last_stage) {
        const unsigned long long pass = __ballot(active && !r.qcfail);
        const unsigned long long fail = __ballot(active && r.qcfail);
        if((threadIdx.x & 63) == 0) {
            if(fail) { atomicAdd(&histogram[0], static_cast<unsigned long long>(__popcll(fail))); }
            if(pass) { atomicAdd(&histogram[1], static_cast<unsigned long long>(__popcll(pass))); }
        }
    }
}

// one global atomic per non zero histogram word per workgroup
// the workgroup's outgoing reads [qc fail | pass filter]: to the job words when the stage is the job's last, and to bin 0
// of every passthrough decoder folded into the stage (StageArgs::echo_offset)
__device__ __forceinline__ void outgoing_flush(const StageArgs& args, const unsigned long long* outgoing) {
    for(int i = threadIdx.x; i < STAT_JOB_WORDS; i += blockDim.x) {
        const unsigned long long v = outgoing[i];
        if(v == 0) { continue; }
        if(args.job_outgoing) { atomicAdd(&args.stats[i], v); }
        for(int e = 0; e < args.echo_count; ++e) { atomicAdd(&args.stats[args.echo_offset[e] + (i == 0 ? 0 : STAT_FIELDS) + FIELD_COUNT], v); }
    }
}

__device__ __forceinline__ void histogram_flush(const StageArgs& args, DecoderRef d, const unsigned long long* histogram) {
    outgoing_flush(args, histogram);
    {
        const int words = d.use_lds_histogram ? (d.n_barcode + 1) * STAT_WORDS_PER_BIN : STAT_WORDS_PER_BIN;
        for(int i = threadIdx.x; i < words; i += blockDim.x) {
            const unsigned long long v = histogram[STAT_JOB_WORDS + i];
            if(v) { atomicAdd(&args.stats[d.stats_offset + i], v); }
        }
    }
}

// A, C, G or T: BAM codes 1, 2, 4, 8 (iupac.h:51-56); branch free, bit c of 0b1_0001_0110
__device__ __forceinline__ bool strict_code(uint32_t c) {
    return ((0x116u >> (c & 0xfu)) & 1u) != 0;
}

// The same power with a plain double Horner evaluation: within ~1 ulp (2.4e-16) of libm's.  Enough for everything
// but deciding an argmax between near equal probabilities; the literal kernel only pays for phred_probability on
// the few barcodes that reach the running maximum.
__device__ __forceinline__ double phred_probability_fast(double sigma, double log2_base_hi, double log2_base_lo) {
    const double z_hi = sigma * log2_base_hi;
    const double z_lo = __builtin_fma(sigma, log2_base_hi, -z_hi) + sigma * log2_base_lo;
    const double n = __builtin_floor(z_hi);
    const double t = ((z_hi - n) + z_lo) * 0.6931471805599453;
    double r = 2.8114572543455206e-15;                   // 1/17!
    r = __builtin_fma(r, t, 4.779477332387385e-14);      // 1/16!
    r = __builtin_fma(r, t, 7.647163731819816e-13);      // 1/15!
    r = __builtin_fma(r, t, 1.1470745597729725e-11);     // 1/14!
    r = __builtin_fma(r, t, 1.6059043836821613e-10);     // 1/13!
    r = __builtin_fma(r, t, 2.08767569878681e-09);       // 1/12!
    r = __builtin_fma(r, t, 2.505210838544172e-08);      // 1/11!
    r = __builtin_fma(r, t, 2.755731922398589e-07);      // 1/10!
    r = __builtin_fma(r, t, 2.7557319223985893e-06);     // 1/9!
    r = __builtin_fma(r, t, 2.48015873015873e-05);       // 1/8!
    r = __builtin_fma(r, t, 0.0001984126984126984);      // 1/7!
    r = __builtin_fma(r, t, 0.001388888888888889);       // 1/6!
    r = __builtin_fma(r, t, 0.008333333333333333);       // 1/5!
    r = __builtin_fma(r, t, 0.041666666666666664);       // 1/4!
    r = __builtin_fma(r, t, 0.16666666666666666);        // 1/3!
    r = __builtin_fma(r, t, 0.5);
    r = __builtin_fma(r, t, 1.0);
    r = __builtin_fma(r, t, 1.0);
    return __builtin_ldexp(r, static_cast<int>(n));
}

// pow(PHRED_PROBABILITY_BASE, sigma) (reference barcode.h:163) as 2^(sigma * log2(base)), base being the
// DOUBLE the reference holds (fl(10^-0.1), phred.h:34), log2(base) carried as a hi + lo pair.
// The exponent argument is formed in double-double (one FMA recovers the product's rounding error), so
// two sums that differ by one ulp — which the reference's Kahan recurrence does produce for the same
// multiset of costs visited in a different order — still map to different probabilities, ordered the way
// libm's pow orders them.  The first four terms of the series are carried in double-double as well (log2(base) to
// 106 bits from the host): the result is within ~0.502 ulp of the true power, and equals glibc's pow bit for bit
// in 99.6 % of 4 M Kahan sums of Phred costs (83 % for a plain double Horner) — what is left decides an argmax only
// between two barcodes whose costs are the same multiset in a different order, summing below ~8.
__device__ __forceinline__ double phred_probability(double sigma, double log2_base_hi, double log2_base_lo) {
    // z = sigma * log2(base) as a double-double, split at its floor
    const double z_hi = sigma * log2_base_hi;
    const double z_lo = __builtin_fma(sigma, log2_base_hi, -z_hi) + sigma * log2_base_lo;
    const double n = __builtin_floor(z_hi);
    const double f0 = z_hi - n;                                   // exact
    const double f_hi = f0 + z_lo;
    const double f_lo = (f0 - f_hi) + z_lo;
    // t = f * ln 2, double-double
    const double t_hi = f_hi * 0.6931471805599453;
    const double t_lo = __builtin_fma(f_hi, 0.6931471805599453, -t_hi) + (f_hi * 2.3190468138462996e-17 + f_lo * 0.6931471805599453);
    // e^t = 1 + t + t^2/2 + t^3/6 in double-double + t^4 * (1/4! + t/5! + ...) in double
    double s = 2.8114572543455206e-15;                           // 1/17!
    s = __builtin_fma(s, t_hi, 4.779477332387385e-14);           // 1/16!
    s = __builtin_fma(s, t_hi, 7.647163731819816e-13);           // 1/15!
    s = __builtin_fma(s, t_hi, 1.1470745597729725e-11);          // 1/14!
    s = __builtin_fma(s, t_hi, 1.6059043836821613e-10);          // 1/13!
    s = __builtin_fma(s, t_hi, 2.08767569878681e-09);            // 1/12!
    s = __builtin_fma(s, t_hi, 2.505210838544172e-08);           // 1/11!
    s = __builtin_fma(s, t_hi, 2.755731922398589e-07);           // 1/10!
    s = __builtin_fma(s, t_hi, 2.7557319223985893e-06);          // 1/9!
    s = __builtin_fma(s, t_hi, 2.48015873015873e-05);            // 1/8!
    s = __builtin_fma(s, t_hi, 0.0001984126984126984);           // 1/7!
    s = __builtin_fma(s, t_hi, 0.001388888888888889);            // 1/6!
    s = __builtin_fma(s, t_hi, 0.008333333333333333);            // 1/5!
    s = __builtin_fma(s, t_hi, 0.041666666666666664);            // 1/4!
    const double t2_hi = t_hi * t_hi;
    const double t2_lo = __builtin_fma(t_hi, t_hi, -t2_hi) + 2.0 * t_hi * t_lo;
    const double t3_hi = t2_hi * t_hi;
    const double t3_lo = __builtin_fma(t2_hi, t_hi, -t3_hi) + (t2_lo * t_hi + t2_hi * t_lo);
    const double tail = s * (t2_hi * t2_hi);
    const double c3_hi = t3_hi * 0.16666666666666666;
    const double c3_lo = __builtin_fma(t3_hi, 0.16666666666666666, -c3_hi) + (t3_hi * 9.25185853854297e-18 + t3_lo * 0.16666666666666666);
    const double h_hi = 0.5 * t2_hi;
    const double h_lo = 0.5 * t2_lo;
    // 1 + t_hi + h_hi + c3_hi with the rounding errors kept (two-sums), everything small added last
    const double a = 1.0 + t_hi;
    const double a_back = a - 1.0;
    const double e1 = (1.0 - (a - a_back)) + (t_hi - a_back);
    const double b = a + h_hi;
    const double b_back = b - a;
    const double e2 = (a - (b - b_back)) + (h_hi - b_back);
    const double c = b + c3_hi;
    const double c_back = c - b;
    const double e3 = (b - (c - c_back)) + (c3_hi - c_back);
    const double low = (((e1 + e2) + e3) + ((t_lo + h_lo) + c3_lo)) + tail;
    return __builtin_ldexp(c + low, static_cast<int>(n));
}

// load the L observed bytes of read i from a plane whose stride is a multiple of 4
template <int L>
__device__ __forceinline__ void load_plane_row(const uint8_t* plane, int64_t i, uint8_t (&out)[(L + 3) / 4 * 4]) {
    constexpr int WORDS = (L + 3) / 4;
    const uint32_t* row = reinterpret_cast<const uint32_t*>(plane + i * (WORDS * 4));
#pragma unroll
    for(int w = 0; w < WORDS; ++w) {
        const uint32_t v = row[w];
        out[4 * w + 0] = static_cast<uint8_t>(v);
        out[4 * w + 1] = static_cast<uint8_t>(v >> 8);
        out[4 * w + 2] = static_cast<uint8_t>(v >> 16);
        out[4 * w + 3] = static_cast<uint8_t>(v >> 24);
    }
}

// More than 64 KiB of dynamic LDS has to be asked for (static + dynamic at most 160 KiB), once per kernel and device (one process may drive several
// devices through several decoder sets, from several threads).
template <typename Kernel>
inline int allow_large_lds(Kernel kernel, std::atomic<uint64_t>& done, int bytes = 160 * 1024) {
    int device = 0;
    if(hipGetDevice(&device) != hipSuccess) { return static_cast<int>(hipErrorInvalidDevice); }
    const uint64_t bit = 1ull << (static_cast<unsigned>(device) & 63u);
    if((done.load(std::memory_order_acquire) & bit) == 0) {
        const hipError_t status = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if(status != hipSuccess) { return static_cast<int>(status); }
        done.fetch_or(bit, std::memory_order_release);
    }
    return 0;
}

// kernel entry points by translation unit
int launch_pamld_literal(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream);
int launch_mdd(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream);
int launch_count(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream);
bool count_results_supports(const DevDecoder& host_decoder);
int launch_count_results(const StageArgs& args, const DevDecoder& host_decoder, int32_t compute_units, hipStream_t stream);
bool mdd_scan_supports(const DevDecoder& host_decoder);
int launch_mdd_scan(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream);
bool mdd_near_supports(const DevDecoder& host_decoder);      // mdd_near.hip: one table lookup per read, the scan on what it leaves
int launch_mdd_near(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream);
bool mdd_pair_supports(const DevDecoder& host_decoder);      // mdd_near.hip: one lookup per segment of a paired codec
int launch_mdd_pair(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream);
bool mdd_cross2_supports(const DevDecoder& host_decoder);    // mdd_near.hip: one lookup per segment of a two segment combinatorial codec
int launch_mdd_cross2(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream);
bool mdd_fused_supports(const DevDecoder& host_decoder);
int launch_mdd_fused(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream);
bool mdd_cross_supports(const DevDecoder& host_decoder);
int launch_mdd_cross(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream);
bool mdd_exact_supports(const DevDecoder& host_decoder);
int launch_mdd_exact(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t lds_bytes, hipStream_t stream);
bool pamld_tiny_supports(const DevDecoder& host_decoder);   // pamld_tiny.hip: at most 8 barcodes, fp64 products, no tables
int launch_pamld_tiny(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream);
bool pamld_few_supports(const DevDecoder& host_decoder);    // pamld_few.hip: the same codecs up to 8 nt in one persistent kernel, a pipeline per wave
int pamld_few_waves();                                      // waves (= worklist segments) per workgroup of that kernel
int launch_pamld_few(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, hipStream_t stream);
bool pamld_lut_supports(const DevDecoder& host_decoder);
bool pamld_lut_preferred(const DevDecoder& host_decoder);          // ... and few enough barcodes that a combinatorial set does not take the factorised kernels instead
int pamld_lut_block(const DevDecoder& host_decoder);       // threads per workgroup of the product form kernel
int launch_pamld_lut(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream);
int pamld_near_block(const DevDecoder& host_decoder);      // threads per workgroup of the neighbour list kernel
bool pamld_near_supports(const DevDecoder& host_decoder);   // pamld_near.hip: neighbour list tier in front of the product form kernel
int launch_pamld_near(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream);
bool pamld_stream_supports(const DevDecoder& host_decoder);   // pamld_stream.hip: neighbour lists and product form scan in one persistent kernel, a pipeline per wave
int pamld_stream_waves();                                      // waves (= worklist segments) per workgroup of that kernel
int launch_pamld_stream(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, hipStream_t stream);
void trace_worklist(const char* tier, const StageArgs& args, unsigned segments, hipStream_t stream);   // stage.hip: PHQ_TRACE_WORKLIST
bool pamld_cross_near_supports(const DevDecoder& host_decoder);   // pamld_cross_near.hip: neighbour lists per segment of a two segment combinatorial set
int pamld_cross_near_block(const DevDecoder& host_decoder);
int launch_pamld_cross_near(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream);
bool pamld_cross_stream_supports(const DevDecoder& host_decoder);   // pamld_cross_stream.hip: the same sets (uniform priors, 8 to 10 nt), lists and factorised scan in one persistent kernel
int pamld_cross_stream_waves();
int launch_pamld_cross_stream(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, hipStream_t stream);
bool pamld_pair_supports(const DevDecoder& host_decoder);   // pamld_pair.hip: paired barcodes, one barcode scored, in front of the product form kernels
int launch_pamld_pair(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream);
bool pamld_plate_supports(const DevDecoder& host_decoder);   // pamld_plate.hip: dual index plates, few words a segment, every word of both segments scored
int pamld_plate_block();                                     // threads of its one workgroup a CU
int pamld_plate_waves();                                     // ... and its waves: a worklist segment each
int launch_pamld_plate(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream);
bool pamld_wide_supports(const DevDecoder& host_decoder);
int pamld_wide_block(const DevDecoder& host_decoder);
int launch_pamld_wide(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream);
bool pamld_wide_exact_supports(const DevDecoder& host_decoder);      // pamld_wide_exact.hip: fp64 tier behind it
int launch_pamld_wide_exact(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream);
bool pamld_cross_supports(const DevDecoder& host_decoder);
int launch_pamld_cross(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream);
bool pamld_exact_supports(const DevDecoder& host_decoder);
bool pamld_exact_legacy_fits(const DevDecoder& host_decoder);   // ... by the kernels of pamld_cross.hpp themselves (pamld_exact_lanes.hip holds larger single segment sets)
int launch_pamld_cross_exact(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream);
int launch_pamld_cross_exact_dense(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream);   // pamld_cross_w2.hip
bool pamld_exact_lanes_supports(const DevDecoder& host_decoder, size_t histogram_bytes);   // pamld_exact_lanes.hip: the same tier with eight lanes a read
int launch_pamld_exact_lanes(const StageArgs& args, const DevDecoder& host_decoder, dim3 grid, size_t histogram_bytes, hipStream_t stream);

}  // namespace phq
