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
// DOUBLE the reference holds (fl(10^-0.1), phred.h:34), log2(base) carried as a hi + lo pair (106 bits, from the host).
// The exponent is formed in double-double (one FMA recovers the product's rounding error), so two sums that differ by
// one ulp — which the reference's Kahan recurrence does produce for the same multiset of costs visited in a different
// order — still map to different probabilities, ordered the way libm's pow orders them.  Round 6: the way libm itself
// goes — the exponent split at the nearest 1/128, 2^(j/128) from a table in double-double, e^t - 1 on |t| <= 0.0028 with
// t in double-double and its powers in double, the product and the last sum with their rounding errors kept — : the
// result equals glibc's pow bit for bit on 99.93 % of 6 M Kahan sums of Phred costs and on 99.98 % of those below 8,
// where ties are decided (the form before this one, a series on |t| <= 0.69 with three terms in double-double: 99.6 %;
// a plain double Horner: 83 %).  What is left is glibc's own distance from the correctly rounded power, which grows with
// the sum (it carries log(base) to 2^-68).
static __device__ const double EXP2_TABLE[128][2] = {            // 2^(j/128) as hi + lo (60 digit arithmetic, rounded)
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.0163da9fb3335p+0, 0x1.b61299ab8cdb7p-54},
    {0x1.02c9a3e778061p+0, -0x1.19083535b085dp-56},
    {0x1.04315e86e7f85p+0, -0x1.0a31c1977c96ep-54},
    {0x1.059b0d3158574p+0, 0x1.d73e2a475b465p-55},
    {0x1.0706b29ddf6dep+0, -0x1.c91dfe2b13c27p-55},
    {0x1.0874518759bc8p+0, 0x1.186be4bb284ffp-57},
    {0x1.09e3ecac6f383p+0, 0x1.1487818316136p-54},
    {0x1.0b5586cf9890fp+0, 0x1.8a62e4adc610bp-54},
    {0x1.0cc922b7247f7p+0, 0x1.01edc16e24f71p-54},
    {0x1.0e3ec32d3d1a2p+0, 0x1.03a1727c57b53p-59},
    {0x1.0fb66affed31bp+0, -0x1.b9bedc44ebd7bp-57},
    {0x1.11301d0125b51p+0, -0x1.6c51039449b3ap-54},
    {0x1.12abdc06c31ccp+0, -0x1.1b514b36ca5c7p-58},
    {0x1.1429aaea92de0p+0, -0x1.32fbf9af1369ep-54},
    {0x1.15a98c8a58e51p+0, 0x1.2406ab9eeab0ap-55},
    {0x1.172b83c7d517bp+0, -0x1.19041b9d78a76p-55},
    {0x1.18af9388c8deap+0, -0x1.11023d1970f6cp-54},
    {0x1.1a35beb6fcb75p+0, 0x1.e5b4c7b4968e4p-55},
    {0x1.1bbe084045cd4p+0, -0x1.95386352ef607p-54},
    {0x1.1d4873168b9aap+0, 0x1.e016e00a2643cp-54},
    {0x1.1ed5022fcd91dp+0, -0x1.1df98027bb78cp-54},
    {0x1.2063b88628cd6p+0, 0x1.dc775814a8495p-55},
    {0x1.21f49917ddc96p+0, 0x1.2a97e9494a5eep-55},
    {0x1.2387a6e756238p+0, 0x1.9b07eb6c70573p-54},
    {0x1.251ce4fb2a63fp+0, 0x1.ac155bef4f4a4p-55},
    {0x1.26b4565e27cddp+0, 0x1.2bd339940e9d9p-55},
    {0x1.284dfe1f56381p+0, -0x1.a4c3a8c3f0d7ep-54},
    {0x1.29e9df51fdee1p+0, 0x1.612e8afad1255p-55},
    {0x1.2b87fd0dad990p+0, -0x1.10adcd6381aa4p-59},
    {0x1.2d285a6e4030bp+0, 0x1.0024754db41d5p-54},
    {0x1.2ecafa93e2f56p+0, 0x1.1ca0f45d52383p-56},
    {0x1.306fe0a31b715p+0, 0x1.6f46ad23182e4p-55},
    {0x1.32170fc4cd831p+0, 0x1.a9ce78e18047cp-55},
    {0x1.33c08b26416ffp+0, 0x1.32721843659a6p-54},
    {0x1.356c55f929ff1p+0, -0x1.b5cee5c4e4628p-55},
    {0x1.371a7373aa9cbp+0, -0x1.63aeabf42eae2p-54},
    {0x1.38cae6d05d866p+0, -0x1.e958d3c9904bdp-54},
    {0x1.3a7db34e59ff7p+0, -0x1.5e436d661f5e3p-56},
    {0x1.3c32dc313a8e5p+0, -0x1.efff8375d29c3p-54},
    {0x1.3dea64c123422p+0, 0x1.ada0911f09ebcp-55},
    {0x1.3fa4504ac801cp+0, -0x1.7d023f956f9f3p-54},
    {0x1.4160a21f72e2ap+0, -0x1.ef3691c309278p-58},
    {0x1.431f5d950a897p+0, -0x1.1c7dde35f7999p-55},
    {0x1.44e086061892dp+0, 0x1.89b7a04ef80d0p-59},
    {0x1.46a41ed1d0057p+0, 0x1.c944bd1648a76p-54},
    {0x1.486a2b5c13cd0p+0, 0x1.3c1a3b69062f0p-56},
    {0x1.4a32af0d7d3dep+0, 0x1.9cb62f3d1be56p-54},
    {0x1.4bfdad5362a27p+0, 0x1.d4397afec42e2p-56},
    {0x1.4dcb299fddd0dp+0, 0x1.8ecdbbc6a7833p-54},
    {0x1.4f9b2769d2ca7p+0, -0x1.4b309d25957e3p-54},
    {0x1.516daa2cf6642p+0, -0x1.f768569bd93efp-55},
    {0x1.5342b569d4f82p+0, -0x1.07abe1db13cadp-55},
    {0x1.551a4ca5d920fp+0, -0x1.d689cefede59bp-55},
    {0x1.56f4736b527dap+0, 0x1.9bb2c011d93adp-54},
    {0x1.58d12d497c7fdp+0, 0x1.295e15b9a1de8p-55},
    {0x1.5ab07dd485429p+0, 0x1.6324c054647adp-54},
    {0x1.5c9268a5946b7p+0, 0x1.c4b1b816986a2p-60},
    {0x1.5e76f15ad2148p+0, 0x1.ba6f93080e65ep-54},
    {0x1.605e1b976dc09p+0, -0x1.3e2429b56de47p-54},
    {0x1.6247eb03a5585p+0, -0x1.383c17e40b497p-54},
    {0x1.6434634ccc320p+0, -0x1.c483c759d8933p-55},
    {0x1.6623882552225p+0, -0x1.bb60987591c34p-54},
    {0x1.68155d44ca973p+0, 0x1.038ae44f73e65p-57},
    {0x1.6a09e667f3bcdp+0, -0x1.bdd3413b26456p-54},
    {0x1.6c012750bdabfp+0, -0x1.2895667ff0b0dp-56},
    {0x1.6dfb23c651a2fp+0, -0x1.bbe3a683c88abp-57},
    {0x1.6ff7df9519484p+0, -0x1.83c0f25860ef6p-55},
    {0x1.71f75e8ec5f74p+0, -0x1.16e4786887a99p-55},
    {0x1.73f9a48a58174p+0, -0x1.0a8d96c65d53cp-54},
    {0x1.75feb564267c9p+0, -0x1.0245957316dd3p-54},
    {0x1.780694fde5d3fp+0, 0x1.866b80a02162dp-54},
    {0x1.7a11473eb0187p+0, -0x1.41577ee04992fp-55},
    {0x1.7c1ed0130c132p+0, 0x1.f124cd1164dd6p-54},
    {0x1.7e2f336cf4e62p+0, 0x1.05d02ba15797ep-56},
    {0x1.80427543e1a12p+0, -0x1.27c86626d972bp-54},
    {0x1.82589994cce13p+0, -0x1.d4c1dd41532d8p-54},
    {0x1.8471a4623c7adp+0, -0x1.8d684a341cdfbp-55},
    {0x1.868d99b4492edp+0, -0x1.fc6f89bd4f6bap-54},
    {0x1.88ac7d98a6699p+0, 0x1.994c2f37cb53ap-54},
    {0x1.8ace5422aa0dbp+0, 0x1.6e9f156864b27p-54},
    {0x1.8cf3216b5448cp+0, -0x1.0d55e32e9e3aap-56},
    {0x1.8f1ae99157736p+0, 0x1.5cc13a2e3976cp-55},
    {0x1.9145b0b91ffc6p+0, -0x1.dd6792e582524p-54},
    {0x1.93737b0cdc5e5p+0, -0x1.75fc781b57ebcp-57},
    {0x1.95a44cbc8520fp+0, -0x1.64b7c96a5f039p-56},
    {0x1.97d829fde4e50p+0, -0x1.d185b7c1b85d1p-54},
    {0x1.9a0f170ca07bap+0, -0x1.173bd91cee632p-54},
    {0x1.9c49182a3f090p+0, 0x1.c7c46b071f2bep-56},
    {0x1.9e86319e32323p+0, 0x1.824ca78e64c6ep-56},
    {0x1.a0c667b5de565p+0, -0x1.359495d1cd533p-54},
    {0x1.a309bec4a2d33p+0, 0x1.6305c7ddc36abp-54},
    {0x1.a5503b23e255dp+0, -0x1.d2f6edb8d41e1p-54},
    {0x1.a799e1330b358p+0, 0x1.bcb7ecac563c7p-54},
    {0x1.a9e6b5579fdbfp+0, 0x1.0fac90ef7fd31p-54},
    {0x1.ac36bbfd3f37ap+0, -0x1.f9234cae76cd0p-55},
    {0x1.ae89f995ad3adp+0, 0x1.7a1cd345dcc81p-54},
    {0x1.b0e07298db666p+0, -0x1.bdef54c80e425p-54},
    {0x1.b33a2b84f15fbp+0, -0x1.2805e3084d708p-57},
    {0x1.b59728de5593ap+0, -0x1.c71dfbbba6de3p-54},
    {0x1.b7f76f2fb5e47p+0, -0x1.5584f7e54ac3bp-56},
    {0x1.ba5b030a1064ap+0, -0x1.efcd30e54292ep-54},
    {0x1.bcc1e904bc1d2p+0, 0x1.23dd07a2d9e84p-55},
    {0x1.bf2c25bd71e09p+0, -0x1.efdca3f6b9c73p-54},
    {0x1.c199bdd85529cp+0, 0x1.11065895048ddp-55},
    {0x1.c40ab5fffd07ap+0, 0x1.b4537e083c60ap-54},
    {0x1.c67f12e57d14bp+0, 0x1.2884dff483cadp-54},
    {0x1.c8f6d9406e7b5p+0, 0x1.1acbc48805c44p-56},
    {0x1.cb720dcef9069p+0, 0x1.503cbd1e949dbp-56},
    {0x1.cdf0b555dc3fap+0, -0x1.dd83b53829d72p-55},
    {0x1.d072d4a07897cp+0, -0x1.cbc3743797a9cp-54},
    {0x1.d2f87080d89f2p+0, -0x1.d487b719d8578p-54},
    {0x1.d5818dcfba487p+0, 0x1.2ed02d75b3707p-55},
    {0x1.d80e316c98398p+0, -0x1.11ec18beddfe8p-54},
    {0x1.da9e603db3285p+0, 0x1.c2300696db532p-54},
    {0x1.dd321f301b460p+0, 0x1.2da5778f018c3p-54},
    {0x1.dfc97337b9b5fp+0, -0x1.1a5cd4f184b5cp-54},
    {0x1.e264614f5a129p+0, -0x1.7b627817a1496p-54},
    {0x1.e502ee78b3ff6p+0, 0x1.39e8980a9cc8fp-55},
    {0x1.e7a51fbc74c83p+0, 0x1.2d522ca0c8de2p-54},
    {0x1.ea4afa2a490dap+0, -0x1.e9c23179c2893p-54},
    {0x1.ecf482d8e67f1p+0, -0x1.c93f3b411ad8cp-54},
    {0x1.efa1bee615a27p+0, 0x1.dc7f486a4b6b0p-54},
    {0x1.f252b376bba97p+0, 0x1.3a1a5bf0d8e43p-54},
    {0x1.f50765b6e4540p+0, 0x1.9d3e12dd8a18bp-54},
    {0x1.f7bfdad9cbe14p+0, -0x1.dbb12d006350ap-54},
    {0x1.fa7c1819e90d8p+0, 0x1.74853f3a5931ep-55},
    {0x1.fd3c22b8f71f1p+0, 0x1.2eb74966579e7p-57},
};
__device__ __forceinline__ double phred_probability(double sigma, double log2_base_hi, double log2_base_lo) {
    // z = sigma * log2(base) as a double-double, split at the nearest multiple of 1/128
    const double z_hi = sigma * log2_base_hi;
    const double z_lo = __builtin_fma(sigma, log2_base_hi, -z_hi) + sigma * log2_base_lo;
    const double kd = __builtin_rint(z_hi * 128.0);
    const double n = __builtin_floor(kd * 0.0078125);
    const int j = static_cast<int>(kd - 128.0 * n);               // 0 .. 127
    const double r0 = z_hi - kd * 0.0078125;                      // exact
    const double r_hi = r0 + z_lo;
    const double r_lo = (r0 - r_hi) + z_lo;
    // t = r * ln 2, double-double; |t| <= 0.0028
    const double t_hi = r_hi * 0.6931471805599453;
    const double t_lo = __builtin_fma(r_hi, 0.6931471805599453, -t_hi) + (r_hi * 2.3190468138462996e-17 + r_lo * 0.6931471805599453);
    // e^t - 1 = t + t^2 (1/2 + t/6 + t^2/24 + t^3/120 + t^4/720): the powers in double (their rounding is 1e-21 of the result)
    double q = 0.001388888888888889;
    q = __builtin_fma(q, t_hi, 0.008333333333333333);
    q = __builtin_fma(q, t_hi, 0.041666666666666664);
    q = __builtin_fma(q, t_hi, 0.16666666666666666);
    q = __builtin_fma(q, t_hi, 0.5);
    q = q * (t_hi * t_hi);
    const double w = t_lo + q;
    const double p_hi = t_hi + w;
    const double p_lo = (t_hi - p_hi) + w;
    // 2^(j/128) (1 + p): the product and the sum with what their roundings leave
    const double table_hi = EXP2_TABLE[j][0], table_lo = EXP2_TABLE[j][1];
    const double product_hi = table_hi * p_hi;
    const double product_lo = __builtin_fma(table_hi, p_hi, -product_hi) + (table_hi * p_lo + table_lo * p_hi);
    const double c = product_lo + table_lo;
    const double s_hi = product_hi + c;
    const double s_lo = (product_hi - s_hi) + c;
    const double f_hi = table_hi + s_hi;
    const double e = (table_hi - f_hi) + s_hi;
    return __builtin_ldexp(f_hi + (e + s_lo), static_cast<int>(n));
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
