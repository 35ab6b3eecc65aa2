// Output records assembled on the device: see record_device.hpp.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "fastq_device_inner.hpp"
#include "record_device.hpp"

namespace phq {

namespace {

struct RecordArgs {
    const FastqTables* tables;      // device
    int32_t slot;
    uint32_t n;
    int64_t capacity;
    const uint8_t* base_planes;
    const uint8_t* quality_planes;
    const uint8_t* keep;
    const uint32_t* keep_block;
    RecordRound round;
    uint32_t output_room;           // bytes of LDS a wave has for the text of its reads (record_write)
    uint32_t* overflow;             // record_quality_control: a segment longer than the cells' cycles
    int32_t n_out, n_sink;          // the kernels of several sinks: what the plan says, for the launch's LDS layout
};

// The bytes of a record: counted (pass 1), or stored (pass 2) — into the wave's LDS, from where the wave copies its reads'
// text to the output 16 bytes a lane at a time, or straight to the output when a wave's text does not fit its room.  (Stored
// straight from the lanes, a byte at a time to 64 lines a wave instruction, the text left at 49 GB/s.)
struct CountOut {
    static constexpr bool WRITES = false;
    uint32_t count;
    __device__ __forceinline__ void put(char) { ++count; }
    __device__ __forceinline__ void room(uint32_t bytes) { count += bytes; }      // bytes whose values do not matter here
};
struct GlobalOut {
    static constexpr bool WRITES = true;
    char* at;
    __device__ __forceinline__ void put(char c) { *at++ = c; }
    __device__ __forceinline__ void room(uint32_t) {}
};
struct LdsOut {
    static constexpr bool WRITES = true;
    uint32_t at;                    // into `staged`
    __device__ __forceinline__ void put(char c) { staged[at++] = static_cast<uint8_t>(c); }
    __device__ __forceinline__ void room(uint32_t) {}
};
template <typename W>
__device__ __forceinline__ void put_literal(W& out, const char* text) { for(; *text; ++text) { out.put(*text); } }
template <typename W>
__device__ __forceinline__ void put_bytes(W& out, const char* from, uint32_t size) {
    if(W::WRITES) { for(uint32_t i = 0; i < size; ++i) { out.put(from[i]); } } else { out.room(size); }
}
template <typename W>
__device__ __forceinline__ void put_number(W& out, uint32_t value) {
    char digits[10];
    int n = 0;
    do { digits[n++] = static_cast<char>('0' + value % 10u); value /= 10u; } while(value);
    while(n > 0) { out.put(digits[--n]); }
}

// printf("%g") of a float in (0, 1]: six significant digits, round half to even on the exact binary value, trailing zeros
// dropped, d.ddddde-XX below 1e-4 (text_format.hpp put_float_g is the host's; the values here are 1 - confidence: at least
// 2^-53, at most 1)
template <typename W>
__device__ void put_float_g(W& out, float value) {
    if(!(value < 1.0f)) { out.put('1'); return; }
    const uint32_t bits = __float_as_uint(value);
    const int binary_exponent = static_cast<int>((bits >> 23) & 0xffu) - 126;       // value = fraction * 2^e, fraction in [0.5, 1)
    const unsigned long long mantissa = (bits & 0x7fffffu) | 0x800000u;             // fraction * 2^24
    const int shift = 24 - binary_exponent;                                         // value = mantissa / 2^shift, shift in [24, 77]
    int exponent = ((binary_exponent - 1) * 78913) >> 18;                           // floor(log10(value)), at most one too small
    unsigned long long digits = 0;
    for(int attempt = 0; attempt < 3; ++attempt) {
        const int scale = 5 - exponent;                                             // 6 .. 22
        unsigned __int128 power = 1;
        for(int k = 0; k < scale; ++k) { power *= 10u; }
        const unsigned __int128 numerator = static_cast<unsigned __int128>(mantissa) * power;      // < 2^24 * 2^74
        const unsigned __int128 whole = numerator >> shift;
        if(whole >= 1000000u) { ++exponent; continue; }
        if(whole < 100000u) { --exponent; continue; }
        const unsigned __int128 remainder = numerator & ((static_cast<unsigned __int128>(1) << shift) - 1);
        const unsigned __int128 half = static_cast<unsigned __int128>(1) << (shift - 1);
        digits = static_cast<unsigned long long>(whole);
        if(remainder > half || (remainder == half && (digits & 1u))) { ++digits; }
        if(digits == 1000000u) { digits = 100000u; ++exponent; }
        break;
    }
    char text[6];
    unsigned long long rest = digits;
    for(int i = 5; i >= 0; --i) { text[i] = static_cast<char>('0' + rest % 10u); rest /= 10u; }
    int significant = 6;
    while(significant > 1 && text[significant - 1] == '0') { --significant; }
    if(exponent >= 0) { out.put('1'); return; }                                     // rounded up to one
    if(exponent < -4) {
        out.put(text[0]);
        if(significant > 1) { out.put('.'); for(int i = 1; i < significant; ++i) { out.put(text[i]); } }
        out.put('e'); out.put('-');
        const int magnitude = -exponent;
        if(magnitude < 10) { out.put('0'); }
        put_number(out, static_cast<uint32_t>(magnitude));
        return;
    }
    out.put('0'); out.put('.');
    for(int i = 0; i < -exponent - 1; ++i) { out.put('0'); }
    for(int i = 0; i < significant; ++i) { out.put(text[i]); }
}

// the float as '#' and the eight hexadecimal digits of its bits: what a line bound for a BAM record states (RECORD_FORMAT_SAM_FOR_BAM)
template <typename W>
__device__ __forceinline__ void put_float_bits(W& out, float value) {
    const uint32_t bits = __float_as_uint(value);
    out.put('#');
    for(int shift = 28; shift >= 0; shift -= 4) {
        const uint32_t digit = (bits >> shift) & 15u;
        out.put(static_cast<char>(digit < 10u ? '0' + digit : 'a' + (digit - 10u)));
    }
}

// one read's view of the round: its row of the planes and results, its segments in the chunks
struct ReadView {
    const RecordArgs* args;
    const RecordPlan* plan;
    uint32_t row;
    Located at;
    const WaveInput* bytes;         // the text of the wave's reads (fastq_device_inner.hpp)

    __device__ __forceinline__ const uint8_t* codes(const RecordDecoder& d) const { return args->base_planes + args->capacity * d.plane_prefix + static_cast<int64_t>(row) * d.plane_stride; }
    __device__ __forceinline__ const uint8_t* qualities(const RecordDecoder& d) const { return args->quality_planes + args->capacity * d.plane_prefix + static_cast<int64_t>(row) * d.plane_stride; }
    __device__ __forceinline__ const DevResult* result(const RecordDecoder& d) const { return d.result_slot >= 0 ? args->round.result + static_cast<int64_t>(d.result_slot) * args->capacity + row : nullptr; }
    // the decoded barcode's ordinal in tag_array order, -1 = unclassified or no result
    __device__ __forceinline__ int ordinal(const RecordDecoder& d) const {
        const DevResult* const res = result(d);
        if(res == nullptr || res->index <= 0 || res->index >= d.n_ordinal) { return -1; }
        return plan->ints[d.ordinal_at + res->index];
    }
    // observed length of segment s of the decoder's barcode: the plane is zero filled past what the tokens brought
    __device__ __forceinline__ int32_t observed(const RecordDecoder& d, int s) const {
        const uint8_t* const oc = codes(d);
        const uint8_t* const oq = qualities(d);
        const int32_t so = d.segment_offset[s];
        int32_t len = d.barcode_length[s];
        while(len > 0 && oc[so + len - 1] == 0 && oq[so + len - 1] == 0) { --len; }
        return len;
    }
    // raw barcode of a kind: append_to_raw_*_barcode (read.h:275-278,301-305,328-332)
    __device__ uint32_t raw_size(int kind) const {
        uint32_t size = 0;
        for(int k = 0; k < plan->n_decoder; ++k) {
            const RecordDecoder& d = plan->decoder[k];
            if(d.kind != kind || d.plane_stride == 0) { continue; }
            for(int s = 0; s < d.segment_cardinality; ++s) { size += static_cast<uint32_t>(observed(d, s)); }
        }
        return size;
    }
    template <typename W>
    __device__ void put_raw(W& out, int kind, bool quality) const {
        for(int k = 0; k < plan->n_decoder; ++k) {
            const RecordDecoder& d = plan->decoder[k];
            if(d.kind != kind || d.plane_stride == 0) { continue; }
            const uint8_t* const from = quality ? qualities(d) : codes(d);
            for(int s = 0; s < d.segment_cardinality; ++s) {
                const int32_t len = observed(d, s);
                if(!W::WRITES) { out.room(static_cast<uint32_t>(len)); continue; }
                for(int32_t i = 0; i < len; ++i) {
                    const uint8_t value = from[d.segment_offset[s] + i];
                    out.put(quality ? static_cast<char>(value + 33) : plan->text_of_bam[value & 15u]);
                }
            }
        }
    }
    __device__ uint32_t corrected_size(int kind) const {
        uint32_t size = 0;
        for(int k = 0; k < plan->n_decoder; ++k) {
            const RecordDecoder& d = plan->decoder[k];
            if(d.kind != kind || d.plane_stride == 0 || !d.closed) { continue; }
            for(int s = 0; s < d.segment_cardinality; ++s) { size += static_cast<uint32_t>(observed(d, s)); }
        }
        return size;
    }
    // Position `wanted` of the corrected barcode of a kind (append_corrected, sequence.h:382-398): the codec's base, and the
    // observed quality where the observation agrees, the decoder's corrected quality elsewhere.  The S9 quirk is kept: the
    // "was this base corrected" test reads the observation at the DESTINATION length.  Visits every position when
    // wanted < 0 (f(code, quality) for each); returns false when the position does not exist.
    template <typename F>
    __device__ bool corrected(int kind, int32_t wanted, F&& f) const {
        int32_t dest = 0;
        for(int k = 0; k < plan->n_decoder; ++k) {
            const RecordDecoder& d = plan->decoder[k];
            if(d.kind != kind || d.plane_stride == 0 || !d.closed) { continue; }
            const uint8_t* const oc = codes(d);
            const uint8_t* const oq = qualities(d);
            const int which = ordinal(d);
            for(int s = 0; s < d.segment_cardinality; ++s) {
                const int32_t len = observed(d, s);
                const int32_t so = d.segment_offset[s];
                if(wanted >= 0 && wanted >= dest + len) { dest += len; continue; }
                const int32_t first = wanted >= 0 ? wanted - dest : 0;
                const int32_t last = wanted >= 0 ? first + 1 : len;
                for(int32_t i = first; i < last; ++i) {
                    const uint8_t fixed = which >= 0 ? plan->bytes[d.code_at + which * d.nucleotide_cardinality + so + i] : 0;
                    const uint8_t probe = (dest + i) < len ? oc[so + dest + i] : 0;
                    f(fixed, (probe == fixed || fixed == 0) ? oq[so + i] : static_cast<uint8_t>(d.corrected_quality));
                }
                if(wanted >= 0) { return true; }
                dest += len;
            }
        }
        return false;
    }
    template <typename W>
    __device__ void put_corrected(W& out, int kind, bool quality) const {
        if(!W::WRITES) { out.room(corrected_size(kind)); return; }
        corrected(kind, -1, [&](uint8_t code, uint8_t q) { out.put(quality ? static_cast<char>(q + 33) : plan->text_of_bam[code & 15u]); });
    }
    // the per type confidence as the decoders' wrappers leave it (pamld.cpp:133-180; read.h:176-183,279-285)
    __device__ void confidences(double (&confidence)[3]) const {
        for(int t = 0; t < 3; ++t) { confidence[t] = 1; }
        for(int k = 0; k < plan->n_decoder; ++k) {
            const RecordDecoder& d = plan->decoder[k];
            if(d.result_slot < 0 || !d.pamld) { continue; }
            const DevResult& res = *result(d);
            if(d.kind == 0 || res.index > 0) { confidence[d.kind] = confidence[d.kind] == 1 ? res.confidence : confidence[d.kind] * res.confidence; }
            else { confidence[d.kind] = 0; }
        }
    }
    // an output segment's bases or qualities from the template tokens (TemplateRule::apply, transform.h:190-226)
    __device__ uint32_t token_size(const RecordToken& token, int32_t& start, int32_t& end) const {
        const int32_t length = token.source >= 0 ? static_cast<int32_t>(at.length[token.source])
                                                 : static_cast<int32_t>(corrected_size(token.source == -1 ? 0 : token.source == -2 ? 2 : 1));
        token_range(token.start, token.end, token.end_terminated, length, start, end);
        return end > start ? static_cast<uint32_t>(end - start) : 0u;
    }
    __device__ uint32_t bases_of(int o) const {
        uint32_t size = 0;
        for(int t = plan->out_first[o]; t < plan->out_first[o + 1]; ++t) { int32_t start, end; size += token_size(plan->token[plan->out_token[t]], start, end); }
        return size;
    }
    template <typename W>
    __device__ void put_segment(W& out, int o, bool quality) const {
        for(int t = plan->out_first[o]; t < plan->out_first[o + 1]; ++t) {
            const RecordToken& token = plan->token[plan->out_token[t]];
            int32_t start, end;
            const uint32_t size = token_size(token, start, end);
            if(!W::WRITES) { out.room(size); continue; }
            if(token.source >= 0) {
                const FeedBytes text = bytes->feed(args->tables->plan.segment_feed[token.source]);
                const uint32_t from = quality ? at.quality_at[token.source] : at.sequence_at[token.source];
                // four bytes a trip: their reads, then their table lookups, are in flight together (one byte a trip is a chain
                // of an LDS read, a table read and an LDS write, and a wave has nothing else to do while it waits)
                const uint32_t first = from + static_cast<uint32_t>(token.reverse_complement ? end - 1 : start);
                const uint32_t step = token.reverse_complement ? ~0u : 1u;       // - 1 or + 1
                const char* const table = token.reverse_complement ? plan->complement_text : plan->canonical_text;
                const char shift = static_cast<char>(plan->quality_shift);
                uint32_t i = 0;
                for(; i + 4u <= size; i += 4u) {
                    const uint8_t c0 = text(first + step * i), c1 = text(first + step * (i + 1u)), c2 = text(first + step * (i + 2u)), c3 = text(first + step * (i + 3u));
                    const char t0 = quality ? static_cast<char>(c0 + shift) : table[c0], t1 = quality ? static_cast<char>(c1 + shift) : table[c1];
                    const char t2 = quality ? static_cast<char>(c2 + shift) : table[c2], t3 = quality ? static_cast<char>(c3 + shift) : table[c3];
                    out.put(t0); out.put(t1); out.put(t2); out.put(t3);
                }
                for(; i < size; ++i) {
                    const uint8_t c = text(first + step * i);
                    out.put(quality ? static_cast<char>(c + shift) : table[c]);
                }
            } else {
                const int kind = token.source == -1 ? 0 : token.source == -2 ? 2 : 1;
                for(uint32_t i = 0; i < size; ++i) {
                    const int32_t from = token.reverse_complement ? end - 1 - static_cast<int32_t>(i) : start + static_cast<int32_t>(i);
                    corrected(kind, from, [&](uint8_t code, uint8_t q) {
                        out.put(quality ? static_cast<char>(q + plan->output_phred_offset)
                                        : plan->text_of_bam[(token.reverse_complement ? plan->complement_of_bam[code & 15u] : code) & 15u]);
                    });
                }
            }
        }
    }
    // the bases of output segment o as the reference's Segment holds them: BAM code and Phred value (what the quality
    // control accumulators count, multiplex.h:219-221)
    template <typename F>
    __device__ void for_each_base(int o, F&& f) const {
        for(int t = plan->out_first[o]; t < plan->out_first[o + 1]; ++t) {
            const RecordToken& token = plan->token[plan->out_token[t]];
            int32_t start, end;
            const uint32_t size = token_size(token, start, end);
            if(token.source >= 0) {
                const FeedBytes text = bytes->feed(args->tables->plan.segment_feed[token.source]);
                for(uint32_t i = 0; i < size; ++i) {
                    const uint32_t from = static_cast<uint32_t>(token.reverse_complement ? end - 1 - static_cast<int32_t>(i) : start + static_cast<int32_t>(i));
                    uint8_t code = args->tables->bam_of_ascii[text(at.sequence_at[token.source] + from)];
                    if(token.reverse_complement) { code = plan->complement_of_bam[code & 15u]; }
                    f(code, static_cast<uint8_t>(text(at.quality_at[token.source] + from) - plan->input_phred_offset));
                }
            } else {
                const int kind = token.source == -1 ? 0 : token.source == -2 ? 2 : 1;
                for(uint32_t i = 0; i < size; ++i) {
                    const int32_t from = token.reverse_complement ? end - 1 - static_cast<int32_t>(i) : start + static_cast<int32_t>(i);
                    corrected(kind, from, [&](uint8_t code, uint8_t q) { f(token.reverse_complement ? plan->complement_of_bam[code & 15u] : code, q); });
                }
            }
        }
    }
    // Multiplexer::push (multiplex.h:198-240): the channel is the multiplexing decoder's barcode; `group` the sample barcode's read group
    __device__ int channel_of(int& group) const {
        int channel = 0;
        group = -1;
        for(int k = 0; k < plan->n_decoder; ++k) {
            const RecordDecoder& d = plan->decoder[k];
            if(d.plane_stride == 0) { continue; }
            const int which = ordinal(d);
            if(d.multiplexing) { channel = which + 1; }
            if(d.result_slot >= 0 && d.kind == 0) { group = which + 1; }
        }
        return channel >= plan->n_channel ? 0 : channel;
    }
};

// Read::flush (read.h:187-237) + Auxiliary::encode order (auxiliary.cpp:327-354): the same tags on every segment
template <typename W>
__device__ void put_tags(W& out, const ReadView& read, int group, const double (&confidence)[3], const uint32_t (&raw_size)[3], const uint32_t (&corrected_size)[3], bool float_bits) {
    const RecordPlan& plan = *read.plan;
    // a line the host turns into a BAM record carries the float itself, not six of its digits (bam_record.hpp)
    auto put_float = [&](float value) { if(float_bits) { put_float_bits(out, value); } else { put_float_g(out, value); } };
    if(group >= 0 && group < plan.n_group) {
        const int32_t first = plan.ints[plan.group_at + group], last = plan.ints[plan.group_at + group + 1];
        if(last > first) { put_literal(out, "\tRG:Z:"); put_bytes(out, reinterpret_cast<const char*>(plan.bytes) + first, static_cast<uint32_t>(last - first)); }
    }
    if(raw_size[0]) { put_literal(out, "\tBC:Z:"); read.put_raw(out, 0, false); put_literal(out, "\tQT:Z:"); read.put_raw(out, 0, true); }
    if(confidence[0] > 0 && confidence[0] < 1) { put_literal(out, "\tXB:f:"); put_float(static_cast<float>(1.0 - confidence[0])); }
    if(corrected_size[1]) { put_literal(out, "\tRX:Z:"); read.put_corrected(out, 1, false); put_literal(out, "\tQX:Z:"); read.put_corrected(out, 1, true); }
    if(raw_size[1]) { put_literal(out, "\tOX:Z:"); read.put_raw(out, 1, false); put_literal(out, "\tBZ:Z:"); read.put_raw(out, 1, true); }
    if(confidence[1] > 0 && confidence[1] < 1) { put_literal(out, "\tXM:f:"); put_float(static_cast<float>(1.0 - confidence[1])); }
    if(corrected_size[2]) { put_literal(out, "\tCB:Z:"); read.put_corrected(out, 2, false); }
    if(raw_size[2]) { put_literal(out, "\tCR:Z:"); read.put_raw(out, 2, false); put_literal(out, "\tCY:Z:"); read.put_raw(out, 2, true); }
    if(confidence[2] > 0 && confidence[2] < 1) { put_literal(out, "\tXC:f:"); put_float(static_cast<float>(1.0 - confidence[2])); }
}

// what the records of a read share: its channel, its read group, the tags' ingredients
struct ReadTags {
    int channel, group;
    bool qcfail;
    double confidence[3];
    uint32_t raw_size[3], corrected_size[3];
};

// Multiplexer::push -> Channel::push (multiplex.h:198-240); false: no record of the read is written
__device__ bool prepare_tags(const ReadView& read, ReadTags& tags) {
    const RecordPlan& plan = *read.plan;
    tags.channel = read.channel_of(tags.group);
    const uint8_t mode = plan.bytes[plan.channel_at + tags.channel];
    tags.qcfail = (read.args->round.word[read.row] >> 31) != 0;
    if(!(mode & 1u) || ((mode & 2u) && tags.qcfail)) { return false; }
    read.confidences(tags.confidence);
    for(int kind = 0; kind < 3; ++kind) { tags.raw_size[kind] = read.raw_size(kind); tags.corrected_size[kind] = read.corrected_size(kind); }
    return true;
}
// where output segment o of a channel goes: a sink of the plan, or -1
__device__ __forceinline__ int sink_of(const RecordPlan& plan, int channel, int o) { return plan.ints[plan.sink_at + channel * plan.n_out + o]; }

// one record: output segment o of the read, as a FASTQ record or a SAM line (format: RECORD_FORMAT_* of its sink)
template <typename W>
__device__ void put_record(W& out, const ReadView& read, const ReadTags& tags, int o, uint8_t format) {
    const bool fastq = format == RECORD_FORMAT_FASTQ;
    const RecordPlan& plan = *read.plan;
    const FeedBytes name_text = read.bytes->feed(read.args->tables->plan.segment_feed[0]);
    auto put_name = [&]() {
        if(!W::WRITES) { out.room(read.at.name_size); return; }
        uint32_t i = 0;
        for(; i + 4u <= read.at.name_size; i += 4u) {
            const uint8_t c0 = name_text(read.at.name_at + i), c1 = name_text(read.at.name_at + i + 1u), c2 = name_text(read.at.name_at + i + 2u), c3 = name_text(read.at.name_at + i + 3u);
            out.put(static_cast<char>(c0)); out.put(static_cast<char>(c1)); out.put(static_cast<char>(c2)); out.put(static_cast<char>(c3));
        }
        for(; i < read.at.name_size; ++i) { out.put(static_cast<char>(name_text(read.at.name_at + i))); }
    };
    if(fastq) {
        // FastqRecord::decode(segment) + encode (fastq.h:79-88,144-171), comment fastq.h:176-194
        out.put('@');
        put_name();
        out.put(' ');
        if(plan.illumina_comment) {
            put_number(out, static_cast<uint32_t>(o + 1));
            put_literal(out, tags.qcfail ? ":Y:0:" : ":N:0:");
            read.put_raw(out, 0, false);
        }
        out.put('\n');
        read.put_segment(out, o, false);
        put_literal(out, "\n+\n");
        read.put_segment(out, o, true);
        out.put('\n');
        return;
    }
    put_name();
    put_bytes(out, plan.columns[o][tags.qcfail ? 1 : 0], static_cast<uint32_t>(plan.columns_size[o][tags.qcfail ? 1 : 0]));
    if(read.bases_of(o) == 0) { put_literal(out, "*\t*"); }
    else {
        read.put_segment(out, o, false);
        out.put('\t');
        read.put_segment(out, o, true);
    }
    if(plan.n_out > 2) {
        put_literal(out, "\tFI:i:"); put_number(out, static_cast<uint32_t>(o + 1));
        put_literal(out, "\tTC:i:"); put_number(out, static_cast<uint32_t>(plan.n_out));
    }
    put_tags(out, read, tags.group, tags.confidence, tags.raw_size, tags.corrected_size, format == RECORD_FORMAT_SAM_FOR_BAM);
    out.put('\n');
}

// every record of one read, in output segment order (a plan of one sink: they are consecutive in it)
template <typename W>
__device__ void put_read(W& out, const ReadView& read) {
    const RecordPlan& plan = *read.plan;
    ReadTags tags;
    if(!prepare_tags(read, tags)) { return; }
    for(int o = 0; o < plan.n_out; ++o) {
        const int sink = sink_of(plan, tags.channel, o);
        if(sink >= 0) { put_record(out, read, tags, o, plan.bytes[plan.format_at + sink]); }
    }
}

// pass 1: the bytes of every read's records, and their sum per workgroup
template <bool FILTER>
__global__ __launch_bounds__(TILE_THREADS) void record_size_kernel(RecordArgs args) {
    __shared__ uint32_t wave_kept[TILE_WAVES];
    __shared__ unsigned long long block_bytes;
    if(threadIdx.x == 0) { block_bytes = 0; }
    const FastqPlan& plan = args.tables->plan;
    const FastqFeedChunk* const chunk = args.tables->chunk[args.slot];
    uint32_t wave_first, wave_count;
    bool present;
    const uint32_t r = read_of_lane(plan, args.n, wave_first, wave_count, present);
    WaveInput bytes_of;
    stage_wave_input(plan, chunk, wave_first, wave_count, (threadIdx.x >> 6) * static_cast<uint32_t>(plan.input_room), bytes_of);
    bool wanted;
    const uint32_t row = row_of<FILTER>(r, present, args.keep, args.keep_block, wave_kept, wanted);
    uint32_t bytes = 0;
    if(wanted) {
        ReadView read;
        read.args = &args;
        read.plan = args.round.plan;
        read.row = row;
        read.bytes = &bytes_of;
        locate(plan, chunk, bytes_of, r, read.at);
        CountOut out{0};
        put_read(out, read);
        bytes = out.count;
    }
    if(present) { args.round.size[r] = bytes; }
    uint32_t sum = bytes;
    for(int step = 32; step > 0; step >>= 1) { sum += __shfl_down(sum, step); }
    if((threadIdx.x & 63) == 0 && sum) { atomicAdd(&block_bytes, static_cast<unsigned long long>(sum)); }
    __syncthreads();
    if(threadIdx.x == 0) { args.round.block[blockIdx.x] = block_bytes; }
}

// The quality control accumulators (record_device.hpp): one lane per read, a global atomic per base.  A wave's lanes are
// at the same cycle of 64 different reads — mostly of different channels, so of different cells — and global atomics run at
// the memory side, a request per lane: ~46 G a second spread like this.
template <bool FILTER>
__global__ __launch_bounds__(TILE_THREADS) void record_quality_control_kernel(RecordArgs args) {
    __shared__ uint32_t wave_kept[TILE_WAVES];
    const FastqPlan& plan = args.tables->plan;
    const FastqFeedChunk* const chunk = args.tables->chunk[args.slot];
    uint32_t wave_first, wave_count;
    bool present;
    const uint32_t r = read_of_lane(plan, args.n, wave_first, wave_count, present);
    WaveInput bytes_of;
    stage_wave_input(plan, chunk, wave_first, wave_count, (threadIdx.x >> 6) * static_cast<uint32_t>(plan.input_room), bytes_of);
    bool wanted;
    const uint32_t row = row_of<FILTER>(r, present, args.keep, args.keep_block, wave_kept, wanted);
    if(!wanted) { return; }
    ReadView read;
    read.args = &args;
    read.plan = args.round.plan;
    read.row = row;
    read.bytes = &bytes_of;
    locate(plan, chunk, bytes_of, r, read.at);
    int group;
    const int channel = read.channel_of(group);
    const size_t cell_words = qc_cell_words(args.round.cycles);
    for(int o = 0; o < read.plan->n_out; ++o) {
        unsigned long long* const cell = args.round.quality_control + (static_cast<size_t>(channel) * static_cast<size_t>(read.plan->n_out) + static_cast<size_t>(o)) * cell_words;
        if(static_cast<int32_t>(read.bases_of(o)) > args.round.cycles) { atomicOr(args.overflow, 1u); continue; }
        uint32_t length = 0;
        unsigned long long total = 0;
        read.for_each_base(o, [&](uint8_t code, uint8_t quality) {
            const uint32_t q = quality < QC_CELL_PHRED ? quality : QC_CELL_PHRED - 1;
            atomicAdd(&cell[QC_CELL_HEAD + (static_cast<size_t>(length) * QC_CELL_CODES + (code & 15u)) * QC_CELL_PHRED + q], 1ull);
            total += quality;
            ++length;
        });
        atomicMax(&cell[0], static_cast<unsigned long long>(length));
        atomicMax(&cell[1], 0x7fffffffull - static_cast<unsigned long long>(length));              // the shortest, so that an untouched cell reads 0x7fffffff
        if(length > 0) {                                                     // AveragePhreadAccumulator::increment, multiplex.h:44-55
            const double value = static_cast<double>(total) / static_cast<double>(length);
            atomicAdd(&cell[2], 1ull);
            atomicAdd(&cell[3], static_cast<unsigned long long>(value * 16777216.0 + 0.5));
            atomicMax(&cell[4], static_cast<unsigned long long>(__double_as_longlong(value)));      // means are not negative: their bits order as they do
            const unsigned long long bin = static_cast<unsigned long long>(value);
            atomicAdd(&cell[5 + (bin < QC_CELL_PHRED ? bin : QC_CELL_PHRED - 1)], 1ull);
        }
    }
}

// exclusive prefix sums of the workgroups' bytes, in place; *total = the bytes of the round's text
__global__ __launch_bounds__(1024) void record_scan_kernel(unsigned long long* block, uint32_t n_block, unsigned long long* total) {
    __shared__ unsigned long long wave_sum[16];
    __shared__ unsigned long long carried;
    if(threadIdx.x == 0) { carried = 0; }
    __syncthreads();
    for(uint32_t first = 0; first < n_block; first += 1024) {
        const uint32_t at = first + threadIdx.x;
        const unsigned long long own = at < n_block ? block[at] : 0ull;
        unsigned long long inclusive = own;
        for(int step = 1; step < 64; step <<= 1) {
            const unsigned long long below = __shfl_up(inclusive, step);
            if(static_cast<int>(threadIdx.x & 63) >= step) { inclusive += below; }
        }
        if((threadIdx.x & 63) == 63) { wave_sum[threadIdx.x >> 6] = inclusive; }
        __syncthreads();
        unsigned long long before = carried;
        for(uint32_t w = 0; w < (threadIdx.x >> 6); ++w) { before += wave_sum[w]; }
        if(at < n_block) { block[at] = before + inclusive - own; }
        __syncthreads();
        if(threadIdx.x == 1023) { carried = before + inclusive; }
        __syncthreads();
    }
    if(threadIdx.x == 0) { *total = carried; }
}

// pass 2: the text, every read's records where the scan put them.  A wave's reads are consecutive in the output: it puts
// their text together in LDS — at the offset that gives LDS and output addresses the same remainder by 16 — and copies it out
// as whole units of 16 bytes a lane, the ragged ends byte by byte.
template <bool FILTER>
__global__ __launch_bounds__(TILE_THREADS) void record_write_kernel(RecordArgs args) {
    __shared__ uint32_t wave_kept[TILE_WAVES];
    __shared__ uint32_t wave_bytes[TILE_WAVES];
    const FastqPlan& plan = args.tables->plan;
    const FastqFeedChunk* const chunk = args.tables->chunk[args.slot];
    uint32_t wave_first, wave_count;
    bool present;
    const uint32_t r = read_of_lane(plan, args.n, wave_first, wave_count, present);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t wave_room = static_cast<uint32_t>(plan.input_room) + args.output_room + 16u;
    WaveInput bytes_of;
    stage_wave_input(plan, chunk, wave_first, wave_count, wave * wave_room, bytes_of);
    bool wanted;
    const uint32_t row = row_of<FILTER>(r, present, args.keep, args.keep_block, wave_kept, wanted);
    const uint32_t own = present ? args.round.size[r] : 0u;
    uint32_t inclusive = own;
    for(int step = 1; step < 64; step <<= 1) {
        const uint32_t below = __shfl_up(inclusive, step);
        if(static_cast<int>(lane) >= step) { inclusive += below; }
    }
    const uint32_t wave_total = static_cast<uint32_t>(__shfl(static_cast<int>(inclusive), 63));
    if(lane == 63) { wave_bytes[wave] = inclusive; }
    __syncthreads();
    unsigned long long wave_place = args.round.block[blockIdx.x];
    for(uint32_t w = 0; w < wave; ++w) { wave_place += wave_bytes[w]; }
    char* const wave_text = args.round.text + wave_place;
    const uint32_t skew = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(wave_text)) & 15u;
    const bool through_lds = wave_total + skew <= args.output_room + 16u;
    const uint32_t out_at = wave * wave_room + static_cast<uint32_t>(plan.input_room);       // a multiple of 16
    if(wanted && own != 0) {
        ReadView read;
        read.args = &args;
        read.plan = args.round.plan;
        read.row = row;
        read.bytes = &bytes_of;
        locate(plan, chunk, bytes_of, r, read.at);
        if(through_lds) {
            LdsOut out{out_at + skew + inclusive - own};
            put_read(out, read);
        } else {
            GlobalOut out{wave_text + (inclusive - own)};
            put_read(out, read);
        }
    }
    if(!through_lds || wave_total == 0) { return; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t head = min(wave_total, (16u - skew) & 15u);
    if(lane < head) { wave_text[lane] = static_cast<char>(staged[out_at + skew + lane]); }
    const uint32_t units = (wave_total - head) >> 4;
    for(uint32_t u = lane; u < units; u += 64u) {
        reinterpret_cast<uint4*>(wave_text + head)[u] = *reinterpret_cast<const uint4*>(&staged[out_at + skew + head + 16u * u]);
    }
    const uint32_t tail_at = head + 16u * units;
    if(lane < wave_total - tail_at) { wave_text[tail_at + lane] = static_cast<char>(staged[out_at + skew + tail_at + lane]); }
}

// ---- several sinks ----------------------------------------------------------------------------------------------------
// pass 1: the bytes of every record, and per workgroup the bytes every sink gets
template <bool FILTER>
__global__ __launch_bounds__(TILE_THREADS) void record_size_sinks_kernel(RecordArgs args) {
    __shared__ uint32_t wave_kept[TILE_WAVES];
    const FastqPlan& plan = args.tables->plan;
    const FastqFeedChunk* const chunk = args.tables->chunk[args.slot];
    uint32_t* const sink_bytes = reinterpret_cast<uint32_t*>(&staged[static_cast<uint32_t>(TILE_WAVES) * static_cast<uint32_t>(plan.input_room)]);      // [n_sink]
    for(int s = threadIdx.x; s < args.n_sink; s += TILE_THREADS) { sink_bytes[s] = 0; }
    uint32_t wave_first, wave_count;
    bool present;
    const uint32_t r = read_of_lane(plan, args.n, wave_first, wave_count, present);
    WaveInput bytes_of;
    stage_wave_input(plan, chunk, wave_first, wave_count, (threadIdx.x >> 6) * static_cast<uint32_t>(plan.input_room), bytes_of);
    bool wanted;
    const uint32_t row = row_of<FILTER>(r, present, args.keep, args.keep_block, wave_kept, wanted);
    if(present) {
        ReadView read;
        read.args = &args;
        read.plan = args.round.plan;
        read.row = row;
        read.bytes = &bytes_of;
        ReadTags tags;
        bool written = false;
        if(wanted) {
            locate(plan, chunk, bytes_of, r, read.at);
            written = prepare_tags(read, tags);
        }
        for(int o = 0; o < args.n_out; ++o) {
            uint32_t bytes = 0;
            const int sink = written ? sink_of(*read.plan, tags.channel, o) : -1;
            if(sink >= 0) {
                CountOut out{0};
                put_record(out, read, tags, o, read.plan->bytes[read.plan->format_at + sink]);
                bytes = out.count;
                atomicAdd(&sink_bytes[sink], bytes);
            }
            args.round.piece_size[static_cast<size_t>(r) * static_cast<size_t>(args.n_out) + static_cast<size_t>(o)] = bytes;
        }
    }
    __syncthreads();
    for(int s = threadIdx.x; s < args.n_sink; s += TILE_THREADS) {
        args.round.sink_block[static_cast<size_t>(blockIdx.x) * static_cast<size_t>(args.n_sink) + static_cast<size_t>(s)] = sink_bytes[s];
    }
}

// one workgroup per sink: exclusive prefix sums of its bytes over the workgroups of pass 1, in place; its total to sink_base[sink]
__global__ __launch_bounds__(256) void record_scan_sinks_kernel(unsigned long long* sink_block, uint32_t n_block, int32_t n_sink, unsigned long long* sink_base) {
    __shared__ unsigned long long wave_sum[4];
    __shared__ unsigned long long carried;
    if(threadIdx.x == 0) { carried = 0; }
    __syncthreads();
    const size_t sink = blockIdx.x;
    for(uint32_t first = 0; first < n_block; first += 256) {
        const uint32_t at = first + threadIdx.x;
        const unsigned long long own = at < n_block ? sink_block[static_cast<size_t>(at) * static_cast<size_t>(n_sink) + sink] : 0ull;
        unsigned long long inclusive = own;
        for(int step = 1; step < 64; step <<= 1) {
            const unsigned long long below = __shfl_up(inclusive, step);
            if(static_cast<int>(threadIdx.x & 63) >= step) { inclusive += below; }
        }
        if((threadIdx.x & 63) == 63) { wave_sum[threadIdx.x >> 6] = inclusive; }
        __syncthreads();
        unsigned long long before = carried;
        for(uint32_t w = 0; w < (threadIdx.x >> 6); ++w) { before += wave_sum[w]; }
        if(at < n_block) { sink_block[static_cast<size_t>(at) * static_cast<size_t>(n_sink) + sink] = before + inclusive - own; }
        __syncthreads();
        if(threadIdx.x == 255) { carried = before + inclusive; }
        __syncthreads();
    }
    if(threadIdx.x == 0) { sink_base[sink] = carried; }
}

// the sinks' totals to where each one's records start in the text; sink_base[n_sink] = the whole text
__global__ void record_base_kernel(unsigned long long* sink_base, int32_t n_sink) {
    if(threadIdx.x != 0 || blockIdx.x != 0) { return; }
    unsigned long long before = 0;
    for(int s = 0; s < n_sink; ++s) { const unsigned long long own = sink_base[s]; sink_base[s] = before; before += own; }
    sink_base[n_sink] = before;
}

// pass 2: a read's records are put together in the wave's LDS, one after the other, and every record then leaves for its
// place in its sink's part of the text, the wave's lanes a byte each (a record is consecutive bytes: whole lines again)
template <bool FILTER>
__global__ __launch_bounds__(TILE_THREADS) void record_write_sinks_kernel(RecordArgs args) {
    __shared__ uint32_t wave_kept[TILE_WAVES];
    const FastqPlan& plan = args.tables->plan;
    const FastqFeedChunk* const chunk = args.tables->chunk[args.slot];
    uint32_t wave_first, wave_count;
    bool present;
    const uint32_t r = read_of_lane(plan, args.n, wave_first, wave_count, present);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t wave_room = static_cast<uint32_t>(plan.input_room) + args.output_room + 16u;
    // behind the waves' text: sink and bytes of every record of the workgroup's reads, in (read, output segment) order
    const uint32_t pieces = static_cast<uint32_t>(TILE_WAVES * plan.reads_per_wave * args.n_out);
    uint32_t* const piece_bytes = reinterpret_cast<uint32_t*>(&staged[TILE_WAVES * wave_room]);
    int32_t* const piece_sink = reinterpret_cast<int32_t*>(piece_bytes + pieces);
    WaveInput bytes_of;
    stage_wave_input(plan, chunk, wave_first, wave_count, wave * wave_room, bytes_of);
    bool wanted;
    const uint32_t row = row_of<FILTER>(r, present, args.keep, args.keep_block, wave_kept, wanted);
    ReadView read;
    read.args = &args;
    read.plan = args.round.plan;
    read.row = row;
    read.bytes = &bytes_of;
    ReadTags tags;
    bool written = false;
    if(wanted) {
        locate(plan, chunk, bytes_of, r, read.at);
        written = prepare_tags(read, tags);
    }
    const uint32_t local = (wave * static_cast<uint32_t>(plan.reads_per_wave) + lane) * static_cast<uint32_t>(args.n_out);      // the read's first piece
    uint32_t own = 0;
    if(lane < static_cast<uint32_t>(plan.reads_per_wave)) {
        for(int o = 0; o < args.n_out; ++o) {
            const uint32_t bytes = present ? args.round.piece_size[static_cast<size_t>(r) * static_cast<size_t>(args.n_out) + static_cast<size_t>(o)] : 0u;
            piece_bytes[local + o] = bytes;
            piece_sink[local + o] = bytes ? sink_of(*read.plan, tags.channel, o) : -1;
            own += bytes;
        }
    }
    uint32_t inclusive = own;
    for(int step = 1; step < 64; step <<= 1) {
        const uint32_t below = __shfl_up(inclusive, step);
        if(static_cast<int>(lane) >= step) { inclusive += below; }
    }
    const uint32_t wave_total = static_cast<uint32_t>(__shfl(static_cast<int>(inclusive), 63));
    __syncthreads();
    // where each of the read's records goes: its sink's part of the text, the workgroup's place in it (pass 1's prefix sums),
    // and the bytes of the workgroup's earlier records for the same sink
    char* place[RECORD_MAX_OUT];
    uint32_t bytes_of_piece[RECORD_MAX_OUT];
#pragma unroll
    for(int o = 0; o < RECORD_MAX_OUT; ++o) {
        place[o] = nullptr;
        bytes_of_piece[o] = 0;
        if(o < args.n_out && own != 0 && piece_bytes[local + o] != 0) {
            const int sink = piece_sink[local + o];
            unsigned long long before = args.round.sink_base[sink] + args.round.sink_block[static_cast<size_t>(blockIdx.x) * static_cast<size_t>(args.n_sink) + static_cast<size_t>(sink)];
            for(uint32_t e = 0; e < local + static_cast<uint32_t>(o); ++e) { before += piece_sink[e] == sink ? piece_bytes[e] : 0u; }
            place[o] = args.round.text + before;
            bytes_of_piece[o] = piece_bytes[local + o];
        }
    }
    const bool through_lds = wave_total <= args.output_room;
    const uint32_t out_at = wave * wave_room + static_cast<uint32_t>(plan.input_room);
    if(written && own != 0) {
        if(through_lds) {
            LdsOut out{out_at + inclusive - own};
#pragma unroll
            for(int o = 0; o < RECORD_MAX_OUT; ++o) {
                if(o < args.n_out && bytes_of_piece[o] != 0) { put_record(out, read, tags, o, read.plan->bytes[read.plan->format_at + piece_sink[local + o]]); }
            }
        } else {
#pragma unroll
            for(int o = 0; o < RECORD_MAX_OUT; ++o) {
                if(o < args.n_out && bytes_of_piece[o] != 0) {
                    GlobalOut out{place[o]};
                    put_record(out, read, tags, o, read.plan->bytes[read.plan->format_at + piece_sink[local + o]]);
                }
            }
        }
    }
    if(!through_lds || wave_total == 0) { return; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t from = out_at + inclusive - own;            // the lane's first record in LDS
#pragma unroll
    for(int o = 0; o < RECORD_MAX_OUT; ++o) {
        if(o >= args.n_out) { break; }
        for(int owner = 0; owner < plan.reads_per_wave; ++owner) {
            const uint32_t bytes = static_cast<uint32_t>(__shfl(static_cast<int>(bytes_of_piece[o]), owner));
            if(bytes == 0) { continue; }
            const uint32_t source = static_cast<uint32_t>(__shfl(static_cast<int>(from), owner));
            const unsigned long long target = (static_cast<unsigned long long>(static_cast<uint32_t>(__shfl(static_cast<int>(reinterpret_cast<uintptr_t>(place[o]) >> 32), owner))) << 32) |
                                              static_cast<uint32_t>(__shfl(static_cast<int>(reinterpret_cast<uintptr_t>(place[o]) & 0xffffffffu), owner));
            char* const to = reinterpret_cast<char*>(target);
            for(uint32_t k = lane; k < bytes; k += 64u) { to[k] = static_cast<char>(staged[source + k]); }
        }
        from += bytes_of_piece[o];
    }
}

RecordArgs arguments_of(const FastqTables* on_device, int32_t slot, uint32_t n, int64_t capacity, const uint8_t* base_planes, const uint8_t* quality_planes,
                        const uint8_t* keep, const uint32_t* keep_block, const RecordRound& round, uint32_t output_room) {
    RecordArgs call;
    call.tables = on_device;
    call.slot = slot;
    call.n = n;
    call.capacity = capacity;
    call.base_planes = base_planes;
    call.quality_planes = quality_planes;
    call.keep = keep;
    call.keep_block = keep_block;
    call.round = round;
    call.output_room = output_room;
    call.overflow = nullptr;
    call.n_out = 0;
    call.n_sink = 0;
    return call;
}

// LDS a wave of record_write has for its reads' text: what the CU's 160 KiB leave four waves beside their input
uint32_t output_room_of(const FastqPlan& plan) {
    // (twice the input's room is enough for the text of most jobs and lets a second workgroup share the CU)
    int32_t room = std::min((152 * 1024) / TILE_WAVES - plan.input_room - 16, 2 * plan.input_room);
    if(const char* text = std::getenv("PHQ_RECORD_OUTPUT_ROOM")) { room = std::min((152 * 1024) / TILE_WAVES - plan.input_room - 16, std::atoi(text)); }      // sweeps
    return static_cast<uint32_t>(std::max(room, 0) & ~15);
}

}  // namespace

int record_sizes(const FastqTables* on_device, const FastqPlan& plan, int32_t slot, uint32_t n, int64_t capacity, const uint8_t* base_planes, const uint8_t* quality_planes,
                 const uint8_t* keep, const uint32_t* keep_block, const RecordRound& round, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t blocks = n > 0 ? fastq_groups(plan, n) : 0u;
    if(blocks > 0) {
        const RecordArgs call = arguments_of(on_device, slot, n, capacity, base_planes, quality_planes, keep, keep_block, round, 0);
        const size_t lds = static_cast<size_t>(TILE_WAVES) * static_cast<size_t>(plan.input_room);
        static std::atomic<uint64_t> prepared[2];
        if(const int status = allow_lds(&record_size_kernel<true>, prepared[0], lds)) { return status; }
        if(const int status = allow_lds(&record_size_kernel<false>, prepared[1], lds)) { return status; }
        if(plan.filtering) { hipLaunchKernelGGL(record_size_kernel<true>, dim3(blocks), dim3(TILE_THREADS), lds, s, call); }
        else { hipLaunchKernelGGL(record_size_kernel<false>, dim3(blocks), dim3(TILE_THREADS), lds, s, call); }
    }
    hipLaunchKernelGGL(record_scan_kernel, dim3(1), dim3(1024), 0, s, round.block, blocks, round.total);
    return static_cast<int>(hipGetLastError());
}

int record_write(const FastqTables* on_device, const FastqPlan& plan, int32_t slot, uint32_t n, int64_t capacity, const uint8_t* base_planes, const uint8_t* quality_planes,
                 const uint8_t* keep, const uint32_t* keep_block, const RecordRound& round, void* stream) {
    if(n == 0) { return 0; }
    const uint32_t blocks = fastq_groups(plan, n);
    const uint32_t output_room = output_room_of(plan);
    const RecordArgs call = arguments_of(on_device, slot, n, capacity, base_planes, quality_planes, keep, keep_block, round, output_room);
    const size_t lds = static_cast<size_t>(TILE_WAVES) * (static_cast<size_t>(plan.input_room) + output_room + 16u);
    static std::atomic<uint64_t> prepared[2];
    if(const int status = allow_lds(&record_write_kernel<true>, prepared[0], lds)) { return status; }
    if(const int status = allow_lds(&record_write_kernel<false>, prepared[1], lds)) { return status; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if(plan.filtering) { hipLaunchKernelGGL(record_write_kernel<true>, dim3(blocks), dim3(TILE_THREADS), lds, s, call); }
    else { hipLaunchKernelGGL(record_write_kernel<false>, dim3(blocks), dim3(TILE_THREADS), lds, s, call); }
    return static_cast<int>(hipGetLastError());
}

int record_sizes_by_sink(const FastqTables* on_device, const FastqPlan& plan, int32_t slot, uint32_t n, int64_t capacity, const uint8_t* base_planes, const uint8_t* quality_planes,
                         const uint8_t* keep, const uint32_t* keep_block, const RecordRound& round, int32_t n_out, int32_t n_sink, void* stream) {
    if(n_out < 1 || n_out > RECORD_MAX_OUT || n_sink < 1 || n_sink > RECORD_MAX_SINK) { return static_cast<int>(hipErrorInvalidValue); }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t blocks = n > 0 ? fastq_groups(plan, n) : 0u;
    if(blocks > 0) {
        RecordArgs call = arguments_of(on_device, slot, n, capacity, base_planes, quality_planes, keep, keep_block, round, 0);
        call.n_out = n_out;
        call.n_sink = n_sink;
        const size_t lds = static_cast<size_t>(TILE_WAVES) * static_cast<size_t>(plan.input_room) + static_cast<size_t>(n_sink) * sizeof(uint32_t);
        static std::atomic<uint64_t> prepared[2];
        if(const int status = allow_lds(&record_size_sinks_kernel<true>, prepared[0], lds)) { return status; }
        if(const int status = allow_lds(&record_size_sinks_kernel<false>, prepared[1], lds)) { return status; }
        if(plan.filtering) { hipLaunchKernelGGL(record_size_sinks_kernel<true>, dim3(blocks), dim3(TILE_THREADS), lds, s, call); }
        else { hipLaunchKernelGGL(record_size_sinks_kernel<false>, dim3(blocks), dim3(TILE_THREADS), lds, s, call); }
    }
    hipLaunchKernelGGL(record_scan_sinks_kernel, dim3(static_cast<unsigned>(n_sink)), dim3(256), 0, s, round.sink_block, blocks, n_sink, round.sink_base);
    hipLaunchKernelGGL(record_base_kernel, dim3(1), dim3(1), 0, s, round.sink_base, n_sink);
    return static_cast<int>(hipGetLastError());
}

int record_write_by_sink(const FastqTables* on_device, const FastqPlan& plan, int32_t slot, uint32_t n, int64_t capacity, const uint8_t* base_planes, const uint8_t* quality_planes,
                         const uint8_t* keep, const uint32_t* keep_block, const RecordRound& round, int32_t n_out, int32_t n_sink, void* stream) {
    if(n == 0) { return 0; }
    if(n_out < 1 || n_out > RECORD_MAX_OUT || n_sink < 1 || n_sink > RECORD_MAX_SINK) { return static_cast<int>(hipErrorInvalidValue); }
    const uint32_t blocks = fastq_groups(plan, n);
    const size_t pieces = static_cast<size_t>(TILE_WAVES) * static_cast<size_t>(plan.reads_per_wave) * static_cast<size_t>(n_out);
    // the pieces' table takes its room from the waves' output
    const int32_t spare = (152 * 1024 - static_cast<int32_t>(pieces * 8)) / TILE_WAVES - plan.input_room - 16;
    const uint32_t output_room = static_cast<uint32_t>(std::max(std::min(spare, 2 * plan.input_room), 0) & ~15);
    RecordArgs call = arguments_of(on_device, slot, n, capacity, base_planes, quality_planes, keep, keep_block, round, output_room);
    call.n_out = n_out;
    call.n_sink = n_sink;
    const size_t lds = static_cast<size_t>(TILE_WAVES) * (static_cast<size_t>(plan.input_room) + output_room + 16u) + pieces * 8;
    static std::atomic<uint64_t> prepared[2];
    if(const int status = allow_lds(&record_write_sinks_kernel<true>, prepared[0], lds)) { return status; }
    if(const int status = allow_lds(&record_write_sinks_kernel<false>, prepared[1], lds)) { return status; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if(plan.filtering) { hipLaunchKernelGGL(record_write_sinks_kernel<true>, dim3(blocks), dim3(TILE_THREADS), lds, s, call); }
    else { hipLaunchKernelGGL(record_write_sinks_kernel<false>, dim3(blocks), dim3(TILE_THREADS), lds, s, call); }
    return static_cast<int>(hipGetLastError());
}

int record_quality_control(const FastqTables* on_device, const FastqPlan& plan, int32_t slot, uint32_t n, int64_t capacity, const uint8_t* base_planes, const uint8_t* quality_planes,
                           const uint8_t* keep, const uint32_t* keep_block, const RecordRound& round, uint32_t* overflow, void* stream) {
    if(n == 0) { return 0; }
    if(round.quality_control == nullptr || overflow == nullptr || round.cycles < 0) { return static_cast<int>(hipErrorInvalidValue); }
    const uint32_t blocks = fastq_groups(plan, n);
    RecordArgs call = arguments_of(on_device, slot, n, capacity, base_planes, quality_planes, keep, keep_block, round, 0);
    call.overflow = overflow;
    const size_t lds = static_cast<size_t>(TILE_WAVES) * static_cast<size_t>(plan.input_room);
    static std::atomic<uint64_t> prepared[2];
    if(const int status = allow_lds(&record_quality_control_kernel<true>, prepared[0], lds)) { return status; }
    if(const int status = allow_lds(&record_quality_control_kernel<false>, prepared[1], lds)) { return status; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if(plan.filtering) { hipLaunchKernelGGL(record_quality_control_kernel<true>, dim3(blocks), dim3(TILE_THREADS), lds, s, call); }
    else { hipLaunchKernelGGL(record_quality_control_kernel<false>, dim3(blocks), dim3(TILE_THREADS), lds, s, call); }
    return static_cast<int>(hipGetLastError());
}

}  // namespace phq
