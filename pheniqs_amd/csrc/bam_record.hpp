// Unaligned BAM output (the SAM specification, sections 4.1 and 4.2) without htslib: the reference writes its reads through htslib's
// bam1_t whatever the format (hts.h:169-239, hts.cpp:26-44: tid -1, pos -1, bin 0, mapq 0, no CIGAR, mate -1 / -1, tlen 0;
// auxiliary.cpp:320-354: FI / TC as type 'i' of four bytes, every string 'Z', XB / XM / XC as type 'f'), and `output: sample.bam` is
// what most Pheniqs runs state.  A record here is made from the SAM text line the host writer has just formatted — the same fields
// in the same order, so what holds the text holds the record — with the three posterior tags taken from their float values, not
// from their six printed digits; the records go into BGZF blocks (fast_deflate.hpp) on the writer's threads and the file ends
// with BGZF's empty block.  CRAM stays out (reference based compression: htslib).
#pragma once

#include <cstdint>
#include <cstring>
#include <string>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "error.hpp"
#include "text_format.hpp"

namespace phq {

// "BAM\1", the header text, no reference sequences
inline std::string bam_header(const std::string& text) {
    std::string out("BAM\1", 4);
    const uint32_t l_text = static_cast<uint32_t>(text.size()), n_ref = 0;
    out.append(reinterpret_cast<const char*>(&l_text), 4);
    out.append(text);
    out.append(reinterpret_cast<const char*>(&n_ref), 4);
    return out;
}

// one SAM line of an unaligned read (no newline): QNAME FLAG * 0 0 * * 0 0 SEQ QUAL [TAG:TYPE:VALUE ...]; posterior[0..2]: the values
// of XB, XM, XC where the line carries them
inline void append_bam_record(ByteBuffer& out, const char* line, size_t size, const float* posterior) {
    // the specification's table "=ACMGRSVTWYHKDBN" (either case); anything else is N
    static const struct Codes {
        uint8_t of[256];
        Codes() {
            std::memset(of, 15, sizeof(of));
            const char* const letters = "=ACMGRSVTWYHKDBN";
            for(int k = 0; k < 16; ++k) {
                of[static_cast<uint8_t>(letters[k])] = static_cast<uint8_t>(k);
                if(letters[k] >= 'A' && letters[k] <= 'Z') { of[static_cast<uint8_t>(letters[k] + 32)] = static_cast<uint8_t>(k); }
            }
        }
    } codes;
    const uint8_t* const CODE = codes.of;
    // where the line's tabs are, in one pass (a line this writer makes has ten in front of the bases and one in front of every tag)
    constexpr int MAX_TABS = 64;
    uint32_t tab_at[MAX_TABS + 1];
    int tabs = 0;
    {
        size_t i = 0;
#if defined(__SSE2__)
        const __m128i wanted = _mm_set1_epi8('\t');
        for(; i + 16 <= size; i += 16) {
            uint32_t mask = static_cast<uint32_t>(_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(line + i)), wanted)));
            while(mask) {
                if(tabs == MAX_TABS) { throw InternalError("a SAM line of more tags than this writer makes"); }
                tab_at[tabs++] = static_cast<uint32_t>(i) + static_cast<uint32_t>(__builtin_ctz(mask));
                mask &= mask - 1;
            }
        }
#endif
        for(; i < size; ++i) {
            if(line[i] == '\t') {
                if(tabs == MAX_TABS) { throw InternalError("a SAM line of more tags than this writer makes"); }
                tab_at[tabs++] = static_cast<uint32_t>(i);
            }
        }
    }
    if(tabs < 10) { throw InternalError("a SAM line of fewer than eleven fields"); }
    tab_at[tabs] = static_cast<uint32_t>(size);
    const char* field[11];
    size_t length[11];
    for(int k = 0; k < 11; ++k) {
        const size_t first = k == 0 ? 0 : tab_at[k - 1] + 1u;
        field[k] = line + first;
        length[k] = tab_at[k] - first;
    }
    const char* const end = line + size;
    int next_tab = 10;                          // the tab in front of the first tag, if there is one
    const char* at = tabs > 10 ? line + tab_at[10] + 1 : end;
    uint32_t flag = 0;
    for(size_t i = 0; i < length[1]; ++i) { flag = flag * 10u + static_cast<uint32_t>(field[1][i] - '0'); }
    const bool no_sequence = length[9] == 1 && field[9][0] == '*';
    const size_t l_seq = no_sequence ? 0 : length[9];
    const bool no_quality = length[10] == 1 && field[10][0] == '*' && l_seq != 1;
    if(length[0] > 254) { throw OverflowError("qname must not exceed 254 characters"); }            // hts.h:180,238
    // the fixed part, the name, the bases two a byte, the qualities, then the tags: at most the line's own size + 64
    char* const begin = out.reserve(size + 64 + l_seq);
    char* to = begin + 4;
    auto put32 = [&](int32_t v) { std::memcpy(to, &v, 4); to += 4; };
    auto put16 = [&](uint16_t v) { std::memcpy(to, &v, 2); to += 2; };
    put32(-1);                                                  // refID
    put32(-1);                                                  // pos
    *to++ = static_cast<char>(length[0] + 1);                   // l_read_name
    *to++ = 0;                                                  // mapq
    put16(0);                                                   // bin (hts.cpp:32: the reference leaves 0)
    put16(0);                                                   // n_cigar_op
    put16(static_cast<uint16_t>(flag));
    put32(static_cast<int32_t>(l_seq));
    put32(-1);                                                  // next refID
    put32(-1);                                                  // next pos
    put32(0);                                                   // tlen
    to = put_bytes(to, field[0], length[0]);
    *to++ = 0;
    for(size_t i = 0; i + 1 < l_seq; i += 2) {
        *to++ = static_cast<char>((CODE[static_cast<uint8_t>(field[9][i])] << 4) | CODE[static_cast<uint8_t>(field[9][i + 1])]);
    }
    if(l_seq & 1) { *to++ = static_cast<char>(CODE[static_cast<uint8_t>(field[9][l_seq - 1])] << 4); }
    if(no_quality) { std::memset(to, 0xff, l_seq); to += l_seq; }
    else {
        const char* const quality = field[10];
        char* const phred = to;
        for(size_t i = 0; i < l_seq; ++i) { phred[i] = static_cast<char>(quality[i] - 33); }
        to += l_seq;
    }
    // TAG:TYPE:VALUE
    while(at < end) {
        ++next_tab;
        const char* const tab = next_tab < tabs ? line + tab_at[next_tab] : nullptr;
        const char* const stop = tab ? tab : end;
        if(stop - at < 5 || at[2] != ':' || at[4] != ':') { throw InternalError("a SAM tag this writer does not make"); }
        *to++ = at[0]; *to++ = at[1];
        const char type = at[3];
        const char* const value = at + 5;
        if(type == 'Z') {
            *to++ = 'Z';
            to = put_bytes(to, value, static_cast<size_t>(stop - value));
            *to++ = 0;
        } else if(type == 'i') {
            int32_t v = 0;
            bool negative = false;
            for(const char* p = value; p < stop; ++p) { if(*p == '-') { negative = true; } else { v = v * 10 + (*p - '0'); } }
            *to++ = 'i';
            put32(negative ? -v : v);
        } else if(type == 'f') {
            *to++ = 'f';
            if(stop - value == 9 && value[0] == '#') {
                // a line assembled on the device for a BAM sink states the float's bits (record_device.hpp RECORD_FORMAT_SAM_FOR_BAM)
                uint32_t bits = 0;
                for(int k = 1; k < 9; ++k) {
                    const char c = value[k];
                    const uint32_t digit = (c >= '0' && c <= '9') ? static_cast<uint32_t>(c - '0') : (c >= 'a' && c <= 'f') ? static_cast<uint32_t>(c - 'a' + 10) : 16u;
                    if(digit > 15u) { throw InternalError("a float tag that states no bits"); }
                    bits = (bits << 4) | digit;
                }
                std::memcpy(to, &bits, 4);
            } else {
                const int which = (at[0] == 'X' && at[1] == 'B') ? 0 : (at[0] == 'X' && at[1] == 'M') ? 1 : (at[0] == 'X' && at[1] == 'C') ? 2 : -1;
                if(which < 0 || posterior == nullptr) { throw InternalError("a float tag this writer does not make"); }
                std::memcpy(to, &posterior[which], 4);
            }
            to += 4;
        } else { throw InternalError("a SAM tag type this writer does not make"); }
        at = tab ? tab + 1 : end;
    }
    const int32_t block_size = static_cast<int32_t>(to - begin - 4);
    std::memcpy(begin, &block_size, 4);
    out.commit(to);
}

// every line of `text` (whole lines, each ended by a newline) as a BAM record behind `out`'s bytes
inline void append_bam_records(ByteBuffer& out, const char* text, size_t size) {
    const char* at = text;
    const char* const end = text + size;
    while(at < end) {
        const char* const newline = static_cast<const char*>(std::memchr(at, '\n', static_cast<size_t>(end - at)));
        const char* const stop = newline ? newline : end;
        if(stop > at) { append_bam_record(out, at, static_cast<size_t>(stop - at), nullptr); }
        at = newline ? newline + 1 : end;
    }
}

}  // namespace phq
