// A gzip member inflated in one go, for gzip_members.hpp: the whole compressed member and the exact size of its text are known
// (BGZF blocks, concatenated members), so the decoder needs no stream state, no window and no output checks beyond one bound —
// a 64 bit bit buffer refilled by unaligned loads, a 11 bit first level table for the literal / length code (8 bit for the
// distances) with second level tables behind it, matches copied eight bytes at a time.  zlib's inflate() is ~0.3 GB/s of FASTQ
// text a thread and was ALL of a BGZF run's time (19.5 Mreads/s on 16 cores); RFC 1951 is the specification followed here.
//
// Safety: whatever this decoder produces is held to the member's own trailer — CRC-32 (carry-less multiplication where the CPU
// has it, zlib's crc32 otherwise) and ISIZE — and it refuses anything it is not sure of (an over- or under-subscribed code, a
// distance before the start of the text, text past the stated size): `false` sends the member through zlib's inflate(), which
// then gives the text or the error.  It can be slower than zlib on odd input; it cannot hand on different text.
#pragma once

#include <cstdint>
#include <cstring>

#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define PHQ_CRC_FOLDED 1
#endif
#include <zlib.h>

namespace phq {
namespace fast_inflate {

constexpr int LITLEN_BITS = 11;
constexpr int DIST_BITS = 8;
constexpr int LITLEN_ENTRIES = (1 << LITLEN_BITS) + 1024;       // second level tables: at most 2^(15 - 11) entries for each of fewer than 64 first level prefixes
constexpr int DIST_ENTRIES = (1 << DIST_BITS) + 512;
// entry: bits 0..4 bits taken at this level, 5..7 kind, 8..11 extra bits, 12..31 payload (a base, a table index); a literal entry
// holds ONE, TWO or THREE literals — kind + 1 of them, in bits 8..15, 16..23, 24..31 — where the codes of the next literals fit
// what is left of the first level's 11 bits: the text of a FASTQ file is half literals of two or three bits (the bases), and a
// lookup a literal is a chain of load -> shift -> mask -> load, seven cycles a byte
enum Kind : uint32_t { LITERAL = 0, LITERAL2 = 1, LITERAL3 = 2, LENGTH = 3, END = 4, SUBTABLE = 5, DISTANCE = 6, INVALID = 7 };
constexpr uint32_t entry_of(uint32_t payload, uint32_t kind, uint32_t extra, uint32_t length) { return (payload << 12) | (extra << 8) | (kind << 5) | length; }
constexpr uint32_t literal_of(uint32_t byte, uint32_t length) { return (byte << 8) | (LITERAL << 5) | length; }
constexpr uint32_t kind_of(uint32_t entry) { return (entry >> 5) & 7u; }
constexpr uint32_t extra_of(uint32_t entry) { return (entry >> 8) & 15u; }

struct Tables {
    uint32_t litlen[LITLEN_ENTRIES];
    uint32_t dist[DIST_ENTRIES];
};

constexpr uint16_t LENGTH_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr uint8_t LENGTH_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
constexpr uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
constexpr uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t reverse_bits(uint32_t code, int length) {
    uint32_t r = 0;
    for(int k = 0; k < length; ++k) { r = (r << 1) | ((code >> k) & 1u); }
    return r;
}

// canonical Huffman code of `count` symbols with the given lengths (RFC 1951 3.2.2) -> a two level table indexed by the next
// bits of the stream, least significant bit first.  false: not a complete prefix code (one code of length 1 alone is allowed
// where `single_allowed`: a distance code of a single symbol), or the second level does not fit
template <typename Describe>
inline bool build_table(const uint8_t* lengths, int count, int table_bits, uint32_t* table, int capacity, bool single_allowed, Describe&& describe) {
    int per_length[16] = {0};
    for(int s = 0; s < count; ++s) { ++per_length[lengths[s]]; }
    per_length[0] = 0;
    int used = 0;
    for(int l = 1; l <= 15; ++l) { used += per_length[l]; }
    const uint32_t invalid = entry_of(0, INVALID, 0, 1);
    for(int k = 0; k < (1 << table_bits); ++k) { table[k] = invalid; }
    if(used == 0) { return single_allowed; }                       // no code at all: a block of literals only has no distance code
    // Kraft: sum 2^(15 - l) == 2^15
    uint32_t space = 0;
    for(int l = 1; l <= 15; ++l) { space += static_cast<uint32_t>(per_length[l]) << (15 - l); }
    if(space != (1u << 15)) {
        if(!(single_allowed && used == 1 && per_length[1] == 1)) { return false; }
    }
    uint32_t next_code[16];
    uint32_t code = 0;
    for(int l = 1; l <= 15; ++l) { code = (code + static_cast<uint32_t>(per_length[l - 1])) << 1; next_code[l] = code; }
    // second level: the longest code behind every first level prefix
    uint8_t longest[1 << LITLEN_BITS];
    std::memset(longest, 0, static_cast<size_t>(1) << table_bits);
    uint32_t assigned[320];
    {
        uint32_t running[16];
        for(int l = 1; l <= 15; ++l) { running[l] = next_code[l]; }
        for(int s = 0; s < count; ++s) {
            const int l = lengths[s];
            if(l == 0) { continue; }
            const uint32_t reversed = reverse_bits(running[l]++, l);
            assigned[s] = reversed;
            if(l > table_bits) {
                uint8_t& deepest = longest[reversed & ((1u << table_bits) - 1u)];
                if(l > deepest) { deepest = static_cast<uint8_t>(l); }
            }
        }
    }
    int next_free = 1 << table_bits;
    for(int prefix = 0; prefix < (1 << table_bits); ++prefix) {
        if(longest[prefix] == 0) { continue; }
        const int bits = longest[prefix] - table_bits;
        if(next_free + (1 << bits) > capacity) { return false; }
        table[prefix] = entry_of(static_cast<uint32_t>(next_free), SUBTABLE, static_cast<uint32_t>(bits), static_cast<uint32_t>(table_bits));
        for(int k = 0; k < (1 << bits); ++k) { table[next_free + k] = invalid; }
        next_free += 1 << bits;
    }
    for(int s = 0; s < count; ++s) {
        const int l = lengths[s];
        if(l == 0) { continue; }
        const uint32_t reversed = assigned[s];
        if(l <= table_bits) {
            const uint32_t entry = describe(s, static_cast<uint32_t>(l));
            for(uint32_t at = reversed; at < (1u << table_bits); at += 1u << l) { table[at] = entry; }
        } else {
            const uint32_t first = table[reversed & ((1u << table_bits) - 1u)];
            const uint32_t base = first >> 12, bits = extra_of(first);
            const uint32_t entry = describe(s, static_cast<uint32_t>(l - table_bits));
            for(uint32_t at = reversed >> table_bits; at < (1u << bits); at += 1u << (l - table_bits)) { table[base + at] = entry; }
        }
    }
    return true;
}

inline bool build_litlen(const uint8_t* lengths, int count, Tables& t) {
    if(!build_table(lengths, count, LITLEN_BITS, t.litlen, LITLEN_ENTRIES, false, [](int symbol, uint32_t length) {
        if(symbol < 256) { return literal_of(static_cast<uint32_t>(symbol), length); }
        if(symbol == 256) { return entry_of(0, END, 0, length); }
        if(symbol > 285) { return entry_of(0, INVALID, 0, length); }
        return entry_of(LENGTH_BASE[symbol - 257], LENGTH, LENGTH_EXTRA[symbol - 257], length);
    })) { return false; }
    // two and three literals an entry: the single literal entries are kept aside while the first level is rewritten
    uint32_t single[1 << LITLEN_BITS];
    std::memcpy(single, t.litlen, sizeof(single));
    for(uint32_t index = 0; index < (1u << LITLEN_BITS); ++index) {
        const uint32_t first = single[index];
        if(kind_of(first) != LITERAL) { continue; }
        const uint32_t taken = first & 31u;
        const uint32_t second = single[index >> taken];                       // the bits behind the first code, zeros above what the index holds
        if(kind_of(second) != LITERAL || taken + (second & 31u) > static_cast<uint32_t>(LITLEN_BITS)) { continue; }
        const uint32_t both = taken + (second & 31u);
        const uint32_t third = single[index >> both];
        if(kind_of(third) == LITERAL && both + (third & 31u) <= static_cast<uint32_t>(LITLEN_BITS)) {
            t.litlen[index] = (((third >> 8) & 0xffu) << 24) | (((second >> 8) & 0xffu) << 16) | (first & 0xff00u) | (LITERAL3 << 5) | (both + (third & 31u));
        } else {
            t.litlen[index] = (((second >> 8) & 0xffu) << 16) | (first & 0xff00u) | (LITERAL2 << 5) | both;
        }
    }
    return true;
}
inline bool build_dist(const uint8_t* lengths, int count, Tables& t) {
    return build_table(lengths, count, DIST_BITS, t.dist, DIST_ENTRIES, true, [](int symbol, uint32_t length) {
        if(symbol > 29) { return entry_of(0, INVALID, 0, length); }
        return entry_of(DIST_BASE[symbol], DISTANCE, DIST_EXTRA[symbol], length);
    });
}

// raw deflate (RFC 1951) of exactly `out_size` bytes; *consumed = bytes of `in` the stream took.  false: anything unexpected.
inline bool inflate_raw(const uint8_t* in, size_t in_size, uint8_t* const out_begin, size_t out_size, size_t* consumed) {
    const uint8_t* in_next = in;
    const uint8_t* const in_end = in + in_size;
    uint8_t* out = out_begin;
    uint8_t* const out_end = out_begin + out_size;
    uint64_t bits = 0;
    int held = 0;                                  // valid bits in `bits`
    // at least 56 bits where the input has them (bytes past the end read as zeros: a code that runs there is caught below)
    auto refill = [&]() {
        if(in_end - in_next >= 8) {
            uint64_t word;
            std::memcpy(&word, in_next, 8);
            bits |= word << held;
            in_next += (63 - held) >> 3;
            held |= 56;
        } else {
            while(held <= 56 && in_next < in_end) { bits |= static_cast<uint64_t>(*in_next++) << held; held += 8; }
        }
    };
    auto take = [&](int n) { const uint32_t v = static_cast<uint32_t>(bits & ((uint64_t(1) << n) - 1u)); bits >>= n; held -= n; return v; };
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    static Tables fixed;
    static const bool fixed_ready = []() {
        uint8_t lengths[288];
        for(int s = 0; s < 144; ++s) { lengths[s] = 8; }
        for(int s = 144; s < 256; ++s) { lengths[s] = 9; }
        for(int s = 256; s < 280; ++s) { lengths[s] = 7; }
        for(int s = 280; s < 288; ++s) { lengths[s] = 8; }
        // (the fixed distance code has 32 codes of 5 bits, two of them unused: complete with them)
        uint8_t all[32];
        for(int s = 0; s < 32; ++s) { all[s] = 5; }
        return build_litlen(lengths, 288, fixed) && build_table(all, 32, DIST_BITS, fixed.dist, DIST_ENTRIES, true, [](int symbol, uint32_t length) {
            return symbol > 29 ? entry_of(0, INVALID, 0, length) : entry_of(DIST_BASE[symbol], DISTANCE, DIST_EXTRA[symbol], length); });
    }();
    if(!fixed_ready) { return false; }
    Tables dynamic;
    bool last = false;
    while(!last) {
        refill();
        if(held < 3) { return false; }
        last = take(1) != 0;
        const uint32_t type = take(2);
        if(type == 0) {
            // stored: to the byte boundary, LEN, NLEN, the bytes.  Whole bytes still in the bit buffer are given back first.
            take(held & 7);
            in_next -= held >> 3;
            bits = 0; held = 0;
            if(in_end - in_next < 4) { return false; }
            const uint32_t length = static_cast<uint32_t>(in_next[0]) | (static_cast<uint32_t>(in_next[1]) << 8);
            const uint32_t check = static_cast<uint32_t>(in_next[2]) | (static_cast<uint32_t>(in_next[3]) << 8);
            in_next += 4;
            if((length ^ check) != 0xffffu || static_cast<size_t>(in_end - in_next) < length || static_cast<size_t>(out_end - out) < length) { return false; }
            std::memcpy(out, in_next, length);
            in_next += length;
            out += length;
            continue;
        }
        const Tables* tables = &fixed;
        if(type == 2) {
            if(held < 14) { return false; }
            const int n_litlen = static_cast<int>(take(5)) + 257, n_dist = static_cast<int>(take(5)) + 1, n_pre = static_cast<int>(take(4)) + 4;
            if(n_litlen > 286 || n_dist > 30) { return false; }
            uint8_t pre_lengths[19] = {0};
            refill();
            for(int k = 0; k < n_pre; ++k) {
                if(held < 3) { refill(); if(held < 3) { return false; } }
                pre_lengths[ORDER[k]] = static_cast<uint8_t>(take(3));
            }
            uint32_t pre_table[128 + 8];
            if(!build_table(pre_lengths, 19, 7, pre_table, 128 + 8, false, [](int symbol, uint32_t length) { return entry_of(static_cast<uint32_t>(symbol), LITERAL, 0, length); })) { return false; }
            uint8_t lengths[286 + 30 + 138];
            int at = 0;
            while(at < n_litlen + n_dist) {
                refill();
                const uint32_t entry = pre_table[bits & 127u];
                if(kind_of(entry) != LITERAL || static_cast<int>(entry & 31u) > held) { return false; }
                take(static_cast<int>(entry & 31u));
                const uint32_t symbol = entry >> 12;
                if(symbol < 16) { lengths[at++] = static_cast<uint8_t>(symbol); continue; }
                uint32_t repeat, value = 0;
                if(symbol == 16) { if(at == 0) { return false; } value = lengths[at - 1]; repeat = 3 + take(2); }
                else if(symbol == 17) { repeat = 3 + take(3); }
                else { repeat = 11 + take(7); }
                if(held < 0 || at + static_cast<int>(repeat) > n_litlen + n_dist) { return false; }
                for(uint32_t k = 0; k < repeat; ++k) { lengths[at++] = static_cast<uint8_t>(value); }
            }
            if(lengths[256] == 0) { return false; }                 // no end of block code
            if(!build_litlen(lengths, n_litlen, dynamic) || !build_dist(lengths + n_litlen, n_dist, dynamic)) { return false; }
            tables = &dynamic;
        } else if(type != 1) { return false; }

        // the block's symbols
        const uint32_t* const litlen = tables->litlen;
        const uint32_t* const dist = tables->dist;
        constexpr uint32_t LITLEN_MASK = (1u << LITLEN_BITS) - 1u, DIST_MASK = (1u << DIST_BITS) - 1u;
        bool open = true;
        while(open) {
            // With sixteen bytes of input and a longest match and three literals of room ahead nothing below needs a check: the
            // refill gives 56 bits, three literals take 45 at most, a length its 20 behind two literals, a distance 28 behind its
            // own refill.  Near either end the careful loop below takes over, symbol by symbol.
            while(in_end - in_next >= 16 && out_end - out >= 258 + 9 + 4 + 8) {
                {
                    uint64_t word;
                    std::memcpy(&word, in_next, 8);
                    bits |= word << held;
                    in_next += (63 - held) >> 3;
                    held |= 56;
                }
                uint32_t entry = litlen[bits & LITLEN_MASK];
                if(kind_of(entry) <= LITERAL3) {
                    // one to three literals an entry, up to three entries a refill (3 x 11 bits): four bytes are stored, kind + 1 of
                    // them count (the loop's guard keeps room for what is stored and not counted)
                    uint32_t bytes = entry >> 8;
                    std::memcpy(out, &bytes, 4);
                    out += kind_of(entry) + 1u;
                    bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
                    entry = litlen[bits & LITLEN_MASK];
                    if(kind_of(entry) <= LITERAL3) {
                        bytes = entry >> 8;
                        std::memcpy(out, &bytes, 4);
                        out += kind_of(entry) + 1u;
                        bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
                        entry = litlen[bits & LITLEN_MASK];
                        if(kind_of(entry) <= LITERAL3) {
                            bytes = entry >> 8;
                            std::memcpy(out, &bytes, 4);
                            out += kind_of(entry) + 1u;
                            bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
                            continue;
                        }
                    }
                }
                if(kind_of(entry) == SUBTABLE) {
                    entry = litlen[(entry >> 12) + ((bits >> LITLEN_BITS) & ((1u << extra_of(entry)) - 1u))];
                    bits >>= LITLEN_BITS; held -= LITLEN_BITS;
                }
                uint32_t kind = kind_of(entry);
                bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
                if(kind == LITERAL) { *out++ = static_cast<uint8_t>(entry >> 8); continue; }        // (from a second level table: a single one)
                if(kind == END) { open = false; break; }
                if(kind != LENGTH) { return false; }
                const uint32_t extra = extra_of(entry);
                const uint32_t length = (entry >> 12) + static_cast<uint32_t>(bits & ((1u << extra) - 1u));
                bits >>= extra; held -= static_cast<int>(extra);
                {
                    uint64_t word;
                    std::memcpy(&word, in_next, 8);
                    bits |= word << held;
                    in_next += (63 - held) >> 3;
                    held |= 56;
                }
                entry = dist[bits & DIST_MASK];
                if(kind_of(entry) == SUBTABLE) {
                    entry = dist[(entry >> 12) + ((bits >> DIST_BITS) & ((1u << extra_of(entry)) - 1u))];
                    bits >>= DIST_BITS; held -= DIST_BITS;
                }
                if(kind_of(entry) != DISTANCE) { return false; }
                bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
                const uint32_t more = extra_of(entry);
                const uint32_t distance = (entry >> 12) + static_cast<uint32_t>(bits & ((1u << more) - 1u));
                bits >>= more; held -= static_cast<int>(more);
                if(distance > static_cast<size_t>(out - out_begin)) { return false; }
                const uint8_t* from = out - distance;
                uint8_t* const stop = out + length;
                if(distance >= 8) {
                    // eight bytes at a time (the source is at least eight behind: no overlap inside a step; up to seven bytes past the
                    // match are written and written again by what follows — never past the member's own text: the loop's guard)
                    do { uint64_t word; std::memcpy(&word, from, 8); std::memcpy(out, &word, 8); from += 8; out += 8; } while(out < stop);
                } else if(distance == 1) {
                    std::memset(out, *from, length);
                } else {
                    // a period of 2 to 7 bytes: the first eight bytes byte by byte, then whole periods' multiples eight at a time
                    for(uint32_t k = 0; k < 8; ++k) { out[k] = from[k]; }
                    if(length > 8) {
                        const uint32_t stride = distance * ((8u + distance - 1u) / distance);      // the smallest multiple of the period of at least eight bytes (8 .. 14): a step never reads what it writes, and starts inside the period's first copy
                        const uint8_t* again = out + 8 - stride;
                        uint8_t* to = out + 8;
                        do { uint64_t word; std::memcpy(&word, again, 8); std::memcpy(to, &word, 8); again += 8; to += 8; } while(to < stop);
                    }
                }
                out = stop;
            }
            if(!open) { break; }
            // the careful loop: one symbol, every bound checked
            refill();
            uint32_t entry = litlen[bits & LITLEN_MASK];
            if(kind_of(entry) == SUBTABLE) {
                entry = litlen[(entry >> 12) + ((bits >> LITLEN_BITS) & ((1u << extra_of(entry)) - 1u))];
                bits >>= LITLEN_BITS; held -= LITLEN_BITS;
            }
            uint32_t kind = kind_of(entry);
            bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
            if(held < 0) { return false; }
            if(kind <= LITERAL3) {
                if(static_cast<size_t>(out_end - out) < kind + 1u) { return false; }
                for(uint32_t k = 0; k <= kind; ++k) { *out++ = static_cast<uint8_t>(entry >> (8 + 8 * k)); }
                continue;
            }
            if(kind == END) { break; }
            if(kind != LENGTH) { return false; }
            const uint32_t length = (entry >> 12) + take(static_cast<int>(extra_of(entry)));
            if(held < 32) { refill(); }
            entry = dist[bits & DIST_MASK];
            if(kind_of(entry) == SUBTABLE) {
                entry = dist[(entry >> 12) + ((bits >> DIST_BITS) & ((1u << extra_of(entry)) - 1u))];
                bits >>= DIST_BITS; held -= DIST_BITS;
            }
            kind = kind_of(entry);
            bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
            if(kind != DISTANCE) { return false; }
            const uint32_t distance = (entry >> 12) + take(static_cast<int>(extra_of(entry)));
            if(held < 0 || distance > static_cast<size_t>(out - out_begin) || length > static_cast<size_t>(out_end - out)) { return false; }
            const uint8_t* from = out - distance;
            for(uint32_t k = 0; k < length; ++k) { out[k] = from[k]; }
            out += length;
        }
    }
    if(out != out_end) { return false; }
    // whole bytes the bit buffer still holds were not the stream's
    in_next -= held >> 3;
    *consumed = static_cast<size_t>(in_next - in);
    return true;
}

#ifdef PHQ_CRC_FOLDED
// CRC-32 (the gzip polynomial) by carry-less multiplication: four 128 bit lanes folded by 64 bytes a step (Gopal et al., "Fast CRC
// computation for generic polynomials using PCLMULQDQ", Intel 2009: its constants for this polynomial); `crc` in and out as
// zlib's crc32() takes and gives it.  At least 64 bytes, a multiple of 16.
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_folded(uint32_t crc, const uint8_t* buffer, size_t bytes) {
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buffer + 0x00));
    x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buffer + 0x10));
    x3 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buffer + 0x20));
    x4 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buffer + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128(static_cast<int>(~crc)));
    x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(k1k2));
    buffer += 64; bytes -= 64;
    while(bytes >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00); x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11); x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buffer + 0x00)); y6 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buffer + 0x10));
        y7 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buffer + 0x20)); y8 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buffer + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buffer += 64; bytes -= 64;
    }
    x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(k3k4));
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while(bytes >= 16) {
        x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buffer));
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buffer += 16; bytes -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64(reinterpret_cast<const __m128i*>(k5k0));
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(poly));
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return ~static_cast<uint32_t>(_mm_extract_epi32(x1, 1));
}
#endif

inline uint32_t crc32_of(const uint8_t* buffer, size_t bytes) {
    uint32_t crc = 0;
#ifdef PHQ_CRC_FOLDED
    static const bool folded = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if(folded && bytes >= 64) {
        const size_t body = bytes & ~static_cast<size_t>(15);
        crc = crc32_folded(crc, buffer, body);
        buffer += body; bytes -= body;
    }
#endif
    while(bytes > 0) {
        const uInt step = static_cast<uInt>(bytes > (size_t(1) << 30) ? (size_t(1) << 30) : bytes);
        crc = static_cast<uint32_t>(crc32(crc, buffer, step));
        buffer += step; bytes -= step;
    }
    return crc;
}

// one whole gzip member (RFC 1952): header, deflate stream of exactly `text` bytes, CRC-32 and ISIZE; true: `out` holds the text
inline bool inflate_member(const uint8_t* in, size_t bytes, uint8_t* out, size_t text) {
    if(bytes < 18 || in[0] != 0x1f || in[1] != 0x8b || in[2] != 8 || (in[3] & 0xe0) != 0) { return false; }
    const uint8_t flags = in[3];
    size_t at = 10;
    if(flags & 4) {
        if(at + 2 > bytes) { return false; }
        at += 2 + (static_cast<size_t>(in[at]) | (static_cast<size_t>(in[at + 1]) << 8));
    }
    if(flags & 8) { while(at < bytes && in[at] != 0) { ++at; } ++at; }
    if(flags & 16) { while(at < bytes && in[at] != 0) { ++at; } ++at; }
    if(flags & 2) { at += 2; }
    if(at + 8 > bytes) { return false; }
    size_t consumed = 0;
    if(!inflate_raw(in + at, bytes - at - 8, out, text, &consumed)) { return false; }
    if(at + consumed + 8 != bytes) { return false; }
    const uint8_t* const trailer = in + at + consumed;
    const uint32_t stated_crc = static_cast<uint32_t>(trailer[0]) | (static_cast<uint32_t>(trailer[1]) << 8) | (static_cast<uint32_t>(trailer[2]) << 16) | (static_cast<uint32_t>(trailer[3]) << 24);
    const uint32_t stated_size = static_cast<uint32_t>(trailer[4]) | (static_cast<uint32_t>(trailer[5]) << 8) | (static_cast<uint32_t>(trailer[6]) << 16) | (static_cast<uint32_t>(trailer[7]) << 24);
    return stated_size == static_cast<uint32_t>(text) && stated_crc == crc32_of(out, text);
}

}  // namespace fast_inflate
}  // namespace phq
