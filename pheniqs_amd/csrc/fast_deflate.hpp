// A gzip member deflated in one go, for the transcoder's gzip outputs (a FASTQ or SAM file per barcode, `.gz`): zlib's deflate at
// the reference's default level 5 (configuration.json:6) is 80 MB/s of FASTQ text a thread, and with the records assembled on the
// device and 16 host threads compressing them a run that writes gzip was bound by it: 8 Mreads/s against 38 for plain files.
//
// RFC 1951 leaves the compressor its choices; these are made for the text this library writes:
//   * one hash probe per position (4 bytes, 2^15 entries, the member's own 32 KiB behind it) and greedy matching — no chains, no
//     lazy evaluation;
//   * a match is taken only where it is CHEAPER than its bytes as literals, priced with the code lengths of the block before:
//     the bases of a read cost two bits each under their Huffman code, so the chance matches of 4 to 8 bases that zlib takes
//     (17+ bits each) lose against their own literals — skipping them is faster AND smaller; read names (a match into the name
//     before) and runs of quality values are where the matches pay;
//   * dynamic Huffman codes per block of up to 64 Ki symbols (lengths limited to 15 / 7 bits by halving the counts until they
//     fit), a stored block where coding does not pay.
// The result is a plain gzip member (header, deflate stream, CRC-32, ISIZE): zlib, the reference's htslib and this library's own
// decoders read it.  check/fast_deflate_check.cpp holds it to zlib's inflate and to fast_inflate.hpp on text, noise, runs and
// empty input under ASan + UBSan; PHQ_GZIP_ZLIB=1 (or a level of 6 and more in the URL) keeps zlib's deflate.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include <zlib.h>

#include "fast_inflate.hpp"

namespace phq {
namespace fast_deflate {

constexpr int HASH_BITS = 15;
constexpr size_t WINDOW = size_t(1) << 15;
constexpr size_t BLOCK_SYMBOLS = size_t(1) << 16;
constexpr int MAX_MATCH = 258;
constexpr int MIN_MATCH = 4;

struct BitWriter {
    uint8_t* next;
    uint8_t* end;
    uint64_t bits = 0;                                   // fewer than 8 of them between two calls: whole bytes leave at once
    int held = 0;
    bool overflow = false;
    void put(uint64_t value, int n) {                    // n <= 48
        bits |= value << held;
        held += n;
        if(end - next < 8) { overflow = true; bits = 0; held &= 7; return; }
        std::memcpy(next, &bits, 8);                     // (the bytes past the whole ones are written again by the next call)
        next += held >> 3;
        bits >>= held & ~7;
        held &= 7;
    }
    void align() {
        if(held > 0) {
            if(next >= end) { overflow = true; held = 0; bits = 0; return; }
            *next++ = static_cast<uint8_t>(bits);
        }
        held = 0; bits = 0;
    }
};

// code lengths of at most `limit` bits for the symbols with a count: Huffman's algorithm over the sorted counts (two queues), the
// counts halved — never to zero — and the tree built again while its deepest leaf is past the limit
inline void code_lengths(const uint32_t* count, int n, int limit, uint8_t* length) {
    struct Node { uint64_t weight; int left, right; };
    std::vector<uint32_t> scaled(count, count + n);
    std::vector<int> order;
    for(;;) {
        order.clear();
        for(int s = 0; s < n; ++s) { length[s] = 0; if(scaled[static_cast<size_t>(s)]) { order.push_back(s); } }
        if(order.empty()) { return; }
        if(order.size() == 1) { length[order[0]] = 1; return; }
        std::sort(order.begin(), order.end(), [&](int a, int b) { return scaled[static_cast<size_t>(a)] != scaled[static_cast<size_t>(b)] ? scaled[static_cast<size_t>(a)] < scaled[static_cast<size_t>(b)] : a < b; });
        const int leaves = static_cast<int>(order.size());
        std::vector<Node> node(static_cast<size_t>(2 * leaves - 1));
        for(int k = 0; k < leaves; ++k) { node[static_cast<size_t>(k)] = {scaled[static_cast<size_t>(order[static_cast<size_t>(k)])], -1, -1}; }
        int leaf = 0, inner = leaves, made = leaves;
        auto take = [&]() {
            if(leaf < leaves && (inner >= made || node[static_cast<size_t>(leaf)].weight <= node[static_cast<size_t>(inner)].weight)) { return leaf++; }
            return inner++;
        };
        while(made < 2 * leaves - 1) {
            const int a = take(), b = take();
            node[static_cast<size_t>(made)] = {node[static_cast<size_t>(a)].weight + node[static_cast<size_t>(b)].weight, a, b};
            ++made;
        }
        // depths, root first (children have smaller indexes than their parent)
        std::vector<int> depth(static_cast<size_t>(2 * leaves - 1), 0);
        int deepest = 0;
        for(int k = 2 * leaves - 2; k >= leaves; --k) {
            depth[static_cast<size_t>(node[static_cast<size_t>(k)].left)] = depth[static_cast<size_t>(k)] + 1;
            depth[static_cast<size_t>(node[static_cast<size_t>(k)].right)] = depth[static_cast<size_t>(k)] + 1;
        }
        for(int k = 0; k < leaves; ++k) { deepest = std::max(deepest, depth[static_cast<size_t>(k)]); }
        if(deepest <= limit) {
            for(int k = 0; k < leaves; ++k) { length[order[static_cast<size_t>(k)]] = static_cast<uint8_t>(depth[static_cast<size_t>(k)]); }
            return;
        }
        for(auto& c : scaled) { if(c) { c = (c + 1) >> 1; } }
    }
}

// canonical codes (RFC 1951 3.2.2), bit reversed: the stream takes a code's most significant bit first, the writer the least
inline void assign_codes(const uint8_t* length, int n, uint16_t* code) {
    int per_length[16] = {0};
    for(int s = 0; s < n; ++s) { ++per_length[length[s]]; }
    per_length[0] = 0;
    uint32_t next[16];
    uint32_t running = 0;
    for(int l = 1; l <= 15; ++l) { running = (running + static_cast<uint32_t>(per_length[l - 1])) << 1; next[l] = running; }
    for(int s = 0; s < n; ++s) {
        const int l = length[s];
        code[s] = l ? static_cast<uint16_t>(fast_inflate::reverse_bits(next[l]++, l)) : 0;
    }
}

struct Coder {
    uint8_t litlen_length[288];
    uint16_t litlen_code[288];
    uint8_t dist_length[32];
    uint16_t dist_code[32];
};

struct LengthSlot { uint16_t symbol; uint8_t extra_bits; uint8_t extra; };

inline const LengthSlot* length_slots() {
    static LengthSlot slot[MAX_MATCH + 1];
    static const bool ready = []() {
        for(int s = 0; s < 29; ++s) {
            const int base = fast_inflate::LENGTH_BASE[s], bits = fast_inflate::LENGTH_EXTRA[s];
            for(int e = 0; e < (1 << bits) && base + e <= MAX_MATCH; ++e) {
                if(s < 28 && base + e == 258) { continue; }                      // 258 is symbol 285 alone
                slot[base + e] = {static_cast<uint16_t>(257 + s), static_cast<uint8_t>(bits), static_cast<uint8_t>(e)};
            }
        }
        return true;
    }();
    (void)ready;
    return slot;
}

inline int dist_symbol(uint32_t distance) {               // 1 .. 32768
    static uint8_t small[513];
    static const bool ready = []() {
        for(int s = 0; s < 30; ++s) {
            const int base = fast_inflate::DIST_BASE[s], bits = fast_inflate::DIST_EXTRA[s];
            for(int e = 0; e < (1 << bits); ++e) { if(base + e <= 512) { small[base + e] = static_cast<uint8_t>(s); } }
        }
        return true;
    }();
    (void)ready;
    if(distance <= 512) { return small[distance]; }
    // past 512 every symbol covers a power of two of distances: two symbols a doubling
    const uint32_t d = distance - 1;
    const int top = 31 - __builtin_clz(d);                // 9 .. 14
    return 2 * top + static_cast<int>((d >> (top - 1)) & 1u);
}

// token: a literal (its byte), or a match: 1 << 31 | (length - 3) << 16 | (distance - 1)
inline void write_block(BitWriter& w, const uint32_t* token, size_t n_token, const uint32_t* litlen_count, const uint32_t* dist_count,
                        const uint8_t* raw, size_t raw_bytes, bool last, Coder& coder) {
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const LengthSlot* const slots = length_slots();
    code_lengths(litlen_count, 286, 15, coder.litlen_length);
    code_lengths(dist_count, 30, 15, coder.dist_length);
    int n_litlen = 286, n_dist = 30;
    while(n_litlen > 257 && coder.litlen_length[n_litlen - 1] == 0) { --n_litlen; }
    while(n_dist > 1 && coder.dist_length[n_dist - 1] == 0) { --n_dist; }
    assign_codes(coder.litlen_length, 286, coder.litlen_code);
    assign_codes(coder.dist_length, 30, coder.dist_code);
    // the code lengths, run length coded with the code length code
    uint8_t all[286 + 30];
    std::memcpy(all, coder.litlen_length, static_cast<size_t>(n_litlen));
    std::memcpy(all + n_litlen, coder.dist_length, static_cast<size_t>(n_dist));
    const int n_all = n_litlen + n_dist;
    struct Run { uint8_t symbol; uint8_t extra; };
    Run run[286 + 30];
    int n_run = 0;
    uint32_t pre_count[19] = {0};
    for(int at = 0; at < n_all;) {
        const uint8_t value = all[at];
        int same = 1;
        while(at + same < n_all && all[at + same] == value) { ++same; }
        int left = same;
        if(value == 0) {
            while(left >= 11) { const int take = std::min(left, 138); run[n_run++] = {18, static_cast<uint8_t>(take - 11)}; ++pre_count[18]; left -= take; }
            if(left >= 3) { run[n_run++] = {17, static_cast<uint8_t>(left - 3)}; ++pre_count[17]; left = 0; }
        } else {
            run[n_run++] = {value, 0}; ++pre_count[value]; --left;
            while(left >= 3) { const int take = std::min(left, 6); run[n_run++] = {16, static_cast<uint8_t>(take - 3)}; ++pre_count[16]; left -= take; }
        }
        while(left > 0) { run[n_run++] = {value, 0}; ++pre_count[value]; --left; }
        at += same;
    }
    uint8_t pre_length[19];
    uint16_t pre_code[19];
    {
        // (a code length code of ONE symbol would be an incomplete code, which inflate does not take there: a second one joins it)
        int used = 0;
        for(int s = 0; s < 19; ++s) { used += pre_count[s] ? 1 : 0; }
        if(used < 2) { ++pre_count[pre_count[0] ? 18 : 0]; }
    }
    code_lengths(pre_count, 19, 7, pre_length);
    assign_codes(pre_length, 19, pre_code);
    int n_pre = 19;
    while(n_pre > 4 && pre_length[ORDER[n_pre - 1]] == 0) { --n_pre; }
    // what the block costs coded, against stored
    uint64_t coded_bits = 3 + 14 + 3u * static_cast<uint64_t>(n_pre);
    for(int k = 0; k < n_run; ++k) { coded_bits += pre_length[run[k].symbol] + (run[k].symbol == 16 ? 2 : run[k].symbol == 17 ? 3 : run[k].symbol == 18 ? 7 : 0); }
    for(int s = 0; s < 286; ++s) { coded_bits += static_cast<uint64_t>(litlen_count[s]) * (coder.litlen_length[s] + (s >= 265 && s < 285 ? static_cast<unsigned>((s - 261) / 4) : 0u)); }
    for(int s = 0; s < 30; ++s) { coded_bits += static_cast<uint64_t>(dist_count[s]) * (coder.dist_length[s] + fast_inflate::DIST_EXTRA[s]); }
    if(coded_bits / 8 + 8 >= raw_bytes + 5 * (raw_bytes / 65535 + 1)) {
        // stored: pieces of at most 65 535 bytes
        size_t at = 0;
        do {
            const size_t piece = std::min<size_t>(raw_bytes - at, 65535);
            w.put((last && at + piece == raw_bytes) ? 1u : 0u, 1);
            w.put(0, 2);
            w.align();
            if(static_cast<size_t>(w.end - w.next) < piece + 4) { w.overflow = true; return; }
            const uint16_t length = static_cast<uint16_t>(piece), check = static_cast<uint16_t>(~length);
            std::memcpy(w.next, &length, 2); std::memcpy(w.next + 2, &check, 2);
            std::memcpy(w.next + 4, raw + at, piece);
            w.next += 4 + piece;
            at += piece;
        } while(at < raw_bytes);
        return;
    }
    w.put(last ? 1u : 0u, 1);
    w.put(2, 2);
    w.put(static_cast<uint32_t>(n_litlen - 257), 5);
    w.put(static_cast<uint32_t>(n_dist - 1), 5);
    w.put(static_cast<uint32_t>(n_pre - 4), 4);
    for(int k = 0; k < n_pre; ++k) { w.put(pre_length[ORDER[k]], 3); }
    for(int k = 0; k < n_run; ++k) {
        w.put(pre_code[run[k].symbol], pre_length[run[k].symbol]);
        if(run[k].symbol == 16) { w.put(run[k].extra, 2); }
        else if(run[k].symbol == 17) { w.put(run[k].extra, 3); }
        else if(run[k].symbol == 18) { w.put(run[k].extra, 7); }
    }
    for(size_t k = 0; k < n_token; ++k) {
        const uint32_t t = token[k];
        if((t >> 31) == 0) { w.put(coder.litlen_code[t], coder.litlen_length[t]); continue; }
        // a match's four fields — length code, its extra bits, distance code, its extra bits: at most 15 + 5 + 15 + 13 — leave as one
        const LengthSlot& slot = slots[((t >> 16) & 0xffu) + 3];
        const uint32_t distance = (t & 0xffffu) + 1;
        const int symbol = dist_symbol(distance);
        uint64_t fields = coder.litlen_code[slot.symbol];
        int n = coder.litlen_length[slot.symbol];
        fields |= static_cast<uint64_t>(slot.extra) << n;
        n += slot.extra_bits;
        fields |= static_cast<uint64_t>(coder.dist_code[symbol]) << n;
        n += coder.dist_length[symbol];
        fields |= static_cast<uint64_t>(distance - fast_inflate::DIST_BASE[symbol]) << n;
        n += fast_inflate::DIST_EXTRA[symbol];
        w.put(fields, n);
    }
    w.put(coder.litlen_code[256], coder.litlen_length[256]);
}

// a raw deflate stream (RFC 1951) of `bytes` of text into `out`; its size, or 0 where `capacity` does not hold it
// (bytes + bytes / 8 + 64 always does: a stored block costs five bytes per 65 535)
inline size_t deflate_raw(const uint8_t* in, size_t bytes, uint8_t* out, size_t capacity) {
    if(capacity < 16) { return 0; }
    BitWriter w;
    w.next = out;
    w.end = out + capacity;
    // (a thread's tables are kept between calls: a BGZF block is 64 KiB of input, its 128 KiB of hash table would be new every time)
    struct Scratch { std::unique_ptr<uint32_t[]> table, token; std::unique_ptr<Coder> coder; };
    static thread_local Scratch scratch;
    if(!scratch.table) { scratch.table.reset(new uint32_t[size_t(1) << HASH_BITS]); scratch.token.reset(new uint32_t[BLOCK_SYMBOLS]); scratch.coder.reset(new Coder); }
    uint32_t* const table = scratch.table.get();
    uint32_t* const token = scratch.token.get();
    Coder* const coder = scratch.coder.get();
    std::memset(table, 0xff, sizeof(uint32_t) << HASH_BITS);
    // what a literal and a match cost, from the block before (the first block: 8 bits a literal, the fixed code's matches)
    uint8_t literal_bits[256];
    std::memset(literal_bits, 8, sizeof(literal_bits));
    int match_bits = 7 + 5;
    const LengthSlot* const slots = length_slots();
    size_t at = 0;
    if(bytes == 0) {
        w.put(1, 1); w.put(1, 2); w.put(0, 7);                    // one fixed block: end of block
    }
    while(at < bytes) {
        uint32_t litlen_count[286] = {0};
        uint32_t dist_count[30] = {0};
        size_t n_token = 0;
        const size_t block_begin = at;
        while(at < bytes && n_token < BLOCK_SYMBOLS) {
            if(bytes - at < 8) {                                   // the member's last bytes: literals
                token[n_token++] = in[at]; ++litlen_count[in[at]]; ++at;
                continue;
            }
            uint32_t four;
            std::memcpy(&four, in + at, 4);
            const uint32_t hash = (four * 0x9e3779b1u) >> (32 - HASH_BITS);
            const uint32_t candidate = table[hash];
            table[hash] = static_cast<uint32_t>(at);
            int length = 0;
            if(candidate != 0xffffffffu && at - candidate <= WINDOW) {
                // eight bytes of both at once (bytes - at >= 8 here, and the candidate lies in front): most candidates of text whose
                // alphabet is small — the bases of a read — are real matches of four to seven bytes, settled by this one comparison
                uint64_t mine, theirs;
                std::memcpy(&mine, in + at, 8);
                std::memcpy(&theirs, in + candidate, 8);
                const uint64_t differ = mine ^ theirs;
                if(differ != 0) { length = static_cast<int>(static_cast<unsigned>(__builtin_ctzll(differ)) >> 3); }
                else {
                    const size_t limit = std::min<size_t>(MAX_MATCH, bytes - at);
                    size_t same = 8;
                    while(same + 8 <= limit) {
                        uint64_t a, b;
                        std::memcpy(&a, in + at + same, 8);
                        std::memcpy(&b, in + candidate + same, 8);
                        if(a != b) { same += static_cast<size_t>(__builtin_ctzll(a ^ b)) >> 3; goto measured; }
                        same += 8;
                    }
                    while(same < limit && in[at + same] == in[candidate + same]) { ++same; }
measured:
                    length = static_cast<int>(std::min(same, limit));
                }
            }
            if(length >= MIN_MATCH) {
                // cheaper than its own bytes as literals?
                int as_literals = 0;
                const int priced = std::min(length, 24);
                for(int k = 0; k < priced; ++k) { as_literals += literal_bits[in[at + static_cast<size_t>(k)]]; }
                if(length > priced) { as_literals += (length - priced) * 2; }
                const uint32_t distance = static_cast<uint32_t>(at - candidate);
                const int symbol = dist_symbol(distance);
                if(match_bits + slots[length].extra_bits + fast_inflate::DIST_EXTRA[symbol] >= as_literals) { length = 0; }
                else {
                    token[n_token++] = (1u << 31) | (static_cast<uint32_t>(length - 3) << 16) | (distance - 1);
                    ++litlen_count[slots[length].symbol];
                    ++dist_count[symbol];
                    // the positions inside a short match join the table; a long one (a run, a repeated line) only with its end
                    const size_t stop = at + static_cast<size_t>(length);
                    if(length <= 32) {
                        for(size_t p = at + 1; p + 4 <= bytes && p < stop; ++p) {
                            uint32_t v;
                            std::memcpy(&v, in + p, 4);
                            table[(v * 0x9e3779b1u) >> (32 - HASH_BITS)] = static_cast<uint32_t>(p);
                        }
                    } else {
                        for(size_t p = stop - 3; p + 4 <= bytes && p < stop; ++p) {
                            uint32_t v;
                            std::memcpy(&v, in + p, 4);
                            table[(v * 0x9e3779b1u) >> (32 - HASH_BITS)] = static_cast<uint32_t>(p);
                        }
                    }
                    at = stop;
                    continue;
                }
            }
            token[n_token++] = in[at]; ++litlen_count[in[at]]; ++at;
        }
        ++litlen_count[256];
        write_block(w, token, n_token, litlen_count, dist_count, in + block_begin, at - block_begin, at == bytes, *coder);
        if(w.overflow) { return 0; }
        // the next block's prices
        for(int b = 0; b < 256; ++b) { literal_bits[b] = coder->litlen_length[b] ? coder->litlen_length[b] : 12; }
        {
            uint64_t bits = 0, matches = 0;
            for(int s = 257; s < 286; ++s) { bits += static_cast<uint64_t>(litlen_count[s]) * coder->litlen_length[s]; matches += litlen_count[s]; }
            uint64_t far = 0;
            for(int s = 0; s < 30; ++s) { far += static_cast<uint64_t>(dist_count[s]) * coder->dist_length[s]; }
            match_bits = matches ? static_cast<int>((bits + far + matches - 1) / matches) : 7 + 5;
        }
    }
    w.align();
    if(w.overflow) { return 0; }
    return static_cast<size_t>(w.next - out);
}

// one gzip member of `bytes` of text into `out`; the member's size, or 0 where `capacity` does not hold it (bytes + bytes / 8 + 128 always does)
inline size_t compress_member(const uint8_t* in, size_t bytes, uint8_t* out, size_t capacity) {
    if(capacity < 32) { return 0; }
    static const uint8_t HEADER[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4, 255};       // XFL 4: the fastest algorithm; OS unknown
    std::memcpy(out, HEADER, 10);
    const size_t packed = deflate_raw(in, bytes, out + 10, capacity - 18);
    if(packed == 0) { return 0; }
    uint8_t* const trailer = out + 10 + packed;
    const uint32_t crc = fast_inflate::crc32_of(in, bytes);
    const uint32_t size = static_cast<uint32_t>(bytes);
    std::memcpy(trailer, &crc, 4);
    std::memcpy(trailer + 4, &size, 4);
    return static_cast<size_t>(trailer + 8 - out);
}

// BGZF (the SAM specification, section 4.1: what a BAM file is made of): `bytes` of data as blocks of at most BGZF_DATA bytes, each
// a gzip member whose extra subfield 'B' 'C' holds its own size - 1.  Appends to `out` (a vector-like with reserve / commit as
// the caller's); bgzf_bound(bytes) bytes of room are enough.
constexpr size_t BGZF_DATA = 0xff00;
constexpr uint8_t BGZF_EOF[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
inline size_t bgzf_bound(size_t bytes) { return bytes + (bytes / BGZF_DATA + 1) * 64 + bytes / 8; }
inline size_t bgzf_blocks(const uint8_t* in, size_t bytes, uint8_t* out, size_t capacity) {
    uint8_t* to = out;
    for(size_t at = 0; at < bytes; at += BGZF_DATA) {
        const size_t piece = std::min(BGZF_DATA, bytes - at);
        if(static_cast<size_t>(out + capacity - to) < piece + 64) { return 0; }
        static const uint8_t HEADER[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        std::memcpy(to, HEADER, 16);
        // (a block is at most 65 536 bytes: data that does not compress is stored, 0xff00 + 5 + 26)
        const size_t packed = deflate_raw(in + at, piece, to + 18, 65536 - 26);
        if(packed == 0) { return 0; }
        const uint16_t block_size = static_cast<uint16_t>(packed + 25);          // the block's size - 1
        std::memcpy(to + 16, &block_size, 2);
        const uint32_t crc = fast_inflate::crc32_of(in + at, piece);
        const uint32_t size = static_cast<uint32_t>(piece);
        std::memcpy(to + 18 + packed, &crc, 4);
        std::memcpy(to + 22 + packed, &size, 4);
        to += 26 + packed;
    }
    return static_cast<size_t>(to - out);
}

}  // namespace fast_deflate
}  // namespace phq
