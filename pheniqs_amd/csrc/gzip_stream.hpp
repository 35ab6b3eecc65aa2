// A gzip input of ONE member (what `gzip file.fastq` and most sequencing pipelines write; the reference reads it through htslib's
// bgzf layer, fastq.h:427-440) inflated by fast_inflate.hpp's decoder in STREAMING form: gzip_members.hpp takes files of many
// small members side by side on every thread, a single member is one sequential deflate stream and was left to zlib's gzread —
// 0.3 GB/s of FASTQ text a file, 4.5 Mreads/s for the bench's two files (40x under the plain text path).
//
// RawStream is fast_inflate::inflate_raw made resumable: the same tables (11 + 8 bit first levels, up to three literals an entry),
// the same unchecked inner loop while sixteen bytes of input and the longest match of room are ahead — but the text goes into a
// work buffer of 1 MiB behind the 32 KiB of history deflate may reach back into (RFC 1951 3.2.5), and when the room runs out the
// decoder returns with its bit buffer, its tables and its place inside the block kept; the caller takes the text, moves the last
// 32 KiB to the front and calls again.  GzipStream wraps it in RFC 1952: header, CRC-32 and ISIZE of every member (several may
// follow one another; each starts with an empty history), the text handed out in pieces of any size.
//
// Safety.  The one shot decoder hands nothing on before the member's CRC holds; a stream cannot wait for its end.  So:
//   * what the decoder is not SURE it can decode — a code length set it cannot table — is found at a block header, before any
//     of the block's text: the rest of the member then goes through zlib's inflate(), primed with the history and the bit
//     offset of that block (inflateSetDictionary / inflatePrime);
//   * damaged data (an invalid code, a distance before the start of the text, input that ends early) and a CRC or size that does
//     not hold at the member's end are an IOError — text may have been handed on by then, exactly as with gzread, and as there
//     the run ends with the error instead of a result.
#pragma once

#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "error.hpp"
#include "fast_inflate.hpp"
#include "host_parallel.hpp"

namespace phq {
namespace fast_inflate {

class RawStream {
public:
    enum class Status { ROOM, END, DAMAGED, REFUSED, BOUNDARY };
    static constexpr size_t MARGIN = 320;          // room the unchecked loop wants ahead: the longest match, three literals, the copies' overshoot

    // the stream starts `bit` bits into `in` (at a block header)
    void start(const uint8_t* in, const uint8_t* in_end, uint64_t bit = 0) {
        origin_ = in; in_next_ = in + (bit >> 3); in_end_ = in_end; bits_ = 0; held_ = 0; blocks_ = 0; last_ = false; in_block_ = false; stored_ = false; stored_left_ = 0; finished_ = false;
        tables_ = nullptr;
        stop_bit_ = ~uint64_t(0);
        if((bit & 7) != 0 && in_next_ < in_end_) { bits_ = static_cast<uint64_t>(*in_next_++) >> (bit & 7); held_ = 8 - static_cast<int>(bit & 7); }
    }
    // bits of the stream taken so far
    uint64_t bit_position() const { return static_cast<uint64_t>(in_next_ - origin_) * 8u - static_cast<uint64_t>(held_); }
    uint64_t header_bit() const { return header_bit_; }
    uint8_t* header_out() const { return header_out_; }
    // run() returns BOUNDARY in front of the first block header at or past this bit (gzip_parallel: the chunk behind starts there)
    void stop_at(uint64_t bit) { stop_bit_ = bit; }
    bool finished() const { return finished_; }
    static long& refuse_block() { static long block = -1; return block; }      // a test seam: the block header with this ordinal is refused
    // the first byte behind the stream (after END of the last block): whole bytes the bit buffer still holds were not the stream's
    const uint8_t* after() const { return in_next_ - (held_ >> 3); }
    // where the block that was REFUSED starts: the byte that holds its first bit, and how many of that byte's low bits came before it
    const uint8_t* boundary_byte() const { return boundary_in_ - ((boundary_held_ + 7) >> 3); }
    int boundary_bit() const { return (8 - (boundary_held_ & 7)) & 7; }

    // Decodes into [out, out_end); `history` <= out is the first byte a distance may reach (the start of the member's text, or of the
    // 32 KiB kept of it).  ROOM: less than MARGIN bytes are left and the stream goes on; END: the last block has ended.
    Status run(const uint8_t* history, uint8_t*& out_io, uint8_t* const out_end) { return run_planes<false>(history, out_io, out_end, 0); }

    // MARKED (a chunk that starts in the middle of the stream, see below): a second plane of flag bytes `delta` bytes behind the
    // text — 0 for a byte that is known; what a match copies out of the 32 KiB in front of the chunk carries its place there, low
    // byte in the text plane, 1 | high bits << 1 in the flag plane.  A literal stores zeros to the flag plane where it stores its
    // bytes, a match copies both planes: the loop is the same one, at most a few stores longer per symbol.
    template <bool MARKED>
    Status run_planes(const uint8_t* history, uint8_t*& out_io, uint8_t* const out_end, ptrdiff_t delta) {
        static const Tables* const fixed = fixed_tables();
        if(fixed == nullptr) { return Status::REFUSED; }
        const uint8_t* in_next = in_next_;
        const uint8_t* const in_end = in_end_;
        uint64_t bits = bits_;
        int held = held_;
        uint8_t* out = out_io;
        auto leave = [&](Status status) { in_next_ = in_next; bits_ = bits; held_ = held; out_io = out; return status; };
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
        for(;;) {
            if(!in_block_) {
                if(finished_) { return leave(Status::END); }
                refill();
                boundary_in_ = in_next; boundary_held_ = held;
                header_bit_ = static_cast<uint64_t>(in_next - origin_) * 8u - static_cast<uint64_t>(held);
                header_out_ = out;
                if(header_bit_ >= stop_bit_) { return leave(Status::BOUNDARY); }
                if(held < 3) { return leave(Status::DAMAGED); }
                const uint64_t bits_before = bits;
                auto refuse = [&]() { in_next = boundary_in_; bits = bits_before; held = boundary_held_; return leave(Status::REFUSED); };
                // (a refused header is read again by zlib from boundary_byte(): only in_next_ and the bit count matter to it)
                if(blocks_ == refuse_block()) { return refuse(); }         // (the check program: the hand over to zlib at a chosen block; refused again when asked again, as a real one is)
                ++blocks_;
                last_ = take(1) != 0;
                const uint32_t type = take(2);
                if(type == 0) {
                    take(held & 7);
                    in_next -= held >> 3;
                    bits = 0; held = 0;
                    if(in_end - in_next < 4) { return leave(Status::DAMAGED); }
                    const uint32_t length = static_cast<uint32_t>(in_next[0]) | (static_cast<uint32_t>(in_next[1]) << 8);
                    const uint32_t check = static_cast<uint32_t>(in_next[2]) | (static_cast<uint32_t>(in_next[3]) << 8);
                    in_next += 4;
                    if((length ^ check) != 0xffffu || static_cast<size_t>(in_end - in_next) < length) { return leave(Status::DAMAGED); }
                    stored_ = true; stored_left_ = length;
                } else if(type == 2) {
                    if(held < 14) { return leave(Status::DAMAGED); }
                    const int n_litlen = static_cast<int>(take(5)) + 257, n_dist = static_cast<int>(take(5)) + 1, n_pre = static_cast<int>(take(4)) + 4;
                    if(n_litlen > 286 || n_dist > 30) { return leave(Status::DAMAGED); }
                    uint8_t pre_lengths[19] = {0};
                    refill();
                    for(int k = 0; k < n_pre; ++k) {
                        if(held < 3) { refill(); if(held < 3) { return leave(Status::DAMAGED); } }
                        pre_lengths[ORDER[k]] = static_cast<uint8_t>(take(3));
                    }
                    uint32_t pre_table[128 + 8];
                    if(!build_table(pre_lengths, 19, 7, pre_table, 128 + 8, false, [](int symbol, uint32_t length) { return entry_of(static_cast<uint32_t>(symbol), LITERAL, 0, length); })) { return refuse(); }
                    uint8_t lengths[286 + 30 + 138];
                    int at = 0;
                    while(at < n_litlen + n_dist) {
                        refill();
                        const uint32_t entry = pre_table[bits & 127u];
                        if(kind_of(entry) != LITERAL || static_cast<int>(entry & 31u) > held) { return refuse(); }
                        take(static_cast<int>(entry & 31u));
                        const uint32_t symbol = entry >> 12;
                        if(symbol < 16) { lengths[at++] = static_cast<uint8_t>(symbol); continue; }
                        uint32_t repeat, value = 0;
                        if(symbol == 16) { if(at == 0) { return refuse(); } value = lengths[at - 1]; repeat = 3 + take(2); }
                        else if(symbol == 17) { repeat = 3 + take(3); }
                        else { repeat = 11 + take(7); }
                        if(held < 0 || at + static_cast<int>(repeat) > n_litlen + n_dist) { return refuse(); }
                        for(uint32_t k = 0; k < repeat; ++k) { lengths[at++] = static_cast<uint8_t>(value); }
                    }
                    if(lengths[256] == 0) { return refuse(); }
                    if(!build_litlen(lengths, n_litlen, dynamic_) || !build_dist(lengths + n_litlen, n_dist, dynamic_)) { return refuse(); }
                    stored_ = false; tables_ = &dynamic_;
                } else if(type == 1) {
                    stored_ = false; tables_ = fixed;
                } else { return leave(Status::DAMAGED); }
                in_block_ = true;
            }
            if(stored_) {
                const size_t n = std::min<size_t>(stored_left_, static_cast<size_t>(out_end - out));
                std::memcpy(out, in_next, n);
                if(MARKED) { std::memset(out + delta, 0, n); }
                in_next += n; out += n; stored_left_ -= static_cast<uint32_t>(n);
                if(stored_left_ > 0) { return leave(Status::ROOM); }
                in_block_ = false;
                finished_ = last_;
                continue;
            }
            const uint32_t* const litlen = tables_->litlen;
            const uint32_t* const dist = tables_->dist;
            constexpr uint32_t LITLEN_MASK = (1u << LITLEN_BITS) - 1u, DIST_MASK = (1u << DIST_BITS) - 1u;
            bool open = true;
            while(open) {
                while(in_end - in_next >= 16 && static_cast<size_t>(out_end - out) >= MARGIN) {
                    {
                        uint64_t word;
                        std::memcpy(&word, in_next, 8);
                        bits |= word << held;
                        in_next += (63 - held) >> 3;
                        held |= 56;
                    }
                    uint32_t entry = litlen[bits & LITLEN_MASK];
                    if(kind_of(entry) <= LITERAL3) {
                        uint32_t bytes = entry >> 8;
                        std::memcpy(out, &bytes, 4);
                        if(MARKED) { std::memset(out + delta, 0, 4); }
                        out += kind_of(entry) + 1u;
                        bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
                        entry = litlen[bits & LITLEN_MASK];
                        if(kind_of(entry) <= LITERAL3) {
                            bytes = entry >> 8;
                            std::memcpy(out, &bytes, 4);
                            if(MARKED) { std::memset(out + delta, 0, 4); }
                            out += kind_of(entry) + 1u;
                            bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
                            entry = litlen[bits & LITLEN_MASK];
                            if(kind_of(entry) <= LITERAL3) {
                                bytes = entry >> 8;
                                std::memcpy(out, &bytes, 4);
                                if(MARKED) { std::memset(out + delta, 0, 4); }
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
                    const uint32_t kind = kind_of(entry);
                    bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
                    if(kind == LITERAL) { if(MARKED) { out[delta] = 0; } *out++ = static_cast<uint8_t>(entry >> 8); continue; }
                    if(kind == END) { open = false; break; }
                    if(kind != LENGTH) { return leave(Status::DAMAGED); }
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
                    if(kind_of(entry) != DISTANCE) { return leave(Status::DAMAGED); }
                    bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
                    const uint32_t more = extra_of(entry);
                    const uint32_t distance = (entry >> 12) + static_cast<uint32_t>(bits & ((1u << more) - 1u));
                    bits >>= more; held -= static_cast<int>(more);
                    if(distance > static_cast<size_t>(out - history)) { return leave(Status::DAMAGED); }
                    const uint8_t* from = out - distance;
                    uint8_t* const stop = out + length;
                    if(distance >= 8) {
                        if(MARKED) {
                            const uint8_t* flags_from = from + delta;
                            uint8_t* flags = out + delta;
                            uint8_t* const flags_stop = stop + delta;
                            do { uint64_t word; std::memcpy(&word, flags_from, 8); std::memcpy(flags, &word, 8); flags_from += 8; flags += 8; } while(flags < flags_stop);
                        }
                        do { uint64_t word; std::memcpy(&word, from, 8); std::memcpy(out, &word, 8); from += 8; out += 8; } while(out < stop);
                    } else if(distance == 1) {
                        if(MARKED) { std::memset(out + delta, from[delta], length); }
                        std::memset(out, *from, length);
                    } else {
                        if(MARKED) { for(uint32_t k = 0; k < length; ++k) { out[delta + static_cast<ptrdiff_t>(k)] = from[delta + static_cast<ptrdiff_t>(k)]; } }
                        for(uint32_t k = 0; k < 8; ++k) { out[k] = from[k]; }
                        if(length > 8) {
                            const uint32_t stride = distance * ((8u + distance - 1u) / distance);
                            const uint8_t* again = out + 8 - stride;
                            uint8_t* to = out + 8;
                            do { uint64_t word; std::memcpy(&word, again, 8); std::memcpy(to, &word, 8); again += 8; to += 8; } while(to < stop);
                        }
                    }
                    out = stop;
                }
                if(!open) { break; }
                if(static_cast<size_t>(out_end - out) < MARGIN) { return leave(Status::ROOM); }       // (inside the block: the tables and the bit buffer wait)
                // the last sixteen bytes of the input: one symbol at a time, every bound checked
                refill();
                uint32_t entry = litlen[bits & LITLEN_MASK];
                if(kind_of(entry) == SUBTABLE) {
                    entry = litlen[(entry >> 12) + ((bits >> LITLEN_BITS) & ((1u << extra_of(entry)) - 1u))];
                    bits >>= LITLEN_BITS; held -= LITLEN_BITS;
                }
                uint32_t kind = kind_of(entry);
                bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
                if(held < 0) { return leave(Status::DAMAGED); }
                if(kind <= LITERAL3) {
                    for(uint32_t k = 0; k <= kind; ++k) { if(MARKED) { out[delta] = 0; } *out++ = static_cast<uint8_t>(entry >> (8 + 8 * k)); }
                    continue;
                }
                if(kind == END) { break; }
                if(kind != LENGTH) { return leave(Status::DAMAGED); }
                const uint32_t length = (entry >> 12) + take(static_cast<int>(extra_of(entry)));
                if(held < 32) { refill(); }
                entry = dist[bits & DIST_MASK];
                if(kind_of(entry) == SUBTABLE) {
                    entry = dist[(entry >> 12) + ((bits >> DIST_BITS) & ((1u << extra_of(entry)) - 1u))];
                    bits >>= DIST_BITS; held -= DIST_BITS;
                }
                kind = kind_of(entry);
                bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
                if(kind != DISTANCE) { return leave(Status::DAMAGED); }
                const uint32_t distance = (entry >> 12) + take(static_cast<int>(extra_of(entry)));
                if(held < 0 || distance > static_cast<size_t>(out - history)) { return leave(Status::DAMAGED); }
                const uint8_t* from = out - distance;
                for(uint32_t k = 0; k < length; ++k) { out[k] = from[k]; if(MARKED) { out[delta + static_cast<ptrdiff_t>(k)] = from[delta + static_cast<ptrdiff_t>(k)]; } }
                out += length;
            }
            in_block_ = false;
            finished_ = last_;
        }
    }

private:
    static const Tables* fixed_tables() {
        static Tables fixed;
        static const bool ready = []() {
            uint8_t lengths[288];
            for(int s = 0; s < 144; ++s) { lengths[s] = 8; }
            for(int s = 144; s < 256; ++s) { lengths[s] = 9; }
            for(int s = 256; s < 280; ++s) { lengths[s] = 7; }
            for(int s = 280; s < 288; ++s) { lengths[s] = 8; }
            uint8_t all[32];
            for(int s = 0; s < 32; ++s) { all[s] = 5; }
            return build_litlen(lengths, 288, fixed) && build_table(all, 32, DIST_BITS, fixed.dist, DIST_ENTRIES, true, [](int symbol, uint32_t length) {
                return symbol > 29 ? entry_of(0, INVALID, 0, length) : entry_of(DIST_BASE[symbol], DISTANCE, DIST_EXTRA[symbol], length); });
        }();
        return ready ? &fixed : nullptr;
    }

    const uint8_t* origin_ = nullptr;
    const uint8_t* in_next_ = nullptr;
    const uint8_t* in_end_ = nullptr;
    uint64_t stop_bit_ = ~uint64_t(0);
    uint64_t header_bit_ = 0;                      // the last block header looked at, and where the text stood in front of it
    uint8_t* header_out_ = nullptr;
    uint64_t bits_ = 0;
    int held_ = 0;
    bool last_ = false, in_block_ = false, stored_ = false, finished_ = false;
    uint32_t stored_left_ = 0;
    long blocks_ = 0;
    const Tables* tables_ = nullptr;
    Tables dynamic_;
    const uint8_t* boundary_in_ = nullptr;
    int boundary_held_ = 0;
};

// ---- a single deflate stream on several threads ------------------------------------------------------------------------------
// One thread inflates ~1 GB/s of FASTQ text; a file of one member gives the others nothing to do.  The way out (Kerbiriou and
// Chikhi, "Parallel decompression of gzip-compressed files and random access to DNA sequences", 2019: pugz) is to start in the
// MIDDLE of the stream: a deflate block can be decoded without what came before it, except that its matches may reach up to
// 32 KiB back into text that is not known yet.  So a chunk that starts at a block header found by search is decoded with a
// second plane of flag bytes beside its text (RawStream::run_planes<true>): a byte is known, or it is a MARKER — "the byte at
// this place of the 32 KiB in front of the chunk" — matches copy markers like literals, and when the chunk before has been
// resolved its last 32 KiB turn every marker into its byte.  (A first version decoded into 16 bit symbols with a loop of its
// own: 0.25 of the one thread rate a thread, profiles/r06_gzip_stream_rate_u16.txt; the planes keep the byte decoder's loop.)
//   * find_block: the first bit position in a range where a dynamic, non final block header parses — the code length code, the
//     literal / length code and the distance code all complete prefix codes — and its first symbols decode.  A false find costs
//     nothing but time: a chunk is taken only when the chunk before ENDS exactly on its first bit.
constexpr size_t MARKED_WINDOW = size_t(1) << 15;

struct BitReader {
    const uint8_t* next = nullptr;
    const uint8_t* end = nullptr;
    uint64_t bits = 0;
    int held = 0;
    void refill() {
        if(end - next >= 8) {
            uint64_t word;
            std::memcpy(&word, next, 8);
            bits |= word << held;
            next += (63 - held) >> 3;
            held |= 56;
        } else {
            while(held <= 56 && next < end) { bits |= static_cast<uint64_t>(*next++) << held; held += 8; }
        }
    }
    uint32_t take(int n) { const uint32_t v = static_cast<uint32_t>(bits & ((uint64_t(1) << n) - 1u)); bits >>= n; held -= n; return v; }
    void seek(const uint8_t* origin, const uint8_t* limit, uint64_t bit) {
        next = origin + (bit >> 3); end = limit; bits = 0; held = 0;
        refill();
        const int r = static_cast<int>(bit & 7);
        bits >>= r; held -= r;
    }
    uint64_t position(const uint8_t* origin) const { return static_cast<uint64_t>(next - origin) * 8u - static_cast<uint64_t>(held); }
};

// the code lengths of a dynamic block (behind its three header bits) into tables; false: not a block this decoder takes
inline bool read_dynamic_tables(BitReader& r, Tables& tables) {
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    r.refill();
    if(r.held < 14) { return false; }
    const int n_litlen = static_cast<int>(r.take(5)) + 257, n_dist = static_cast<int>(r.take(5)) + 1, n_pre = static_cast<int>(r.take(4)) + 4;
    if(n_litlen > 286 || n_dist > 30) { return false; }
    uint8_t pre_lengths[19] = {0};
    r.refill();
    for(int k = 0; k < n_pre; ++k) {
        if(r.held < 3) { r.refill(); if(r.held < 3) { return false; } }
        pre_lengths[ORDER[k]] = static_cast<uint8_t>(r.take(3));
    }
    uint32_t pre_table[128 + 8];
    if(!build_table(pre_lengths, 19, 7, pre_table, 128 + 8, false, [](int symbol, uint32_t length) { return entry_of(static_cast<uint32_t>(symbol), LITERAL, 0, length); })) { return false; }
    uint8_t lengths[286 + 30 + 138];
    int at = 0;
    while(at < n_litlen + n_dist) {
        r.refill();
        const uint32_t entry = pre_table[r.bits & 127u];
        if(kind_of(entry) != LITERAL || static_cast<int>(entry & 31u) > r.held) { return false; }
        r.take(static_cast<int>(entry & 31u));
        const uint32_t symbol = entry >> 12;
        if(symbol < 16) { lengths[at++] = static_cast<uint8_t>(symbol); continue; }
        uint32_t repeat, value = 0;
        if(symbol == 16) { if(at == 0) { return false; } value = lengths[at - 1]; repeat = 3 + r.take(2); }
        else if(symbol == 17) { repeat = 3 + r.take(3); }
        else { repeat = 11 + r.take(7); }
        if(r.held < 0 || at + static_cast<int>(repeat) > n_litlen + n_dist) { return false; }
        for(uint32_t k = 0; k < repeat; ++k) { lengths[at++] = static_cast<uint8_t>(value); }
    }
    if(lengths[256] == 0) { return false; }
    return build_litlen(lengths, n_litlen, tables) && build_dist(lengths + n_litlen, n_dist, tables);
}

// the first bit in [from_bit, until_bit) where a dynamic, non final block header parses into complete codes and the block's first
// symbols decode; false: none
inline bool find_block(const uint8_t* origin, const uint8_t* limit, uint64_t from_bit, uint64_t until_bit, uint64_t* found) {
    std::unique_ptr<Tables> tables(new Tables);
    const uint64_t last_bit = static_cast<uint64_t>(limit - origin) * 8u;
    for(uint64_t bit = from_bit; bit < until_bit && bit + 200 < last_bit; ++bit) {
        const uint8_t* const at = origin + (bit >> 3);
        if(limit - at < 24) { return false; }
        uint64_t low, high;
        std::memcpy(&low, at, 8);
        std::memcpy(&high, at + 8, 8);
        const int shift = static_cast<int>(bit & 7);
        const uint64_t v = shift ? ((low >> shift) | (high << (64 - shift))) : low;
        // BFINAL 0, BTYPE 2 (bits 1..2 = 10 binary, least significant first), HLIT <= 29, HDIST <= 29
        if((v & 7u) != 4u) { continue; }
        if(((v >> 3) & 31u) > 29u || ((v >> 8) & 31u) > 29u) { continue; }
        const int n_pre = static_cast<int>((v >> 13) & 15u) + 4;
        // the code length code: 3 bits a length from bit 17 — 57 bits at most, 47 of them in v — must be a complete prefix code
        const uint64_t w = shift ? ((high >> shift) | 0) : high;           // bits 64.. of the candidate
        uint32_t space = 0;
        int used = 0;
        for(int k = 0; k < n_pre; ++k) {
            const int lo = 17 + 3 * k;
            uint32_t l;
            if(lo + 3 <= 64) { l = static_cast<uint32_t>((v >> lo) & 7u); }
            else if(lo >= 64) { l = static_cast<uint32_t>((w >> (lo - 64)) & 7u); }
            else { l = static_cast<uint32_t>(((v >> lo) | (w << (64 - lo))) & 7u); }
            if(l) { space += 128u >> l; ++used; }
        }
        if(space != 128u || used < 2) { continue; }
        BitReader r;
        r.seek(origin, limit, bit + 3);
        if(!read_dynamic_tables(r, *tables)) { continue; }
        // the block's first symbols: no invalid code, no distance past the window
        bool plausible = true;
        size_t made = 0;
        constexpr uint32_t LITLEN_MASK = (1u << LITLEN_BITS) - 1u, DIST_MASK = (1u << DIST_BITS) - 1u;
        for(int symbol = 0; symbol < 512 && plausible; ++symbol) {
            r.refill();
            uint32_t entry = tables->litlen[r.bits & LITLEN_MASK];
            if(kind_of(entry) == SUBTABLE) {
                entry = tables->litlen[(entry >> 12) + ((r.bits >> LITLEN_BITS) & ((1u << extra_of(entry)) - 1u))];
                r.bits >>= LITLEN_BITS; r.held -= LITLEN_BITS;
            }
            uint32_t kind = kind_of(entry);
            r.bits >>= entry & 31u; r.held -= static_cast<int>(entry & 31u);
            if(r.held < 0) { plausible = false; break; }
            if(kind <= LITERAL3) { made += kind + 1u; continue; }
            if(kind == END) { break; }
            if(kind != LENGTH) { plausible = false; break; }
            const uint32_t length = (entry >> 12) + r.take(static_cast<int>(extra_of(entry)));
            if(r.held < 32) { r.refill(); }
            entry = tables->dist[r.bits & DIST_MASK];
            if(kind_of(entry) == SUBTABLE) {
                entry = tables->dist[(entry >> 12) + ((r.bits >> DIST_BITS) & ((1u << extra_of(entry)) - 1u))];
                r.bits >>= DIST_BITS; r.held -= DIST_BITS;
            }
            kind = kind_of(entry);
            r.bits >>= entry & 31u; r.held -= static_cast<int>(entry & 31u);
            if(kind != DISTANCE) { plausible = false; break; }
            const uint32_t distance = (entry >> 12) + r.take(static_cast<int>(extra_of(entry)));
            if(r.held < 0 || distance > MARKED_WINDOW + made) { plausible = false; break; }
            made += length;
        }
        if(plausible) { *found = bit; return true; }
    }
    return false;
}

}  // namespace fast_inflate

// the text of a gzip file of one member — or of a few large ones — in pieces of any size
class GzipStream {
public:
    GzipStream() = default;
    GzipStream(const GzipStream&) = delete;
    GzipStream& operator=(const GzipStream&) = delete;
    ~GzipStream() {
        if(zlib_open_) { inflateEnd(&zlib_); }
        if(map_ != nullptr) { ::munmap(const_cast<uint8_t*>(map_), size_); }
    }

    // true: a regular file that starts with a gzip header.  threads > 1: a member of several MiB is inflated chunk parallel (above)
    bool open(const std::string& path, int threads = 1) {
        threads_ = threads < 1 ? 1 : threads > 64 ? 64 : threads;
        const int descriptor = ::open(path.c_str(), O_RDONLY);
        if(descriptor < 0) { return false; }
        struct stat status;
        if(::fstat(descriptor, &status) != 0 || !S_ISREG(status.st_mode) || status.st_size < 18) { ::close(descriptor); return false; }
        size_ = static_cast<size_t>(status.st_size);
        void* const mapped = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, descriptor, 0);
        ::close(descriptor);
        if(mapped == MAP_FAILED) { return false; }
        map_ = static_cast<const uint8_t*>(mapped);
        (void)::madvise(mapped, size_, MADV_SEQUENTIAL);
        work_.reset(new uint8_t[WINDOW + CHUNK + fast_inflate::RawStream::MARGIN]);
        at_ = 0;
        return begin_member();
    }

    // up to `want` bytes of the text, in order; 0: the end.  IOError: damaged data, or a member whose CRC or size does not hold.
    size_t read(uint8_t* to, size_t want) {
        size_t done = 0;
        while(done < want) {
            if(ready_ == produced_) {
                if(piece_ + 1 < pieces_.size()) {
                    // the chunks of a span: copied out side by side where they all fit (one thread's copy of a span's text took as long as
                    // its decoding on sixteen)
                    size_t behind = 0;
                    for(size_t k = piece_ + 1; k < pieces_.size(); ++k) { behind += pieces_[k].second; }
                    if(threads_ > 1 && behind <= want - done && pieces_.size() - piece_ > 2) {
                        std::vector<size_t> place(pieces_.size(), 0);
                        size_t at = done;
                        for(size_t k = piece_ + 1; k < pieces_.size(); ++k) { place[k] = at; at += pieces_[k].second; }
                        const size_t first_piece = piece_ + 1;
                        parallel_ranges(static_cast<int64_t>(pieces_.size() - first_piece), 1, threads_, [&](int64_t first, int64_t last, int) {
                            for(int64_t k = first; k < last; ++k) { const size_t p = first_piece + static_cast<size_t>(k); std::memcpy(to + place[p], pieces_[p].first, pieces_[p].second); }
                        });
                        done = at;
                        piece_ = pieces_.size() - 1;
                        ready_ = produced_ = pieces_[piece_].first + pieces_[piece_].second;
                        continue;
                    }
                    ++piece_; ready_ = pieces_[piece_].first; produced_ = ready_ + pieces_[piece_].second; continue;
                }
                if(ended_) { break; }
                produce();
                continue;
            }
            const size_t take = std::min(want - done, static_cast<size_t>(produced_ - ready_));
            if(threads_ > 1 && take >= (size_t(8) << 20)) {
                // a span's text: tens of MiB, copied on several threads (one thread's copy was a sixth of a span's time)
                uint8_t* const target = to + done;
                const uint8_t* const source = ready_;
                parallel_ranges(static_cast<int64_t>((take + (1 << 20) - 1) >> 20), 1, std::min(threads_, 8), [&](int64_t first, int64_t last, int) {
                    const size_t begin = static_cast<size_t>(first) << 20, end = std::min(take, static_cast<size_t>(last) << 20);
                    std::memcpy(target + begin, source + begin, end - begin);
                });
            } else { std::memcpy(to + done, ready_, take); }
            ready_ += take;
            done += take;
        }
        return done;
    }

    static std::atomic<uint64_t>& through_zlib() { static std::atomic<uint64_t> count{0}; return count; }      // members the decoder handed to zlib part way (diagnosis, the check program)
    // chunks taken from the other threads / chunks decoded for nothing (a find that was no block header, a chunk behind one that stopped early)
    static std::atomic<uint64_t>& chunks_taken() { static std::atomic<uint64_t> count{0}; return count; }
    static std::atomic<uint64_t>& chunks_dropped() { static std::atomic<uint64_t> count{0}; return count; }

private:
    static constexpr size_t WINDOW = size_t(1) << 15;
    static constexpr size_t CHUNK = size_t(1) << 20;
    static constexpr size_t MIN_CHUNK = size_t(256) << 10;       // compressed bytes a thread: a search for a block header costs a few ms
    static constexpr size_t MAX_CHUNK = size_t(1) << 20;
    static constexpr size_t SEARCH = size_t(128) << 10;          // input bytes a chunk is searched for a block header
    static constexpr size_t CHUNK_TEXT = size_t(5) << 20;        // text a chunk should make

    bool begin_member() {
        const uint8_t* const in = map_ + at_;
        const size_t bytes = size_ - at_;
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
        deflate_ = in + at;
        raw_.start(deflate_, map_ + size_);
        tail_pending_ = false;
        pieces_.clear();
        piece_ = 0;
        history_ = work_.get() + WINDOW;               // nothing to reach back into yet
        ready_ = produced_ = work_.get() + WINDOW;
        crc_ = 0;
        text_ = 0;
        return true;
    }

    // the next piece of the text into the work buffer, behind the history (called when everything made so far has been handed out)
    void produce() {
        if(next_member_) {                              // cat a.gz b.gz: the member before has ended and its text is out
            next_member_ = false;
            if(!begin_member()) { throw IOError("damaged gzip header"); }
        }
        if(threads_ >= 4 && !zlib_open_) {              // (two or three threads do not pay: a chunk behind this thread's own costs twice its decoding — search, markers, copy)
            // (a span that took no chunk — no header found, or none that chained — is followed by more and more pieces on one
            // thread before the next attempt)
            if(span_wait_ > 0) { --span_wait_; }
            else if(span()) { return; }
        }
        // the last 32 KiB of the member's text move to the front
        uint8_t* const base = work_.get() + WINDOW;
        pieces_.clear();
        piece_ = 0;
        const size_t keep = history_into(base);
        history_ = base - keep;
        ready_ = produced_ = base;
        uint8_t* out = base;
        uint8_t* const out_end = base + CHUNK + fast_inflate::RawStream::MARGIN;
        bool member_ended = false;
        if(!zlib_open_) {
            const auto status = raw_.run(history_, out, out_end);
            if(status == fast_inflate::RawStream::Status::DAMAGED) { throw IOError("damaged gzip data"); }
            if(status == fast_inflate::RawStream::Status::REFUSED) {
                // a block this decoder cannot table: zlib takes the member from here, with the history as its dictionary
                std::memset(&zlib_, 0, sizeof(zlib_));
                if(inflateInit2(&zlib_, -MAX_WBITS) != Z_OK) { throw InternalError("inflateInit2"); }
                zlib_open_ = true;
                ++through_zlib();
                const size_t dictionary = std::min(static_cast<size_t>(out - history_), WINDOW);
                if(dictionary > 0 && inflateSetDictionary(&zlib_, out - dictionary, static_cast<uInt>(dictionary)) != Z_OK) { throw InternalError("inflateSetDictionary"); }
                const uint8_t* byte = raw_.boundary_byte();
                const int bit = raw_.boundary_bit();
                if(bit > 0) {
                    if(inflatePrime(&zlib_, 8 - bit, *byte >> bit) != Z_OK) { throw InternalError("inflatePrime"); }
                    ++byte;
                }
                zlib_in_ = byte;
            } else { member_ended = status == fast_inflate::RawStream::Status::END; }
        }
        if(zlib_open_) {
            while(static_cast<size_t>(out_end - out) >= fast_inflate::RawStream::MARGIN) {
                zlib_.next_in = const_cast<Bytef*>(zlib_in_);
                zlib_.avail_in = static_cast<uInt>(std::min<size_t>(static_cast<size_t>(map_ + size_ - zlib_in_), size_t(1) << 30));
                zlib_.next_out = out;
                zlib_.avail_out = static_cast<uInt>(static_cast<size_t>(out_end - out) - (fast_inflate::RawStream::MARGIN - 1));
                const int status = inflate(&zlib_, Z_NO_FLUSH);
                const bool progressed = zlib_.next_out != out || zlib_.next_in != zlib_in_;
                zlib_in_ = zlib_.next_in;
                out = zlib_.next_out;
                if(status == Z_STREAM_END) { member_ended = true; break; }
                if(status != Z_OK && status != Z_BUF_ERROR) { throw IOError("damaged gzip data"); }
                if(!progressed) { throw IOError("a gzip stream ends early"); }
            }
        }
        produced_ = out;
        const size_t made = static_cast<size_t>(out - base);
        crc_ = crc_more(crc_, base, made);
        text_ += made;
        if(member_ended) {
            const uint8_t* const trailer = zlib_open_ ? zlib_in_ : raw_.after();
            if(zlib_open_) { inflateEnd(&zlib_); zlib_open_ = false; }
            end_member(trailer);
        } else if(made == 0) { throw IOError("damaged gzip data"); }       // no progress and no end
    }

    // the last 32 KiB (or all there is) of the member's text so far to the bytes in front of `to`; how many
    size_t history_into(uint8_t* to) {
        size_t keep;
        if(tail_pending_) {
            keep = static_cast<size_t>(std::min<uint64_t>(text_, WINDOW));
            std::memcpy(to - keep, tail_window_.data() + WINDOW - keep, keep);
            tail_pending_ = false;
        } else {
            keep = std::min(static_cast<size_t>(produced_ - history_), WINDOW);
            std::memmove(to - keep, produced_ - keep, keep);
        }
        return keep;
    }

    void end_member(const uint8_t* trailer) {
        if(trailer + 8 > map_ + size_) { throw IOError("a gzip stream ends early"); }
        if(le32(trailer) != crc_ || le32(trailer + 4) != static_cast<uint32_t>(text_)) { throw IOError("a gzip member does not hold to its CRC or its size"); }
        at_ = static_cast<size_t>(trailer + 8 - map_);
        // another member behind it, or the end (bytes that are no gzip header end the text, as they end gzread's)
        next_member_ = at_ + 18 <= size_ && map_[at_] == 0x1f && map_[at_ + 1] == 0x8b;
        ended_ = !next_member_;
    }

    // One span of the member on threads_ threads (see above): this thread goes on from where the stream stands, the
    // others start at block headers found in the chunks behind, every chunk runs to the first block header in the next one;
    // then the chunks are taken in order for as long as each ENDS where the next one STARTS, their markers resolved from the
    // 32 KiB in front of them.  Whatever does not chain — a find that was no header, a chunk out of room — is dropped, and the
    // stream goes on from the last bit taken, on this path or the one thread one.  false: not worth it here (little input left).
    bool span() {
        const uint8_t* const origin = deflate_;
        const uint8_t* const limit = map_ + size_;
        const uint64_t here = raw_.bit_position();
        const size_t left = static_cast<size_t>(limit - origin) - static_cast<size_t>(here >> 3);
        if(raw_.finished() || left < 4 * MIN_CHUNK) { return false; }
        consumed_ = static_cast<size_t>(here >> 3);
        // a chunk: about CHUNK_TEXT bytes of text at the ratio the stream has shown so far (FASTQ: 4 to 5; with constant qualities 8
        // and more), and half as much again of room — a chunk that makes more is cut at its last whole block and the chain ends there
        const double ratio = consumed_ > (size_t(64) << 10) ? std::max(1.0, static_cast<double>(text_) / static_cast<double>(consumed_)) : 5.0;
        const size_t wanted = std::max(MIN_CHUNK, std::min(MAX_CHUNK, static_cast<size_t>(static_cast<double>(CHUNK_TEXT) / ratio)));
        const int chunks = static_cast<int>(std::min<size_t>(static_cast<size_t>(threads_), left / wanted));
        if(chunks < 2) { return false; }
        const size_t chunk_bytes = std::min(wanted, left / static_cast<size_t>(chunks));
        const size_t room = static_cast<size_t>(static_cast<double>(chunk_bytes) * ratio * 1.5) + (size_t(256) << 10);      // text a chunk may make before it is cut
        // the window in front of the span: the last 32 KiB of the member's text so far (they may sit in the buffer that grows below)
        std::vector<uint8_t> kept(WINDOW);
        const size_t keep = history_into(kept.data() + WINDOW);
        if(span_text_room_ < WINDOW + room + fast_inflate::RawStream::MARGIN + 64) {
            span_text_room_ = WINDOW + room + fast_inflate::RawStream::MARGIN + 64;
            span_text_.reset(new uint8_t[span_text_room_]);
        }
        if(marked_.size() < static_cast<size_t>(chunks)) { marked_.resize(static_cast<size_t>(chunks)); }
        uint8_t* const base = span_text_.get() + WINDOW;
        if(keep > 0) { std::memcpy(base - keep, kept.data() + WINDOW - keep, keep); }
        history_ = base - keep;
        ready_ = produced_ = base;
        pieces_.clear();
        piece_ = 0;

        struct Chunk { uint64_t begin = 0, end = 0; bool found = false, ended = false, damaged = false; size_t made = 0; };
        std::vector<Chunk> chunk(static_cast<size_t>(chunks));
        const uint64_t first_byte = here >> 3;
        // Every chunk behind this thread's own starts at the first block header found in its part of the input and runs to the
        // first block header in the next part: the same header, when the search takes every header there is (a FASTQ stream's
        // blocks are all dynamic); where it is not — a stored or fixed block, a find that was none — the chain below breaks
        // there and the stream goes on from the last bit taken.  (Search and decoding are ONE parallel phase: a search phase of
        // its own in front cost a spawn and a join of every thread per span.)
        auto nominal = [&](int64_t k) { return (first_byte + static_cast<uint64_t>(k) * chunk_bytes) * 8u; };
        // a chunk's planes: WINDOW of markers, then `plane` bytes of text; the flag plane `plane_stride` behind
        const size_t plane = room + fast_inflate::RawStream::MARGIN;
        const size_t plane_stride = WINDOW + plane + 64;
        fast_inflate::RawStream::Status own = fast_inflate::RawStream::Status::ROOM;
        uint8_t* own_out = base;
        parallel_ranges(chunks, 1, chunks, [&](int64_t first, int64_t last, int) {
            for(int64_t k = first; k < last; ++k) {
                if(k == 0) {
                    // this thread's own chunk: bytes, with the history it has
                    raw_.stop_at(nominal(1));
                    own = raw_.run(history_, own_out, base + room + fast_inflate::RawStream::MARGIN);
                    raw_.stop_at(~uint64_t(0));
                    continue;
                }
                Chunk& c = chunk[static_cast<size_t>(k)];
                // (a block of zlib's or pigz's is tens of KiB of input: what is not found within SEARCH bytes is not there — text that
                // does not compress, stored or fixed blocks — and a search through all of a chunk would cost more than its decoding)
                c.found = fast_inflate::find_block(origin, limit, nominal(k), std::min(nominal(k) + SEARCH * 8u, nominal(k + 1)), &c.begin);
                if(!c.found) { continue; }
                std::vector<uint8_t>& planes = marked_[static_cast<size_t>(k)];
                if(planes.size() < 2 * plane_stride) { planes.resize(2 * plane_stride); }
                uint8_t* const text = planes.data();
                for(size_t i = 0; i < WINDOW; ++i) { text[i] = static_cast<uint8_t>(i); text[plane_stride + i] = static_cast<uint8_t>(1u | ((i >> 8) << 1)); }
                std::unique_ptr<fast_inflate::RawStream> stream(new fast_inflate::RawStream);
                stream->start(origin, limit, c.begin);
                stream->stop_at(nominal(k + 1));
                uint8_t* out = text + WINDOW;
                const auto status = stream->run_planes<true>(text, out, text + WINDOW + plane, static_cast<ptrdiff_t>(plane_stride));
                if(status == fast_inflate::RawStream::Status::BOUNDARY) { c.end = stream->bit_position(); c.made = static_cast<size_t>(out - (text + WINDOW)); }
                else if(status == fast_inflate::RawStream::Status::END) { c.end = static_cast<uint64_t>(stream->after() - origin) * 8u; c.made = static_cast<size_t>(out - (text + WINDOW)); c.ended = true; }
                else {
                    // out of room, or a block the stream itself will look at again (damage, or one for zlib): what was made up to
                    // the last block header stands
                    c.end = stream->header_bit();
                    c.made = stream->header_out() != nullptr ? static_cast<size_t>(stream->header_out() - (text + WINDOW)) : 0;
                    c.damaged = status != fast_inflate::RawStream::Status::ROOM;
                }
            }
        });
        if(own == fast_inflate::RawStream::Status::DAMAGED) { throw IOError("damaged gzip data"); }
        const size_t own_made = static_cast<size_t>(own_out - base);
        size_t made = own_made;
        bool member_ended = own == fast_inflate::RawStream::Status::END;
        uint64_t end_bit = raw_.bit_position();
        // the chain: chunk k is taken when what has been taken so far ends on its first bit
        std::vector<int> taken;
        if(own == fast_inflate::RawStream::Status::BOUNDARY) {
            for(int k = 1; k < chunks; ++k) {
                const Chunk& c = chunk[static_cast<size_t>(k)];
                if(!c.found || c.begin < end_bit) { continue; }
                if(c.begin != end_bit || c.made == 0) { break; }
                taken.push_back(k);
                made += c.made;
                end_bit = c.end;
                if(c.ended) { member_ended = true; break; }
                if(c.end < nominal(k + 1)) { break; }       // it stopped early: the stream goes on from there (and finds the damage, if it is one)
            }
        }
        int found = 0;
        for(const Chunk& c : chunk) { found += c.found ? 1 : 0; }
        span_wait_ = taken.empty() ? (span_backoff_ = std::min(2 * span_backoff_ + 1, 255)) : (span_backoff_ = 0);
        chunks_taken() += taken.size();
        chunks_dropped() += static_cast<uint64_t>(found) - taken.size();
        // the window in front of every chunk taken — the last 32 KiB of the text before it — and the one behind the last: resolved
        // one behind another (32 KiB each)
        std::vector<std::vector<uint8_t>> window(taken.size() + 1);
        {
            std::vector<uint8_t>& w = window[0];
            w.assign(WINDOW, 0);
            const size_t have = std::min(WINDOW, static_cast<size_t>(base + own_made - history_));
            std::memcpy(w.data() + WINDOW - have, base + own_made - have, have);
        }
        for(size_t t = 0; t < taken.size(); ++t) {
            const Chunk& c = chunk[static_cast<size_t>(taken[t])];
            const uint8_t* const text = marked_[static_cast<size_t>(taken[t])].data() + WINDOW;
            const uint8_t* const flags = text + plane_stride;
            const std::vector<uint8_t>& older = window[t];
            std::vector<uint8_t>& w = window[t + 1];
            w.assign(WINDOW, 0);
            // place i of the window behind this chunk = byte (made - WINDOW + i) of the chunk, or (made < WINDOW) the older window's tail
            const size_t from_chunk = std::min(c.made, WINDOW), from_older = WINDOW - from_chunk;
            if(from_older > 0) { std::memcpy(w.data(), older.data() + c.made, from_older); }
            std::memcpy(w.data() + from_older, text + c.made - from_chunk, from_chunk);
            resolve(w.data() + from_older, flags + c.made - from_chunk, from_chunk, older.data());
        }
        // every chunk's markers into bytes, in place, and its CRC
        std::vector<uint32_t> crc(taken.size(), 0);
        parallel_ranges(static_cast<int64_t>(taken.size()), 1, threads_, [&](int64_t first, int64_t last, int) {
            for(int64_t t = first; t < last; ++t) {
                const Chunk& c = chunk[static_cast<size_t>(taken[static_cast<size_t>(t)])];
                uint8_t* const text = marked_[static_cast<size_t>(taken[static_cast<size_t>(t)])].data() + WINDOW;
                const uint8_t* const flags = text + plane_stride;
                const uint8_t* const w = window[static_cast<size_t>(t)].data();
                const size_t n = c.made;
                resolve(text, flags, n, w);
                crc[static_cast<size_t>(t)] = crc_more(0, text, n);
            }
        });
        crc_ = crc_more(crc_, base, own_made);
        for(size_t t = 0; t < taken.size(); ++t) {
            crc_ = static_cast<uint32_t>(crc32_combine(crc_, crc[t], static_cast<z_off_t>(chunk[static_cast<size_t>(taken[t])].made)));
        }
        text_ += made;
        // the span's text in pieces: this thread's own, then every chunk taken, where they were made
        pieces_.clear();
        piece_ = 0;
        pieces_.push_back({base, own_made});
        for(const int k : taken) { pieces_.push_back({marked_[static_cast<size_t>(k)].data() + WINDOW, chunk[static_cast<size_t>(k)].made}); }
        ready_ = pieces_[0].first;
        produced_ = ready_ + pieces_[0].second;
        if(!taken.empty()) {
            // the history behind the span: the window behind the last chunk taken
            tail_window_.swap(window[taken.size()]);
            tail_pending_ = true;
            // the stream goes on behind the last chunk taken (a block header, or the trailer)
            raw_.start(origin, limit, end_bit);
            if(member_ended) { end_member(origin + (end_bit >> 3)); }
        } else if(member_ended) {
            end_member(raw_.after());
        } else if(made == 0) { return false; }           // a block the decoder does not take (the one thread path hands it to zlib): nothing was made here
        return true;
    }

    // markers into bytes: text[i] = window[place] wherever flags[i] says so.  (Markers do not die out in FASTQ text: a read's name
    // is a match into the name before it, and what the first name of the chunk took from the window is copied from record to
    // record to the chunk's end — a third of the bytes.)
    static void resolve(uint8_t* text, const uint8_t* flags, size_t n, const uint8_t* w) {
        size_t i = 0;
#ifdef PHQ_CRC_FOLDED
        for(; i + 16 <= n; i += 16) {
            const __m128i sixteen = _mm_loadu_si128(reinterpret_cast<const __m128i*>(flags + i));
            unsigned marked = ~static_cast<unsigned>(_mm_movemask_epi8(_mm_cmpeq_epi8(sixteen, _mm_setzero_si128()))) & 0xffffu;
            while(marked) {
                const size_t j = i + static_cast<size_t>(__builtin_ctz(marked));
                marked &= marked - 1u;
                text[j] = w[static_cast<size_t>(text[j]) | (static_cast<size_t>(flags[j] >> 1) << 8)];
            }
        }
#endif
        for(; i < n; ++i) { if(flags[i]) { text[i] = w[static_cast<size_t>(text[i]) | (static_cast<size_t>(flags[i] >> 1) << 8)]; } }
    }

    static uint32_t le32(const uint8_t* p) { return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) | (static_cast<uint32_t>(p[2]) << 16) | (static_cast<uint32_t>(p[3]) << 24); }
    static uint32_t crc_more(uint32_t crc, const uint8_t* buffer, size_t bytes) {
#ifdef PHQ_CRC_FOLDED
        static const bool folded = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
        if(folded && bytes >= 64) {
            const size_t body = bytes & ~static_cast<size_t>(15);
            crc = fast_inflate::crc32_folded(crc, buffer, body);
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

    const uint8_t* map_ = nullptr;
    size_t size_ = 0;
    int threads_ = 1;
    int span_wait_ = 0, span_backoff_ = 0;
    size_t consumed_ = 0;                          // bytes of the member's deflate stream behind the decoder
    const uint8_t* deflate_ = nullptr;             // the member's deflate stream (behind its header)
    std::unique_ptr<uint8_t[]> span_text_;         // WINDOW of history, then a span's text
    size_t span_text_room_ = 0;
    std::vector<std::vector<uint8_t>> marked_;     // per chunk: its text plane and its flag plane (run_planes<true>)
    std::vector<std::pair<const uint8_t*, size_t>> pieces_;      // a span's text where it was made: this thread's own piece, then every chunk taken
    size_t piece_ = 0;
    std::vector<uint8_t> tail_window_;             // the last 32 KiB of the text behind a span whose last piece is a chunk's
    bool tail_pending_ = false;
    size_t at_ = 0;                                // the member's first byte in the file
    std::unique_ptr<uint8_t[]> work_;              // WINDOW of history, then CHUNK of new text (and the decoder's margin)
    const uint8_t* history_ = nullptr;             // the first byte of the member's text that is still held
    const uint8_t* ready_ = nullptr;               // [ready_, produced_): text not handed out yet
    const uint8_t* produced_ = nullptr;
    fast_inflate::RawStream raw_;
    z_stream zlib_;
    bool zlib_open_ = false;
    const uint8_t* zlib_in_ = nullptr;
    uint32_t crc_ = 0;
    uint64_t text_ = 0;
    bool ended_ = false, next_member_ = false;
};

}  // namespace phq
