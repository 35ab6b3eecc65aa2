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

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "error.hpp"
#include "fast_inflate.hpp"

namespace phq {
namespace fast_inflate {

class RawStream {
public:
    enum class Status { ROOM, END, DAMAGED, REFUSED };
    static constexpr size_t MARGIN = 320;          // room the unchecked loop wants ahead: the longest match, three literals, the copies' overshoot

    void start(const uint8_t* in, const uint8_t* in_end) {
        in_next_ = in; in_end_ = in_end; bits_ = 0; held_ = 0; blocks_ = 0; last_ = false; in_block_ = false; stored_ = false; stored_left_ = 0; finished_ = false;
        tables_ = nullptr;
    }
    bool finished() const { return finished_; }
    static long& refuse_block() { static long block = -1; return block; }      // a test seam: the block header with this ordinal is refused
    // the first byte behind the stream (after END of the last block): whole bytes the bit buffer still holds were not the stream's
    const uint8_t* after() const { return in_next_ - (held_ >> 3); }
    // where the block that was REFUSED starts: the byte that holds its first bit, and how many of that byte's low bits came before it
    const uint8_t* boundary_byte() const { return boundary_in_ - ((boundary_held_ + 7) >> 3); }
    int boundary_bit() const { return (8 - (boundary_held_ & 7)) & 7; }

    // Decodes into [out, out_end); `history` <= out is the first byte a distance may reach (the start of the member's text, or of the
    // 32 KiB kept of it).  ROOM: less than MARGIN bytes are left and the stream goes on; END: the last block has ended.
    Status run(const uint8_t* history, uint8_t*& out_io, uint8_t* const out_end) {
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
                if(held < 3) { return leave(Status::DAMAGED); }
                const uint64_t bits_before = bits;
                auto refuse = [&]() { in_next = boundary_in_; bits = bits_before; held = boundary_held_; return leave(Status::REFUSED); };
                // (a refused header is read again by zlib from boundary_byte(): only in_next_ and the bit count matter to it)
                if(blocks_++ == refuse_block()) { return refuse(); }       // (the check program: the hand over to zlib at a chosen block)
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
                    const uint32_t kind = kind_of(entry);
                    bits >>= entry & 31u; held -= static_cast<int>(entry & 31u);
                    if(kind == LITERAL) { *out++ = static_cast<uint8_t>(entry >> 8); continue; }
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
                        do { uint64_t word; std::memcpy(&word, from, 8); std::memcpy(out, &word, 8); from += 8; out += 8; } while(out < stop);
                    } else if(distance == 1) {
                        std::memset(out, *from, length);
                    } else {
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
                    for(uint32_t k = 0; k <= kind; ++k) { *out++ = static_cast<uint8_t>(entry >> (8 + 8 * k)); }
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
                for(uint32_t k = 0; k < length; ++k) { out[k] = from[k]; }
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

    const uint8_t* in_next_ = nullptr;
    const uint8_t* in_end_ = nullptr;
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

    // true: a regular file that starts with a gzip header
    bool open(const std::string& path) {
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
                if(ended_) { break; }
                produce();
                continue;
            }
            const size_t take = std::min(want - done, static_cast<size_t>(produced_ - ready_));
            std::memcpy(to + done, ready_, take);
            ready_ += take;
            done += take;
        }
        return done;
    }

    static uint64_t& through_zlib() { static uint64_t count = 0; return count; }      // members the decoder handed to zlib part way (diagnosis, the check program)

private:
    static constexpr size_t WINDOW = size_t(1) << 15;
    static constexpr size_t CHUNK = size_t(1) << 20;

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
        raw_.start(in + at, map_ + size_);
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
        // the last 32 KiB of the member's text move to the front
        uint8_t* const base = work_.get() + WINDOW;
        const size_t keep = std::min(static_cast<size_t>(produced_ - history_), WINDOW);
        std::memmove(base - keep, produced_ - keep, keep);
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
            if(trailer + 8 > map_ + size_) { throw IOError("a gzip stream ends early"); }
            if(le32(trailer) != crc_ || le32(trailer + 4) != static_cast<uint32_t>(text_)) { throw IOError("a gzip member does not hold to its CRC or its size"); }
            at_ = static_cast<size_t>(trailer + 8 - map_);
            // another member behind it, or the end (bytes that are no gzip header end the text, as they end gzread's)
            next_member_ = at_ + 18 <= size_ && map_[at_] == 0x1f && map_[at_ + 1] == 0x8b;
            ended_ = !next_member_;
        } else if(made == 0) { throw IOError("damaged gzip data"); }       // no progress and no end
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
