// A gzip input of SEVERAL members — BGZF blocks (bgzip, htslib: what the reference reads FASTQ through, fastq.h:427-440 with its
// thread pool, transcode.cpp:1599-1605) or plain concatenated members (what this library writes: one member per batch part) —
// inflated member parallel on the host's threads, straight into the reader's buffer.
//
// zlib inflates ~0.3 GB/s of FASTQ text a thread and a gzip stream is sequential inside a member, so a single member file
// stays one gzread thread (the caller's old path: open() says no); members are independent deflate streams, each with its own
// CRC and its inflated size in its last four bytes, so with the member boundaries known every member's place in the text is
// known before anything is inflated and the threads write straight to it.
//   * BGZF: every block's header carries its compressed size (extra subfield 'B' 'C'): the index is a walk over headers.
//   * plain members: a scan for gzip headers — 1f 8b 08, no reserved flag bit, XFL 0 / 2 / 4, a known OS byte: by chance once
//     in ~190 GB of deflate data — and every member is held to Z_STREAM_END at exactly the next boundary with exactly the
//     size its trailer states and zlib's own CRC check; anything else is an IO error (the device path then hands the file to
//     the host reader, which reads it through gzread from where the last good round ended).
#pragma once

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
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

class MemberInflater {
public:
    MemberInflater() = default;
    MemberInflater(const MemberInflater&) = delete;
    MemberInflater& operator=(const MemberInflater&) = delete;
    ~MemberInflater() {
        if(map_ != nullptr) { ::munmap(const_cast<uint8_t*>(map_), size_); }
    }

    // true: the file is a gzip stream of at least two members and is indexed; false: something else (one member, not gzip, not
    // a regular file): the caller reads it as before
    bool open(const std::string& path, int threads) {
        threads_ = threads < 1 ? 1 : threads;
        const int descriptor = ::open(path.c_str(), O_RDONLY);
        if(descriptor < 0) { return false; }
        struct stat status;
        if(::fstat(descriptor, &status) != 0 || !S_ISREG(status.st_mode) || status.st_size < 36) { ::close(descriptor); return false; }
        size_ = static_cast<size_t>(status.st_size);
        void* const mapped = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, descriptor, 0);
        ::close(descriptor);
        if(mapped == MAP_FAILED) { return false; }
        map_ = static_cast<const uint8_t*>(mapped);
        (void)::madvise(mapped, size_, MADV_SEQUENTIAL);
        if(!(index_blocks() || index_members()) || member_.size() < 2) {
            ::munmap(mapped, size_);
            map_ = nullptr;
            member_.clear();
            return false;
        }
        return true;
    }

    bool blocked() const { return blocked_; }                 // BGZF
    size_t members() const { return member_.size(); }

    // up to `want` bytes of the text, in order; 0: the end.  Throws IOError for a member that does not inflate to what it states.
    // The members that fit `to` whole are inflated STRAIGHT INTO IT, side by side on every thread (their places in the text are
    // known from their trailers before anything is inflated); the one member that straddles the end of `to` goes to a buffer of
    // its own and is handed out in two pieces.  (A first version inflated batches ahead into buffers of their own and copied them
    // out: the copy — 190 MB a round of 1 M reads, through the same memory system the inflating threads use — and the fresh
    // pages of every batch were as much of a BGZF run as inflating itself: 19.5 -> 28 Mreads/s with the buffers reused, the rest
    // with the copy gone.)
    size_t read(uint8_t* to, size_t want) {
        size_t done = 0;
        auto from_carry = [&]() {
            const size_t take = std::min(want - done, carry_size_ - carry_at_);
            std::memcpy(to + done, carry_.get() + carry_at_, take);
            done += take;
            carry_at_ += take;
        };
        if(carry_at_ < carry_size_) { from_carry(); }
        while(done < want && next_ < member_.size()) {
            // the run of whole members that fits
            const size_t first = next_;
            size_t text = 0;
            while(next_ < member_.size() && text + member_[next_].text <= want - done) { text += member_[next_++].text; }
            if(next_ > first) {
                const size_t last = next_;
                uint8_t* const out = to + done;
                if(last - first == 1) {
                    inflate_member(map_ + member_[first].begin, member_[first].end - member_[first].begin, out, member_[first].text);
                } else {
                    std::vector<size_t> place(last - first);
                    size_t at = 0;
                    for(size_t k = first; k < last; ++k) { place[k - first] = at; at += member_[k].text; }
                    std::atomic<size_t> turn{first};
                    const int workers = static_cast<int>(std::min<size_t>(static_cast<size_t>(threads_), (last - first + 3) / 4));
                    parallel_ranges(workers, 1, workers, [&](int64_t, int64_t, int) {
                        for(size_t k = turn.fetch_add(4); k < last; k = turn.fetch_add(4)) {               // four members a turn
                            for(size_t m = k; m < std::min(k + 4, last); ++m) {
                                inflate_member(map_ + member_[m].begin, member_[m].end - member_[m].begin, out + place[m - first], member_[m].text);
                            }
                        }
                    });
                }
                done += text;
                continue;
            }
            // the next member is larger than what is left of `to`: through the carry buffer
            const Member& member = member_[next_++];
            if(carry_capacity_ < member.text) { carry_.reset(new uint8_t[member.text]); carry_capacity_ = member.text; }
            inflate_member(map_ + member.begin, member.end - member.begin, carry_.get(), member.text);
            carry_size_ = member.text;
            carry_at_ = 0;
            from_carry();
        }
        return done;
    }

private:
    struct Member { size_t begin, end, text; };
    static constexpr size_t MEMBER_LIMIT = size_t(64) << 20;      // plain members: compressed or stated text bytes

    static uint32_t le32(const uint8_t* p) { return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) | (static_cast<uint32_t>(p[2]) << 16) | (static_cast<uint32_t>(p[3]) << 24); }

    // BGZF (the SAM specification, section 4.1): FLG.FEXTRA, XLEN, subfield 'B' 'C' of two bytes = the block's size - 1
    bool index_blocks() {
        size_t at = 0;
        std::vector<Member> found;
        while(at + 28 <= size_) {
            const uint8_t* const h = map_ + at;
            if(h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 4) == 0) { return false; }
            const size_t extra = static_cast<size_t>(h[10]) | (static_cast<size_t>(h[11]) << 8);
            if(at + 12 + extra > size_) { return false; }
            size_t block = 0;
            for(size_t e = 0; e + 4 <= extra;) {
                const uint8_t* const sub = h + 12 + e;
                const size_t length = static_cast<size_t>(sub[2]) | (static_cast<size_t>(sub[3]) << 8);
                if(sub[0] == 'B' && sub[1] == 'C' && length == 2 && e + 6 <= extra) { block = (static_cast<size_t>(sub[4]) | (static_cast<size_t>(sub[5]) << 8)) + 1; }
                e += 4 + length;
            }
            if(block < 26 || at + block > size_) { return false; }
            const size_t text = le32(map_ + at + block - 4);
            if(text > 65536) { return false; }
            if(text > 0) { found.push_back({at, at + block, text}); }             // (the empty block at the end says end of file)
            at += block;
        }
        if(at != size_) { return false; }
        member_.swap(found);
        blocked_ = true;
        return true;
    }

    static bool header_at(const uint8_t* h) {
        return h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 0xe0) == 0 && (h[8] == 0 || h[8] == 2 || h[8] == 4) && (h[9] <= 13 || h[9] == 255);
    }

    // plain members: where a gzip header starts, scanned on every thread
    bool index_members() {
        if(!header_at(map_)) { return false; }
        const size_t last = size_ - 18;                       // a header and a trailer fit behind it
        const int parts = static_cast<int>(std::min<size_t>(static_cast<size_t>(threads_), (size_ >> 22) + 1));
        std::vector<std::vector<size_t>> found(static_cast<size_t>(parts));
        parallel_ranges(static_cast<int64_t>(last), 1 << 20, parts, [&](int64_t first, int64_t until, int part) {
            const uint8_t* at = map_ + (first == 0 ? 1 : first);
            const uint8_t* const end = map_ + until;
            while(at < end) {
                at = static_cast<const uint8_t*>(std::memchr(at, 0x1f, static_cast<size_t>(end - at)));
                if(at == nullptr) { break; }
                if(header_at(at)) { found[static_cast<size_t>(part)].push_back(static_cast<size_t>(at - map_)); }
                ++at;
            }
        });
        std::vector<size_t> boundary{0};
        for(const auto& part : found) { boundary.insert(boundary.end(), part.begin(), part.end()); }
        boundary.push_back(size_);
        for(size_t k = 0; k + 1 < boundary.size(); ++k) {
            if(boundary[k + 1] - boundary[k] < 18) { return false; }
            // A trailer states a member's text size modulo 2^32, and a member that does not fit a read goes through a buffer
            // of its own size: members past MEMBER_LIMIT bytes either way (two lanes' files joined with cat, each gigabytes) are
            // not this reader's shape — open() says no and the caller streams the file (a member of less than 4 MiB cannot
            // inflate to 4 GiB, deflate expands at most 1 032 times; a larger one whose trailer wrapped fails its size check in
            // inflate_member and the caller falls back as well).
            const size_t text = le32(map_ + boundary[k + 1] - 4);
            if(boundary[k + 1] - boundary[k] > MEMBER_LIMIT || text > MEMBER_LIMIT) { member_.clear(); return false; }
            member_.push_back({boundary[k], boundary[k + 1], text});
        }
        return true;
    }

public:
    // how many members each decoder took (diagnosis, the check program)
    static std::atomic<uint64_t>& fast_members() { static std::atomic<uint64_t> count{0}; return count; }
    static std::atomic<uint64_t>& zlib_members() { static std::atomic<uint64_t> count{0}; return count; }

private:
    static void inflate_member(const uint8_t* from, size_t bytes, uint8_t* to, size_t text) {
        // the one shot decoder first (fast_inflate.hpp: 2-3x zlib's rate, held to the member's CRC and size); whatever it is not
        // sure of goes through zlib's inflate(), which gives the text or the error
        static const bool only_zlib = std::getenv("PHQ_GZIP_ZLIB") != nullptr;
        if(!only_zlib && fast_inflate::inflate_member(from, bytes, to, text)) { ++fast_members(); return; }
        ++zlib_members();
        z_stream z;
        std::memset(&z, 0, sizeof(z));
        if(inflateInit2(&z, 16 + MAX_WBITS) != Z_OK) { throw InternalError("inflateInit2"); }
        size_t in_at = 0, out_at = 0;
        int status = Z_OK;
        while(status == Z_OK) {
            z.next_in = const_cast<Bytef*>(from + in_at);
            z.avail_in = static_cast<uInt>(std::min<size_t>(bytes - in_at, size_t(1) << 30));
            z.next_out = to + out_at;
            z.avail_out = static_cast<uInt>(std::min<size_t>(text - out_at, size_t(1) << 30));
            const size_t in_before = z.avail_in, out_before = z.avail_out;
            status = inflate(&z, Z_NO_FLUSH);
            in_at += in_before - z.avail_in;
            out_at += out_before - z.avail_out;
            if(status == Z_OK && in_before == z.avail_in && out_before == z.avail_out) { break; }      // no room and no end: more text than stated
        }
        inflateEnd(&z);
        if(status != Z_STREAM_END || in_at != bytes || out_at != text) { throw IOError("a gzip member does not inflate to the size its trailer states"); }
    }

    const uint8_t* map_ = nullptr;
    size_t size_ = 0;
    int threads_ = 1;
    bool blocked_ = false;
    std::vector<Member> member_;
    size_t next_ = 0;
    std::unique_ptr<uint8_t[]> carry_;         // the text of the one member that straddled the end of a read
    size_t carry_capacity_ = 0, carry_size_ = 0, carry_at_ = 0;
};

}  // namespace phq
