// Transcoder: the steps either side of the classification path for FASTQ in / FASTQ or SAM text out
// (SURVEY.md §8f rows f1 and f2), host side C++.
//
//   f1  FASTQ -> batch      FastqRecord::decode (fastq.h:55-78: ASCII -> BAM code, quality - phred offset),
//                           Illumina comment filter flag (fastq.h:253-285), Read::validate name sync
//                           (read.h:253-264); one reader thread per input feed, a path stated k times in `input`
//                           being one interleaved feed of resolution k (transcode.cpp:693-701); then
//                           DecoderSet::stage_reads.
//   f2  results -> records  the per type accumulators and barcode strings the decoder wrappers leave on the
//                           output Read (pamld.cpp:133-180, mdd.cpp:96-138, naive.h:40-45, read.h:269-348),
//                           TemplateRule::apply (transform.h:190-226), Read::flush (read.h:187-237),
//                           Auxiliary::encode tag order (auxiliary.cpp:327-354), SAM text as htslib prints it,
//                           FASTQ records (fastq.h:144-194), @HD / @RG / @PG header lines (atom.cpp:454-472,906-965),
//                           Multiplexer::push: the channel of the multiplexing decoder's barcode decides the
//                           output feeds (multiplex.h:198-240), optionally without QC failed reads.
//
// Out of scope here: CRAM (htslib), bz2 / xz.  Unaligned BAM out: bam_record.hpp.
#include "transcoder.hpp"
#include "fastq_device.hpp"
#include "record_device.hpp"

#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "gzip_members.hpp"
#include "gzip_stream.hpp"
#include "fast_deflate.hpp"
#include "bam_record.hpp"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <type_traits>

#include "error.hpp"
#include "host_parallel.hpp"
#include "text_format.hpp"

namespace phq {

namespace {

// Work a call does not wait for (unmapping input files, giving batch buffers back: page table work in the kernel) runs on ONE
// helper thread owned by this object, which its destructor drains and joins — at process exit or when the library is
// unloaded nothing of it is left running library code (detached threads could be, and made leak checks flaky).
class Afterwards {
public:
    static Afterwards& instance() { static Afterwards one; return one; }
    void run(std::function<void()> task) {
        try {
            std::unique_lock<std::mutex> lock(mutex_);
            if(!worker_.joinable()) { worker_ = std::thread([this]() { loop(); }); }
            queue_.push_back(std::move(task));
            lock.unlock();
            wake_.notify_one();
        } catch(...) {
            task();                              // no thread to be had: done here and now
        }
    }
    ~Afterwards() {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            stop_ = true;
        }
        wake_.notify_one();
        if(worker_.joinable()) { worker_.join(); }
    }
private:
    void loop() {
        std::unique_lock<std::mutex> lock(mutex_);
        for(;;) {
            wake_.wait(lock, [this]() { return stop_ || !queue_.empty(); });
            if(queue_.empty()) { return; }       // stopping, and drained
            std::function<void()> task = std::move(queue_.front());
            queue_.pop_front();
            lock.unlock();
            task();
            lock.lock();
        }
    }
    std::mutex mutex_;
    std::condition_variable wake_;
    std::deque<std::function<void()>> queue_;
    std::thread worker_;
    bool stop_ = false;
};

// std::vector whose resize() leaves trivially constructible elements uninitialised: the batch arrays below are
// written in full by the parallel passes right after they are sized, and a value initialising resize would first
// touch (and zero) hundreds of megabytes on the one thread that calls it
template <typename T, typename A = std::allocator<T>>
class default_init_allocator : public A {
    typedef std::allocator_traits<A> traits;
public:
    template <typename U> struct rebind { typedef default_init_allocator<U, typename traits::template rebind_alloc<U>> other; };
    using A::A;
    // large blocks on 2 MiB boundaries and marked for transparent huge pages: a fresh batch array of a hundred
    // megabytes then takes fifty page faults instead of twenty five thousand
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if(bytes >= (size_t(4) << 20)) {
            const size_t huge = size_t(2) << 20;
            void* block = nullptr;
            if(::posix_memalign(&block, huge, (bytes + huge - 1) / huge * huge) != 0) { throw std::bad_alloc(); }
            ::madvise(block, (bytes + huge - 1) / huge * huge, MADV_HUGEPAGE);
            return static_cast<T*>(block);
        }
        void* block = std::malloc(bytes ? bytes : 1);
        if(block == nullptr) { throw std::bad_alloc(); }
        return static_cast<T*>(block);
    }
    void deallocate(T* pointer, size_t) noexcept { std::free(pointer); }
    template <typename U> void construct(U* pointer) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new(static_cast<void*>(pointer)) U; }
    template <typename U, typename... Args> void construct(U* pointer, Args&&... args) { traits::construct(static_cast<A&>(*this), pointer, std::forward<Args>(args)...); }
};
template <typename T> using Bare = std::vector<T, default_init_allocator<T>>;
struct Line { size_t first; uint32_t second; };          // where a line starts in the stream's bytes, and its length

// one input file, plain or gzip (gzread reads both), 4 line FASTQ records
//
// A batch is read in two steps: the reader thread pulls bytes and indexes line starts until it holds the lines of
// `capacity` records (memchr speed), then the records are decoded — name, Illumina filter flag, ASCII -> BAM codes,
// quality - offset — on a few threads over contiguous record ranges (two passes: sizes, then a prefix sum places
// every thread's output).  The reference decodes under the feed lock, one record at a time (feed.h:320-333).
class FastqStream {
public:
    // A plain (not gzip) regular file is mapped: its bytes are read where they lie, a batch's lines are located on
    // several threads straight in the mapping, and records may keep pointing into it for the whole run (SegmentBatch
    // borrows the printable text instead of copying it).  Anything else goes through gzread as before.
    explicit FastqStream(const std::string& path) : path_(path), file_(nullptr), map_(nullptr), map_size_(0), begin_(0), end_(0), eof_(false) {
        const int descriptor = ::open(path.c_str(), O_RDONLY);
        if(descriptor < 0) { throw IOError("could not open " + path + " for reading"); }
        struct stat status;
        unsigned char magic[2] = {0, 0};
        const bool regular = ::fstat(descriptor, &status) == 0 && S_ISREG(status.st_mode);
        const bool zipped = regular && ::pread(descriptor, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if(regular && !zipped && status.st_size > 0 && std::getenv("PHQ_NO_MMAP") == nullptr) {
            void* mapped = ::mmap(nullptr, static_cast<size_t>(status.st_size), PROT_READ, MAP_PRIVATE, descriptor, 0);
            if(mapped != MAP_FAILED) {
                map_ = static_cast<const char*>(mapped);
                map_size_ = static_cast<size_t>(status.st_size);
                ::madvise(mapped, map_size_, MADV_SEQUENTIAL);
                end_ = map_size_;
                eof_ = true;
            }
        }
        ::close(descriptor);
        if(map_ == nullptr) {
            file_ = gzopen(path.c_str(), "rb");
            if(file_ == nullptr) { throw IOError("could not open " + path + " for reading"); }
            gzbuffer(file_, 1 << 20);
            buffer_.resize(1 << 24);
        }
    }
    ~FastqStream() {
        if(file_) { gzclose(file_); }
        // unmapping hundreds of megabytes is page table work in the kernel (26 ms after 4 M reads: a fifth of a report only
        // run of that size): it is left to the helper thread, the call does not wait for it
        if(map_) {
            char* const address = const_cast<char*>(map_);
            const size_t size = map_size_;
            Afterwards::instance().run([address, size]() { ::munmap(address, size); });
        }
    }
    bool mapped() const { return map_ != nullptr; }
    FastqStream(const FastqStream&) = delete;
    FastqStream& operator=(const FastqStream&) = delete;

    // the next up to `capacity` records: line[4 i + k] = (offset, size) of line k of record i inside data().
    // Line ends are found on `threads` threads over slices of the buffered bytes (a memchr per 40 byte line costs
    // more in call overhead than in scanning), about as many bytes at a time as the lines still wanted should take.
    int64_t index_records(int64_t capacity, int32_t threads, Bare<Line>& line) {
        line.clear();
        size_t scan = begin_;                       // first byte that belongs to no line yet
        size_t searched = begin_;                   // bytes before this hold no unseen line end
        const size_t wanted = static_cast<size_t>(capacity) * 4;
        while(line.size() < wanted) {
            if(searched >= end_) {
                if(eof_) {
                    if(scan < end_) {                                          // last line without a newline
                        push(line, scan, end_ - scan);
                        scan = end_;
                    }
                    break;
                }
                // more bytes: keep what is not consumed yet at the front (offsets already taken stay valid: they are
                // relative to begin_, rebased below)
                const size_t shift = begin_;
                refill();
                scan -= shift;
                searched -= shift;
                for(auto& l : line) { l.first -= shift; }
                continue;
            }
            const size_t missing = wanted - line.size();
            const size_t guess = static_cast<size_t>(static_cast<double>(missing) * bytes_per_line_ * 1.02) + 4096;
            const size_t until = std::min(end_, searched + guess);
            find_line_ends(searched, until, threads);
            // the lines between consecutive newlines, filled in place on several threads; a blank line (tolerated
            // between records) or a carriage return sends the window through the serial loop below instead
            const size_t take = std::min(line_end_.size(), wanted - line.size());
            bool filled = false;
            if(take >= (1u << 16)) {
                const size_t base = line.size();
                line.resize(base + take);
                std::vector<char> plain(static_cast<size_t>(std::max(1, threads)), 1);
                const char* const bytes_at = data();
                const size_t first_start = scan;
                parallel_ranges(static_cast<int64_t>(take), 1 << 16, threads, [&](int64_t from, int64_t to, int part) {
                    for(int64_t k = from; k < to; ++k) {
                        const size_t start = k == 0 ? first_start : line_end_[static_cast<size_t>(k) - 1] + 1;
                        const size_t stop = line_end_[static_cast<size_t>(k)];
                        if(stop == start || bytes_at[stop - 1] == '\r') { plain[static_cast<size_t>(part)] = 0; }
                        line[base + static_cast<size_t>(k)] = Line{start, static_cast<uint32_t>(stop - start)};
                    }
                });
                filled = std::find(plain.begin(), plain.end(), 0) == plain.end();
                if(filled) { scan = line_end_[take - 1] + 1; }
                else { line.resize(base); }
            }
            if(!filled) {
                for(const size_t newline : line_end_) {
                    if(line.size() >= wanted) { break; }
                    push(line, scan, newline - scan);
                    scan = newline + 1;
                }
            }
            if(line.size() >= wanted) { break; }
            searched = until;
        }
        const size_t records = line.size() / 4;
        if(line.size() % 4 != 0) {
            if(eof_ && scan == end_) { throw SequenceError("truncated FASTQ record in " + path_); }
            line.resize(records * 4);
        }
        consumed_ = records > 0 ? line[records * 4 - 1].first + line[records * 4 - 1].second + 1 : scan;
        if(records * 4 == line.size() && line.size() < wanted) { consumed_ = scan; }      // end of input: trailing blank lines too
        if(consumed_ > end_) { consumed_ = end_; }
        if(records > 0 && consumed_ > begin_) { bytes_per_line_ = static_cast<double>(consumed_ - begin_) / static_cast<double>(records * 4); }
        return static_cast<int64_t>(records);
    }
    void release() { begin_ = consumed_; }                 // the indexed records have been decoded
    // the records before byte `offset` of the (uncompressed) file have been taken elsewhere: the device path, which handed
    // the rest of the run over (Transcoder::run_on_device).  Called before the first index_records.
    void start_at(size_t offset) {
        if(map_ != nullptr) { begin_ = consumed_ = std::min(offset, map_size_); return; }
        if(gzseek(file_, static_cast<z_off_t>(offset), SEEK_SET) < 0) { throw IOError("error reading " + path_); }
    }
    const char* data() const { return map_ ? map_ : buffer_.data(); }
    const std::string& path() const { return path_; }

private:
    // offsets of every '\n' in [first, last), in order, into line_end_
    void find_line_ends(size_t first, size_t last, int32_t threads) {
        line_end_.clear();
        const int64_t bytes = static_cast<int64_t>(last - first);
        static const int index_threads = []() { const char* text = std::getenv("PHQ_INDEX_THREADS"); return text ? std::max(1, std::atoi(text)) : 0; }();      // measurement aid
        const int parts = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(index_threads > 0 ? index_threads : threads, bytes >> 20)));
        if(static_cast<int>(part_end_.size()) < parts) { part_end_.resize(static_cast<size_t>(parts)); }
        const char* const bytes_at = data();
        parallel_ranges(bytes, 1 << 20, parts, [&](int64_t from, int64_t to, int part) {
            Bare<size_t>& found = part_end_[static_cast<size_t>(part)];
            found.clear();
            const char* at = bytes_at + first + from;
            const char* const stop = bytes_at + first + to;
            // room for a line end every 8 bytes, grown when a stretch of shorter lines needs it
            found.resize(static_cast<size_t>(stop - at) / 8 + 64);
            size_t count = 0;
            if(have_avx2) { count = scan_newlines_avx2(at, stop, bytes_at, found); }
            else {
                while(at < stop) {
                    const void* newline = std::memchr(at, '\n', static_cast<size_t>(stop - at));
                    if(newline == nullptr) { break; }
                    if(count == found.size()) { found.resize(found.size() * 2); }
                    found[count++] = static_cast<size_t>(static_cast<const char*>(newline) - bytes_at);
                    at = static_cast<const char*>(newline) + 1;
                }
            }
            found.resize(count);
        });
        for(int part = 0; part < parts; ++part) { line_end_.insert(line_end_.end(), part_end_[static_cast<size_t>(part)].begin(), part_end_[static_cast<size_t>(part)].end()); }
    }
    // every '\n' of [at, stop): 32 bytes per compare (a memchr per 40 byte line spends its time in call overhead)
    __attribute__((target("avx2"))) static size_t scan_newlines_avx2(const char* at, const char* stop, const char* origin, Bare<size_t>& found) {
        size_t count = 0;
        const __m256i newline = _mm256_set1_epi8('\n');
        while(at + 32 <= stop) {
            uint32_t mask = static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(at)), newline)));
            if(mask != 0) {
                if(count + 32 > found.size()) { found.resize(found.size() * 2); }
                const size_t base = static_cast<size_t>(at - origin);
                while(mask != 0) {
                    found[count++] = base + static_cast<size_t>(__builtin_ctz(mask));
                    mask &= mask - 1;
                }
            }
            at += 32;
        }
        for(; at < stop; ++at) {
            if(*at == '\n') {
                if(count == found.size()) { found.resize(found.size() * 2); }
                found[count++] = static_cast<size_t>(at - origin);
            }
        }
        return count;
    }
    void push(Bare<Line>& line, size_t offset, size_t size) {
        if(size > 0 && data()[offset + size - 1] == '\r') { --size; }
        if(size == 0 && line.size() % 4 == 0) { return; }                      // blank line between records
        line.push_back(Line{offset, static_cast<uint32_t>(size)});
    }
    void refill() {
        if(begin_ > 0) {
            std::memmove(buffer_.data(), buffer_.data() + begin_, end_ - begin_);
            end_ -= begin_;
            begin_ = 0;
        }
        if(end_ == buffer_.size()) { buffer_.resize(buffer_.size() * 2); }
        const int got = gzread(file_, buffer_.data() + end_, static_cast<unsigned>(std::min<size_t>(buffer_.size() - end_, 1u << 30)));
        if(got < 0) { throw IOError("error reading " + path_); }
        if(got == 0) { eof_ = true; }
        end_ += static_cast<size_t>(got);
    }
    std::string path_;
    gzFile file_;
    const char* map_;
    size_t map_size_;
    std::vector<char> buffer_;
    size_t begin_;
    size_t end_;
    size_t consumed_ = 0;
    bool eof_;
    double bytes_per_line_ = 64;                     // of the previous batch: how far to look for the lines of the next
    Bare<size_t> line_end_;
    std::vector<Bare<size_t>> part_end_;
    const bool have_avx2 = __builtin_cpu_supports("avx2") != 0;
};

struct SegmentBatch {
    Bare<uint8_t> code;           // BAM codes / Phred values: only for feeds a decoder token reads (stage_reads)
    Bare<uint8_t> quality;
    Bare<char> text;              // the same bases as the output prints them (BamToAmbiguousAscii of the code) and the
    Bare<char> quality_text;      // qualities at the output Phred offset: only for feeds the template reads
    Bare<int64_t> offset;
    Bare<int32_t> length;
    Bare<uint8_t> qcfail;
    Bare<char> name;              // concatenated, name_offset[i] .. name_offset[i + 1]
    Bare<int64_t> name_offset;
    Bare<Line> line;
    // a feed read from a mapped file lends its bytes instead: record i's bases and qualities as the file holds them
    // start at source + source_sequence[i] / source_quality[i]; the printable form (BamToAmbiguousAscii of the code, the
    // output Phred offset) is made while the record is written
    const char* source = nullptr;
    Bare<uint64_t> source_sequence;
    Bare<uint64_t> source_quality;
    int32_t quality_shift = 0;           // output Phred offset - input Phred offset
    std::string error;
    int error_code = 0;                  // its class (error.hpp): a read error stays an IO error
    void clear() {
        code.clear(); quality.clear(); text.clear(); quality_text.clear(); offset.clear(); length.clear(); qcfail.clear(); name.clear();
        name_offset.assign(1, 0);
        source = nullptr; source_sequence.clear(); source_quality.clear();
        error.clear();
        error_code = 0;
    }
    int64_t size() const { return static_cast<int64_t>(length.size()); }
};

// what a feed's batch has to hold: codes for the decoders' tokens, printable text for the template's
struct FeedNeeds { bool codes; bool text; int32_t output_phred_offset; bool illumina_comment; };

void read_segment_batch(FastqStream& stream, int64_t capacity, int32_t phred_offset, int32_t threads, FeedNeeds needs, SegmentBatch& out) {
    static const std::vector<uint8_t> ascii_table = []() {                  // AsciiToAmbiguousBam, iupac.h:153-171
        std::vector<uint8_t> table(256);
        for(int c = 0; c < 256; ++c) { table[static_cast<size_t>(c)] = ascii_to_bam(static_cast<uint8_t>(c)); }
        return table;
    }();
    // ... followed by BamToAmbiguousAscii (iupac.h:62-79): the letter the reference prints for what it read
    static const std::vector<char> canonical_table = []() {
        std::vector<char> table(256);
        for(int c = 0; c < 256; ++c) { table[static_cast<size_t>(c)] = bam_to_ascii(ascii_to_bam(static_cast<uint8_t>(c))); }
        return table;
    }();
    out.clear();
    try {
        const int64_t n = stream.index_records(capacity, threads, out.line);
        const char* const data = stream.data();
        const auto& line = out.line;
        // name up to the first whitespace (kseq), comment after it
        auto name_end_of = [&](int64_t i) {
            const char* header = data + line[static_cast<size_t>(4 * i)].first;
            const size_t size = line[static_cast<size_t>(4 * i)].second;
            size_t name_end = 1;
            while(name_end < size && header[name_end] != ' ' && header[name_end] != '\t') { ++name_end; }
            return name_end;
        };
        // pass 1: validate, sizes per thread range
        const int parts = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(threads, (n + (1 << 14) - 1) >> 14)));
        std::vector<int64_t> sequence_bytes(static_cast<size_t>(parts) + 1, 0), name_bytes(static_cast<size_t>(parts) + 1, 0);
        out.offset.resize(static_cast<size_t>(n));
        out.length.resize(static_cast<size_t>(n));
        out.qcfail.resize(static_cast<size_t>(n));
        out.name_offset.resize(static_cast<size_t>(n) + 1);
        parallel_ranges(n, 1 << 14, parts, [&](int64_t first, int64_t last, int part) {
            int64_t bases = 0;
            int64_t names = 0;
            for(int64_t i = first; i < last; ++i) {
                const auto& header = line[static_cast<size_t>(4 * i)];
                const auto& sequence = line[static_cast<size_t>(4 * i + 1)];
                const auto& separator = line[static_cast<size_t>(4 * i + 2)];
                const auto& quality = line[static_cast<size_t>(4 * i + 3)];
                if(header.second == 0 || data[header.first] != '@') { throw SequenceError("malformed FASTQ record in " + stream.path()); }
                if(separator.second == 0 || data[separator.first] != '+') { throw SequenceError("malformed FASTQ separator in " + stream.path()); }
                if(quality.second != sequence.second) { throw SequenceError("sequence and quality differ in length in " + stream.path()); }
                out.length[static_cast<size_t>(i)] = static_cast<int32_t>(sequence.second);
                bases += sequence.second;
                names += static_cast<int64_t>(name_end_of(i)) - 1;
            }
            sequence_bytes[static_cast<size_t>(part) + 1] = bases;
            name_bytes[static_cast<size_t>(part) + 1] = names;
        });
        for(int part = 0; part < parts; ++part) {
            sequence_bytes[static_cast<size_t>(part) + 1] += sequence_bytes[static_cast<size_t>(part)];
            name_bytes[static_cast<size_t>(part) + 1] += name_bytes[static_cast<size_t>(part)];
        }
        const size_t total_bases = static_cast<size_t>(sequence_bytes[static_cast<size_t>(parts)]);
        const bool lend = needs.text && stream.mapped();
        if(needs.codes) { out.code.resize(total_bases); out.quality.resize(total_bases); }
        if(needs.text && !lend) { out.text.resize(total_bases); out.quality_text.resize(total_bases); }
        if(lend) {
            out.source = data;
            out.source_sequence.resize(static_cast<size_t>(n));
            out.source_quality.resize(static_cast<size_t>(n));
            out.quality_shift = needs.output_phred_offset - phred_offset;
        }
        out.name.resize(static_cast<size_t>(name_bytes[static_cast<size_t>(parts)]));
        out.name_offset[static_cast<size_t>(n)] = name_bytes[static_cast<size_t>(parts)];
        // pass 2: decode into place
        parallel_ranges(n, 1 << 14, parts, [&](int64_t first, int64_t last, int part) {
            int64_t at = sequence_bytes[static_cast<size_t>(part)];
            int64_t name_at = name_bytes[static_cast<size_t>(part)];
            for(int64_t i = first; i < last; ++i) {
                const char* header = data + line[static_cast<size_t>(4 * i)].first;
                const size_t header_size = line[static_cast<size_t>(4 * i)].second;
                const size_t name_end = name_end_of(i);
                std::memcpy(out.name.data() + name_at, header + 1, name_end - 1);
                out.name_offset[static_cast<size_t>(i)] = name_at;
                name_at += static_cast<int64_t>(name_end) - 1;
                // Illumina comment `segment:filtered:control:barcode` (fastq.h:219-285): filtered == 'Y' sets QC fail
                // ... on the platforms whose comment has that form only (ILLUMINA, ELEMENT: fastq.h:100-126); any other
                // platform leaves the flag alone
                bool qcfail = false;
                if(needs.illumina_comment) {
                    size_t p = name_end < header_size ? name_end + 1 : header_size;
                    while(p < header_size && header[p] != ':') { ++p; }
                    if(p + 1 < header_size) { qcfail = header[p + 1] == 'Y'; }
                }
                out.qcfail[static_cast<size_t>(i)] = qcfail ? 1 : 0;
                const size_t size = line[static_cast<size_t>(4 * i + 1)].second;
                const char* sequence = data + line[static_cast<size_t>(4 * i + 1)].first;
                const char* quality = data + line[static_cast<size_t>(4 * i + 3)].first;
                out.offset[static_cast<size_t>(i)] = at;
                if(needs.codes) {
                    uint8_t* to_code = out.code.data() + at;
                    uint8_t* to_quality = out.quality.data() + at;
                    for(size_t k = 0; k < size; ++k) { to_code[k] = ascii_table[static_cast<uint8_t>(sequence[k])]; }
                    for(size_t k = 0; k < size; ++k) { to_quality[k] = static_cast<uint8_t>(quality[k] - phred_offset); }
                }
                if(lend) {
                    out.source_sequence[static_cast<size_t>(i)] = line[static_cast<size_t>(4 * i + 1)].first;
                    out.source_quality[static_cast<size_t>(i)] = line[static_cast<size_t>(4 * i + 3)].first;
                } else if(needs.text) {
                    // what FastqRecord::decode then ::encode make of the record (fastq.h:55-78,144-171), in one step
                    char* to_text = out.text.data() + at;
                    char* to_quality = out.quality_text.data() + at;
                    for(size_t k = 0; k < size; ++k) { to_text[k] = canonical_table[static_cast<uint8_t>(sequence[k])]; }
                    if(needs.output_phred_offset == phred_offset) { std::memcpy(to_quality, quality, size); }
                    else { for(size_t k = 0; k < size; ++k) { to_quality[k] = static_cast<char>(static_cast<uint8_t>(quality[k] - phred_offset) + needs.output_phred_offset); } }
                }
                at += static_cast<int64_t>(size);
            }
        });
        stream.release();
    } catch(const std::exception& e) {
        const Error* const classified = dynamic_cast<const Error*>(&e);
        out.error = classified ? classified->message : std::string(e.what());
        out.error_code = classified ? static_cast<int>(classified->code) : static_cast<int>(SEQUENCE_ERROR);
        if(out.error.empty()) { out.error = "error reading " + stream.path(); out.error_code = IO_ERROR; }
        out.code.clear(); out.quality.clear(); out.text.clear(); out.quality_text.clear(); out.offset.clear(); out.length.clear(); out.qcfail.clear(); out.name.clear();
        out.name_offset.assign(1, 0);
    }
}

// a barcode accumulator of the output read (raw or corrected, per type): at most the nucleotide width of the job
struct Text {
    std::vector<uint8_t> code;
    std::vector<uint8_t> quality;
    int32_t size = 0;
    void allocate(int32_t capacity) { code.resize(static_cast<size_t>(capacity)); quality.resize(static_cast<size_t>(capacity)); }
    void clear() { size = 0; }
    bool empty() const { return size == 0; }
    void append(const uint8_t* c, const uint8_t* q, int32_t n) {
        std::memcpy(code.data() + size, c, static_cast<size_t>(n));
        std::memcpy(quality.data() + size, q, static_cast<size_t>(n));
        size += n;
    }
    void push(uint8_t c, uint8_t q) { code[static_cast<size_t>(size)] = c; quality[static_cast<size_t>(size)] = q; ++size; }
};

const char BAM_TO_ASCII[] = "=ACMGRSVTWYHKDBN";                              // BamToAmbiguousAscii, iupac.h:62-79
inline char* put_sequence(char* to, const uint8_t* code, int32_t n) {
    for(int32_t i = 0; i < n; ++i) { to[i] = BAM_TO_ASCII[code[i] & 0xf]; }
    return to + n;
}
inline char* put_quality(char* to, const uint8_t* quality, int32_t n, int32_t phred_offset) {
    for(int32_t i = 0; i < n; ++i) { to[i] = static_cast<char>(quality[i] + phred_offset); }
    return to + n;
}
inline char* put_sequence(char* to, const Text& text) { return put_sequence(to, text.code.data(), text.size); }
inline char* put_quality(char* to, const Text& text, int32_t phred_offset) { return put_quality(to, text.quality.data(), text.size, phred_offset); }

// one input segment of the batch: record i of the segment is record i * stride + slot of its feed's batch
struct SegmentView {
    const SegmentBatch* feed = nullptr;
    int32_t stride = 1;
    int32_t slot = 0;
    std::vector<int64_t> offset_copy;       // contiguous copies for stage_reads, interleaved feeds only
    std::vector<int32_t> length_copy;
    std::vector<uint8_t> qcfail_copy;
    int64_t size() const { return feed->size() / stride; }
    size_t at(int64_t i) const { return static_cast<size_t>(i) * static_cast<size_t>(stride) + static_cast<size_t>(slot); }
    void bind(const SegmentBatch& batch, int32_t resolution, int32_t position) {
        feed = &batch; stride = resolution; slot = position;
        offset_copy.clear(); length_copy.clear(); qcfail_copy.clear();
        if(stride > 1) {
            const int64_t n = size();
            offset_copy.resize(static_cast<size_t>(n)); length_copy.resize(static_cast<size_t>(n)); qcfail_copy.resize(static_cast<size_t>(n));
            for(int64_t i = 0; i < n; ++i) {
                offset_copy[static_cast<size_t>(i)] = batch.offset[at(i)];
                length_copy[static_cast<size_t>(i)] = batch.length[at(i)];
                qcfail_copy[static_cast<size_t>(i)] = batch.qcfail[at(i)];
            }
        }
    }
    const int64_t* offsets() const { return stride > 1 ? offset_copy.data() : feed->offset.data(); }
    const int32_t* lengths() const { return stride > 1 ? length_copy.data() : feed->length.data(); }
    const uint8_t* qcfails() const { return stride > 1 ? qcfail_copy.data() : feed->qcfail.data(); }
    const char* name(int64_t i) const { return feed->name.data() + feed->name_offset[at(i)]; }
    size_t name_size(int64_t i) const { return static_cast<size_t>(feed->name_offset[at(i) + 1] - feed->name_offset[at(i)]); }
};

// one gzip member per call: members concatenate into a valid gzip stream, so the parts of a batch compress in parallel
void gzip_member(const char* text, size_t size, int level, ByteBuffer& out) {
    // up to the reference's default level 5 (configuration.json:6) the library's own compressor (fast_deflate.hpp: 4x zlib's level
    // 5 a thread on FASTQ text, 5 % more bytes; between zlib's levels 1 and 5 in size, past both in speed); a URL that asks for 6
    // and more, or PHQ_GZIP_ZLIB, gets zlib's deflate
    static const bool only_zlib = std::getenv("PHQ_GZIP_ZLIB") != nullptr;
    if(!only_zlib && level <= 5) {
        out.clear();
        const size_t bound = size + size / 8 + 128;
        char* const to = out.reserve(bound);
        const size_t produced = fast_deflate::compress_member(reinterpret_cast<const uint8_t*>(text), size, reinterpret_cast<uint8_t*>(to), bound);
        if(produced > 0) { out.size = produced; return; }
    }
    z_stream stream;
    std::memset(&stream, 0, sizeof(stream));
    if(deflateInit2(&stream, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) { throw InternalError("deflateInit2 failed"); }
    out.clear();
    const size_t bound = deflateBound(&stream, static_cast<uLong>(size)) + 32;
    char* to = out.reserve(bound);
    stream.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(text));
    stream.avail_in = static_cast<uInt>(size);
    stream.next_out = reinterpret_cast<Bytef*>(to);
    stream.avail_out = static_cast<uInt>(bound);
    const int status = deflate(&stream, Z_FINISH);
    const size_t produced = bound - stream.avail_out;
    deflateEnd(&stream);
    if(status != Z_STREAM_END) { throw InternalError("deflate failed"); }
    out.size = produced;
}

struct OutputSink {
    TranscodePlan::OutputFeed feed;
    int descriptor = -1;
    bool owns = false;
    bool wrote = false;
    bool null = false;                 // /dev/null: nothing is formatted for it (reference feed.h:105 NullFeed, transcode.cpp:1634,1753)
    bool seekable = false;             // a regular file: the parts of a batch are written in place concurrently
    uint64_t position = 0;
    void open() {
        if(feed.path == "/dev/null") { null = true; return; }
        if(feed.path == "-" || feed.path == "/dev/stdout") { std::fflush(stdout); descriptor = STDOUT_FILENO; }
        else {
            descriptor = ::open(feed.path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
            owns = true;
            if(descriptor < 0) { throw IOError("could not open " + feed.path + " for writing"); }
            struct stat status;
            seekable = ::fstat(descriptor, &status) == 0 && S_ISREG(status.st_mode);
        }
    }
    // `size` bytes at `at` (a seekable sink) or at the end of the stream
    void put(const char* bytes, size_t size, uint64_t at) {
        while(size > 0) {
            const ssize_t done = seekable ? ::pwrite(descriptor, bytes, size, static_cast<off_t>(at)) : ::write(descriptor, bytes, size);
            if(done < 0) { if(errno == EINTR) { continue; } throw IOError("error writing " + feed.path); }
            bytes += done; size -= static_cast<size_t>(done); at += static_cast<uint64_t>(done);
        }
    }
    void write(const char* bytes, size_t size) {
        if(null || size == 0) { return; }
        put(bytes, size, position);
        position += size;
        wrote = true;
    }
    // room for `size` bytes the workers copy into place themselves: the file grows and the new range is mapped (writes
    // to one file serialise on its lock whichever thread makes them; stores into a shared mapping do not)
    char* window = nullptr;
    size_t window_size = 0;
    uint64_t window_base = 0;
    char* map_more(size_t size) {
        const uint64_t at = position;
        const uint64_t page = static_cast<uint64_t>(::sysconf(_SC_PAGESIZE));
        window_base = at / page * page;
        window_size = static_cast<size_t>(at + size - window_base);
        // The file grows by ONE fallocate: the pages of the new range are allocated in a single walk (tmpfs: 11 GB/s on a thread of
        // this pool's hosts) and the workers' stores then take minor faults, sixteen pages each, at the memory's pace — where a
        // mapping of a merely truncated file allocates page by page inside the faults, every thread on the same tree and lock:
        // 3.6-4.0 GB/s on 8 or 16 threads against 5.9 / 9.1 (profiles/tools/micro/file_write.cpp, profiles/r06_file_write.txt;
        // one thread's plain write() is 6.3, sixteen threads' pwrite() 3.4-5.5: writes to one file serialise on its lock).
        if(::fallocate(descriptor, 0, static_cast<off_t>(at), static_cast<off_t>(size)) != 0) {
            if(::ftruncate(descriptor, static_cast<off_t>(at + size)) != 0) { throw IOError("error writing " + feed.path); }      // a file system without fallocate
        }
        void* mapped = ::mmap(nullptr, window_size, PROT_READ | PROT_WRITE, MAP_SHARED, descriptor, static_cast<off_t>(window_base));
        if(mapped == MAP_FAILED) { window = nullptr; return nullptr; }
        window = static_cast<char*>(mapped);
        position += size;
        wrote = wrote || size > 0;
        return window + (at - window_base);
    }
    void unmap() { if(window != nullptr) { ::munmap(window, window_size); window = nullptr; } }
    uint64_t reserve(size_t size) { const uint64_t at = position; position += size; wrote = wrote || size > 0; return at; }
    void close() {
        if(descriptor >= 0) {
            if(feed.format == FileFormat::BAM && !null) {    // BGZF's end of file marker: an empty block
                try { write(reinterpret_cast<const char*>(fast_deflate::BGZF_EOF), sizeof(fast_deflate::BGZF_EOF)); } catch(...) {}
            }
            if(feed.gzip && !wrote) {                        // an empty channel is still a valid gzip file
                ByteBuffer member;
                try { gzip_member("", 0, feed.level, member); write(member.data.get(), member.size); } catch(...) {}
            }
            if(owns) { ::close(descriptor); }
            descriptor = -1;
        }
    }
};

FileFormat output_format_of(const Url& url, const std::string& path) {
    switch(url.format) {
        case FileFormat::FASTQ: case FileFormat::SAM: return url.format;
        case FileFormat::UNKNOWN: case FileFormat::NONE: return FileFormat::SAM;                  // "default output format": "sam" (configuration.json:7)
        case FileFormat::BAM: return url.format;                                                  // unaligned BAM: bam_record.hpp, BGZF by fast_deflate.hpp
        case FileFormat::CRAM: throw ConfigurationError("CRAM output needs htslib: " + path);
        default: throw ConfigurationError("unknown output format " + path);                      // transcode.cpp:1752
    }
}

}  // namespace

TranscodePlan TranscodePlan::from_job(const JobConfig& job) {
    TranscodePlan plan;
    plan.sam_header = true;
    // compile_explicit_input: unique URLs in order of first appearance, resolution = times stated
    if(job.input.empty()) { throw ConfigurationError("the job states no input"); }
    for(const auto& encoded : job.input) {
        const Url url = Url::parse(encoded);
        if(url.format != FileFormat::UNKNOWN && url.format != FileFormat::FASTQ) {
            throw ConfigurationError("only FASTQ input is supported (SAM, BAM and CRAM need htslib): " + url.path);
        }
        int32_t feed = -1;
        for(size_t f = 0; f < plan.input_feed.size(); ++f) { if(plan.input_feed[f].path == url.path) { feed = static_cast<int32_t>(f); } }
        if(feed < 0) {
            feed = static_cast<int32_t>(plan.input_feed.size());
            plan.input_feed.push_back({url.path, 0});
        }
        plan.input_segment.emplace_back(feed, plan.input_feed[static_cast<size_t>(feed)].resolution++);
    }

    // compile_output: the channels of the multiplexing decoder, or one channel for everything
    std::vector<const BarcodeConfig*> record;
    if(job.multiplexing_decoder >= 0) {
        const DecoderConfig& d = job.decoder[static_cast<size_t>(job.multiplexing_decoder)];
        record.push_back(&d.undetermined);
        for(const auto& b : d.codec) { record.push_back(&b); }
    }
    bool any = !job.output.empty();
    for(const auto* r : record) { any = any || !r->output.empty(); }
    const std::vector<std::string> fallback = any ? job.output : std::vector<std::string>{"/dev/stdout"};     // configuration.json:13-15
    const int32_t n_out = job.output_segment_cardinality;
    auto add_channel = [&](const std::vector<std::string>& stated, bool filter) {
        Channel channel;
        channel.filter_outgoing_qc_fail = filter;
        std::vector<std::string> output = stated;
        if(output.size() == 1) { output.resize(static_cast<size_t>(n_out), output.front()); }               // pad_url_array_by_key
        if(!output.empty() && static_cast<int32_t>(output.size()) != n_out) { throw ConfigurationError("incorrect number of output URLs in channel"); }
        std::vector<int32_t> stated_in_channel(plan.output_feed.size(), 0);
        for(const auto& encoded : output) {
            const Url url = Url::parse(encoded);
            if(url.is_stdin()) { throw ConfigurationError("output stream can not be set to standard input"); }
            if(url.is_stderr()) { throw ConfigurationError("output stream can not be set to standard error"); }
            int32_t feed = -1;
            for(size_t f = 0; f < plan.output_feed.size(); ++f) { if(plan.output_feed[f].path == url.path) { feed = static_cast<int32_t>(f); } }
            if(feed < 0) {
                feed = static_cast<int32_t>(plan.output_feed.size());
                const FileCompression compression = url.compression();
                if(compression == FileCompression::BZ2 || compression == FileCompression::XZ) { throw ConfigurationError("bz2 and xz output are not supported: " + url.path); }
                const bool gzip = compression == FileCompression::GZIP || compression == FileCompression::BGZF;
                const FileFormat format = output_format_of(url, url.path);
                plan.output_feed.push_back({url.path, format, gzip && format != FileFormat::BAM, url.compression_level >= 0 ? url.compression_level : 5, 0});       // (a BAM file is BGZF already)
                stated_in_channel.push_back(0);
            }
            ++stated_in_channel[static_cast<size_t>(feed)];
            channel.feed_by_segment.push_back(feed);
        }
        for(size_t f = 0; f < stated_in_channel.size(); ++f) {                                               // transcode.cpp:1383-1393
            if(stated_in_channel[f] == 0) { continue; }
            int32_t& resolution = plan.output_feed[f].resolution;
            if(resolution == 0) { resolution = stated_in_channel[f]; }
            else if(resolution != stated_in_channel[f]) { throw ConfigurationError("inconsistent resolution for " + plan.output_feed[f].path); }
        }
        plan.channel.push_back(channel);
    };
    if(record.empty()) { add_channel(fallback, false); }
    for(const auto* r : record) { add_channel(r->output.empty() ? fallback : r->output, r->filter_outgoing_qc_fail); }
    return plan;
}

TranscodePlan TranscodePlan::explicit_files(const JobConfig& job, const std::vector<std::string>& input, const std::string& sam_output) {
    TranscodePlan plan;
    plan.sam_header = true;
    for(size_t s = 0; s < input.size(); ++s) {
        plan.input_feed.push_back({input[s], 1});
        plan.input_segment.emplace_back(static_cast<int32_t>(s), 0);
    }
    plan.output_feed.push_back({sam_output, FileFormat::SAM, false, 5, job.output_segment_cardinality});
    size_t channels = 1;
    if(job.multiplexing_decoder >= 0) { channels += job.decoder[static_cast<size_t>(job.multiplexing_decoder)].codec.size(); }
    Channel channel;
    channel.filter_outgoing_qc_fail = false;
    channel.feed_by_segment.assign(static_cast<size_t>(job.output_segment_cardinality), 0);
    plan.channel.assign(channels, channel);
    return plan;
}

struct Transcoder::Impl {
    DecoderSet& set;
    TranscodePlan plan;
    std::vector<std::unique_ptr<FastqStream>> stream;
    std::vector<OutputSink> sink;
    int64_t capacity;
    std::vector<SegmentBatch> batch;              // per input feed
    std::vector<SegmentBatch> ahead;              // the batch being read while `batch` is in work
    uint8_t* base;
    uint8_t* quality;
    uint8_t* flag;
    DevResult* result;
    uint32_t* word;
    std::vector<int64_t> kept;
    Transcoder::Classifier classifier;           // empty: DecoderSet::classify_host
    std::vector<DecoderSet*> copy;               // further copies of the set, on devices of their own (Transcoder::add_copy)

    Impl(DecoderSet& set, const TranscodePlan& plan) : set(set), plan(plan), capacity(0), base(nullptr), quality(nullptr), flag(nullptr), result(nullptr), word(nullptr) {}
    ~Impl() {
        for(auto& s : sink) { s.close(); }      // the staging planes belong to the set (DecoderSet::host_stage)
        // giving the batches' buffers back (and, in ~FastqStream, unmapping the input files) is page table work in the kernel:
        // 17 + 26 ms after 4 M reads, a third of a report only run of that size.  Both go to the helper thread (Afterwards).
        auto* const doomed = new std::pair<std::vector<SegmentBatch>, std::vector<SegmentBatch>>(std::move(batch), std::move(ahead));
        Afterwards::instance().run([doomed]() { delete doomed; });
    }
};

Transcoder::Transcoder(DecoderSet& set, const TranscodePlan& plan, int64_t batch_capacity) :
    impl_(new Impl(set, plan)) {
    try {
        const JobConfig& job = set.job();
        if(job.input_segment_cardinality > 0 && static_cast<int32_t>(plan.input_segment.size()) != job.input_segment_cardinality) {
            throw ConfigurationError("expected " + std::to_string(job.input_segment_cardinality) + " input files");
        }
        if(job.template_transform.empty()) { throw ConfigurationError("the job has no template transform"); }
        for(const auto& t : job.template_transform) {
            if(t.input_segment_index >= static_cast<int32_t>(plan.input_segment.size())) {
                throw ConfigurationError("invalid input feed reference " + std::to_string(t.input_segment_index) + " in token");     // transcode.cpp:1284-1288
            }
        }
        for(const auto& channel : plan.channel) {
            if(!channel.feed_by_segment.empty() && static_cast<int32_t>(channel.feed_by_segment.size()) != job.output_segment_cardinality) {
                throw ConfigurationError("incorrect number of output URLs in channel");
            }
        }
        for(const auto& feed : plan.input_feed) { impl_->stream.emplace_back(new FastqStream(feed.path)); }
        for(const auto& feed : plan.output_feed) {
            impl_->sink.emplace_back();
            impl_->sink.back().feed = feed;
            impl_->sink.back().open();
        }
        impl_->capacity = batch_capacity;
        impl_->batch.resize(plan.input_feed.size());
        const DecoderSet::HostStage& stage = set.host_stage(batch_capacity);
        impl_->base = stage.base; impl_->quality = stage.quality; impl_->flag = stage.flag; impl_->result = stage.result; impl_->word = stage.word;
        impl_->kept.resize(static_cast<size_t>(batch_capacity));
    } catch(...) {
        delete impl_;
        throw;
    }
}

Transcoder::~Transcoder() { delete impl_; }

void Transcoder::set_classifier(Classifier classifier) { impl_->classifier = std::move(classifier); }

void Transcoder::add_copy(DecoderSet& copy) {
    if(copy.job().stats_words != impl_->set.job().stats_words || copy.job().decoder.size() != impl_->set.job().decoder.size()) {
        throw ConfigurationError("a copy of the decoder set must be built from the same job");
    }
    impl_->copy.push_back(&copy);
}


// ---- report only runs: FASTQ text decoded on the device (fastq_device.hpp) ---------------------------------------------
namespace {

// what a set keeps from one report only run to the next: pinning and allocating these costs more than a small run
struct DeviceFastqCache {
    size_t raw_bytes[FASTQ_MAX_FEED] = {0};
    uint8_t* pinned[2][FASTQ_MAX_FEED] = {{nullptr}, {nullptr}};      // a round's bytes, and the next round's read ahead of its start being known
    uint8_t* raw[2][FASTQ_MAX_FEED] = {{nullptr}, {nullptr}};
    uint32_t* newline[2][FASTQ_MAX_FEED] = {{nullptr}, {nullptr}};
    uint32_t* block_count[2][FASTQ_MAX_FEED] = {{nullptr}, {nullptr}};
    uint32_t* total = nullptr;                     // [2][FASTQ_MAX_FEED] device words
    FastqRound* round_device = nullptr;
    FastqRound* round_host = nullptr;              // pinned
    uint32_t* irregular = nullptr;                 // device
    unsigned long long* flagged = nullptr;         // device
    uint64_t* counters_before = nullptr;           // device: the counter buffer when the run began
    int64_t capacity = 0;
    uint8_t* base = nullptr;
    uint8_t* quality = nullptr;
    uint8_t* flag = nullptr;
    DevResult* result = nullptr;
    uint32_t* word = nullptr;
    uint8_t* keep = nullptr;                       // device: a job with pre-filters: which reads of the round are decoded
    uint32_t* keep_block = nullptr;                // device: kept reads per workgroup's reads (at least 32), then their prefix sums
    uint32_t* kept_device = nullptr;
    uint32_t* kept_host = nullptr;                 // pinned
    void prepare_rounds(int64_t reads, const JobConfig& job);               // planes, results, the filters' scratch for rounds of `reads` reads
    void prepare_records(const DeviceText& text_plan, int64_t reads, const JobConfig& job, size_t n_channel, bool counts);
    void prepare_feed(int32_t f, size_t bytes);
    void prepare_tables(const FastqPlan& device_plan, int32_t n_feed, const size_t* chunk_bytes);
    hipStream_t stream = nullptr;
    FastqTables* tables = nullptr;                 // device: the plan, the chunks of both slots, the code tables (the kernels read them through this)
    // a run that writes: the records' text is assembled on the device (record_device.hpp) and leaves on a stream of its own
    RecordPlan* record_plan = nullptr;             // device
    int32_t* record_ints = nullptr;                // device
    uint8_t* record_bytes = nullptr;               // device
    size_t record_ints_room = 0, record_bytes_room = 0;
    int64_t record_capacity = 0;
    uint32_t* piece_size = nullptr;                // several sinks: device [capacity x n_out]
    unsigned long long* sink_block = nullptr;      // ... device [(capacity / 32 + 2) x n_sink]
    unsigned long long* sink_base = nullptr;       // ... device [n_sink + 1]
    unsigned long long* sink_base_host[2] = {nullptr, nullptr};      // pinned, per text buffer
    size_t piece_room = 0, sink_block_room = 0, sink_base_room = 0;
    unsigned long long* qc_device = nullptr;       // the quality control accumulators (record_device.hpp), cells of qc_cycles cycles
    int32_t qc_cycles = 0;
    size_t qc_cells = 0;
    uint32_t* qc_overflow = nullptr;               // device
    uint32_t* longest = nullptr;                   // device: [FASTQ_MAX_FEED] the longest input segment so far
    uint32_t* record_size = nullptr;               // device: [capacity]
    unsigned long long* record_block = nullptr;    // device: [capacity / 32 + 2]
    unsigned long long* record_total = nullptr;    // device
    struct RoundStatus { unsigned long long text_bytes; unsigned long long flagged; uint32_t irregular; uint32_t kept; uint32_t longest[FASTQ_MAX_FEED]; uint32_t overflow; };
    RoundStatus* status_host = nullptr;            // pinned
    char* text_device[2] = {nullptr, nullptr};
    char* text_host[2] = {nullptr, nullptr};       // pinned
    size_t text_device_room[2] = {0, 0}, text_host_room[2] = {0, 0};
    hipStream_t copy_stream = nullptr;
    hipEvent_t written[2] = {nullptr, nullptr};    // the round's text is in text_device[b]
    hipEvent_t copied[2] = {nullptr, nullptr};     // ... and in text_host[b]
    ~DeviceFastqCache() {
        (void)hipFree(piece_size); (void)hipFree(sink_block); (void)hipFree(sink_base);
        for(int b = 0; b < 2; ++b) { if(sink_base_host[b]) { (void)hipHostFree(sink_base_host[b]); } }
        (void)hipFree(tables); (void)hipFree(qc_device); (void)hipFree(qc_overflow); (void)hipFree(longest);
        (void)hipFree(record_plan); (void)hipFree(record_ints); (void)hipFree(record_bytes); (void)hipFree(record_size); (void)hipFree(record_block); (void)hipFree(record_total);
        if(status_host) { (void)hipHostFree(status_host); }
        for(int b = 0; b < 2; ++b) {
            (void)hipFree(text_device[b]);
            if(text_host[b]) { (void)hipHostFree(text_host[b]); }
            if(written[b]) { (void)hipEventDestroy(written[b]); }
            if(copied[b]) { (void)hipEventDestroy(copied[b]); }
        }
        if(copy_stream) { (void)hipStreamDestroy(copy_stream); }
        (void)hipFree(keep); (void)hipFree(keep_block); (void)hipFree(kept_device);
        if(kept_host) { (void)hipHostFree(kept_host); }
        for(int f = 0; f < FASTQ_MAX_FEED; ++f) {
            for(int b = 0; b < 2; ++b) { if(pinned[b][f]) { (void)hipHostFree(pinned[b][f]); } }
            for(int s = 0; s < 2; ++s) { (void)hipFree(raw[s][f]); (void)hipFree(newline[s][f]); (void)hipFree(block_count[s][f]); }
        }
        (void)hipFree(total); (void)hipFree(round_device); (void)hipFree(irregular); (void)hipFree(flagged); (void)hipFree(counters_before);
        if(round_host) { (void)hipHostFree(round_host); }
        (void)hipFree(base); (void)hipFree(quality); (void)hipFree(flag); (void)hipFree(result); (void)hipFree(word);
        if(stream) { (void)hipStreamDestroy(stream); }
    }
};

void device_check(hipError_t status, const char* what) {
    if(status != hipSuccess) { throw InternalError(std::string(what) + ": " + hipGetErrorString(status)); }
}

struct PlainFile {                     // an input file of the device path: plain, or a gzip stream inflated into the pinned buffer
    int descriptor = -1;
    size_t size = 0;
    size_t position = 0;               // of the round's first byte, in the (inflated) text
    double bytes_per_record = 0;
    gzFile zipped = nullptr;
    std::unique_ptr<MemberInflater> members;      // gzip of several members (BGZF blocks, concatenated members): inflated member parallel, a batch ahead (gzip_members.hpp)
    std::unique_ptr<GzipStream> stream;           // gzip of one member (or a few large ones): the streaming form of the same decoder (gzip_stream.hpp); gzread where neither opens
    bool ended = false;                // gzip: the stream has no more bytes
    size_t held = 0;                   // gzip: bytes of text of the round in its pinned buffer, from `offset` on
    size_t offset = 0;                 // gzip: where the round's text starts in its pinned buffer (behind the room kept for what the round before left)
    size_t ahead_filled = 0;           // gzip: bytes the task inflating ahead has put behind that room in the NEXT round's buffer
    std::vector<uint8_t> head;         // gzip: the text inflated while the file was looked at, until the first round takes it
    ~PlainFile() {
        if(zipped) { gzclose(zipped); }
        if(descriptor >= 0) { ::close(descriptor); }
    }
};

}  // namespace

namespace {
std::atomic<uint64_t> device_rounds{0}, device_reads{0}, device_record_bytes{0}, device_resumed{0};
}
void device_path_counters(uint64_t* counters) {
    counters[0] = device_rounds; counters[1] = device_reads; counters[2] = device_record_bytes; counters[3] = device_resumed;
}

// A run whose every written channel sends all its output segments to ONE uncompressed file: which sink, and the tables
// the device needs to assemble the records (record_device.hpp).  `ints` and `bytes` are the two pools RecordPlan points into.
struct DeviceText {
    std::vector<int32_t> sinks;     // the output files records go to (indices into the transcoder's sinks); empty: a run that only counts
    bool quality_control = false;   // the accumulators of `enable quality control` are kept on the device (record_quality_control)
    RecordPlan plan;
    std::vector<int32_t> ints;
    std::vector<uint8_t> bytes;
};

namespace {

// false: the job is not one the device writes (the host writer takes it)
bool plan_device_text(const JobConfig& job, const TranscodePlan& plan, const std::vector<OutputSink>& sink, const std::vector<std::string>& rg_by_index,
                      const std::vector<std::vector<int>>& ordinal_by_index, DeviceText& out) {
    if(std::getenv("PHQ_HOST_RECORDS") != nullptr) { return false; }
    const int32_t n_out = job.output_segment_cardinality;
    if(n_out < 1 || n_out > RECORD_MAX_OUT || job.template_transform.size() > static_cast<size_t>(RECORD_MAX_TOKEN) || job.decoder.size() > static_cast<size_t>(FASTQ_MAX_DECODER)) { return false; }
    // the channels: an empty one, or one whose every URL is /dev/null, takes its reads without a record; of the others every
    // output segment goes to the file of its URL (uncompressed or gzip, SAM or FASTQ records): the sinks of the plan
    out.sinks.clear();
    std::vector<int32_t> device_sink(sink.size(), -1);
    std::vector<int32_t> sink_of_segment(plan.channel.size() * static_cast<size_t>(n_out), -1);
    std::vector<uint8_t> mode(plan.channel.size(), 0);
    for(size_t c = 0; c < plan.channel.size(); ++c) {
        const auto& channel = plan.channel[c];
        if(channel.feed_by_segment.empty()) { continue; }
        bool writes = false;
        for(int32_t o = 0; o < n_out; ++o) {
            const int32_t f = channel.feed_by_segment[static_cast<size_t>(o)];
            const OutputSink& to = sink[static_cast<size_t>(f)];
            if(to.null) { continue; }
            if(to.feed.format != FileFormat::SAM && to.feed.format != FileFormat::FASTQ && to.feed.format != FileFormat::BAM) { return false; }
            if(device_sink[static_cast<size_t>(f)] < 0) { device_sink[static_cast<size_t>(f)] = static_cast<int32_t>(out.sinks.size()); out.sinks.push_back(f); }
            sink_of_segment[c * static_cast<size_t>(n_out) + static_cast<size_t>(o)] = device_sink[static_cast<size_t>(f)];
            writes = true;
        }
        if(writes) { mode[c] = static_cast<uint8_t>(1u | (channel.filter_outgoing_qc_fail ? 2u : 0u)); }
    }
    if(out.sinks.size() > static_cast<size_t>(RECORD_MAX_SINK)) { return false; }

    RecordPlan& r = out.plan;
    std::memset(&r, 0, sizeof(r));
    out.ints.clear();
    out.bytes.clear();
    r.n_decoder = static_cast<int32_t>(job.decoder.size());
    r.n_out = n_out;
    r.n_channel = static_cast<int32_t>(plan.channel.size());
    r.n_sink = static_cast<int32_t>(out.sinks.size());
    r.illumina_comment = (job.platform == "ILLUMINA" || job.platform == "ELEMENT") ? 1 : 0;      // fastq.h:176-194
    r.quality_shift = job.output_phred_offset - job.input_phred_offset;
    r.input_phred_offset = job.input_phred_offset;
    r.output_phred_offset = job.output_phred_offset;
    // read group ids: where each one's bytes end
    r.n_group = static_cast<int32_t>(rg_by_index.size());
    r.group_at = static_cast<int32_t>(out.ints.size());
    for(const auto& id : rg_by_index) {
        out.ints.push_back(static_cast<int32_t>(out.bytes.size()));
        out.bytes.insert(out.bytes.end(), id.begin(), id.end());
    }
    out.ints.push_back(static_cast<int32_t>(out.bytes.size()));
    r.channel_at = static_cast<int32_t>(out.bytes.size());
    out.bytes.insert(out.bytes.end(), mode.begin(), mode.end());
    r.format_at = static_cast<int32_t>(out.bytes.size());
    for(const int32_t f : out.sinks) {
        const FileFormat format = sink[static_cast<size_t>(f)].feed.format;
        out.bytes.push_back(format == FileFormat::FASTQ ? RECORD_FORMAT_FASTQ : format == FileFormat::BAM ? RECORD_FORMAT_SAM_FOR_BAM : RECORD_FORMAT_SAM);
    }
    out.bytes.push_back(0);
    r.sink_at = static_cast<int32_t>(out.ints.size());
    out.ints.insert(out.ints.end(), sink_of_segment.begin(), sink_of_segment.end());
    // the tokens of every output segment, in template order
    int32_t placed = 0;
    for(int32_t o = 0; o < n_out; ++o) {
        r.out_first[o] = placed;
        for(size_t t = 0; t < job.template_transform.size(); ++t) {
            if(job.template_transform[t].output_segment_index == o) { r.out_token[placed++] = static_cast<int32_t>(t); }
        }
    }
    r.out_first[n_out] = placed;
    for(size_t t = 0; t < job.template_transform.size(); ++t) {
        const TransformConfig& token = job.template_transform[t];
        if(token.input_segment_index < -3 || token.output_segment_index < 0 || token.output_segment_index >= n_out) { return false; }
        r.token[t] = RecordToken{token.input_segment_index, token.start, token.end, token.end_terminated ? 1 : 0, token.reverse_complement ? 1 : 0};
    }
    for(int32_t o = 0; o < n_out; ++o) {
        for(int failed = 0; failed < 2; ++failed) {
            int flag = 4 | 8;                                                 // BAM_FUNMAP | BAM_FMUNMAP (read.h:134-135)
            if(n_out > 1) { flag |= 1; }
            if(o == 0) { flag |= 64; }
            if(n_out > 1 && o == n_out - 1) { flag |= 128; }
            if(failed) { flag |= 512; }
            const std::string text = "\t" + std::to_string(flag) + "\t*\t0\t0\t*\t*\t0\t0\t";
            if(text.size() > sizeof(r.columns[o][failed])) { return false; }
            std::memcpy(r.columns[o][failed], text.data(), text.size());
            r.columns_size[o][failed] = static_cast<int32_t>(text.size());
        }
    }
    for(size_t k = 0; k < job.decoder.size(); ++k) {
        const DecoderConfig& d = job.decoder[k];
        RecordDecoder& to_decoder = r.decoder[k];
        if(d.segment_cardinality > DEV_MAX_SEGMENT) { return false; }
        to_decoder.kind = static_cast<int32_t>(d.kind);
        to_decoder.plane_prefix = d.plane_prefix;
        to_decoder.plane_stride = d.plane_stride;
        to_decoder.result_slot = d.result_slot;
        to_decoder.segment_cardinality = d.segment_cardinality;
        to_decoder.nucleotide_cardinality = d.nucleotide_cardinality;
        to_decoder.closed = d.closed() ? 1 : 0;
        to_decoder.pamld = d.algorithm == Algorithm::PAMLD ? 1 : 0;
        to_decoder.corrected_quality = d.corrected_quality;
        to_decoder.multiplexing = d.multiplexing_classifier ? 1 : 0;
        to_decoder.n_ordinal = static_cast<int32_t>(ordinal_by_index[k].size());
        to_decoder.ordinal_at = static_cast<int32_t>(out.ints.size());
        out.ints.insert(out.ints.end(), ordinal_by_index[k].begin(), ordinal_by_index[k].end());
        to_decoder.code_at = static_cast<int32_t>(out.bytes.size());
        out.bytes.insert(out.bytes.end(), d.barcode_code.begin(), d.barcode_code.end());
        for(int32_t s = 0; s < d.segment_cardinality; ++s) {
            to_decoder.segment_offset[s] = d.segment_offset[static_cast<size_t>(s)];
            to_decoder.barcode_length[s] = d.barcode_length[static_cast<size_t>(s)];
        }
    }
    for(int c = 0; c < 16; ++c) {
        r.text_of_bam[c] = bam_to_ascii(static_cast<uint8_t>(c));
        r.complement_of_bam[c] = reverse_complement_bam(static_cast<uint8_t>(c));
    }
    for(int c = 0; c < 256; ++c) {
        r.complement_text[c] = bam_to_ascii(reverse_complement_bam(ascii_to_bam(static_cast<uint8_t>(c))));
        r.canonical_text[c] = bam_to_ascii(ascii_to_bam(static_cast<uint8_t>(c)));
    }
    out.ints.push_back(0);
    out.bytes.push_back(0);
    return true;
}

}  // namespace

namespace {

// the job's input side as the decode kernels read it (fastq_device.hpp); false: a job the device path does not take
bool device_fastq_plan(const JobConfig& job, const TranscodePlan& plan, FastqPlan& device_plan) {
    const int32_t n_feed = static_cast<int32_t>(plan.input_feed.size());
    const int32_t n_segment = static_cast<int32_t>(plan.input_segment.size());
    if(n_feed < 1 || n_feed > FASTQ_MAX_FEED || static_cast<int32_t>(job.decoder.size()) > FASTQ_MAX_DECODER || n_segment < 1 || n_segment > FASTQ_MAX_FEED) { return false; }
    std::memset(&device_plan, 0, sizeof(device_plan));
    device_plan.n_feed = n_feed;
    device_plan.n_decoder = static_cast<int32_t>(job.decoder.size());
    // interleaved feeds (a path stated k times in `input`: transcode.cpp:693-701): segment s is record r k + slot of its file
    device_plan.n_segment = n_segment;
    for(int32_t f = 0; f < n_feed; ++f) { device_plan.feed_resolution[f] = std::max(1, plan.input_feed[static_cast<size_t>(f)].resolution); }
    for(int32_t s = 0; s < n_segment; ++s) {
        device_plan.segment_feed[s] = plan.input_segment[static_cast<size_t>(s)].first;
        device_plan.segment_slot[s] = plan.input_segment[static_cast<size_t>(s)].second;
        if(device_plan.segment_feed[s] < 0 || device_plan.segment_feed[s] >= n_feed || device_plan.segment_slot[s] < 0 ||
           device_plan.segment_slot[s] >= device_plan.feed_resolution[device_plan.segment_feed[s]]) { return false; }
    }
    // the pre-filters of the pull loop (transcode.h:193-205), on the device too: kept reads are compacted into the planes
    device_plan.filter_incoming_qc_fail = job.filter_incoming_qc_fail ? 1 : 0;
    for(int32_t s = 1; s < n_segment && s < static_cast<int32_t>(job.min_input_length.size()); ++s) { device_plan.min_length[s] = job.min_input_length[static_cast<size_t>(s)]; }
    device_plan.filtering = device_plan.filter_incoming_qc_fail;
    for(int32_t s = 1; s < n_segment; ++s) { if(device_plan.min_length[s] > 0) { device_plan.filtering = 1; } }
    device_plan.leader_segment = job.leading_segment_index;
    device_plan.phred_offset = job.input_phred_offset;
    device_plan.illumina_comment = (job.platform == "ILLUMINA" || job.platform == "ELEMENT") ? 1 : 0;
    int32_t n_token = 0;
    for(size_t k = 0; k < job.decoder.size(); ++k) {
        const DecoderConfig& d = job.decoder[k];
        FastqDecoderPlan& to = device_plan.decoder[k];
        to.plane_prefix = d.plane_prefix;
        to.plane_stride = d.plane_stride;
        to.first_token = n_token;
        for(const auto& t : d.transform) {
            if(n_token >= FASTQ_MAX_TOKEN || t.input_segment_index < 0 || t.input_segment_index >= static_cast<int32_t>(plan.input_segment.size()) ||
               t.output_segment_index < 0 || t.output_segment_index >= DEV_MAX_SEGMENT || d.plane_stride == 0) { return false; }
            FastqToken& token = device_plan.token[n_token++];
            token.segment = t.input_segment_index;
            token.start = t.start;
            token.end = t.end;
            token.end_terminated = t.end_terminated ? 1 : 0;
            token.reverse_complement = t.reverse_complement ? 1 : 0;
            token.output_offset = d.segment_offset[static_cast<size_t>(t.output_segment_index)];
            token.output_room = d.barcode_length[static_cast<size_t>(t.output_segment_index)];
            token.output_segment = t.output_segment_index;
        }
        to.n_token = n_token - to.first_token;
    }
    return true;
}

// the input files of a run on the device: regular files, plain or gzip, that start like FASTQ; false: the host reader's
bool open_device_inputs(const TranscodePlan& plan, std::vector<PlainFile>& file) {
    const int32_t n_feed = static_cast<int32_t>(plan.input_feed.size());
    file = std::vector<PlainFile>(static_cast<size_t>(n_feed));
    for(int32_t f = 0; f < n_feed; ++f) {
        PlainFile& p = file[static_cast<size_t>(f)];
        p.descriptor = ::open(plan.input_feed[static_cast<size_t>(f)].path.c_str(), O_RDONLY);
        if(p.descriptor < 0) { return false; }
        struct stat status;
        if(::fstat(p.descriptor, &status) != 0 || !S_ISREG(status.st_mode) || status.st_size <= 0 || static_cast<uint64_t>(status.st_size) >= (uint64_t(1) << 40)) { return false; }
        p.size = static_cast<size_t>(status.st_size);
        char head[1 << 16];
        ssize_t got = ::pread(p.descriptor, head, sizeof(head), 0);
        if(got >= 2 && static_cast<unsigned char>(head[0]) == 0x1f && static_cast<unsigned char>(head[1]) == 0x8b) {
            // a gzip stream: inflated on a host thread of the feed's own into the pinned buffer, the text then goes the same way
            // (inflating bounds such a run whoever does it: 0.3 GB/s of text a stream)
            p.zipped = gzopen(plan.input_feed[static_cast<size_t>(f)].path.c_str(), "rb");
            if(p.zipped == nullptr) { return false; }
            gzbuffer(p.zipped, 1 << 20);
            // ... unless the stream has several members (bgzip's blocks, the members this library writes): those inflate side by
            // side on every host thread, a batch ahead of the rounds (gzip_members.hpp)
            static const bool one_thread = std::getenv("PHQ_GZIP_ONE_THREAD") != nullptr;
            if(!one_thread) {
                p.members.reset(new MemberInflater());
                if(!p.members->open(plan.input_feed[static_cast<size_t>(f)].path, host_threads())) { p.members.reset(); }
            }
            p.head.resize(sizeof(head));
            int inflated = -1;
            if(p.members) {
                // a member that does not hold to its trailer (a text of 4 GiB and more wraps it), or no memory for it: the
                // stream is still at its start, gzread takes the file as before
                try { inflated = static_cast<int>(p.members->read(p.head.data(), p.head.size())); }
                catch(const std::exception&) { p.members.reset(); inflated = -1; }
            }
            static const bool only_zlib = std::getenv("PHQ_GZIP_ZLIB") != nullptr;
            if(!p.members && !one_thread && !only_zlib) {
                // one member: a sequential stream, through fast_inflate.hpp's decoder in its streaming form (gzip_stream.hpp)
                p.stream.reset(new GzipStream());
                try {
                    // (twice the threads: a span's phases leave gaps — this thread's own chunk is done first, the windows are resolved by one — that a second set of chunks fills: 25 -> 34 Mreads/s for two files on 16 cores)
                    if(p.stream->open(plan.input_feed[static_cast<size_t>(f)].path, std::min(32, 2 * host_threads()))) { inflated = static_cast<int>(p.stream->read(p.head.data(), p.head.size())); }
                    else { p.stream.reset(); }
                } catch(const std::exception&) { p.stream.reset(); inflated = -1; }
            }
            if(!p.members && !p.stream) { inflated = gzread(p.zipped, p.head.data(), static_cast<unsigned>(p.head.size())); }
            if(inflated < 4) { return false; }
            p.head.resize(static_cast<size_t>(inflated));
            std::memcpy(head, p.head.data(), p.head.size());
            got = inflated;
            p.size = size_t(1) << 62;                   // not known
        }
        if(got < 4 || head[0] != '@') { return false; }
        size_t lines = 0;
        for(ssize_t k = 0; k < got; ++k) { lines += head[k] == '\n' ? 1 : 0; }
        if(lines < 4) { return false; }
        p.bytes_per_record = 4.0 * static_cast<double>(got) / static_cast<double>(lines);
    }

    return true;
}

// the device's quality control cells (record_device.hpp) as the host's accumulators (qc.hpp)
QcState quality_control_of_cells(const std::vector<unsigned long long>& words, size_t n_cell, int32_t cycles, int32_t n_channel, int32_t n_out) {
    const size_t cell_words = qc_cell_words(cycles);
    QcState state;
    state.shape(n_channel, n_out);
    for(size_t k = 0; k < n_cell; ++k) {
        const unsigned long long* const w = words.data() + k * cell_words;
        QcSegment& cell = state.cell[k];
        cell.capacity = static_cast<int32_t>(w[0]);
        cell.shortest = static_cast<int32_t>(0x7fffffffull - w[1]);
        cell.average_count = w[2];
        cell.average_sum_fixed = w[3];
        std::memcpy(&cell.average_max, &w[4], sizeof(double));
        for(int q = 0; q < QC_PHRED_RANGE; ++q) { cell.average_distribution[q] = w[5 + q]; }
        const size_t bins = static_cast<size_t>(cell.capacity) * QC_CODES * QC_PHRED_RANGE;
        cell.cycle.assign(w + QC_CELL_HEAD, w + QC_CELL_HEAD + bins);
        uint64_t every = 0;
        for(size_t bin = 0; bin < bins; ++bin) {
            cell.nucleic[(bin / QC_PHRED_RANGE) % QC_CODES] += cell.cycle[bin];
            every += cell.cycle[bin];
        }
        cell.nucleic[0] += every;                                           // multiplex.h:155-156: [0] counts every base (and code 0 twice)
    }
    return state;
}

// how many cycles a quality control cell needs: the longest any output segment can be — every token at most the longest of
// its input segment so far, a corrected barcode at most the job's nucleotides
int32_t quality_control_cycles(const RecordPlan& plan, const uint32_t* longest, int32_t nucleotide_width) {
    int32_t cycles = 0;
    for(int32_t o = 0; o < plan.n_out; ++o) {
        int32_t bound = 0;
        for(int32_t t = plan.out_first[o]; t < plan.out_first[o + 1]; ++t) {
            const RecordToken& token = plan.token[plan.out_token[t]];
            bound += token.source >= 0 ? static_cast<int32_t>(longest[token.source]) : nucleotide_width;
        }
        cycles = std::max(cycles, bound);
    }
    return std::max(64, (cycles + 31) / 32 * 32);
}

// SAM lines the device assembled for a BAM sink (RECORD_FORMAT_SAM_FOR_BAM) as BAM records in BGZF blocks behind `packed`'s bytes
void bam_blocks_of_lines(const char* lines, size_t size, ByteBuffer& records, ByteBuffer& packed) {
    if(size == 0) { return; }
    append_bam_records(records, lines, size);
    const size_t bound = fast_deflate::bgzf_bound(records.size);
    char* const to = packed.reserve(bound);
    const size_t made = fast_deflate::bgzf_blocks(reinterpret_cast<const uint8_t*>(records.data.get()), records.size, reinterpret_cast<uint8_t*>(to), bound);
    if(made == 0) { throw InternalError("BGZF"); }
    packed.commit(to + made);
}

// A round's text into the files: sink k's records are bytes [base_of_sink[k], base_of_sink[k + 1]) of `from`.  A large part for
// a regular file goes through a mapping on several host threads; gzip sinks are compressed a member per piece, the sinks (or
// the pieces of a single one) side by side.
void write_round_text(const std::vector<OutputSink*>& to, const char* from, const unsigned long long* base_of_sink, int threads) {
    const int64_t n_sink = static_cast<int64_t>(to.size());
    if(n_sink == 1) {
        OutputSink& sink = *to[0];
        const size_t bytes = static_cast<size_t>(base_of_sink[1]);
        if(sink.feed.format == FileFormat::BAM) {
            // pieces of about 2 MiB of lines, each from the start of a line: records, then BGZF blocks, side by side; written in order
            const size_t piece = size_t(2) << 20;
            const int64_t pieces = static_cast<int64_t>((bytes + piece - 1) / piece);
            std::vector<size_t> begin_of(static_cast<size_t>(pieces) + 1, bytes);
            begin_of[0] = 0;
            for(int64_t k = 1; k < pieces; ++k) {
                const size_t from_byte = static_cast<size_t>(k) * piece - 1;
                const char* const newline = static_cast<const char*>(std::memchr(from + from_byte, '\n', bytes - from_byte));
                begin_of[static_cast<size_t>(k)] = newline ? static_cast<size_t>(newline - from) + 1 : bytes;
            }
            std::vector<ByteBuffer> packed(static_cast<size_t>(pieces));
            parallel_ranges(pieces, 1, threads, [&](int64_t first, int64_t last, int) {
                ByteBuffer records;
                for(int64_t k = first; k < last; ++k) {
                    records.clear();
                    bam_blocks_of_lines(from + begin_of[static_cast<size_t>(k)], begin_of[static_cast<size_t>(k) + 1] - begin_of[static_cast<size_t>(k)], records, packed[static_cast<size_t>(k)]);
                }
            });
            for(const auto& blocks : packed) { if(blocks.size) { sink.write(blocks.data.get(), blocks.size); } }
            return;
        }
        if(sink.feed.gzip) {
            const size_t piece = size_t(4) << 20;
            const int64_t pieces = static_cast<int64_t>((bytes + piece - 1) / piece);
            std::vector<ByteBuffer> packed(static_cast<size_t>(pieces));
            parallel_ranges(pieces, 1, threads, [&](int64_t first, int64_t last, int) {
                for(int64_t k = first; k < last; ++k) {
                    const size_t begin = static_cast<size_t>(k) * piece;
                    gzip_member(from + begin, std::min(piece, bytes - begin), sink.feed.level, packed[static_cast<size_t>(k)]);
                }
            });
            for(const auto& member : packed) { sink.write(member.data.get(), member.size); }
            return;
        }
        char* const window = (sink.seekable && bytes >= (1u << 20)) ? sink.map_more(bytes) : nullptr;
        if(window == nullptr) { sink.write(from, bytes); return; }
        parallel_ranges(static_cast<int64_t>((bytes + (1 << 20) - 1) >> 20), 1, threads, [&](int64_t first, int64_t last, int) {
            const size_t begin = static_cast<size_t>(first) << 20;
            const size_t end = std::min(bytes, static_cast<size_t>(last) << 20);
            std::memcpy(window + begin, from + begin, end - begin);
        });
        sink.unmap();
        return;
    }
    parallel_ranges(n_sink, 1, threads, [&](int64_t first, int64_t last, int) {
        ByteBuffer packed, records;
        for(int64_t k = first; k < last; ++k) {
            OutputSink& sink = *to[static_cast<size_t>(k)];
            const size_t begin = static_cast<size_t>(base_of_sink[k]), bytes = static_cast<size_t>(base_of_sink[k + 1] - base_of_sink[k]);
            if(bytes == 0) { continue; }
            if(sink.feed.format == FileFormat::BAM) {
                packed.clear();
                records.clear();
                bam_blocks_of_lines(from + begin, bytes, records, packed);
                sink.write(packed.data.get(), packed.size);
            } else if(sink.feed.gzip) {
                packed.clear();
                gzip_member(from + begin, bytes, sink.feed.level, packed);
                sink.write(packed.data.get(), packed.size);
            } else { sink.write(from + begin, bytes); }
        }
    });
}

}  // namespace

void DeviceFastqCache::prepare_rounds(int64_t reads, const JobConfig& job) {
    if(capacity < reads) {
        (void)hipFree(base); (void)hipFree(quality); (void)hipFree(flag); (void)hipFree(result); (void)hipFree(word);
        base = quality = flag = nullptr; result = nullptr; word = nullptr; capacity = 0;
        const size_t planes = static_cast<size_t>(reads) * static_cast<size_t>(std::max(job.plane_width, 4));
        device_check(hipMalloc(reinterpret_cast<void**>(&base), planes), "hipMalloc");
        device_check(hipMalloc(reinterpret_cast<void**>(&quality), planes), "hipMalloc");
        device_check(hipMalloc(reinterpret_cast<void**>(&flag), static_cast<size_t>(reads)), "hipMalloc");
        device_check(hipMalloc(reinterpret_cast<void**>(&result), static_cast<size_t>(reads) * static_cast<size_t>(std::max(job.result_count, 1)) * sizeof(DevResult)), "hipMalloc");
        device_check(hipMalloc(reinterpret_cast<void**>(&word), static_cast<size_t>(reads) * sizeof(uint32_t)), "hipMalloc");
        (void)hipFree(keep); (void)hipFree(keep_block); keep = nullptr; keep_block = nullptr;
        device_check(hipMalloc(reinterpret_cast<void**>(&keep), static_cast<size_t>(reads)), "hipMalloc");
        device_check(hipMalloc(reinterpret_cast<void**>(&keep_block), (static_cast<size_t>(reads) / 32 + 2) * sizeof(uint32_t)), "hipMalloc");
        capacity = reads;
    }
    if(total == nullptr) {
        device_check(hipMalloc(reinterpret_cast<void**>(&total), 2 * FASTQ_MAX_FEED * sizeof(uint32_t)), "hipMalloc");
        device_check(hipMalloc(reinterpret_cast<void**>(&round_device), sizeof(FastqRound)), "hipMalloc");
        device_check(hipHostMalloc(reinterpret_cast<void**>(&round_host), sizeof(FastqRound), hipHostMallocDefault), "hipHostMalloc");
        device_check(hipMalloc(reinterpret_cast<void**>(&irregular), sizeof(uint32_t)), "hipMalloc");
        device_check(hipMalloc(reinterpret_cast<void**>(&flagged), sizeof(unsigned long long)), "hipMalloc");
        device_check(hipMalloc(reinterpret_cast<void**>(&counters_before), static_cast<size_t>(job.stats_words) * sizeof(uint64_t)), "hipMalloc");
        device_check(hipMalloc(reinterpret_cast<void**>(&kept_device), sizeof(uint32_t)), "hipMalloc");
        device_check(hipHostMalloc(reinterpret_cast<void**>(&kept_host), sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc");
    }
}

// what a run that writes or counts needs beside: the record plan and its pools, the passes' scratch, the sinks' tables
void DeviceFastqCache::prepare_records(const DeviceText& text_plan, int64_t reads, const JobConfig& job, size_t n_channel, bool counts) {
    const DeviceText* const text = &text_plan;
    {
        if(copy_stream == nullptr) {
            device_check(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking), "hipStreamCreate");
            for(int b = 0; b < 2; ++b) {
                device_check(hipEventCreateWithFlags(&written[b], hipEventDisableTiming), "hipEventCreate");
                device_check(hipEventCreateWithFlags(&copied[b], hipEventDisableTiming), "hipEventCreate");
            }
            device_check(hipHostMalloc(reinterpret_cast<void**>(&status_host), sizeof(*status_host), hipHostMallocDefault), "hipHostMalloc");
            device_check(hipMalloc(reinterpret_cast<void**>(&record_plan), sizeof(RecordPlan)), "hipMalloc");
            device_check(hipMalloc(reinterpret_cast<void**>(&record_total), sizeof(unsigned long long)), "hipMalloc");
        }
        if(record_capacity < reads) {
            (void)hipFree(record_size); (void)hipFree(record_block); record_size = nullptr; record_block = nullptr; record_capacity = 0;
            device_check(hipMalloc(reinterpret_cast<void**>(&record_size), static_cast<size_t>(reads) * sizeof(uint32_t)), "hipMalloc");
            device_check(hipMalloc(reinterpret_cast<void**>(&record_block), (static_cast<size_t>(reads) / 32 + 2) * sizeof(unsigned long long)), "hipMalloc");
            record_capacity = reads;
        }
        {
            const size_t n_sink = std::max<size_t>(text->sinks.size(), 1);
            const size_t pieces = static_cast<size_t>(reads) * static_cast<size_t>(std::max(job.output_segment_cardinality, 1));
            const size_t blocks = (static_cast<size_t>(reads) / 32 + 2) * n_sink;
            if(piece_room < pieces) {
                (void)hipFree(piece_size); piece_size = nullptr; piece_room = 0;
                device_check(hipMalloc(reinterpret_cast<void**>(&piece_size), pieces * sizeof(uint32_t)), "hipMalloc");
                piece_room = pieces;
            }
            if(sink_block_room < blocks) {
                (void)hipFree(sink_block); sink_block = nullptr; sink_block_room = 0;
                device_check(hipMalloc(reinterpret_cast<void**>(&sink_block), blocks * sizeof(unsigned long long)), "hipMalloc");
                sink_block_room = blocks;
            }
            if(sink_base_room < n_sink + 1) {
                (void)hipFree(sink_base); sink_base = nullptr;
                device_check(hipMalloc(reinterpret_cast<void**>(&sink_base), (n_sink + 1) * sizeof(unsigned long long)), "hipMalloc");
                for(int b = 0; b < 2; ++b) {
                    if(sink_base_host[b]) { (void)hipHostFree(sink_base_host[b]); sink_base_host[b] = nullptr; }
                    device_check(hipHostMalloc(reinterpret_cast<void**>(&sink_base_host[b]), (n_sink + 1) * sizeof(unsigned long long), hipHostMallocDefault), "hipHostMalloc");
                }
                sink_base_room = n_sink + 1;
            }
        }
        if(record_ints_room < text->ints.size()) {
            (void)hipFree(record_ints); record_ints = nullptr; record_ints_room = 0;
            device_check(hipMalloc(reinterpret_cast<void**>(&record_ints), text->ints.size() * sizeof(int32_t)), "hipMalloc");
            record_ints_room = text->ints.size();
        }
        if(record_bytes_room < text->bytes.size()) {
            (void)hipFree(record_bytes); record_bytes = nullptr; record_bytes_room = 0;
            device_check(hipMalloc(reinterpret_cast<void**>(&record_bytes), text->bytes.size()), "hipMalloc");
            record_bytes_room = text->bytes.size();
        }
        RecordPlan on_device = text->plan;
        on_device.ints = record_ints;
        on_device.bytes = record_bytes;
        device_check(hipMemcpy(record_ints, text->ints.data(), text->ints.size() * sizeof(int32_t), hipMemcpyHostToDevice), "hipMemcpy");
        device_check(hipMemcpy(record_bytes, text->bytes.data(), text->bytes.size(), hipMemcpyHostToDevice), "hipMemcpy");
        device_check(hipMemcpy(record_plan, &on_device, sizeof(RecordPlan), hipMemcpyHostToDevice), "hipMemcpy");
        std::memset(status_host, 0, sizeof(*status_host));
        if(counts) {
            if(longest == nullptr) {
                device_check(hipMalloc(reinterpret_cast<void**>(&longest), FASTQ_MAX_FEED * sizeof(uint32_t)), "hipMalloc");
                device_check(hipMalloc(reinterpret_cast<void**>(&qc_overflow), sizeof(uint32_t)), "hipMalloc");
            }
            device_check(hipMemset(longest, 0, FASTQ_MAX_FEED * sizeof(uint32_t)), "hipMemset");
            device_check(hipMemset(qc_overflow, 0, sizeof(uint32_t)), "hipMemset");
            // the accumulators start empty with every run (what a run leaves is merged into the set's when it ends)
            (void)hipFree(qc_device); qc_device = nullptr; qc_cycles = 0;
            qc_cells = n_channel * static_cast<size_t>(job.output_segment_cardinality);
        }
    }
}

// two pinned and two device buffers of a feed's text, the newline index of either slot
void DeviceFastqCache::prepare_feed(int32_t f, size_t bytes) {
    size_t chunk_bytes[FASTQ_MAX_FEED] = {0};
    chunk_bytes[f] = bytes;
    if(raw_bytes[f] < chunk_bytes[f]) {
        for(int b = 0; b < 2; ++b) { if(pinned[b][f]) { (void)hipHostFree(pinned[b][f]); pinned[b][f] = nullptr; } }
        for(int s = 0; s < 2; ++s) { (void)hipFree(raw[s][f]); (void)hipFree(newline[s][f]); (void)hipFree(block_count[s][f]); raw[s][f] = nullptr; newline[s][f] = nullptr; block_count[s][f] = nullptr; }
        raw_bytes[f] = 0;
        for(int b = 0; b < 2; ++b) { device_check(hipHostMalloc(reinterpret_cast<void**>(&pinned[b][f]), chunk_bytes[f] + chunk_bytes[f] / 8 + 8192 + 16, hipHostMallocDefault), "hipHostMalloc"); }
        for(int s = 0; s < 2; ++s) {
            device_check(hipMalloc(reinterpret_cast<void**>(&raw[s][f]), chunk_bytes[f] + 16), "hipMalloc");
            device_check(hipMalloc(reinterpret_cast<void**>(&newline[s][f]), (chunk_bytes[f] / 8 + 1024) * sizeof(uint32_t)), "hipMalloc");
            device_check(hipMalloc(reinterpret_cast<void**>(&block_count[s][f]), (chunk_bytes[f] / 4096 + 2) * sizeof(uint32_t)), "hipMalloc");
        }
        raw_bytes[f] = chunk_bytes[f];
    }
}

// what the kernels read through a pointer: the plan, where the chunks of either slot are, the code tables
void DeviceFastqCache::prepare_tables(const FastqPlan& device_plan, int32_t n_feed, const size_t* chunk_bytes) {
    DeviceFastqCache& c = *this;
    {
        FastqTables tables;
        std::memset(&tables, 0, sizeof(tables));
        tables.plan = device_plan;
        fastq_fill_tables(tables);
        for(int s = 0; s < 2; ++s) {
            for(int32_t f = 0; f < n_feed; ++f) {
                FastqFeedChunk& at = tables.chunk[s][f];
                at.raw = c.raw[s][f];
                at.newline = c.newline[s][f];
                at.newline_capacity = static_cast<uint32_t>(chunk_bytes[f] / 8 + 1024);
                at.block_count = c.block_count[s][f];
                at.total = c.total + s * FASTQ_MAX_FEED + f;
            }
        }
        if(c.tables == nullptr) { device_check(hipMalloc(reinterpret_cast<void**>(&c.tables), sizeof(FastqTables)), "hipMalloc"); }
        device_check(hipMemcpy(c.tables, &tables, sizeof(FastqTables), hipMemcpyHostToDevice), "hipMemcpy");
    }
}

Transcoder::OnDevice Transcoder::run_on_device(uint64_t& incoming_count, uint64_t& incoming_pf_count, const DeviceText* text) {
    Impl& m = *impl_;
    const JobConfig& job = m.set.job();
    const TranscodePlan& plan = m.plan;
    const int32_t n_feed = static_cast<int32_t>(plan.input_feed.size());
    if(m.set.device() < 0 || m.classifier || std::getenv("PHQ_HOST_FASTQ") != nullptr) { return OnDevice::NOT_TAKEN; }
    if(job.enable_quality_control && (text == nullptr || !text->quality_control)) { return OnDevice::NOT_TAKEN; }
    const bool writes = text != nullptr && !text->sinks.empty();
    const bool counts = text != nullptr && text->quality_control;
    FastqPlan device_plan;
    if(!device_fastq_plan(job, plan, device_plan)) { return OnDevice::NOT_TAKEN; }
    std::vector<PlainFile> file;
    if(!open_device_inputs(plan, file)) { return OnDevice::NOT_TAKEN; }
    {
        // a wave's share of the reads, by what their text takes in its LDS
        double bytes_per_read = 0;
        for(int32_t f = 0; f < n_feed; ++f) { bytes_per_read += file[static_cast<size_t>(f)].bytes_per_record * device_plan.feed_resolution[f]; }
        fastq_plan_waves(device_plan, bytes_per_read);
    }
    device_check(hipSetDevice(m.set.device()), "hipSetDevice");
    if(!m.set.device_fastq_cache) { m.set.device_fastq_cache = std::make_shared<DeviceFastqCache>(); }
    DeviceFastqCache& c = *static_cast<DeviceFastqCache*>(m.set.device_fastq_cache.get());
    if(c.stream == nullptr) { device_check(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking), "hipStreamCreate"); }

    // records per round: the batch capacity, in chunks of at most 192 MiB of text per file
    const int64_t capacity = std::max<int64_t>(1 << 12, std::min<int64_t>(m.capacity, int64_t(1) << 22));
    c.prepare_rounds(capacity, job);
    if(text != nullptr) { c.prepare_records(*text, capacity, job, plan.channel.size(), counts); }
    size_t chunk_bytes[FASTQ_MAX_FEED];
    for(int32_t f = 0; f < n_feed; ++f) {
        const double wanted = static_cast<double>(capacity) * device_plan.feed_resolution[f] * file[static_cast<size_t>(f)].bytes_per_record * 1.03 + 65536.0;      // a read is `resolution` records of an interleaved file
        chunk_bytes[f] = static_cast<size_t>(std::min(wanted, 192.0 * 1048576.0));
        if(file[static_cast<size_t>(f)].zipped == nullptr) { chunk_bytes[f] = std::min(chunk_bytes[f], file[static_cast<size_t>(f)].size + 1); }
        chunk_bytes[f] = (chunk_bytes[f] + 4095) & ~static_cast<size_t>(4095);
        c.prepare_feed(f, chunk_bytes[f]);
    }
    c.prepare_tables(device_plan, n_feed, chunk_bytes);
    hipStream_t stream = c.stream;
    // the counter buffer as it is now: a run that turns out to need the host reader starts over from it
    device_check(hipMemcpyAsync(c.counters_before, m.set.stats_device_pointer(), static_cast<size_t>(job.stats_words) * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync");
    device_check(hipMemsetAsync(c.irregular, 0, sizeof(uint32_t), stream), "hipMemsetAsync");
    device_check(hipMemsetAsync(c.flagged, 0, sizeof(unsigned long long), stream), "hipMemsetAsync");
    int slot = 0;
    int buffer = 0;                                  // which pinned buffer of a feed this round's bytes are in
    bool ahead_valid[FASTQ_MAX_FEED] = {false};
    size_t ahead_start[FASTQ_MAX_FEED] = {0}, ahead_bytes[FASTQ_MAX_FEED] = {0};
    size_t source_bytes[FASTQ_MAX_FEED] = {0};
    const int threads = host_threads();
    std::future<void> ahead_task;                    // reads the next round's bytes while this one is on the device; declared after what it touches
    // PHQ_TRANSCODE_TRACE=1: where the calling thread waited, summed over the rounds, on stderr at the end of the run
    const bool trace = std::getenv("PHQ_TRANSCODE_TRACE") != nullptr;
    double waited[6] = {0, 0, 0, 0, 0, 0};           // the read ahead, the plan, the decode, the sizes, the output buffer, the last round's text
    uint64_t rounds = 0;
    auto clock_now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double began = clock_now();
    auto timed = [&](int which, auto&& wait) { const double t = clock_now(); wait(); waited[which] += clock_now() - t; };
    auto report = [&]() {
        if(!trace) { return; }
        std::fprintf(stderr, "device path: %llu rounds, %.1f ms; waited (ms) for the read ahead %.1f, the plan %.1f, the decode %.1f, the sizes %.1f, the output buffer %.1f, the last text %.1f\n",
                     static_cast<unsigned long long>(rounds), (clock_now() - began) * 1e3, waited[0] * 1e3, waited[1] * 1e3, waited[2] * 1e3, waited[3] * 1e3, waited[4] * 1e3, waited[5] * 1e3);
    };
    uint64_t reads = 0;
    unsigned long long flagged_so_far = 0;          // a run that writes: the reads of the finished rounds whose filter flag is set
    // a run that writes: the text of the rounds leaves in order on threads of its own, two rounds in flight
    std::shared_future<void> drained;               // the round before this one is in the file
    std::shared_future<void> buffer_drained[2];     // ... the round that used text_host[b]
    int text_buffer = 0;
    // Without text nothing has left the process: the counters go back to what they were and the host reader takes the run
    // from its start.  With text the finished rounds are in the file: the host reader takes over at the bytes they end at.
    // what the rounds of this run left in the device's quality control accumulators, into the set's (multiplex.cpp:405-411)
    auto merge_quality_control = [&]() {
        if(!counts || c.qc_device == nullptr) { return; }
        device_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
        uint32_t overflow = 0;
        device_check(hipMemcpy(&overflow, c.qc_overflow, sizeof(overflow), hipMemcpyDeviceToHost), "hipMemcpy");
        if(overflow) { throw InternalError("quality control: an output segment longer than the bound its tokens give"); }
        std::vector<unsigned long long> words(c.qc_cells * qc_cell_words(c.qc_cycles));
        device_check(hipMemcpy(words.data(), c.qc_device, words.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost), "hipMemcpy");
        const QcState state = quality_control_of_cells(words, c.qc_cells, c.qc_cycles, static_cast<int32_t>(plan.channel.size()), job.output_segment_cardinality);
        m.set.quality_control_merge(state);
        (void)hipFree(c.qc_device); c.qc_device = nullptr; c.qc_cycles = 0;
    };
    auto give_up = [&]() {
        (void)hipStreamSynchronize(stream);
        if(ahead_task.valid()) { ahead_task.wait(); }
        if(text == nullptr) {
            (void)hipMemcpy(m.set.stats_device_pointer(), c.counters_before, static_cast<size_t>(job.stats_words) * sizeof(uint64_t), hipMemcpyDeviceToDevice);
            return OnDevice::NOT_TAKEN;
        }
        if(drained.valid()) { drained.get(); }
        merge_quality_control();
        incoming_count += reads;
        incoming_pf_count += reads - flagged_so_far;
        for(int32_t f = 0; f < n_feed; ++f) { m.stream[static_cast<size_t>(f)]->start_at(file[static_cast<size_t>(f)].position); }
        ++device_resumed;
        return OnDevice::RESUMED;
    };
    // gzip feeds inflate into the pinned buffer of a round: behind what the round before left at its front, up to the round's bytes
    std::atomic<bool> zipped_failed{false};
    auto fill_zipped = [&](int into) {
        parallel_ranges(n_feed, 1, n_feed, [&](int64_t first_feed, int64_t last_feed, int) {
            for(int64_t f = first_feed; f < last_feed; ++f) {
                PlainFile& p = file[static_cast<size_t>(f)];
                if(p.zipped == nullptr) { continue; }
                uint8_t* const to = c.pinned[into][f] + p.offset;
                const size_t want = chunk_bytes[f] - 1;
                if(!p.head.empty() && p.held < want) {             // text inflated while the file was looked at (or set aside by a round that took little)
                    const size_t take = std::min(p.head.size(), want - p.held);
                    std::memcpy(to + p.held, p.head.data(), take);
                    p.held += take;
                    p.head.erase(p.head.begin(), p.head.begin() + static_cast<std::ptrdiff_t>(take));
                }
                while(!p.ended && p.head.empty() && p.held < want) {
                    int got;
                    try {
                        const size_t room = std::min<size_t>(want - p.held, size_t(1) << 30);
                        got = p.members ? static_cast<int>(p.members->read(to + p.held, room)) : p.stream ? static_cast<int>(p.stream->read(to + p.held, room))
                                        : gzread(p.zipped, to + p.held, static_cast<unsigned>(room));
                    } catch(const std::exception&) { got = -1; }       // a member that does not inflate to what it states: the host reader takes the file
                    if(got < 0) { zipped_failed = true; return; }
                    if(got == 0) { p.ended = true; break; }
                    p.held += static_cast<size_t>(got);
                }
            }
        });
    };
    std::future<void> zipped_ahead;
    struct JoinZipped { std::future<void>& task; ~JoinZipped() { if(task.valid()) { task.wait(); } } } join_zipped{zipped_ahead};      // (whichever way the run leaves: the task holds references)
    bool any_zipped = false;
    for(const auto& p : file) { any_zipped = any_zipped || p.zipped != nullptr; }
    while(true) {
        if(ahead_task.valid()) { timed(0, [&]() { ahead_task.get(); }); }
        bool any_left = false;
        FastqFeedChunk chunk[FASTQ_MAX_FEED];
        {
            // gzip feeds: what the round before left is at the front of the pinned buffer; the feeds inflate side by side — from the
            // second round on AHEAD, beside the round before on the device (zipped_ahead below): nothing is left to do here then
            if(zipped_ahead.valid()) { timed(0, [&]() { zipped_ahead.get(); }); }
            for(auto& p : file) { p.held += p.ahead_filled; p.ahead_filled = 0; }
            fill_zipped(buffer);
            if(zipped_failed) { return give_up(); }
        }
        for(int32_t f = 0; f < n_feed; ++f) {
            PlainFile& p = file[static_cast<size_t>(f)];
            const size_t left = p.zipped ? (p.ended ? p.held : p.held + 1) : p.size - p.position;
            size_t bytes = p.zipped ? p.held : std::min(left, chunk_bytes[f] - 1);
            any_left = any_left || left > 0;
            // the file's bytes into a pinned buffer, on every host thread — unless the read ahead of the round before already
            // holds them: that one started a little before where this round could begin at the earliest
            uint8_t* to = c.pinned[buffer][f];
            std::atomic<bool> failed{false};
            auto read_into = [&](uint8_t* target, size_t from_position, size_t count) {
                parallel_ranges(static_cast<int64_t>((count + (1 << 20) - 1) >> 20), 1, threads, [&](int64_t first, int64_t last, int) {
                    const size_t from = static_cast<size_t>(first) << 20;
                    const size_t until = std::min(count, static_cast<size_t>(last) << 20);
                    size_t done = from;
                    while(done < until) {
                        const ssize_t got = ::pread(p.descriptor, target + done, until - done, static_cast<off_t>(from_position + done));
                        if(got <= 0) { failed = true; return; }
                        done += static_cast<size_t>(got);
                    }
                });
            };
            if(p.zipped) {
                to += p.offset;                                     // inflated above
            } else if(ahead_valid[f] && p.position >= ahead_start[f] && p.position + bytes <= ahead_start[f] + ahead_bytes[f]) {
                to = c.pinned[buffer][f] + (p.position - ahead_start[f]);
            } else {
                read_into(to, p.position, bytes);
                if(failed) { return give_up(); }
            }
            source_bytes[f] = p.zipped ? 0 : bytes;                 // (nothing is read ahead of a gzip stream)
            if(bytes == left && bytes > 0 && to[bytes - 1] != '\n') { to[bytes++] = '\n'; }      // the last line of a file may lack its newline
            // (the copies of the feeds on streams of their own, beside the last kernel of the round before, were measured: no
            // gain that stands out of the box to box spread of the link's rate under the host's own memory traffic)
            hipStream_t const up = stream;
            device_check(hipMemcpyAsync(c.raw[slot][f], to, bytes, hipMemcpyHostToDevice, up), "hipMemcpyAsync");
            device_check(hipMemsetAsync(c.raw[slot][f] + bytes, 0, 16, up), "hipMemsetAsync");      // zero padded to whole units of 16 bytes
            chunk[f].raw = c.raw[slot][f];
            chunk[f].bytes = static_cast<uint32_t>(bytes);
            chunk[f].newline = c.newline[slot][f];
            chunk[f].newline_capacity = static_cast<uint32_t>(chunk_bytes[f] / 8 + 1024);
            chunk[f].block_count = c.block_count[slot][f];
            chunk[f].total = c.total + slot * FASTQ_MAX_FEED + f;
            device_check(static_cast<hipError_t>(fastq_index(chunk[f], up)), "fastq_index");
        }
        if(!any_left) { break; }
        device_check(static_cast<hipError_t>(fastq_plan_round(chunk, n_feed, device_plan.feed_resolution, static_cast<uint32_t>(capacity), c.round_device, stream)), "fastq_plan_round");
        device_check(hipMemcpyAsync(c.round_host, c.round_device, sizeof(FastqRound), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
        // While the round is on the device the host reads AHEAD, on threads of its own: where the next round starts is only
        // known once this one is planned (it takes whole records), but never before 7/8 of this round's bytes — a round's
        // bytes are sized for the capacity plus 3 % — so the next round's bytes are read from there on into the other
        // pinned buffer, and the round finds them at whatever offset its start turns out to have.  (Read between this
        // round's upload and its plan only, the files' bytes — 5.5 ms a round of 1 M reads out of the page cache — and
        // the round's kernels took turns: 7 ms a round report only, 155 Mreads/s.)
        const int other = buffer ^ 1;
        struct Ahead { size_t start, count; } planned[FASTQ_MAX_FEED];
        for(int32_t f = 0; f < n_feed; ++f) {
            const PlainFile& p = file[static_cast<size_t>(f)];
            ahead_valid[f] = false;
            planned[f].start = p.position + (source_bytes[f] - source_bytes[f] / 8);
            planned[f].count = 0;
            if(source_bytes[f] < (size_t(1) << 20) || planned[f].start >= p.size) { continue; }      // a small file, or its end: nothing to gain
            planned[f].count = std::min(p.size - planned[f].start, chunk_bytes[f] - 1 + source_bytes[f] / 8 + 4096);
        }
        ahead_task = std::async(std::launch::async, [&file, &c, &ahead_valid, &ahead_start, &ahead_bytes, n_feed, other, threads, planned]() {
            for(int32_t f = 0; f < n_feed; ++f) {
                const size_t start = planned[f].start, count = planned[f].count;
                if(count == 0) { continue; }
                const int descriptor = file[static_cast<size_t>(f)].descriptor;
                uint8_t* const target = c.pinned[other][f];
                std::atomic<bool> failed{false};
                parallel_ranges(static_cast<int64_t>((count + (1 << 20) - 1) >> 20), 1, threads, [&](int64_t first, int64_t last, int) {
                    const size_t from = static_cast<size_t>(first) << 20;
                    const size_t until = std::min(count, static_cast<size_t>(last) << 20);
                    size_t done = from;
                    while(done < until) {
                        const ssize_t got = ::pread(descriptor, target + done, until - done, static_cast<off_t>(start + done));
                        if(got <= 0) { failed = true; return; }
                        done += static_cast<size_t>(got);
                    }
                });
                if(failed) { continue; }
                ahead_valid[f] = true;
                ahead_start[f] = start;
                ahead_bytes[f] = count;
            }
        });
        // gzip feeds likewise: the stream goes on where this round's text ended whatever the round takes of it, so the next round's text
        // is inflated NOW — beside this round's upload, its plan, its kernels — into the other pinned buffer, behind room for what this
        // round will turn out to have left (an eighth of a round's bytes; a round that leaves more sets the text aside, below).  A round
        // of gzip input was inflate THEN device; now it is the longer of the two but for the last eighth, inflated when the round starts.
        if(any_zipped && !zipped_failed) {
            zipped_ahead = std::async(std::launch::async, [&file, &c, &chunk_bytes, &zipped_failed, n_feed, other]() {
                parallel_ranges(n_feed, 1, n_feed, [&](int64_t first_feed, int64_t last_feed, int) {
                    for(int64_t f = first_feed; f < last_feed; ++f) {
                        PlainFile& p = file[static_cast<size_t>(f)];
                        if(p.zipped == nullptr || p.ended || !p.head.empty()) { continue; }
                        const size_t reserve = chunk_bytes[f] / 8 + 4096;
                        uint8_t* const to = c.pinned[other][f] + reserve;
                        const size_t want = chunk_bytes[f] - 1 - reserve;
                        size_t filled = 0;
                        while(filled < want) {
                            int got;
                            try {
                                const size_t room = std::min<size_t>(want - filled, size_t(1) << 30);
                                got = p.members ? static_cast<int>(p.members->read(to + filled, room)) : p.stream ? static_cast<int>(p.stream->read(to + filled, room))
                                                : gzread(p.zipped, to + filled, static_cast<unsigned>(room));
                            } catch(const std::exception&) { got = -1; }
                            if(got < 0) { zipped_failed = true; break; }
                            if(got == 0) { p.ended = true; break; }
                            filled += static_cast<size_t>(got);
                        }
                        p.ahead_filled = filled;
                    }
                });
            });
        }
        timed(1, [&]() { device_check(hipStreamSynchronize(stream), "hipStreamSynchronize"); });       // this round's pinned buffers are free again, the round is known
        ++rounds;
        buffer = other;
        const FastqRound round = *c.round_host;
        if(round.irregular) { return give_up(); }
        if(round.records == 0) {
            // not one whole record in a chunk: a record longer than a chunk, a truncated file, files of different lengths —
            // the host reader says which
            return give_up();
        }
        size_t round_began_at[FASTQ_MAX_FEED];
        for(int32_t f = 0; f < n_feed; ++f) {
            PlainFile& p = file[static_cast<size_t>(f)];
            round_began_at[f] = p.position;
            if(p.zipped) {
                // what the round did not take goes in FRONT of what is being inflated into the next round's buffer (`buffer` is that one
                // by now), into the room kept for it
                const size_t consumed = std::min<size_t>(round.consumed[f], p.held);          // (the newline the last line lacked is not held)
                const size_t left_over = p.held - consumed;
                const size_t reserve = chunk_bytes[f] / 8 + 4096;
                const uint8_t* const from = c.pinned[buffer ^ 1][f] + p.offset + consumed;
                if(left_over <= reserve) {
                    std::memcpy(c.pinned[buffer][f] + reserve - left_over, from, left_over);
                    p.offset = reserve - left_over;
                } else {
                    // the round took less than seven eighths of its bytes (records longer than the file's first ones said): what was
                    // inflated ahead is set aside and follows when there is room again
                    if(zipped_ahead.valid()) { zipped_ahead.get(); }
                    p.head.insert(p.head.end(), c.pinned[buffer][f] + reserve, c.pinned[buffer][f] + reserve + p.ahead_filled);
                    p.ahead_filled = 0;
                    std::memcpy(c.pinned[buffer][f], from, left_over);
                    p.offset = 0;
                }
                p.held = left_over;
                p.position += consumed;
                continue;
            }
            p.position += round.consumed[f];
            if(p.position > p.size) { p.position = p.size; }                      // the newline the last line lacked
        }
        device_check(static_cast<hipError_t>(fastq_decode(c.tables, device_plan, slot, round.records, c.capacity, c.base, c.quality, c.flag, c.irregular, c.flagged,
                                                          c.keep, c.keep_block, c.kept_device, counts ? c.longest : nullptr, stream)), "fastq_decode");
        uint32_t decoded = round.records;
        if(text != nullptr) {
            // A run that writes learns now whether the round holds anything out of the ordinary: its reads are neither
            // counted nor written then, and the host reader continues from where the round began.
            device_check(hipMemcpyAsync(&c.status_host->irregular, c.irregular, sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
            device_check(hipMemcpyAsync(&c.status_host->flagged, c.flagged, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
            if(device_plan.filtering) { device_check(hipMemcpyAsync(&c.status_host->kept, c.kept_device, sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync"); }
            if(counts) { device_check(hipMemcpyAsync(c.status_host->longest, c.longest, FASTQ_MAX_FEED * sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync"); }
            timed(2, [&]() { device_check(hipStreamSynchronize(stream), "hipStreamSynchronize"); });
            if(c.status_host->irregular) {
                for(int32_t f = 0; f < n_feed; ++f) { file[static_cast<size_t>(f)].position = round_began_at[f]; }
                return give_up();
            }
            if(device_plan.filtering) { decoded = c.status_host->kept; }
        } else if(device_plan.filtering) {
            // how many reads the pre-filters kept: the rows the planes hold
            device_check(hipMemcpyAsync(c.kept_host, c.kept_device, sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
            device_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
            decoded = *c.kept_host;
        }
        if(decoded > 0) { m.set.classify_device(c.capacity, 0, decoded, c.base, c.quality, c.flag, c.result, c.word, stream); }
        RecordRound record;
        record.plan = c.record_plan;
        record.result = c.result;
        record.word = c.word;
        record.size = c.record_size;
        record.block = c.record_block;
        record.total = c.record_total;
        record.text = nullptr;
        record.quality_control = nullptr;
        record.cycles = 0;
        if(counts && decoded > 0) {
            // The quality control accumulators of the round's reads (record_quality_control).  A cell holds as many cycles as
            // the longest output segment can have: every token at most the longest of its input segment so far (the decode
            // kernel keeps that), a corrected barcode at most the job's nucleotides; the cells grow when reads grow.
            const int32_t cycles = quality_control_cycles(text->plan, c.status_host->longest, job.nucleotide_width);
            if(cycles > c.qc_cycles) {
                unsigned long long* grown = nullptr;
                const size_t words = c.qc_cells * qc_cell_words(cycles);
                device_check(hipMalloc(reinterpret_cast<void**>(&grown), words * sizeof(unsigned long long)), "hipMalloc");
                device_check(hipMemsetAsync(grown, 0, words * sizeof(unsigned long long), stream), "hipMemsetAsync");
                if(c.qc_device != nullptr) {
                    for(size_t cell = 0; cell < c.qc_cells; ++cell) {
                        device_check(hipMemcpyAsync(grown + cell * qc_cell_words(cycles), c.qc_device + cell * qc_cell_words(c.qc_cycles),
                                                    qc_cell_words(c.qc_cycles) * sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync");
                    }
                    device_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
                    (void)hipFree(c.qc_device);
                }
                c.qc_device = grown;
                c.qc_cycles = cycles;
            }
            record.quality_control = c.qc_device;
            record.cycles = c.qc_cycles;
            device_check(static_cast<hipError_t>(record_quality_control(c.tables, device_plan, slot, round.records, c.capacity, c.base, c.quality, c.keep, c.keep_block,
                                                                        record, c.qc_overflow, stream)), "record_quality_control");
        }
        if(writes && decoded > 0) {
            // the records of the round: sizes, places, text (record_device.hpp), then off the device on the copy stream while
            // the next round comes in on this one.  One sink: its records are the text; several: every sink's records are a
            // part of the text, in the sinks' order (sink_base says where each part starts).
            const int b = text_buffer;
            const int32_t n_sink = static_cast<int32_t>(text->sinks.size());
            const int32_t n_out = text->plan.n_out;
            record.piece_size = c.piece_size;
            record.sink_block = c.sink_block;
            record.sink_base = c.sink_base;
            if(buffer_drained[b].valid()) { timed(4, [&]() { buffer_drained[b].get(); }); }        // text_host[b] and sink_base_host[b] are in the files
            unsigned long long* const base_of_sink = c.sink_base_host[b];
            if(n_sink == 1) {
                device_check(static_cast<hipError_t>(record_sizes(c.tables, device_plan, slot, round.records, c.capacity, c.base, c.quality, c.keep, c.keep_block, record, stream)), "record_sizes");
                device_check(hipMemcpyAsync(&base_of_sink[1], c.record_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
                base_of_sink[0] = 0;
            } else {
                device_check(static_cast<hipError_t>(record_sizes_by_sink(c.tables, device_plan, slot, round.records, c.capacity, c.base, c.quality, c.keep, c.keep_block, record,
                                                                          n_out, n_sink, stream)), "record_sizes_by_sink");
                device_check(hipMemcpyAsync(base_of_sink, c.sink_base, static_cast<size_t>(n_sink + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
            }
            timed(3, [&]() { device_check(hipStreamSynchronize(stream), "hipStreamSynchronize"); });
            const size_t bytes = static_cast<size_t>(base_of_sink[n_sink]);
            if(bytes > 0) {
                if(c.text_device_room[b] < bytes) {
                    device_check(hipStreamSynchronize(c.copy_stream), "hipStreamSynchronize");      // the copy of two rounds ago reads the old one
                    (void)hipFree(c.text_device[b]); c.text_device[b] = nullptr; c.text_device_room[b] = 0;
                    const size_t room = bytes + bytes / 8 + (size_t(1) << 20);
                    device_check(hipMalloc(reinterpret_cast<void**>(&c.text_device[b]), room), "hipMalloc");
                    c.text_device_room[b] = room;
                }
                device_check(hipStreamWaitEvent(stream, c.copied[b], 0), "hipStreamWaitEvent");       // text_device[b] has left (two rounds ago)
                record.text = c.text_device[b];
                if(n_sink == 1) {
                    device_check(static_cast<hipError_t>(record_write(c.tables, device_plan, slot, round.records, c.capacity, c.base, c.quality, c.keep, c.keep_block, record, stream)), "record_write");
                } else {
                    device_check(static_cast<hipError_t>(record_write_by_sink(c.tables, device_plan, slot, round.records, c.capacity, c.base, c.quality, c.keep, c.keep_block, record,
                                                                              n_out, n_sink, stream)), "record_write_by_sink");
                }
                device_check(hipEventRecord(c.written[b], stream), "hipEventRecord");
                if(c.text_host_room[b] < bytes) {
                    if(c.text_host[b]) { (void)hipHostFree(c.text_host[b]); c.text_host[b] = nullptr; c.text_host_room[b] = 0; }
                    const size_t room = bytes + bytes / 8 + (size_t(1) << 20);
                    device_check(hipHostMalloc(reinterpret_cast<void**>(&c.text_host[b]), room, hipHostMallocDefault), "hipHostMalloc");
                    c.text_host_room[b] = room;
                }
                device_check(hipStreamWaitEvent(c.copy_stream, c.written[b], 0), "hipStreamWaitEvent");
                device_check(hipMemcpyAsync(c.text_host[b], c.text_device[b], bytes, hipMemcpyDeviceToHost, c.copy_stream), "hipMemcpyAsync");
                device_check(hipEventRecord(c.copied[b], c.copy_stream), "hipEventRecord");
                // Into the files, after the round before.  A large part for a regular file goes through a mapping on several host
                // threads; gzip sinks are compressed a member per piece, the sinks (or the pieces of a single one) side by side.
                std::vector<OutputSink*> to(static_cast<size_t>(n_sink));
                for(int32_t k = 0; k < n_sink; ++k) { to[static_cast<size_t>(k)] = &m.sink[static_cast<size_t>(text->sinks[static_cast<size_t>(k)])]; }
                const char* const from = c.text_host[b];
                hipEvent_t const arrived = c.copied[b];
                const int device = m.set.device();
                const std::shared_future<void> before = drained;
                drained = std::async(std::launch::async, [to, from, base_of_sink, arrived, device, before, threads]() {
                    if(before.valid()) { before.get(); }
                    device_check(hipSetDevice(device), "hipSetDevice");
                    device_check(hipEventSynchronize(arrived), "hipEventSynchronize");
                    write_round_text(to, from, base_of_sink, threads);
                }).share();
                buffer_drained[b] = drained;
                text_buffer ^= 1;
                device_record_bytes += bytes;
            }
        }
        if(text != nullptr) { flagged_so_far = c.status_host->flagged; }
        reads += round.records;
        ++device_rounds;
        device_reads += round.records;
        slot ^= 1;
    }
    if(text != nullptr) {
        timed(5, [&]() { if(drained.valid()) { drained.get(); } });
        device_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
        merge_quality_control();
        incoming_count += reads;
        incoming_pf_count += reads - flagged_so_far;
        report();
        return OnDevice::DONE;
    }
    uint32_t irregular = 0;
    unsigned long long flagged = 0;
    device_check(hipMemcpyAsync(&irregular, c.irregular, sizeof(irregular), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
    device_check(hipMemcpyAsync(&flagged, c.flagged, sizeof(flagged), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
    device_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if(irregular) { return give_up(); }
    incoming_count += reads;                                                      // transcode.cpp:304-309
    incoming_pf_count += reads - flagged;
    report();
    return OnDevice::DONE;
}

void Transcoder::run(uint64_t& incoming_count, uint64_t& incoming_pf_count) {
    Impl& m = *impl_;
    const double run_begin = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const JobConfig& job = m.set.job();
    const TranscodePlan& plan = m.plan;
    const int32_t n_segment = static_cast<int32_t>(plan.input_segment.size());
    const int32_t n_feed = static_cast<int32_t>(plan.input_feed.size());
    const int32_t n_sink = static_cast<int32_t>(m.sink.size());
    const int32_t leader = job.leading_segment_index;
    const bool illumina_comment = job.platform == "ILLUMINA" || job.platform == "ELEMENT";       // fastq.h:176-194

    // the read group of every sample barcode: decode_tag_id_by_index (classifier.cpp:79-98)
    std::vector<std::string> rg_by_index;
    std::string header = "@HD\tVN:1.0\tSO:unknown\tGO:query\n";                                  // atom.cpp:454-472, the values of transcode.cpp:1661-1667
    for(const auto& d : job.decoder) {
        if(d.kind == Kind::SAMPLE) {
            rg_by_index.push_back(d.undetermined.id);
            for(const auto& b : d.codec) { rg_by_index.push_back(b.id); }
            // load_output (transcode.cpp:1679-1700): every read group of the sample decoder on every output
            auto put_read_group = [&](const BarcodeConfig& b) {
                if(b.id.empty()) { return; }
                header += "@RG";
                for(const auto& field : b.read_group) { header += "\t" + field.first + ":" + field.second; }
                header += "\n";
            };
            put_read_group(d.undetermined);
            for(const auto& b : d.codec) { put_read_group(b); }
        }
    }
    header += "@PG\tID:pheniqs_amd\tPN:pheniqs_amd\tVN:1\n";
    if(plan.sam_header) {
        for(auto& sink : m.sink) {
            if(sink.feed.format == FileFormat::BAM) {
                // "BAM\1", the same header text, no reference sequences — in BGZF blocks of its own (hts.h:96-130 hands the header to htslib)
                const std::string binary = bam_header(header);
                ByteBuffer blocks;
                char* const to = blocks.reserve(fast_deflate::bgzf_bound(binary.size()));
                const size_t packed = fast_deflate::bgzf_blocks(reinterpret_cast<const uint8_t*>(binary.data()), binary.size(), reinterpret_cast<uint8_t*>(to), fast_deflate::bgzf_bound(binary.size()));
                if(packed == 0) { throw InternalError("BGZF"); }
                sink.write(to, packed);
                continue;
            }
            if(sink.feed.format != FileFormat::SAM) { continue; }
            if(sink.feed.gzip) { ByteBuffer member; gzip_member(header.data(), header.size(), sink.feed.level, member); sink.write(member.data.get(), member.size); }
            else { sink.write(header.data(), header.size()); }
        }
    }

    // barcode index -> ordinal in tag_array order, per decoder
    std::vector<std::vector<int>> ordinal_by_index(job.decoder.size());
    for(size_t k = 0; k < job.decoder.size(); ++k) {
        int32_t largest = 0;
        for(const auto& b : job.decoder[k].codec) { largest = std::max(largest, b.index); }
        ordinal_by_index[k].assign(static_cast<size_t>(largest) + 1, -1);
        for(size_t b = 0; b < job.decoder[k].codec.size(); ++b) {
            int& slot = ordinal_by_index[k][static_cast<size_t>(job.decoder[k].codec[b].index)];
            if(slot < 0) { slot = static_cast<int>(b); }                        // the first of equals, as the scan did
        }
    }

    // NullFeed (reference feed.h:105; transcode.cpp:1634,1753): a channel whose every URL is /dev/null takes its reads
    // without a record being assembled; when no channel writes anything the whole formatting step is skipped
    std::vector<char> channel_silent(plan.channel.size(), 0);
    bool all_silent = true;
    for(size_t c = 0; c < plan.channel.size(); ++c) {
        bool silent = true;
        for(const int32_t f : plan.channel[c].feed_by_segment) { silent = silent && m.sink[static_cast<size_t>(f)].null; }
        channel_silent[c] = silent ? 1 : 0;
        all_silent = all_silent && silent;
    }
    // MODE_AUTO (device_types.hpp): a run that writes SAM records of a small job classifies in the literal kernel's arithmetic, so
    // that XB / XM / XC print the reference's own digits (read.h:187-199); FASTQ records carry no posterior and stay fast
    bool writes_sam = false;
    for(const auto& sink : m.sink) { writes_sam = writes_sam || (!sink.null && (sink.feed.format == FileFormat::SAM || sink.feed.format == FileFormat::BAM)); }
    struct ExactRecords {
        std::vector<DecoderSet*> sets;
        ~ExactRecords() { for(DecoderSet* set : sets) { set->set_records_exact(false); } }
    } exact_records;
    if(writes_sam && m.set.literal_cells_per_read() <= RECORD_LITERAL_CELLS) {
        exact_records.sets.push_back(&m.set);
        exact_records.sets.insert(exact_records.sets.end(), m.copy.begin(), m.copy.end());
        for(DecoderSet* set : exact_records.sets) { set->set_records_exact(true); }
    }
    // nothing is written: the FASTQ text is decoded on the device when it can be (plain files, no pre-filters); a file that
    // holds anything out of the ordinary sends the whole run through the host reader below
    if(all_silent && !job.enable_quality_control) {
        if(run_on_device(incoming_count, incoming_pf_count, nullptr) == OnDevice::DONE) { return; }
    } else {
        // Every record into one uncompressed file, or none at all but the quality control accumulators wanted: records and
        // accumulators come from the device too (record_device.hpp).  An input that turns out of the ordinary part way
        // leaves the rest of the run to the host reader and writer below.
        DeviceText text;
        if(plan_device_text(job, plan, m.sink, rg_by_index, ordinal_by_index, text)) {
            text.quality_control = job.enable_quality_control;
            if(run_on_device(incoming_count, incoming_pf_count, &text) == OnDevice::DONE) {
                for(auto& sink : m.sink) { sink.close(); }
                return;
            }
        }
    }
    const DecoderConfig* const multiplexer = job.multiplexing_decoder >= 0 ? &job.decoder[static_cast<size_t>(job.multiplexing_decoder)] : nullptr;
    const std::vector<int>* const multiplexer_ordinal = multiplexer ? &ordinal_by_index[static_cast<size_t>(job.multiplexing_decoder)] : nullptr;

    // f1: one reader thread per input feed (the reference has one producer thread per feed, feed.h:309-314); the
    // next batch is read while this one is staged, classified and written
    std::vector<SegmentBatch>& ahead = m.ahead;          // torn down with the other buffers (Impl::~Impl)
    ahead.resize(m.batch.size());
    // decoding threads per feed: the feeds are read concurrently while the previous batch is being formatted; the
    // readers are the critical path (measured, 2 feeds on 16 host threads: 4 threads per feed 16.2 Mreads/s, 8: 19.6)
    int32_t parse_threads = std::max(1, host_threads() / std::max(1, n_feed));
    if(const char* text = std::getenv("PHQ_PARSE_THREADS")) { parse_threads = std::max(1, std::atoi(text)); }      // measurement aid
    // what each feed's batches must hold: BAM codes where a decoder token reads one of its segments (stage_reads),
    // printable text where the template does; a template-only feed (the biological reads) is never turned into codes
    std::vector<FeedNeeds> needs(static_cast<size_t>(n_feed), FeedNeeds{false, false, job.output_phred_offset, illumina_comment});
    for(const auto& d : job.decoder) {
        for(const auto& t : d.transform) {
            if(t.input_segment_index >= 0 && t.input_segment_index < n_segment) { needs[static_cast<size_t>(plan.input_segment[static_cast<size_t>(t.input_segment_index)].first)].codes = true; }
        }
    }
    for(const auto& t : job.template_transform) {
        if(all_silent && !job.enable_quality_control) { break; }                 // nothing is written or counted: no feed needs its printable text
        if(t.input_segment_index >= 0 && t.input_segment_index < n_segment) { needs[static_cast<size_t>(plan.input_segment[static_cast<size_t>(t.input_segment_index)].first)].text = true; }
    }
    // the first batches are short ones, doubling up to the capacity: nothing else can start before the first batch is read
    // (a full first batch of a million records cost a third of a 4 M read report only run), and the staging planes are laid
    // out for the capacity whatever a batch holds
    int64_t requested = std::min<int64_t>(m.capacity, int64_t(1) << 16);     // records asked of the batch in work
    int64_t requested_ahead = requested;                                      // ... and of the one being read
    auto start_readers = [&](std::vector<SegmentBatch>& into, int64_t records) {
        std::vector<std::thread> readers;
        for(int32_t f = 0; f < n_feed; ++f) {
            readers.emplace_back(read_segment_batch, std::ref(*m.stream[static_cast<size_t>(f)]), records * plan.input_feed[static_cast<size_t>(f)].resolution,
                                 job.input_phred_offset, parse_threads, needs[static_cast<size_t>(f)], std::ref(into[static_cast<size_t>(f)]));
        }
        return readers;
    };
    {
        auto readers = start_readers(m.batch, requested);
        for(auto& t : readers) { t.join(); }
    }
    const double first_read_end = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    // the complement of a printed base, for `~` tokens of the template over printable text (iupac.h:107-124)
    char complement_text[256];
    char canonical_text[256];              // BamToAmbiguousAscii of AsciiToAmbiguousBam: what the reference prints for what it read
    for(int c = 0; c < 256; ++c) {
        complement_text[c] = bam_to_ascii(reverse_complement_bam(ascii_to_bam(static_cast<uint8_t>(c))));
        canonical_text[c] = bam_to_ascii(ascii_to_bam(static_cast<uint8_t>(c)));
    }
    const int workers = host_threads();
    // quality control accumulators (qc.hpp): one private copy per record thread, merged into the set's at the end of the run
    const bool quality_control = job.enable_quality_control;
    std::vector<QcState> quality_by_part(quality_control ? static_cast<size_t>(workers) : 0);
    for(auto& q : quality_by_part) { q.shape(static_cast<int32_t>(plan.channel.size()), job.output_segment_cardinality); }
    std::vector<ByteBuffer> part_text(static_cast<size_t>(workers) * static_cast<size_t>(n_sink));
    std::vector<ByteBuffer> part_packed(part_text.size());
    // the tokens of every output segment, the widest tag block, the flag column per (output segment, QC fail)
    const int32_t n_out = job.output_segment_cardinality;
    std::vector<std::vector<int32_t>> token_of_output(static_cast<size_t>(n_out));
    for(size_t t = 0; t < job.template_transform.size(); ++t) { token_of_output[static_cast<size_t>(job.template_transform[t].output_segment_index)].push_back(static_cast<int32_t>(t)); }
    size_t rg_widest = 0;
    for(const auto& id : rg_by_index) { rg_widest = std::max(rg_widest, id.size()); }
    const int32_t barcode_capacity = job.nucleotide_width + 8;
    const size_t tag_capacity = rg_widest + 8 * static_cast<size_t>(barcode_capacity) + 256;
    std::vector<std::string> flag_text(static_cast<size_t>(n_out) * 2);
    for(int32_t o = 0; o < n_out; ++o) {
        for(int failed = 0; failed < 2; ++failed) {
            int flag = 4 | 8;                                                 // BAM_FUNMAP | BAM_FMUNMAP (read.h:134-135)
            if(n_out > 1) { flag |= 1; }
            if(o == 0) { flag |= 64; }
            if(n_out > 1 && o == n_out - 1) { flag |= 128; }
            if(failed) { flag |= 512; }
            std::string& text = flag_text[static_cast<size_t>(o) * 2 + static_cast<size_t>(failed)];
            text = "\t" + std::to_string(flag) + "\t*\t0\t0\t*\t*\t0\t0\t";
            // FI and TC go in front of the other tags when there are more than two segments (auxiliary.cpp:324-329)
        }
    }
    std::vector<SegmentView> view(static_cast<size_t>(n_segment));
    // PHQ_TRANSCODE_TRACE=1: wall time per phase on stderr at the end of the run
    const bool trace = std::getenv("PHQ_TRANSCODE_TRACE") != nullptr;
    double phase[6] = {0, 0, 0, 0, 0, 0};      // validate, stage, classify, format, write, wait for the readers
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double mark = now();
    auto lap = [&](int k) { const double t = now(); phase[k] += t - mark; mark = t; };
    // N copies of the decoders behind one pull loop (reference transcode.cpp:1769-1775: one TranscodingDecoder per thread, the
    // reads dealt to whichever pulls next): batch b goes to copy b mod N — each copy on a device of its own —, records leave in
    // input order because batches are written in the order they were read, and the copies' counters are added into this
    // set's when the input ends (transcode.cpp:1816-1818 -> selector.cpp:68-77)
    std::vector<DecoderSet*> taker{&m.set};
    taker.insert(taker.end(), m.copy.begin(), m.copy.end());
    size_t batch_number = 0;
    while(true) {
        for(const auto& b : m.batch) { if(!b.error.empty()) { throw_error(b.error_code, b.error); } }
        for(int32_t f = 0; f < n_feed; ++f) {
            if(m.batch[static_cast<size_t>(f)].size() % plan.input_feed[static_cast<size_t>(f)].resolution != 0) {
                throw SequenceError("interleaved input " + plan.input_feed[static_cast<size_t>(f)].path + " ends inside a read");
            }
        }
        for(int32_t s = 0; s < n_segment; ++s) {
            const auto& where = plan.input_segment[static_cast<size_t>(s)];
            view[static_cast<size_t>(s)].bind(m.batch[static_cast<size_t>(where.first)], plan.input_feed[static_cast<size_t>(where.first)].resolution, where.second);
        }
        const int64_t n_read = view[0].size();
        for(const auto& v : view) {
            if(v.size() != n_read) { throw SequenceError("input files hold different numbers of records"); }
        }
        if(n_read == 0) { break; }
        const bool last_batch = n_read < requested;
        std::vector<std::thread> readers;
        struct Joiner { std::vector<std::thread>& threads; ~Joiner() { for(auto& t : threads) { if(t.joinable()) { t.join(); } } } } joiner{readers};
        if(!last_batch) {
            requested_ahead = std::min<int64_t>(m.capacity, requested * 4);
            readers = start_readers(ahead, requested_ahead);
        }
        // Read::validate (read.h:253-264): every segment of a read carries the same identifier
        if(n_segment > 1) {
            try {
                parallel_ranges(n_read, 1 << 14, host_threads(), [&](int64_t first, int64_t last, int) {
                    const SegmentView& a = view[0];
                    for(int32_t s = 1; s < n_segment; ++s) {
                        const SegmentView& b = view[static_cast<size_t>(s)];
                        for(int64_t i = first; i < last; ++i) {
                            const size_t la = a.name_size(i);
                            const size_t lb = b.name_size(i);
                            if(la != lb || std::memcmp(a.name(i), b.name(i), la) != 0) {
                                throw SequenceError("read out of sync " + std::string(b.name(i), lb) + " and " + std::string(a.name(i), la));
                            }
                        }
                    }
                });
            } catch(const Error&) {
                throw;
            } catch(const std::exception& e) {                 // from a worker thread: the text with its class prefix
                std::string text = e.what();
                const std::string prefix = "Sequence error : ";
                if(text.compare(0, prefix.size(), prefix) == 0) { text.erase(0, prefix.size()); }
                throw SequenceError(text);
            }
        }
        std::vector<const uint8_t*> code(static_cast<size_t>(n_segment)), quality(static_cast<size_t>(n_segment));
        std::vector<const int64_t*> offset(static_cast<size_t>(n_segment));
        std::vector<const int32_t*> length(static_cast<size_t>(n_segment));
        for(int32_t s = 0; s < n_segment; ++s) {
            code[static_cast<size_t>(s)] = view[static_cast<size_t>(s)].feed->code.data();
            quality[static_cast<size_t>(s)] = view[static_cast<size_t>(s)].feed->quality.data();
            offset[static_cast<size_t>(s)] = view[static_cast<size_t>(s)].offsets();
            length[static_cast<size_t>(s)] = view[static_cast<size_t>(s)].lengths();
        }
        lap(0);
        int64_t n_kept = 0;
        m.set.stage_reads(n_read, n_segment, code.data(), quality.data(), offset.data(), length.data(),
                          view[static_cast<size_t>(leader)].qcfails(), m.capacity, m.base, m.quality, m.flag, m.kept.data(), &n_kept,
                          &incoming_count, &incoming_pf_count);
        lap(1);
        if(m.classifier) { m.classifier(m.capacity, n_kept, m.base, m.quality, m.flag, m.result, m.word); }
        else { taker[batch_number++ % taker.size()]->classify_host(m.capacity, n_kept, m.base, m.quality, m.flag, m.result, m.word); }
        lap(2);

        // f2: records, a contiguous range of the batch per thread, written in order
        for(auto& text : part_text) { text.clear(); }
        for(auto& text : part_packed) { text.clear(); }
        parallel_ranges((all_silent && !quality_control) ? 0 : n_kept, 1 << 13, workers, [&](int64_t first_kept, int64_t last_kept, int part) {
        QcState* const counted = quality_control ? &quality_by_part[static_cast<size_t>(part)] : nullptr;
        std::vector<uint8_t> counted_code, counted_quality;
        ByteBuffer* const text_of_sink = &part_text[static_cast<size_t>(part) * static_cast<size_t>(n_sink)];
        Text raw[3], corrected[3];
        for(auto& t : raw) { t.allocate(barcode_capacity); }
        for(auto& t : corrected) { t.allocate(barcode_capacity); }
        std::vector<char> tags(tag_capacity);
        ByteBuffer line;                          // a BAM sink's record as the SAM line it is made from (bam_record.hpp)
        // a token's bases: BAM codes and Phred values (a corrected barcode) or printable text (an input segment)
        struct Piece { const uint8_t* code; const uint8_t* quality; const char* text; const char* quality_text; int32_t size; bool reverse_complement; bool lent; int32_t quality_shift; };
        std::vector<Piece> piece(job.template_transform.size());
        for(int64_t k = first_kept; k < last_kept; ++k) {
            const int64_t r = m.kept[static_cast<size_t>(k)];
            if(multiplexer != nullptr && multiplexer->result_slot >= 0) {
                // the channel is known from the multiplexing decoder's record alone: a silent one needs nothing else
                const DevResult& res = m.result[static_cast<int64_t>(multiplexer->result_slot) * m.capacity + k];
                size_t early = 0;
                if(res.index > 0 && static_cast<size_t>(res.index) < multiplexer_ordinal->size()) { early = static_cast<size_t>((*multiplexer_ordinal)[static_cast<size_t>(res.index)] + 1); }
                if(early >= plan.channel.size()) { early = 0; }
                if(!quality_control && (channel_silent[early] || plan.channel[early].feed_by_segment.empty())) { continue; }
            } else if(!quality_control && channel_silent[0]) { continue; }
            uint32_t distance[3] = {0, 0, 0};
            double confidence[3] = {1, 1, 1};                               // read.h:176-183
            for(auto& t : raw) { t.clear(); }
            for(auto& t : corrected) { t.clear(); }
            const std::string* rg = nullptr;
            size_t channel_index = 0;                                        // classifier.h:83-85
            for(const auto& d : job.decoder) {
                const int kind = static_cast<int>(d.kind);
                if(d.plane_stride == 0) { continue; }
                const uint8_t* oc = m.base + m.capacity * d.plane_prefix + k * d.plane_stride;
                const uint8_t* oq = m.quality + m.capacity * d.plane_prefix + k * d.plane_stride;
                const DevResult* res = d.result_slot >= 0 ? m.result + static_cast<int64_t>(d.result_slot) * m.capacity + k : nullptr;
                // the decoded barcode: ordinal in tag_array order, -1 = unclassified
                int ordinal = -1;
                if(res != nullptr && res->index > 0) {
                    const auto& by_index = ordinal_by_index[static_cast<size_t>(&d - job.decoder.data())];
                    if(static_cast<size_t>(res->index) < by_index.size()) { ordinal = by_index[static_cast<size_t>(res->index)]; }
                }
                if(d.multiplexing_classifier) { channel_index = static_cast<size_t>(ordinal + 1); }
                for(int32_t s = 0; s < d.segment_cardinality; ++s) {
                    const int32_t so = d.segment_offset[static_cast<size_t>(s)];
                    int32_t len = d.barcode_length[static_cast<size_t>(s)];
                    while(len > 0 && oc[so + len - 1] == 0 && oq[so + len - 1] == 0) { --len; }      // observed length
                    // append_to_raw_*_barcode (read.h:275-278,301-305,328-332)
                    raw[kind].append(oc + so, oq + so, len);
                    if(d.closed() && len > 0) {
                        // append_to_corrected_*_barcode_sequence -> append_corrected (sequence.h:382-398), S9 quirk kept:
                        // the "was this base corrected" test reads the observation at the DESTINATION length
                        Text& to = corrected[kind];
                        const int32_t dest = to.size;
                        const uint8_t* const expected = ordinal >= 0 ? d.barcode_code.data() + static_cast<size_t>(ordinal) * static_cast<size_t>(d.nucleotide_cardinality) + static_cast<size_t>(so) : nullptr;
                        for(int32_t i = 0; i < len; ++i) {
                            const uint8_t fixed = expected != nullptr ? expected[i] : 0;
                            const uint8_t probe = (dest + i) < len ? oc[so + dest + i] : 0;
                            to.push(fixed, (probe == fixed || fixed == 0) ? oq[so + i] : static_cast<uint8_t>(d.corrected_quality));
                        }
                    }
                }
                if(res == nullptr) { continue; }                             // naive / passthrough: raw barcode only
                if(d.kind == Kind::SAMPLE) { rg = &rg_by_index[static_cast<size_t>(ordinal + 1)]; }
            }
            // the per type accumulators as the wrappers leave them (pamld.cpp:133-180, mdd.cpp:96-138): phq_summarize's loop
            summarize_read(job, m.result, m.capacity, k, distance, confidence);

            // Multiplexer::push -> Channel::push (multiplex.h:198-240)
            const bool qcfail = (m.word[k] >> 31) != 0;
            if(channel_index >= plan.channel.size()) { channel_index = 0; }
            const TranscodePlan::Channel& channel = plan.channel[channel_index];
            const bool written = !(channel.feed_by_segment.empty() || (channel.filter_outgoing_qc_fail && qcfail)) && !channel_silent[channel_index];
            if(!written && !quality_control) { continue; }

            // TemplateRule::apply (transform.h:190-226): where every token's bases are
            for(size_t t = 0; t < job.template_transform.size(); ++t) {
                const TransformConfig& token = job.template_transform[t];
                const uint8_t* fc = nullptr; const uint8_t* fq = nullptr; const char* ft = nullptr; const char* fqt = nullptr; int32_t flen;
                bool lent = false; int32_t quality_shift = 0;
                if(token.input_segment_index < 0) {
                    const Text& from = corrected[token.input_segment_index == -1 ? 0 : token.input_segment_index == -2 ? 2 : 1];
                    fc = from.code.data(); fq = from.quality.data(); flen = from.size;
                } else {
                    const SegmentView& v = view[static_cast<size_t>(token.input_segment_index)];
                    const size_t at = v.at(r);
                    if(v.feed->source != nullptr) {                       // lent by a mapped file: made printable while it is copied
                        ft = v.feed->source + v.feed->source_sequence[at]; fqt = v.feed->source + v.feed->source_quality[at];
                        lent = true; quality_shift = v.feed->quality_shift;
                    } else { ft = v.feed->text.data() + v.feed->offset[at]; fqt = v.feed->quality_text.data() + v.feed->offset[at]; }
                    flen = v.feed->length[at];
                }
                const int32_t start = token.absolute_start(flen);
                const int32_t end = token.absolute_end(flen);
                piece[t] = Piece{fc ? fc + start : nullptr, fq ? fq + start : nullptr, ft ? ft + start : nullptr, fqt ? fqt + start : nullptr,
                                 end - start > 0 ? end - start : 0, token.reverse_complement, lent, quality_shift};
            }
            // Channel::push counts the read whether or not it is written (multiplex.h:219-221): BAM codes and Phred values of
            // every output segment, as the reference's Segment holds them
            if(counted != nullptr) {
                for(int32_t o = 0; o < n_out; ++o) {
                    counted_code.clear();
                    counted_quality.clear();
                    for(const int32_t t : token_of_output[static_cast<size_t>(o)]) {
                        const Piece& p = piece[static_cast<size_t>(t)];
                        for(int32_t i = 0; i < p.size; ++i) {
                            const int32_t from = p.reverse_complement ? p.size - i - 1 : i;
                            uint8_t code;
                            uint8_t quality;
                            if(p.text != nullptr) {
                                code = ascii_to_bam(static_cast<uint8_t>(p.text[from]));
                                quality = static_cast<uint8_t>(p.quality_text[from] - (p.lent ? job.input_phred_offset : job.output_phred_offset));
                            } else { code = p.code[from]; quality = p.quality[from]; }
                            counted_code.push_back(p.reverse_complement ? reverse_complement_bam(code) : code);
                            counted_quality.push_back(quality);
                        }
                    }
                    counted->at(static_cast<int32_t>(channel_index), o).increment(counted_code.data(), counted_quality.data(), static_cast<int32_t>(counted_code.size()));
                }
            }
            if(!written) { continue; }

            // Read::flush (read.h:187-237) + Auxiliary::encode order (auxiliary.cpp:327-354): the same tags on every segment
            char* tag_end = tags.data();
            if(rg != nullptr && !rg->empty()) { tag_end = put_literal(tag_end, "\tRG:Z:"); tag_end = put_bytes(tag_end, rg->data(), rg->size()); }
            if(!raw[0].empty()) { tag_end = put_literal(tag_end, "\tBC:Z:"); tag_end = put_sequence(tag_end, raw[0]); tag_end = put_literal(tag_end, "\tQT:Z:"); tag_end = put_quality(tag_end, raw[0], 33); }
            if(confidence[0] > 0 && confidence[0] < 1) { tag_end = put_literal(tag_end, "\tXB:f:"); tag_end = put_float_g(tag_end, static_cast<float>(1.0 - confidence[0])); }
            if(!corrected[1].empty()) { tag_end = put_literal(tag_end, "\tRX:Z:"); tag_end = put_sequence(tag_end, corrected[1]); tag_end = put_literal(tag_end, "\tQX:Z:"); tag_end = put_quality(tag_end, corrected[1], 33); }
            if(!raw[1].empty()) { tag_end = put_literal(tag_end, "\tOX:Z:"); tag_end = put_sequence(tag_end, raw[1]); tag_end = put_literal(tag_end, "\tBZ:Z:"); tag_end = put_quality(tag_end, raw[1], 33); }
            if(confidence[1] > 0 && confidence[1] < 1) { tag_end = put_literal(tag_end, "\tXM:f:"); tag_end = put_float_g(tag_end, static_cast<float>(1.0 - confidence[1])); }
            if(!corrected[2].empty()) { tag_end = put_literal(tag_end, "\tCB:Z:"); tag_end = put_sequence(tag_end, corrected[2]); }
            if(!raw[2].empty()) { tag_end = put_literal(tag_end, "\tCR:Z:"); tag_end = put_sequence(tag_end, raw[2]); tag_end = put_literal(tag_end, "\tCY:Z:"); tag_end = put_quality(tag_end, raw[2], 33); }
            if(confidence[2] > 0 && confidence[2] < 1) { tag_end = put_literal(tag_end, "\tXC:f:"); tag_end = put_float_g(tag_end, static_cast<float>(1.0 - confidence[2])); }
            const size_t tag_size = static_cast<size_t>(tag_end - tags.data());

            const SegmentView& lead = view[static_cast<size_t>(leader)];
            const char* const name = lead.name(r);
            const size_t name_size = lead.name_size(r);
            for(int32_t o = 0; o < n_out; ++o) {
                const int32_t to_sink = channel.feed_by_segment[static_cast<size_t>(o)];
                const bool as_bam = m.sink[static_cast<size_t>(to_sink)].feed.format == FileFormat::BAM;
                if(as_bam) { line.clear(); }
                ByteBuffer& buffer = as_bam ? line : text_of_sink[to_sink];
                const auto& tokens = token_of_output[static_cast<size_t>(o)];
                size_t bases = 0;
                for(const int32_t t : tokens) { bases += static_cast<size_t>(piece[static_cast<size_t>(t)].size); }
                char* to = buffer.reserve(name_size + 2 * bases + tag_size + static_cast<size_t>(raw[0].size) + 128);
                auto put_bases = [&](char* at) {
                    for(const int32_t t : tokens) {
                        const Piece& p = piece[static_cast<size_t>(t)];
                        if(p.text != nullptr) {
                            if(p.reverse_complement) { for(int32_t i = 0; i < p.size; ++i) { at[i] = complement_text[static_cast<uint8_t>(p.text[p.size - i - 1])]; } at += p.size; }
                            else if(p.lent) { for(int32_t i = 0; i < p.size; ++i) { at[i] = canonical_text[static_cast<uint8_t>(p.text[i])]; } at += p.size; }
                            else { at = put_bytes(at, p.text, static_cast<size_t>(p.size)); }
                        } else if(p.reverse_complement) { for(int32_t i = 0; i < p.size; ++i) { at[i] = BAM_TO_ASCII[reverse_complement_bam(p.code[p.size - i - 1]) & 0xf]; } at += p.size; }
                        else { at = put_sequence(at, p.code, p.size); }
                    }
                    return at;
                };
                auto put_qualities = [&](char* at) {
                    for(const int32_t t : tokens) {
                        const Piece& p = piece[static_cast<size_t>(t)];
                        if(p.quality_text != nullptr) {
                            if(p.reverse_complement) { for(int32_t i = 0; i < p.size; ++i) { at[i] = static_cast<char>(p.quality_text[p.size - i - 1] + p.quality_shift); } at += p.size; }
                            else if(p.quality_shift != 0) { for(int32_t i = 0; i < p.size; ++i) { at[i] = static_cast<char>(p.quality_text[i] + p.quality_shift); } at += p.size; }
                            else { at = put_bytes(at, p.quality_text, static_cast<size_t>(p.size)); }
                        } else if(p.reverse_complement) { for(int32_t i = 0; i < p.size; ++i) { at[i] = static_cast<char>(p.quality[p.size - i - 1] + job.output_phred_offset); } at += p.size; }
                        else { at = put_quality(at, p.quality, p.size, job.output_phred_offset); }
                    }
                    return at;
                };
                if(m.sink[static_cast<size_t>(to_sink)].feed.format == FileFormat::FASTQ) {
                    // FastqRecord::decode(segment) + encode (fastq.h:79-88,144-171), comment fastq.h:176-194
                    *to++ = '@';
                    to = put_bytes(to, name, name_size);
                    *to++ = ' ';
                    if(illumina_comment) {
                        to = put_unsigned(to, static_cast<uint32_t>(o + 1));
                        to = qcfail ? put_literal(to, ":Y:0:") : put_literal(to, ":N:0:");
                        to = put_sequence(to, raw[0]);
                    }
                    *to++ = '\n';
                    to = put_bases(to);
                    to = put_literal(to, "\n+\n");
                    to = put_qualities(to);
                    *to++ = '\n';
                    buffer.commit(to);
                    continue;
                }
                const std::string& columns = flag_text[static_cast<size_t>(o) * 2 + (qcfail ? 1 : 0)];
                to = put_bytes(to, name, name_size);
                to = put_bytes(to, columns.data(), columns.size());
                if(bases == 0) { to = put_literal(to, "*\t*"); }
                else {
                    to = put_bases(to);
                    *to++ = '\t';
                    to = put_qualities(to);
                }
                if(n_out > 2) {
                    to = put_literal(to, "\tFI:i:"); to = put_unsigned(to, static_cast<uint32_t>(o + 1));
                    to = put_literal(to, "\tTC:i:"); to = put_unsigned(to, static_cast<uint32_t>(n_out));
                }
                to = put_bytes(to, tags.data(), tag_size);
                *to++ = '\n';
                buffer.commit(to);
                if(as_bam) {
                    // the same fields as a BAM record; XB / XM / XC from their values, not from their six printed digits (auxiliary.cpp:335-349)
                    const float posterior[3] = {static_cast<float>(1.0 - confidence[0]), static_cast<float>(1.0 - confidence[1]), static_cast<float>(1.0 - confidence[2])};
                    append_bam_record(text_of_sink[to_sink], line.data.get(), line.size - 1, posterior);
                }
            }
        }
        for(int32_t f = 0; f < n_sink; ++f) {
            ByteBuffer& packed = part_packed[static_cast<size_t>(part) * static_cast<size_t>(n_sink) + static_cast<size_t>(f)];
            if(m.sink[static_cast<size_t>(f)].feed.format == FileFormat::BAM && !text_of_sink[f].empty()) {
                // the part's records in BGZF blocks (a record may end in the next block: section 4.1)
                packed.clear();
                const size_t bound = fast_deflate::bgzf_bound(text_of_sink[f].size);
                char* const to = packed.reserve(bound);
                const size_t made = fast_deflate::bgzf_blocks(reinterpret_cast<const uint8_t*>(text_of_sink[f].data.get()), text_of_sink[f].size, reinterpret_cast<uint8_t*>(to), bound);
                if(made == 0) { throw InternalError("BGZF"); }
                packed.size = made;
            } else if(m.sink[static_cast<size_t>(f)].feed.gzip && !text_of_sink[f].empty()) {
                gzip_member(text_of_sink[f].data.get(), text_of_sink[f].size, m.sink[static_cast<size_t>(f)].feed.level, packed);
            }
        }
        });
        lap(3);
        // the parts in order; on a regular file every part goes to its place on its own thread
        {
            std::vector<char*> place(part_text.size(), nullptr);
            bool concurrent = false;
            for(int32_t f = 0; f < n_sink; ++f) {
                OutputSink& sink = m.sink[static_cast<size_t>(f)];
                if(sink.null) { continue; }
                size_t total = 0;
                for(int part = 0; part < workers; ++part) {
                    const size_t at = static_cast<size_t>(part) * static_cast<size_t>(n_sink) + static_cast<size_t>(f);
                    total += ((sink.feed.gzip || sink.feed.format == FileFormat::BAM) ? part_packed[at] : part_text[at]).size;
                }
                char* window = (sink.seekable && total >= (1u << 20)) ? sink.map_more(total) : nullptr;
                for(int part = 0; part < workers; ++part) {
                    const size_t at = static_cast<size_t>(part) * static_cast<size_t>(n_sink) + static_cast<size_t>(f);
                    const ByteBuffer& bytes = (sink.feed.gzip || sink.feed.format == FileFormat::BAM) ? part_packed[at] : part_text[at];
                    if(bytes.size == 0) { continue; }
                    if(window != nullptr) { place[at] = window; window += bytes.size; concurrent = true; }
                    else { sink.write(bytes.data.get(), bytes.size); }
                }
            }
            if(concurrent) {
                parallel_ranges(workers, 1, workers, [&](int64_t first_part, int64_t last_part, int) {
                    for(int64_t part = first_part; part < last_part; ++part) {
                        for(int32_t f = 0; f < n_sink; ++f) {
                            const size_t at = static_cast<size_t>(part) * static_cast<size_t>(n_sink) + static_cast<size_t>(f);
                            const ByteBuffer& bytes = (m.sink[static_cast<size_t>(f)].feed.gzip || m.sink[static_cast<size_t>(f)].feed.format == FileFormat::BAM) ? part_packed[at] : part_text[at];
                            if(place[at] != nullptr) { std::memcpy(place[at], bytes.data.get(), bytes.size); }
                        }
                    }
                });
                for(auto& sink : m.sink) { sink.unmap(); }
            }
        }
        lap(4);
        if(last_batch) { break; }
        for(auto& t : readers) { t.join(); }
        lap(5);
        m.batch.swap(ahead);
        requested = requested_ahead;
    }
    for(auto& sink : m.sink) { sink.close(); }
    for(DecoderSet* const copy : m.copy) { m.set.stats_take(*copy); }         // the end of run merge: integer sums, word for word
    if(quality_control) {                                                      // Multiplexer::collect, multiplex.cpp:405-411
        QcState total;
        for(const auto& q : quality_by_part) { total.merge(q); }
        m.set.quality_control_merge(total);
    }
    if(trace) {
        std::fprintf(stderr, "transcode phases (s): first batch read %.3f | validate %.3f stage %.3f classify %.3f format %.3f write %.3f wait for readers %.3f | run %.3f\n",
                     first_read_end - run_begin, phase[0], phase[1], phase[2], phase[3], phase[4], phase[5], now() - run_begin);
    }
}

}  // namespace phq
