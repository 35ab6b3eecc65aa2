// Transcoder: the steps either side of the classification path for FASTQ in / SAM text out
// (SURVEY.md §8f rows f1 and f2), host side C++.
//
//   f1  FASTQ -> batch      FastqRecord::decode (fastq.h:55-78: ASCII -> BAM code, quality - phred offset),
//                           Illumina comment filter flag (fastq.h:253-285), Read::validate name sync
//                           (read.h:253-264); one reader thread per input file; then DecoderSet::stage_reads.
//   f2  results -> records  the per type accumulators and barcode strings the decoder wrappers leave on the
//                           output Read (pamld.cpp:133-180, mdd.cpp:96-138, naive.h:40-45, read.h:269-348),
//                           TemplateRule::apply (transform.h:190-226), Read::flush (read.h:187-237),
//                           Auxiliary::encode tag order (auxiliary.cpp:327-354), SAM text as htslib prints it.
//
// Out of scope here: BAM/CRAM, SAM header atoms (@HD/@RG/@PG), split outputs per channel, QC histograms.
#include "transcoder.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "error.hpp"
#include "host_parallel.hpp"

namespace phq {

namespace {

// one input file, plain or gzip (gzread reads both), 4 line FASTQ records
//
// A batch is read in two steps: the reader thread pulls bytes and indexes line starts until it holds the lines of
// `capacity` records (memchr speed), then the records are decoded — name, Illumina filter flag, ASCII -> BAM codes,
// quality - offset — on a few threads over contiguous record ranges (two passes: sizes, then a prefix sum places
// every thread's output).  The reference decodes under the feed lock, one record at a time (feed.h:320-333).
class FastqStream {
public:
    explicit FastqStream(const std::string& path) : path_(path), begin_(0), end_(0), eof_(false) {
        file_ = gzopen(path.c_str(), "rb");
        if(file_ == nullptr) { throw IOError("could not open " + path + " for reading"); }
        gzbuffer(file_, 1 << 20);
        buffer_.resize(1 << 24);
    }
    ~FastqStream() { if(file_) { gzclose(file_); } }
    FastqStream(const FastqStream&) = delete;
    FastqStream& operator=(const FastqStream&) = delete;

    // the next up to `capacity` records: line[4 i + k] = (offset, size) of line k of record i inside data()
    int64_t index_records(int64_t capacity, std::vector<std::pair<size_t, uint32_t>>& line) {
        line.clear();
        size_t scan = begin_;
        const size_t wanted = static_cast<size_t>(capacity) * 4;
        while(line.size() < wanted) {
            const char* start = buffer_.data() + scan;
            const void* newline = scan < end_ ? std::memchr(start, '\n', end_ - scan) : nullptr;
            if(newline == nullptr) {
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
                for(auto& l : line) { l.first -= shift; }
                continue;
            }
            const size_t size = static_cast<const char*>(newline) - start;
            push(line, scan, size);
            scan += size + 1;
        }
        const size_t records = line.size() / 4;
        if(line.size() % 4 != 0) {
            if(eof_ && scan == end_) { throw SequenceError("truncated FASTQ record in " + path_); }
            line.resize(records * 4);
        }
        consumed_ = records > 0 ? line[records * 4 - 1].first + line[records * 4 - 1].second + 1 : scan;
        if(records * 4 == line.size() && line.size() < wanted) { consumed_ = scan; }      // end of input: trailing blank lines too
        if(consumed_ > end_) { consumed_ = end_; }
        return static_cast<int64_t>(records);
    }
    void release() { begin_ = consumed_; }                 // the indexed records have been decoded
    const char* data() const { return buffer_.data(); }
    const std::string& path() const { return path_; }

private:
    void push(std::vector<std::pair<size_t, uint32_t>>& line, size_t offset, size_t size) {
        if(size > 0 && buffer_[offset + size - 1] == '\r') { --size; }
        if(size == 0 && line.size() % 4 == 0) { return; }                      // blank line between records
        line.emplace_back(offset, static_cast<uint32_t>(size));
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
    std::vector<char> buffer_;
    size_t begin_;
    size_t end_;
    size_t consumed_ = 0;
    bool eof_;
};

struct SegmentBatch {
    std::vector<uint8_t> code;
    std::vector<uint8_t> quality;
    std::vector<int64_t> offset;
    std::vector<int32_t> length;
    std::vector<uint8_t> qcfail;
    std::vector<char> name;              // concatenated, name_offset[i] .. name_offset[i + 1]
    std::vector<int64_t> name_offset;
    std::vector<std::pair<size_t, uint32_t>> line;
    std::string error;
    void clear() {
        code.clear(); quality.clear(); offset.clear(); length.clear(); qcfail.clear(); name.clear();
        name_offset.assign(1, 0);
        error.clear();
    }
    int64_t size() const { return static_cast<int64_t>(length.size()); }
};

void read_segment_batch(FastqStream& stream, int64_t capacity, int32_t phred_offset, int32_t threads, SegmentBatch& out) {
    static const std::vector<uint8_t> ascii_table = []() {                  // AsciiToAmbiguousBam, iupac.h:153-171
        std::vector<uint8_t> table(256);
        for(int c = 0; c < 256; ++c) { table[static_cast<size_t>(c)] = ascii_to_bam(static_cast<uint8_t>(c)); }
        return table;
    }();
    out.clear();
    try {
        const int64_t n = stream.index_records(capacity, out.line);
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
        out.code.resize(static_cast<size_t>(sequence_bytes[static_cast<size_t>(parts)]));
        out.quality.resize(out.code.size());
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
                bool qcfail = false;
                size_t p = name_end < header_size ? name_end + 1 : header_size;
                while(p < header_size && header[p] != ':') { ++p; }
                if(p + 1 < header_size) { qcfail = header[p + 1] == 'Y'; }
                out.qcfail[static_cast<size_t>(i)] = qcfail ? 1 : 0;
                const size_t size = line[static_cast<size_t>(4 * i + 1)].second;
                const char* sequence = data + line[static_cast<size_t>(4 * i + 1)].first;
                const char* quality = data + line[static_cast<size_t>(4 * i + 3)].first;
                out.offset[static_cast<size_t>(i)] = at;
                uint8_t* to_code = out.code.data() + at;
                uint8_t* to_quality = out.quality.data() + at;
                for(size_t k = 0; k < size; ++k) { to_code[k] = ascii_table[static_cast<uint8_t>(sequence[k])]; }
                for(size_t k = 0; k < size; ++k) { to_quality[k] = static_cast<uint8_t>(quality[k] - phred_offset); }
                at += static_cast<int64_t>(size);
            }
        });
        stream.release();
    } catch(const std::exception& e) {
        out.error = e.what();
        if(out.error.empty()) { out.error = "error reading " + stream.path(); }
        out.code.clear(); out.quality.clear(); out.offset.clear(); out.length.clear(); out.qcfail.clear(); out.name.clear();
        out.name_offset.assign(1, 0);
    }
}

struct Text {
    std::vector<uint8_t> code;
    std::vector<uint8_t> quality;
    void clear() { code.clear(); quality.clear(); }
    bool empty() const { return code.empty(); }
    void append(const uint8_t* c, const uint8_t* q, int32_t n) {
        code.insert(code.end(), c, c + n);
        quality.insert(quality.end(), q, q + n);
    }
};

void put_sequence(std::string& out, const std::vector<uint8_t>& code) {
    static const char table[] = "=ACMGRSVTWYHKDBN";                        // BamToAmbiguousAscii, iupac.h:62-79
    const size_t at = out.size();
    out.resize(at + code.size());
    char* to = &out[at];
    for(size_t i = 0; i < code.size(); ++i) { to[i] = table[code[i] & 0xf]; }
}
void put_quality(std::string& out, const std::vector<uint8_t>& quality, int32_t phred_offset) {
    const size_t at = out.size();
    out.resize(at + quality.size());
    char* to = &out[at];
    for(size_t i = 0; i < quality.size(); ++i) { to[i] = static_cast<char>(quality[i] + phred_offset); }
}
void put_float(std::string& out, float value) {
    char scratch[48];
    std::snprintf(scratch, sizeof(scratch), "%g", static_cast<double>(value));     // htslib kputd/ksprintf "%g"
    out += scratch;
}

}  // namespace

struct Transcoder::Impl {
    DecoderSet& set;
    std::vector<std::unique_ptr<FastqStream>> stream;
    std::FILE* output;
    bool owns_output;
    int64_t capacity;
    std::vector<SegmentBatch> batch;
    uint8_t* base;
    uint8_t* quality;
    uint8_t* flag;
    DevResult* result;
    uint32_t* word;
    std::vector<int64_t> kept;
    std::string text;

    Impl(DecoderSet& set) : set(set), output(nullptr), owns_output(false), capacity(0), base(nullptr), quality(nullptr), flag(nullptr), result(nullptr), word(nullptr) {}
    ~Impl() {
        host_free(base); host_free(quality); host_free(flag); host_free(result); host_free(word);
        if(output && owns_output) { std::fclose(output); }
    }
};

Transcoder::Transcoder(DecoderSet& set, const std::vector<std::string>& input, const std::string& sam_output, int64_t batch_capacity) :
    impl_(new Impl(set)) {
    const JobConfig& job = set.job();
    if(job.input_segment_cardinality > 0 && static_cast<int32_t>(input.size()) != job.input_segment_cardinality) {
        throw ConfigurationError("expected " + std::to_string(job.input_segment_cardinality) + " input files");
    }
    if(job.template_transform.empty()) { throw ConfigurationError("the job has no template transform"); }
    for(const auto& path : input) { impl_->stream.emplace_back(new FastqStream(path)); }
    if(sam_output == "-" || sam_output == "/dev/stdout") { impl_->output = stdout; }
    else {
        impl_->output = std::fopen(sam_output.c_str(), "wb");
        impl_->owns_output = true;
        if(impl_->output == nullptr) { throw IOError("could not open " + sam_output + " for writing"); }
    }
    impl_->capacity = batch_capacity;
    impl_->batch.resize(input.size());
    const size_t planes = static_cast<size_t>(batch_capacity) * static_cast<size_t>(std::max(job.plane_width, 4));
    impl_->base = static_cast<uint8_t*>(host_alloc(planes));
    impl_->quality = static_cast<uint8_t*>(host_alloc(planes));
    impl_->flag = static_cast<uint8_t*>(host_alloc(static_cast<size_t>(batch_capacity)));
    impl_->result = static_cast<DevResult*>(host_alloc(static_cast<size_t>(batch_capacity) * static_cast<size_t>(std::max(job.result_count, 1)) * sizeof(DevResult)));
    impl_->word = static_cast<uint32_t*>(host_alloc(static_cast<size_t>(batch_capacity) * sizeof(uint32_t)));
    impl_->kept.resize(static_cast<size_t>(batch_capacity));
}

Transcoder::~Transcoder() { delete impl_; }

void Transcoder::run(uint64_t& incoming_count, uint64_t& incoming_pf_count) {
    Impl& m = *impl_;
    const JobConfig& job = m.set.job();
    const int32_t n_segment = static_cast<int32_t>(m.stream.size());
    const int32_t leader = job.leading_segment_index;

    // the read group of every sample barcode: decode_tag_id_by_index (classifier.cpp:79-98)
    std::vector<std::string> rg_by_index;
    for(const auto& d : job.decoder) {
        if(d.kind == Kind::SAMPLE) {
            rg_by_index.push_back(d.undetermined.id);
            for(const auto& b : d.codec) { rg_by_index.push_back(b.id); }
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

    // f1: one reader thread per input file (the reference has one producer thread per feed, feed.h:309-314); the
    // next batch is read while this one is staged, classified and written
    std::vector<SegmentBatch> ahead(m.batch.size());
    // decoding threads per file: the files are read concurrently and the previous batch is being formatted meanwhile
    const int32_t parse_threads = std::max(1, host_threads() / (2 * std::max(1, n_segment)));
    auto start_readers = [&](std::vector<SegmentBatch>& into) {
        std::vector<std::thread> readers;
        for(int32_t s = 0; s < n_segment; ++s) {
            readers.emplace_back(read_segment_batch, std::ref(*m.stream[static_cast<size_t>(s)]), m.capacity, job.input_phred_offset, parse_threads, std::ref(into[static_cast<size_t>(s)]));
        }
        return readers;
    };
    {
        auto readers = start_readers(m.batch);
        for(auto& t : readers) { t.join(); }
    }
    const int workers = host_threads();
    std::vector<std::string> part_text(static_cast<size_t>(workers));
    // PHQ_TRANSCODE_TRACE=1: wall time per phase on stderr at the end of the run
    const bool trace = std::getenv("PHQ_TRANSCODE_TRACE") != nullptr;
    double phase[6] = {0, 0, 0, 0, 0, 0};      // validate, stage, classify, format, write, wait for the readers
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double mark = now();
    auto lap = [&](int k) { const double t = now(); phase[k] += t - mark; mark = t; };
    while(true) {
        for(const auto& b : m.batch) { if(!b.error.empty()) { throw SequenceError(b.error); } }
        const int64_t n_read = m.batch[0].size();
        for(const auto& b : m.batch) {
            if(b.size() != n_read) { throw SequenceError("input files hold different numbers of records"); }
        }
        if(n_read == 0) { break; }
        const bool last_batch = n_read < m.capacity;
        std::vector<std::thread> readers;
        struct Joiner { std::vector<std::thread>& threads; ~Joiner() { for(auto& t : threads) { if(t.joinable()) { t.join(); } } } } joiner{readers};
        if(!last_batch) { readers = start_readers(ahead); }
        // Read::validate (read.h:253-264): every segment of a read carries the same identifier
        for(int32_t s = 1; s < n_segment; ++s) {
            const SegmentBatch& a = m.batch[0];
            const SegmentBatch& b = m.batch[static_cast<size_t>(s)];
            for(int64_t i = 0; i < n_read; ++i) {
                const int64_t la = a.name_offset[static_cast<size_t>(i) + 1] - a.name_offset[static_cast<size_t>(i)];
                const int64_t lb = b.name_offset[static_cast<size_t>(i) + 1] - b.name_offset[static_cast<size_t>(i)];
                if(la != lb || std::memcmp(a.name.data() + a.name_offset[static_cast<size_t>(i)], b.name.data() + b.name_offset[static_cast<size_t>(i)], static_cast<size_t>(la)) != 0) {
                    throw SequenceError("read out of sync " + std::string(b.name.data() + b.name_offset[static_cast<size_t>(i)], static_cast<size_t>(lb)) + " and " +
                                        std::string(a.name.data() + a.name_offset[static_cast<size_t>(i)], static_cast<size_t>(la)));
                }
            }
        }
        std::vector<const uint8_t*> code(static_cast<size_t>(n_segment)), quality(static_cast<size_t>(n_segment));
        std::vector<const int64_t*> offset(static_cast<size_t>(n_segment));
        std::vector<const int32_t*> length(static_cast<size_t>(n_segment));
        for(int32_t s = 0; s < n_segment; ++s) {
            code[static_cast<size_t>(s)] = m.batch[static_cast<size_t>(s)].code.data();
            quality[static_cast<size_t>(s)] = m.batch[static_cast<size_t>(s)].quality.data();
            offset[static_cast<size_t>(s)] = m.batch[static_cast<size_t>(s)].offset.data();
            length[static_cast<size_t>(s)] = m.batch[static_cast<size_t>(s)].length.data();
        }
        lap(0);
        int64_t n_kept = 0;
        m.set.stage_reads(n_read, n_segment, code.data(), quality.data(), offset.data(), length.data(),
                          m.batch[static_cast<size_t>(leader)].qcfail.data(), m.capacity, m.base, m.quality, m.flag, m.kept.data(), &n_kept,
                          &incoming_count, &incoming_pf_count);
        lap(1);
        m.set.classify_host(m.capacity, n_kept, m.base, m.quality, m.flag, m.result, m.word);
        lap(2);

        // f2: records, a contiguous range of the batch per thread, written in order
        for(auto& text : part_text) { text.clear(); }
        parallel_ranges(n_kept, 1 << 13, workers, [&](int64_t first_kept, int64_t last_kept, int part) {
        std::string& text = part_text[static_cast<size_t>(part)];
        text.reserve(static_cast<size_t>(last_kept - first_kept) * 256);
        Text raw[3], corrected[3], out_segment[8];
        for(int64_t k = first_kept; k < last_kept; ++k) {
            const int64_t r = m.kept[static_cast<size_t>(k)];
            uint32_t distance[3] = {0, 0, 0};
            double confidence[3] = {1, 1, 1};                               // read.h:176-183
            for(auto& t : raw) { t.clear(); }
            for(auto& t : corrected) { t.clear(); }
            const std::string* rg = nullptr;
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
                        const int32_t dest = static_cast<int32_t>(to.code.size());
                        for(int32_t i = 0; i < len; ++i) {
                            const uint8_t fixed = ordinal >= 0 ? d.barcode_code[static_cast<size_t>(ordinal) * static_cast<size_t>(d.nucleotide_cardinality) + static_cast<size_t>(so + i)] : 0;
                            const uint8_t probe = (dest + i) < len ? oc[so + dest + i] : 0;
                            to.code.push_back(fixed);
                            to.quality.push_back((probe == fixed || fixed == 0) ? oq[so + i] : static_cast<uint8_t>(d.corrected_quality));
                        }
                    }
                }
                if(res == nullptr) { continue; }                             // naive / passthrough: raw barcode only
                const bool classified = res->index > 0;
                if(d.algorithm == Algorithm::PAMLD) {
                    if(d.kind == Kind::SAMPLE || classified) {               // pamld.cpp:133-140,149-160,169-180
                        distance[kind] += res->distance;
                        confidence[kind] = confidence[kind] == 1 ? res->confidence : confidence[kind] * res->confidence;
                    } else {
                        confidence[kind] = 0;
                        if(d.kind == Kind::CELLULAR) { distance[kind] = 0; }   // read.h:319-321 adds for molecular
                    }
                } else {                                                     // mdd.cpp:96-138
                    if(d.kind == Kind::SAMPLE || classified) { distance[kind] += res->distance; }
                    else { distance[static_cast<int>(Kind::CELLULAR)] = 0; }
                }
                if(d.kind == Kind::SAMPLE) { rg = &rg_by_index[static_cast<size_t>(ordinal + 1)]; }
            }

            // TemplateRule::apply (transform.h:190-226)
            const int32_t n_out = job.output_segment_cardinality;
            for(int32_t o = 0; o < n_out; ++o) { out_segment[o].clear(); }
            for(const auto& t : job.template_transform) {
                const uint8_t* fc; const uint8_t* fq; int32_t flen;
                if(t.input_segment_index < 0) {
                    const Text& from = corrected[t.input_segment_index == -1 ? 0 : t.input_segment_index == -2 ? 2 : 1];
                    fc = from.code.data(); fq = from.quality.data(); flen = static_cast<int32_t>(from.code.size());
                } else {
                    const SegmentBatch& b = m.batch[static_cast<size_t>(t.input_segment_index)];
                    fc = b.code.data() + b.offset[static_cast<size_t>(r)]; fq = b.quality.data() + b.offset[static_cast<size_t>(r)]; flen = b.length[static_cast<size_t>(r)];
                }
                const int32_t start = t.absolute_start(flen);
                const int32_t end = t.absolute_end(flen);
                Text& to = out_segment[t.output_segment_index];
                if(end - start > 0) {
                    if(t.reverse_complement) {
                        for(int32_t i = 0; i < end - start; ++i) { to.code.push_back(reverse_complement_bam(fc[end - i - 1])); to.quality.push_back(fq[end - i - 1]); }
                    } else { to.append(fc + start, fq + start, end - start); }
                }
            }

            // Read::flush (read.h:187-237) + Auxiliary::encode order (auxiliary.cpp:327-354) + SAM text
            const bool qcfail = (m.word[k] >> 31) != 0;
            const SegmentBatch& lead = m.batch[static_cast<size_t>(leader)];
            for(int32_t o = 0; o < n_out; ++o) {
                int flag = 4 | 8;                                             // BAM_FUNMAP | BAM_FMUNMAP (read.h:134-135)
                if(n_out > 1) { flag |= 1; }
                if(o == 0) { flag |= 64; }
                if(n_out > 1 && o == n_out - 1) { flag |= 128; }
                if(qcfail) { flag |= 512; }
                text.append(lead.name.data() + lead.name_offset[static_cast<size_t>(r)], static_cast<size_t>(lead.name_offset[static_cast<size_t>(r) + 1] - lead.name_offset[static_cast<size_t>(r)]));
                text += '\t'; text += std::to_string(flag); text += "\t*\t0\t0\t*\t*\t0\t0\t";
                if(out_segment[o].empty()) { text += "*\t*"; }
                else {
                    put_sequence(text, out_segment[o].code);
                    text += '\t';
                    put_quality(text, out_segment[o].quality, job.output_phred_offset);
                }
                if(n_out > 2) { text += "\tFI:i:" + std::to_string(o + 1) + "\tTC:i:" + std::to_string(n_out); }
                if(rg != nullptr && !rg->empty()) { text += "\tRG:Z:"; text += *rg; }
                if(!raw[0].empty()) { text += "\tBC:Z:"; put_sequence(text, raw[0].code); text += "\tQT:Z:"; put_quality(text, raw[0].quality, 33); }
                if(confidence[0] > 0 && confidence[0] < 1) { text += "\tXB:f:"; put_float(text, static_cast<float>(1.0 - confidence[0])); }
                if(!corrected[1].empty()) { text += "\tRX:Z:"; put_sequence(text, corrected[1].code); text += "\tQX:Z:"; put_quality(text, corrected[1].quality, 33); }
                if(!raw[1].empty()) { text += "\tOX:Z:"; put_sequence(text, raw[1].code); text += "\tBZ:Z:"; put_quality(text, raw[1].quality, 33); }
                if(confidence[1] > 0 && confidence[1] < 1) { text += "\tXM:f:"; put_float(text, static_cast<float>(1.0 - confidence[1])); }
                if(!corrected[2].empty()) { text += "\tCB:Z:"; put_sequence(text, corrected[2].code); }
                if(!raw[2].empty()) { text += "\tCR:Z:"; put_sequence(text, raw[2].code); text += "\tCY:Z:"; put_quality(text, raw[2].quality, 33); }
                if(confidence[2] > 0 && confidence[2] < 1) { text += "\tXC:f:"; put_float(text, static_cast<float>(1.0 - confidence[2])); }
                text += '\n';
            }
            (void)distance;
        }
        });
        lap(3);
        for(const auto& text : part_text) {
            if(!text.empty() && std::fwrite(text.data(), 1, text.size(), m.output) != text.size()) { throw IOError("error writing the SAM output"); }
        }
        lap(4);
        if(last_batch) { break; }
        for(auto& t : readers) { t.join(); }
        lap(5);
        m.batch.swap(ahead);
    }
    std::fflush(m.output);
    if(trace) {
        std::fprintf(stderr, "transcode phases (s): validate %.3f stage %.3f classify %.3f format %.3f write %.3f wait for readers %.3f\n",
                     phase[0], phase[1], phase[2], phase[3], phase[4], phase[5]);
    }
}

}  // namespace phq
