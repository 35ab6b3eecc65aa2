// The transcoder's reader and record writers (transcoder.cpp: mapped input, 32-byte line scans, raw pointer writers) under
// AddressSanitizer + UBSan (make -C pheniqs_amd/csrc asan).  No GPU code can run under the sanitizers on the target pool, so
// the batch is NOT classified here: the check injects result planes of its own through Transcoder::set_classifier — an
// exact word match per decoder, enough to send classified and unclassified reads, QC failed or not, through every writer —
// and the product path (DecoderSet::classify_host) is never reached.
//
// usage: transcoder_check <job.json> <rounds> <work directory> <segment 0 FASTQ> [<segment 1 FASTQ> ...]
//
// Round 0 runs the files as they are (every record must come out); every later round runs copies with bytes flipped,
// dropped, doubled, lines cut or the tail truncated, alternating SAM text to a file, FASTQ / SAM / BAM by the job's own output
// directives when it has some, and report only (/dev/null).  An Error (the reference's error codes) or a result every time.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../decoder_set.hpp"
#include "../error.hpp"
#include "../job.hpp"
#include "../transcoder.hpp"

using namespace phq;

static std::string slurp(const std::string& path) {
    std::ifstream file(path, std::ios::binary);
    std::stringstream content;
    content << file.rdbuf();
    return content.str();
}

int main(int argc, char** argv) {
    if(argc < 5) { std::fprintf(stderr, "usage: transcoder_check <job.json> <rounds> <work directory> <fastq> ...\n"); return 2; }
    const std::string job_text = slurp(argv[1]);
    const int rounds = std::atoi(argv[2]);
    const std::string work = argv[3];
    std::vector<std::string> original;
    for(int i = 4; i < argc; ++i) { original.push_back(slurp(argv[i])); }
    int failures = 0;
    int errors = 0;
    int results = 0;
    try {
        DecoderSet set(job_text, -1);                          // PHQ_DEVICE_NONE: configuration, staging, statistics
        const JobConfig& job = set.job();
        // result planes by exact word match (the check's own, not a decoder)
        auto classify = [&](int64_t capacity, int64_t n, const uint8_t* base, const uint8_t*, const uint8_t* flag, DevResult* result, uint32_t* word) {
            for(int64_t i = 0; i < n; ++i) {
                bool qcfail = flag != nullptr && (flag[i] & 1);
                uint32_t channel = 0;
                for(const auto& d : job.decoder) {
                    if(d.result_slot < 0) { continue; }
                    const uint8_t* const row = base + capacity * d.plane_prefix + i * d.plane_stride;
                    DevResult out;
                    std::memset(&out, 0, sizeof(out));
                    out.status = STATUS_UNCLASSIFIED;
                    for(size_t b = 0; b < d.codec.size(); ++b) {
                        if(std::memcmp(row, &d.barcode_code[b * static_cast<size_t>(d.nucleotide_cardinality)], static_cast<size_t>(d.nucleotide_cardinality)) == 0) {
                            out.index = d.codec[b].index;
                            out.status = STATUS_PASS;
                            out.confidence = d.algorithm == Algorithm::PAMLD ? 0.96875 + 1.0 / static_cast<double>(64 + (i & 63)) / 64.0 : 0.0;
                            break;
                        }
                    }
                    if(out.status != STATUS_PASS) { qcfail = true; }
                    out.qcfail = qcfail ? 1 : 0;
                    result[static_cast<int64_t>(d.result_slot) * capacity + i] = out;
                    if(d.multiplexing_classifier) { channel = static_cast<uint32_t>(out.index); }
                }
                word[i] = channel | (qcfail ? 0x80000000u : 0u);
            }
        };
        uint64_t state = 0x9e3779b97f4a7c15ull;
        auto next = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
        for(int round = 0; round <= rounds; ++round) {
            std::vector<std::string> input;
            for(size_t f = 0; f < original.size(); ++f) {
                std::string text = original[f];
                if(round > 0 && (next() % original.size() == f || next() % 3 == 0) && !text.empty()) {
                    const int edits = 1 + static_cast<int>(next() % 4);
                    for(int e = 0; e < edits && !text.empty(); ++e) {
                        const size_t at = static_cast<size_t>(next() % text.size());
                        if(round % 2 == 1) {
                            // gentle rounds: a byte of a line replaced by another printable one (a base that is no base, a quality
                            // out of range, a name that no longer matches): the record structure mostly survives
                            size_t line = 0;
                            for(size_t k = 0; k < at; ++k) { line += text[k] == '\n' ? 1 : 0; }
                            const bool whole_record = e % 4 == 3;                     // now and then a name or a separator line too
                            if(text[at] != '\n' && (whole_record || (line & 1u) == 1u)) { text[at] = static_cast<char>(33 + next() % 94); }
                            continue;
                        }
                        switch(next() % 7) {
                            case 0: text[at] = static_cast<char>(next() % 256); break;
                            case 1: text.erase(at, 1 + static_cast<size_t>(next() % 12)); break;
                            case 2: text.insert(at, text.substr(at, 1 + static_cast<size_t>(next() % 40))); break;
                            case 3: text[at] = '\n'; break;
                            case 4: text.resize(at); break;                                              // truncated
                            case 5: { const size_t end = text.find('\n', at); if(end != std::string::npos) { text.erase(at, end - at); } break; }    // a line cut short
                            default: text.insert(at, "\n@"); break;
                        }
                    }
                }
                const std::string path = work + "/segment_" + std::to_string(f) + ".fastq";
                std::ofstream(path, std::ios::binary) << text;
                input.push_back(path);
            }
            const int64_t capacity = round % 3 == 0 ? 1 << 12 : round % 3 == 1 ? 37 : 1 << 16;          // batches smaller and larger than the file
            const std::string output = round % 2 == 0 ? work + "/out.sam" : "/dev/null";
            try {
                uint64_t incoming = 0, incoming_pf = 0;
                // the job's own output directives (a file per channel: FASTQ, gzip, SAM, BAM) over this round's copies of the input
                const bool own_outputs = round % 5 == 4 && job.multiplexing_decoder >= 0 && job.input.size() == input.size();
                TranscodePlan plan = own_outputs ? TranscodePlan::from_job(job) : TranscodePlan::explicit_files(job, input, output);
                if(own_outputs && plan.input_feed.size() == input.size()) {
                    for(size_t f = 0; f < input.size(); ++f) { plan.input_feed[f].path = input[f]; }
                }
                {
                    Transcoder transcoder(set, plan, capacity);
                    transcoder.set_classifier(classify);
                    transcoder.run(incoming, incoming_pf);
                }
                ++results;
                if(round == 0) {
                    // the files as they are: every record comes in, and one SAM record per output segment goes out
                    size_t lines = 0;
                    for(char c : original[0]) { lines += c == '\n' ? 1 : 0; }
                    if(incoming != lines / 4) { std::fprintf(stderr, "round 0: %llu records in, the file holds %zu\n", static_cast<unsigned long long>(incoming), lines / 4); ++failures; }
                    const std::string sam = slurp(output);
                    size_t records = 0;
                    size_t at = 0;
                    while(at < sam.size()) { if(sam[at] != '@') { ++records; } const size_t end = sam.find('\n', at); if(end == std::string::npos) { break; } at = end + 1; }
                    // (the job may filter incoming reads: at most one record per output segment and read that came in)
                    if(records == 0 || records % static_cast<size_t>(job.output_segment_cardinality) != 0 || records > incoming * static_cast<uint64_t>(job.output_segment_cardinality)) { std::fprintf(stderr, "round 0: %zu SAM records for %llu reads\n", records, static_cast<unsigned long long>(incoming)); ++failures; }
                }
            } catch(const Error&) {
                ++errors;
                if(round == 0) { std::fprintf(stderr, "round 0 (undamaged input) raised an error\n"); ++failures; }
            }
        }
        // last, the undamaged files by the job's own directives, so that what the channels' files hold can be read back by the caller
        if(job.multiplexing_decoder >= 0 && job.input.size() == original.size()) {
            TranscodePlan plan = TranscodePlan::from_job(job);
            if(plan.input_feed.size() == original.size()) {
                for(size_t f = 0; f < original.size(); ++f) {
                    const std::string path = work + "/segment_" + std::to_string(f) + ".fastq";
                    std::ofstream(path, std::ios::binary) << original[f];
                    plan.input_feed[f].path = path;
                }
                uint64_t incoming = 0, incoming_pf = 0;
                Transcoder transcoder(set, plan, 1 << 10);
                transcoder.set_classifier(classify);
                transcoder.run(incoming, incoming_pf);
                std::printf("own outputs: %llu records in\n", static_cast<unsigned long long>(incoming));
            }
        }
    } catch(const std::exception& e) {
        std::fprintf(stderr, "transcoder check: %s\n", e.what());
        return 1;
    }
    std::printf("transcoder check: %d runs, %d results, %d errors\n", rounds + 1, results, errors);
    return failures ? 1 : 0;
}
