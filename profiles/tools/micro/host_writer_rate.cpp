// The host writer's rate by output format, without a GPU: the transcoder (transcoder.cpp) on a set without a device, the batch
// "classified" by an injected word match (as check/transcoder_check.cpp does), so that what is timed is the reader, the record
// writers (SAM text, BAM records + BGZF, FASTQ) and the file — the part of a run that BAM output adds to.
//
// build (repo root):
//   g++ -std=c++17 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include profiles/tools/micro/host_writer_rate.cpp pheniqs_amd/csrc/check/no_device.cpp \
//       pheniqs_amd/csrc/{transcoder,decoder_set,json,compile,job,phred,stats,qc}.cpp -o /tmp/host_writer_rate -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lz -lpthread -ldl
// usage: host_writer_rate <job.json> <work directory> <segment 0 FASTQ> [<segment 1 FASTQ> ...]
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../../pheniqs_amd/csrc/decoder_set.hpp"
#include "../../../pheniqs_amd/csrc/error.hpp"
#include "../../../pheniqs_amd/csrc/job.hpp"
#include "../../../pheniqs_amd/csrc/transcoder.hpp"

using namespace phq;

int main(int argc, char** argv) {
    if(argc < 4) { std::fprintf(stderr, "usage: host_writer_rate <job.json> <work directory> <fastq> ...\n"); return 2; }
    std::ifstream file(argv[1], std::ios::binary);
    std::stringstream content;
    content << file.rdbuf();
    const std::string work = argv[2];
    std::vector<std::string> input;
    for(int i = 3; i < argc; ++i) { input.push_back(argv[i]); }
    try {
        DecoderSet set(content.str(), -1);
        const JobConfig& job = set.job();
        double classify_seconds = 0;
        auto classify = [&](int64_t capacity, int64_t n, const uint8_t* base, const uint8_t*, const uint8_t* flag, DevResult* result, uint32_t* word) {
            const auto t0 = std::chrono::steady_clock::now();
            for(int64_t i = 0; i < n; ++i) {
                const bool qcfail = flag != nullptr && (flag[i] & 1);
                uint32_t channel = 0;
                for(const auto& d : job.decoder) {
                    if(d.result_slot < 0) { continue; }
                    const uint8_t* const row = base + capacity * d.plane_prefix + i * d.plane_stride;
                    DevResult out;
                    std::memset(&out, 0, sizeof(out));
                    // (a cheap stand-in: the barcode by the row's bytes, nine reads in ten pass)
                    uint32_t h = 0;
                    for(int32_t k = 0; k < d.plane_stride; ++k) { h = h * 31u + row[k]; }
                    const size_t b = h % d.codec.size();
                    out.index = d.codec[b].index;
                    out.status = STATUS_PASS;
                    out.confidence = 0.96875 + 1.0 / static_cast<double>(64 + (i & 63)) / 64.0;
                    out.qcfail = qcfail ? 1 : 0;
                    result[static_cast<int64_t>(d.result_slot) * capacity + i] = out;
                    if(d.multiplexing_classifier) { channel = static_cast<uint32_t>(out.index); }
                }
                word[i] = channel | (qcfail ? 0x80000000u : 0u);
            }
            classify_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        };
        for(const char* name : {"/dev/null", "out.sam", "out.bam", "out.sam", "out.bam"}) {
            const std::string output = name[0] == '/' ? std::string(name) : work + "/" + name;
            uint64_t incoming = 0, incoming_pf = 0;
            classify_seconds = 0;
            const auto t0 = std::chrono::steady_clock::now();
            {
                Transcoder transcoder(set, TranscodePlan::explicit_files(job, input, output), 1 << 20);
                transcoder.set_classifier(classify);
                transcoder.run(incoming, incoming_pf);
            }
            const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("%-10s %8.3f s (%.3f s of it the stand-in classifier)  %6.2f Mreads/s\n", name, seconds, classify_seconds, static_cast<double>(incoming) / seconds / 1e6);
        }
    } catch(const std::exception& e) {
        std::fprintf(stderr, "host_writer_rate: %s\n", e.what());
        return 1;
    }
    return 0;
}
