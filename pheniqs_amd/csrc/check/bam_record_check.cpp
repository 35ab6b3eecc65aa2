// bam_record.hpp + fast_deflate.hpp's BGZF blocks under AddressSanitizer + UBSan, without the transcoder around them: SAM lines
// in (a file), an unaligned BAM file out — the header text given, every line a record, the records in BGZF blocks, the empty block
// at the end — which tests/bam_reader.py reads back.  A line this writer does not make (fewer than eleven fields, a name past 254
// characters, a float tag that states neither bits nor a value the caller holds) is an Error: exit code 3 and its message.
//
// usage: bam_record_check <lines.sam> <out.bam> [<header text file>]
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>

#include "../bam_record.hpp"
#include "../fast_deflate.hpp"

using namespace phq;

static std::string slurp(const char* path) {
    std::ifstream file(path, std::ios::binary);
    std::stringstream content;
    content << file.rdbuf();
    return content.str();
}

int main(int argc, char** argv) {
    if(argc < 3) { std::fprintf(stderr, "usage: bam_record_check <lines.sam> <out.bam> [<header>]\n"); return 2; }
    try {
        const std::string lines = slurp(argv[1]);
        const std::string header = bam_header(argc > 3 ? slurp(argv[3]) : std::string());
        ByteBuffer records, packed;
        append_bam_records(records, lines.data(), lines.size());
        auto pack = [&](const char* data, size_t size) {
            if(size == 0) { return; }
            const size_t bound = fast_deflate::bgzf_bound(size);
            char* const to = packed.reserve(bound);
            const size_t made = fast_deflate::bgzf_blocks(reinterpret_cast<const uint8_t*>(data), size, reinterpret_cast<uint8_t*>(to), bound);
            if(made == 0) { throw InternalError("BGZF"); }
            packed.commit(to + made);
        };
        pack(header.data(), header.size());
        pack(records.data.get(), records.size);
        std::ofstream out(argv[2], std::ios::binary);
        out.write(packed.data.get(), static_cast<std::streamsize>(packed.size));
        out.write(reinterpret_cast<const char*>(fast_deflate::BGZF_EOF), sizeof(fast_deflate::BGZF_EOF));
        std::printf("bam record check: %zu bytes of lines, %zu of records, %zu of blocks\n", lines.size(), records.size, packed.size);
    } catch(const Error& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 3;
    }
    return 0;
}
