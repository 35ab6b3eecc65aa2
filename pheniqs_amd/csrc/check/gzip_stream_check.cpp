// GzipStream (gzip_stream.hpp) on the CPU, under ASan + UBSan: gzip_stream_check FILE [expected text FILE | - [block to refuse | -1 [threads]]]
//   exit 0 and "stream bytes T crc C zlib Z": the file inflated to T bytes (compared with the expected text when one is given),
//   read back in pieces of uneven sizes; Z = members handed to zlib part way; exit 3: not a gzip file; exit 4: an IOError
//   (damaged data, a CRC or size that does not hold, a stream that ends early).  tests/test_capi_cpu.py drives it.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "../gzip_stream.hpp"

int main(int argc, char** argv) {
    if(argc < 2) { return 2; }
    if(argc > 3) { phq::fast_inflate::RawStream::refuse_block() = std::atol(argv[3]); }
    phq::GzipStream stream;
    if(!stream.open(argv[1], argc > 4 ? std::atoi(argv[4]) : 1)) { std::printf("not taken\n"); return 3; }
    std::vector<uint8_t> text;
    size_t piece = 1;
    try {
        while(true) {
            const size_t before = text.size();
            text.resize(before + piece);
            const size_t got = stream.read(text.data() + before, piece);
            text.resize(before + got);
            if(got == 0) { break; }
            piece = piece * 3 + 7;                       // 1, 10, 37, ... bytes, then several work buffers at once
            if(piece > (size_t(5) << 20)) { piece = 12345; }
        }
    } catch(const phq::Error& error) {
        std::printf("error: %s\n", error.message.c_str());
        return 4;
    }
    if(argc > 2 && argv[2][0] != '-') {
        std::ifstream expected_file(argv[2], std::ios::binary);
        const std::vector<uint8_t> expected((std::istreambuf_iterator<char>(expected_file)), std::istreambuf_iterator<char>());
        if(expected != text) { std::printf("text differs: %zu bytes, expected %zu\n", text.size(), expected.size()); return 5; }
    }
    std::printf("stream bytes %zu crc %lu zlib %llu chunks %llu dropped %llu\n", text.size(), crc32(0L, text.data(), static_cast<uInt>(text.size())), static_cast<unsigned long long>(phq::GzipStream::through_zlib().load()),
                static_cast<unsigned long long>(phq::GzipStream::chunks_taken().load()), static_cast<unsigned long long>(phq::GzipStream::chunks_dropped().load()));
    return 0;
}
