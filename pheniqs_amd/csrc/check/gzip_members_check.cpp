// MemberInflater (gzip_members.hpp) on the CPU, under ASan + UBSan: gzip_members_check FILE [expected text FILE]
//   exit 0 and "members N blocked B bytes T crc C": the file has several members and inflated to T bytes (compared with the
//   expected text when one is given, read back in pieces of uneven sizes); exit 3: open() said no (one member, not gzip);
//   exit 4: a member did not inflate to what it states.  tests/test_capi_cpu.py drives it.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "../gzip_members.hpp"

int main(int argc, char** argv) {
    if(argc < 2) { return 2; }
    phq::MemberInflater inflater;
    if(!inflater.open(argv[1], 4)) { std::printf("not taken\n"); return 3; }
    std::vector<uint8_t> text;
    size_t piece = 1;
    try {
        while(true) {
            const size_t before = text.size();
            text.resize(before + piece);
            const size_t got = inflater.read(text.data() + before, piece);
            text.resize(before + got);
            if(got == 0) { break; }
            piece = piece * 3 + 7;                       // 1, 10, 37, ... bytes, then whole batches
            if(piece > (size_t(48) << 20)) { piece = 12345; }
        }
    } catch(const phq::Error& error) {
        std::printf("error: %s\n", error.message.c_str());
        return 4;
    }
    if(argc > 2) {
        std::ifstream expected_file(argv[2], std::ios::binary);
        const std::vector<uint8_t> expected((std::istreambuf_iterator<char>(expected_file)), std::istreambuf_iterator<char>());
        if(expected != text) { std::printf("text differs: %zu bytes, expected %zu\n", text.size(), expected.size()); return 5; }
    }
    std::printf("members %zu blocked %d bytes %zu crc %lu fast %llu zlib %llu\n", inflater.members(), inflater.blocked() ? 1 : 0, text.size(), crc32(0L, text.data(), static_cast<uInt>(text.size())),
                static_cast<unsigned long long>(phq::MemberInflater::fast_members().load()), static_cast<unsigned long long>(phq::MemberInflater::zlib_members().load()));
    return 0;
}
