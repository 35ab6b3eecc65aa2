// fast_inflate.hpp against zlib, on the CPU (ASan + UBSan in the test suite): fast_inflate_check [speed]
// Every text below is deflated by zlib into a gzip member under every level and strategy and must come back byte for byte;
// damaged members must be refused (false), never answered with other text.  `speed`: rates of both decoders on FASTQ like text.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../fast_inflate.hpp"

static std::vector<uint8_t> member_of(const std::vector<uint8_t>& text, int level, int strategy, int memory) {
    z_stream z;
    std::memset(&z, 0, sizeof z);
    if(deflateInit2(&z, level, Z_DEFLATED, 31, memory, strategy) != Z_OK) { std::abort(); }
    std::vector<uint8_t> out(deflateBound(&z, static_cast<uLong>(text.size())) + 64);
    z.next_in = const_cast<Bytef*>(text.data()); z.avail_in = static_cast<uInt>(text.size());
    z.next_out = out.data(); z.avail_out = static_cast<uInt>(out.size());
    if(deflate(&z, Z_FINISH) != Z_STREAM_END) { std::abort(); }
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

static std::vector<uint8_t> fastq_like(std::mt19937& rng, int records) {
    std::vector<uint8_t> text;
    for(int r = 0; r < records; ++r) {
        char name[64];
        const int n = std::snprintf(name, sizeof name, "@M:1:FC:1:1101:%07d:%05d 1:N:0:1\n", r, r % 77777);
        text.insert(text.end(), name, name + n);
        const int length = 30 + static_cast<int>(rng() % 80);
        for(int j = 0; j < length; ++j) { text.push_back("ACGTN"[rng() % 100 < 1 ? 4 : rng() & 3]); }
        text.push_back('\n'); text.push_back('+'); text.push_back('\n');
        for(int j = 0; j < length; ++j) { const unsigned v = rng() % 100; text.push_back(v < 80 ? 'F' : v < 90 ? ':' : v < 97 ? '-' : '#'); }
        text.push_back('\n');
    }
    return text;
}

int main(int argc, char** argv) {
    std::mt19937 rng(20240613);
    // the CRC by carry-less multiplication against zlib's, at every length around the blocks of 64 and 16
    {
        std::vector<uint8_t> bytes(5000);
        for(auto& b : bytes) { b = static_cast<uint8_t>(rng()); }
        for(size_t n = 0; n <= bytes.size(); n += (n < 300 ? 1 : 97)) {
            if(phq::fast_inflate::crc32_of(bytes.data(), n) != static_cast<uint32_t>(crc32(0L, bytes.data(), static_cast<uInt>(n)))) { std::printf("crc differs at %zu bytes\n", n); return 1; }
        }
    }
    std::vector<std::pair<std::string, std::vector<uint8_t>>> texts;
    texts.push_back({"empty", {}});
    texts.push_back({"one byte", {'x'}});
    texts.push_back({"fastq", fastq_like(rng, 3000)});
    { std::vector<uint8_t> t(70000); for(auto& b : t) { b = static_cast<uint8_t>(rng()); } texts.push_back({"random bytes", t}); }
    { std::vector<uint8_t> t(100000, 'F'); texts.push_back({"one long run", t}); }
    { std::vector<uint8_t> t; for(int k = 0; k < 40000; ++k) { t.push_back("ACGT"[k % 3]); if(k % 1000 == 0) { t.push_back(static_cast<uint8_t>(rng())); } } texts.push_back({"short periods", t}); }
    { std::vector<uint8_t> t; for(int k = 0; k < 300000; ++k) { t.push_back(static_cast<uint8_t>((rng() % 7 == 0) ? rng() : "the quick brown fox "[k % 20])); } texts.push_back({"text with noise", t}); }
    { std::vector<uint8_t> t; for(int k = 0; k < 66000; ++k) { t.push_back(static_cast<uint8_t>(static_cast<uint64_t>(k) * static_cast<uint64_t>(k) >> 3)); } texts.push_back({"far matches", t}); }
    size_t members = 0, refused = 0;
    for(const auto& named : texts) {
        const std::vector<uint8_t>& text = named.second;
        for(int level = 0; level <= 9; ++level) {
            for(int strategy : {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED}) {
                for(int memory : {1, 8}) {
                    const std::vector<uint8_t> packed = member_of(text, level, strategy, memory);
                    std::vector<uint8_t> out(text.size() + 16, 0xAB);            // the guard bytes behind the text must stay
                    ++members;
                    if(!phq::fast_inflate::inflate_member(packed.data(), packed.size(), out.data(), text.size())) {
                        ++refused;
                        std::printf("refused: %s level %d strategy %d memory %d (%zu -> %zu bytes)\n", named.first.c_str(), level, strategy, memory, text.size(), packed.size());
                        continue;
                    }
                    if(!text.empty() && std::memcmp(out.data(), text.data(), text.size()) != 0) { std::printf("text differs: %s level %d strategy %d\n", named.first.c_str(), level, strategy); return 1; }
                    for(size_t k = text.size(); k < out.size(); ++k) { if(out[k] != 0xAB) { std::printf("wrote past the text: %s level %d strategy %d\n", named.first.c_str(), level, strategy); return 1; } }
                }
            }
        }
    }
    // damaged members: never `true` with other text (the CRC stands behind everything), never a read or write out of bounds
    {
        const std::vector<uint8_t> text = fastq_like(rng, 400);
        const std::vector<uint8_t> packed = member_of(text, 6, Z_DEFAULT_STRATEGY, 8);
        size_t accepted = 0;
        for(int trial = 0; trial < 3000; ++trial) {
            std::vector<uint8_t> damaged = packed;
            const size_t at = rng() % damaged.size();
            if(trial % 3 == 0) { damaged[at] ^= static_cast<uint8_t>(1u << (rng() % 8)); }
            else if(trial % 3 == 1) { damaged.resize(at < 18 ? 18 : at); }
            else { damaged[at] = static_cast<uint8_t>(rng()); }
            std::vector<uint8_t> out(text.size() + 16, 0xAB);
            const size_t stated = trial % 7 == 0 ? text.size() - 1 - rng() % 5 : text.size();
            if(phq::fast_inflate::inflate_member(damaged.data(), damaged.size(), out.data(), stated)) {
                if(stated != text.size() || (!text.empty() && std::memcmp(out.data(), text.data(), text.size()) != 0)) { std::printf("a damaged member was accepted with other text\n"); return 1; }
                ++accepted;                                   // the damage was in a place that does not matter (MTIME, XFL, OS)
            }
            for(size_t k = text.size(); k < out.size(); ++k) { if(out[k] != 0xAB) { std::printf("a damaged member wrote past the text\n"); return 1; } }
        }
        std::printf("damaged members accepted unchanged: %zu of 3000\n", accepted);
    }
    std::printf("members %zu refused %zu\n", members, refused);
    if(argc > 1 && std::string(argv[1]) == "speed") {
        const std::vector<uint8_t> text = fastq_like(rng, 400000);
        for(int level : {1, 6}) {
            const std::vector<uint8_t> packed = member_of(text, level, Z_DEFAULT_STRATEGY, 8);
            std::vector<uint8_t> out(text.size());
            auto t0 = std::chrono::steady_clock::now();
            for(int k = 0; k < 5; ++k) { if(!phq::fast_inflate::inflate_member(packed.data(), packed.size(), out.data(), text.size())) { return 1; } }
            const double fast = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            t0 = std::chrono::steady_clock::now();
            for(int k = 0; k < 5; ++k) {
                z_stream z; std::memset(&z, 0, sizeof z); inflateInit2(&z, 31);
                z.next_in = const_cast<Bytef*>(packed.data()); z.avail_in = static_cast<uInt>(packed.size()); z.next_out = out.data(); z.avail_out = static_cast<uInt>(out.size());
                inflate(&z, Z_FINISH); inflateEnd(&z);
            }
            const double slow = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            t0 = std::chrono::steady_clock::now();
            uint32_t c = 0; for(int k = 0; k < 5; ++k) { c ^= phq::fast_inflate::crc32_of(out.data(), out.size()); }
            const double check = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("level %d: this decoder %.3f GB/s of text (its CRC alone %.2f GB/s, %x), zlib %.3f GB/s\n", level, 5.0 * text.size() / fast / 1e9, 5.0 * text.size() / check / 1e9, c, 5.0 * text.size() / slow / 1e9);
        }
    }
    return refused == 0 ? 0 : 2;
}
