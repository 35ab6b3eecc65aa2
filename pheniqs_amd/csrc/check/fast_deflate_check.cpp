// fast_deflate.hpp on the CPU, under ASan + UBSan: members of text, noise, runs, short periods, every size from nothing to a few MB
// come back through zlib's inflate AND through fast_inflate.hpp byte for byte, with the CRC and size their trailers state;
// "members N ratio R (zlib level 5: Z)" on the FASTQ text; exit 1 on any difference.  fast_deflate_check [FILE]: that file's rates.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <vector>

#include "../fast_deflate.hpp"

static std::vector<uint8_t> fastq_like(std::mt19937& rng, int records) {
    std::vector<uint8_t> text;
    for(int k = 0; k < records; ++k) {
        char name[96];
        const int n = std::snprintf(name, sizeof(name), "@A00123:45:HXXXXXXX:1:1101:%d:%d 1:N:0:ACGTACGT\n", 1000 + k % 30000, 1000 + k / 30);
        text.insert(text.end(), name, name + n);
        const int length = 30 + static_cast<int>(rng() % 60);
        for(int j = 0; j < length; ++j) { text.push_back("ACGTN"[(rng() % 100) < 1 ? 4 : rng() % 4]); }
        text.push_back('\n'); text.push_back('+'); text.push_back('\n');
        for(int j = 0; j < length; ++j) { const unsigned r = rng() % 100; text.push_back(r < 3 ? '#' : r < 10 ? '-' : r < 20 ? ':' : 'F'); }
        text.push_back('\n');
    }
    return text;
}

static bool round_trip(const std::vector<uint8_t>& text, const char* name, size_t* packed_size) {
    std::vector<uint8_t> packed(text.size() + text.size() / 8 + 128);
    const size_t size = phq::fast_deflate::compress_member(text.data(), text.size(), packed.data(), packed.size());
    if(size == 0) { std::printf("%s: no room\n", name); return false; }
    *packed_size = size;
    // zlib
    std::vector<uint8_t> back(text.size() + 16, 0xAB);
    z_stream z;
    std::memset(&z, 0, sizeof(z));
    if(inflateInit2(&z, 16 + MAX_WBITS) != Z_OK) { return false; }
    z.next_in = packed.data(); z.avail_in = static_cast<uInt>(size);
    z.next_out = back.data(); z.avail_out = static_cast<uInt>(back.size());
    const int status = inflate(&z, Z_FINISH);
    const size_t made = back.size() - z.avail_out;
    const bool whole = z.avail_in == 0;
    inflateEnd(&z);
    if(status != Z_STREAM_END || !whole || made != text.size() || (made && std::memcmp(back.data(), text.data(), made) != 0)) {
        std::printf("%s: zlib does not give the text back (status %d, %zu of %zu bytes)\n", name, status, made, text.size());
        return false;
    }
    // this library's one shot decoder
    std::vector<uint8_t> again(text.size() + 16, 0xCD);
    if(phq::fast_inflate::inflate_member(packed.data(), size, again.data(), text.size())) {
        if(!text.empty() && std::memcmp(again.data(), text.data(), text.size()) != 0) { std::printf("%s: fast_inflate gives other text\n", name); return false; }
    } else { std::printf("%s: fast_inflate refused the member\n", name); return false; }
    return true;
}

int main(int argc, char** argv) {
    if(argc > 1) {
        std::ifstream file(argv[1], std::ios::binary);
        const std::vector<uint8_t> text((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
        std::vector<uint8_t> packed(text.size() + text.size() / 8 + 128);
        for(int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            const size_t size = phq::fast_deflate::compress_member(text.data(), text.size(), packed.data(), packed.size());
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("fast_deflate: %.1f MB/s, ratio %.3f\n", text.size() / dt / 1e6, static_cast<double>(text.size()) / static_cast<double>(size));
            for(int level : {1, 5}) {
                uLongf bound = compressBound(static_cast<uLong>(text.size()));
                std::vector<uint8_t> other(bound);
                t0 = std::chrono::steady_clock::now();
                compress2(other.data(), &bound, text.data(), static_cast<uLong>(text.size()), level);
                dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                std::printf("zlib level %d: %.1f MB/s, ratio %.3f\n", level, text.size() / dt / 1e6, static_cast<double>(text.size()) / static_cast<double>(bound));
            }
        }
        return 0;
    }
    std::mt19937 rng(11);
    size_t members = 0, packed_size = 0;
    std::vector<std::pair<std::string, std::vector<uint8_t>>> texts;
    texts.push_back({"empty", {}});
    for(int n : {1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 255, 256, 257, 258, 259, 260, 1000}) { std::vector<uint8_t> t(static_cast<size_t>(n)); for(auto& b : t) { b = static_cast<uint8_t>("ACGT"[rng() % 4]); } texts.push_back({"short " + std::to_string(n), t}); }
    const std::vector<uint8_t> fastq = fastq_like(rng, 30000);
    texts.push_back({"fastq", fastq});
    { std::vector<uint8_t> t(300000); for(auto& b : t) { b = static_cast<uint8_t>(rng()); } texts.push_back({"noise", t}); }
    { std::vector<uint8_t> t(700000, 'F'); texts.push_back({"one long run", t}); }
    { std::vector<uint8_t> t; for(int k = 0; k < 200000; ++k) { t.push_back("ACGT"[k % 3]); if(k % 1000 == 0) { t.push_back(static_cast<uint8_t>(rng())); } } texts.push_back({"short periods", t}); }
    { std::vector<uint8_t> t; for(int k = 0; k < 400000; ++k) { t.push_back(static_cast<uint8_t>((rng() % 7 == 0) ? rng() : "the quick brown fox "[k % 20])); } texts.push_back({"text with noise", t}); }
    { std::vector<uint8_t> t; for(int k = 0; k < 70000; ++k) { t.push_back(static_cast<uint8_t>(static_cast<uint64_t>(k) * static_cast<uint64_t>(k) >> 3)); } texts.push_back({"far matches", t}); }
    { std::vector<uint8_t> t(100000, 0); for(size_t k = 0; k < t.size(); k += 33000) { t[k] = 1; } texts.push_back({"zeros", t}); }
    { std::vector<uint8_t> t; for(int k = 0; k < 256 * 300; ++k) { t.push_back(static_cast<uint8_t>(k)); } texts.push_back({"every byte", t}); }
    for(const auto& named : texts) {
        if(!round_trip(named.second, named.first.c_str(), &packed_size)) { return 1; }
        ++members;
    }
    // prefixes around window and block boundaries
    for(size_t cut : {size_t(32767), size_t(32768), size_t(32769), size_t(65535), size_t(65536), size_t(65537), size_t(200001)}) {
        const std::vector<uint8_t> prefix(fastq.begin(), fastq.begin() + static_cast<std::ptrdiff_t>(std::min(cut, fastq.size())));
        if(!round_trip(prefix, "fastq prefix", &packed_size)) { return 1; }
        ++members;
    }
    // BGZF (what a BAM file is made of): every block a gzip member that states its own size, at most 64 KiB either way
    for(const std::vector<uint8_t>* data : std::vector<const std::vector<uint8_t>*>{&fastq, &texts[19].second /* noise */, &texts[0].second /* nothing */}) {
        std::vector<uint8_t> blocks(phq::fast_deflate::bgzf_bound(data->size()) + 64);
        const size_t size = phq::fast_deflate::bgzf_blocks(data->data(), data->size(), blocks.data(), blocks.size());
        if(size == 0 && !data->empty()) { std::printf("bgzf: no room\n"); return 1; }
        std::vector<uint8_t> back;
        size_t at = 0;
        while(at < size) {
            const uint8_t* const h = blocks.data() + at;
            if(size - at < 28 || h[0] != 0x1f || h[1] != 0x8b || h[3] != 4 || h[10] != 6 || h[12] != 'B' || h[13] != 'C') { std::printf("bgzf: no block header at %zu\n", at); return 1; }
            const size_t block = (static_cast<size_t>(h[16]) | (static_cast<size_t>(h[17]) << 8)) + 1;
            if(block > 65536 || at + block > size) { std::printf("bgzf: block size %zu\n", block); return 1; }
            std::vector<uint8_t> piece(65536);
            z_stream z;
            std::memset(&z, 0, sizeof(z));
            if(inflateInit2(&z, 16 + MAX_WBITS) != Z_OK) { return 1; }
            z.next_in = const_cast<Bytef*>(h); z.avail_in = static_cast<uInt>(block);
            z.next_out = piece.data(); z.avail_out = static_cast<uInt>(piece.size());
            const int status = inflate(&z, Z_FINISH);
            const size_t made = piece.size() - z.avail_out;
            const bool whole = z.avail_in == 0;
            inflateEnd(&z);
            if(status != Z_STREAM_END || !whole || made > phq::fast_deflate::BGZF_DATA) { std::printf("bgzf: block at %zu does not inflate (status %d)\n", at, status); return 1; }
            back.insert(back.end(), piece.begin(), piece.begin() + static_cast<std::ptrdiff_t>(made));
            at += block;
            ++members;
        }
        if(back != *data) { std::printf("bgzf: other data back\n"); return 1; }
    }
    if(!round_trip(fastq, "fastq", &packed_size)) { return 1; }
    uLongf bound = compressBound(static_cast<uLong>(fastq.size()));
    std::vector<uint8_t> other(bound);
    compress2(other.data(), &bound, fastq.data(), static_cast<uLong>(fastq.size()), 5);
    std::printf("members %zu ratio %.3f (zlib level 5: %.3f)\n", members, static_cast<double>(fastq.size()) / static_cast<double>(packed_size), static_cast<double>(fastq.size()) / static_cast<double>(bound));
    return 0;
}
