// The text rate of csrc/gzip_stream.hpp on one gzip member by thread count, beside zlib's gzread.
//   g++ -std=c++17 -O3 -o gzip_stream_rate gzip_stream_rate.cpp -lz -lpthread ; gzip_stream_rate FILE.gz
#include <chrono>
#include <cstdio>
#include <vector>
#include "../../../pheniqs_amd/csrc/gzip_stream.hpp"
int main(int argc, char** argv) {
    if(argc < 2) { return 2; }
    std::vector<uint8_t> buffer(size_t(128) << 20);
    {
        gzFile z = gzopen(argv[1], "rb");
        gzbuffer(z, 1 << 20);
        const auto t0 = std::chrono::steady_clock::now();
        size_t total = 0;
        for(;;) { const int got = gzread(z, buffer.data(), 64 << 20); if(got <= 0) { break; } total += static_cast<size_t>(got); }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("gzread             : %.3f GB/s of text (%zu bytes)\n", total / dt / 1e9, total);
        gzclose(z);
    }
    for(int threads : {1, 2, 4, 8, 16, 32}) {
        double best = 0, first = 0;
        for(int rep = 0; rep < 3; ++rep) {
            phq::GzipStream stream;
            if(!stream.open(argv[1], threads)) { return 3; }
            const auto t0 = std::chrono::steady_clock::now();
            size_t total = 0;
            for(;;) { const size_t got = stream.read(buffer.data(), buffer.size()); if(got == 0) { break; } total += got; }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if(rep == 0) { first = total / dt / 1e9; }
            best = std::max(best, total / dt / 1e9);
        }
        std::printf("GzipStream %2d threads: %.3f GB/s of text (first run %.3f; chunks taken %llu, dropped %llu so far)\n", threads, best, first,
                    static_cast<unsigned long long>(phq::GzipStream::chunks_taken().load()), static_cast<unsigned long long>(phq::GzipStream::chunks_dropped().load()));
    }
    return 0;
}
