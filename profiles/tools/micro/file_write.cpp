// How fast can a round's SAM text (~206 MB per Mi reads of the bench's job) go into a file of memory backed storage?
// Strategies for csrc/transcoder.cpp's write_round_text.   g++ -O2 -o file_write file_write.cpp -lpthread
//   usage: file_write PATH THREADS      (8 rounds of 206 MB; best of 3 per strategy)
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename F> void par(int threads, F f) { std::vector<std::thread> t; for(int k = 0; k < threads; ++k) { t.emplace_back(f, k); } for(auto& x : t) { x.join(); } }
static const size_t ROUND = size_t(206) << 20;
static const int ROUNDS = 8;

static double run(const char* path, const std::string& mode, int threads, const char* src) {
    ::unlink(path);
    const int fd = ::open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    std::thread ahead;
    const double t0 = now();
    size_t pos = 0;
    if(mode == "fallocate_all_first") { if(posix_fallocate(fd, 0, ROUND * ROUNDS)) { abort(); } }
    for(int r = 0; r < ROUNDS; ++r) {
        auto slice = [&](int k, size_t& b, size_t& e) { b = (ROUND * k / threads) & ~size_t(4095); e = k + 1 == threads ? ROUND : (ROUND * (k + 1) / threads) & ~size_t(4095); };
        if(mode == "pwrite" || mode == "fallocate_pwrite") {
            if(mode == "fallocate_pwrite") { if(posix_fallocate(fd, pos, ROUND)) { abort(); } }
            par(threads, [&](int k) { size_t b, e; slice(k, b, e); size_t at = b; while(at < e) { const ssize_t d = ::pwrite(fd, src + at, std::min<size_t>(e - at, 1 << 20), pos + at); if(d <= 0) { abort(); } at += d; } });
        } else if(mode == "write1") {
            size_t at = 0; while(at < ROUND) { const ssize_t d = ::write(fd, src + at, ROUND - at); if(d <= 0) { abort(); } at += d; }
        } else {
            if(mode == "mmap" || mode == "populate_write" || mode == "hugepage") { if(ftruncate(fd, pos + ROUND)) { abort(); } }
            if(mode == "fallocate_mmap") { if(posix_fallocate(fd, pos, ROUND)) { abort(); } }
            if(mode == "fallocate_ahead") {          // a helper allocates the NEXT round's pages while this one is copied
                if(r == 0) { if(posix_fallocate(fd, 0, ROUND)) { abort(); } } else { ahead.join(); }
                if(r + 1 < ROUNDS) { ahead = std::thread([fd, pos] { if(posix_fallocate(fd, pos + ROUND, ROUND)) { abort(); } }); }
            }
            char* const w = static_cast<char*>(mmap(nullptr, ROUND, PROT_READ | PROT_WRITE, MAP_SHARED, fd, pos));
            if(w == MAP_FAILED) { abort(); }
            if(mode == "hugepage") { madvise(w, ROUND, MADV_HUGEPAGE); }
            par(threads, [&](int k) { size_t b, e; slice(k, b, e); if(mode == "populate_write") { madvise(w + b, e - b, 23 /* MADV_POPULATE_WRITE */); } memcpy(w + b, src + b, e - b); });
            munmap(w, ROUND);
        }
        pos += ROUND;
    }
    const double dt = now() - t0;
    close(fd);
    ::unlink(path);
    return ROUNDS * ROUND / dt / 1e9;
}

int main(int argc, char** argv) {
    if(argc < 3) { return 2; }
    const int threads = atoi(argv[2]);
    char* const src = static_cast<char*>(aligned_alloc(4096, ROUND));
    memset(src, 'A', ROUND);
    for(const char* mode : {"mmap", "fallocate_mmap", "fallocate_ahead", "fallocate_all_first", "populate_write", "hugepage", "pwrite", "fallocate_pwrite", "write1"}) {
        double best = 0;
        for(int rep = 0; rep < 3; ++rep) { best = std::max(best, run(argv[1], mode, threads, src)); }
        printf("%-20s threads %2d: %6.2f GB/s\n", mode, threads, best);
    }
    return 0;
}
