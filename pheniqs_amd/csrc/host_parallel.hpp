// Host side data parallelism for the batch steps around the kernels (staging, record formatting): contiguous
// ranges of a batch on a few threads.  The reference runs its whole per read loop on `decoding threads` workers
// (transcode.cpp:1491-1500); here only the host steps of a batch are split, the classification is on the GPU.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>

#include "error.hpp"

namespace phq {

// threads the process may really use: PHQ_HOST_THREADS, else the affinity mask capped by the cgroup CPU quota, at most 32
inline int host_threads() {
    static const int cached = []() {
        if(const char* text = std::getenv("PHQ_HOST_THREADS")) {
            const int asked = std::atoi(text);
            if(asked > 0) { return std::min(asked, 256); }
        }
        int cores = 1;
        cpu_set_t set;
        if(sched_getaffinity(0, sizeof(set), &set) == 0) { cores = std::max(1, CPU_COUNT(&set)); }
        else { cores = std::max(1u, std::thread::hardware_concurrency()); }
        std::ifstream quota_file("/sys/fs/cgroup/cpu.max");
        std::string quota, period;
        if(quota_file >> quota >> period) {
            if(quota != "max") {
                const double q = std::atof(quota.c_str());
                const double p = std::atof(period.c_str());
                if(q > 0 && p > 0) { cores = std::max(1, std::min(cores, static_cast<int>(q / p))); }
            }
        }
        return std::min(cores, 32);
    }();
    return cached;
}

// f(begin, end, part) over `parts` contiguous ranges of [0, n); inline when the batch is small
template <typename F>
void parallel_ranges(int64_t n, int64_t grain, int parts, F&& f) {
    if(n <= 0) { return; }
    parts = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(parts, (n + grain - 1) / grain)));
    if(parts == 1) { f(static_cast<int64_t>(0), n, 0); return; }
    std::vector<std::thread> workers;
    // a worker's failure crosses to the caller as (error class, text): an IO error stays an IO error
    std::vector<std::string> failure(static_cast<size_t>(parts));
    std::vector<int> failure_code(static_cast<size_t>(parts), 0);
    for(int part = 0; part < parts; ++part) {
        const int64_t begin = n * part / parts;
        const int64_t end = n * (part + 1) / parts;
        workers.emplace_back([&, begin, end, part]() {
            try { f(begin, end, part); }
            catch(const Error& e) { failure_code[static_cast<size_t>(part)] = e.code; failure[static_cast<size_t>(part)] = e.message.empty() ? std::string("error") : e.message; }
            catch(const std::exception& e) { failure[static_cast<size_t>(part)] = e.what(); if(failure[static_cast<size_t>(part)].empty()) { failure[static_cast<size_t>(part)] = "error"; } }
        });
    }
    for(auto& worker : workers) { worker.join(); }
    for(size_t part = 0; part < failure.size(); ++part) {
        if(failure[part].empty()) { continue; }
        if(failure_code[part] != 0) { throw_error(failure_code[part], failure[part]); }
        throw std::runtime_error(failure[part]);
    }
}

}  // namespace phq
