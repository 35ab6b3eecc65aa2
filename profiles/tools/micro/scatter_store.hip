// How fast do scattered partial line stores complete?  n threads each store one 16 byte record and one 4 byte word at
// every k-th place of two arrays (the tail kernels' pattern: a deferred read's result record and status word among the
// records of the reads around it), against the same number of stores to consecutive places.
//   hipcc --offload-arch=gfx950 -O3 scatter_store.hip -o scatter_store && ./scatter_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t vector4 __attribute__((ext_vector_type(4)));

template <bool STREAMING>
__global__ void scatter(vector4* records, uint32_t* words, int64_t n, int64_t every) {
    for(int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        vector4 value;
        value.x = static_cast<uint32_t>(i); value.y = 1; value.z = 2; value.w = 3;
        if(STREAMING) {
            __builtin_nontemporal_store(value, &records[i * every]);
            __builtin_nontemporal_store(static_cast<uint32_t>(i), &words[i * every]);
        } else {
            records[i * every] = value;
            words[i * every] = static_cast<uint32_t>(i);
        }
    }
}

int main() {
    const int64_t total = 100000000;             // places
    vector4* records; uint32_t* words;
    hipMalloc(&records, total * sizeof(vector4));
    hipMalloc(&words, total * sizeof(uint32_t));
    hipMemset(records, 0, total * sizeof(vector4));
    hipMemset(words, 0, total * sizeof(uint32_t));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for(int64_t every : {1, 2, 4, 10, 16, 64}) {
        const int64_t n = total / every > 9600000 ? 9600000 : total / every;
        for(int streaming = 0; streaming < 2; ++streaming) {
            float best = 1e9f;
            for(int rep = 0; rep < 5; ++rep) {
                hipEventRecord(a);
                if(streaming) { hipLaunchKernelGGL(scatter<true>, dim3(4096), dim3(256), 0, 0, records, words, n, every); }
                else { hipLaunchKernelGGL(scatter<false>, dim3(4096), dim3(256), 0, 0, records, words, n, every); }
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if(ms < best) { best = ms; }
            }
            std::printf("every %3lld-th place, %s stores: %lld records + words in %.3f ms (%.1f G stores/s)\n", static_cast<long long>(every),
                        streaming ? "streaming" : "cached   ", static_cast<long long>(n), best, 2.0 * n / best / 1e6);
        }
    }
    return 0;
}
