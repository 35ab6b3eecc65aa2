// Rate of LDS atomic adds with scattered addresses (the access pattern of count_results_kernel): ns per lane-atomic per CU.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int loops, int bins) {
    extern __shared__ unsigned long long lds[];
    for(int i = threadIdx.x; i < bins * 2; i += 1024) lds[i] = 0;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
    for(int l = 0; l < loops; ++l) {
        x = x * 1664525u + 1013904223u;
        const unsigned bin = (x >> 8) % bins;
        if(KIND == 0) { atomicAdd(&lds[2 * bin], 1ull); }
        else if(KIND == 1) { atomicAdd(reinterpret_cast<unsigned*>(&lds[2 * bin]), 1u); }
        else if(KIND == 2) { atomicAdd(&lds[2 * bin], 1ull); atomicAdd(&lds[2 * bin + 1], (unsigned long long)x); }
        else if(KIND == 3) { atomicAdd(reinterpret_cast<unsigned*>(&lds[2 * bin]), 1u); atomicAdd(reinterpret_cast<unsigned*>(&lds[2 * bin]) + 1, x & 15u); }
        else if(KIND == 4) { lds[2 * bin] += 1; }                       // plain read modify write (wrong under conflicts): the cost of the traffic alone
    }
    __syncthreads();
    if(threadIdx.x == 0) out[blockIdx.x] = lds[0] + lds[3];
}

template <int KIND> void run(const char* name, int bins) {
    unsigned long long* out; (void)hipMalloc(&out, 8 * 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int loops = 2000;
    k<KIND><<<256, 1024, bins * 16>>>(out, 10, bins);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    k<KIND><<<256, 1024, bins * 16>>>(out, loops, bins);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-34s bins %5d: %.3f ms, %.3f ns per lane iteration per CU\n", name, bins, ms, ms * 1e6 / (double(loops) * 1024));
    (void)hipFree(out);
}

int main() {
    for(int bins : {97, 1024, 9217}) {
        run<0>("ds_add_u64", bins);
        run<1>("ds_add_u32", bins);
        run<2>("2 x ds_add_u64", bins);
        run<3>("2 x ds_add_u32", bins);
        run<4>("plain read + write u64", bins);
    }
    return 0;
}
