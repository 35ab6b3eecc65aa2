// A plain streaming copy, 16 bytes a lane: the practical HBM ceiling of the node next to the 8 TB/s the rooflines are priced
// against (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).  Measurement aid only (phq_copy_bandwidth).
#include <hip/hip_runtime.h>

#include <cstdint>

namespace phq {

namespace {
typedef uint32_t Quad __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_probe_kernel(const Quad* __restrict__ from, Quad* __restrict__ to, size_t quads) {
    const size_t stride = static_cast<size_t>(gridDim.x) * 256u;
    for(size_t i = static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x; i < quads; i += stride) {
        __builtin_nontemporal_store(__builtin_nontemporal_load(&from[i]), &to[i]);
    }
}
}  // namespace

// copies `bytes` (a multiple of 16) `repeats` times on the device's default stream between two buffers it allocates; *gigabytes_per_second counts
// bytes read + bytes written.  Returns a hipError_t value.
int copy_bandwidth(int device, size_t bytes, int repeats, double* gigabytes_per_second) {
    if(const hipError_t status = hipSetDevice(device)) { return static_cast<int>(status); }
    void* const stream = nullptr;
    void* from = nullptr;
    void* to = nullptr;
    hipEvent_t begin = nullptr, end = nullptr;
    hipError_t status = hipMalloc(&from, bytes);
    if(status == hipSuccess) { status = hipMalloc(&to, bytes); }
    if(status == hipSuccess) { status = hipMemsetAsync(from, 1, bytes, static_cast<hipStream_t>(stream)); }
    if(status == hipSuccess) { status = hipEventCreate(&begin); }
    if(status == hipSuccess) { status = hipEventCreate(&end); }
    if(status == hipSuccess) {
        const size_t quads = bytes / 16;
        const unsigned blocks = 256u * 32u;                     // 32 workgroups a CU in flight: the loop strides
        for(int k = 0; k < 2; ++k) { hipLaunchKernelGGL(copy_probe_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const Quad*>(from), static_cast<Quad*>(to), quads); }
        status = hipEventRecord(begin, static_cast<hipStream_t>(stream));
        for(int k = 0; k < repeats; ++k) { hipLaunchKernelGGL(copy_probe_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const Quad*>(from), static_cast<Quad*>(to), quads); }
        if(status == hipSuccess) { status = hipEventRecord(end, static_cast<hipStream_t>(stream)); }
        if(status == hipSuccess) { status = hipEventSynchronize(end); }
        float ms = 0;
        if(status == hipSuccess) { status = hipEventElapsedTime(&ms, begin, end); }
        if(status == hipSuccess && ms > 0) { *gigabytes_per_second = 2.0 * static_cast<double>(bytes) * repeats / (static_cast<double>(ms) * 1e-3) / 1e9; }
    }
    if(begin) { (void)hipEventDestroy(begin); }
    if(end) { (void)hipEventDestroy(end); }
    (void)hipFree(from);
    (void)hipFree(to);
    return static_cast<int>(status);
}

}  // namespace phq
