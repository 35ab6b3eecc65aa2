// Issue rate of a few vector instructions on gfx950, N waves per SIMD: cycles per wave-instruction per SIMD.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void rate_kernel(unsigned* out, int loops, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
    float f0 = a0 * 1e-9f, f1 = a1 * 1e-9f, f2 = a2 * 1e-9f, f3 = a3 * 1e-9f, f4 = a4 * 1e-9f, f5 = a5 * 1e-9f, f6 = a6 * 1e-9f, f7 = a7 * 1e-9f;
    double d0 = f0, d1 = f1, d2 = f2, d3 = f3, d4 = f4, d5 = f5, d6 = f6, d7 = f7;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7}, p4 = {f1, f0}, p5 = {f3, f2}, p6 = {f5, f4}, p7 = {f7, f6};
    __shared__ float lds[4096];
    for(int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    for(int l = 0; l < loops; ++l) {
        if(KIND == 0) {          // v_xor_b32
            REP16(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %3\n v_xor_b32 %4, %4, %5\n v_xor_b32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if(KIND == 1) {   // v_fma_f32
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %2, %2, %3, %3\n v_fma_f32 %4, %4, %5, %5\n v_fma_f32 %6, %6, %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));)
        } else if(KIND == 2) {   // v_pk_mul_f32
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %6, %6, %7" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
        } else if(KIND == 3) {   // v_mul_f64
            REP16(asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %2, %2, %3\n v_mul_f64 %4, %4, %5\n v_mul_f64 %6, %6, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
        } else if(KIND == 4) {   // v_fma_f64
            REP16(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %2, %2, %3, %3\n v_fma_f64 %4, %4, %5, %5\n v_fma_f64 %6, %6, %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
        } else if(KIND == 5) {   // v_perm_b32
            REP16(asm volatile("v_perm_b32 %0, %0, %1, %1\n v_perm_b32 %2, %2, %3, %3\n v_perm_b32 %4, %4, %5, %5\n v_perm_b32 %6, %6, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if(KIND == 6) {   // v_cndmask_b32 (vcc)
            REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");)
        } else if(KIND == 7) {   // v_mul_f32
            REP16(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %2, %2, %3\n v_mul_f32 %4, %4, %5\n v_mul_f32 %6, %6, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));)
        } else if(KIND == 8) {   // v_add_f64
            REP16(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %2, %2, %3\n v_add_f64 %4, %4, %5\n v_add_f64 %6, %6, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
        } else if(KIND == 9) {   // v_and_or_b32 (VOP3)
            REP16(asm volatile("v_and_or_b32 %0, %0, %1, %1\n v_and_or_b32 %2, %2, %3, %3\n v_and_or_b32 %4, %4, %5, %5\n v_and_or_b32 %6, %6, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if(KIND == 10) {  // v_rcp_f64
            REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
        } else if(KIND == 11) {  // ds_read_b32, conflict free
            unsigned addr = (threadIdx.x & 1023u) * 4u;
            REP16(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4096\n ds_read_b32 %2, %4 offset:8192\n ds_read_b32 %3, %4 offset:12288\n s_waitcnt lgkmcnt(0)" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(addr));)
        } else if(KIND == 12) {  // v_sad_u8
            REP16(asm volatile("v_sad_u8 %0, %0, %1, %1\n v_sad_u8 %2, %2, %3, %3\n v_sad_u8 %4, %4, %5, %5\n v_sad_u8 %6, %6, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if(KIND == 13) {  // v_bfe_u32
            REP16(asm volatile("v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if(KIND == 14) {  // v_exp_f32
            REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
        } else if(KIND == 15) {  // v_dot8 u4? use v_dot4_u32_u8
            REP16(asm volatile("v_dot4_u32_u8 %0, %0, %1, %1\n v_dot4_u32_u8 %2, %2, %3, %3\n v_dot4_u32_u8 %4, %4, %5, %5\n v_dot4_u32_u8 %6, %6, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
    }
    unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    float f = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x;
    double d = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    if(r == 0x12345u && f == 1.5f && d == 2.5) out[0] = 1;
}

template <int KIND> double run(const char* name, int per_loop, int waves_per_simd) {
    unsigned* out; hipMalloc(&out, 4);
    const int loops = 2000;
    const int threads = 256;                                   // one wave per SIMD per workgroup
    const int blocks = 256 * waves_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    rate_kernel<KIND><<<blocks, threads>>>(out, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    rate_kernel<KIND><<<blocks, threads>>>(out, loops, 1);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = double(loops) * per_loop * waves_per_simd;      // wave-instructions per SIMD
    const double ns_per = ms * 1e6 / instr_per_simd;
    printf("%-16s waves/SIMD %d: %.3f ms, %.3f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz\n", name, waves_per_simd, ms, ns_per, ns_per * 2.4);
    hipFree(out);
    return ns_per;
}

int main() {
    for(int w : {1, 2, 4, 8}) {
        run<0>("v_xor_b32", 64, w);
        run<1>("v_fma_f32", 64, w);
        run<7>("v_mul_f32", 64, w);
        run<2>("v_pk_mul_f32", 64, w);
        run<3>("v_mul_f64", 64, w);
        run<4>("v_fma_f64", 64, w);
        run<8>("v_add_f64", 64, w);
        run<5>("v_perm_b32", 64, w);
        run<6>("v_cndmask_b32", 64, w);
        run<9>("v_and_or_b32", 64, w);
        run<10>("v_rcp_f64", 64, w);
        run<11>("ds_read_b32", 64, w);
        run<12>("v_sad_u8", 64, w);
        run<13>("v_bfe_u32", 64, w);
        run<14>("v_exp_f32", 64, w);
        run<15>("v_dot4_u32_u8", 64, w);
    }
    return 0;
}
