// Micro-benchmarks behind two design questions of the march kernels (gfx950):
//  1. Does a VALU instruction cost less when most lanes are exec-masked?  (If all-inactive 16/32-lane groups were skipped, lanes
//     that need the sphere-fold block could be packed into one part of the wave.)  A wave runs ITERS x 16 v_fma / v_rcp under
//     different exec masks; 8 waves per SIMD, all CUs.
//  2. What does a workgroup that exits at once cost?  (Sizing a grid for the host's UPPER BOUND of a device-resident count.)
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define REP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define KERNEL(name, OP)                                                                              \
    __global__ void __launch_bounds__(256) name(float* out, float seed, unsigned long long mask) {     \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        float b = seed * 0.5f, c = 0.25f;                                                              \
        if ((mask >> (threadIdx.x & 63)) & 1ull)                                                       \
            for (int i = 0; i < ITERS; i++) { REP8(OP) REP8(OP) }                                      \
        if (seed == 12345.678f) out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;             \
    }
KERNEL(k_fma, FMA)
KERNEL(k_rcp, RCP)
KERNEL(k_mul, MUL)
__global__ void __launch_bounds__(256) k_empty(const unsigned* n, float* out) {
    if (blockIdx.x * 256u + threadIdx.x >= *n) return;
    out[threadIdx.x] = 1.0f;
}
template <typename K> double run(K k, unsigned long long mask) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<256 * 8, 256>>>(out, 1.0f, mask); hipDeviceSynchronize();
    hipEventRecord(a);
    k<<<256 * 8, 256>>>(out, 1.0f, mask);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipFree(out);
    return ms * 1e6 / ((double)ITERS * 16 * 8.0); // ns per wave-instruction per SIMD (8 waves per SIMD)
}
int main() {
    struct { const char* name; unsigned long long m; } masks[] = {
        {"all 64", ~0ull}, {"lanes 0-31", 0xFFFFFFFFull}, {"lanes 0-15", 0xFFFFull}, {"lanes 0-7", 0xFFull}, {"lane 0", 1ull},
        {"1 per 16 (0,16,32,48)", 0x0001000100010001ull}, {"every other lane", 0x5555555555555555ull}, {"lanes 32-63", 0xFFFFFFFF00000000ull}};
    for (auto& m : masks)
        printf("%-24s v_fma %.3f ns   v_mul %.3f ns   v_rcp %.3f ns   (per wave-instruction per SIMD)\n", m.name, run(k_fma, m.m), run(k_mul, m.m), run(k_rcp, m.m));
    unsigned* dn; float* out; hipMalloc(&dn, 4); hipMalloc(&out, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (unsigned blocks : {2048u, 65536u, 524288u, 2097152u}) {
        unsigned zero = 0; hipMemcpy(dn, &zero, 4, hipMemcpyHostToDevice);
        k_empty<<<blocks, 256>>>(dn, out); hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 10; r++) k_empty<<<blocks, 256>>>(dn, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("empty grid of %8u blocks x 256 threads: %.1f us per launch  (%.2f ns per block)\n", blocks, ms * 100.0, ms * 1e5 / blocks);
    }
    return 0;
}
