// Follow-up of exec_mask.hip: at which active-lane count / pattern does a VALU instruction get slower, and for which opcodes?
// 8 waves per SIMD, all CUs; values stay normal numbers (no denormal effects): fma x*1+0, mul x*1, rcp(1) etc.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2048
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define FMAC(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(c), "v"(c));
#define RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
#define MAXF(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(b));
#define REP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define KERNEL(name, OP)                                                                              \
    __global__ void __launch_bounds__(256) name(float* out, float seed, unsigned long long mask) {     \
        float a0 = seed, a1 = seed, a2 = seed, a3 = seed, a4 = seed, a5 = seed, a6 = seed, a7 = seed;  \
        float b = seed, c = seed - 1.0f; /* b = 1, c = 0 */                                            \
        if ((mask >> (threadIdx.x & 63)) & 1ull)                                                       \
            for (int i = 0; i < ITERS; i++) { REP8(OP) REP8(OP) }                                      \
        if (seed == 12345.678f) out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;             \
    }
KERNEL(k_fma, FMA) KERNEL(k_fmac, FMAC) KERNEL(k_rcp, RCP) KERNEL(k_mul, MUL) KERNEL(k_add, ADD) KERNEL(k_max, MAXF) KERNEL(k_med3, MED3)
template <typename K> double run(K k, unsigned long long mask, int blocks_per_cu) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<256 * blocks_per_cu, 256>>>(out, 1.0f, mask); hipDeviceSynchronize();
    hipEventRecord(a);
    k<<<256 * blocks_per_cu, 256>>>(out, 1.0f, mask);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipFree(out);
    return ms * 1e6 / ((double)ITERS * 16 * blocks_per_cu); // ns per wave-instruction per SIMD
}
static unsigned long long contiguous(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1); }
static unsigned long long spread(int per16) { unsigned long long q = (1ull << per16) - 1; return q | (q << 16) | (q << 32) | (q << 48); }
int main() {
    printf("%-22s %8s %8s %8s %8s %8s %8s %8s   [ns per wave-instruction per SIMD, 8 waves/SIMD]\n", "exec mask", "fma", "fmac", "mul", "add", "max", "med3", "rcp");
    auto row = [&](const char* name, unsigned long long m, int bpc) {
        printf("%-22s %8.3f %8.3f %8.3f %8.3f %8.3f %8.3f %8.3f\n", name, run(k_fma, m, bpc), run(k_fmac, m, bpc), run(k_mul, m, bpc), run(k_add, m, bpc), run(k_max, m, bpc), run(k_med3, m, bpc), run(k_rcp, m, bpc));
    };
    char nm[64];
    for (int n : {64, 32, 24, 17, 16, 15, 14, 12, 10, 9, 8, 4, 2, 1}) { snprintf(nm, sizeof nm, "contiguous %d", n); row(nm, contiguous(n), 8); }
    for (int n : {8, 5, 4, 3, 2, 1}) { snprintf(nm, sizeof nm, "%d per 16-lane group", n); row(nm, spread(n), 8); }
    row("lanes 8-15", 0xFF00ull, 8); row("lanes 16-23", 0xFF0000ull, 8); row("lanes 56-63", 0xFF00000000000000ull, 8);
    printf("-- 2 waves per SIMD\n");
    for (int n : {64, 16, 8, 1}) { snprintf(nm, sizeof nm, "contiguous %d", n); row(nm, contiguous(n), 2); }
    return 0;
}
