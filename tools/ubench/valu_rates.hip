// Micro-benchmark: issue cost of the VALU instructions the MandelBox fold uses, on gfx950.
// Each kernel runs ITERS x 16 instructions of one kind per wave, 8 waves per SIMD, all CUs.
// Prints ns per wave-instruction per SIMD and the ratio to v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITERS 4096

#define KERNEL(name, DECL, BODY)                                                         \
    __global__ void __launch_bounds__(256) name(float* out, float seed) {                 \
        DECL;                                                                             \
        for (int i = 0; i < ITERS; i++) { BODY }                                          \
        if (seed == 12345.678f) out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
    }
#define DECL8 float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; float b = seed * 0.5f, c = 0.25f
#define REP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define REP16(OP) REP8(OP) REP8(OP)

#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define MAXF(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define SQRT(x) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
#define DIVFIX(x) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define DIVSCALE(x) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x) : "v"(b) : "vcc");
#define DIVFMAS(x) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c) : "vcc");
#define CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define CMP(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");

// integer forms that can stand in for a float compare / max on NON-NEGATIVE, non-NaN floats (same bit order)
#define CMPU(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define CMPS(x) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(x), "v"(b) : "s20", "s21");
#define MAXU(x) asm volatile("v_max_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define MINU(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define MINF(x) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define ANDB(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define BFI(x) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(b), "v"(c));
#define ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define SUBF(x) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x) : "v"(b));
#define FMAC(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define FMAK(x) asm volatile("v_fma_f32 %0, %0, 2.0, -%1" : "+v"(x) : "v"(b));
KERNEL(k_cmp_u32, DECL8, REP16(CMPU))
KERNEL(k_cmp_sdst, DECL8, REP16(CMPS))
KERNEL(k_max_u32, DECL8, REP16(MAXU))
KERNEL(k_min_u32, DECL8, REP16(MINU))
KERNEL(k_min_f32, DECL8, REP16(MINF))
KERNEL(k_and_b32, DECL8, REP16(ANDB))
KERNEL(k_bfi_b32, DECL8, REP16(BFI))
KERNEL(k_add_u32, DECL8, REP16(ADDU))
KERNEL(k_sub_f32, DECL8, REP16(SUBF))
KERNEL(k_fmac_f32, DECL8, REP16(FMAC))
KERNEL(k_fma_neg, DECL8, REP16(FMAK))
KERNEL(k_fma, DECL8, REP16(FMA))
KERNEL(k_mul, DECL8, REP16(MUL))
KERNEL(k_add, DECL8, REP16(ADD))
KERNEL(k_med3, DECL8, REP16(MED3))
KERNEL(k_max, DECL8, REP16(MAXF))
KERNEL(k_rcp, DECL8, REP16(RCP))
KERNEL(k_sqrt, DECL8, REP16(SQRT))
KERNEL(k_divfix, DECL8, REP16(DIVFIX))
KERNEL(k_divscale, DECL8, REP16(DIVSCALE))
KERNEL(k_divfmas, DECL8, REP16(DIVFMAS))
KERNEL(k_cndmask, DECL8, REP16(CNDMASK))
KERNEL(k_cmp, DECL8, REP16(CMP))

// packed f32: 8 register pairs
typedef float float2v __attribute__((ext_vector_type(2)));
#define KERNEL2(name, BODY)                                                                \
    __global__ void __launch_bounds__(256) name(float* out, float seed) {                   \
        float2v a0 = {seed, seed + 1}, a1 = {seed + 2, seed + 3}, a2 = {seed + 4, seed}, a3 = {seed, seed}, a4 = {seed, 1}, a5 = {2, seed}, a6 = {seed, 3}, a7 = {4, seed}; \
        float2v b = {seed * 0.5f, seed * 0.25f}, c = {0.25f, 0.5f};                        \
        for (int i = 0; i < ITERS; i++) { BODY }                                            \
        if (seed == 12345.678f) out[threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y; \
    }
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
KERNEL2(k_pkfma, REP16(PKFMA))
KERNEL2(k_pkmul, REP16(PKMUL))
KERNEL2(k_pkadd, REP16(PKADD))

// dependent chain (one accumulator) to see latency with 8 waves/SIMD
#define FMA1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
#define R16(X) X X X X X X X X X X X X X X X X
KERNEL(k_fma_dep, DECL8, R16(FMA1))
// f64
#define KERNEL3(name, BODY)                                                                \
    __global__ void __launch_bounds__(256) name(float* out, float seed) {                   \
        double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5, c = 0.25; \
        for (int i = 0; i < ITERS; i++) { BODY }                                            \
        if (seed == 12345.678f) out[threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
    }
#define DFMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define DMUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
KERNEL3(k_dfma, REP16(DFMA))
KERNEL3(k_dmul, REP16(DMUL))

template <typename K> double run(K k, int blocks_per_cu, int threads) {
    float* out; hipMalloc(&out, 4096);
    int cus = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<cus * blocks_per_cu, threads>>>(out, 1.0f); hipDeviceSynchronize();
    hipEventRecord(a);
    k<<<cus * blocks_per_cu, threads>>>(out, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipFree(out);
    double waves_per_simd = blocks_per_cu * (threads / 64) / 4.0;
    return ms * 1e6 / ((double)ITERS * 16 * waves_per_simd); // ns per wave-instruction per SIMD
}
#define RUN(k) { double v8 = run(k, 8, 256), v2 = run(k, 2, 256), v1 = run(k, 1, 256); printf("%-12s 8w/SIMD %.3f ns (x%.2f)   2w/SIMD %.3f   1w/SIMD %.3f\n", #k, v8, v8 / base, v2, v1); }
int main() {
    double base = run(k_fma, 8, 256);
    RUN(k_fma) RUN(k_mul) RUN(k_add) RUN(k_med3) RUN(k_max) RUN(k_cmp) RUN(k_cndmask) RUN(k_rcp) RUN(k_sqrt) RUN(k_divscale) RUN(k_divfmas) RUN(k_divfix)
    RUN(k_cmp_u32) RUN(k_cmp_sdst) RUN(k_max_u32) RUN(k_min_u32) RUN(k_min_f32) RUN(k_and_b32) RUN(k_bfi_b32) RUN(k_add_u32) RUN(k_sub_f32) RUN(k_fmac_f32) RUN(k_fma_neg)
    RUN(k_pkfma) RUN(k_pkmul) RUN(k_pkadd) RUN(k_fma_dep) RUN(k_dfma) RUN(k_dmul)
    return 0;
}
