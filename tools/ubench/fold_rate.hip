// Where do the cycles of the MandelBox fold iteration go?  Variants of the fast-path loop body on escaping orbits
// (the sphere-fold block is never entered).  Prints VALU "cycles" per iteration per wave assuming 2.4 GHz x 1024 SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V>
__global__ void __launch_bounds__(256) k_fold(float* out, int reps, float l, float s, float frs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t hsh = i * 2654435761u;
    float ox = 4.0f + (hsh & 1023) / 256.0f, oy = -3.0f - ((hsh >> 10) & 1023) / 256.0f, oz = 2.5f + ((hsh >> 20) & 1023) / 512.0f;
    float acc = 0.0f;
    for (int r = 0; r < reps; r++) {
        float px = ox, py = oy, pz = oz, dr = 1.0f;
        for (int it = 0; it < 12; it++) {
            if (V != 3) {
                px = __builtin_fmaf(__builtin_amdgcn_fmed3f(px, -l, l), 2.0f, -px);
                py = __builtin_fmaf(__builtin_amdgcn_fmed3f(py, -l, l), 2.0f, -py);
                pz = __builtin_fmaf(__builtin_amdgcn_fmed3f(pz, -l, l), 2.0f, -pz);
            } else {
                px = __builtin_fmaf(px, 2.0f, -l); py = __builtin_fmaf(py, 2.0f, -l); pz = __builtin_fmaf(pz, 2.0f, -l);
            }
            if (V == 0 || V == 1 || V == 3) {
                const float r2 = px * px + (py * py + pz * pz);
                if (V == 0 || V == 3) {
                    if (r2 < frs) { const float m = frs / r2; px *= m; py *= m; pz *= m; dr *= m; }
                } else acc += r2;
            }
            if (V == 4) { // unlikely hint
                const float r2 = px * px + (py * py + pz * pz);
                if (__builtin_expect(r2 < frs, 0)) { const float m = frs / r2; px *= m; py *= m; pz *= m; dr *= m; }
            }
            if (V == 5) { // branch-free: Newton-Raphson division always, select 1.0 when not folding
                const float r2 = px * px + (py * py + pz * pz);
                const float d = __builtin_amdgcn_fmed3f(r2, 1e-4f, __builtin_inff());
                float rc = __builtin_amdgcn_rcpf(d);
                rc = __builtin_fmaf(__builtin_fmaf(-d, rc, 1.0f), rc, rc);
                float q = frs * rc;
                q = __builtin_fmaf(__builtin_fmaf(-d, q, frs), rc, q);
                q = __builtin_fmaf(__builtin_fmaf(-d, q, frs), rc, q);
                const float m = r2 < frs ? q : 1.0f;
                px *= m; py *= m; pz *= m; dr *= m;
            }
            if (V == 6) { // wave-uniform scalar branch on the ballot, lane select inside
                const float r2 = px * px + (py * py + pz * pz);
                const bool need = r2 < frs;
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(need) != 0, 0)) {
                    const float m = need ? frs / r2 : 1.0f; px *= m; py *= m; pz *= m; dr *= m;
                }
            }
            px = px * s + ox; py = py * s + oy; pz = pz * s + oz;
            dr = -dr * s + 1.0f;
            if (V == 3) { px *= 0.25f; py *= 0.25f; pz *= 0.25f; }
        }
        acc += px + py + pz + dr;
        ox += 1e-6f; oy -= 1e-6f; oz += 2e-6f;
    }
    out[i] = acc;
}
template <int V>
void run(const char* name, float* out) {
    const int blocks = 2048, reps = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_fold<V><<<blocks, 256>>>(out, 10, 1.0f, -2.1f, 3.61f); hipDeviceSynchronize();
    hipEventRecord(a); k_fold<V><<<blocks, 256>>>(out, reps, 1.0f, -2.1f, 3.61f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_iters = (double)blocks * 4 * reps * 12;
    printf("%-44s %7.2f ms  %6.1f cycles/iteration/wave (2.4 GHz x 1024 SIMDs)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / wave_iters);
}

// ILP experiment: NP independent orbits per thread in lock step, realistic sphere fold (exec-masked block with the
// 4-instruction division) on points near the fractal so that the block is entered in most iterations.
template <int NP>
__global__ void __launch_bounds__(256) k_fold_ilp(float* out, int reps, float l, float s, float frs, float mrs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float ox[NP], oy[NP], oz[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) {
        uint32_t hsh = (i * NP + k) * 2654435761u;
        ox[k] = -2.0f + 4.0f * ((hsh & 1023) / 1023.0f); oy[k] = -2.0f + 4.0f * (((hsh >> 10) & 1023) / 1023.0f); oz[k] = -2.0f + 4.0f * (((hsh >> 20) & 1023) / 1023.0f);
    }
    float acc = 0.0f;
    for (int r = 0; r < reps; r++) {
        float px[NP], py[NP], pz[NP], dr[NP];
#pragma unroll
        for (int k = 0; k < NP; k++) { px[k] = ox[k]; py[k] = oy[k]; pz[k] = oz[k]; dr[k] = 1.0f; }
#pragma unroll
        for (int it = 0; it < 12; it++) {
#pragma unroll
            for (int k = 0; k < NP; k++) {
                px[k] = __builtin_fmaf(__builtin_amdgcn_fmed3f(px[k], -l, l), 2.0f, -px[k]);
                py[k] = __builtin_fmaf(__builtin_amdgcn_fmed3f(py[k], -l, l), 2.0f, -py[k]);
                pz[k] = __builtin_fmaf(__builtin_amdgcn_fmed3f(pz[k], -l, l), 2.0f, -pz[k]);
            }
            float r2[NP];
            bool fold[NP];
            bool any = false;
#pragma unroll
            for (int k = 0; k < NP; k++) { r2[k] = px[k] * px[k] + (py[k] * py[k] + pz[k] * pz[k]); fold[k] = r2[k] < frs; any = any || fold[k]; }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(any) != 0, 0)) {
#pragma unroll
                for (int k = 0; k < NP; k++)
                    if (fold[k]) {
                        const float d = r2[k] < mrs ? mrs : r2[k];
                        const float rc = __builtin_amdgcn_rcpf(d);
                        float q = frs * rc;
                        q = __builtin_fmaf(__builtin_fmaf(-d, q, frs), rc, q);
                        px[k] *= q; py[k] *= q; pz[k] *= q; dr[k] *= q;
                    }
            }
#pragma unroll
            for (int k = 0; k < NP; k++) {
                px[k] = px[k] * s + ox[k]; py[k] = py[k] * s + oy[k]; pz[k] = pz[k] * s + oz[k];
                dr[k] = -dr[k] * s + 1.0f;
            }
        }
#pragma unroll
        for (int k = 0; k < NP; k++) { acc += px[k] + py[k] + pz[k] + dr[k]; ox[k] += 1e-6f; oy[k] -= 1e-6f; oz[k] += 2e-6f; }
    }
    out[i] = acc;
}
template <int NP>
void run_ilp(const char* name, float* out) {
    const int blocks = 2048, reps = 1000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_fold_ilp<NP><<<blocks, 256>>>(out, 10, 1.0f, -2.1f, 3.61f, 1e-4f); hipDeviceSynchronize();
    hipEventRecord(a); k_fold_ilp<NP><<<blocks, 256>>>(out, reps, 1.0f, -2.1f, 3.61f, 1e-4f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_iters = (double)blocks * 4 * reps * 12 * NP;
    printf("%-44s %7.2f ms  %6.1f cycles/iteration/orbit-wave (2.4 GHz x 1024 SIMDs)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / wave_iters);
}
int main() {
    float* out; hipMalloc(&out, 2048 * 256 * 4);
    run<0>("V0 full body (branch never taken)", out);
    run<1>("V1 r2 computed, no compare/branch", out);
    run<2>("V2 no r2, no branch", out);
    run<3>("V3 no med3 (fma instead), with r2+branch", out);
    run<4>("V4 branch with unlikely hint", out);
    run<5>("V5 branch-free (always divide, select)", out);
    run<6>("V6 scalar branch on ballot (unlikely)", out);
    run_ilp<1>("ILP1 near-fractal points, masked fold block", out);
    run_ilp<2>("ILP2 two orbits per thread", out);
    run_ilp<3>("ILP3 three orbits per thread", out);
    return 0;
}
