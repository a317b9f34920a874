// Upper bound for the march kernels: nothing but MandelBox::dist evaluations on all 64 lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../rayn_amd/csrc/device_core.h"
using namespace rayn;
using namespace rayn_p0;
__global__ void __launch_bounds__(256) k_dist(const DScene* scp, float* out, int reps) {
    const DHitable& h = scp->h[0];
    uint32_t ev = 0;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // incoherent points across the lanes of a wave (hash of the thread index), spread over the fractal's bounding region
    uint32_t hsh = i * 2654435761u;
    f3 p = f3{-2.0f + 4.0f * ((hsh & 1023) / 1023.0f), -2.0f + 4.0f * (((hsh >> 10) & 1023) / 1023.0f), -2.0f + 4.0f * (((hsh >> 20) & 1023) / 1023.0f)};
    float acc = 0.0f;
    for (int r = 0; r < reps; r++) {
        float d = sdf_dist<false>(h, p, ev);
        acc += d;
        p.x += d * 0.31f; p.y -= d * 0.17f; p.z += d * 0.05f;
        if (!(mag_sq(p) < 9.0f)) p = f3{p.x * 0.25f, p.y * 0.25f, p.z * 0.25f}; // stay near the fractal
    }
    out[i] = acc;
}
int main() {
    DScene hs; memset(&hs, 0, sizeof hs);
    hs.n_hitables = 1; hs.h[0].kind = 1; hs.h[0].sdf_kind = 1; hs.h[0].iterations = 12; hs.h[0].box_l = 1.0f;
    hs.h[0].min_rad_sq = 1e-4f; hs.h[0].fixed_rad_sq = 3.61f; hs.h[0].scale = -2.1f; hs.h[0].fast_div = 1;
    DScene* d; hipMalloc(&d, sizeof hs); hipMemcpy(d, &hs, sizeof hs, hipMemcpyHostToDevice);
    const int blocks = 2048, reps = 4000;
    float* out; hipMalloc(&out, blocks * 256 * 4);
    for (int fast = 1; fast >= 0; fast--) {
        hs.h[0].fast_div = fast; hipMemcpy(d, &hs, sizeof hs, hipMemcpyHostToDevice);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        k_dist<<<blocks, 256>>>(d, out, 100); hipDeviceSynchronize();
        hipEventRecord(a); k_dist<<<blocks, 256>>>(d, out, reps); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double evals = (double)blocks * 256 * reps;
        printf("fast_div=%d: %.1f G evals/s  (%.1f TFLOP/s at 404 flop/eval), %.2f ms\n", fast, evals / ms / 1e6, evals * 404 / ms / 1e9, ms);
    }
    return 0;
}
