// alloc_rate.hip - what does device memory cost at first use on MI355X, and can it be had incrementally while kernels run?
//   1. hipMalloc of 1 / 8 / 32 / 96 GB: wall time, time of the first kernel touching it, hipFree.
//   2. virtual-memory API: reserve a 128 GB range once, then create + map + set-access physical chunks of 1 / 2 / 4 GB: time per
//      chunk (the arena of a worker could grow IN PLACE on a side thread while the first batches render).
//   3. both of the above on a side thread WHILE a long compute kernel runs on another stream: does the kernel slow down, does the
//      allocation slow down?
// build: hipcc --offload-arch=gfx950 -O2 -o alloc_rate alloc_rate.hip -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_touch(uint4* p, size_t n) { // one 16-byte store per 4 KB page-ish stride: touches every 64 KB
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = i; k < n; k += stride) p[k * 4096] = make_uint4(1, 2, 3, 4);
}
__global__ void k_fill(uint4* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = i; k < n; k += stride) p[k] = make_uint4(1, 2, 3, 4);
}
__global__ void k_spin(float* out, int iters) { // VALU-bound busy kernel
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; i++) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
    if (a == 12345.0f) out[0] = a + b;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const bool do_vmm = argc > 1 && atoi(argv[1]) != 0; // VMM stages only on request
    CK(hipSetDevice(0));
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    float* d_out; CK(hipMalloc(&d_out, 4));
    // warm-up
    hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s1, d_out, 1000); CK(hipStreamSynchronize(s1));
    // ---- 1. plain hipMalloc
    for (size_t gb : {1, 8, 32, 96}) {
        void* p = nullptr;
        double t0 = now(); hipError_t e = hipMalloc(&p, gb << 30); double t1 = now();
        if (e != hipSuccess) { printf("hipMalloc %zu GB failed: %s\n", gb, hipGetErrorString(e)); continue; }
        hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, s1, (uint4*)p, (gb << 30) / 65536); CK(hipStreamSynchronize(s1)); double t2 = now();
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s1, (uint4*)p, (gb << 30) / 16); CK(hipStreamSynchronize(s1)); double t3 = now();
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s1, (uint4*)p, (gb << 30) / 16); CK(hipStreamSynchronize(s1)); double t4 = now();
        CK(hipFree(p)); double t5 = now();
        printf("hipMalloc %3zu GB: malloc %8.2f ms (%.2f ms/GB)  first touch (1 store / 64 KB) %7.2f ms  first full fill %7.2f ms  second fill %7.2f ms  free %7.2f ms\n",
               gb, t1 - t0, (t1 - t0) / gb, t2 - t1, t3 - t2, t4 - t3, t5 - t4);
    }
    // second allocation of the same size right after a free: does the runtime cache it?
    { void* p; double t0 = now(); CK(hipMalloc(&p, (size_t)32 << 30)); double t1 = now(); CK(hipFree(p)); printf("hipMalloc 32 GB again: %.2f ms\n", t1 - t0); }
    // ---- 2. VMM
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("VMM granularity (recommended) %zu bytes\n", gran);
    const size_t RANGE = (size_t)128 << 30;
    void* va = nullptr;
    if (do_vmm) { double t0 = now(); hipError_t e = hipMemAddressReserve(&va, RANGE, 0, nullptr, 0); printf("hipMemAddressReserve 128 GB: %.3f ms (%s)\n", now() - t0, hipGetErrorString(e)); if (e != hipSuccess) va = nullptr; }
    std::vector<hipMemGenericAllocationHandle_t> handles;
    size_t mapped = 0;
    auto map_chunk = [&](size_t bytes, bool verbose) -> bool {
        hipMemGenericAllocationHandle_t h;
        double t0 = now();
        if (hipMemCreate(&h, bytes, &prop, 0) != hipSuccess) { printf("hipMemCreate failed\n"); return false; }
        double t1 = now();
        if (hipMemMap((char*)va + mapped, bytes, 0, h, 0) != hipSuccess) { printf("hipMemMap failed\n"); return false; }
        double t2 = now();
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess((char*)va + mapped, bytes, &acc, 1) != hipSuccess) { printf("hipMemSetAccess failed\n"); return false; }
        double t3 = now();
        if (verbose) printf("  VMM chunk %5.1f GB at +%5.1f GB: create %7.2f map %6.2f access %7.2f ms  (%.2f ms/GB)\n", bytes / 1073741824.0, mapped / 1073741824.0, t1 - t0, t2 - t1, t3 - t2, (t3 - t0) / (bytes / 1073741824.0));
        handles.push_back(h); mapped += bytes;
        return true;
    };
    if (va) {
        for (size_t gb : {1, 1, 2, 2, 4, 4, 8, 8}) if (!map_chunk(gb << 30, true)) break;
        // the mapped range is ONE contiguous buffer: fill across chunk borders
        double t0 = now();
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s1, (uint4*)va, mapped / 16); hipError_t e = hipStreamSynchronize(s1);
        printf("fill of the %zu GB mapped range: %.2f ms (%s)\n", mapped >> 30, now() - t0, hipGetErrorString(e));
    }
    // ---- 3. concurrency with a running kernel
    auto spin_ms = [&](int iters) { double t0 = now(); hipLaunchKernelGGL(k_spin, dim3(256 * 8), dim3(256), 0, s2, d_out, iters); CK(hipStreamSynchronize(s2)); return now() - t0; };
    int iters = 2000000;
    double base = spin_ms(iters);
    printf("busy kernel alone: %.1f ms\n", base);
    {
        double t_alloc = 0; void* p = nullptr;
        std::thread th([&]() { CK(hipSetDevice(0)); double t0 = now(); CK(hipMalloc(&p, (size_t)32 << 30)); t_alloc = now() - t0; });
        double t = spin_ms(iters); th.join();
        printf("busy kernel with a concurrent hipMalloc(32 GB): kernel %.1f ms, malloc %.1f ms\n", t, t_alloc);
        double t0 = now(); CK(hipFree(p)); printf("hipFree 32 GB: %.1f ms\n", now() - t0);
    }
    if (va) {
        double t_map = 0;
        std::thread th([&]() { CK(hipSetDevice(0)); double t0 = now(); for (int i = 0; i < 8; i++) map_chunk((size_t)4 << 30, false); t_map = now() - t0; });
        double t = spin_ms(iters); th.join();
        printf("busy kernel with 8 concurrent 4 GB VMM chunk maps: kernel %.1f ms, maps %.1f ms (%.2f ms/GB)\n", t, t_map, t_map / 32);
        // a kernel that USES the already-mapped part while more is being mapped
        std::thread th2([&]() { CK(hipSetDevice(0)); for (int i = 0; i < 4; i++) map_chunk((size_t)4 << 30, false); });
        double t0 = now();
        for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s1, (uint4*)va, ((size_t)30 << 30) / 16);
        hipError_t e = hipStreamSynchronize(s1); double t1 = now(); th2.join();
        printf("4 fills of the first 30 GB while 16 GB more are mapped behind them: %.2f ms (%s)\n", t1 - t0, hipGetErrorString(e));
        double t2 = now();
        CK(hipMemUnmap(va, mapped)); for (auto h : handles) CK(hipMemRelease(h)); CK(hipMemAddressFree(va, RANGE));
        printf("unmap + release %zu GB: %.2f ms\n", mapped >> 30, now() - t2);
    }
    // pinned vs pageable host transfers (83 MB film)
    {
        const size_t n = 83 << 20; void* d; CK(hipMalloc(&d, n));
        void* hp = malloc(n); void* hpin; CK(hipHostMalloc(&hpin, n));
        memset(hp, 1, n); memset(hpin, 1, n);
        for (int r = 0; r < 2; r++) {
            double t0 = now(); CK(hipMemcpy(d, hp, n, hipMemcpyHostToDevice)); double t1 = now(); CK(hipMemcpy(hp, d, n, hipMemcpyDeviceToHost)); double t2 = now();
            CK(hipMemcpy(d, hpin, n, hipMemcpyHostToDevice)); double t3 = now(); CK(hipMemcpy(hpin, d, n, hipMemcpyDeviceToHost)); double t4 = now();
            printf("83 MB: pageable H2D %.2f ms D2H %.2f ms | pinned H2D %.2f ms D2H %.2f ms\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3);
        }
    }
    return 0;
}
