// How fast can 604 MB of pinned host PCM reach HBM?  One hipMemcpyAsync, the same bytes split over 2 / 4 streams (several SDMA
// engines), and a kernel that reads the pinned host buffer directly (zero-copy over PCIe).  -> profiles/h2d_probe_r04.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>
__global__ void __launch_bounds__(256) pull(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) dst[i] = src[i];
}
int main()
{
    const size_t bytes = 131072ull * 2304 * 2;
    void *h, *d; hipHostMalloc(&h, bytes, hipHostMallocDefault); hipMalloc(&d, bytes);
    memset(h, 1, bytes);
    hipStream_t st[8]; for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    auto timeit = [&](const char *name, auto fn) {
        fn(); hipDeviceSynchronize();
        double best = 1e9;
        for (int r = 0; r < 5; r++) { auto t0 = std::chrono::steady_clock::now(); fn(); hipDeviceSynchronize(); double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); if (dt < best) best = dt; }
        printf("%-44s %7.2f ms  %6.1f GB/s\n", name, best * 1e3, bytes / best / 1e9);
    };
    for (int k : {1, 2, 4, 8}) {
        char nm[64]; snprintf(nm, sizeof nm, "hipMemcpyAsync H2D, %d stream(s)", k);
        timeit(nm, [&] { for (int i = 0; i < k; i++) hipMemcpyAsync((char *)d + bytes / k * i, (char *)h + bytes / k * i, bytes / k, hipMemcpyHostToDevice, st[i]); });
    }
    for (int blocks : {256, 1024, 4096}) {
        char nm[64]; snprintf(nm, sizeof nm, "kernel reads pinned host memory, %d blocks", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL(pull, dim3(blocks), dim3(256), 0, st[0], (const uint4 *)h, (uint4 *)d, bytes / 16); });
    }
    timeit("D2H hipMemcpyAsync 1 stream (same bytes)", [&] { hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st[0]); });
    timeit("H2D + D2H together (full duplex)", [&] { hipMemcpyAsync(d, h, bytes / 2, hipMemcpyHostToDevice, st[0]); hipMemcpyAsync((char *)h + bytes / 2, (char *)d + bytes / 2, bytes / 2, hipMemcpyDeviceToHost, st[1]); });
    return 0;
}
