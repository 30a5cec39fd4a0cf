// tools/probe_streams.hip -- why does a sweep that reads R far-apart streams at once (the m-variable folds read 2^m rows, the degree-2
// folds 8 rows of each of 4 tables) run below the one-stream read rate on this box?  Read-only kernels over the same 8 GiB:
//   R streams at a power-of-two stride, all at the same offset (what a fold does);
//   the same with stream r skewed by r * SKEW bytes (rotating inside its stream): does the aliasing of the streams matter?
//   the same addresses read in longer contiguous runs per stream (RUN KiB per wave before it turns to the next stream).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_streams.hip -o tools/probe_streams
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// n16 = units (16 B) per stream, a power of two; skew16 = units stream r is rotated by, times r
template <int R>
__global__ void __launch_bounds__(256) k_streams(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, size_t skew16) {
    u32x4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        u32x4 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = __builtin_nontemporal_load(src + (size_t)r * n16 + ((i + (size_t)r * skew16) & (n16 - 1)));
#pragma unroll
        for (int r = 0; r < R; ++r) acc ^= v[r];
    }
    if (acc[0] == 0x12345 && acc[1] == 77) dst[threadIdx.x] = acc;
}
// each wave reads RUN KiB of one stream (RUN trips of 1 KiB, all in flight), then the same offsets of the next stream
template <int R, int RUN>
__global__ void __launch_bounds__(256) k_runs(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    u32x4 acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * 4, wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr size_t run16 = (size_t)RUN * 64; // units per run
    for (size_t base = wave * run16; base < n16; base += waves * run16) {
#pragma unroll 1
        for (int r = 0; r < R; ++r) {
            u32x4 v[RUN];
#pragma unroll
            for (int t = 0; t < RUN; ++t) v[t] = __builtin_nontemporal_load(src + (size_t)r * n16 + base + (size_t)t * 64 + lane);
#pragma unroll
            for (int t = 0; t < RUN; ++t) acc ^= v[t];
        }
    }
    if (acc[0] == 0x12345 && acc[1] == 77) dst[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_write(u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = u32x4{1, 2, 3, (unsigned)i};
}

int main() {
    const size_t total = (size_t)8 << 30;
    u32x4 *src, *dst;
    CHK(hipMalloc(&src, total));
    CHK(hipMalloc(&dst, 1 << 20));
    const size_t src16 = total / 16;
    hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, src, src16);
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    auto timeit = [&](const char* name, auto&& launch) {
        std::vector<float> t;
        for (int rep = 0; rep < 6; ++rep) {
            CHK(hipEventRecord(a));
            launch();
            CHK(hipEventRecord(b));
            CHK(hipEventSynchronize(b));
            float ms;
            CHK(hipEventElapsedTime(&ms, a, b));
            if (rep) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("%-72s %8.1f GB/s (%.3f ms)\n", name, (double)total / t[2] / 1e6, t[2]);
        fflush(stdout);
    };
    for (int grid : {1024, 2048, 4096}) {
        printf("--- grid %d x 256 ---\n", grid);
        char nm[128];
        timeit("1 stream", [&] { hipLaunchKernelGGL(k_streams<1>, dim3(grid), dim3(256), 0, 0, src, dst, src16, (size_t)0); });
        timeit("2 streams, same offset", [&] { hipLaunchKernelGGL(k_streams<2>, dim3(grid), dim3(256), 0, 0, src, dst, src16 / 2, (size_t)0); });
        timeit("4 streams, same offset", [&] { hipLaunchKernelGGL(k_streams<4>, dim3(grid), dim3(256), 0, 0, src, dst, src16 / 4, (size_t)0); });
        timeit("8 streams, same offset", [&] { hipLaunchKernelGGL(k_streams<8>, dim3(grid), dim3(256), 0, 0, src, dst, src16 / 8, (size_t)0); });
        for (size_t skew : {(size_t)256, (size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536, (size_t)(1 << 20), (size_t)((1 << 20) + 4096 + 256), (size_t)(33 << 20)}) {
            snprintf(nm, sizeof nm, "8 streams, stream r rotated by r x %zu B", skew);
            timeit(nm, [&] { hipLaunchKernelGGL(k_streams<8>, dim3(grid), dim3(256), 0, 0, src, dst, src16 / 8, skew / 16); });
        }
        timeit("8 streams, runs of 1 KiB per wave and stream", [&] { hipLaunchKernelGGL((k_runs<8, 1>), dim3(grid), dim3(256), 0, 0, src, dst, src16 / 8); });
        timeit("8 streams, runs of 4 KiB per wave and stream", [&] { hipLaunchKernelGGL((k_runs<8, 4>), dim3(grid), dim3(256), 0, 0, src, dst, src16 / 8); });
        timeit("8 streams, runs of 8 KiB per wave and stream", [&] { hipLaunchKernelGGL((k_runs<8, 8>), dim3(grid), dim3(256), 0, 0, src, dst, src16 / 8); });
        timeit("8 streams, runs of 16 KiB per wave and stream", [&] { hipLaunchKernelGGL((k_runs<8, 16>), dim3(grid), dim3(256), 0, 0, src, dst, src16 / 8); });
        timeit("1 stream, runs of 8 KiB per wave", [&] { hipLaunchKernelGGL((k_runs<1, 8>), dim3(grid), dim3(256), 0, 0, src, dst, src16); });
        timeit("32 streams, same offset", [&] { hipLaunchKernelGGL(k_streams<32>, dim3(grid), dim3(256), 0, 0, src, dst, src16 / 32, (size_t)0); });
        timeit("32 streams, stream r rotated by r x 4352 B", [&] { hipLaunchKernelGGL(k_streams<32>, dim3(grid), dim3(256), 0, 0, src, dst, src16 / 32, (size_t)4352 / 16); });
    }
    CHK(hipGetLastError());
    CHK(hipDeviceSynchronize());
    return 0;
}
