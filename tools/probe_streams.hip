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
// the wide fold's pattern: NR rows of `row16` units each at a stride of row16 (the table seen as NR x row16), a wave has U consecutive
// rows of one 1 KiB tile in flight (tile j = the same offset in every row); rot16: row b rotated by (b % U) * rot16 units inside its row
template <int U>
__global__ void __launch_bounds__(256) k_rows(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t row16, int nrows, size_t rot16) {
    u32x4 acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * 4, wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (size_t j = wave * 64; j < row16; j += waves * 64) {
        for (int b0 = 0; b0 < nrows; b0 += U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + (size_t)(b0 + u) * row16 + ((j + lane + (size_t)u * rot16) & (row16 - 1)));
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
    }
    if (acc[0] == 0x12345 && acc[1] == 77) dst[threadIdx.x] = acc;
}
// what an interleaved schedule of the wide fold would have in flight: U rows (same offset: aliased) of each of G tiles 4 KiB apart;
// a wave owns the G tiles j + 4 KiB * k and walks the rows U at a time
template <int U, int G>
__global__ void __launch_bounds__(256) k_rows_tiles(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t row16, int nrows) {
    u32x4 acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * 4, wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    // windows of 2 G KiB-tiles... a wave pair shares a window of 2 G tiles of 2 KiB: wave parity picks the odd or even 2 KiB halves
    for (size_t win = wave >> 1; win * (size_t)(G * 256) < row16; win += waves >> 1) {
        const size_t j = win * (size_t)(G * 256) + (wave & 1) * 128; // units of 16 B: a window is G x 4 KiB, this wave's half of each 4 KiB
        for (int b0 = 0; b0 < nrows; b0 += U) {
            u32x4 v[U][G][2];
#pragma unroll
            for (int k = 0; k < G; ++k)
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int h = 0; h < 2; ++h) v[u][k][h] = __builtin_nontemporal_load(src + (size_t)(b0 + u) * row16 + j + (size_t)k * 256 + h * 64 + lane);
#pragma unroll
            for (int k = 0; k < G; ++k)
#pragma unroll
                for (int u = 0; u < U; ++u) acc ^= v[u][k][0] ^ v[u][k][1];
        }
    }
    if (acc[0] == 0x12345 && acc[1] == 77) dst[threadIdx.x] = acc;
}
// register loads in the operand layout of the matrix-core wide fold: per step a wave reads rows 2 bp, 2 bp + 1 of its group (64
// outputs = 2 KiB per row) as four 16-byte loads per lane -- lane (g, n), tile nt: row 2 bp + (g >> 1), output 16 nt + n, half g & 1,
// i.e. each instruction covers 2 rows x 512 contiguous bytes; S steps in flight
template <int S>
__global__ void __launch_bounds__(256) k_rows_mfma(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t row16, int nrows) {
    u32x4 acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const size_t waves = (size_t)gridDim.x * 4, wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t ngroups = row16 / 128; // 128 units = 2 KiB
    for (size_t grp = wave; grp < ngroups; grp += waves) {
        const u32x4* base = src + (size_t)(g >> 1) * row16 + grp * 128 + (size_t)n * 2 + (g & 1);
        for (int bp0 = 0; bp0 < nrows / 2; bp0 += S) {
            u32x4 v[S][4];
#pragma unroll
            for (int q = 0; q < S; ++q)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) v[q][nt] = __builtin_nontemporal_load(base + (size_t)(2 * (bp0 + q)) * row16 + nt * 32);
#pragma unroll
            for (int q = 0; q < S; ++q)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc ^= v[q][nt];
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
    // the wide fold 2^28 -> 2^20: 256 rows of 32 MiB; the degree-2 folds: 8 rows of 1 GiB
    for (int grid : {256, 512, 1024, 2048}) {
        printf("--- rows of one tile in flight per wave, grid %d x 256 ---\n", grid);
        const size_t r256 = src16 / 256, r8 = src16 / 8;
        timeit("256 rows x 32 MiB, 4 rows in flight", [&] { hipLaunchKernelGGL(k_rows<4>, dim3(grid), dim3(256), 0, 0, src, dst, r256, 256, (size_t)0); });
        timeit("256 rows x 32 MiB, 4 rows in flight, rotated by 4 KiB", [&] { hipLaunchKernelGGL(k_rows<4>, dim3(grid), dim3(256), 0, 0, src, dst, r256, 256, (size_t)256); });
        timeit("256 rows x 32 MiB, 8 rows in flight", [&] { hipLaunchKernelGGL(k_rows<8>, dim3(grid), dim3(256), 0, 0, src, dst, r256, 256, (size_t)0); });
        timeit("256 rows x 32 MiB, 8 rows in flight, rotated by 4 KiB", [&] { hipLaunchKernelGGL(k_rows<8>, dim3(grid), dim3(256), 0, 0, src, dst, r256, 256, (size_t)256); });
        timeit("256 rows x 32 MiB, 16 rows in flight", [&] { hipLaunchKernelGGL(k_rows<16>, dim3(grid), dim3(256), 0, 0, src, dst, r256, 256, (size_t)0); });
        timeit("256 rows x 32 MiB, 16 rows in flight, rotated by 4 KiB", [&] { hipLaunchKernelGGL(k_rows<16>, dim3(grid), dim3(256), 0, 0, src, dst, r256, 256, (size_t)256); });
        timeit("256 rows x 32 MiB, 2 KiB x 2 rows x 1 tile in flight", [&] { hipLaunchKernelGGL((k_rows_tiles<2, 1>), dim3(grid), dim3(256), 0, 0, src, dst, r256, 256); });
        timeit("256 rows x 32 MiB, 2 KiB x 4 rows x 1 tile in flight", [&] { hipLaunchKernelGGL((k_rows_tiles<4, 1>), dim3(grid), dim3(256), 0, 0, src, dst, r256, 256); });
        timeit("256 rows x 32 MiB, 2 KiB x 2 rows x 2 tiles (4 KiB apart) in flight", [&] { hipLaunchKernelGGL((k_rows_tiles<2, 2>), dim3(grid), dim3(256), 0, 0, src, dst, r256, 256); });
        timeit("256 rows x 32 MiB, 2 KiB x 2 rows x 4 tiles (4 KiB apart) in flight", [&] { hipLaunchKernelGGL((k_rows_tiles<2, 4>), dim3(grid), dim3(256), 0, 0, src, dst, r256, 256); });
        timeit("256 rows x 32 MiB, 2 KiB x 1 row x 4 tiles (4 KiB apart) in flight", [&] { hipLaunchKernelGGL((k_rows_tiles<1, 4>), dim3(grid), dim3(256), 0, 0, src, dst, r256, 256); });
        timeit("256 rows x 32 MiB, 2 KiB x 2 rows x 8 tiles (4 KiB apart) in flight", [&] { hipLaunchKernelGGL((k_rows_tiles<2, 8>), dim3(grid), dim3(256), 0, 0, src, dst, r256, 256); });
        timeit("256 rows x 32 MiB, matrix-operand layout, 2 steps (8 KiB) in flight", [&] { hipLaunchKernelGGL(k_rows_mfma<2>, dim3(grid), dim3(256), 0, 0, src, dst, r256, 256); });
        timeit("256 rows x 32 MiB, matrix-operand layout, 4 steps (16 KiB) in flight", [&] { hipLaunchKernelGGL(k_rows_mfma<4>, dim3(grid), dim3(256), 0, 0, src, dst, r256, 256); });
        timeit("256 rows x 32 MiB, matrix-operand layout, 8 steps (32 KiB) in flight", [&] { hipLaunchKernelGGL(k_rows_mfma<8>, dim3(grid), dim3(256), 0, 0, src, dst, r256, 256); });
        timeit("8 rows x 1 GiB, 8 rows in flight", [&] { hipLaunchKernelGGL(k_rows<8>, dim3(grid), dim3(256), 0, 0, src, dst, r8, 8, (size_t)0); });
        timeit("8 rows x 1 GiB, 8 rows in flight, rotated by 4 KiB", [&] { hipLaunchKernelGGL(k_rows<8>, dim3(grid), dim3(256), 0, 0, src, dst, r8, 8, (size_t)256); });
        timeit("8 rows x 1 GiB, 2 rows in flight", [&] { hipLaunchKernelGGL(k_rows<2>, dim3(grid), dim3(256), 0, 0, src, dst, r8, 8, (size_t)0); });
        timeit("8 rows x 1 GiB, 2 rows in flight, rotated by 4 KiB", [&] { hipLaunchKernelGGL(k_rows<2>, dim3(grid), dim3(256), 0, 0, src, dst, r8, 8, (size_t)256); });
    }
    CHK(hipGetLastError());
    CHK(hipDeviceSynchronize());
    return 0;
}
