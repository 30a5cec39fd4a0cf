// tools/copybench.hip -- what a read stream mixed with a store stream costs on this box: plain copies against R:1 read:write mixes
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/copybench.hip -o tools/copybench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// R read streams (n16 units each, contiguous one after another) xor-ed into one write stream of n16 units; grid-stride, 16 B per lane
template <int R, bool NTS>
__global__ void __launch_bounds__(256) k_mix(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        u32x4 a = __builtin_nontemporal_load(src + i);
#pragma unroll
        for (int r = 1; r < R; ++r) a ^= __builtin_nontemporal_load(src + (size_t)r * n16 + i);
        if (NTS) __builtin_nontemporal_store(a, dst + i); else dst[i] = a;
    }
}
// the same with the workgroup's outputs gathered in LDS and written out in bursts of BURST KiB (reads of the next burst are not
// issued until the burst is stored: reads and writes alternate per workgroup in long runs)
template <int R, int BURST>
__global__ void __launch_bounds__(256) k_mix_burst(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    __shared__ u32x4 buf[BURST * 64];
    constexpr size_t per = (size_t)BURST * 64; // units per burst
    for (size_t base = (size_t)blockIdx.x * per; base < n16; base += (size_t)gridDim.x * per) {
        for (size_t k = threadIdx.x; k < per; k += 256) {
            const size_t i = base + k;
            u32x4 a = __builtin_nontemporal_load(src + i);
#pragma unroll
            for (int r = 1; r < R; ++r) a ^= __builtin_nontemporal_load(src + (size_t)r * n16 + i);
            buf[k] = a;
        }
        __syncthreads();
        for (size_t k = threadIdx.x; k < per; k += 256) dst[base + k] = buf[k];
        __syncthreads();
    }
}
// chunked: each workgroup owns a contiguous chunk of the output (and the matching chunks of the inputs)
template <int R>
__global__ void __launch_bounds__(256) k_mix_chunk(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x, lo = (size_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        u32x4 a = __builtin_nontemporal_load(src + i);
#pragma unroll
        for (int r = 1; r < R; ++r) a ^= __builtin_nontemporal_load(src + (size_t)r * n16 + i);
        dst[i] = a;
    }
}
template <int R>
__global__ void __launch_bounds__(256) k_read(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc ^= __builtin_nontemporal_load(src + (size_t)r * n16 + i);
    }
    if (acc[0] == 0x12345 && acc[1] == 77) dst[threadIdx.x] = acc;
}
// read-only sweeps in the shapes the block-sum kernels use: U independent 16-B loads per lane and trip
template <int U>
__global__ void __launch_bounds__(256) k_read_u(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    u32x4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(src + i);
    if (acc[0] == 0x12345 && acc[1] == 77) dst[threadIdx.x] = acc;
}
// 32 B per lane (an element per lane: two 16-B loads at a 32-B lane stride), as k_block_sums reads
__global__ void __launch_bounds__(256) k_read_32(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256, n32 = n16 / 2;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n32; i += stride)
        acc ^= __builtin_nontemporal_load(src + 2 * i) ^ __builtin_nontemporal_load(src + 2 * i + 1);
    if (acc[0] == 0x12345 && acc[1] == 77) dst[threadIdx.x] = acc;
}
// LDS-DMA ring per wave (D rows of 1 KiB = one DMA instruction each), nt policy, as k_block_sums_dma
template <int D>
__global__ void __launch_bounds__(256) k_read_dma(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    __shared__ u32x4 ring[4][D][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t nrows = n16 / 64, wstride = (size_t)gridDim.x * 4, row0 = (size_t)blockIdx.x * 4 + wave;
    auto issue = [&](size_t r, int slot) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + r * 64 + lane),
                                         (__attribute__((address_space(3))) void*)&ring[wave][slot][0], 16, 0, 2);
    };
    size_t ri = row0;
    int inflight = 0;
#pragma unroll
    for (int sl = 0; sl < D; ++sl)
        if (ri < nrows) { issue(ri, sl); ri += wstride; ++inflight; }
    int slot = 0;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t r = row0; r < nrows; r += wstride) {
        if (inflight == D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u32x4 v = ring[wave][slot][lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        --inflight;
        if (ri < nrows) { issue(ri, slot); ri += wstride; ++inflight; }
        acc ^= v;
        slot = slot + 1 == D ? 0 : slot + 1;
    }
    if (acc[0] == 0x12345 && acc[1] == 77) dst[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_write(u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = u32x4{1, 2, 3, (unsigned)i};
}

int main() {
    const size_t total = (size_t)8 << 30; // source bytes
    u32x4 *src, *dst;
    CHK(hipMalloc(&src, total));
    CHK(hipMalloc(&dst, total));
    CHK(hipMemset(src, 0x11, total));
    CHK(hipMemset(dst, 0x22, total));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    auto timeit = [&](const char* name, double bytes, auto&& launch) {
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
        printf("%-64s %8.1f GB/s (%.3f ms)\n", name, bytes / t[2] / 1e6, t[2]);
        fflush(stdout);
    };
    const size_t src16 = total / 16;
    hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, src, src16); // varied data
    for (int grid : {512, 1024, 2048, 4096, 8192}) {
        printf("--- read only, grid %d ---\n", grid);
        timeit("16 B/lane, 1 load per trip", (double)total, [&] { hipLaunchKernelGGL(k_read_u<1>, dim3(grid), dim3(256), 0, 0, src, dst, src16); });
        timeit("16 B/lane, 2 loads per trip", (double)total, [&] { hipLaunchKernelGGL(k_read_u<2>, dim3(grid), dim3(256), 0, 0, src, dst, src16); });
        timeit("16 B/lane, 4 loads per trip", (double)total, [&] { hipLaunchKernelGGL(k_read_u<4>, dim3(grid), dim3(256), 0, 0, src, dst, src16); });
        timeit("16 B/lane, 8 loads per trip", (double)total, [&] { hipLaunchKernelGGL(k_read_u<8>, dim3(grid), dim3(256), 0, 0, src, dst, src16); });
        timeit("32 B/lane (2 x 16 B at a 32-B lane stride)", (double)total, [&] { hipLaunchKernelGGL(k_read_32, dim3(grid), dim3(256), 0, 0, src, dst, src16); });
        timeit("LDS-DMA ring, 8 x 1 KiB per wave, nt", (double)total, [&] { hipLaunchKernelGGL(k_read_dma<8>, dim3(grid), dim3(256), 0, 0, src, dst, src16); });
        timeit("LDS-DMA ring, 16 x 1 KiB per wave, nt", (double)total, [&] { hipLaunchKernelGGL(k_read_dma<16>, dim3(grid), dim3(256), 0, 0, src, dst, src16); });
    }
    timeit("hipMemcpyDtoD 8 GiB (read + write bytes)", 2.0 * total, [&] { CHK(hipMemcpyAsync(dst, src, total, hipMemcpyDeviceToDevice, 0)); });
    for (int grid : {2048}) {
        printf("--- grid %d ---\n", grid);
        timeit("read only, 1 stream", (double)total, [&] { hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(256), 0, 0, src, dst, src16); });
        timeit("write only", (double)total, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, dst, src16); });
        timeit("copy 1R:1W grid-stride", 2.0 * total, [&] { hipLaunchKernelGGL((k_mix<1, false>), dim3(grid), dim3(256), 0, 0, src, dst, src16); });
        timeit("copy 1R:1W grid-stride, nt store", 2.0 * total, [&] { hipLaunchKernelGGL((k_mix<1, true>), dim3(grid), dim3(256), 0, 0, src, dst, src16); });
        timeit("copy 1R:1W chunked per workgroup", 2.0 * total, [&] { hipLaunchKernelGGL((k_mix_chunk<1>), dim3(grid), dim3(256), 0, 0, src, dst, src16); });
        timeit("2R:1W grid-stride", 1.5 * total, [&] { hipLaunchKernelGGL((k_mix<2, false>), dim3(grid), dim3(256), 0, 0, src, dst, src16 / 2); });
        timeit("4R:1W grid-stride", 1.25 * total, [&] { hipLaunchKernelGGL((k_mix<4, false>), dim3(grid), dim3(256), 0, 0, src, dst, src16 / 4); });
        timeit("8R:1W grid-stride", 1.125 * total, [&] { hipLaunchKernelGGL((k_mix<8, false>), dim3(grid), dim3(256), 0, 0, src, dst, src16 / 8); });
        timeit("8R:1W grid-stride, nt store", 1.125 * total, [&] { hipLaunchKernelGGL((k_mix<8, true>), dim3(grid), dim3(256), 0, 0, src, dst, src16 / 8); });
        timeit("8R:1W chunked per workgroup", 1.125 * total, [&] { hipLaunchKernelGGL((k_mix_chunk<8>), dim3(grid), dim3(256), 0, 0, src, dst, src16 / 8); });
        timeit("8R:1W stores in 16 KiB bursts through LDS", 1.125 * total, [&] { hipLaunchKernelGGL((k_mix_burst<8, 16>), dim3(grid), dim3(256), 0, 0, src, dst, src16 / 8); });
        timeit("8R:1W stores in 64 KiB bursts through LDS", 1.125 * total, [&] { hipLaunchKernelGGL((k_mix_burst<8, 64>), dim3(grid), dim3(256), 0, 0, src, dst, src16 / 8); });
        timeit("8 streams read only", (double)total, [&] { hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, 0, src, dst, src16 / 8); });
    }
    CHK(hipGetLastError());
    return 0;
}
