// tools/mfbench.hip -- ablations of the 16-stream m-variable fold pass (k_multifold_lazy / k_multifold_roll):
//   mem  : the same 16 reads + 1 write per output with XOR instead of field arithmetic (memory ceiling of the pattern)
//   alu  : the same arithmetic per output on register data, no global loads (ALU ceiling)
//   real : the library kernels
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Izk_amd/csrc tools/mfbench.hip -o tools/mfbench
#include "../zk_amd/csrc/kernels.hip"
#include "fold_variants.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
using namespace zkk;

// MODE 0: grid-stride over outputs (the library's mapping); MODE 1: each workgroup owns a contiguous chunk
// STORE: 0 none (xor-reduce kept in a register, one store per thread at the end), 1 plain, 2 nontemporal
template <int NB, int INFLIGHT, int STORE>
__global__ void __launch_bounds__(256) k_mem2(const fq* __restrict__ src, fq* __restrict__ dst, size_t Lout, size_t sstride) {
    fq keep = fq_zero();
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < Lout; i += step) {
        fq o = fq_zero();
#pragma unroll
        for (int b0 = 0; b0 < NB; b0 += INFLIGHT) {
            fq t[INFLIGHT];
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) t[k] = fq_load_stream(src + (size_t)(b0 + k) * sstride + i);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) o.v[j] ^= t[k].v[j];
        }
        if (STORE == 1) fq_store(dst + i, o);
        else if (STORE == 3) {
            // transpose through LDS so that each store instruction writes 1 KiB contiguous (16 B per lane, no gaps)
            __shared__ u32x4 tr[256 * 2];
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            u32x4* w = tr + wave * 128;
            u32x4 lo = {o.v[0], o.v[1], o.v[2], o.v[3]}, hi = {o.v[4], o.v[5], o.v[6], o.v[7]};
            w[2 * lane] = lo;
            w[2 * lane + 1] = hi;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            u32x4 c0 = w[lane], c1 = w[64 + lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            u32x4* g = reinterpret_cast<u32x4*>(dst + (i - lane)); // the wave's 64 outputs are contiguous; Lout is a multiple of 64 here
            g[lane] = c0;
            g[64 + lane] = c1;
        } else if (STORE >= 4) {
            // explicit cache-policy bits on the store (gfx940+: sc0, sc1, nt)
            u32x4 lo = {o.v[0], o.v[1], o.v[2], o.v[3]}, hi = {o.v[4], o.v[5], o.v[6], o.v[7]};
            const fq* p = dst + i;
            if (STORE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0" ::"v"(p), "v"(lo), "v"(hi) : "memory");
            if (STORE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" ::"v"(p), "v"(lo), "v"(hi) : "memory");
            if (STORE == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1" ::"v"(p), "v"(lo), "v"(hi) : "memory");
            if (STORE == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1 nt" ::"v"(p), "v"(lo), "v"(hi) : "memory");
            if (STORE == 8) asm volatile("global_store_dwordx4 %0, %1, off nt\n\tglobal_store_dwordx4 %0, %2, off offset:16 nt" ::"v"(p), "v"(lo), "v"(hi) : "memory");
        } else if (STORE == 2) {
            u32x4* q = reinterpret_cast<u32x4*>(dst + i);
            u32x4 lo = {o.v[0], o.v[1], o.v[2], o.v[3]}, hi = {o.v[4], o.v[5], o.v[6], o.v[7]};
            __builtin_nontemporal_store(lo, q); __builtin_nontemporal_store(hi, q + 1);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) keep.v[j] ^= o.v[j];
        }
    }
    if (STORE == 0) fq_store(dst + (size_t)blockIdx.x * 256 + threadIdx.x, keep);
}

template <int NB, int MODE, int INFLIGHT>
__global__ void __launch_bounds__(256) k_mem(const fq* __restrict__ src, fq* __restrict__ dst, size_t Lout) {
    size_t i, end, step;
    if (MODE == 0) { i = (size_t)blockIdx.x * 256 + threadIdx.x; end = Lout; step = (size_t)gridDim.x * 256; }
    else { const size_t per = (Lout + gridDim.x - 1) / gridDim.x; i = (size_t)blockIdx.x * per + threadIdx.x; end = min(Lout, (size_t)(blockIdx.x + 1) * per); step = 256; }
    for (; i < end; i += step) {
        fq o = fq_zero();
#pragma unroll
        for (int b0 = 0; b0 < NB; b0 += INFLIGHT) {
            fq t[INFLIGHT];
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) t[k] = fq_load_stream(src + (size_t)(b0 + k) * Lout + i);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) o.v[j] ^= t[k].v[j];
        }
        fq_store(dst + i, o);
    }
}

// the wide pass (k_multifold_wide) with knobs: G elements per load buffer; ALU 0 = xor only, 1 = full inner product,
// 2 = half the limb products (sensitivity to the arithmetic); weights from LDS as in the library kernel
template <int G, int ALU, int WS = 0>
__global__ void __launch_bounds__(256) k_wide_exp(const fq* __restrict__ src, fq* __restrict__ dst, size_t Lout, int m, const u32* __restrict__ wsrc) {
    __shared__ u32 wl[256 * 10];
    if (!WS) {
        for (int t = threadIdx.x; t < (10 << m); t += 256) wl[t] = wsrc[t];
        __syncthreads();
    }
    const int ng = (1 << m) / G;
    const size_t stride = (size_t)gridDim.x * 256, gstep = (size_t)G * Lout;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    fq A[G], B[G];
    if (i < Lout) lazy_load<G>(A, src, Lout, i, 0);
    while (i < Lout) {
        const size_t n = i + stride;
        u64 acc[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) acc[k] = 0;
        const fq* __restrict__ p = src + i;
        const u32* __restrict__ w = WS ? wsrc : wl;
        for (int g = 0; g < ng; g += 2) {
            lazy_load<G>(B, p + gstep, Lout, 0, 0);
#pragma unroll
            for (int b = 0; b < G; ++b) {
                if (ALU == 1) fq28_mac(acc, A[b], w + b * 10);
                else if (ALU == 2) { u32 l[10]; fq28_unpack(A[b], l);
#pragma unroll
                    for (int j = 0; j < 5; ++j) { const u32 wj = w[b * 10 + j];
#pragma unroll
                        for (int q = 0; q < 10; ++q) acc[q + j] += (u64)l[q] * wj; } }
                else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[q] ^= A[b].v[q];
                }
            }
            if (g + 2 < ng) lazy_load<G>(A, p + 2 * gstep, Lout, 0, 0);
            else if (n < Lout) lazy_load<G>(A, src, Lout, n, 0);
#pragma unroll
            for (int b = 0; b < G; ++b) {
                if (ALU == 1) fq28_mac(acc, B[b], w + (G + b) * 10);
                else if (ALU == 2) { u32 l[10]; fq28_unpack(B[b], l);
#pragma unroll
                    for (int j = 0; j < 5; ++j) { const u32 wj = w[(G + b) * 10 + j];
#pragma unroll
                        for (int q = 0; q < 10; ++q) acc[q + j] += (u64)l[q] * wj; } }
                else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[q] ^= B[b].v[q];
                }
            }
            if (ALU) {
#pragma unroll
                for (int k = 0; k < 19; ++k) { acc[k + 1] += acc[k] >> 28; acc[k] &= FQ28_MASK; }
            }
            p += 2 * gstep;
            w += 2 * G * 10;
        }
        fq o;
        if (ALU) o = fq28_redc(acc);
        else {
#pragma unroll
            for (int q = 0; q < 8; ++q) o.v[q] = (u32)acc[q];
        }
        fq_store(dst + i, o);
        i = n;
    }
}

template <int M>
__global__ void __launch_bounds__(256) k_alu(fq* __restrict__ io, W28<M> W, int outputs_per_thread) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    fq a = fq_load(io + tid);
    fq sum = fq_zero();
    for (int it = 0; it < outputs_per_thread; ++it) {
        u64 acc[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) acc[k] = 0;
#pragma unroll
        for (int b = 0; b < (1 << M); ++b) {
            a.v[b & 7] ^= (u32)it; // keep the unpack in the loop
            a.v[7] &= 0x3fffffffu;
            fq28_mac(acc, a, W.w[b]);
        }
        fq o = fq28_redc(acc);
        sum = fq_add(sum, o);
    }
    fq_store(io + tid, sum);
}

template <class F>
static float timeit(F&& f, hipEvent_t a, hipEvent_t b) {
    std::vector<float> t;
    for (int rep = 0; rep < 6; ++rep) {
        CHK(hipEventRecord(a)); f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (rep) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[2];
}

int main(int argc, char** argv) {
    const int logn = argc > 1 ? atoi(argv[1]) : 28;
    const size_t n = (size_t)1 << logn, Lout = n >> 4;
    fq *src, *dst, *part, *sums;
    CHK(hipMalloc(&src, n * 32));
    CHK(hipMalloc(&dst, n * 32)); // sized for the widest ablation (1 stream: as many outputs as inputs)
    CHK(hipMalloc(&part, 2048 * 3 * 32)); // >= 2 * kMaxBlocks partials of the wide pass
    CHK(hipMalloc(&sums, 256 * 32)); // up to 2^8 block sums
    launch_generate(src, 1, 0, 1, n, 0);
    CHK(hipDeviceSynchronize());
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const double bytes = (double)n * 32 + (double)Lout * 32;
    uint32_t w28[160];
    for (int i = 0; i < 160; ++i) w28[i] = (0x9E3779B9u * (i + 1)) & 0x0fffffffu;
    W28<4> W;
    for (int b2 = 0; b2 < 16; ++b2) for (int j = 0; j < 10; ++j) W.w[b2][j] = w28[b2 * 10 + j];

    printf("--- m = 4 fold pass over 2^%d entries: 16 read streams + 1 write stream (GB/s algorithmic; median of 5) ---\n", logn);
    for (int grid : {2048, 4096, 8192}) {
        float ms;
        ms = timeit([&] { hipLaunchKernelGGL((k_mem<16, 0, 4>), dim3(grid), dim3(256), 0, 0, src, dst, Lout); }, a, b);
        printf("mem  xor grid-stride  4 in flight  grid %5d  %8.1f GB/s (%.3f ms)\n", grid, bytes / ms / 1e6, ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_mem<16, 0, 8>), dim3(grid), dim3(256), 0, 0, src, dst, Lout); }, a, b);
        printf("mem  xor grid-stride  8 in flight  grid %5d  %8.1f GB/s (%.3f ms)\n", grid, bytes / ms / 1e6, ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_mem<16, 0, 16>), dim3(grid), dim3(256), 0, 0, src, dst, Lout); }, a, b);
        printf("mem  xor grid-stride 16 in flight  grid %5d  %8.1f GB/s (%.3f ms)\n", grid, bytes / ms / 1e6, ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_mem<16, 1, 8>), dim3(grid), dim3(256), 0, 0, src, dst, Lout); }, a, b);
        printf("mem  xor wg-chunked   8 in flight  grid %5d  %8.1f GB/s (%.3f ms)\n", grid, bytes / ms / 1e6, ms);
    }
    printf("--- stream-count / store ablation, grid 2048, same 2^%d-entry source (GB/s over bytes actually moved) ---\n", logn);
#define RUN2(NBv, INF, ST, LO, SS, label)                                                                                       \
    {                                                                                                                           \
        float ms = timeit([&] { hipLaunchKernelGGL((k_mem2<NBv, INF, ST>), dim3(2048), dim3(256), 0, 0, src, dst, (size_t)(LO), (size_t)(SS)); }, a, b); \
        double by = (double)(LO) * 32 * NBv + (ST ? (double)(LO) * 32 : 0);                                                     \
        printf("%-58s %8.1f GB/s (%.3f ms)\n", label, by / ms / 1e6, ms);                                                        \
    }
    RUN2(16, 8, 0, Lout, Lout, "16 streams, no store");
    RUN2(16, 8, 1, Lout, Lout, "16 streams, plain store");
    RUN2(16, 8, 2, Lout, Lout, "16 streams, nontemporal store");
    RUN2(16, 8, 3, Lout, Lout, "16 streams, LDS-transposed contiguous store");
    RUN2(16, 8, 4, Lout, Lout, "16 streams, store sc0");
    RUN2(16, 8, 5, Lout, Lout, "16 streams, store sc1");
    RUN2(16, 8, 6, Lout, Lout, "16 streams, store sc0 sc1");
    RUN2(16, 8, 7, Lout, Lout, "16 streams, store sc0 sc1 nt");
    RUN2(16, 8, 8, Lout, Lout, "16 streams, store nt");
    RUN2(32, 8, 0, Lout / 2, Lout / 2, "32 streams, no store");
    RUN2(32, 8, 1, Lout / 2, Lout / 2, "32 streams, plain store");
    RUN2(64, 8, 0, Lout / 4, Lout / 4, "64 streams, no store");
    RUN2(64, 8, 1, Lout / 4, Lout / 4, "64 streams, plain store");
    RUN2(64, 8, 2, Lout / 4, Lout / 4, "64 streams, nontemporal store");
    RUN2(256, 8, 0, Lout / 16, Lout / 16, "256 streams, no store");
    RUN2(256, 8, 1, Lout / 16, Lout / 16, "256 streams, plain store");
    RUN2(256, 8, 2, Lout / 16, Lout / 16, "256 streams, nontemporal store");
    RUN2(8, 8, 0, 2 * Lout, 2 * Lout, "8 streams, no store");
    RUN2(8, 8, 1, 2 * Lout, 2 * Lout, "8 streams, plain store");
    RUN2(4, 4, 0, 4 * Lout, 4 * Lout, "4 streams, no store");
    RUN2(4, 4, 1, 4 * Lout, 4 * Lout, "4 streams, plain store");
    RUN2(2, 2, 0, 8 * Lout, 8 * Lout, "2 streams, no store");
    RUN2(2, 2, 1, 8 * Lout, 8 * Lout, "2 streams, plain store");
    RUN2(1, 1, 0, 16 * Lout, 16 * Lout, "1 stream, no store");
    RUN2(16, 8, 0, Lout - 4096, Lout - 4096 + 264, "16 streams padded apart by 8 KiB + 256 B, no store");   // 16 * (Lout - 3832) < n
    RUN2(16, 8, 1, Lout - 4096, Lout - 4096 + 264, "16 streams padded apart by 8 KiB + 256 B, plain store");
    {
        const int per_thread = (int)(Lout / (2048 * 256));
        float ms = timeit([&] { hipLaunchKernelGGL((k_alu<4>), dim3(2048), dim3(256), 0, 0, dst, W, per_thread); }, a, b);
        printf("alu  16 x (unpack + 100 mad) + redc per output, %d outputs/thread, grid 2048: %.3f ms  (= %.1f GB/s if it were the pass)\n", per_thread, ms,
               bytes / ms / 1e6);
    }
    for (int group : {4, 2, 108, 104}) {
        float ms = timeit([&] { launch_multifold_lazy(4, src, dst, Lout, w28, 4, part, sums, 0); }, a, b);
        printf("real launch_multifold_lazy group %3d: %8.1f GB/s (%.3f ms)\n", group, bytes / ms / 1e6, ms);
    }
    u32* wbuf; CHK(hipMalloc(&wbuf, kWideWeightWords * 4));
    for (int m : {5, 6, 7, 8}) {
        uint64_t r[4 * 8], two24[4] = {1u << 24, 0, 0, 0}; // any values below p: timing only
        for (int i = 0; i < 32; ++i) r[i] = 0x0123456789abcdefull >> (i % 4 == 3 ? 8 : 0);
        const size_t lo = n >> m;
        for (int log_ob : {4, 8}) {
            float ms = timeit([&] { launch_multifold_wide(m, src, dst, lo, r, two24, wbuf, 19, 0, log_ob, part, sums, 0); }, a, b);
            printf("real launch_multifold_wide m = %d (%3d streams), %3d output blocks: %8.1f GB/s (%.3f ms)\n", m, 1 << m, 1 << log_ob,
                   ((double)n * 32 + (double)lo * 32) / ms / 1e6, ms);
        }
    }
    {
        u32* wdev; CHK(hipMalloc(&wdev, 2560 * 4));
        std::vector<u32> wh(2560);
        for (int i = 0; i < 2560; ++i) wh[i] = (0x9E3779B9u * (i + 1)) & 0x0fffffffu;
        CHK(hipMemcpy(wdev, wh.data(), 2560 * 4, hipMemcpyHostToDevice));
        const size_t lo = n >> 8;
#define RUNW(Gv, ALUv, GRID, label) RUNWS(Gv, ALUv, 0, GRID, label)
#define RUNWS(Gv, ALUv, WSv, GRID, label)                                                                                                  \
    {                                                                                                                                      \
        float ms = timeit([&] { hipLaunchKernelGGL((k_wide_exp<Gv, ALUv, WSv>), dim3(GRID), dim3(256), 0, 0, src, dst, lo, 8, wdev); }, a, b); \
        printf("wide exp m = 8, G = %d, %-18s grid %5d: %8.1f GB/s (%.3f ms)\n", Gv, label, GRID, ((double)n * 32 + (double)lo * 32) / ms / 1e6, ms); \
    }
        RUNW(4, 0, 2048, "xor only");
        RUNW(4, 2, 2048, "half the products");
        RUNW(4, 1, 2048, "full");
        RUNW(4, 1, 1024, "full");
        RUNW(4, 1, 4096, "full");
        RUNW(8, 0, 2048, "xor only");
        RUNW(8, 1, 2048, "full");
        RUNW(2, 1, 2048, "full");
        RUNW(2, 1, 4096, "full");
        RUNWS(2, 1, 1, 2048, "full, scalar w");
        RUNWS(2, 1, 1, 4096, "full, scalar w");
        RUNWS(4, 1, 1, 2048, "full, scalar w");
    }
    {
        float ms = timeit([&] { launch_block_sums(src, n, 8, part, sums, 0); }, a, b);
        printf("block_sums, 256 blocks: %8.1f GB/s (%.3f ms)\n", (double)n * 32 / ms / 1e6, ms);
    }
    {
        float ms = timeit([&] { launch_block_sums(src, n, 4, part, sums, 0); }, a, b);
        printf("block_sums (read only, 1 stream per workgroup): %8.1f GB/s (%.3f ms)\n", (double)n * 32 / ms / 1e6, ms);
    }
    CHK(hipGetLastError());
    return 0;
}
