// tools/streambench.hip -- A/B variants of the fused fold+sum streaming kernel (one process, interleaved).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Izk_amd/csrc tools/streambench.hip -o tools/streambench
#include "fq_dev.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ fq ld(const fq* p) { return NT ? fq_load_stream(p) : fq_load(p); }
template <bool NT> __device__ __forceinline__ void st(fq* p, const fq& a) {
    if (NT) {
        u32x4* q = reinterpret_cast<u32x4*>(p);
        u32x4 lo = {a.v[0], a.v[1], a.v[2], a.v[3]}, hi = {a.v[4], a.v[5], a.v[6], a.v[7]};
        __builtin_nontemporal_store(lo, q); __builtin_nontemporal_store(hi, q + 1);
    } else fq_store(p, a);
}
struct Quad { fq a0, a1, b0, b1; };
template <bool NT> __device__ __forceinline__ void load_quad(Quad& Q, const fq* __restrict__ src, size_t q, size_t i) {
    Q.a0 = ld<NT>(src + i); Q.a1 = ld<NT>(src + 2 * q + i); Q.b0 = ld<NT>(src + q + i); Q.b1 = ld<NT>(src + 3 * q + i);
}
// MATH: 0 = xor only (memory pattern ceiling), 1 = fold without sums, 2 = fold + sums (the real kernel)
template <int MATH, bool NTS> __device__ __forceinline__ void work(const Quad& Q, fq* __restrict__ dst, size_t q, size_t i, const fq& r, fq (&acc)[2]) {
    fq o0, o1;
    if (MATH == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { o0.v[j] = Q.a0.v[j] ^ Q.a1.v[j]; o1.v[j] = Q.b0.v[j] ^ Q.b1.v[j]; }
    } else { o0 = fq_fold(Q.a0, Q.a1, r); o1 = fq_fold(Q.b0, Q.b1, r); }
    st<NTS>(dst + i, o0); st<NTS>(dst + q + i, o1);
    if (MATH == 2) { acc[0] = fq_add(acc[0], o0); acc[1] = fq_add(acc[1], o1); }
}
template <int MATH, bool NTL, bool NTS, bool DB, int WPS>
__global__ void __launch_bounds__(256, WPS) k_stream(const fq* __restrict__ src, fq* __restrict__ dst, size_t q, fq r, fq* __restrict__ partials) {
    fq acc[2] = {fq_zero(), fq_zero()};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (DB) {
        Quad A, B;
        if (i < q) load_quad<NTL>(A, src, q, i);
        while (i < q) {
            size_t j = i + stride;
            if (j < q) load_quad<NTL>(B, src, q, j);
            asm volatile("" ::: "memory");
            work<MATH, NTS>(A, dst, q, i, r, acc);
            i = j; if (i >= q) break;
            j = i + stride;
            if (j < q) load_quad<NTL>(A, src, q, j);
            asm volatile("" ::: "memory");
            work<MATH, NTS>(B, dst, q, i, r, acc);
            i = j;
        }
    } else {
        for (; i < q; i += stride) { Quad A; load_quad<NTL>(A, src, q, i); work<MATH, NTS>(A, dst, q, i, r, acc); }
    }
    if (MATH == 2) {
        fq s = fq_add(fq_wave_sum(acc[0]), fq_wave_sum(acc[1]));
        if ((threadIdx.x & 63) == 0) fq_store(partials + blockIdx.x * 4 + (threadIdx.x >> 6), s);
    }
}


// diagnostic: same byte counts, different access shapes.  SHAPE 0: plain float4 copy 2 reads : 1 write on ONE read stream
// SHAPE 1: 4 read streams at power-of-two offsets, 16-B-per-lane fully coalesced (1 KiB per wave instruction)
// SHAPE 2: as 1 but the four streams padded apart by odd multiples of 4 KiB + 256 B
// SHAPE 3: 32-B-per-lane loads (the real pattern), block-contiguous chunks instead of grid-stride
template <int SHAPE>
__global__ void __launch_bounds__(256) k_shape(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t q16 /* 16-byte units per quarter */, size_t pad16) {
    const size_t stride = (size_t)gridDim.x * 256;
    if (SHAPE == 0) { // reads 4*q16 units contiguous, writes 2*q16
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 2 * q16; i += stride) {
            u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + 2 * q16 + i);
            dst[i] = a ^ b;
        }
    } else if (SHAPE == 1 || SHAPE == 2) {
        const size_t off = (SHAPE == 2) ? pad16 : 0;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < q16; i += stride) {
            u32x4 a0 = __builtin_nontemporal_load(src + i), a1 = __builtin_nontemporal_load(src + 2 * q16 + 2 * off + i);
            u32x4 b0 = __builtin_nontemporal_load(src + q16 + off + i), b1 = __builtin_nontemporal_load(src + 3 * q16 + 3 * off + i);
            dst[i] = a0 ^ a1; dst[q16 + off + i] = b0 ^ b1;
        }
    } else if (SHAPE == 3) {
        const size_t qe = q16 / 2; // elements per quarter
        const size_t per_block = (qe + gridDim.x - 1) / gridDim.x;
        const size_t lo = (size_t)blockIdx.x * per_block, hi = lo + per_block < qe ? lo + per_block : qe;
        const fq* s = reinterpret_cast<const fq*>(src); fq* d = reinterpret_cast<fq*>(dst);
        for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
            fq a0 = fq_load_stream(s + i), a1 = fq_load_stream(s + 2 * qe + i), b0 = fq_load_stream(s + qe + i), b1 = fq_load_stream(s + 3 * qe + i);
            fq o0, o1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0.v[j] = a0.v[j] ^ a1.v[j]; o1.v[j] = b0.v[j] ^ b1.v[j]; }
            fq_store(d + i, o0); fq_store(d + qe + i, o1);
        }
    }
}

struct Variant { const char* name; void (*launch)(int, const fq*, fq*, size_t, fq, fq*); };
#define V(NAME, ...) {NAME, [](int g, const fq* s, fq* d, size_t q, fq r, fq* p) { hipLaunchKernelGGL((k_stream<__VA_ARGS__>), dim3(g), dim3(256), 0, 0, s, d, q, r, p); }}

int main(int argc, char** argv) {
    int logn = argc > 1 ? atoi(argv[1]) : 26;
    size_t n = (size_t)1 << logn, q = n / 4;
    fq *src, *dst, *part;
    CHK(hipMalloc(&src, n * 32)); CHK(hipMalloc(&dst, n / 2 * 32)); CHK(hipMalloc(&part, 8192 * 4 * 32));
    CHK(hipMemset(src, 0x11, n * 32));
    fq r; for (int j = 0; j < 8; ++j) r.v[j] = 0x12345678u + j; r.v[7] &= 0x0fffffffu;
    std::vector<Variant> vs = {
        V("xor  nt-ld     db  g2048", 0, true, false, true, 1),
        V("xor  plain-ld  db  g2048", 0, false, false, true, 1),
        V("xor  nt-ld nt-st db     ", 0, true, true, true, 1),
        V("xor  nt-ld     nodb     ", 0, true, false, false, 1),
        V("fold nt-ld     db       ", 1, true, false, true, 1),
        V("f+s  nt-ld     db       ", 2, true, false, true, 1),
        V("f+s  plain-ld  db       ", 2, false, false, true, 1),
        V("f+s  nt-ld nt-st db     ", 2, true, true, true, 1),
        V("f+s  nt-ld     nodb     ", 2, true, false, false, 1),
        V("f+s  nt-ld     db  wps2 ", 2, true, false, true, 2),
        V("f+s  nt-ld     nodb wps2", 2, true, false, false, 2),
    };
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    double bytes = (double)n * 48;
    for (int grid : {2048}) {
        printf("--- 2^%d entries, grid %d blocks x 256 (GB/s algorithmic, 48 B/entry; median of 5) ---\n", logn, grid);
        for (auto& v : vs) {
            std::vector<float> t;
            for (int rep = 0; rep < 6; ++rep) {
                CHK(hipEventRecord(a)); v.launch(grid, src, dst, q, r, part); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
                float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (rep) t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            printf("%-28s %8.1f GB/s  (%.3f ms)\n", v.name, bytes / t[2] / 1e6, t[2]);
        }
    }
    {
        const char* names[] = {"copy 1 stream 2R:1W 16B/lane", "4 streams pow2, 16B/lane coalesced", "4 streams padded, 16B/lane coalesced", "4 streams pow2, 32B/lane, block-chunked"};
        size_t q16 = q * 2, pad16 = (4096 * 5 + 256) / 16;
        for (int grid : {2048, 8192}) {
            printf("--- shapes, grid %d ---\n", grid);
            for (int sh = 0; sh < 4; ++sh) {
                std::vector<float> t;
                for (int rep = 0; rep < 6; ++rep) {
                    CHK(hipEventRecord(a));
                    if (sh == 0) hipLaunchKernelGGL(k_shape<0>, dim3(grid), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)dst, q16 - 3 * pad16, pad16);
                    if (sh == 1) hipLaunchKernelGGL(k_shape<1>, dim3(grid), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)dst, q16 - 3 * pad16, pad16);
                    if (sh == 2) hipLaunchKernelGGL(k_shape<2>, dim3(grid), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)dst, q16 - 3 * pad16, pad16);
                    if (sh == 3) hipLaunchKernelGGL(k_shape<3>, dim3(grid), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)dst, q16 - 3 * pad16, pad16);
                    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
                    float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (rep) t.push_back(ms);
                }
                std::sort(t.begin(), t.end());
                printf("%-42s %8.1f GB/s (%.3f ms)\n", names[sh], (double)(q16 - 3 * pad16) * 16 * 6 / t[2] / 1e6, t[2]);
            }
        }
    }
    CHK(hipGetLastError());
    return 0;
}
