// tools/microbench.hip -- standalone gfx950 microbenchmarks for the int32-ALU side of the roofline.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Izk_amd/csrc tools/microbench.hip -o tools/microbench
#include "fq_dev.hpp"
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int VARIANT>
__global__ void __launch_bounds__(256) k_probe(fq* __restrict__ io, int iters, fq rr) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    fq a = fq_load(io + i), b = a;
    b.v[0] ^= 0x05a5a5a5u; b.v[7] &= 0x0fffffffu; a.v[7] &= 0x0fffffffu;
    if (VARIANT == 0) for (int k = 0; k < iters; ++k) a = fq_mul(a, b);
    if (VARIANT == 1) for (int k = 0; k < iters; ++k) a = fq_mul_uniform(a, rr);
    if (VARIANT == 2) { // two independent chains
        fq c = b;
        for (int k = 0; k < iters; k += 2) { a = fq_mul(a, b); c = fq_mul(c, b); }
        a = fq_add(a, c);
    }
    if (VARIANT == 3) for (int k = 0; k < iters; ++k) a = fq_fold(a, b, rr); // sub_lazy + mul_uniform + add
    fq_store(io + i, a);
}

// raw instruction issue rates: 256 dependent / independent ops per loop trip
template <int OP>
__global__ void __launch_bounds__(256) k_inst(u32* out, int iters, u32 x, u32 y) {
    u32 t = threadIdx.x;
    u64 acc0 = t, acc1 = t + 1, acc2 = t + 2, acc3 = t + 3;
    u32 c0 = t, c1 = t + 5;
    for (int k = 0; k < iters; ++k) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            if (OP == 0) { // 4 independent mad_u64_u32 chains
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(c0), "v"(c1) : "vcc");
            } else if (OP == 1) { // one dependent chain of 4
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0"
                             : "+v"(acc0) : "v"(c0), "v"(c1) : "vcc");
            } else if (OP == 2) { // mad + addc pairs (the FIPS inner step)
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                             : "+v"(acc0), "+v"(c0) : "v"(c1), "v"((u32)acc1) : "vcc");
            } else if (OP == 3) { // 4 dependent v_add_u32
                asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(c0) : "v"(c1));
            } else if (OP == 4) { // 4 v_mul_lo_u32 independent
                u32 a = (u32)acc0, b = (u32)acc1, c = (u32)acc2, d = (u32)acc3;
                asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(c1));
                acc0 = a; acc1 = b; acc2 = c; acc3 = d;
            } else if (OP == 5) { // mad with SGPR operand, independent
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(c0), "s"(x) : "vcc");
            } else if (OP == 6) { // v_mul_u32_u24 / v_mad_u32_u24 full rate?
                u32 a = (u32)acc0, b = (u32)acc1, c = (u32)acc2, d = (u32)acc3;
                asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n\tv_mad_u32_u24 %1, %1, %4, %2\n\tv_mad_u32_u24 %2, %2, %4, %3\n\tv_mad_u32_u24 %3, %3, %4, %0"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(c1));
                acc0 = a; acc1 = b; acc2 = c; acc3 = d;
            } else if (OP == 7) { // v_mul_hi_u32 independent
                u32 a = (u32)acc0, b = (u32)acc1, c = (u32)acc2, d = (u32)acc3;
                asm volatile("v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\tv_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(c1));
                acc0 = a; acc1 = b; acc2 = c; acc3 = d;
            } else if (OP == 8) { // f64 fma independent
                double a = __longlong_as_double(acc0), b = __longlong_as_double(acc1), c = __longlong_as_double(acc2), d = __longlong_as_double(acc3);
                double m = __longlong_as_double(acc0 ^ 7);
                asm volatile("v_fma_f64 %0, %0, %4, %0\n\tv_fma_f64 %1, %1, %4, %1\n\tv_fma_f64 %2, %2, %4, %2\n\tv_fma_f64 %3, %3, %4, %3"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
                acc0 = __double_as_longlong(a); acc1 = __double_as_longlong(b); acc2 = __double_as_longlong(c); acc3 = __double_as_longlong(d);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (u32)(acc0 + acc1 + acc2 + acc3) + c0;
}

template <class F>
static float timeit(F&& f) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f();
    (void)hipEventRecord(a); f(); (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms;
}

int main() {
    const int maxblocks = 2048;
    fq* buf; u32* out;
    CHK(hipMalloc(&buf, (size_t)maxblocks * 256 * 32));
    CHK(hipMalloc(&out, (size_t)maxblocks * 256 * 4));
    std::vector<u32> h((size_t)maxblocks * 256 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (u32)(i * 2654435761u) >> ((i % 8 == 7) ? 4 : 0);
    CHK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    fq rr; for (int j = 0; j < 8; ++j) rr.v[j] = 0x12345678u + j; rr.v[7] &= 0x0fffffffu;
    const int iters = 1000;
    printf("== fq_mul throughput (G modmul/s); waves/SIMD = blocks*4/1024 when <= 8 ==\n");
    for (int blocks : {256, 512, 1024, 2048}) {
        float m0 = timeit([&] { hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(256), 0, 0, buf, iters, rr); });
        float m1 = timeit([&] { hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(256), 0, 0, buf, iters, rr); });
        float m2 = timeit([&] { hipLaunchKernelGGL(k_probe<2>, dim3(blocks), dim3(256), 0, 0, buf, iters, rr); });
        float m3 = timeit([&] { hipLaunchKernelGGL(k_probe<3>, dim3(blocks), dim3(256), 0, 0, buf, iters, rr); });
        double n = (double)blocks * 256 * iters;
        printf("blocks %4d: vv %.1f  uniform %.1f  vv-2chains %.1f  fold(sub+mulu+add) %.1f\n", blocks, n / m0 / 1e6, n / m1 / 1e6, n / m2 / 1e6, n / m3 / 1e6);
    }
    CHK(hipDeviceSynchronize());
    printf("== raw issue: cycles per wave-instruction per SIMD at 2.4 GHz nominal (blocks=2048 -> 8 waves/SIMD; 256 -> 1) ==\n");
    const char* names[] = {"mad_u64_u32 indep", "mad_u64_u32 dep", "mad+addc", "v_add_u32 dep", "v_mul_lo_u32", "mad_u64 sgpr", "v_mad_u32_u24", "v_mul_hi_u32", "v_fma_f64"};
    for (int blocks : {256, 1024, 2048}) {
        const int it = 200;
        float ms[9];
        ms[0] = timeit([&] { hipLaunchKernelGGL(k_inst<0>, dim3(blocks), dim3(256), 0, 0, out, it, 3u, 5u); });
        ms[1] = timeit([&] { hipLaunchKernelGGL(k_inst<1>, dim3(blocks), dim3(256), 0, 0, out, it, 3u, 5u); });
        ms[2] = timeit([&] { hipLaunchKernelGGL(k_inst<2>, dim3(blocks), dim3(256), 0, 0, out, it, 3u, 5u); });
        ms[3] = timeit([&] { hipLaunchKernelGGL(k_inst<3>, dim3(blocks), dim3(256), 0, 0, out, it, 3u, 5u); });
        ms[4] = timeit([&] { hipLaunchKernelGGL(k_inst<4>, dim3(blocks), dim3(256), 0, 0, out, it, 3u, 5u); });
        ms[5] = timeit([&] { hipLaunchKernelGGL(k_inst<5>, dim3(blocks), dim3(256), 0, 0, out, it, 3u, 5u); });
        ms[6] = timeit([&] { hipLaunchKernelGGL(k_inst<6>, dim3(blocks), dim3(256), 0, 0, out, it, 3u, 5u); });
        ms[7] = timeit([&] { hipLaunchKernelGGL(k_inst<7>, dim3(blocks), dim3(256), 0, 0, out, it, 3u, 5u); });
        ms[8] = timeit([&] { hipLaunchKernelGGL(k_inst<8>, dim3(blocks), dim3(256), 0, 0, out, it, 3u, 5u); });
        double waves_per_simd = blocks * 4.0 / 1024.0;
        for (int k = 0; k < 9; ++k) {
            double insts = (double)it * 64 * 4 * waves_per_simd; // wave-instructions per SIMD
            printf("blocks %4d  %-18s %.2f ns/inst/SIMD = %.2f cyc@2.4GHz\n", blocks, names[k], ms[k] * 1e6 / insts, ms[k] * 1e6 / insts * 2.4);
        }
    }
    CHK(hipDeviceSynchronize());
    return 0;
}
