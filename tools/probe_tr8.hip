// tools/probe_tr8.hip -- what does ds_read_b64_tr_b8 deliver?  Every LDS byte holds its own address (low byte in run 0, high
// byte in run 1); every lane supplies an address; the 8 result bytes of each lane are printed as the LDS addresses they came from.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_tr8.hip -o tools/probe_tr8
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
__global__ void k(const uint32_t* addr_in, u64* out, int plane) {
    __shared__ unsigned char lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = plane ? (unsigned char)(i >> 8) : (unsigned char)(i & 0xff);
    __syncthreads();
    const uint32_t a = addr_in[threadIdx.x] + (uint32_t)(uintptr_t)lds;
    u64 r;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    out[threadIdx.x] = r;
}
int main(int argc, char** argv) {
    const int stride = argc > 1 ? atoi(argv[1]) : 32; // row stride in bytes
    std::vector<uint32_t> addr(64);
    // hypothesis (by analogy with tr_b16): per 16-lane group a block of 8 rows x 16 byte-columns; lane 2q+p supplies row q, bytes 8p..8p+7
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15, q = i >> 1, p = i & 1;
        addr[l] = (uint32_t)(g * 8 * stride + q * stride + 8 * p);
    }
    uint32_t* d_a;
    u64* d_o;
    hipMalloc(&d_a, 256);
    hipMalloc(&d_o, 512);
    hipMemcpy(d_a, addr.data(), 256, hipMemcpyHostToDevice);
    std::vector<u64> lo(64), hi(64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_a, d_o, 0);
    hipMemcpy(lo.data(), d_o, 512, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_a, d_o, 1);
    hipMemcpy(hi.data(), d_o, 512, hipMemcpyDeviceToHost);
    printf("row stride %d bytes; lane: supplied address -> source addresses of its 8 result bytes (as row:col with that stride)\n", stride);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d addr %4u ->", l, addr[l]);
        for (int b = 0; b < 8; ++b) {
            const unsigned src = (unsigned)((lo[l] >> (8 * b)) & 0xff) | ((unsigned)((hi[l] >> (8 * b)) & 0xff) << 8);
            printf(" %u:%u", src / stride, src % stride);
        }
        printf("\n");
    }
    return 0;
}
