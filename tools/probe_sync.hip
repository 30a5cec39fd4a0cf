// probe_sync.hip -- how long one "kernel result -> host -> next kernel" round trip takes on this box, two ways:
//   (a) the kernel stores its result into mapped pinned memory, the host calls hipStreamSynchronize, launches the next
//   (b) the kernel stores the result, fences at system scope and stores a sequence number; the host polls that word
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_sync tools/probe_sync.hip ; run: tools/probe_sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_work(const uint32_t* __restrict__ src, uint32_t* __restrict__ host_out, unsigned* ticket, volatile uint32_t* flag, uint32_t seq, int n) {
    uint32_t acc = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += src[i];
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) {
        host_out[blockIdx.x] = acc;
        if (flag) {
            __threadfence_system();
            const unsigned t = atomicAdd(ticket, 1u);
            if (t == gridDim.x - 1) { *ticket = 0; __threadfence_system(); *flag = seq; }
        }
    }
}
int main() {
    const int n = 1 << 20, blocks = 256, iters = 2000;
    uint32_t* src; CK(hipMalloc(&src, n * 4)); CK(hipMemset(src, 1, n * 4));
    unsigned* ticket; CK(hipMalloc(&ticket, 4)); CK(hipMemset(ticket, 0, 4));
    uint32_t* host; CK(hipHostMalloc(&host, 4096 * 4, hipHostMallocMapped)); 
    uint32_t* host_dev; CK(hipHostGetDevicePointer((void**)&host_dev, host, 0));
    volatile uint32_t* flag = host + 2048; uint32_t* flag_dev = host_dev + 2048;
    *flag = 0;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            uint64_t sink = 0;
            for (int i = 1; i <= iters; ++i) {
                const uint32_t seq = (uint32_t)(mode * 1000000 + rep * 10000 + i);
                hipLaunchKernelGGL(k_work, dim3(blocks), dim3(64), 0, st, src, host_dev, ticket, mode ? flag_dev : nullptr, seq, n);
                if (mode == 0) { CK(hipStreamSynchronize(st)); }
                else { while (__atomic_load_n((const uint32_t*)flag, __ATOMIC_ACQUIRE) != seq) {} }
                sink += host[i & 255];
            }
            CK(hipStreamSynchronize(st));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
            printf("%s rep %d: %.2f us per round trip (sink %llu)\n", mode ? "poll flag" : "stream sync", rep, us, (unsigned long long)sink);
        }
    }
    return 0;
}
