// probe_sync.hip -- how long one "kernel result -> host -> next kernel" round trip takes on this box, two ways:
//   (a) the kernel stores its result into mapped pinned memory, the host calls hipStreamSynchronize, launches the next
//   (b) the kernel stores the result, fences at system scope and stores a sequence number; the host polls that word
//   (c) as (b), but the NEXT kernel is already enqueued and its workgroups wait (bounded: 50 ms by s_memrealtime) for a "go" word the
//       host writes into mapped memory once it has the previous result -- the launch path is off the critical path
//   (d) as (c) with a one-workgroup relay kernel doing the waiting and the 256-workgroup kernel enqueued behind it
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
// waits until *go >= seq; false on timeout (5e6 ticks of the 100 MHz counter = 50 ms): every wave reaches an exit
__device__ bool wait_go(const volatile uint32_t* go, uint32_t seq) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load((const uint32_t*)go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) return false;
        __builtin_amdgcn_s_sleep(8);
    }
    return true;
}
__global__ void k_work_wait(const uint32_t* __restrict__ src, uint32_t* __restrict__ host_out, unsigned* ticket, volatile uint32_t* flag, uint32_t seq, int n,
                            const volatile uint32_t* go, uint32_t* timeouts) {
    __shared__ int ok;
    if (go) {
        if (threadIdx.x == 0) ok = wait_go(go, seq);
        __syncthreads();
        if (!ok) { if (threadIdx.x == 0) atomicAdd(timeouts, 1u); }
    }
    uint32_t acc = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += src[i];
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) {
        host_out[blockIdx.x] = acc;
        __threadfence_system();
        const unsigned t = atomicAdd(ticket, 1u);
        if (t == gridDim.x - 1) { *ticket = 0; __threadfence_system(); *flag = seq; }
    }
}
__global__ void k_relay(const volatile uint32_t* go, uint32_t seq, uint32_t* dev_word, uint32_t* timeouts) {
    if (threadIdx.x == 0) {
        if (!wait_go(go, seq)) atomicAdd(timeouts, 1u);
        *dev_word = seq;
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
    volatile uint32_t* go = host + 3072; uint32_t* go_dev = host_dev + 3072;
    uint32_t* timeouts; CK(hipMalloc(&timeouts, 8)); CK(hipMemset(timeouts, 0, 8));
    uint32_t* dev_word = timeouts + 1;
    for (int mode = 2; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            const uint32_t base = (uint32_t)(mode * 1000000 + rep * 10000);
            *go = 0; *flag = 0;
            auto enqueue = [&](uint32_t seq) {
                if (mode == 3) {
                    hipLaunchKernelGGL(k_relay, dim3(1), dim3(64), 0, st, go_dev, seq, dev_word, timeouts);
                    hipLaunchKernelGGL(k_work_wait, dim3(blocks), dim3(64), 0, st, src, host_dev, ticket, flag_dev, seq, n, (const volatile uint32_t*)nullptr, timeouts);
                } else {
                    hipLaunchKernelGGL(k_work_wait, dim3(blocks), dim3(64), 0, st, src, host_dev, ticket, flag_dev, seq, n, go_dev, timeouts);
                }
            };
            enqueue(base + 1);
            auto t0 = std::chrono::steady_clock::now();
            uint64_t sink = 0;
            for (int i = 1; i <= iters; ++i) {
                __atomic_store_n((uint32_t*)go, base + i, __ATOMIC_RELEASE); // "the challenge is known": pass i may start
                if (i < iters) enqueue(base + i + 1);                     // the next pass goes into the queue while pass i runs
                while (__atomic_load_n((const uint32_t*)flag, __ATOMIC_ACQUIRE) != base + i) {}
                sink += host[i & 255];
            }
            CK(hipStreamSynchronize(st));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
            uint32_t to = 0; CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
            printf("%s rep %d: %.2f us per round trip (timeouts %u, sink %llu)\n", mode == 2 ? "pre-enqueued, workgroups wait for go" : "pre-enqueued, relay kernel waits for go", rep, us, to,
                   (unsigned long long)sink);
        }
    }
    return 0;
}
