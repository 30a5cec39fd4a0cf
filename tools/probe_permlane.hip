#include <hip/hip_runtime.h>
typedef unsigned u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(u32* out) {
    u32 a = threadIdx.x, b = threadIdx.x + 100;
    u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    u32x2 q = __builtin_amdgcn_permlane16_swap(r[0], r[1], false, false);
    out[threadIdx.x] = q[0];
    out[64 + threadIdx.x] = q[1];
    out[128 + threadIdx.x] = r[0];
    out[192 + threadIdx.x] = r[1];
}
int main() {
    u32* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    u32 h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int part = 2; part < 4; ++part) { printf("r[%d]:", part - 2); for (int i = 0; i < 64; ++i) printf(" %u", h[part * 64 + i]); printf("\n"); }
    for (int part = 0; part < 2; ++part) { printf("q[%d]:", part); for (int i = 0; i < 64; ++i) printf(" %u", h[part * 64 + i]); printf("\n"); }
}
