// Developer probe: what does ds_read_b32 return for a byte address that is not a multiple of 4?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t *out)
{
    __shared__ uint32_t t[64];
    t[threadIdx.x] = 0x11111111u * (threadIdx.x & 15) + (threadIdx.x << 28);
    __syncthreads();
    uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)t + 8 + (threadIdx.x & 3);   // word 2, misaligned by 0..3
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x] = v;
    if (threadIdx.x < 8) out[64 + threadIdx.x] = t[threadIdx.x];
}
int main()
{
    uint32_t *d, h[72];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("table: "); for (int i = 0; i < 8; ++i) printf("%08x ", h[64 + i]); printf("\n");
    for (int i = 0; i < 4; ++i) printf("misalign %d -> %08x\n", i, h[i]);
    return 0;
}
