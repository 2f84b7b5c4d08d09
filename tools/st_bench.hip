// Developer microbenchmark 3: cost of K1's 64 MiB of output stores on their own, by pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// P0: K1 pattern: 2048 waves, each 128 x (64 lanes x 4 B) into its own 32 KiB region, back to back
// P1: same bytes, dwordx4 per lane (32 x 1 KiB per wave)
// P2: plain streaming fill, 256-thread blocks, dwordx4, grid covers 64 MiB
// P3: K1 pattern but paced with s_sleep (128 stores spread over ~0.17 ms)
__global__ __launch_bounds__(64) void p0(uint32_t *out, int pace)
{
    uint32_t *ob = out + (size_t)blockIdx.x * 8192;
    for (int w = 0; w < 128; ++w) { ob[w * 64 + threadIdx.x] = w; if (pace) __builtin_amdgcn_s_sleep(40); }
}
__global__ __launch_bounds__(64) void p1(uint4 *out)
{
    uint4 *ob = out + (size_t)blockIdx.x * 2048;
    for (int w = 0; w < 32; ++w) ob[w * 64 + threadIdx.x] = make_uint4(w, w, w, w);
}
__global__ __launch_bounds__(256) void p2(uint4 *out) { out[(size_t)blockIdx.x * 256 + threadIdx.x] = make_uint4(1, 2, 3, 4); }
int main()
{
    uint32_t *o; hipMalloc(&o, 64 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 4; ++v) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(p0, dim3(2048), dim3(64), 0, 0, o, 0);
            if (v == 1) hipLaunchKernelGGL(p1, dim3(2048), dim3(64), 0, 0, (uint4 *)o);
            if (v == 2) hipLaunchKernelGGL(p2, dim3((64 << 20) / 16 / 256), dim3(256), 0, 0, (uint4 *)o);
            if (v == 3) hipLaunchKernelGGL(p0, dim3(2048), dim3(64), 0, 0, o, 1);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("pattern %d: %.4f ms  %.1f GB/s\n", v, best, (64 << 20) / best / 1e6);
    }
    return 0;
}
