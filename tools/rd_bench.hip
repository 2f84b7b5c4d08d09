// Developer microbenchmark: how fast 2049 x 2 waves can read a 64 MiB buffer 1 KiB per wave-instruction with PF loads in
// flight, by address pattern.  mode 0: tile-major in lock-step (wave T reads T*32K + k*1K: all waves at the same k);
// mode 1: same, start chunk rotated per tile; mode 2: chunk-major (lock-step waves read one contiguous region);
// mode 3: like 0 with a per-tile XOR of the chunk index (a layout permutation K1 could write).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <int MODE, int NCH, int PF>
__global__ __launch_bounds__(128) void k(const v4u *src, uint32_t *out, uint32_t ntiles)
{
    const uint32_t T = blockIdx.x, lane = threadIdx.x & 63, v = threadIdx.x >> 6;
    v4u r[PF + 1];
    uint32_t acc = 0;
    auto addr = [&](uint32_t k) -> const v4u * {
        uint32_t kk = v * 16 + k;                 // chunk of the two-row stream (0..63): second row = next tile's, here the same tile
        uint32_t t = T + (kk >> 5); kk &= 31; if (t >= ntiles) t = 0;
        size_t ci;
        if (MODE == 0) ci = (size_t)t * 32 + kk;
        else if (MODE == 1) ci = (size_t)t * 32 + ((kk + t * 7) & 31);
        else if (MODE == 2) ci = (size_t)kk * ntiles + t;
        else ci = (size_t)t * 32 + (kk ^ (t & 31));
        return src + ci * 64 + lane;
    };
#pragma unroll
    for (int i = 0; i <= PF; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[i]) : "v"(addr(i)) : "memory");
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r[k % (PF + 1)]) : "n"(PF) : "memory");
        const v4u x = r[k % (PF + 1)];
        acc ^= x.x ^ x.y ^ x.z ^ x.w;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[k % (PF + 1)]) : "v"(addr(k + PF + 1 < NCH ? k + PF + 1 : NCH - 1)) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[T * 128 + threadIdx.x] = acc;
}
template <int MODE, int NCH, int PF>
void run(const v4u *d, uint32_t *o, uint32_t nt)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms;
    for (int rep = 0; rep < 12; ++rep) {
        hipExtLaunchKernelGGL((k<MODE, NCH, PF>), dim3(nt), dim3(128), 0, 0, e0, e1, 0, d, o, nt);
        hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); if (rep >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)nt * 2 * NCH * 1024;
    printf("mode %d chunks/wave %d pf %d: med %.4f ms min %.4f  %.2f TB/s requested\n", MODE, NCH, PF, ms[ms.size() / 2], ms[0], bytes / ms[ms.size() / 2] * 1e-9);
}
int main()
{
    const uint32_t nt = 2049;
    v4u *d; uint32_t *o;
    hipMalloc(&d, (size_t)(nt + 2) * 32768); hipMalloc(&o, nt * 128 * 4);
    hipMemset(d, 1, (size_t)(nt + 2) * 32768);
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL((k<0, 16, 4>), dim3(nt), dim3(128), 0, 0, d, o, nt);   // clock ramp
    hipDeviceSynchronize();
    run<0, 16, 4>(d, o, nt); run<1, 16, 4>(d, o, nt); run<2, 16, 4>(d, o, nt); run<3, 16, 4>(d, o, nt);
    run<0, 38, 4>(d, o, nt); run<1, 38, 4>(d, o, nt); run<2, 38, 4>(d, o, nt); run<3, 38, 4>(d, o, nt);
    run<0, 38, 8>(d, o, nt); run<2, 38, 8>(d, o, nt); run<3, 38, 8>(d, o, nt);
    run<0, 16, 8>(d, o, nt); run<2, 16, 8>(d, o, nt);
    return 0;
}
