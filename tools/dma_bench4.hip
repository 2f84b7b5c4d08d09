// Developer microbenchmark 6: does a second tile of look-ahead help when per-tile compute time is comparable to
// the memory latency?  Same occupancy for both arms (6 waves per CU via a 24 KiB LDS footprint), nt LDS-DMA,
// FILL dependent VALU ops per 8-sample group as stand-in for K1's arithmetic, one dwordx4 store per 2 tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;
template <int NBUF, int FILL>
__global__ __launch_bounds__(64, 2) void k(const uint8_t *src, uint32_t *out, int ntiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t tiles[3 * 8192];   // always 24 KiB: equal occupancy
    const uint32_t lane = threadIdx.x, wg = blockIdx.x, rl = lane >> 3;
    const uint8_t *base = src + (size_t)wg * 64 * 8192;
    uint32_t acc = 0;
    auto issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint8_t *g = base + (size_t)(q * 8 + rl) * 8192 + t * 128 + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(tiles + (t % NBUF) * 8192 + q * 1024), 16, 0, 2);
        }
    };
    for (int t = 0; t < NBUF - 1; ++t) issue(t);
    for (int t = 0; t < ntiles; ++t) {
        if (NBUF == 2 || t + 2 > ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // the younger tile stays in flight (stores are older here)
        if (t + NBUF - 1 < ntiles) issue(t + NBUF - 1);
#pragma unroll
        for (int gt = 0; gt < 8; ++gt) {
            uint4 v;
            uint32_t addr = (uint32_t)(uintptr_t)(lds_ptr_t)tiles + (t % NBUF) * 8192 + lane * 128 + ((gt * 16) ^ (((lane >> 1) & 7) * 16));
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
            float f = __uint_as_float(acc & 0x3fffffff);
#pragma unroll
            for (int i = 0; i < FILL; ++i) f = f * 1.0001f + 0.5f;
            acc ^= __float_as_uint(f) & 1;
        }
    }
    out[wg * 64 + lane] = acc;
}
template <int NBUF, int FILL>
void run(const uint8_t *d, uint32_t *o)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NBUF, FILL>), dim3(2048), dim3(64), 0, 0, d, o, 64);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("buffers %d fill %3d: %.4f ms  %.1f GB/s\n", NBUF, FILL, best, (1ull << 30) / best / 1e6);
}
int main()
{
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, (1ull << 30) + (1 << 20)); hipMalloc(&o, 2048 * 64 * 4);
    hipMemset(d, 1, (1ull << 30) + (1 << 20));
    run<2, 0>(d, o); run<3, 0>(d, o);
    run<2, 40>(d, o); run<3, 40>(d, o);
    run<2, 60>(d, o); run<3, 60>(d, o);
    run<2, 80>(d, o); run<3, 80>(d, o);
    run<2, 100>(d, o); run<3, 100>(d, o);
    return 0;
}
