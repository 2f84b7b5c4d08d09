// Developer microbenchmark 5: LDS-DMA staging with 64-byte (half-line) vs 128-byte row pieces, nt policy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;
// W = bytes per row per tile (64 or 128 or 256); NBUF buffers of 64*W bytes
template <int W, int NBUF>
__global__ __launch_bounds__(64, 2) void k(const uint8_t *src, uint32_t *out, int ntiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t tiles[NBUF * 64 * W];
    const uint32_t lane = threadIdx.x, wg = blockIdx.x;
    constexpr int LPR = W / 16;          // lanes per row
    constexpr int RPI = 64 / LPR;        // rows per instruction
    constexpr int NI = 64 / RPI;         // instructions per tile
    const uint8_t *base = src + (size_t)wg * 64 * 8192;
    uint32_t acc = 0;
    auto issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const uint8_t *g = base + (size_t)(q * RPI + lane / LPR) * 8192 + (size_t)t * W + (lane % LPR) * 16;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(tiles + (t % NBUF) * 64 * W + q * 1024), 16, 0, 2);
        }
    };
    for (int t = 0; t < NBUF - 1; ++t) issue(t);
    for (int t = 0; t < ntiles; ++t) {
        if (NBUF == 2 || t + NBUF - 1 > ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NBUF == 3) { if (NI == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else if (NBUF == 4) { if (NI == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
        if (t + NBUF - 1 < ntiles) issue(t + NBUF - 1);
#pragma unroll
        for (int gt = 0; gt < W / 16; ++gt) {
            uint4 v;
            uint32_t addr = (uint32_t)(uintptr_t)(lds_ptr_t)tiles + (t % NBUF) * 64 * W + lane * W + gt * 16;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    out[wg * 64 + lane] = acc;
}
template <int W, int NBUF>
void run(const uint8_t *d, uint32_t *o)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<W, NBUF>), dim3(2048), dim3(64), 0, 0, d, o, 8192 / W);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("row piece %3d B, %d buffers (%2d KiB LDS/wave): %.4f ms  %.1f GB/s\n", W, NBUF, NBUF * 64 * W / 1024, best, (1ull << 30) / best / 1e6);
}
int main()
{
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, (1ull << 30) + (1 << 20)); hipMalloc(&o, 2048 * 64 * 4);
    hipMemset(d, 1, (1ull << 30) + (1 << 20));
    run<128, 2>(d, o); run<64, 2>(d, o); run<64, 3>(d, o); run<64, 4>(d, o); run<256, 2>(d, o);
    return 0;
}
