// Developer microbenchmark 2: does the output-store penalty depend on the load flavour (LDS-DMA vs VGPR loads),
// the load cache policy (nt) or the store flavour?  K1-shaped traffic: 2048 waves x 64 rows x 128 B tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

// LT: 0 = LDS-DMA, 1 = LDS-DMA nt, 2 = global_load_dwordx4 into VGPRs (no LDS), 3 = same, nt
// ST: 0 none, 1 global_store_dword x2 per tile, 2 = sc0 sc1, 3 = nt, 4 = one dwordx4 store per 2 tiles (nt)
template <int LT, int ST, int CH>
__global__ __launch_bounds__(64, 2) void k(const uint8_t *src, uint32_t *out, int ntiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t tiles[2 * 8192];
    const uint32_t lane = threadIdx.x, wg = blockIdx.x, rl = lane >> 3;
    const uint8_t *base = src + (size_t)wg * 64 * 8192;
    uint32_t acc = 0;
    uint4 r[2][8];
    auto issue = [&](int t, int q0 = 0, int q1 = 8) {
#pragma unroll
        for (int q = q0; q < q1; ++q) {
            const uint8_t *g = base + (size_t)(q * 8 + rl) * 8192 + t * 128 + (lane & 7) * 16;
            if (LT == 0) __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(tiles + (t & 1) * 8192 + q * 1024), 16, 0, 0);
            if (LT == 1) __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(tiles + (t & 1) * 8192 + q * 1024), 16, 0, 2);
            if (LT == 2) r[t & 1][q] = *reinterpret_cast<const uint4 *>(g);
            if (LT == 3) { typedef uint32_t v4 __attribute__((ext_vector_type(4))); v4 x = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(g)); r[t & 1][q] = make_uint4(x.x, x.y, x.z, x.w); }
        }
    };
    uint32_t *ob = out + (size_t)wg * 8192;
    issue(0);
#pragma unroll 2
    for (int t = 0; t < ntiles; ++t) {
        if (LT < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t + 1 < ntiles && CH == 0) issue(t + 1);
        if (t + 1 < ntiles && CH == 1) issue(t + 1, 0, 4);
        if (LT < 2) {
#pragma unroll
            for (int gt = 0; gt < 8; ++gt) {
                uint4 v;
                if (CH == 1 && gt == 3 && t + 1 < ntiles) issue(t + 1, 4, 8);
                if (CH == 2 && t + 1 < ntiles) issue(t + 1, gt, gt + 1);
                if (CH == 3 && t + 1 < ntiles && gt < 4) issue(t + 1, 2 * gt, 2 * gt + 2);
                uint32_t addr = (uint32_t)(uintptr_t)(lds_ptr_t)tiles + (t & 1) * 8192 + lane * 128 + ((gt * 16) ^ (((lane >> 1) & 7) * 16));
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc ^= r[t & 1][q].x ^ r[t & 1][q].y ^ r[t & 1][q].z ^ r[t & 1][q].w;
        }
        uint32_t *p0 = ob + (2 * t) * 64 + lane, *p1 = p0 + 64;
        if (ST == 1) { *p0 = acc; *p1 = ~acc; }
        if (ST == 2) { asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p0), "v"(acc) : "memory"); asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p1), "v"(~acc) : "memory"); }
        if (ST == 3) { __builtin_nontemporal_store(acc, p0); __builtin_nontemporal_store(~acc, p1); }
        if (ST == 4 && (t & 1)) { typedef uint32_t v4 __attribute__((ext_vector_type(4))); v4 x = {acc, ~acc, acc + 1, acc + 2}; __builtin_nontemporal_store(x, reinterpret_cast<v4 *>(ob) + (t >> 1) * 64 + lane); }
    }
    if (ST == 0) out[wg * 64 + lane] = acc;
}
template <int LT, int ST, int CH = 0>
void run(const uint8_t *d, uint32_t *o, int grid = 2048)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<LT, ST, CH>), dim3(grid), dim3(64), 0, 0, d, o, 64);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("load %d store %d chunk %d grid %d: %.4f ms  %.1f GB/s\n", LT, ST, CH, grid, best, (double)grid * 64 * 8192 / best / 1e6);
}
int main(int argc, char **argv)
{
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, (1ull << 30) + (1 << 20)); hipMalloc(&o, 2048 * 8192 * 4 + 4096);
    hipMemset(d, 1, (1ull << 30) + (1 << 20));
    if (argc > 1) { run<1, 0>(d, o); run<1, 1>(d, o); return 0; }
    run<1, 0, 0>(d, o); run<1, 1, 0>(d, o);
    run<1, 0, 1>(d, o); run<1, 1, 1>(d, o);
    run<1, 0, 2>(d, o); run<1, 1, 2>(d, o);
    run<1, 0, 3>(d, o); run<1, 1, 3>(d, o);
    run<1, 0, 0>(d, o, 1024); run<1, 1, 0>(d, o, 1024);
    run<1, 0, 0>(d, o, 1536); run<1, 1, 0>(d, o, 1536);
    return 0;
    run<0, 0>(d, o); run<0, 1>(d, o); run<0, 2>(d, o); run<0, 3>(d, o); run<0, 4>(d, o);
    run<1, 0>(d, o); run<1, 1>(d, o); run<1, 2>(d, o); run<1, 3>(d, o); run<1, 4>(d, o);
    run<2, 0>(d, o); run<2, 1>(d, o); run<2, 3>(d, o);
    run<3, 0>(d, o); run<3, 1>(d, o); run<3, 3>(d, o);
    return 0;
}
