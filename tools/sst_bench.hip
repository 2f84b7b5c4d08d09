// Developer microbenchmark 4: K1-shaped LDS-DMA read stream + output written with SCALAR stores
// (s_store_dwordx4 of wave-uniform data = ballot masks), verifying the data and timing the penalty.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

// ST: 0 none, 1 = 2 vector dword stores per tile, 5 = 32 x s_store_dwordx4 per tile (512 B per wave per tile)
template <int ST, int FILL>
__global__ __launch_bounds__(64, 2) void k(const uint8_t *src, uint32_t *out, int ntiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t tiles[2 * 8192];
    const uint32_t lane = threadIdx.x, wg = blockIdx.x, rl = lane >> 3;
    const uint8_t *base = src + (size_t)wg * 64 * 8192;
    uint32_t acc = 0;
    auto issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint8_t *g = base + (size_t)(q * 8 + rl) * 8192 + t * 128 + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(tiles + (t & 1) * 8192 + q * 1024), 16, 0, 2);
        }
    };
    uint32_t *ob = out + (size_t)wg * 8192;   // 32 KiB per wave
    issue(0);
    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t + 1 < ntiles) issue(t + 1);
#pragma unroll
        for (int gt = 0; gt < 8; ++gt) {
            uint4 v;
            uint32_t addr = (uint32_t)(uintptr_t)(lds_ptr_t)tiles + (t & 1) * 8192 + lane * 128 + ((gt * 16) ^ (((lane >> 1) & 7) * 16));
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
            // filler: FILL dependent VALU ops per group (K1 issues ~62 per group)
            float f = __uint_as_float(acc & 0x3fffffff);
#pragma unroll
            for (int i = 0; i < FILL; ++i) f = f * 1.0001f + 0.5f;
            acc ^= __float_as_uint(f) & 1;
            if (ST == 5) {
                // 8 sample steps = 4 x dwordx4; data: {wg, t*8+gt, k, 0xabcd0000 + k}
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                    v4 d = {wg, (uint32_t)(t * 8 + gt), (uint32_t)kk, 0xabcd0000u + (uint32_t)kk};
                    const uint32_t *p = ob + ((t * 8 + gt) * 4 + kk) * 4;
                    asm volatile("s_store_dwordx4 %0, %1, 0x0" :: "s"(d), "s"(p) : "memory");
                }
            }
        }
        if (ST == 1) { ob[(2 * t) * 64 + lane] = acc; ob[(2 * t + 1) * 64 + lane] = ~acc; }
        if (ST >= 10) {   // flush every N = ST-10+... tiles: 2 words per tile kept in registers, written as dwordx4 bursts
            constexpr int N = ST - 10;   // tiles per flush (8, 16, 32, 64)
            if ((t % N) == N - 1) {
                typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                v4 *o4 = reinterpret_cast<v4 *>(ob) + (size_t)(t / N) * (N / 2) * 64;
#pragma unroll
                for (int j = 0; j < N / 2; ++j) { v4 x = {acc + j, ~acc, acc ^ j, acc}; o4[j * 64 + lane] = x; }
            }
        }
        if (ST == 6) { const int tr = (t + wg * 5) & 63; ob[(2 * tr) * 64 + lane] = acc; ob[(2 * tr + 1) * 64 + lane] = ~acc; }
        if (ST == 7) { uint32_t *oc = out + ((size_t)t * 2048 + wg) * 128; oc[lane] = acc; oc[64 + lane] = ~acc; }   // time-major: all waves' tile-t output contiguous
        if (ST == 8) { const uint32_t h = (wg * 2654435761u) >> 21; uint32_t *oc = out + ((size_t)t * 2048 + h) * 128; oc[lane] = acc; oc[64 + lane] = ~acc; }
    }
    if (ST == 5) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    if (ST == 0) out[wg * 64 + lane] = acc;
    if (ST == 5 && acc == 0x12345678) out[0] = acc;
}
template <int ST, int FILL>
void run(const uint8_t *d, uint32_t *o)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    hipMemset(o, 0, 2048 * 8192 * 4);
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<ST, FILL>), dim3(2048), dim3(64), 0, 0, d, o, 64);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    long bad = -1;
    if (ST == 5) {
        std::vector<uint32_t> h(2048 * 8192);
        hipMemcpy(h.data(), o, h.size() * 4, hipMemcpyDeviceToHost);
        bad = 0;
        for (uint32_t wg = 0; wg < 2048; ++wg)
            for (uint32_t s = 0; s < 512; ++s)
                for (uint32_t kk = 0; kk < 4; ++kk) {
                    const uint32_t *p = &h[(size_t)wg * 8192 + (s * 4 + kk) * 4];
                    if (p[0] != wg || p[1] != s || p[2] != kk || p[3] != 0xabcd0000u + kk) ++bad;
                }
    }
    printf("fill %d store %d: %.4f ms  %.1f GB/s  bad %ld\n", FILL, ST, best, (1ull << 30) / best / 1e6, bad);
}
int main()
{
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, (1ull << 30) + (1 << 20)); hipMalloc(&o, 2048 * 8192 * 4 + 4096);
    hipMemset(d, 1, (1ull << 30) + (1 << 20));
    run<0, 0>(d, o); run<1, 0>(d, o); run<18, 0>(d, o); run<26, 0>(d, o); run<42, 0>(d, o); run<74, 0>(d, o);
    run<0, 40>(d, o); run<1, 40>(d, o); run<18, 40>(d, o); run<26, 40>(d, o); run<42, 40>(d, o); run<74, 40>(d, o);
    return 0;
}
