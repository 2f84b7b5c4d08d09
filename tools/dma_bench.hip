// Developer microbenchmark (not product code): LDS-DMA staging patterns for K1's memory side.
// Each wave stages 64 "rows" x 128 B per tile into a double-buffered 16 KiB LDS area, one wave per
// workgroup, 2048 workgroups x 67 tiles ~ 1 GiB; prints achieved GB/s per pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

// mode 0: row stride 8192, all rows same column (K1 today)
// mode 1: column skew (t + row%8)            mode 2: linear stream (wave reads contiguous 8 KiB per tile)
// mode 3: column skew (t + row%16)           mode 4: column skew by wave id only ((t + wg) % ncols)
// mode 5: skew (t + row/8)  (rows of ONE load instruction share the column, instructions differ)
template <int NBUF>
__global__ __launch_bounds__(128, 2) void dma(const uint8_t *src, uint32_t *out, int mode, int ntiles, int ncols, int store, int check, int delay)
{
    __shared__ __attribute__((aligned(16))) uint8_t tiles[NBUF * 8192];
    const uint32_t lane = threadIdx.x & 63, wg = blockIdx.x, rl = lane >> 3;
    if (threadIdx.x >= 64) {   // store == 30/31: a second wave does nothing but the output stores, paced to the kernel's duration
        uint32_t *ob = out + (size_t)wg * 8192;
        for (int t = 0; t < ntiles; ++t) {
            ob[(2 * t) * 64 + lane] = t; ob[(2 * t + 1) * 64 + lane] = ~t;
            for (int k = 0; k < delay; ++k) __builtin_amdgcn_s_sleep(100);
        }
        return;
    }
    const uint8_t *base = src + (size_t)wg * 64 * 8192;
    uint32_t acc = 0;
    auto issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t row = q * 8 + rl;
            const uint8_t *g;
            if (mode == 2) g = base + (size_t)t * 8192 + q * 1024 + lane * 16;
            else {
                uint32_t col = t;
                if (mode == 1) col = (t + (row & 7)) % ncols;
                if (mode == 3) col = (t + (row & 15)) % ncols;
                if (mode == 4) col = (t + wg) % ncols;
                if (mode == 5) col = (t + q) % ncols;
                g = base + (size_t)row * 8192 + col * 128 + (lane & 7) * 16;
            }
            __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(tiles + (t % NBUF) * 8192 + q * 1024), 16, 0, 0);
        }
    };
    // K1 order: wait for tile t, issue t+1, consume t, then the two output stores of the tile.
    // store==3: the stores are younger than the DMA of t+1, so vmcnt(2) waits for the DMA only
    // (gfx9 vmcnt returns in order); bad counts LDS words that differ from the source pattern.
    uint32_t bad = 0;
    issue(0);
    for (int t = 0; t < ntiles; ++t) {
        if ((store == 3 || store == 20) && t > 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (store == 22 && t > 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // deliberately too lax: the checker must see it
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t + 1 < ntiles) issue(t + 1);
        if (store >= 20) {   // stores right behind the DMA issue: younger than tile t+1, a whole tile-time older than t+2
            uint32_t *ob2 = out + (size_t)wg * 8192;
            ob2[(2 * t) * 64 + lane] = acc; ob2[(2 * t + 1) * 64 + lane] = ~acc;
        }
#pragma unroll
        for (int gt = 0; gt < 8; ++gt) {
            uint4 v;
            uint32_t addr = (uint32_t)(uintptr_t)(lds_ptr_t)tiles + (t % NBUF) * 8192 + lane * 128 + ((gt * 16) ^ (((lane >> 1) & 7) * 16));
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
            if (check) {   // source word i holds i: lane reads row `lane`, tile t, 16-byte column gt (unswizzled here)
                const uint32_t col16 = gt ^ ((lane >> 1) & 7);
                const uint32_t w0 = (uint32_t)(((size_t)wg * 64 + lane) * 8192 + (size_t)t * 128 + col16 * 16) / 4;
                // LDS slot (lane*128 + c*16) was filled by loader lane (row=lane: q=lane>>3, rl=lane&7) column c
                bad += (v.x != w0) + (v.y != w0 + 1) + (v.z != w0 + 2) + (v.w != w0 + 3);
            }
        }
        uint32_t *ob = out + (size_t)wg * 8192;
        if (store == 1 || store == 3) { ob[(2 * t) * 64 + lane] = acc; ob[(2 * t + 1) * 64 + lane] = ~acc; }
        if (store == 4) ob[(2 * t) * 64 + lane] = acc;                                        // half bytes, half instrs
        if (store == 5) reinterpret_cast<uint2 *>(ob)[t * 64 + lane] = make_uint2(acc, ~acc);  // same bytes, half instrs
        if (store == 6 && (t & 1)) reinterpret_cast<uint4 *>(ob)[(t >> 1) * 64 + lane] = make_uint4(acc, ~acc, acc, ~acc);  // quarter instrs
        if (store == 7) { ob[lane] = acc; ob[64 + lane] = ~acc; }                             // same instrs, no new lines
        if (store == 10 && lane == 0) { ob[(2 * t) * 64] = acc; ob[(2 * t + 1) * 64] = ~acc; }
        if (store == 11 && t == ntiles - 1) { ob[(2 * t) * 64 + lane] = acc; ob[(2 * t + 1) * 64 + lane] = ~acc; }
        if (store == 12 && (t & 7) == 7) ob[(2 * t) * 64 + lane] = acc;
        if (store == 8 && (t & 7) == 7) {                                                     // 16 dword stores in a burst every 8 tiles
#pragma unroll
            for (int k = 0; k < 16; ++k) ob[(2 * (t - 7) + k) * 64 + lane] = acc + k;
        }
    }
    if (check) atomicAdd(&out[2048 * 8192], bad);
    out[wg * 64 + lane] = acc;
}

int main()
{
    const size_t bytes = 1ull << 30;
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, bytes + (1 << 20)); hipMalloc(&o, 2048 * 8192 * 4 + 4096);
    hipMemset(d, 1, bytes + (1 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ntiles = 64, ncols = 64;   // 64 tiles x 128 B = the whole 8 KiB row
    std::vector<uint32_t> pat((bytes + (1 << 20)) / 4);
    for (size_t i = 0; i < pat.size(); ++i) pat[i] = (uint32_t)i;
    hipMemcpy(d, pat.data(), pat.size() * 4, hipMemcpyHostToDevice);
    for (int check = 0; check <= 0; ++check)
    for (int delay = 0; delay <= 1; delay += 1)
    for (int store : {0, 1, 30})
    for (int nbuf = 2; nbuf <= 2; ++nbuf)
        for (int mode = 0; mode <= 0; mode += 2) {
            hipMemset(o + 2048 * 8192, 0, 4);
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (nbuf == 2) hipLaunchKernelGGL(dma<2>, dim3(2048), dim3(store >= 30 ? 128 : 64), 0, 0, d, o, mode, ntiles, ncols, store, check, delay);
                if (nbuf == 3) hipLaunchKernelGGL(dma<3>, dim3(2048), dim3(store >= 30 ? 128 : 64), 0, 0, d, o, mode, ntiles, ncols, store, check, delay);
                if (nbuf == 4) hipLaunchKernelGGL(dma<4>, dim3(2048), dim3(store >= 30 ? 128 : 64), 0, 0, d, o, mode, ntiles, ncols, store, check, delay);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            uint32_t bad = 0; hipMemcpy(&bad, o + 2048 * 8192, 4, hipMemcpyDeviceToHost);
            printf("check %d delay %d bad %u ", check, delay, bad);
            printf("store %d nbuf %d mode %d: %.4f ms  %.1f GB/s\n", store, nbuf, mode, best, bytes / best / 1e6);
        }
    return 0;
}
