// K2 harness: k2_search_walk<SL, SET> alone on a random bitstream, with the per-wave phase stamps of AMR_K2W_DBG
// (developer tool, not product code).  Geometry: scm at chip length 72 by default (rows of 128 words), or "idm" / "all"
// (rows of 256 words).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAMR_K2W_DBG=1 -DAMR_K2R_LPR2=1 -Irtlamr_amd/csrc -Iinclude -o build/k2b tools/k2_bench.hip
// usage: k2b [scm|idm|all] [n_tiles incl. the history tile] [reps] [cold]
// "scm" also runs k2_search_row<144, 0, 128> (k2_row.h) on the same bitstream and compares counts and staging with the walk;
// cold: a 1 GiB buffer is streamed through the chip in front of every launch (what K1 does to the caches in the product)
// Note: the bitstream stays in the 256 MB Infinity Cache between launches here; in the product K1 has just streamed a GiB
// through it and K2 reads a cold bitstream (scm: 20-25 us here, 34 us in the bench's --depth 1 profile).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "k2_row.h"

#ifndef K2B_PF1
#define K2B_PF1 AMR_K2W_PF1
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint64_t bits_of(const char *s) { uint64_t v = 0; for (int p = 0; s[p]; ++p) v |= (uint64_t)(s[p] == '1') << p; return v; }

__global__ void k_stream_read(const uint4 *p, size_t n16, uint32_t *sink)
{
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p) + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345679u) *sink = acc;
}
// "coldw": what K1 does in front of K2 in the product -- one wave per tile, XCD-contiguous tile order, streams its share of
// a GiB (nt loads) and WRITES its tile of the bitstream (sc1 stores, from a shadow copy of the same bits)
__global__ __launch_bounds__(64) void k_k1_like(const uint4 *cold, size_t n16_per_wave, const uint4 *shadow, uint4 *qt, uint32_t tile16, uint32_t *sink)
{
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const uint32_t wt = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const v4 *src = reinterpret_cast<const v4 *>(cold) + (size_t)wt * n16_per_wave;
    uint32_t acc = 0;
    for (size_t i = threadIdx.x; i < n16_per_wave; i += 64) { const v4 v = __builtin_nontemporal_load(src + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    const uint4 *sh = shadow + (size_t)(wt + 1) * tile16;      // tile 0 = history tile: not written by K1
    uint4 *dst = qt + (size_t)(wt + 1) * tile16;
    for (uint32_t i = threadIdx.x; i < tile16; i += 64) {
        const v4 x = reinterpret_cast<const v4 *>(sh)[i];
        v4 *pd = reinterpret_cast<v4 *>(dst) + i;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(pd), "v"(x) : "memory");
    }
    if (acc == 0x12345679u) *sink = acc;
}
static uint4 *g_cold = nullptr; static uint32_t *g_sink = nullptr; static const size_t kColdBytes = (size_t)1 << 30;
static const uint4 *g_shadow = nullptr; static uint4 *g_qt = nullptr; static uint32_t g_tile16 = 0, g_wtiles = 0;
static void cold_pass()
{
    if (g_shadow) hipLaunchKernelGGL(k_k1_like, dim3(g_wtiles), dim3(64), 0, 0, g_cold, kColdBytes / 16 / g_wtiles, g_shadow, g_qt, g_tile16, g_sink);
    else if (g_cold) hipLaunchKernelGGL(k_stream_read, dim3(4096), dim3(256), 0, 0, g_cold, kColdBytes / 16, g_sink);
}

// the row kernel (one preamble, rows of up to 128 words) on the same arguments; compares with what the walk left behind
template <int SL, int KIND, int WPB, int LPR = 1>
static void run_row(const char *name, amr::K2Args a, uint32_t n_tiles, int reps, unsigned long long *d_dbg)
{
    using namespace amr;
    std::vector<uint32_t> cnt0((size_t)n_tiles), st0((size_t)n_tiles * a.cap);
    CK(hipMemcpy(cnt0.data(), a.counts, cnt0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(st0.data(), a.staging, st0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(a.counts, 0xff, cnt0.size() * 4)); CK(hipMemset(a.staging, 0xff, st0.size() * 4));
    const uint32_t n_wg = (n_tiles * LPR + kK2WWaves - 1) / kK2WWaves;
    const uint32_t grid = 8u * ((n_wg + 7u) / 8u);
    const size_t lds = k2_walk_lds_bytes(0);
    CK(hipFuncSetAttribute((const void *)k2_search_row<SL, KIND, WPB, LPR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    const uint32_t groups = k2_groups(n_tiles);
    for (int r = 0; r < reps + 30; ++r) {
        CK(hipMemsetAsync(a.gcnt, 0, (size_t)groups * a.g.n_pre * 4 * kGroupStride, 0));
        CK(hipMemsetAsync(a.overflow, 0, 4, 0));
        cold_pass();
        a.dbg = (r == reps + 29) ? d_dbg : nullptr;
        hipExtLaunchKernelGGL((k2_search_row<SL, KIND, WPB, LPR>), dim3(grid), dim3(64 * kK2WWaves), lds, 0, e0, e1, 0, a);
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 30) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    uint32_t ovf = 0; CK(hipMemcpy(&ovf, a.overflow, 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> cnt((size_t)n_tiles), st((size_t)n_tiles * a.cap);
    CK(hipMemcpy(cnt.data(), a.counts, cnt.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(st.data(), a.staging, st.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long hits = 0, bad_c = 0, bad_s = 0;
    for (uint32_t T = 0; T < n_tiles; ++T) {
        hits += cnt[T];
        if (cnt[T] != cnt0[T]) { if (bad_c++ < 5) printf("    tile %u: count %u, walk %u\n", T, cnt[T], cnt0[T]); continue; }
        for (uint32_t i = 0; i < cnt[T]; ++i)
            if (st[(size_t)T * a.cap + i] != st0[(size_t)T * a.cap + i]) { if (bad_s++ < 5) printf("    tile %u hit %u: %u, walk %u\n", T, i, st[(size_t)T * a.cap + i], st0[(size_t)T * a.cap + i]); }
    }
    printf("k2r %-4s SL=%d KIND=%d WPB=%d tiles=%u grid=%u: min %.1f med %.1f p90 %.1f us   hits %llu overflow %u   vs walk: %llu count mismatches, %llu position mismatches %s\n",
           name, SL, KIND, WPB, n_tiles, grid, ms[0] * 1e3, ms[ms.size() / 2] * 1e3, ms[ms.size() * 9 / 10] * 1e3, hits, ovf, bad_c, bad_s,
           (bad_c || bad_s) ? "** MISMATCH **" : "(identical)");
    if (!AMR_K2W_DBG) return;
    std::vector<unsigned long long> d((size_t)n_tiles * 16);
    CK(hipMemcpy(d.data(), d_dbg, d.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> walk, life, start, end;
    unsigned long long t0 = ~0ull;
    for (uint32_t T = 0; T < n_tiles; ++T) t0 = std::min(t0, d[(size_t)T * 16 + 8]);
    for (uint32_t T = 0; T < n_tiles; ++T) {
        const unsigned long long *w = &d[(size_t)T * 16];
        walk.push_back((double)(w[1] - w[0]));
        life.push_back((w[9] - w[8]) * 0.01); start.push_back((w[8] - t0) * 0.01); end.push_back((w[9] - t0) * 0.01);
    }
    auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
    printf("    cycles  walk %.0f / %.0f / %.0f (p10 / med / p90)\n", q(walk, .1), q(walk, .5), q(walk, .9));
    printf("    us      wave life %.1f / %.1f / %.1f   start %.1f / %.1f / %.1f   end %.1f / %.1f / %.1f (max %.1f)\n",
           q(life, .1), q(life, .5), q(life, .9), q(start, .1), q(start, .5), q(start, .9), q(end, .1), q(end, .5), q(end, .9), q(end, 1.0));
}

template <int SL, int SET>
static void run(const char *name, amr::K2Args a, uint32_t n_tiles, int reps, unsigned long long *d_dbg)
{
    using namespace amr;
    const uint32_t n_wg = (n_tiles + kK2WWaves - 1) / kK2WWaves;
    const uint32_t grid = 8u * ((n_wg + 7u) / 8u);
    const size_t lds = k2_walk_lds_bytes(0);
    CK(hipFuncSetAttribute((const void *)k2_search_walk<SL, SET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    const uint32_t groups = k2_groups(n_tiles);
    for (int r = 0; r < reps + 30; ++r) {
        CK(hipMemsetAsync(a.gcnt, 0, (size_t)groups * a.g.n_pre * 4 * kGroupStride, 0));
        CK(hipMemsetAsync(a.overflow, 0, 4, 0));
        cold_pass();
        a.dbg = (r == reps + 29) ? d_dbg : nullptr;
        hipExtLaunchKernelGGL((k2_search_walk<SL, SET>), dim3(grid), dim3(64 * kK2WWaves), lds, 0, e0, e1, 0, a);
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 30) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    uint32_t ovf = 0; CK(hipMemcpy(&ovf, a.overflow, 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> cnt((size_t)n_tiles * a.g.n_pre);
    CK(hipMemcpy(cnt.data(), a.counts, cnt.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long hits = 0; for (uint32_t c : cnt) hits += c;
    printf("k2b %-4s SL=%d SET=%d tiles=%u grid=%u: min %.1f med %.1f p90 %.1f us   hits %llu overflow %u\n", name, SL, SET, n_tiles, grid,
           ms[0] * 1e3, ms[ms.size() / 2] * 1e3, ms[ms.size() * 9 / 10] * 1e3, hits, ovf);
    if (!AMR_K2W_DBG) return;
    // phase stamps of the last launch: shader-clock deltas per wave, real-time start/end (100 MHz)
    std::vector<unsigned long long> d((size_t)n_tiles * 16);
    CK(hipMemcpy(d.data(), d_dbg, d.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> walk, st2, emit, life, start, end;
    unsigned long long t0 = ~0ull;
    for (uint32_t T = 0; T < n_tiles; ++T) t0 = std::min(t0, d[(size_t)T * 16 + 8]);
    unsigned long long cand = 0, keep = 0;
    for (uint32_t T = 0; T < n_tiles; ++T) {
        const unsigned long long *w = &d[(size_t)T * 16];
        walk.push_back((double)(w[1] - w[0])); st2.push_back((double)(w[2] - w[1])); emit.push_back((double)(w[3] - w[2]));
        life.push_back((w[9] - w[8]) * 0.01); start.push_back((w[8] - t0) * 0.01); end.push_back((w[9] - t0) * 0.01);
        cand += w[7] >> 32; keep += w[7] & 0xffffffffu;
    }
    auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
    printf("    cycles  walk %.0f / %.0f / %.0f   stage2 %.0f / %.0f / %.0f   ranks+emit %.0f / %.0f / %.0f   (p10 / med / p90)\n",
           q(walk, .1), q(walk, .5), q(walk, .9), q(st2, .1), q(st2, .5), q(st2, .9), q(emit, .1), q(emit, .5), q(emit, .9));
    printf("    us      wave life %.1f / %.1f / %.1f   start %.1f / %.1f / %.1f   end %.1f / %.1f / %.1f (max %.1f)   candidates %llu kept %llu\n",
           q(life, .1), q(life, .5), q(life, .9), q(start, .1), q(start, .5), q(start, .9), q(end, .1), q(end, .5), q(end, .9), q(end, 1.0), cand, keep);
    // per-XCD picture: tiles are dealt in 8 contiguous runs
    const uint32_t per = (n_tiles + 7) / 8;
    printf("    end of the last wave per XCD run (us):");
    for (int x = 0; x < 8; ++x) { double m = 0; for (uint32_t T = x * per; T < std::min(n_tiles, (x + 1) * per); ++T) m = std::max(m, end[T]); printf(" %.1f", m); }
    printf("\n");
}

int main(int argc, char **argv)
{
    using namespace amr;
    const char *kind = argc > 1 ? argv[1] : "scm";
    const bool wide = strcmp(kind, "scm") != 0;
    const uint32_t n_tiles = argc > 2 ? (uint32_t)atoi(argv[2]) : (wide ? 4097u : 2049u);
    const int reps = argc > 3 ? atoi(argv[3]) : 40;
    const bool coldw = argc > 4 && !strcmp(argv[4], "coldw");
    if (argc > 4 && (!strcmp(argv[4], "cold") || coldw)) {
        CK(hipMalloc((void **)&g_cold, kColdBytes)); CK(hipMalloc((void **)&g_sink, 4));
        CK(hipMemset(g_cold, 1, kColdBytes));
    }
    K2Args a{};
    SearchGeom &g = a.g;
    g.block_size = wide ? 8192 : 4096; g.lg_block_size = wide ? 13 : 12; g.wpb = g.block_size / 32; g.lg_wpb = wide ? 8 : 7;
    g.symbol_length = 144;
    const char *pre[4] = {"111110010101001100000", "0001011010100011", "01010101010101010001011010100011", "00000000000000001110010101100100"};
    if (!strcmp(kind, "scm")) { g.n_pre = 1; g.pre_len[0] = 21; g.pre_bits[0] = bits_of(pre[0]); g.packet_symbols = 96; a.walk_pids = 0; }
    else if (!strcmp(kind, "idm")) { g.n_pre = 1; g.pre_len[0] = 32; g.pre_bits[0] = bits_of(pre[2]); g.packet_symbols = 736; a.walk_pids = 0; }
    else { g.n_pre = 4; for (int q = 0; q < 4; ++q) { g.pre_len[q] = (uint32_t)strlen(pre[q]); g.pre_bits[q] = bits_of(pre[q]); }
           g.packet_symbols = 736; a.walk_pids = 0u | (1u << 8) | (2u << 16) | (3u << 24); }
    g.max_pre_len = 0; for (uint32_t q = 0; q < g.n_pre; ++q) g.max_pre_len = std::max(g.max_pre_len, g.pre_len[q]);
    g.packet_length = g.packet_symbols * g.symbol_length; g.pkt_bytes = (g.packet_symbols + 7) / 8;
    const size_t tile_words = (size_t)64 * g.wpb;
    const size_t qt_bytes = (size_t)(n_tiles + 2) * tile_words * 4 + kQtSlackBytes;   // history tile + n_tiles + one more, as ensure_qt
    uint32_t *d_qt; CK(hipMalloc((void **)&d_qt, qt_bytes));
    {   // random bits (what noise quantizes to)
        std::vector<uint32_t> h(qt_bytes / 4);
        uint64_t s = 0x9e3779b97f4a7c15ull;
        for (auto &w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
        if (getenv("K2B_BURSTS")) {   // what planted packets look like: K2B_BURSTS per tile, each 72 adjacent positions that match the preamble
            const int per_tile = atoi(getenv("K2B_BURSTS"));
            const uint32_t lg_wpb = g.lg_wpb, wpb = g.wpb;
            auto setbit = [&](uint64_t n, int v) {      // stream bit n counted from row 64 (first batch row), first sample in bit 31
                const uint64_t R = 64 + n / g.block_size; const uint32_t w = (uint32_t)((n % g.block_size) >> 5), b = 31 - (uint32_t)(n & 31);
                const size_t i = ((R >> 6) << (6 + lg_wpb)) + ((size_t)(w >> 2) << 8) + ((R & 63) << 2) + (w & 3);
                if (i < h.size()) h[i] = (h[i] & ~(1u << b)) | ((uint32_t)v << b);
            };
            const uint64_t tile_bits = (uint64_t)64 * g.block_size;
            (void)wpb;
            for (uint32_t T = 0; T + 2 < n_tiles; ++T)
                for (int k = 0; k < per_tile; ++k) {
                    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                    const uint64_t o = (uint64_t)T * tile_bits + (s >> 20) % (tile_bits - 40000);
                    for (uint32_t p = 0; p < g.pre_len[0]; ++p)
                        for (int d = 0; d < 72; ++d) setbit(o + (uint64_t)p * g.symbol_length + d, (int)((g.pre_bits[0] >> p) & 1));
                }
        }
        CK(hipMemcpy(d_qt, h.data(), qt_bytes, hipMemcpyHostToDevice));
    }
    if (coldw) {
        uint4 *sh; CK(hipMalloc((void **)&sh, qt_bytes)); CK(hipMemcpy(sh, d_qt, qt_bytes, hipMemcpyDeviceToDevice));
        g_shadow = sh; g_qt = reinterpret_cast<uint4 *>(d_qt); g_tile16 = (uint32_t)(tile_words / 4); g_wtiles = (n_tiles - 1) & ~7u;
    }
    a.qt = d_qt; a.n_tiles = n_tiles; a.cap = 1024;
    CK(hipMalloc((void **)&a.counts, (size_t)n_tiles * g.n_pre * 4));
    CK(hipMalloc((void **)&a.gcnt, (size_t)k2_groups(n_tiles) * g.n_pre * 4 * kGroupStride));
    CK(hipMalloc((void **)&a.staging, (size_t)n_tiles * g.n_pre * a.cap * 4));
    CK(hipMalloc((void **)&a.overflow, 4));
    unsigned long long *d_dbg; CK(hipMalloc((void **)&d_dbg, (size_t)n_tiles * 16 * 8));
    a.n_lo = -(int64_t)g.packet_length; a.n_hi = (int64_t)(n_tiles - 1) * 64 * g.block_size - g.packet_length;   // tile 0 = history tile
    if (!strcmp(kind, "scm")) { run<144, 1>("scm", a, n_tiles, reps, d_dbg); run_row<144, 0, 128>("scm", a, n_tiles, reps, d_dbg); }
    else if (!strcmp(kind, "idm")) { run<144, 4>("idm", a, n_tiles, reps, d_dbg); run_row<144, 2, 128, 2>("idm", a, n_tiles, reps, d_dbg); }
    else run<144, 15>("all", a, n_tiles, reps, d_dbg);
    return 0;
}
