// Developer harness (not product code): times K1 alone, without Python/torch, so that many kernel variants can be
// compared in one short GPU session.  Build one binary per variant, e.g.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Irtlamr_amd/csrc -Iinclude \
//         -DK1B_CL=72 [-DAMR_K1T_SCHED=1 ...] -o build/k1b_<tag> tools/k1_bench.hip
// usage: k1b [variants: comma list of names, or "all"] [n_blocks] [reps] [input: 0 noise, 1 uniform bytes] [n_allocs] [block_size] [spin-up launches]
// All variants live in one binary and run round-robin on the same buffers (K1's time depends on the process's
// allocation and on the shader clock's ramp-up, so only same-process, interleaved comparisons mean anything).
// Prints per-variant launch times (events on the dispatch itself) and a checksum of the tiled bitstream: every
// variant must print the same checksum for the same input (bit-exactness check before the real parity tests).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "k1_tile.h"
#include "synth.h"

#ifndef K1B_CL
#define K1B_CL 72
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_uniform(uint8_t *iq, uint64_t n16, uint64_t seed)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n16) return;
    const uint64_t a = amr::splitmix64(seed ^ (2 * t)), b = amr::splitmix64(seed ^ (2 * t + 1));
    reinterpret_cast<uint4 *>(iq)[t] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}

__global__ void k_checksum(const uint32_t *q, uint64_t n, unsigned long long *out)
{
    unsigned long long s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        s += (unsigned long long)q[i] * (2654435761ull * (i + 1) | 1ull);
    atomicAdd(out, s);
}


struct Variant { const char *name; void (*launch)(const amr::K1Args &, uint32_t full, uint32_t rem, hipEvent_t, hipEvent_t); };

template <class C>
void launch_tile(const amr::K1Args &a0, uint32_t full, uint32_t rem, hipEvent_t e0, hipEvent_t e1)
{
    amr::K1Args a = a0;
    constexpr int CL = K1B_CL;
    if (full) { a.wg_first = 0; hipExtLaunchKernelGGL((amr::k1t_demod<CL, false, C>), dim3(full), dim3(64), C::kLds, 0, e0, rem ? nullptr : e1, 0, a); }
    if (rem) { a.wg_first = full; hipExtLaunchKernelGGL((amr::k1t_demod<CL, true, C>), dim3(1), dim3(64), C::kLds, 0, full ? nullptr : e0, e1, 0, a); }
}
// name = s<SCHED>p<DEPTH>x<XCD>n<NW>
#ifndef K1B_VARIANTS
#define K1B_VARIANTS V(1,1,1,16) V(1,1,0,16) V(0,1,1,16) V(1,2,1,16) V(1,1,1,32) V(1,1,1,8)
#endif
static const Variant kVariants[] = {
#define V(S, P, X, N) {"s" #S "p" #P "x" #X "n" #N, launch_tile<amr::K1TCfg<S, P, X, N>>},
#define D(S, P, X, N, DG) {"s" #S "p" #P "x" #X "n" #N "d" #DG, launch_tile<amr::K1TCfg<S, P, X, N, DG>>},
#define W(S, P, X, N, POL, AFT) {"s" #S "p" #P "x" #X "n" #N "w" #POL "a" #AFT, launch_tile<amr::K1TCfg<S, P, X, N, 0, POL, AFT>>},
#define L(S, P, X, N, POL, AFT, NLC) {"s" #S "p" #P "x" #X "n" #N "w" #POL "a" #AFT "l" #NLC, launch_tile<amr::K1TCfg<S, P, X, N, 0, POL, AFT, NLC>>},
#define P(S, P_, X, N, POL, AFT, NLC, PR) {"s" #S "p" #P_ "x" #X "n" #N "w" #POL "a" #AFT "l" #NLC "r" #PR, launch_tile<amr::K1TCfg<S, P_, X, N, 0, POL, AFT, NLC, PR>>},
#define H(S, P_, X, N, POL, AFT, NLC, PR, HD) {"s" #S "p" #P_ "x" #X "n" #N "w" #POL "a" #AFT "l" #NLC "r" #PR "h" #HD, launch_tile<amr::K1TCfg<S, P_, X, N, 0, POL, AFT, NLC, PR, HD>>},
#define T(NAME, ...) {#NAME, launch_tile<amr::K1TCfg<__VA_ARGS__>>},     /* any configuration, named freely */
    K1B_VARIANTS
#undef T
#undef H
#undef P
#undef V
#undef D
#undef W
#undef L
};

int main(int argc, char **argv)
{
    const char *sel = argc > 1 ? argv[1] : "all";
    const uint32_t n_blocks = argc > 2 ? (uint32_t)atoll(argv[2]) : 131072;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    const int input = argc > 4 ? atoi(argv[4]) : 0;
    const int n_allocs = argc > 5 ? atoi(argv[5]) : 1;
    const uint32_t bs = argc > 6 ? (uint32_t)atoi(argv[6]) : 4096;
    const int spin = argc > 7 ? atoi(argv[7]) : 300;   // untimed launches first: the shader clock takes tens of ms to ramp up
    constexpr int CL = K1B_CL;
    const uint32_t hba = (4 * CL + 127) & ~127;
    const size_t iq_bytes = (size_t)n_blocks * bs * 2;
    const uint32_t wpb = bs / 32;
    const uint32_t full = n_blocks / 64, rem = n_blocks % 64;
    const size_t qt_words = (size_t)(full + (rem ? 1 : 0) + 1) * 64 * wpb;

    std::vector<Variant> vs;
    for (const Variant &v : kVariants) {
        std::string s = std::string(",") + sel + ",";
        if (std::string(sel) == "all" || s.find(std::string(",") + v.name + ",") != std::string::npos) vs.push_back(v);
    }
    if (vs.empty()) { fprintf(stderr, "no variant selected\n"); return 1; }

    float lut[256];
    for (int i = 0; i < 256; ++i) {   // decode.go:209-216, float32 arithmetic
        volatile float x = (127.5f - (float)i) / 127.5f;
        volatile float y = x * x;
        lut[i] = y;
    }
    float *d_lut; uint8_t *d_carry; uint32_t *d_qt; unsigned long long *d_sum;
    CK(hipMalloc(&d_lut, 1024)); CK(hipMemcpy(d_lut, lut, 1024, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_carry, hba)); CK(hipMemset(d_carry, 127, hba));
    const bool qt_late = argc > 8 && atoi(argv[8]) == 4;
    if (!qt_late) CK(hipMalloc(&d_qt, qt_words * 4));
    CK(hipMalloc(&d_sum, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    // how the IQ buffers are obtained (K1's time depends on it): 0 = hipMalloc each; 1 = a 1 GiB dummy hipMalloc first;
    // 2 = slices of ONE hipMalloc; 3 = virtual-memory API (hipMemCreate/hipMemMap), VA aligned to 1 GiB; 4 = like 0 but
    // the bitstream buffer is allocated after the IQ buffers
    const int amode = argc > 8 ? atoi(argv[8]) : 0;
    std::vector<uint8_t *> iqs(n_allocs);
    uint8_t *big = nullptr, *dummy = nullptr;
    if (amode == 1) CK(hipMalloc(&dummy, (size_t)1 << 30));
    if (amode == 2) CK(hipMalloc(&big, iq_bytes * n_allocs));
    for (int ai = 0; ai < n_allocs; ++ai) {
        if (amode == 2) iqs[ai] = big + iq_bytes * ai;
        else if (amode == 3) {
            hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
            const size_t sz = (iq_bytes + gran - 1) / gran * gran;
            hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, sz, &prop, 0));
            void *va = nullptr; CK(hipMemAddressReserve(&va, sz, (size_t)1 << 30, nullptr, 0));
            CK(hipMemMap(va, sz, 0, h, 0));
            hipMemAccessDesc ad{}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, sz, &ad, 1));
            iqs[ai] = (uint8_t *)va;
            if (ai == 0) printf("VMM granularity %zu\n", gran);
        } else CK(hipMalloc(&iqs[ai], iq_bytes));
        const uint64_t n_samples = (uint64_t)n_blocks * bs;
        if (input == 0)
            hipLaunchKernelGGL(amr::k_synth_noise<false>, dim3((unsigned)((n_samples / 8 + 255) / 256)), dim3(256), 0, 0, iqs[ai], n_samples, 1ull, 0ull);
        else
            hipLaunchKernelGGL(k_uniform, dim3((unsigned)((iq_bytes / 16 + 255) / 256)), dim3(256), 0, 0, iqs[ai], iq_bytes / 16, 7ull);
        CK(hipDeviceSynchronize());
    }
    if (qt_late) CK(hipMalloc(&d_qt, qt_words * 4));
    amr::K1Args a{};
    a.carry = d_carry; a.lut = d_lut; a.qt = d_qt; a.n_blocks = n_blocks; a.block_size = bs; a.zero_halo = 0;
    a.iq = iqs[0];
    for (int r = 0; r < spin; ++r) vs[r % vs.size()].launch(a, full, rem, e0, e1);
    CK(hipDeviceSynchronize());

    const double gb = 2.0 * n_blocks * bs;
    // K1B_ROTATE=1: every launch reads another of the n_allocs inputs (nothing of an input survives in a cache from one launch
    // of it to the next); default: all repetitions on one input, then the next
    const bool rotate = getenv("K1B_ROTATE") && n_allocs > 1;
    for (int ai = 0; ai < (rotate ? 1 : n_allocs); ++ai) {
        a.iq = iqs[ai];
        std::vector<std::vector<float>> ms(vs.size());
        for (int r = 0; r < reps; ++r)
            for (size_t vi = 0; vi < vs.size(); ++vi) {
                if (rotate) a.iq = iqs[(r * vs.size() + vi) % n_allocs];
                vs[vi].launch(a, full, rem, e0, e1);
                CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                ms[vi].push_back(t);
            }
        CK(hipGetLastError());
        a.iq = iqs[ai];
        for (size_t vi = 0; vi < vs.size(); ++vi) {
            CK(hipMemset(d_qt, 0, qt_words * 4));
            vs[vi].launch(a, full, rem, e0, e1);
            CK(hipMemset(d_sum, 0, 8));
            const size_t first = (size_t)64 * wpb;   // skip the history tile; whole tiles only (a partial last tile interleaves unused rows)
            hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, 0, d_qt + first, (size_t)full * 64 * wpb, d_sum);
            unsigned long long sum; CK(hipMemcpy(&sum, d_sum, 8, hipMemcpyDeviceToHost));
            std::vector<float> &m = ms[vi];
            std::sort(m.begin(), m.end());
            const double med = m[m.size() / 2];
            printf("k1b %-12s CL=%d bs=%u blocks=%u in=%d alloc=%d(%p): min %.4f med %.4f p90 %.4f ms  %.3f TB/s frac %.3f  checksum %016llx",
                   vs[vi].name, CL, bs, n_blocks, input, ai, (void *)iqs[ai], m.front(), med, m[m.size() * 9 / 10], gb / med * 1e-9, gb / med * 1e-9 / 8.0, sum);
#if AMR_K1T_CLK
            { uint32_t c[4]; CK(hipMemcpy(c, d_qt, 16, hipMemcpyDeviceToHost));
              const double cyc = (double)(((uint64_t)c[1] << 32) | c[0]), rt = (double)(((uint64_t)c[3] << 32) | c[2]);
              if (rt > 0) printf("  clk %.2f GHz", cyc / (rt * 10.0));
              static std::vector<unsigned long long> tl(3 * amr::kK1TWgs);      // slot 0 of the timeline (tl_seq stays 0 here)
              CK(hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(amr::k1t_timeline), tl.size() * 8));
              const unsigned nwg = std::min<unsigned>(amr::kK1TWgs, n_blocks / 64);
              unsigned long long t0 = ~0ull, t1 = 0;
              for (unsigned i = 0; i < nwg; ++i) { t0 = std::min(t0, tl[3 * i]); t1 = std::max(t1, tl[3 * i + 1]); }
              std::vector<double> st, du, en;
              for (unsigned i = 0; i < nwg; ++i) { st.push_back((tl[3 * i] - t0) * 0.01); du.push_back((tl[3 * i + 1] - tl[3 * i]) * 0.01); en.push_back((t1 - tl[3 * i + 1]) * 0.01); }
              std::sort(st.begin(), st.end()); std::sort(du.begin(), du.end()); std::sort(en.begin(), en.end());
              auto pc = [](const std::vector<double> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
              printf("\n     timeline us (last launch): span %.1f | start p50 %.1f p90 %.1f p99 %.1f max %.1f | duration p1 %.1f p50 %.1f p99 %.1f | end before last: p50 %.1f p90 %.1f max %.1f",
                     (t1 - t0) * 0.01, pc(st, .5), pc(st, .9), pc(st, .99), pc(st, 1), pc(du, .01), pc(du, .5), pc(du, .99), pc(en, .5), pc(en, .9), pc(en, 1));
              printf("\n     mean duration per XCD (b %% 8):");
              for (unsigned x = 0; x < 8; ++x) { double m = 0, mx = 0; unsigned n = 0; for (unsigned i = x; i < nwg; i += 8) { const double d = (tl[3 * i + 1] - tl[3 * i]) * 0.01; m += d; mx = std::max(mx, d); ++n; } printf(" %.1f(max %.1f)", m / n, mx); }
              printf("\n     mean duration by dispatch order (b >> 3, 8 groups of 32):");
              for (unsigned gq = 0; gq < 8; ++gq) { double m = 0; unsigned n = 0; for (unsigned i = 0; i < nwg; ++i) if (((i >> 3) * 8 / (nwg / 8)) == gq) { m += (tl[3 * i + 1] - tl[3 * i]) * 0.01; ++n; } printf(" %.1f", n ? m / n : 0.0); }
            }
#endif
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
