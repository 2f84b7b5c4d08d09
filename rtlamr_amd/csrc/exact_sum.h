// The exact parallel form of the reference's sequential float32 running sum: what k1_single.h (one block per call, a whole
// workgroup) and k1_coop.h (one wave per block) share.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amr {

// ---------------------------------------------------------------------------------------------------------------------
// The running sum of decode.go:232-236, csum[j+1] = fl32(csum[j] + Signal[j]), is sequential by definition: every
// addition rounds, and the rounding depends on the sum so far.  One lane doing the reference's additions one after the
// other costs 15 shader cycles per sample once operands and results travel through LDS (45 us for 4240 samples at the
// clock a nearly idle chip runs at; the DPP chain across a wave, k1_coop.h: 16.4).  But the terms are non-negative, so
// the sum only grows, and WHILE IT STAYS INSIDE ONE BINADE [2^e, 2^(e+1)) every partial sum is a multiple of
// ulp = 2^(e-23), the spacing of float32 there.  With c = n * ulp and m = x * ulp (x exact: a power-of-two scaling),
//     fl32(c + m) = (n + RN(x)) * ulp                  -- round-to-nearest of the term alone, whatever n is --
// unless x lies exactly half way between two integers, where round-half-to-even looks at n:
//     fl32(c + m) = (n + a + ((n + a) & 1)) * ulp,     a = floor(x).
// So inside a binade the sequential float32 sum is an INTEGER recurrence n -> n + d(n & 1): a two-state (parity)
// transducer per term, and transducers compose associatively -- a parallel scan.  The block scans all remaining terms at
// once, finds the first term at which the sum leaves the binade (n reaches 2^24), takes every sum in front of it from
// the scan -- bit for bit the reference's values --, performs THAT one addition in float32 and starts over in the new
// binade.  A block of receiver noise crosses 5 to 6 binades after the first few hundred samples (which one lane adds
// up sequentially: the sum doubles every few samples there); the worst case seen in tests is 15.  oracle twin:
// tests/test_exact_sum_cpu.py (numpy, the same decisions) against the plain sequential loop.
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kSumSeq = 256;        // leading terms added sequentially by one lane (>= the longest zero history, 192)
constexpr uint32_t kSumCap = 1u << 26;   // saturation of the integer deltas (anything >= 2^24 - n0 is "left the binade")
constexpr uint32_t kSumMaxPhases = 48;   // beyond this one lane finishes sequentially (never seen; correctness either way)

struct KsPair { uint32_t d0, d1; };      // the sum's increase over a run of terms, for even / odd n in front of the run

__device__ __forceinline__ uint32_t ks_sat(uint32_t x) { return x < kSumCap ? x : kSumCap; }
// first f, then g
__device__ __forceinline__ KsPair ks_compose(KsPair f, KsPair g)
{
    KsPair h;
    h.d0 = ks_sat(f.d0 + ((f.d0 & 1u) ? g.d1 : g.d0));
    h.d1 = ks_sat(f.d1 + ((f.d1 & 1u) ? g.d0 : g.d1));     // n odd, plus an odd delta: even
    return h;
}
// one term: a = RN(x) (or floor(x) at a tie), tie = x exactly half way
__device__ __forceinline__ void ks_term(float m, float inv_ulp, uint32_t &a, bool &tie)
{
    const float x = m * inv_ulp;                            // exact: inv_ulp is a power of two
    const bool big = x >= 33554432.0f;                      // 2^25 ulps or more: the sum leaves the binade whatever it is
    const float xs = big ? 0.0f : x;                        // (branch-free: the callers run this for every lane's every term)
    const float t = truncf(xs);
    tie = (xs - t) == 0.5f;                                 // exact difference
    const uint32_t r = (uint32_t)(tie ? t : rintf(xs));
    a = big ? kSumCap : r;
}
__device__ __forceinline__ uint32_t ks_step(uint32_t n, uint32_t a, bool tie) { return ks_sat(n + a + (tie ? ((n + a) & 1u) : 0u)); }

// inclusive scan of transducers over the 64 lanes of a wave (DPP: the pattern of k3_wave_scan; (0, 0) is the identity)
__device__ __forceinline__ KsPair ks_wave_scan(KsPair x)
{
#define KS_STEP(CTRL, RMASK, BC)                                                                                       \
    {                                                                                                                  \
        KsPair y;                                                                                                      \
        y.d0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.d0, CTRL, RMASK, 0xf, BC);                              \
        y.d1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.d1, CTRL, RMASK, 0xf, BC);                              \
        x = ks_compose(y, x);                                                                                          \
    }
    KS_STEP(0x111, 0xf, true) KS_STEP(0x112, 0xf, true) KS_STEP(0x114, 0xf, true) KS_STEP(0x118, 0xf, true)   // row_shr:1, 2, 4, 8
    KS_STEP(0x142, 0xa, false)                                                                                // row_bcast:15 into rows 1 and 3
    KS_STEP(0x143, 0xc, false)                                                                                // row_bcast:31 into rows 2 and 3
#undef KS_STEP
    return x;
}

}  // namespace amr
