// K1 for the blocks that do not fill a wave-tile -- the last < 64 blocks of a batch, every block of a small batch -- as ONE
// WAVE PER BLOCK instead of one lane per block.
//
// The tile kernel (k1_tile.h) gives a lane a whole reference block because the reference's running sum is
// sequential float32 (decode.go:232-236) and 64 lanes = 64 blocks fill a wave; a lone block then costs a whole wave
// life, 150-175 us, with 63 lanes idle.  Here the 64 lanes of a wave share ONE block:
//   * the first 256 samples of the Signal, 64 at a time: magnitudes in parallel (two LUT gathers from LDS, one float add:
//     decode.go:222), the running sum as the exact sequential chain k4_r900.h uses (lane i adds its magnitude to lane i-1's
//     sum, 63 dependent v_add_f32 with the DPP wave shift: 1050 cycles per 64 samples) -- the sum doubles every few
//     samples there;
//   * everything behind them, 512 samples at a time, eight consecutive samples per lane, with the EXACT PARALLEL form of
//     that sum (exact_sum.h, round 5): while the sum stays inside one binade every term is an integer transducer on its
//     mantissa; one DPP scan over the wave gives every partial sum of the chunk, bit for bit the reference's; the term at
//     which the sum leaves the binade (five or six per block) is added in float32 and the rest of the chunk scanned again;
//   * the sums go to an LDS ring, and the matched filter (decode.go:239-244: (c[i+CL]-c[i]) - (c[i+SL]-c[i+CL]), same
//     three roundings) runs in parallel over 64 outputs once their farthest sum exists; a ballot of the sign bits is two
//     bitstream words.
// A block of 4096 samples: ~70 k cycles as a DPP chain throughout (rounds 3-4: 50 us on an idle chip), ~25 k this way.
// Output, halo and carry conventions are K1Args' (the "tiled4" bitstream, the head buffer, zero history of a fresh Decoder).
#pragma once
#include "exact_sum.h"
#include "k1_common.h"

namespace amr {

template <int CL>
struct K1CGeom {
    static constexpr int SL = 2 * CL;
    static constexpr int HB = 4 * CL;                      // halo bytes = SL samples of history (decode.go:165-166)
    static constexpr int HBA = (HB + 127) & ~127;          // what the head buffer holds in front of block 0
    static constexpr uint32_t kWarm = 256;                 // samples summed by the DPP chain (>= the longest zero history, 192)
#ifndef AMR_K1C_TPL
#define AMR_K1C_TPL 8
#endif
    static constexpr uint32_t kTpl = AMR_K1C_TPL;          // consecutive samples per lane and scan chunk (4 or 8)
    static constexpr uint32_t kChunk = 64 * kTpl;          // samples per scan chunk
    static constexpr uint32_t kRing = 1024;                // sums kept (a power of two): the oldest output's c[i] .. the chunk being summed
    static constexpr uint32_t kLut = 0, kRingOff = 1024;   // LDS byte offsets
    static constexpr uint32_t kLds = kRingOff + kRing * 4;
    static_assert(SL + 64 + kChunk + 1 <= kRing && SL <= (int)kWarm, "ring span");
};


// p(lane) = carry + m(0) + ... + m(lane), added in that order (the reference's loop): see k4_r900.h k4_chain
__device__ __forceinline__ float k1c_chain(float carry, float mag)
{
    float p;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(p) : "v"(carry), "v"(mag));
#pragma unroll
    for (int i = 0; i < 63; ++i)
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(p) : "v"(mag));
    return p;
}

// grid = blocks handled this way, one wave each; block index = a.wg_first (a BLOCK index here) + blockIdx.x
template <int CL>
__global__ __launch_bounds__(64) void k1c_demod(const K1Args a)
{
    using G = K1CGeom<CL>;
    __shared__ __attribute__((aligned(16))) uint8_t lds[G::kLds];
    float *lut = reinterpret_cast<float *>(lds + G::kLut);
    float *ring = reinterpret_cast<float *>(lds + G::kRingOff);          // ring[k & (kRing - 1)] = csum[k] (decode.go:234)
    constexpr uint32_t RM = G::kRing - 1;
    const uint32_t lane = threadIdx.x;
    k1_announce(a, lane);
    const uint32_t b = a.wg_first + blockIdx.x;            // block of the launch (row 64 + b of the bitstream)
    const uint32_t bs = a.block_size, bs2 = bs * 2, wpb = bs >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) lut[lane + 64 * i] = a.lut[lane + 64 * i];

    // the block's bytes and its halo: the previous block's tail lies right in front of it, in the caller's batch or in
    // the head buffer (k1_tile_base); only block 0 of a launch without head rows takes its halo from the head buffer
    const uint32_t wg = b >> 6, r = b & 63u;
    const uint8_t *base = k1_tile_base<G::HBA>(a, wg, bs2) + (size_t)r * bs2;
    const bool halo_in_carry = b == 0 && !a.head_rows;
    const bool zero_hist = a.zero_halo && b == 0;          // fresh Decoder: Signal starts as zeros (decode.go:144)
    uint32_t *qrow = a.qt + (size_t)(wg + 1) * kRows * wpb + r * 4;   // word w of the row at qrow[(w >> 2) * 256 + (w & 3)]

    const uint32_t n_sig = bs + G::SL;                     // Signal = SL history samples + the block (decode.go:163-170)
    uint32_t q_next = 0;                                   // next group of 64 outputs to emit
    // outputs 64q .. 64q + 63 once csum[64q + 63 + SL] exists, i.e. `pos` samples have been summed
    auto emit = [&](uint32_t pos) {
        while (q_next * 64 < bs && q_next * 64 + 63 + G::SL <= pos) {
            const uint32_t i = q_next * 64 + lane;
            const float c0 = ring[i & RM], c1 = ring[(i + CL) & RM], c2 = ring[(i + G::SL) & RM];
            const float lo = c1 - c0;                      // decode.go:241
            const float up = c2 - c1;                      // decode.go:242
            const float f = lo - up;                       // decode.go:243
            const uint64_t neg = __ballot(__float_as_uint(f) >> 31);
            // Quantized = 1 - signbit (decode.go:244); first sample in bit 31 of its word
            const uint32_t w0 = __builtin_bitreverse32(~(uint32_t)neg), w1 = __builtin_bitreverse32(~(uint32_t)(neg >> 32));
            if (lane == 0) {
                const uint32_t w = 2 * q_next;
                *reinterpret_cast<uint2 *>(qrow + (size_t)(w >> 2) * 256 + (w & 3)) = make_uint2(w0, w1);
            }
            ++q_next;
        }
    };

    // ---- the first kWarm samples: 64 at a time, the sum as a DPP chain ----
    float cur = 0.0f;                                      // csum[0] = 0 (decode.go:232); wave-uniform
    if (lane == 0) ring[0] = 0.0f;
    auto load_iq = [&](uint32_t s) -> uint32_t {           // the two bytes of Signal sample s (0 outside the Signal)
        const int32_t off = (int32_t)(2 * s) - G::HB;      // byte offset from the block's first byte
        if (s >= n_sig) return 0u;
        const uint8_t *p = (off < 0 && halo_in_carry) ? a.carry + G::HBA + off : base + off;
        return *reinterpret_cast<const uint16_t *>(p);
    };
    {
        uint32_t iq = load_iq(lane);
        for (uint32_t s0 = 0; s0 < G::kWarm; s0 += 64) {
            const uint32_t iq_next = load_iq(s0 + 64 + lane);   // a chunk is a chain of 63 dependent additions: nothing else hides the load
            const uint32_t s = s0 + lane;
            float m = lut[iq & 0xff] + lut[iq >> 8];       // decode.go:222
            if (s >= n_sig || (zero_hist && s < (uint32_t)G::SL)) m = 0.0f;
            const float P = k1c_chain(cur, m);             // P = csum[s + 1] (decode.go:234)
            cur = __shfl(P, 63);
            ring[(s + 1) & RM] = P;
            iq = iq_next;
            __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): one wave, its own LDS writes
            __builtin_amdgcn_wave_barrier();
            emit(s0 + 64 < n_sig ? s0 + 64 : n_sig);
        }
    }

    // ---- the rest: kChunk samples at a time, lane l its kTpl consecutive samples, the sum by exact parallel scan ----
    constexpr uint32_t TPL = G::kTpl;
    struct Raw { uint2 v[TPL / 4]; };
    auto load4 = [&](uint32_t s0) -> Raw {                 // the bytes of samples s0 + TPL lane .. + TPL - 1 (behind kWarm: inside the block)
        Raw x;
#pragma unroll
        for (uint32_t h = 0; h < TPL / 4; ++h) {
            const uint32_t s = s0 + TPL * lane + 4 * h;    // n_sig is a multiple of 4: a group of four is all in or all out
            x.v[h] = s < n_sig ? *reinterpret_cast<const uint2 *>(base + (2 * s - G::HB)) : make_uint2(0u, 0u);
        }
        return x;
    };
    Raw raw = load4(G::kWarm);
    for (uint32_t s0 = G::kWarm; s0 < n_sig; s0 += G::kChunk) {
        const Raw raw_next = load4(s0 + G::kChunk);
        const uint32_t s = s0 + TPL * lane;
        float m[TPL];
        bool in[TPL];
#pragma unroll
        for (uint32_t h = 0; h < TPL / 4; ++h) {
            const uint2 w = raw.v[h];
            m[4 * h + 0] = lut[w.x & 0xff] + lut[(w.x >> 8) & 0xff];    // decode.go:222
            m[4 * h + 1] = lut[(w.x >> 16) & 0xff] + lut[w.x >> 24];
            m[4 * h + 2] = lut[w.y & 0xff] + lut[(w.y >> 8) & 0xff];
            m[4 * h + 3] = lut[(w.y >> 16) & 0xff] + lut[w.y >> 24];
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) in[4 * h + k] = s + 4 * h < n_sig;
        }
        const uint32_t n_here = n_sig - s0 < G::kChunk ? n_sig - s0 : G::kChunk;   // samples of this chunk
        uint32_t o = 0;                                    // samples of the chunk summed so far (wave-uniform)
        while (o < n_here) {
            const uint32_t cbits = __builtin_amdgcn_readfirstlane(__float_as_uint(cur));
            uint32_t j = 0xffffffffu, nb = 0;              // the term that leaves the binade, the mantissa in front of it
            float mj = 0.0f;
            if (cbits != 0) {
                const uint32_t E = cbits >> 23;            // biased exponent of the sum (positive, normal)
                const float ulp = __uint_as_float((E - 23u) << 23), inv_ulp = __uint_as_float((277u - E) << 23);
                const uint32_t n0 = (cbits & 0x7fffffu) | 0x800000u;
                uint32_t ta[TPL]; bool tt[TPL];
                KsPair f{0u, 0u};
#pragma unroll
                for (uint32_t k = 0; k < TPL; ++k) {
                    const bool act = in[k] && TPL * lane + k >= o;
                    ks_term(act ? m[k] : 0.0f, inv_ulp, ta[k], tt[k]);          // a term of 0.0 is the identity
                    KsPair g{ks_sat(ta[k] + (tt[k] ? (ta[k] & 1u) : 0u)), ks_sat(ta[k] + (tt[k] ? ((ta[k] + 1u) & 1u) : 0u))};
                    f = ks_compose(f, g);
                }
                const KsPair inc = ks_wave_scan(f);
                KsPair ex;
                ex.d0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc.d0, 0x138, 0xf, 0xf, true);   // wave_shr:1, lane 0: identity
                ex.d1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc.d1, 0x138, 0xf, 0xf, true);
                uint32_t nn = ks_sat(n0 + ((n0 & 1u) ? ex.d1 : ex.d0));
                uint32_t ev = 0xffffffffu, evn = 0;
                float evm = 0.0f;
                if (nn < (1u << 24)) {
#pragma unroll
                    for (uint32_t k = 0; k < TPL; ++k) {
                        if (ev != 0xffffffffu || !(in[k] && TPL * lane + k >= o)) continue;
                        const uint32_t before = nn;
                        nn = ks_step(nn, ta[k], tt[k]);
                        if (nn >= (1u << 24)) { ev = TPL * lane + k; evn = before; evm = m[k]; }
                        else ring[(s + k + 1) & RM] = (float)nn * ulp;          // exact: nn < 2^24, ulp a power of two
                    }
                }
                const uint64_t em = __ballot(ev != 0xffffffffu);
                if (em == 0) {                             // the whole rest of the chunk inside the binade
                    const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)nn, 63);
                    cur = (float)last * ulp;
                    o = n_here;
                    continue;
                }
                const int L = __ffsll((unsigned long long)em) - 1;              // lanes are in sample order: the first one
                j = (uint32_t)__builtin_amdgcn_readlane((int)ev, L);
                nb = (uint32_t)__builtin_amdgcn_readlane((int)evn, L);
                mj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(evm), L));
                cur = (float)nb * ulp;                     // the sum in front of that term
            } else {
                // a sum still zero behind kWarm samples (never with real samples: the LUT has no zero): one term per round
                j = o;
                float mine = 0.0f;
#pragma unroll
                for (uint32_t k = 0; k < TPL; ++k) mine = (j % TPL) == k && in[k] ? m[k] : mine;
                mj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(mine), (int)(j / TPL)));
            }
            cur = cur + mj;                                // THAT addition in float32 (decode.go:234)
            if (lane == 0) ring[(s0 + j + 1) & RM] = cur;
            o = j + 1;
        }
        raw = raw_next;
        __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): one wave, its own LDS writes
        __builtin_amdgcn_wave_barrier();
        emit(s0 + n_here);
    }
}

}  // namespace amr
