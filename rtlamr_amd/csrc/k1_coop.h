// K1 for the blocks that do not fill a wave-tile -- the single block of the unchanged main.go loop (main.go:235: one
// Decode call per block), the last < 64 blocks of a batch -- as ONE WAVE PER BLOCK instead of one lane per block.
//
// The tile kernels (k1_tile.h, k1_demod.h) give a lane a whole reference block because the reference's running sum is
// sequential float32 (decode.go:232-236) and 64 lanes = 64 blocks fill a wave; a lone block then costs a whole wave
// life, 150-175 us, with 63 lanes idle.  Here the 64 lanes of a wave take 64 CONSECUTIVE SAMPLES of one block:
//   * magnitudes in parallel (two byte loads, two LUT gathers from LDS, one float add: decode.go:222);
//   * the running sum as the exact sequential chain k4_r900.h uses: lane i adds its magnitude to lane i-1's sum, 63
//     dependent v_add_f32 with the DPP wave shift -- the same additions in the same order as the reference's loop, about
//     1050 cycles per 64 samples (tools/chain_bench.hip);
//   * the sums go to a small LDS ring, and the matched filter (decode.go:239-244: (c[i+CL]-c[i]) - (c[i+SL]-c[i+CL]), same
//     three roundings) runs in parallel over 64 outputs once their farthest sum exists; a ballot of the sign bits is two
//     bitstream words.
// A block of 4096 samples is 67 chunks: ~30 us instead of ~160.  Output, halo and carry conventions are K1Args' (the
// "tiled4" bitstream, the head buffer, zero history of a fresh Decoder).
#pragma once
#include "k1_demod.h"

namespace amr {

template <int CL>
struct K1CGeom {
    static constexpr int SL = 2 * CL;
    static constexpr int HB = 4 * CL;                      // halo bytes = SL samples of history (decode.go:165-166)
    static constexpr int HBA = (HB + 127) & ~127;          // what the head buffer holds in front of block 0
    static constexpr int D = (SL - 1 + 63) / 64;           // output group q is complete after sample chunk q + D
    static constexpr int RING = 64 * (D + 2);              // sums kept: P[64(j-D)] .. P[64j+64]
    static constexpr uint32_t kLut = 0, kRing = 1024;      // LDS byte offsets
    static constexpr uint32_t kLds = kRing + RING * 4;
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
    float *ring = reinterpret_cast<float *>(lds + G::kRing);
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
    const uint32_t n_chunks = (n_sig + 63) >> 6;
    float carry = 0.0f;                                    // csum[0] = 0 (decode.go:232)
    if (lane == 0) ring[0] = 0.0f;
    // the two bytes of Signal sample 64j + lane (0 outside the Signal); loaded one chunk ahead: a chunk is a chain of 63
    // dependent additions, nothing else could hide the load behind it
    auto load_iq = [&](uint32_t j) -> uint32_t {
        const uint32_t s = j * 64 + lane;
        const int32_t off = (int32_t)(2 * s) - G::HB;      // byte offset from the block's first byte
        if (s >= n_sig) return 0u;
        const uint8_t *p = (off < 0 && halo_in_carry) ? a.carry + G::HBA + off : base + off;
        return *reinterpret_cast<const uint16_t *>(p);
    };
    uint32_t iq = load_iq(0);
    for (uint32_t j = 0; j < n_chunks + G::D; ++j) {       // D more rounds drain the last output groups
        if (j < n_chunks) {
            const uint32_t iq_next = load_iq(j + 1);       // j + 1 == n_chunks: past the Signal, 0
            const uint32_t s = j * 64 + lane;              // sample of Signal
            float m = lut[iq & 0xff] + lut[iq >> 8];       // decode.go:222
            if (s >= n_sig || (zero_hist && s < (uint32_t)G::SL)) m = 0.0f;
            const float P = k1c_chain(carry, m);           // P = csum[s + 1] (decode.go:234)
            carry = __shfl(P, 63);
            ring[(s + 1) % G::RING] = P;
            iq = iq_next;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): one wave, its own LDS writes
        __builtin_amdgcn_wave_barrier();
        if (j >= (uint32_t)G::D) {
            const uint32_t q = j - G::D;                   // outputs 64q .. 64q+63
            const uint32_t i = q * 64 + lane;
            if (q * 64 < bs) {                             // wave-uniform
                const float c0 = ring[i % G::RING], c1 = ring[(i + CL) % G::RING], c2 = ring[(i + G::SL) % G::RING];
                const float lo = c1 - c0;                  // decode.go:241
                const float up = c2 - c1;                  // decode.go:242
                const float f = lo - up;                   // decode.go:243
                const uint64_t neg = __ballot(__float_as_uint(f) >> 31);
                // Quantized = 1 - signbit (decode.go:244); first sample in bit 31 of its word
                const uint32_t w0 = __builtin_bitreverse32(~(uint32_t)neg), w1 = __builtin_bitreverse32(~(uint32_t)(neg >> 32));
                if (lane == 0) {
                    const uint32_t w = 2 * q;
                    *reinterpret_cast<uint2 *>(qrow + (size_t)(w >> 2) * 256 + (w & 3)) = make_uint2(w0, w1);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace amr
