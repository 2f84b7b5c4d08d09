// Deterministic integer-only synthetic IQ for bench and tests.
// Not part of the decode path.  The same generator exists in numpy
// (rtlamr_amd/synth.py) and the two are compared byte for byte in the tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amr {

__host__ __device__ inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// 8 samples (16 bytes) per thread; n_samples must be a multiple of 8.
// UNIFORM: the second input distribution of SURVEY.md 8d -- uniform random bytes, I = bits 32..39 and Q = bits 40..47 of
// the same hash: every LUT entry equally likely (the worst case for LDS bank conflicts in K1's gathers).
template <bool UNIFORM>
__global__ void k_synth_noise(uint8_t *iq, uint64_t n_samples, uint64_t seed, uint64_t first_sample)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t s0 = t * 8;
    if (s0 >= n_samples) return;
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint64_t h = splitmix64(seed ^ (first_sample + s0 + 2 * k + j));
            const uint32_t I = UNIFORM ? (uint32_t)(h >> 32) & 0xFFu : 119u + __popc((uint32_t)h & 0xFFFFu);
            const uint32_t Q = UNIFORM ? (uint32_t)(h >> 40) & 0xFFu : 120u + __popc((uint32_t)(h >> 16) & 0xFFFFu);
            v |= (I | (Q << 8)) << (16 * j);
        }
        w[k] = v;
    }
    reinterpret_cast<uint4 *>(iq)[t] = make_uint4(w[0], w[1], w[2], w[3]);
}

struct PlantArgs {
    uint8_t *iq;
    uint64_t n_samples, first_sample;
    const uint64_t *start;   // [n_packets] stream sample index of the first chip
    const uint8_t *bits;     // [n_packets*stride], MSB first
    const int8_t *d_i, *d_q; // [n_packets]
    uint32_t n_packets, n_bits, stride, chip_length;
};

// grid.y = packet, grid.x*blockDim.x covers n_bits*2*CL samples of the packet.
__global__ void k_synth_plant(const PlantArgs a)
{
    const uint32_t j = blockIdx.y;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sl = 2 * a.chip_length;
    if (s >= a.n_bits * sl) return;
    const uint32_t p = s / sl, within = s % sl;
    const uint32_t bit = (a.bits[(size_t)j * a.stride + (p >> 3)] >> (7 - (p & 7))) & 1u;
    const bool high = (within < a.chip_length) == (bit == 1u);  // bit 1 = high,low ; bit 0 = low,high
    if (!high) return;
    const uint64_t n = a.start[j] + s;
    if (n < a.first_sample || n >= a.first_sample + a.n_samples) return;
    uint8_t *px = a.iq + 2 * (n - a.first_sample);
    int I = (int)px[0] + a.d_i[j], Q = (int)px[1] + a.d_q[j];
    px[0] = (uint8_t)(I < 0 ? 0 : I > 255 ? 255 : I);
    px[1] = (uint8_t)(Q < 0 ? 0 : Q > 255 ? 255 : Q);
}

}  // namespace amr
