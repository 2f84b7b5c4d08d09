// K4 -- the r900 parser's second-stage matched filter, evaluated only where it is consumed.
//
// Reference (r900/r900.go): on EVERY Decode call Parser.Parse slides its own BufferLength-long copy of the
// magnitudes (r900.go:168-170), recomputes a sequential float32 running sum over all of it (r900.go:96-100),
// quantizes every position into one of six symbols (r900.go:119-149) and then reads 42 of those symbols per
// preamble hit, 4 chips apart (r900.go:187-193).  Only those 42 values per hit are observable, so this kernel
// computes exactly them: one lane = one r900 preamble hit; the call's running sum is replayed from the start of
// the parser's buffer -- same order, same float32 roundings, once per wave and call -- the lane samples it at the
// 169 chip boundaries its 42 symbols span, and applies the reference's a0/a1/a2 arithmetic operation for operation.
//
// The parser's buffer at call k holds the samples [k*BS - PL, k*BS + BS) of the stream (zero magnitude before
// the stream starts, r900.go:163-165 allocates zeros); samples before the current batch come from `hist`, the
// last PL samples that preceded it.  Hits are sorted by (call, idx): the lanes of a wave mostly share their
// call, so their loads hit the same addresses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k2_common.h"

namespace amr {

constexpr uint32_t kK4Split = 4;   // waves per 64-hit chunk, one distinct Decode call each

struct K4Args {
    const uint8_t *iq;        // batch block 0 (device)
    const uint8_t *hist;      // 2*PL bytes: the PL samples that precede the batch (only the last hist_valid are real)
    const float *lut;         // NewMagLUT
    const uint8_t *out_packed;   // packed result of K3: [hit_block u64 x n | hit_idx u32 x n | ...]
    const uint64_t *offs_pre;    // [n_pre+1]
    const uint32_t *overflow;    // K2's overflow word: non-zero = K3 left the packed result unwritten, the host searches again
    uint8_t *digits;          // [cap][42]
    uint64_t cap;             // hits `digits` and out_packed hold
    uint64_t block_base;      // call index of batch block 0
    uint32_t n_pre, pid;      // the r900 preamble id
    uint32_t hist_valid;      // real samples in hist (<= PL)
    uint32_t block_size, lg_block_size, packet_length, preamble_length, symbol_length, chip_length;
};

// One wave = 64 consecutive r900 hits (sorted by call, idx).  All hits of one Decode call need the SAME running sum
// (r900.go:96-100 restarts it at the start of the parser's buffer on every call), so the wave computes it once per
// distinct call among its hits, cooperatively: 64 lanes load 64 consecutive samples (one coalesced 128-byte read), take
// their magnitudes from the LUT in parallel, and a 64-step chain of v_add_f32 with the DPP wave shift -- lane i adds its
// magnitude to the sum of lane i-1 -- reproduces the reference's sequential float32 additions exactly, one instruction
// per sample.  A hit's lane then picks the sums at its chip boundaries out of the tile with ds_bpermute (lane index =
// csum index - tile start - 1) and applies a0/a1/a2 every fourth boundary.  (The first version replayed the sum in
// every lane, ~15 instructions per sample and lane: 1.5 ms per GiB at 36 k hits.)
__device__ __forceinline__ float k4_chain(float carry, float mag)
{
    // p(lane) = csum after this lane's sample.  Lane 0 first (carry + m0, computed by every lane, right in lane 0);
    // then 63 steps "p = p(lane-1) + m": lane i is final after step i, later steps recompute the same value; lane 0 has
    // no lane below it and is left alone by the shift (bound_ctrl 0).  s_nop 1: VALU write -> DPP read of the same VGPR.
    float p;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(p) : "v"(carry), "v"(mag));
#pragma unroll
    for (int i = 0; i < 63; ++i)
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(p) : "v"(mag));
    return p;
}

__global__ __launch_bounds__(64) void k4_r900_digits(const K4Args a)
{
    __shared__ float lut[256];
    for (int i = threadIdx.x; i < 256; i += 64) lut[i] = a.lut[i];
    __syncthreads();
    // after an overflow K3 has published counts but no hit records: the positions in out_packed are stale or
    // uninitialised and must not be turned into addresses
    if (*a.overflow) return;
    const uint64_t total = a.offs_pre[a.n_pre];
    if (total > a.cap) return;                       // the host grows the buffers and runs the search again
    const uint64_t lo = a.offs_pre[a.pid], n = a.offs_pre[a.pid + 1] - lo;
    const uint64_t gid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const bool active = gid < n;
    if ((uint64_t)blockIdx.x * 64 >= n) return;
    const uint32_t lane = threadIdx.x;
    const uint64_t slot = lo + (active ? gid : 0);
    const uint64_t *hit_block = reinterpret_cast<const uint64_t *>(a.out_packed);
    const uint32_t *hit_idx = reinterpret_cast<const uint32_t *>(a.out_packed + total * 8);
    const int64_t k = active ? (int64_t)(hit_block[slot] - a.block_base) : -1;   // call index inside the batch
    const uint32_t idx = hit_idx[slot];
    const uint32_t CL = a.chip_length, PL = a.packet_length;
    const uint32_t payload = idx + a.preamble_length - a.symbol_length;   // r900.go:183
    const uint32_t last = payload + 168 * CL;                              // csum index of the last chip boundary
    uint8_t *dg = a.digits + gid * kR900Digits;

    float c[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // running-sum samples at the last five chip boundaries
    uint32_t m = 0;                            // boundaries recorded so far (boundary m sits at csum index payload + m*CL)
    uint32_t next_pt = payload;
    auto record = [&](float v) {
        c[0] = c[1]; c[1] = c[2]; c[2] = c[3]; c[3] = c[4]; c[4] = v;
        if (m >= 4 && (m & 3) == 0) {          // boundaries m-4..m = one symbol: r900.go:119-148, operation for operation
            const float c0 = c[0];
            const float c1 = c[1] + c[1];
            const float c2 = c[2] + c[2];
            const float c3 = c[3] + c[3];
            const float c4 = c[4];
            const float a0 = c2 - c4 - c0;               // 1100
            const float a1 = c1 - c2 + c3 - c4 - c0;     // 1010
            const float a2 = c1 - c3 + c4 - c0;          // 1001
            float max_abs = fabsf(a0), val = a0;
            uint32_t arg = 0;
            if (fabsf(a1) > max_abs) { max_abs = fabsf(a1); arg = 1; val = a1; }
            if (fabsf(a2) > max_abs) { max_abs = fabsf(a2); arg = 2; val = a2; }
            if (val > 0.f) arg += 3;
            dg[(m >> 2) - 1] = (uint8_t)arg;
        }
        m += 1;
        next_pt += CL;
    };

    // A walk costs ~0.3 ms of latency whatever the number of hits that share it (tools/chain_bench.hip: 1050 cycles per
    // 64 samples for the chain), so the distinct calls of a 64-hit chunk are spread over the kK4Split waves launched
    // for it (blockIdx.y): wave r takes the r-th distinct call, the last one also whatever remains.
    uint64_t todo = __ballot(active);
    for (uint32_t round = 0; todo; ++round) {
        const int lead = __ffsll((unsigned long long)todo) - 1;
        const uint32_t k_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, lead);
        const uint32_t k_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)k >> 32), lead);
        const int64_t kc = (int64_t)(((uint64_t)k_hi << 32) | k_lo);
        const bool mine = active && k == kc;
        if (round != blockIdx.y && !(blockIdx.y == kK4Split - 1 && round >= kK4Split)) {   // another wave's call
            todo &= ~__ballot(mine);
            continue;
        }
        // sample j of the parser's buffer = batch sample kc*BS - PL + j; nothing exists before the stream start
        const int64_t n0 = (kc << a.lg_block_size) - (int64_t)PL;
        const int64_t first_real = -(int64_t)a.hist_valid - n0;           // first buffer index with a real sample
        const uint32_t j0 = first_real < 0 ? 0u : (uint32_t)first_real;
        // chip boundaries that lie in the all-zero prefix (before the stream start): the running sum is still 0 there
        while (mine && m <= 168 && next_pt <= j0) record(0.f);
        uint32_t wave_last = mine ? last : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_xor(wave_last, d);
            wave_last = o > wave_last ? o : wave_last;
        }
        float carry = 0.f;
        // the samples of a tile: one 16-bit load per lane (128 contiguous bytes per wave), issued two tiles ahead of
        // their chain so that the memory latency hides behind two chains
        auto fetch = [&](uint32_t tile) -> uint32_t {
            const uint32_t j = tile + lane;
            const int64_t nb = n0 + j;
            if (j >= j0 && j < wave_last)
                return nb >= 0 ? *reinterpret_cast<const uint16_t *>(a.iq + 2 * nb)
                               : *reinterpret_cast<const uint16_t *>(a.hist + 2 * ((int64_t)PL + nb));
            return 0xffffffffu;                        // no sample here (before the stream start / past the last boundary)
        };
        const uint32_t t0 = j0 & ~63u;
        uint32_t v0 = fetch(t0), v1 = fetch(t0 + 64);
        for (uint32_t tile = t0; tile < wave_last; tile += 64) {
            const uint32_t v = v0;
            v0 = v1;
            v1 = fetch(tile + 128);
            // samples before the stream start add nothing (x + 0.0 is exact)
            const float mag = v == 0xffffffffu ? 0.f : lut[v & 0xff] + lut[v >> 8];      // decode.go:222
            const float p = k4_chain(carry, mag);     // p(lane) = csum[tile + lane + 1]   (r900.go:97-99)
            carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 63));   // (the builtin is typed int)
            for (;;) {                                // boundaries inside (tile, tile + 64]: several when CL < 64
                const bool at = mine && m <= 168 && next_pt > tile && next_pt <= tile + 64;
                if (!__any(at)) break;
                const int src = at ? (int)(next_pt - tile - 1) : 0;
                const float cv = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(p)));
                if (at) record(cv);
            }
        }
        todo &= ~__ballot(mine);
    }
}

// The last PL samples that precede the next batch: new[i] = sample (i - PL + n_batch) of the batch just processed,
// taken from the batch or, where that index is negative, from the old history.
struct IqHistArgs {
    const uint8_t *iq;      // batch just processed
    const uint8_t *old_hist;
    uint8_t *new_hist;
    uint64_t n_batch;       // samples in the batch
    uint32_t packet_length; // PL (multiple of 16)
};

__global__ __launch_bounds__(256) void k_iqhist_update(const IqHistArgs a)
{
    const uint32_t chunks = a.packet_length / 8;     // 16-byte chunks of 8 samples
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < chunks; i += gridDim.x * 256) {
        const int64_t n = (int64_t)i * 8 - (int64_t)a.packet_length + (int64_t)a.n_batch;
        const uint4 v = n >= 0 ? *reinterpret_cast<const uint4 *>(a.iq + 2 * n)
                               : *reinterpret_cast<const uint4 *>(a.old_hist + 2 * ((int64_t)a.packet_length + n));
        reinterpret_cast<uint4 *>(a.new_hist)[i] = v;
    }
}

}  // namespace amr
