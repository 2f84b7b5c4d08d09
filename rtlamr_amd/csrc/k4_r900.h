// K4 -- the r900 parser's second-stage matched filter, evaluated only where it is consumed.
//
// Reference (r900/r900.go): on EVERY Decode call Parser.Parse slides its own BufferLength-long copy of the
// magnitudes (r900.go:168-170), recomputes a sequential float32 running sum over all of it (r900.go:96-100),
// quantizes every position into one of six symbols (r900.go:119-149) and then reads 42 of those symbols per
// preamble hit, 4 chips apart (r900.go:187-193).  Only those 42 values per hit are observable, so this kernel
// computes exactly them: one lane = one r900 preamble hit; the lane replays the call's running sum from the
// start of the parser's buffer -- same order, same float32 roundings -- samples it at the 169 chip boundaries
// its 42 symbols span, and applies the reference's a0/a1/a2 arithmetic operation for operation.
//
// The parser's buffer at call k holds the samples [k*BS - PL, k*BS + BS) of the stream (zero magnitude before
// the stream starts, r900.go:163-165 allocates zeros); samples before the current batch come from `hist`, the
// last PL samples that preceded it.  Hits are sorted by (call, idx): the lanes of a wave mostly share their
// call, so their loads hit the same addresses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amr {

constexpr int kR900Digits = 42;   // PayloadSymbols, r900.go:30

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
    const uint64_t slot = lo + (active ? gid : 0);
    if ((uint64_t)blockIdx.x * 64 >= n) return;
    const uint64_t *hit_block = reinterpret_cast<const uint64_t *>(a.out_packed);
    const uint32_t *hit_idx = reinterpret_cast<const uint32_t *>(a.out_packed + total * 8);
    const int64_t k = (int64_t)(hit_block[slot] - a.block_base);          // call index inside the batch
    const uint32_t idx = hit_idx[slot];
    const uint32_t CL = a.chip_length, PL = a.packet_length;
    const uint32_t payload = idx + a.preamble_length - a.symbol_length;   // r900.go:183
    const uint32_t last = active ? payload + 168 * CL : 0;                 // csum index of the last chip boundary
    // sample j of the parser's buffer = batch sample k*BS - PL + j; nothing exists before the stream start
    const int64_t n0 = (k << a.lg_block_size) - (int64_t)PL;              // batch-relative sample of buffer index 0
    int64_t first_real = -(int64_t)a.hist_valid - n0;                     // first buffer index with a real sample
    const uint32_t j0 = first_real < 0 ? 0u : (uint32_t)first_real;

    float c[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // running-sum samples at the last five chip boundaries
    uint32_t m = 0;                            // boundaries recorded so far (boundary m sits at csum index payload + m*CL)
    uint32_t next_pt = payload;
    uint8_t *dg = a.digits + gid * kR900Digits;
    float sum = 0.f;

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
            if (active) dg[(m >> 2) - 1] = (uint8_t)arg;
        }
        m += 1;
        next_pt += CL;
    };
    // chip boundaries that lie in the all-zero prefix (before the stream start): the running sum is still 0 there
    while (active && m <= 168 && next_pt <= j0) record(0.f);

    // walk the buffer 8 samples (16 bytes, never straddling the batch start: PL and BS are multiples of 16) at a time
    uint32_t wave_last = last;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_xor(wave_last, d);
        wave_last = o > wave_last ? o : wave_last;
    }
    // The walk.  Chip boundaries are CL samples apart, so in ANY window of CL consecutive samples a lane meets exactly
    // one of its own; the lanes of a wave meet theirs at different samples (adjacent hits), which would run the
    // record() body -- under a one-lane mask -- for almost every sample.  Instead a boundary is only CAPTURED when it
    // passes (two predicated moves) and all lanes record together once per CL samples.  32 samples per round, the four
    // 16-byte loads issued together.
    const uint32_t chunks_per_epoch = CL >> 3;       // every legal chip length is a multiple of 8
    uint32_t cc = 0;
    float cap = 0.f;
    bool pending = false;
    for (uint32_t j32 = j0 & ~31u; j32 < wave_last; j32 += 32) {
        uint4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t nb = n0 + j32 + 8 * q;     // batch-relative sample of the chunk; chunks past `last` are not used
            const bool need = active && j32 + 8 * q < last && j32 + 8 * q + 8 > j0;
            w[q] = !need ? make_uint4(0, 0, 0, 0)
                 : nb >= 0 ? *reinterpret_cast<const uint4 *>(a.iq + 2 * nb)
                           : *reinterpret_cast<const uint4 *>(a.hist + 2 * ((int64_t)PL + nb));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t dw[4] = {w[q].x, w[q].y, w[q].z, w[q].w};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const uint32_t j = j32 + 8 * q + s;
                const uint32_t v = dw[s >> 1] >> ((s & 1) * 16);
                const float mag = lut[v & 0xff] + lut[(v >> 8) & 0xff];       // decode.go:222
                const bool in = j >= j0 && j < last;
                sum = in ? sum + mag : sum;                                   // r900.go:97-99
                const bool at = in && j + 1 == next_pt;
                cap = at ? sum : cap;
                pending = pending || at;
            }
            if (++cc == chunks_per_epoch) {          // wave-uniform
                cc = 0;
                if (pending) { record(cap); pending = false; }
            }
        }
    }
    if (pending) record(cap);
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
