// The bits Decoder.Slice never clears (decode.go:353-375).  Slice shifts every symbol into d.pkt[p >> 3]
// (pkt[p>>3] <<= 1; pkt[p>>3] |= bit) and d.pkt lives as long as the Decoder: when PacketSymbols is not a multiple of 8
// -- r900 registered alone or with scm: 116 symbols -- the last byte receives only r = PacketSymbols % 8 shifts per hit,
// so above its r fresh bits it still holds what the hits sliced BEFORE it left there:
//     B(j) = ( B(j-1) << r  |  fresh(j) ) & 0xff  =  fresh(j) | fresh(j-1) << r | fresh(j-2) << 2r | ...   (8 bits of it)
// where j-1, j-2 .. are the hits in the order the Decoder slices them: call by call, inside a call preamble by preamble
// in registration order (the oracle's order; the Go map's iteration order is random with more than one preamble), inside
// a preamble idx ascending -- across calls and across batches.  Parsers never look at those bits, NewData copies them
// into Data.Bytes all the same, and so does this kernel: K3 writes the r fresh bits right-aligned (high bits zero), this
// pass ORs in the predecessors' fresh bits.  One thread per hit; the low r bits of a byte never change, so a thread may read
// its predecessors' bytes while their threads rewrite them, and the pass may run twice (a re-search).  The hit before the
// batch's first one is the previous batch's last: its final byte travels in a device byte per slot (carry).
#pragma once
#include "k2_common.h"

namespace amr {

struct StaleArgs {
    uint8_t *out;               // packed result [hit_block u64 x n | hit_idx u32 x n | pkt x n] (K3Args::out)
    const uint64_t *offs_pre;   // [n_pre + 1] per-preamble bases, [n_pre] = n
    const uint32_t *overflow;   // K2's overflow word: non-zero = K3 wrote nothing, the host searches again
    const uint8_t *carry_in;    // final last byte of the last hit sliced before this batch
    uint8_t *carry_out;         // ... of this batch's last hit (carry_in when the batch has none)
    uint64_t cap;               // hits the buffer holds
    uint32_t n_pre, pkt_bytes, r;   // r = PacketSymbols % 8 (1 .. 7)
};

// index (into the packed result) of the hit sliced right before hit `cur`, or ~0 when `cur` is the batch's first
__device__ __forceinline__ uint64_t stale_prev(const uint64_t *hb, const uint64_t *offs, uint32_t n_pre, uint64_t cur, uint32_t q)
{
    const uint64_t b = hb[cur];
    if (cur > offs[q] && hb[cur - 1] == b) return cur - 1;             // same call, same preamble: the idx before it
    uint64_t best = ~0ull, best_b = 0;
    uint32_t best_q = 0;
    for (uint32_t p = 0; p < n_pre; ++p) {
        // preamble p's last hit in a call <= b (p in front of q: the same call counts) or < b
        const uint64_t lim = b + (p < q ? 1u : 0u);                    // first block that no longer qualifies
        uint64_t lo = offs[p], hi = offs[p + 1];
        while (lo < hi) {                                              // first index with hb >= lim
            const uint64_t mid = (lo + hi) >> 1;
            if (hb[mid] < lim) lo = mid + 1; else hi = mid;
        }
        if (lo == offs[p]) continue;
        const uint64_t c = lo - 1, cb = hb[c];
        if (best == ~0ull || cb > best_b || (cb == best_b && p > best_q)) { best = c; best_b = cb; best_q = p; }
    }
    return best;
}

__global__ __launch_bounds__(256) void k_stale_bits(const StaleArgs a)
{
    const uint64_t n = a.offs_pre[a.n_pre];
    if (*a.overflow != 0 || n > a.cap) return;                          // searched again: that run's pass writes the carry
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0) { if (j == 0) *a.carry_out = *a.carry_in; return; }
    if (j >= n) return;
    const uint64_t *hb = reinterpret_cast<const uint64_t *>(a.out);
    uint8_t *last = a.out + n * 12 + (uint64_t)a.pkt_bytes - 1;        // last byte of packet i: last[i * pkt_bytes]
    uint32_t q = 0;
    while (q + 1 < a.n_pre && j >= a.offs_pre[q + 1]) ++q;
    const uint32_t r = a.r, mask = (1u << r) - 1u;
    uint32_t acc = last[j * a.pkt_bytes] & mask;
    uint64_t cur = j;
    uint32_t cq = q;
    for (uint32_t sh = r; sh < 8; sh += r) {
        const uint64_t p = stale_prev(hb, a.offs_pre, a.n_pre, cur, cq);
        if (p == ~0ull) { acc |= (uint32_t)*a.carry_in << sh; break; } // the batch's first hits: what the previous batch left
        acc |= (last[p * a.pkt_bytes] & mask) << sh;
        cur = p;
        cq = 0;
        while (cq + 1 < a.n_pre && cur >= a.offs_pre[cq + 1]) ++cq;
    }
    acc &= 0xffu;
    // the batch's last hit in slicing order = the largest (call, preamble) among the lists' last hits: its byte is the carry
    bool is_last = j + 1 == a.offs_pre[q + 1];
    for (uint32_t p = 0; p < a.n_pre && is_last; ++p) {
        if (p == q || a.offs_pre[p + 1] == a.offs_pre[p]) continue;
        const uint64_t ob = hb[a.offs_pre[p + 1] - 1];
        if (ob > hb[j] || (ob == hb[j] && p > q)) is_last = false;
    }
    last[j * a.pkt_bytes] = (uint8_t)acc;
    if (is_last) *a.carry_out = (uint8_t)acc;
}

}  // namespace amr
