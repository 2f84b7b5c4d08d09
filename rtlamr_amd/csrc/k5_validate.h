// K5 -- per-hit validation on the GPU: the checksum test every rtlamr parser applies first and
// the removal of repeated packets, so that only hits a parser could turn into a message leave the device.
//
// Reference semantics restated (the parsers still run unchanged on what is left and emit the same messages):
//   crc.Checksum            crc/crc.go:49-55   crc = crc<<8 ^ table[crc>>8 ^ byte], table from crc/crc.go:34-47
//   SCM    Parse            scm/scm.go:61-90   Checksum(Bytes[2:12]) != 0        -> skip     (BCH 0x6F63, init 0)
//   SCM+   Parse            scmplus/scmplus.go:62-92   Checksum(Bytes[2:16]) != Residue -> skip  (CCITT)
//   IDM / NetIDM Parse      idm/idm.go:62-98, netidm/netidm.go:73-110
//                           Checksum(Bytes[4:92]) != Residue -> skip; Checksum(Bytes[9:13] ++ Bytes[88:90]) != Residue -> skip
//   seen[string(Bytes)]     first line of every Parse loop: a byte string is handled once per Decode call.
// A hit is dropped here when a check fails, or when its first dedupe_bytes packet bytes equal those of the hit
// right before it in the same (preamble, block) list -- a subset of what `seen` drops, so the parser's own `seen`
// finishes the job and the message stream is unchanged.  The order of the surviving hits is kept.
//
// Two small kernels (the hit count is known on the device only): flag + per-chunk counts, then an ordered
// compaction into a second packed buffer of the same layout as K3's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "amrdemod.h"

namespace amr {

constexpr int kValChunk = 256;   // hits per workgroup of k5_flag / k5_compact

struct ValCheck {
    uint16_t init, poly, residue, n_spans;
    uint16_t off[2], len[2];
};
struct ValRule {
    int32_t n_checks;       // 0: no checksum test
    int32_t dedupe_bytes;   // 0: keep repeated packets
    ValCheck chk[2];
};

struct K5Args {
    const uint8_t *in;          // K3's packed result [hit_block u64 x n | hit_idx u32 x n | pkt x n]
    uint8_t *out;               // same layout, n' = surviving hits
    const uint64_t *offs_pre;   // [n_pre+1] from k3_slice
    uint64_t *offs_val;         // [n_pre+1] offsets into the validated list (device)
    uint64_t *h_offs_val;       // the same in pinned host memory
    uint32_t *chunk;            // [cap/kValChunk + 2]: survivors per chunk
    uint8_t *keep;              // [cap]
    const uint32_t *overflow;   // K2's overflow word: the host searches again, nothing here is used
    uint64_t cap;               // hits the buffers hold
    uint32_t n_pre, pkt_bytes;
    ValRule rule[AMR_MAX_PREAMBLES];
};

__device__ __forceinline__ bool k5_usable(const K5Args &a, uint64_t &total)
{
    total = a.offs_pre[a.n_pre];
    return *a.overflow == 0 && total <= a.cap;
}

__device__ __forceinline__ uint32_t k5_preamble_of(const K5Args &a, uint64_t g)
{
    uint32_t p = 0;
    for (uint32_t q = 1; q < a.n_pre; ++q) p += g >= a.offs_pre[q] ? 1u : 0u;
    return p;
}

__global__ __launch_bounds__(kValChunk) void k5_flag(const K5Args a)
{
    __shared__ uint16_t tbl[AMR_MAX_PREAMBLES * 2][256];
    __shared__ uint32_t wsum[kValChunk / 64];
    uint64_t total;
    if (!k5_usable(a, total)) return;
    const uint64_t g0 = (uint64_t)blockIdx.x * kValChunk;
    if (g0 >= total) return;
    // crc.NewTable (crc/crc.go:34-47), one table per configured check
    for (uint32_t q = 0; q < a.n_pre; ++q)
        for (int c = 0; c < a.rule[q].n_checks; ++c) {
            const uint16_t poly = a.rule[q].chk[c].poly;
            uint16_t crc = (uint16_t)(threadIdx.x << 8);
            for (int b = 0; b < 8; ++b) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ poly) : (uint16_t)(crc << 1);
            tbl[q * 2 + c][threadIdx.x] = crc;
        }
    __syncthreads();
    const uint64_t g = g0 + threadIdx.x;
    bool keep = false;
    if (g < total) {
        const uint64_t *hit_block = reinterpret_cast<const uint64_t *>(a.in);
        const uint8_t *pkt = a.in + total * 12 + g * a.pkt_bytes;
        const uint32_t p = k5_preamble_of(a, g);
        const ValRule &r = a.rule[p];
        keep = true;
        for (int c = 0; c < r.n_checks; ++c) {
            const ValCheck &k = r.chk[c];
            uint16_t crc = k.init;
            for (uint32_t s = 0; s < k.n_spans; ++s)
                for (uint32_t i = 0; i < k.len[s]; ++i)
                    crc = (uint16_t)((crc << 8) ^ tbl[p * 2 + c][(crc >> 8) ^ pkt[k.off[s] + i]]);   // crc/crc.go:52
            keep = keep && crc == k.residue;
        }
        if (keep && r.dedupe_bytes > 0 && g > a.offs_pre[p] && hit_block[g - 1] == hit_block[g]) {
            const uint8_t *prev = pkt - a.pkt_bytes;
            bool same = true;
            for (int i = 0; i < r.dedupe_bytes; ++i) same = same && prev[i] == pkt[i];
            keep = !same;
        }
        a.keep[g] = keep ? 1 : 0;
    }
    const uint64_t m = __ballot(keep);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int w = 0; w < kValChunk / 64; ++w) s += wsum[w];
        a.chunk[blockIdx.x] = s;
    }
}

// Ordered compaction.  Every workgroup adds up the chunk counts itself (those before it = its base, all of them =
// the size of the validated list, which fixes the packed layout): a few loads per lane, cheaper than a scan kernel
// of its own between the two passes (each dispatch costs ~4.5 us on the stream).
__global__ __launch_bounds__(kValChunk) void k5_compact(const K5Args a)
{
    __shared__ uint32_t wcnt[kValChunk / 64], wred[2][kValChunk / 64];
    uint64_t total;
    const bool usable = k5_usable(a, total);
    if (!usable || total == 0) {
        if (blockIdx.x == 0 && threadIdx.x <= a.n_pre) { a.offs_val[threadIdx.x] = 0; a.h_offs_val[threadIdx.x] = 0; }
        return;
    }
    const uint64_t g0 = (uint64_t)blockIdx.x * kValChunk;
    if (g0 >= total) return;
    const uint32_t n = (uint32_t)((total + kValChunk - 1) / kValChunk);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t before = 0, all = 0;
    for (uint32_t i = threadIdx.x; i < n; i += kValChunk) {
        const uint32_t c = a.chunk[i];
        all += c;
        before += i < blockIdx.x ? c : 0u;
    }
    for (int d = 32; d; d >>= 1) { before += __shfl_down(before, d); all += __shfl_down(all, d); }
    const uint64_t g = g0 + threadIdx.x;
    const bool keep = g < total && a.keep[g] != 0;
    const uint64_t m = __ballot(keep);
    if (lane == 0) { wred[0][w] = before; wred[1][w] = all; wcnt[w] = (uint32_t)__popcll(m); }
    __syncthreads();
    uint64_t base = 0, kept = 0;
    uint32_t inblock = 0;
    for (uint32_t i = 0; i < kValChunk / 64; ++i) {
        base += wred[0][i];
        kept += wred[1][i];
        inblock += i < w ? wcnt[i] : 0u;
    }
    const uint64_t rank = base + inblock + (uint32_t)__popcll(m & ((1ull << lane) - 1));
    // new preamble offsets: the lane that holds a preamble's first hit knows how many hits survive before it;
    // preambles whose range starts at the end of the list (empty tail ranges, and the end marker) get the total
    if (g < total)
        for (uint32_t q = 0; q < a.n_pre; ++q)
            if (a.offs_pre[q] == g) { a.offs_val[q] = rank; a.h_offs_val[q] = rank; }
    if (blockIdx.x == 0 && threadIdx.x <= a.n_pre && a.offs_pre[threadIdx.x] >= total) {
        a.offs_val[threadIdx.x] = kept;
        a.h_offs_val[threadIdx.x] = kept;
    }
    if (!keep) return;
    const uint64_t *ib = reinterpret_cast<const uint64_t *>(a.in);
    const uint32_t *ii = reinterpret_cast<const uint32_t *>(a.in + total * 8);
    const uint8_t *ip = a.in + total * 12 + g * a.pkt_bytes;
    uint64_t *ob = reinterpret_cast<uint64_t *>(a.out);
    uint32_t *oi = reinterpret_cast<uint32_t *>(a.out + kept * 8);
    uint8_t *op = a.out + kept * 12 + rank * a.pkt_bytes;
    ob[rank] = ib[g];
    oi[rank] = ii[g];
    for (uint32_t i = 0; i < a.pkt_bytes; ++i) op[i] = ip[i];
}

}  // namespace amr
