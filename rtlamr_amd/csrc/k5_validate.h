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
// Three small kernels (the hit count is known on the device only): flag + per-chunk counts, scan of the chunk
// counts, ordered compaction into a second packed buffer of the same layout as K3's.
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
    const uint64_t *offs_pre;   // [n_pre+1] from k2s_scan
    uint64_t *offs_val;         // [n_pre+1] offsets into the validated list (device)
    uint64_t *h_offs_val;       // the same in pinned host memory
    uint32_t *chunk;            // [cap/kValChunk + 2]: survivors per chunk, then their exclusive scan
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

// exclusive scan of the chunk counts (one workgroup), the total behind the last chunk
__global__ __launch_bounds__(1024) void k5_scan(const K5Args a)
{
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x;
    uint64_t total;
    if (!k5_usable(a, total)) {
        if (tid <= a.n_pre) { a.offs_val[tid] = 0; a.h_offs_val[tid] = 0; }
        return;
    }
    const uint32_t n = (uint32_t)((total + kValChunk - 1) / kValChunk);
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += a.chunk[i];
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t t = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t c = a.chunk[i];
        a.chunk[i] = run;
        run += c;
    }
    const uint32_t kept = part[1023];
    if (tid == 0) a.chunk[n] = kept;
    // preambles whose first hit does not exist (empty tail ranges) start at the end of the validated list;
    // the others get their offset from the lane of k5_compact that holds their first hit
    if (tid <= a.n_pre && a.offs_pre[tid] >= total) { a.offs_val[tid] = kept; a.h_offs_val[tid] = kept; }
}

__global__ __launch_bounds__(kValChunk) void k5_compact(const K5Args a)
{
    __shared__ uint32_t wbase[kValChunk / 64];
    uint64_t total;
    if (!k5_usable(a, total)) return;
    const uint64_t g0 = (uint64_t)blockIdx.x * kValChunk;
    if (g0 >= total) return;
    const uint32_t n = (uint32_t)((total + kValChunk - 1) / kValChunk);
    const uint64_t kept = a.chunk[n];
    const uint64_t g = g0 + threadIdx.x;
    const bool keep = g < total && a.keep[g] != 0;
    const uint64_t m = __ballot(keep);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wbase[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t i = 0; i < w; ++i) before += wbase[i];
    const uint64_t rank = (uint64_t)a.chunk[blockIdx.x] + before + (uint32_t)__popcll(m & ((1ull << lane) - 1));
    if (g < total)
        for (uint32_t q = 0; q < a.n_pre; ++q)
            if (a.offs_pre[q] == g) { a.offs_val[q] = rank; a.h_offs_val[q] = rank; }
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
