// K2 / K2s / K3 -- preamble search, hit compaction and packet slicing on the
// tiled bitstream K1 wrote.
//
// Reference semantics (protocol/decode.go:255-328, Decoder.Search): call k
// reports every idx in [0, BlockSize) with
//     Quantized[idx + p*SymbolLength] == preamble[p]   for all p,
// ascending.  With pos = k*BlockSize + idx (counted from the first call of the
// batch) the bit tested for tap p is q[pos - PacketLength + p*SymbolLength],
// q = the stream of bit decisions, q[n] = 0 before the stream starts
// (decode.go:145).  For every legal -symbollength the byte prefilter of
// decode.go:268-294 selects exactly this set, so the
// search below evaluates the set directly, 32 positions per lane at a time:
//     M &= preamble[p] ? W_p : ~W_p,    W_p = the 32 stream bits starting at
//                                       n + p*SymbolLength (one funnel shift).
// All preambles share the windows W_p (every parser uses the same
// SymbolLength), so one pass serves scm, scm+, idm/netidm and r900 together.
//
// Work decomposition: one workgroup = one tile = 64 consecutive rows (reference
// blocks) of the tiled bitstream, staged in LDS together with row 0 of the
// next tile (a window never reaches further: (L-1)*SL < PreambleLength <=
// BlockSize).  Threads walk the tile in stream order, so hits leave the tile
// already sorted; a per-tile count plus an exclusive scan over tiles (K2s)
// gives every tile its slot in the final per-preamble arrays, which K3 fills
// (hit position + the sliced packet, decode.go:353-375).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amr {

constexpr int kMaxPre = 8;

struct SearchGeom {
    uint32_t block_size;     // BS
    uint32_t lg_block_size;
    uint32_t wpb;            // BS/32 words per row
    uint32_t lg_wpb;
    uint32_t symbol_length;  // SL (multiple of 16)
    uint32_t packet_length;  // PL (multiple of 64)
    uint32_t packet_symbols;
    uint32_t pkt_bytes;
    uint32_t n_pre;
    uint32_t max_pre_len;
    uint32_t pre_len[kMaxPre];
    uint64_t pre_bits[kMaxPre];  // bit p = preamble[p]
};

struct K2Args {
    const uint32_t *qt;    // tiled bitstream, tile 0 = history tile
    uint32_t *counts;      // [n_pre][n_tiles]
    uint32_t *staging;     // [n_tiles][n_pre][cap] tile-local positions (row*BS + bit), ascending
    uint32_t *overflow;    // set to 1 when a tile found more than cap hits for a preamble
    uint32_t n_tiles;      // tiles searched: ceil(n_blocks/64) + 1 (history tile first)
    uint32_t cap;
    int64_t n_lo, n_hi;    // valid positions: n_lo <= n < n_hi, n relative to batch sample 0
    SearchGeom g;
};

// 32 stream bits starting at bit `o` (word x = o>>5, shift sh = o&31) of row `l`; LDS tile is
// [word][65]: column 64 holds row 0 of the next tile, so a row overrun is "same word index in
// the next column".
__device__ __forceinline__ uint32_t k2_word(const uint32_t *lds, uint32_t x, uint32_t l, uint32_t wpb_mask, uint32_t lg_wpb)
{
    return lds[(x & wpb_mask) * 65 + l + (x >> lg_wpb)];
}

__global__ __launch_bounds__(256) void k2_search(const K2Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // wpb*65 words + 8 counters
    const SearchGeom &g = a.g;
    const uint32_t T = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t wpb = g.wpb, lg_wpb = g.lg_wpb, wpb_mask = wpb - 1;
    const uint32_t tile_words = 64u << lg_wpb;
    uint32_t *wave_tot = lds + wpb * 65;  // [4] wave totals for the block scan

    const uint32_t *src = a.qt + (size_t)T * tile_words;
    for (uint32_t i = tid; i < tile_words; i += 256) lds[(i >> 6) * 65 + (i & 63)] = src[i];
    for (uint32_t w = tid; w < wpb; w += 256) lds[w * 65 + 64] = src[tile_words + (w << 6)];
    __syncthreads();

    uint32_t running[kMaxPre];
#pragma unroll
    for (int p = 0; p < kMaxPre; ++p) running[p] = 0;

    const uint32_t iters = tile_words >> 8;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t i = it * 256 + tid;       // stream order inside the tile
        const uint32_t l = i >> lg_wpb;          // row
        const uint32_t w = i & wpb_mask;         // word in row
        // first position of this word, relative to batch sample 0 (tile 1 row 0 = batch block 0)
        const int64_t n0 = ((int64_t)T * 64 + l - 64) * (int64_t)g.block_size + (int64_t)w * 32;
        const bool valid = n0 >= a.n_lo && n0 < a.n_hi;
        uint32_t M[kMaxPre];
#pragma unroll
        for (int p = 0; p < kMaxPre; ++p) M[p] = (valid && p < (int)g.n_pre) ? 0xffffffffu : 0u;

        for (uint32_t p = 0; p < g.max_pre_len; ++p) {
            uint32_t any = 0;
#pragma unroll
            for (int q = 0; q < kMaxPre; ++q) any |= M[q];
            if (!__any(any != 0)) break;
            const uint32_t o = p * g.symbol_length;
            const uint32_t x = w + (o >> 5);
            uint32_t W = k2_word(lds, x, l, wpb_mask, lg_wpb);
            if (o & 31) {  // SL is a multiple of 16: the only non-zero shift is 16
                const uint32_t B = k2_word(lds, x + 1, l, wpb_mask, lg_wpb);
                W = (W << 16) | (B >> 16);
            }
#pragma unroll
            for (int q = 0; q < kMaxPre; ++q) {
                if (q < (int)g.n_pre && p < g.pre_len[q]) M[q] &= ((g.pre_bits[q] >> p) & 1) ? W : ~W;
            }
        }

        uint32_t any = 0;
#pragma unroll
        for (int q = 0; q < kMaxPre; ++q) any |= M[q];
        if (!__syncthreads_or(any != 0)) continue;

        // ordered emission: exclusive scan of popcounts in thread (= stream) order
#pragma unroll
        for (int q = 0; q < kMaxPre; ++q) {
            if (q >= (int)g.n_pre) break;
            uint32_t m = M[q];
            uint32_t c = __popc(m);
            uint32_t inc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t t = __shfl_up(inc, d);
                if ((tid & 63) >= (uint32_t)d) inc += t;
            }
            if ((tid & 63) == 63) wave_tot[tid >> 6] = inc;
            __syncthreads();
            uint32_t base = running[q];
            for (uint32_t v = 0; v < (tid >> 6); ++v) base += wave_tot[v];
            const uint32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
            uint32_t rank = base + inc - c;
            uint32_t *dst = a.staging + ((size_t)T * g.n_pre + q) * a.cap;
            while (m) {
                const uint32_t bit = __clz(m);
                if (rank < a.cap) dst[rank] = (l << g.lg_block_size) + (w << 5) + bit;
                rank++;
                m &= ~(0x80000000u >> bit);
            }
            running[q] += total;
            __syncthreads();
        }
    }

#pragma unroll
    for (int q = 0; q < kMaxPre; ++q) {
        if (q < (int)g.n_pre && tid == 0) {
            a.counts[q * a.n_tiles + T] = running[q] < a.cap ? running[q] : a.cap;
            if (running[q] > a.cap) atomicOr(a.overflow, 1u);
        }
    }
}

// K2s: exclusive scan of counts[n_pre*n_tiles] (preamble-major) -> offsets, plus
// per-preamble bases offs_pre[n_pre+1].  One workgroup; n is a few thousand.
struct ScanArgs {
    const uint32_t *counts;
    uint64_t *offsets;   // [n_pre*n_tiles]
    uint64_t *offs_pre;  // [n_pre+1]
    uint32_t n_tiles;
    uint32_t n_pre;
};

__global__ __launch_bounds__(1024) void k2s_scan(const ScanArgs a)
{
    __shared__ uint64_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t n = a.n_tiles * a.n_pre;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = tid * per, hi = lo + per < n ? lo + per : n;
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += a.counts[i];
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan, integers: exact
        uint64_t t = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    uint64_t run = tid ? part[tid - 1] : 0;
    for (uint32_t i = lo; i < hi; ++i) {
        a.offsets[i] = run;
        if (i % a.n_tiles == 0) a.offs_pre[i / a.n_tiles] = run;
        run += a.counts[i];
    }
    if (tid == 1023) a.offs_pre[a.n_pre] = part[1023];
}

// K3: move each tile's hits to their final slot and slice the packets.
struct K3Args {
    const uint32_t *qt;
    const uint32_t *counts;
    const uint64_t *offsets;
    const uint32_t *staging;
    uint64_t *hit_pos;     // [out_cap] pos = k_rel*BS + idx, k_rel = call index inside the batch
    uint8_t *pkt;          // [out_cap * pkt_bytes]
    uint64_t out_cap;
    uint32_t n_tiles;
    uint32_t cap;
    SearchGeom g;
};

// bit q[n] for n relative to batch sample 0 (n >= -64*BS): tiled row 64 + floor(n/BS)
__device__ __forceinline__ uint32_t k3_bit(const uint32_t *qt, int64_t n, const SearchGeom &g)
{
    const uint64_t u = (uint64_t)(n + ((int64_t)64 << g.lg_block_size));
    const uint64_t R = u >> g.lg_block_size;
    const uint32_t b = (uint32_t)u & (g.block_size - 1);
    const uint32_t word = qt[((R >> 6) << (6 + g.lg_wpb)) + ((b >> 5) << 6) + (R & 63)];
    return (word >> (31 - (b & 31))) & 1u;
}

__global__ __launch_bounds__(256) void k3_slice(const K3Args a)
{
    const SearchGeom &g = a.g;
    const uint32_t T = blockIdx.x, q = blockIdx.y;
    const uint32_t cnt = a.counts[q * a.n_tiles + T];
    if (cnt == 0) return;
    const uint64_t off = a.offsets[q * a.n_tiles + T];
    const uint32_t *src = a.staging + ((size_t)T * g.n_pre + q) * a.cap;
    const uint32_t work = cnt * g.pkt_bytes;
    for (uint32_t i = threadIdx.x; i < work; i += 256) {
        const uint32_t h = i / g.pkt_bytes, j = i % g.pkt_bytes;
        const uint64_t slot = off + h;
        if (slot >= a.out_cap) continue;
        const uint32_t local = src[h];
        // n relative to batch sample 0 of the first preamble bit
        const int64_t n = ((int64_t)T * 64 - 64) * (int64_t)g.block_size + local;
        if (j == 0) a.hit_pos[slot] = (uint64_t)(n + g.packet_length);
        uint32_t byte = 0;
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t p = j * 8 + k;
            if (p < g.packet_symbols) byte = (byte << 1) | k3_bit(a.qt, n + (int64_t)p * g.symbol_length, g);
        }
        a.pkt[slot * g.pkt_bytes + j] = (uint8_t)byte;
    }
}

// After a batch: the last `hr` rows (reference blocks) become the history rows 64-hr..63 of tile 0.
// Single workgroup, reads everything before writing anything (rows may move inside tile 0).
struct HistArgs {
    uint32_t *qt;
    uint32_t n_blocks;  // rows in the batch just processed
    uint32_t hr;        // history rows kept = ceil(PL/BS) (<= 63)
    uint32_t wpb, lg_wpb;
};

__global__ __launch_bounds__(1024) void k_hist_update(const HistArgs a)
{
    extern __shared__ uint32_t tmp[];  // hr*wpb words
    const uint32_t n = a.hr << a.lg_wpb;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const uint32_t j = i >> a.lg_wpb, w = i & (a.wpb - 1);
        // new history row j = stream row (n_blocks - hr + j) of the batch; negative -> old history
        const int64_t srow = (int64_t)64 + a.n_blocks - a.hr + j;  // tiled row index (tile 0 rows 0..63 = old history)
        tmp[i] = a.qt[((srow >> 6) << (6 + a.lg_wpb)) + (w << 6) + (srow & 63)];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const uint32_t j = i >> a.lg_wpb, w = i & (a.wpb - 1);
        a.qt[(w << 6) + (64 - a.hr + j)] = tmp[i];
    }
}

// Tests: tiled rows 64.. -> linear MSB-first byte stream (decode.go:259-265 packing).
__global__ void k_untile(const uint32_t *qt, uint32_t *out, uint32_t n_blocks, uint32_t lg_wpb)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = (uint64_t)n_blocks << lg_wpb;
    if (i >= n) return;
    const uint64_t R = 64 + (i >> lg_wpb);
    const uint32_t w = (uint32_t)i & ((1u << lg_wpb) - 1);
    const uint32_t v = qt[((R >> 6) << (6 + lg_wpb)) + (w << 6) + (R & 63)];
    out[i] = __builtin_bswap32(v);
}

}  // namespace amr
