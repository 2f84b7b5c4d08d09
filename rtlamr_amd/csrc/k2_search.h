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
// already sorted; the per-tile counts (and their sums over groups of 64 tiles)
// give every tile its slot in the final per-preamble arrays, which K3 fills
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

// After a batch: the last `hr` rows (reference blocks) become the history rows 64-hr..63 of tile 0 of the next slot, the
// last HBA IQ bytes the carry, and the next slot's search words are reset -- the state the Go Decoder carries from call
// to call (decode.go:165-166).  One workgroup; reads everything before writing anything.
struct HistArgs {
    const uint32_t *qt;     // bitstream of the batch just processed (its tile 0 = the old history)
    uint32_t *qt_next;      // bitstream buffer the next batch will use: receives the new history tile
    uint32_t n_blocks;  // rows in the batch just processed
    uint32_t hr;        // history rows kept = ceil(PL/BS) (<= 63)
    uint32_t wpb, lg_wpb;
    // the other per-batch state, folded into this launch: the IQ halo of the next batch's block 0 (last HBA stream
    // bytes, decode.go:165) and the reset of the overflow word the next batch's search will use
    const uint8_t *carry_src;
    uint8_t *carry_dst;
    uint32_t carry_bytes;   // multiple of 16
    // blocks deferred to the next launch (amr_set_deferral): they follow the carry bytes in the stream and in the head
    // buffer (carry_src + carry_bytes -> carry_dst + carry_bytes), copied by `defer_wgs` extra workgroups of the launch
    uint32_t defer_bytes;   // multiple of 16
    uint32_t defer_wgs;
    uint32_t *ovf_next;
    uint32_t *gcnt_next;    // the group sums the next batch's K2 adds into
    uint32_t gcnt_words;
    // completion ticket of the batch, stored to pinned host memory by the last thread of this last kernel
    uint64_t *done_flag;
    uint64_t done_value;
    // ticket of the stream-A part of the batch (K1, search, this kernel), always published; done_flag may be null
    // when K3 and what follows it run later on the second stream and publish the batch ticket themselves
    uint64_t *adone_flag;
    // Pipelined callers: K3.. of the PREVIOUS batch run on the second stream next to this batch's search.  When they
    // take longer than the search, the next K1 launch (which needs every wave slot of the chip) has to wait for them:
    // this kernel, the last one in front of it, spins until the device word `wait_flag` reaches `wait_value`
    // (k_done of that batch) -- for at most ~2 ms, in case the host never launches them.
    const uint64_t *wait_flag;
    uint64_t wait_value;
};

__device__ __forceinline__ size_t qt_index_fwd(uint64_t R, uint32_t w, uint32_t lg_wpb)   // = qt_index, defined below
{
    return ((R >> 6) << (6 + lg_wpb)) + ((size_t)(w >> 2) << 8) + ((R & 63) << 2) + (w & 3);
}

// the work, by a workgroup of `nt` threads with hr * wpb words of LDS at tmp
__device__ __forceinline__ void hist_body(const HistArgs &a, uint32_t *tmp, uint32_t nt)
{
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < a.carry_bytes / 16; i += nt)
        reinterpret_cast<uint4 *>(a.carry_dst)[i] = reinterpret_cast<const uint4 *>(a.carry_src)[i];
    if (tid == nt - 1) *a.ovf_next = 0;
    for (uint32_t i = tid; i < a.gcnt_words; i += nt) a.gcnt_next[i] = 0;
    const uint32_t n = a.hr << a.lg_wpb;
    for (uint32_t i = tid; i < n; i += nt) {
        const uint32_t j = i >> a.lg_wpb, w = i & (a.wpb - 1);
        // new history row j = stream row (n_blocks - hr + j) of the batch; negative -> old history
        const int64_t srow = (int64_t)64 + a.n_blocks - a.hr + j;  // tiled row index (tile 0 rows 0..63 = old history)
        tmp[i] = a.qt[qt_index_fwd((uint64_t)srow, w, a.lg_wpb)];
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += nt) {
        const uint32_t j = i >> a.lg_wpb, w = i & (a.wpb - 1);
        a.qt_next[qt_index_fwd(64 - a.hr + j, w, a.lg_wpb)] = tmp[i];
    }
    __syncthreads();
}

// slice `part` of `parts` of the deferred blocks, by a workgroup of `nt` threads
__device__ __forceinline__ void defer_copy_body(const HistArgs &a, uint32_t part, uint32_t nt)
{
    const uint32_t n16 = a.defer_bytes / 16, per = (n16 + a.defer_wgs - 1) / a.defer_wgs;
    const uint32_t lo = part * per, hi = lo + per < n16 ? lo + per : n16;
    const uint4 *src = reinterpret_cast<const uint4 *>(a.carry_src + a.carry_bytes);
    uint4 *dst = reinterpret_cast<uint4 *>(a.carry_dst + a.carry_bytes);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += nt) dst[i] = src[i];
}

// the tickets, by one thread, once everything of the batch on this stream has completed
__device__ __forceinline__ void hist_publish(const HistArgs &a)
{
    if (a.adone_flag) __hip_atomic_store(a.adone_flag, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.done_flag) __hip_atomic_store(a.done_flag, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.wait_flag) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        while (__hip_atomic_load(a.wait_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < a.wait_value &&
               __builtin_amdgcn_s_memrealtime() - t0 < 200000ull)
            __builtin_amdgcn_s_sleep(32);
    }
}

__global__ __launch_bounds__(1024) void k_hist_update(const HistArgs a)
{
    extern __shared__ uint32_t hist_tmp[];  // hr*wpb words
    if (blockIdx.x) { defer_copy_body(a, blockIdx.x - 1, 1024); return; }
    hist_body(a, hist_tmp, 1024);
    // every earlier kernel of the batch has completed (same stream); the host polls these words
    if (threadIdx.x == 0) hist_publish(a);
}

struct K2Args {
    const uint32_t *qt;    // tiled bitstream, tile 0 = history tile
    uint32_t *counts;      // [n_pre][n_tiles]
    uint32_t *gcnt;        // [n_pre][n_groups] sums of counts over groups of 64 tiles (atomicAdd; zero before K2 runs)
    uint32_t *staging;     // [n_tiles][n_pre][cap] tile-local positions (row*BS + bit), ascending
    uint32_t *overflow;    // set to 1 when a tile found more than cap hits for a preamble
    uint32_t n_tiles;      // tiles searched: ceil(n_blocks/64) + 1 (history tile first)
    uint32_t cap;
    int64_t n_lo, n_hi;    // valid positions: n_lo <= n < n_hi, n relative to batch sample 0
    unsigned long long *dbg;   // harness builds only (AMR_K2S_DBG): 16 words of timestamps per workgroup, or null
    // pinned host word that receives `started_value` when the search starts, i.e. when everything before it on the
    // stream (this batch's K1) has finished: the host then launches the previous batch's K3 on the second stream
    uint64_t *started;
    uint64_t started_value;
    // pipelined callers: the state update rides along as one more workgroup (tile index n_tiles; the hist.defer_wgs
    // workgroups behind it copy the deferred blocks) instead of a 5 us kernel of its own behind the search.  It carries
    // no completion ticket (the search is still running
    // when it is done; a ticket from inside the kernel would also need every workgroup to release its writes, an L2
    // write-back each): the host takes "the next search has started" or "the stream is idle" as the signal instead.
    uint32_t do_hist;
    HistArgs hist;
    SearchGeom g;
};

// Workgroups behind the last tile of a search launch: the folded state update and the deferred-block copies.
// Returns true when this workgroup was one of them (and is done).
__device__ __forceinline__ bool k2_extra_workgroup(const K2Args &a, uint32_t T, uint32_t *lds, uint32_t nt)
{
    if (T < a.n_tiles) return false;
    if (a.do_hist) {
        if (T == a.n_tiles) {
            hist_body(a.hist, lds, nt);
            if (threadIdx.x == 0) hist_publish(a.hist);   // no tickets here (the search is still running): only the wait
        } else if (T - a.n_tiles - 1 < a.hist.defer_wgs) {
            defer_copy_body(a.hist, T - a.n_tiles - 1, nt);
        }
    }
    return true;
}

__device__ __forceinline__ void k2_announce(const K2Args &a)
{
    if (a.started && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(a.started, a.started_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Index of word w (32 decisions) of tiled row R (row 64 + b = batch block b; rows 0..63 = history tile) in the
// "tiled4" bitstream K1 writes: per 64-row tile, 4-word chunks, a row's chunk = 16 contiguous bytes.
__device__ __forceinline__ size_t qt_index(uint64_t R, uint32_t w, uint32_t lg_wpb)
{
    return ((R >> 6) << (6 + lg_wpb)) + ((size_t)(w >> 2) << 8) + ((R & 63) << 2) + (w & 3);
}

// 32 stream bits starting at bit `o` (word x = o>>5, shift sh = o&31) of row `l`; LDS tile is
// [word][65]: column 64 holds row 0 of the next tile, so a row overrun is "same word index in
// the next column".
__device__ __forceinline__ uint32_t k2_word(const uint32_t *lds, uint32_t x, uint32_t l, uint32_t wpb_mask, uint32_t lg_wpb)
{
    return lds[(x & wpb_mask) * 65 + l + (x >> lg_wpb)];
}

// The per-(preamble, tile) hit counts are also summed per group of 64 tiles, so that K3 finds the slot of a list in
// the packed result from <= n_pre * n_groups + 63 values instead of a scan over all of them.
__host__ __device__ __forceinline__ uint32_t k2_groups(uint32_t n_tiles) { return (n_tiles + 63) >> 6; }

__global__ __launch_bounds__(256) void k2_search_dense(const K2Args a)
{
    k2_announce(a);
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // wpb*65 words + 8 counters
    const SearchGeom &g = a.g;
    const uint32_t T = blockIdx.x;
    if (k2_extra_workgroup(a, T, lds, 256)) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t wpb = g.wpb, lg_wpb = g.lg_wpb, wpb_mask = wpb - 1;
    const uint32_t tile_words = 64u << lg_wpb;
    uint32_t *wave_tot = lds + wpb * 65;  // [4] wave totals for the block scan

    const uint32_t *src = a.qt + (size_t)T * tile_words;
    for (uint32_t i = tid; i < tile_words; i += 256) lds[((i >> 8) * 4 + (i & 3)) * 65 + ((i >> 2) & 63)] = src[i];
    for (uint32_t w = tid; w < wpb; w += 256) lds[w * 65 + 64] = src[tile_words + ((w >> 2) << 8) + (w & 3)];
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
            const uint32_t c = running[q] < a.cap ? running[q] : a.cap;
            a.counts[q * a.n_tiles + T] = c;
            if (c) atomicAdd(&a.gcnt[q * k2_groups(a.n_tiles) + (T >> 6)], c);
            if (running[q] > a.cap) atomicOr(a.overflow, 1u);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k2_search_fast<NPRE>: the production search.  Same result as k2_search_dense, organised for the
// common case that hits are sparse:
//   * lane = row (reference block) of the tile, the unit K1 computed, so the tile is laid out in LDS as
//     [word][65] (column 64 = row 0 of the next tile) and a lane's window never leaves its column pair;
//   * stage 1: a wave handles 8 consecutive words (256 positions; 4 with the 4-wave variant) of all 64 rows per step and applies
//     only the first D taps (D = 9 + log2 NPRE), without any early-out test: in noise 2^-D of the
//     positions survive.  The per-tap scalar work (offset, shift, preamble bit) is shared by 4 words,
//     the 4-5 LDS reads use immediate offsets and are issued one tap ahead of their use;
//   * the (rare) non-zero masks go to a small per-wave list; stage 2 finishes the remaining taps on
//     the list entries, one entry per lane, and compacts the list in place (order preserved);
//   * a per-tile exclusive scan over (row, wave) popcounts then gives every surviving entry its rank,
//     and each entry is emitted by 32 lanes at once (lane b = bit b), in stream order.
// If a wave's list overflows (pathological input), bit 1 of *overflow is set and the host re-runs the
// tile set with k2_search_dense.
constexpr int kListCap = 448;  // (key, mask) entries per wave

#ifndef AMR_K2_DEPTH
#define AMR_K2_DEPTH 9
#endif
__device__ __forceinline__ uint32_t k2_depth(uint32_t npre) { return (uint32_t)AMR_K2_DEPTH + (npre > 2 ? 2u : npre > 1 ? 1u : 0u); }

// NWV waves share one tile (8 when a row has >= 64 words, so that each wave still gets a whole 8-word step: 24 waves per CU hide the LDS latency of the tap loop,
// which is what bounds this kernel; 4 for the 512-sample blocks of chip length 8).
template <int NPRE, int NWV, int JW>
__global__ __launch_bounds__(64 * NWV) void k2_search_fast(const K2Args a)
{
    k2_announce(a);
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t T = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (k2_extra_workgroup(a, T, lds, 64 * NWV)) return;
    // the wave index is wave-uniform, but hipcc cannot know that of tid >> 6: without readfirstlane the whole
    // window addressing below is done per lane in VALU and its branches become exec-masked double execution
    const uint32_t v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wpb = a.g.wpb, lg_wpb = a.g.lg_wpb, wpb_mask = wpb - 1;
    const uint32_t lg_bs = a.g.lg_block_size;
    const uint32_t SL = a.g.symbol_length, maxL = a.g.max_pre_len;
    const uint32_t tile_words = 64u << lg_wpb;
    uint32_t *tile = lds;                              // [wpb][65]
    constexpr int NT = 64 * NWV;                       // threads
    // JW = words per lane per stage-1 step: the per-tap scalar work is shared by JW words (8; 4 for 512-sample blocks)
    constexpr int LCAP = kListCap * 4 / NWV;           // list entries per wave: the candidates split with the words
    uint32_t *lists = tile + wpb * 65;                 // [NWV][LCAP][2]
    uint32_t *cnts = lists + 4 * kListCap * 2;         // [NPRE][NT], index row*NWV+wave
    uint32_t *bases = cnts + NPRE * NT;                // [NPRE][NT]
    uint32_t *wtot = bases + NPRE * NT;                // [NWV]

    uint64_t pbits[NPRE];
    uint32_t plen[NPRE];
#pragma unroll
    for (int q = 0; q < NPRE; ++q) { pbits[q] = a.g.pre_bits[q]; plen[q] = a.g.pre_len[q]; }
    const uint32_t D = k2_depth(NPRE) < maxL ? k2_depth(NPRE) : maxL;   // stage-1 depth

    // ---- stage the tile: 16 bytes per lane per load (4 words of one row), transposed into [word][row] ----
    {
        const uint4 *src4 = reinterpret_cast<const uint4 *>(a.qt + (size_t)T * tile_words);
        for (uint32_t i = tid; i < tile_words / 4; i += NT) {
            const uint4 x = src4[i];                      // words 4c..4c+3 of row l
            const uint32_t c = i >> 6, l = i & 63;
            uint32_t *d = tile + (c * 4) * 65 + l;
            d[0] = x.x; d[65] = x.y; d[130] = x.z; d[195] = x.w;
        }
        const uint32_t *nxt = a.qt + (size_t)(T + 1) * tile_words;
        for (uint32_t w = tid; w < wpb; w += NT) tile[w * 65 + 64] = nxt[((w >> 2) << 8) + (w & 3)];
#pragma unroll
        for (int q = 0; q < NPRE; ++q) cnts[q * NT + tid] = 0;
    }
    __syncthreads();

    // ---- valid word range of this lane's row: n_lo <= R*BS + 32w < n_hi ----
    const int64_t rowbase = ((int64_t)T * 64 + lane - 64) << lg_bs;
    int64_t lo64 = (a.n_lo - rowbase) >> 5, hi64 = (a.n_hi - rowbase) >> 5;
    const uint32_t w_lo = (uint32_t)(lo64 < 0 ? 0 : lo64 > (int64_t)wpb ? wpb : lo64);
    const uint32_t w_hi = (uint32_t)(hi64 < 0 ? 0 : hi64 > (int64_t)wpb ? wpb : hi64);

    uint32_t list_n = 0;                                // wave-uniform
    uint32_t *mylist = lists + v * (LCAP * 2);
    const uint32_t wq = wpb / NWV;                      // words per wave
    const uint32_t *lane_tile = tile + lane;

    // window words A[0..4] of tap p for the step that starts at word w0 (uniform addressing)
    auto load_tap = [&](uint32_t w0, uint32_t p, uint32_t (&A)[JW + 1]) {
        const uint32_t o = p * SL;
        const uint32_t x0 = w0 + (o >> 5);
        const uint32_t xm = x0 & wpb_mask;
        if (xm + JW + 1 <= wpb) {                        // all words in one row: immediate offsets
            const uint32_t *src = lane_tile + xm * 65 + (x0 >> lg_wpb);
#pragma unroll
            for (int j = 0; j < JW + 1; ++j) A[j] = src[j * 65];
        } else {                                         // the window crosses into the next row
#pragma unroll
            for (int j = 0; j < JW + 1; ++j) {
                const uint32_t x = x0 + j;
                A[j] = lane_tile[(x & wpb_mask) * 65 + (x >> lg_wpb)];
            }
        }
    };

    // ---- stage 1 ----
#ifndef AMR_K2_DIAG
#define AMR_K2_DIAG 0   // developer diagnostics: 1 = no search at all (staging, barriers, scan only), 2 = stage 1 only
#endif
    for (uint32_t c = 0; c < (AMR_K2_DIAG == 1 ? 0u : wq / JW); ++c) {
        const uint32_t w0 = v * wq + JW * c;
        uint32_t M[NPRE][JW];
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            const uint32_t ok = (w0 + j >= w_lo && w0 + j < w_hi) ? 0xffffffffu : 0u;
#pragma unroll
            for (int q = 0; q < NPRE; ++q) M[q][j] = ok;
        }
        // one tap: W = the 4 windows (plain or 16-bit funnel shift, a wave-uniform choice), M &= W ^ inv
        auto apply_tap = [&](uint32_t p, const uint32_t (&A)[JW + 1]) {
            uint32_t W[JW];
            if ((p * SL) & 31) {                         // SL multiple of 16: shift is 0 or 16
#pragma unroll
                for (int j = 0; j < JW; ++j) W[j] = __builtin_amdgcn_alignbit(A[j], A[j + 1], 16);
            } else {
#pragma unroll
                for (int j = 0; j < JW; ++j) W[j] = A[j];
            }
#pragma unroll
            for (int q = 0; q < NPRE; ++q) {
                if (p < plen[q]) {
                    const uint32_t inv = ((pbits[q] >> p) & 1) ? 0u : 0xffffffffu;
#pragma unroll
                    for (int j = 0; j < JW; ++j) M[q][j] &= W[j] ^ inv;
                }
            }
        };
        // two taps per iteration on ping-pong buffers: the loads of tap p+1 are in flight while tap p is applied
        uint32_t A0[JW + 1], A1[JW + 1];
        load_tap(w0, 0, A0);
        for (uint32_t p = 0; p < D; p += 2) {
            if (p + 1 < D) load_tap(w0, p + 1, A1);
            apply_tap(p, A0);
            if (p + 1 < D) {
                if (p + 2 < D) load_tap(w0, p + 2, A0);
                apply_tap(p + 1, A1);
            }
        }
        // record the (rare) non-zero masks
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
#pragma unroll
            for (int j = 0; j < JW; ++j) {
                const uint32_t m = M[q][j];
                const uint64_t b = __ballot(m != 0);
                if (b) {
                    const uint32_t idx = list_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
                    if (m != 0 && idx < (uint32_t)LCAP) {
                        mylist[idx * 2] = ((uint32_t)q << 16) | (lane << 8) | (w0 + j);
                        mylist[idx * 2 + 1] = m;
                    }
                    list_n += __popcll(b);
                }
            }
        }
    }

    // ---- stage 2: remaining taps on the list entries (one per lane), compaction in place ----
    const uint32_t n_cand = AMR_K2_DIAG == 2 ? 0u : (list_n < (uint32_t)LCAP ? list_n : (uint32_t)LCAP);
    uint32_t n_keep = 0;                                // wave-uniform
    for (uint32_t e0 = 0; e0 < n_cand; e0 += 64) {
        const uint32_t e = e0 + lane;
        uint32_t key = 0, m = 0;
        if (e < n_cand) { key = mylist[e * 2]; m = mylist[e * 2 + 1]; }
        const uint32_t q = key >> 16, l = (key >> 8) & 63, w = key & 0xff;
        uint64_t pb = pbits[0];
        uint32_t pl = plen[0];
#pragma unroll
        for (int qq = 1; qq < NPRE; ++qq)
            if (q == (uint32_t)qq) { pb = pbits[qq]; pl = plen[qq]; }
        for (uint32_t p = D; p < maxL; ++p) {
            if (!__any(m != 0)) break;
            const uint32_t o = p * SL;
            const uint32_t x = w + (o >> 5);
            uint32_t Wd = tile[(x & wpb_mask) * 65 + l + (x >> lg_wpb)];
            if (o & 31) {
                const uint32_t B = tile[((x + 1) & wpb_mask) * 65 + l + ((x + 1) >> lg_wpb)];
                Wd = __builtin_amdgcn_alignbit(Wd, B, 16);
            }
            if (p < pl) m &= ((pb >> p) & 1) ? Wd : ~Wd;
        }
        const uint64_t b = __ballot(m != 0);
        if (m != 0) {   // survivors move to the front, order preserved (slot <= e, earlier chunks already read)
            const uint32_t slot = n_keep + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
            mylist[slot * 2] = key;
            mylist[slot * 2 + 1] = m;
            atomicAdd(&cnts[q * NT + l * NWV + v], __popc(m));
        }
        n_keep += __popcll(b);
    }
    __syncthreads();

    // ---- ranks: exclusive scan over (row, wave) in stream order, per preamble ----
    uint32_t total[NPRE];
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
        const uint32_t val = cnts[q * NT + tid];
        uint32_t inc = val;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d);
            if (lane >= (uint32_t)d) inc += t;
        }
        if (lane == 63) wtot[v] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t u = 0; u < v; ++u) base += wtot[u];
        uint32_t tot = 0;
#pragma unroll
        for (int u = 0; u < NWV; ++u) tot += wtot[u];
        total[q] = tot;
        bases[q * NT + tid] = base + inc - val;
        __syncthreads();
    }

    // ---- emit: every surviving entry by 32 lanes at once, lane b = bit b (MSB first = stream order) ----
    uint32_t run[NPRE];
#pragma unroll
    for (int q = 0; q < NPRE; ++q) run[q] = 0;
    for (uint32_t e = 0; e < n_keep; ++e) {
        const uint32_t key = __builtin_amdgcn_readfirstlane(mylist[e * 2]);
        const uint32_t m = __builtin_amdgcn_readfirstlane(mylist[e * 2 + 1]);
        const uint32_t q = key >> 16, l = (key >> 8) & 63, w = key & 0xff;
        uint32_t r = 0;
#pragma unroll
        for (int qq = 0; qq < NPRE; ++qq)
            if (q == (uint32_t)qq) r = __builtin_amdgcn_readlane(run[qq], l);
        const uint32_t base = bases[q * NT + l * NWV + v] + r;
        if (lane < 32 && ((m >> (31 - lane)) & 1)) {
            const uint32_t before = lane ? __popc(m >> (32 - lane)) : 0;
            const uint32_t rank = base + before;
            if (rank < a.cap) a.staging[((size_t)T * NPRE + q) * a.cap + rank] = (l << lg_bs) + (w << 5) + lane;
        }
        const uint32_t add = (lane == l) ? __popc(m) : 0;
#pragma unroll
        for (int qq = 0; qq < NPRE; ++qq)
            if (q == (uint32_t)qq) run[qq] += add;
    }

    if (tid == 0) {
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const uint32_t c = total[q] < a.cap ? total[q] : a.cap;
            a.counts[q * a.n_tiles + T] = c;
            if (c) atomicAdd(&a.gcnt[q * k2_groups(a.n_tiles) + (T >> 6)], c);
            if (total[q] > a.cap) atomicOr(a.overflow, 1u);
        }
    }
    if (lane == 0 && list_n > (uint32_t)LCAP) atomicOr(a.overflow, 2u);
}

inline size_t k2_fast_lds_bytes(uint32_t wpb, int npre, int nwv)
{
    return ((size_t)wpb * 65 + 4 * kListCap * 2 + 2 * (size_t)npre * 64 * nwv + 8) * 4;
}

// K3: move each tile's hits to their final slot and slice the packets.
struct K3Args {
    const uint32_t *qt;
    const uint32_t *counts;     // [n_pre][n_tiles] from K2
    const uint32_t *gcnt;       // [n_pre][n_groups] from K2
    const uint32_t *staging;
    // packed result [hit_block u64 x n | hit_idx u32 x n | pkt x n], n = total hits:
    //   hit_block = block_base + (pos >> lg BS), pos = n + PacketLength;  hit_idx = pos & (BS-1)  (Data.Idx, decode.go:371)
    uint8_t *out;
    uint64_t *offs_pre;         // [n_pre+1] per-preamble bases, written here (K4/K5 and device-side consumers read them)
    // what the host needs to size / accept the result, written straight into pinned host memory (no D2H copy on
    // the compute stream): the per-preamble bases and K2's overflow word
    uint64_t *h_offs_pre;       // [n_pre+1]
    uint32_t *h_overflow;
    uint64_t block_base;        // call index of the first block of the batch
    uint64_t out_cap;           // hits the buffer holds
    const uint32_t *overflow;   // K2's overflow word: non-zero = the host will grow a capacity and search again
    uint32_t n_tiles;
    uint32_t cap;
    SearchGeom g;
};

// K3 slices by bitstream word, not by hit.  The hits of a real packet (and most noise hits' neighbours) come in runs of
// adjacent positions, so slicing hit by hit (round 1: one lane = 32 symbols of one hit) read every bitstream word ~20
// times and spent ~13 VALU operations per (hit, symbol).  Here the unit of work is a bitstream WORD that holds hits: for symbol p the 32 positions of the word
// need the 32 stream bits starting at word*32 + p*SL -- one window, one or two word loads (SL is a multiple of 16) --
// and the packets of all 32 positions are the columns of the bit matrix [symbol][position].  A wave takes 64 symbols
// at a time, lane = symbol (two 32 x 32 blocks), transposes the blocks in five exchange steps (ds_swizzle, no LDS
// memory), after which lane c of a block holds 32 consecutive packet bits of position 31-c: one dword of that packet,
// already in the byte order of Decoder.Slice (decode.go:363-366) because the symbols were dealt to the lanes
// bit-reversed inside every byte.  Positions that are hits store their dword, the others are dropped.
// Input: positions in the staging slots, ascending; output: the packed result (K3Args).
// (Tried and dropped: one workgroup per tile that first stages the tile's 64 rows in LDS -- LDS-DMA, rows XOR-swizzled
// against bank conflicts -- and takes the windows from there: 64 vs 54 us of search per 1 GiB with scm, 1.05 vs 0.67 ms
// per 4 GiB with four preambles.  The preambles of a tile run one after the other, half as many workgroups fit a CU,
// and the staging is one more latency in front of a kernel that is made of latencies.)
__device__ __forceinline__ uint32_t k3_transpose32(uint32_t x, uint32_t lane)
{
#define K3_TSTEP(S, M)                                                                                                \
    {                                                                                                                 \
        const uint32_t y = (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, ((S) << 10) | 0x1f);   /* lane ^ S */       \
        x = (lane & (S)) ? ((x & ~(M)) | ((y >> (S)) & (M))) : ((x & (M)) | ((y << (S)) & ~(M)));                     \
    }
    K3_TSTEP(16, 0x0000ffffu) K3_TSTEP(8, 0x00ff00ffu) K3_TSTEP(4, 0x0f0f0f0fu) K3_TSTEP(2, 0x33333333u) K3_TSTEP(1, 0x55555555u)
#undef K3_TSTEP
    return x;
}

constexpr int kK3Batch = 4;    // words (entries) a wave works on together: their bitstream loads are in flight at once

__global__ __launch_bounds__(256) void k3_slice_words(const K3Args a)
{
    const SearchGeom &g = a.g;
    const uint32_t T = blockIdx.x, q = blockIdx.y;
    __shared__ uint64_t red[2][4];
    __shared__ uint32_t tab[4][kK3Batch][32];   // per wave and entry: staging index of the hit at bit b of the word, or ~0
    // Slot of this (tile, preamble) list in the packed result = the hits of all lists before it (preamble-major), and the
    // layout needs the grand total.  No scan kernel between K2 and K3 (a dispatch costs the stream ~5 us): K2 left sums
    // over groups of 64 tiles, so a workgroup adds up the group sums before its group, the <= 63 counts before it inside
    // the group, and all group sums for the total -- one load per lane, all in flight before the first use.  The
    // workgroups of tile 0 publish the per-preamble bases, (0,0) also the total and K2's overflow word.
    const uint32_t n_groups = k2_groups(a.n_tiles), my_g = q * n_groups + (T >> 6);
    const uint32_t cnt = a.counts[q * a.n_tiles + T];
    uint64_t before = 0, all = 0;
    for (uint32_t i = threadIdx.x; i < g.n_pre * n_groups; i += 256) {
        const uint32_t c = a.gcnt[i];
        all += c;
        before += i < my_g ? c : 0u;
    }
    if (threadIdx.x < (T & 63)) before += a.counts[q * a.n_tiles + (T & ~63u) + threadIdx.x];
    const uint32_t ovf = *a.overflow;
    if (cnt == 0 && T != 0) return;
    for (int d = 32; d; d >>= 1) {
        before += __shfl_down((unsigned long long)before, d);
        all += __shfl_down((unsigned long long)all, d);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = before; red[1][threadIdx.x >> 6] = all; }
    __syncthreads();
    const uint64_t off = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const uint64_t total = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (T == 0 && threadIdx.x == 0) {
        a.offs_pre[q] = off;
        a.h_offs_pre[q] = off;
        if (q == 0) { a.offs_pre[g.n_pre] = total; a.h_offs_pre[g.n_pre] = total; *a.h_overflow = ovf; }
    }
    if (ovf || cnt == 0) return;
    if (total > a.out_cap) return;   // the host grows the buffer and searches again
    uint64_t *hit_block = reinterpret_cast<uint64_t *>(a.out);
    uint32_t *hit_idx = reinterpret_cast<uint32_t *>(a.out + total * 8);
    uint8_t *pkt = a.out + total * 12;
    const uint32_t *src = a.staging + ((size_t)T * g.n_pre + q) * a.cap;
    const uint32_t *__restrict__ tbase = a.qt + ((size_t)T << (6 + g.lg_wpb));
    const uint32_t lg_bs = g.lg_block_size, bs_mask = g.block_size - 1, lg_tw = 6 + g.lg_wpb;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l32 = lane & 31, half = lane >> 5;
    const uint32_t PS = g.packet_symbols, SL = g.symbol_length, PB = g.pkt_bytes;
    const bool dword_ok = (PB & 3) == 0 && (PS & 7) == 0;
    // symbol offset of this lane inside a 64-symbol step: the 32 lanes of a block take the symbols bit-reversed
    // within every byte, so that bit i of the transposed dword is the symbol Decoder.Slice puts into bit i
    const uint32_t sym_lane = half * 32 + ((l32 & ~7u) | (7u - (l32 & 7u)));
    const uint32_t bad = 64u << lg_bs;                 // defensive: never index the bitstream with a bad position
    auto word_at = [&](uint32_t v) {                   // bitstream word holding bit v (counted from row 0 of tile T)
        const uint32_t row = v >> lg_bs, w = (v & bs_mask) >> 5;
        return tbase[((row >> 6) << lg_tw) + ((w >> 2) << 8) + ((row & 63) << 2) + (w & 3)];
    };
    for (uint32_t i0 = wv * 64; i0 < cnt; i0 += 256) {
        const uint32_t i = i0 + lane;
        const bool have = i < cnt;
        const uint32_t local = have ? src[i] : 0xffffffffu;
        const bool ok = have && local < bad;
        if (ok) {
            const int64_t n = ((int64_t)T * 64 - 64) * (int64_t)g.block_size + local;
            const uint64_t pos = (uint64_t)(n + g.packet_length);
            hit_block[off + i] = a.block_base + (pos >> lg_bs);
            hit_idx[off + i] = (uint32_t)pos & bs_mask;
        }
        const uint32_t key = ok ? local >> 5 : 0xffffffffu;
        const uint32_t prev = __shfl_up(key, 1);
        uint64_t leaders = __ballot(ok && (lane == 0 || key != prev));
        while (leaders) {
            uint32_t v0[kK3Batch], slot[kK3Batch];
            int nb = 0;
#pragma unroll
            for (int e = 0; e < kK3Batch; ++e) {
                v0[e] = 0; slot[e] = 0xffffffffu;
                if (leaders) {                                         // wave-uniform
                    const uint32_t L = (uint32_t)__ffsll((unsigned long long)leaders) - 1;
                    leaders &= leaders - 1;
                    const uint32_t key_s = __builtin_amdgcn_readlane(key, L);
                    if (lane < 32) tab[wv][e][lane] = 0xffffffffu;
                    if (ok && key == key_s) tab[wv][e][local & 31] = i;   // same wave: LDS operations execute in order
                    slot[e] = tab[wv][e][31 - l32];                    // lane c of a block ends up with position 31-c
                    v0[e] = key_s << 5;
                    nb = e + 1;
                }
            }
            for (uint32_t p0 = 0; p0 < PS; p0 += 128) {               // two 64-symbol steps of up to four words per round
                uint32_t A[kK3Batch][2], B[kK3Batch][2];
#pragma unroll
                for (int e = 0; e < kK3Batch; ++e)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        A[e][k] = 0; B[e][k] = 0;
                        if (e < nb && p0 + 64 * k < PS) {
                            const uint32_t sy = p0 + 64 * k + sym_lane;
                            const uint32_t v = v0[e] + (sy < PS ? sy : PS - 1) * SL;
                            A[e][k] = word_at(v);
                            B[e][k] = word_at(v + 32);
                        }
                    }
#pragma unroll
                for (int e = 0; e < kK3Batch; ++e)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        if (e >= nb || p0 + 64 * k >= PS) continue;
                        const uint32_t sy = p0 + 64 * k + sym_lane;
                        const uint32_t W = ((sy < PS ? sy : PS - 1) * SL & 16) ? __builtin_amdgcn_alignbit(A[e][k], B[e][k], 16) : A[e][k];
                        const uint32_t Y = k3_transpose32(W, lane);
                        const uint32_t b0 = (p0 + 64 * k) / 8 + half * 4;   // first packet byte of this lane's dword
                        if (slot[e] != 0xffffffffu && b0 < PB) {
                            uint8_t *out = pkt + (off + slot[e]) * (uint64_t)PB;
                            if (b0 + 4 <= PB && dword_ok) {
                                *reinterpret_cast<uint32_t *>(out + b0) = Y;
                            } else {
#pragma unroll
                                for (uint32_t j = 0; j < 4; ++j) {
                                    const uint32_t bj = b0 + j;
                                    if (bj < PB) {
                                        uint32_t byte = (Y >> (8 * j)) & 0xffu;
                                        const uint32_t valid = PS - bj * 8;
                                        if (valid < 8) byte >>= (8 - valid);   // PacketSymbols % 8 != 0: right-aligned like Go's shift-in
                                        out[bj] = (uint8_t)byte;
                                    }
                                }
                            }
                        }
                    }
            }
        }
    }
}

// last kernel of a batch whose K3 (K4, K5) ran on the second stream: publishes the batch ticket
__global__ void k_done(uint64_t *flag, uint64_t value, uint64_t *dev_flag)
{
    __hip_atomic_store(dev_flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // k_hist_update of the next batch waits here
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Tests: tiled rows 64.. -> linear MSB-first byte stream (decode.go:259-265 packing).
__global__ void k_untile(const uint32_t *qt, uint32_t *out, uint32_t n_blocks, uint32_t lg_wpb)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = (uint64_t)n_blocks << lg_wpb;
    if (i >= n) return;
    const uint64_t R = 64 + (i >> lg_wpb);
    const uint32_t w = (uint32_t)i & ((1u << lg_wpb) - 1);
    const uint32_t v = qt[qt_index(R, w, lg_wpb)];
    out[i] = __builtin_bswap32(v);
}

}  // namespace amr
