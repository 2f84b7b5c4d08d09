// K2 fallbacks -- k2_search_fast (list-based, first generation) and k2_search_dense, for what k2_walk.h does not take:
// preambles shorter than 16 symbols, more than four preambles, rows under 16 words, and the re-run after a candidate
// list overflowed.  Same reference semantics, arguments and output as the walk kernel (k2_common.h).
#pragma once
#include "k2_common.h"

namespace amr {

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
            if (c) atomicAdd(&a.gcnt[(q * k2_groups(a.n_tiles) + (T >> 6)) * kGroupStride], c);
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
            if (c) atomicAdd(&a.gcnt[(q * k2_groups(a.n_tiles) + (T >> 6)) * kGroupStride], c);
            if (total[q] > a.cap) atomicOr(a.overflow, 1u);
        }
    }
    if (lane == 0 && list_n > (uint32_t)LCAP) atomicOr(a.overflow, 2u);
}


}  // namespace amr
