// K2 (second generation) -- the preamble search of k2_search.h (Decoder.Search, protocol/decode.go:255-328: every
// idx with Quantized[idx + p*SymbolLength] == preamble[p] for all p, ascending), reorganised around what bounded the
// first kernel (47 us per 64 MiB of bitstream: a transposing copy into LDS behind a barrier, then one LDS read per tap
// and word with computed addresses -- 47 % of the wave time parked):
//   * the 64-row tile goes into LDS with LDS-DMA (global_load_lds_dwordx4), 1 KiB per instruction, in the layout the
//     bitstream has in memory ("tiled4": chunk c = words 4c..4c+3 of all 64 rows, 16 bytes per row): no VGPR round trip,
//     no transpose.  Every chunk gets a 65th row slot that holds row 0 of the next tile, so "the row behind row l" is
//     row slot l+1 for every lane;
//   * a wave sweeps 32 words (8 groups of 4) of all 64 rows, lane = row = one reference block as before, and keeps the
//     words its first D taps can reach in a REGISTER ring filled with ds_read_b128 (consecutive lanes read consecutive
//     16 bytes: conflict-free).  SymbolLength and D are template parameters, so every window word of every tap is a
//     fixed register: per word and tap one v_perm (odd multiples of 16 bits only) and ONE v_bitop3 (M & (W ^ inv), the
//     preamble bit as a scalar 0 / ~0) -- no address arithmetic, no LDS access in the tap loop;
//   * survivors of the D taps (2^-D of the positions in noise) go to the per-wave list of k2_search.h; everything behind
//     that (remaining taps on the list entries, per-tile ranks, ordered emission into the staging slots, counts, overflow
//     protocol) is the first kernel's code reading the new tile layout -- K3 and the host see no difference.
// D = 10 for one preamble, 12 for several (AMR_K2S_D1 / AMR_K2S_DN; 16 in the first version: 47 / 39 / 37 us at 16 / 12 /
// 10).  Used when every preamble has at least D symbols (all of rtlamr's have 16 or more: scm+ 16, scm 21, idm / netidm
// / r900 32) and a row has 64..256 words; otherwise k2_search_fast / k2_search_dense run.
// (A first version without the LDS tile -- every wave streaming its words and the look-ahead straight from global memory --
// read 141 MB instead of 64 MiB, the look-ahead of D = 16 taps being longer than a wave's own segment, and was no faster
// than the first kernel.)
#pragma once
#include "k2_search.h"

#ifndef AMR_K2S_D1
#define AMR_K2S_D1 10   // taps of the register sweep, one preamble
#endif
#ifndef AMR_K2S_DN
#define AMR_K2S_DN 12   // taps of the register sweep, several preambles
#endif
#ifndef AMR_K2S_DBG
#define AMR_K2S_DBG 0
#endif
#ifndef AMR_K2S_OCC
#define AMR_K2S_OCC 4   // waves per SIMD the register allocation aims at (launch bound)
#endif

namespace amr {

template <int SL, int D>
struct K2SGeom {
    static constexpr int LOOK = ((D - 1) * SL + 31) / 32;           // words beyond w that the taps of word w reach
    static constexpr int NEED = (3 + LOOK) / 4 + 1;                 // chunks (4 words) a group of 4 words needs
    static constexpr int PF = 1;                                    // chunks of look-ahead (the source is LDS)
    static constexpr int RC = NEED + PF;                            // ring size in chunks
    static constexpr int RW = RC * 4;                               // ring size in words
};

// LDS bytes per chunk: 64 rows + the row slot of the next tile's row 0.  LDS-DMA addresses LDS through a 16-bit offset in
// M0, so the staged tile must end below 64 KiB: rows of 256 words (8 waves) have no room for the 65th slot; there the
// next tile's row 0 sits in a compact array behind the tile (16 bytes per chunk) and lane 63 is pointed at it.
template <int NWV> struct K2SLds {
    static constexpr bool kExtra = NWV == 8;
    static constexpr uint32_t kChunk = kExtra ? 64 * 16 : 65 * 16;
};
constexpr int kK2SWords = 32;             // words of a row per wave
constexpr int kK2SList = 128;             // (key, mask) entries per wave: with D = 16 a wave of noise yields < 1, a packet a few

inline int k2_stream_waves(uint32_t wpb) { return (int)(wpb / kK2SWords); }
// LDS per workgroup: the tile, the per-wave candidate lists, the hit counts per (row, wave) -- two preambles share a
// word (16 bits each: a wave's 32 words of a row hold at most 1024 hits) -- and the per-wave totals of the scan.  The
// ranks (bases) reuse the tile's space once the sweep and stage 2 are done.  Rows of 256 words with four preambles come
// to 78.9 KB: two workgroups per CU.
inline size_t k2_stream_lds_bytes(uint32_t wpb, uint32_t n_pre)
{
    const int nwv = k2_stream_waves(wpb);
    const size_t tile = nwv == 8 ? (size_t)(wpb / 4) * (1024 + 16) : (size_t)(wpb / 4) * 1040;   // same bytes, different layout
    return tile + ((size_t)nwv * kK2SList * 2 + (size_t)((n_pre + 1) / 2) * 64 * nwv + 4 * 8) * 4;
}

// inclusive prefix sum over the 64 lanes of a wave: four row_shr steps inside the rows of 16, then the two row broadcasts
__device__ __forceinline__ uint32_t k2s_wave_scan(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);    // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);    // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);    // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);    // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return x;
}

typedef uint32_t k2s_v4u __attribute__((ext_vector_type(4)));

// word x (0 .. 2*wpb-1) of the stream that starts with row l: LDS byte offset in the staged tile
template <int NWV>
__device__ __forceinline__ uint32_t k2s_off(uint32_t l, uint32_t x, uint32_t wpb_mask, uint32_t lg_wpb)
{
    const uint32_t c = (x & wpb_mask) >> 2, r = l + (x >> lg_wpb);
    if (K2SLds<NWV>::kExtra && r == 64) return ((wpb_mask + 1) >> 2) * K2SLds<NWV>::kChunk + c * 16 + (x & 3) * 4;
    return c * K2SLds<NWV>::kChunk + r * 16 + (x & 3) * 4;
}

template <int SL, int D>
struct K2SRing { k2s_v4u c[K2SGeom<SL, D>::RC]; };

struct K2SCtx {
    uint32_t a_own, a_next;        // LDS byte address of this wave's first chunk in row slot `lane` / in row slot lane+1
    uint32_t a_next63;             // kExtra: lane 63's base into the compact next-tile row (minus the chunk stride it does not have)
    uint32_t k_row;                // chunks from the wave's first chunk to the end of the row
    uint32_t w_lo, w_hi, lane, n_pre;
    uint32_t npb[4];               // ~preamble bits
    uint32_t *mylist;
    uint32_t list_n, lcap;
};

// Ring loads are ordinary LDS loads: hipcc waits for all of them (lgkmcnt(0)) at the first use behind a branch, which
// costs an LDS latency here and there but other waves fill it.  (Issued from inline asm with hand-counted waits they
// were faster on paper and wrong in practice: the register allocator saves and restores ring registers around the
// rarely taken candidate path, and a chunk still in flight was restored with the value from before it landed.)
template <int SL, int D, int NWV, int SLOT, int K>
__device__ __forceinline__ void k2s_load(K2SRing<SL, D> &R, const K2SCtx &cx)   // chunk K of the wave's stream
{
    // past the row end the stream continues in the next row slot, at the chunk index counted from the row start
    constexpr int CS = (int)K2SLds<NWV>::kChunk;
    typedef const __attribute__((address_space(3))) k2s_v4u *lds_v4;
    uint32_t nx = cx.a_next;
    if (K2SLds<NWV>::kExtra) nx = cx.lane == 63 ? cx.a_next63 - (uint32_t)(K * (CS - 16)) : nx;
    const uint32_t ad = (uint32_t)K < cx.k_row ? cx.a_own : nx;
    R.c[SLOT] = *(lds_v4)(uintptr_t)(ad + (uint32_t)(K * CS));
}

template <int SL, int D, int NWV, int K>
__device__ __forceinline__ void k2s_fill(K2SRing<SL, D> &R, const K2SCtx &cx)
{
    if constexpr (K < K2SGeom<SL, D>::RC) {
        k2s_load<SL, D, NWV, K, K>(R, cx);
        k2s_fill<SL, D, NWV, K + 1>(R, cx);
    }
}
// The sweep: groups GG .. 7 of 4 words each (template recursion: every ring index has to be a constant, and the
// optimizer refuses to unroll a loop of this size on request).
template <int SL, int D, int NWV, int GG>
__device__ __forceinline__ void k2s_sweep(K2SRing<SL, D> &R, K2SCtx &cx, uint32_t w0)
{
    using G = K2SGeom<SL, D>;
    if constexpr (GG < kK2SWords / 4) {
        for (uint32_t q = 0; q < cx.n_pre; ++q) {                 // rolled: the ring registers do not depend on q
            uint32_t npb = cx.npb[0];                             // explicit select: a dynamic index would go to scratch
#pragma unroll
            for (int qq = 1; qq < 4; ++qq) npb = q == (uint32_t)qq ? cx.npb[qq] : npb;
            uint32_t M[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#pragma unroll
            for (int p = 0; p < D; ++p) {
                const int x = (p * SL) >> 5;
                const bool half = ((p * SL) & 31) != 0;        // SL is a multiple of 16: the shift is 0 or 16
                const uint32_t inv = (uint32_t)((int32_t)(npb << (31 - p)) >> 31);   // scalar: ~0 where preamble[p] == 0
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i0 = (GG * 4 + j + x) % G::RW, i1 = (i0 + 1) % G::RW;
                    const uint32_t W = half ? __builtin_amdgcn_alignbit(R.c[i0 >> 2][i0 & 3], R.c[i1 >> 2][i1 & 3], 16) : R.c[i0 >> 2][i0 & 3];
                    M[j] = __builtin_amdgcn_bitop3_b32(M[j], W, inv, 0x60);   // M & (W ^ inv) in one pass
                }
            }
            if (__ballot((M[0] | M[1] | M[2] | M[3]) != 0)) {  // rare: record the non-zero masks of valid words
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t w = w0 + GG * 4 + j;
                    const uint32_t m = (w >= cx.w_lo && w < cx.w_hi) ? M[j] : 0u;
                    const uint64_t b = __ballot(m != 0);
                    if (b) {
                        const uint32_t idx = cx.list_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32),
                                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
                        if (m != 0 && idx < cx.lcap) {
                            cx.mylist[idx * 2] = (q << 16) | (cx.lane << 8) | w;
                            cx.mylist[idx * 2 + 1] = m;
                        }
                        cx.list_n += __popcll(b);
                    }
                }
            }
        }
        // the group's chunk is dead: refill its ring slot (chunks past the last group's reach are never used)
        if constexpr (GG + G::RC <= kK2SWords / 4 - 1 + G::NEED - 1)
        k2s_load<SL, D, NWV, GG % G::RC, GG + G::RC>(R, cx);
        k2s_sweep<SL, D, NWV, GG + 1>(R, cx, w0);
    }
}

template <int SL, int D, int NWV>
__global__ __launch_bounds__(64 * NWV, AMR_K2S_OCC) void k2_search_stream(const K2Args a)
{
    using G = K2SGeom<SL, D>;
    constexpr int MAXP = 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds != 0) __builtin_trap();   // the M0 values below
#if AMR_K2S_DBG   // harness builds: per-workgroup phase stamps (shader clock) and 100 MHz start / end ticks
#define K2S_STAMP(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define K2S_STAMP(i) do { } while (0)
#endif
    // workgroup b runs on XCD b % 8: every XCD gets one contiguous run of tiles (the grid is rounded up to 8 equal runs;
    // -5 us: the prologue of a later round finds the next tile's row 0 in its own L2)
    const uint32_t T = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    k2_announce(a);
    if (k2_extra_workgroup(a, T, lds, 64 * NWV)) return;   // the state update of the batch, next to the search
    K2S_STAMP(0);
#if AMR_K2S_DBG
    if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wpb = a.g.wpb, lg_wpb = a.g.lg_wpb, wpb_mask = wpb - 1;
    const uint32_t lg_bs = a.g.lg_block_size;
    const uint32_t maxL = a.g.max_pre_len, n_pre = a.g.n_pre;
    const uint32_t tile_words = 64u << lg_wpb;
    const uint32_t cpr = wpb >> 2;                      // chunks per row
    constexpr int NT = 64 * NWV;
    constexpr int LCAP = kK2SList;
    constexpr uint32_t CS = K2SLds<NWV>::kChunk;
    constexpr bool kExtra = K2SLds<NWV>::kExtra;
    uint8_t *tileb = reinterpret_cast<uint8_t *>(lds);  // [cpr][65][16 B], or [cpr][64][16 B] + [cpr][16 B]
    uint32_t *lists = lds + cpr * 260;                 // [NWV][LCAP][2]
    const uint32_t planes = (n_pre + 1) >> 1;
    uint32_t *cnts = lists + NWV * LCAP * 2;           // [planes][NT], index row*NWV+wave; preamble q in half q&1 of plane q>>1
    uint32_t *wtot = cnts + planes * NT;               // [MAXP][NWV]
    uint32_t *bases = lds;                             // [n_pre][NT]: over the tile, once nothing reads it any more

    // ---- stage the tile: chunk c of all 64 rows is 1 KiB contiguous in memory and in LDS ----
    {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(a.qt + (size_t)T * tile_words) + lane * 16;
        for (uint32_t c = v; c < cpr; c += NWV) {
            const uint32_t m0 = c * CS;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :: "v"(src + (size_t)c * 1024), "s"(m0) : "memory");
        }
        // row 0 of the next tile -> row slot 64 of every chunk
        if (tid < cpr) {
            const k2s_v4u x = *reinterpret_cast<const k2s_v4u *>(a.qt + (size_t)(T + 1) * tile_words + (size_t)tid * 256);
            *reinterpret_cast<k2s_v4u *>(tileb + (kExtra ? cpr * CS + tid * 16 : tid * CS + 1024)) = x;
        }
        for (uint32_t q = 0; q < planes; ++q) cnts[q * NT + tid] = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    K2S_STAMP(1);

    uint64_t pbits[MAXP];
    uint32_t plen[MAXP];
#pragma unroll
    for (int q = 0; q < MAXP; ++q) { pbits[q] = q < (int)n_pre ? a.g.pre_bits[q] : 0; plen[q] = q < (int)n_pre ? a.g.pre_len[q] : 0; }

    // ---- valid word range of this lane's row: n_lo <= R*BS + 32w < n_hi ----
    const int64_t rowbase = ((int64_t)T * 64 + lane - 64) << lg_bs;
    int64_t lo64 = (a.n_lo - rowbase) >> 5, hi64 = (a.n_hi - rowbase) >> 5;
    const uint32_t w_lo = (uint32_t)(lo64 < 0 ? 0 : lo64 > (int64_t)wpb ? wpb : lo64);
    const uint32_t w_hi = (uint32_t)(hi64 < 0 ? 0 : hi64 > (int64_t)wpb ? wpb : hi64);

    uint32_t *mylist = lists + v * (LCAP * 2);

    // ---- stage 1: register-ring sweep over words [32v, 32v+32) of all 64 rows ----
    K2SRing<SL, D> R;
    K2SCtx cx;
    const uint32_t c0 = v * (kK2SWords / 4);            // first chunk of this wave
    cx.a_own = c0 * CS + lane * 16;
    cx.a_next = (uint32_t)((int32_t)c0 - (int32_t)cpr) * CS + (lane + 1) * 16;   // + K * chunk: chunk c0 + K - cpr of row slot lane+1
    cx.a_next63 = cpr * CS + (uint32_t)((int32_t)c0 - (int32_t)cpr) * 16;        // + K * 16: entry c0 + K - cpr of the compact row
    cx.k_row = cpr - c0;
    cx.w_lo = w_lo; cx.w_hi = w_hi; cx.lane = lane; cx.n_pre = n_pre; cx.mylist = mylist; cx.list_n = 0; cx.lcap = (uint32_t)LCAP;
    cx.npb[0] = ~(uint32_t)pbits[0]; cx.npb[1] = ~(uint32_t)pbits[1]; cx.npb[2] = ~(uint32_t)pbits[2]; cx.npb[3] = ~(uint32_t)pbits[3];
    k2s_fill<SL, D, NWV, 0>(R, cx);
    k2s_sweep<SL, D, NWV, 0>(R, cx, v * kK2SWords);
    const uint32_t list_n = cx.list_n;
    K2S_STAMP(2);

    // ---- stage 2: remaining taps on the list entries (one per lane), compaction in place ----
    const uint32_t n_cand = list_n < (uint32_t)LCAP ? list_n : (uint32_t)LCAP;
    uint32_t n_keep = 0;                                // wave-uniform
    for (uint32_t e0 = 0; e0 < n_cand; e0 += 64) {
        const uint32_t e = e0 + lane;
        uint32_t key = 0, m = 0;
        if (e < n_cand) { key = mylist[e * 2]; m = mylist[e * 2 + 1]; }
        const uint32_t q = key >> 16, l = (key >> 8) & 63, w = key & 0xff;
        uint64_t pb = pbits[0];
        uint32_t pl = plen[0];
#pragma unroll
        for (int qq = 1; qq < MAXP; ++qq)
            if (q == (uint32_t)qq) { pb = pbits[qq]; pl = plen[qq]; }
        for (uint32_t p = D; p < maxL; p += 4) {        // four taps per round: their eight LDS reads are in flight together
            if (!__any(m != 0)) break;
            uint32_t Wd[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t pk = p + k < maxL ? p + k : maxL - 1;
                const uint32_t o = pk * SL;
                const uint32_t x = w + (o >> 5);
                const uint32_t A = *reinterpret_cast<const uint32_t *>(tileb + k2s_off<NWV>(l, x, wpb_mask, lg_wpb));
                const uint32_t B = *reinterpret_cast<const uint32_t *>(tileb + k2s_off<NWV>(l, x + 1, wpb_mask, lg_wpb));
                Wd[k] = (o & 31) ? __builtin_amdgcn_alignbit(A, B, 16) : A;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (p + k < pl) m &= ((pb >> (p + k)) & 1) ? Wd[k] : ~Wd[k];
        }
        const uint64_t b = __ballot(m != 0);
        if (m != 0) {   // survivors move to the front, order preserved (slot <= e, earlier chunks already read)
            const uint32_t slot = n_keep + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
            mylist[slot * 2] = key;
            mylist[slot * 2 + 1] = m;
            atomicAdd(&cnts[(q >> 1) * NT + l * NWV + v], (uint32_t)__popc(m) << ((q & 1) * 16));
        }
        n_keep += __popcll(b);
    }
    K2S_STAMP(3);
    __syncthreads();
    K2S_STAMP(4);

    // ---- ranks: exclusive scan over (row, wave) in stream order, all preambles in one pass ----
    uint32_t total[MAXP], val[MAXP], inc[MAXP];
#pragma unroll
    for (int q = 0; q < MAXP; ++q) {
        val[q] = q < (int)n_pre ? (cnts[(q >> 1) * NT + tid] >> ((q & 1) * 16)) & 0xffffu : 0u;
        inc[q] = k2s_wave_scan(val[q]);
        if (lane == 63) wtot[q * NWV + v] = inc[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MAXP; ++q) {
        total[q] = 0;
        if (q >= (int)n_pre) continue;
        uint32_t base = 0, tot = 0;
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const uint32_t t = wtot[q * NWV + u];
            tot += t;
            base += (uint32_t)u < v ? t : 0u;
        }
        total[q] = tot;
        bases[q * NT + tid] = base + inc[q] - val[q];
    }
    __syncthreads();
    K2S_STAMP(5);

    // ---- emit: every surviving entry by 32 lanes at once, lane b = bit b (MSB first = stream order) ----
    uint32_t run[MAXP];
#pragma unroll
    for (int q = 0; q < MAXP; ++q) run[q] = 0;
    for (uint32_t e = 0; e < n_keep; ++e) {
        const uint32_t key = __builtin_amdgcn_readfirstlane(mylist[e * 2]);
        const uint32_t m = __builtin_amdgcn_readfirstlane(mylist[e * 2 + 1]);
        const uint32_t q = key >> 16, l = (key >> 8) & 63, w = key & 0xff;
        uint32_t r = 0;
#pragma unroll
        for (int qq = 0; qq < MAXP; ++qq)
            if (q == (uint32_t)qq) r = __builtin_amdgcn_readlane(run[qq], l);
        const uint32_t base = bases[q * NT + l * NWV + v] + r;
        if (lane < 32 && ((m >> (31 - lane)) & 1)) {
            const uint32_t before = lane ? __popc(m >> (32 - lane)) : 0;
            const uint32_t rank = base + before;
            if (rank < a.cap) a.staging[((size_t)T * n_pre + q) * a.cap + rank] = (l << lg_bs) + (w << 5) + lane;
        }
        const uint32_t add = (lane == l) ? __popc(m) : 0;
#pragma unroll
        for (int qq = 0; qq < MAXP; ++qq)
            if (q == (uint32_t)qq) run[qq] += add;
    }
    K2S_STAMP(6);
#if AMR_K2S_DBG
    if (a.dbg && tid == 0) {
        a.dbg[(size_t)blockIdx.x * 16 + 7] = ((unsigned long long)n_cand << 32) | n_keep;
        a.dbg[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    if (tid == 0) {
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            if (q >= (int)n_pre) continue;
            const uint32_t c = total[q] < a.cap ? total[q] : a.cap;
            a.counts[q * a.n_tiles + T] = c;
            if (c) atomicAdd(&a.gcnt[q * k2_groups(a.n_tiles) + (T >> 6)], c);
            if (total[q] > a.cap) atomicOr(a.overflow, 1u);
        }
    }
    if (lane == 0 && list_n > (uint32_t)LCAP) atomicOr(a.overflow, 2u);
}

}  // namespace amr
