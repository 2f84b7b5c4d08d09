// K2 (fourth generation, one preamble) -- Decoder.Search (protocol/decode.go:255-328) with the WHOLE ROW IN REGISTERS.
//
// k2_walk.h walks a row through a register ring of NEED + PF chunks and, past the row end, goes on reading the next row
// from memory: every row's look-ahead (LOOK words: 68 of 128 at SymbolLength 144, scm) is fetched twice -- 1.53 x the
// bitstream at BlockSize 4096, 1.70 x measured (profiles/r03/pmc_summary_cfg2.json) -- and the walk is a chain of
// exposed latencies (PF chunks of 1 KiB in flight per wave, 60 % of the wave-cycles waiting).  For ONE preamble and
// rows of up to 128 words neither is necessary:
//   * a lane loads its whole row at once, CPR = WPB / 4 coalesced global_load_dwordx4 back to back (the tiled4 layout:
//     chunk c of the wave's 64 rows is one contiguous KiB): the tile's 32 KiB are in flight together, the memory
//     system streams instead of answering five requests per wave at a time;
//   * the look-ahead of a lane is the head of the NEXT row -- which the neighbour lane holds in its registers.  Chunk c
//     of a lane's own row is dead once group c has been swept (the lanes of a wave run in lockstep: dead for all of
//     them at once), so right then the register is shifted IN PLACE by one lane (v_mov_b32 dpp wave_shl:1): slot c now
//     holds chunk CPR + c of the lane's stream.  The row is its own ring, RW = WPB, and the window code of k2_walk.h
//     (ring index = stream word mod RW) applies unchanged.  Lane 63 continues in row 0 of the next tile: lanes
//     0 .. NEED-2 fetch those chunks once (one load instruction) and v_readlane / v_writelane patch lane 63;
//   * nothing is over-fetched: the bitstream is read 1.0 x plus 16 bytes per chunk of look-ahead and tile (7 % at scm).
// Candidate list, stage 2 and ranks are those of k2_walk.h, the sweep and the emission are restated below (same staging /
// counts / overflow protocol:
// K3 and the host see no difference).  Used for a decoder whose ONE preamble is one of rtlamr's four and whose rows have
// 16, 64 or 128 words (every single-preamble parser set except idm / netidm / r900 alone, BlockSize 8192: 256 row
// registers do not exist; they keep the ring walk, whose re-read is 27 % there).
#pragma once
#include "k2_walk.h"

#ifndef AMR_K2R_WPE
#define AMR_K2R_WPE 3      // waves per SIMD the register allocation aims at
#endif

namespace amr {

// Taps the sweep applies to every position: all of the preamble when its reach still fits the row-as-ring scheme (scm: 21
// symbols -- no candidate list of near-misses, no second stage; scm+ has 16 anyway), the first sixteen otherwise (idm,
// netidm, r900: 32 symbols reach past the row; scm at SymbolLength 96, where 21 symbols fill the 2048-sample block).
template <int SL, int KIND, int WPB>
constexpr int k2r_taps()
{
    constexpr int len = (int)kK2WKnownLen[KIND];
    return (len <= 24 && (((len - 1) * SL) >> 5) + 4 < WPB) ? len : kK2WTaps;
}

template <int SL, int WPB, int DT = kK2WTaps>
struct K2RGeom {
    static constexpr int D = DT;
    static constexpr int LOOK = ((D - 1) * SL + 31) / 32;           // words beyond w that the taps of word w reach
    static constexpr int CPR = WPB / 4;                             // chunks per row = ring size in chunks
    static constexpr int RC = CPR;
    static constexpr int RW = WPB;
    static constexpr int NLA = (LOOK + 3) / 4;                      // look-ahead chunks a row walk needs (chunks 0 .. NLA-1 of the next row)
    // slot c is shifted to the next row after group c; group g reads stream words up to 4g + 3 + LOOK + 1 (the funnel
    // shift's second word): they must lie in slots already shifted, i.e. in chunks <= g - 1 of the next row
    static_assert(((D - 1) * SL >> 5) + 4 < WPB, "look-ahead does not fit the row-as-ring scheme");
    static_assert(NLA <= CPR && NLA <= 64, "look-ahead longer than a row");
};

// The BlockSize a decoder with only preamble KIND registered gets (decode.go:131-141: NextPowerOf2 of the preamble's
// length in samples), in words; and whether k2_search_row exists for it.
template <int SL, int KIND>
constexpr int k2r_wpb()
{
    uint32_t pl = kK2WKnownLen[KIND] * (uint32_t)SL, bs = 1;
    while (bs < pl) bs <<= 1;
    return (int)(bs / 32);
}
// lanes a row is split over: 1 up to 128 words, 2 for rows of 256 words (idm / netidm / r900 alone at chip length 72 .. 96),
// 0: no row kernel for this geometry.  Two lanes per row read the bitstream once where the walk re-reads every row's
// look-ahead: bit-identical, 20 % faster on its own (idm, 4 GiB: 64.0 -> 51.4 us in tools/k2_bench.hip).  Round 4 had it in
// the harness only, because in the pipelined product the FIRST of the two K1 rounds behind it took 500 us instead of 390;
// round 5 found why (the previous batch's gate kernel, which came onto the chip next to this kernel's waves and ended up in
// the middle of a SIMD's register file: submit() in amr_pipeline.hip, profiles/r05/k1_gate_fragmentation.txt) and it is
// the product's search for these geometries now.  AMR_K2R_LPR2=0 builds the round-4 library (tools/build_variant.sh).
#ifndef AMR_K2R_LPR2
#define AMR_K2R_LPR2 1
#endif
template <int SL, int KIND>
constexpr int k2r_lpr()
{
    constexpr int w = k2r_wpb<SL, KIND>();
    return (w == 16 || w == 32 || w == 64 || w == 128) ? 1 : (w == 256 && AMR_K2R_LPR2) ? 2 : 0;
}
template <int SL, int KIND>
constexpr bool k2r_ok()
{
    constexpr int lpr = k2r_lpr<SL, KIND>();
    if (lpr == 0) return false;
    return (((kK2WTaps - 1) * SL) >> 5) + 4 < k2r_wpb<SL, KIND>() / lpr;
}

// ---- the sweep: sixteen taps on the four words of group GG ------------------------------------------------------------
// Tap P of the positions [32w, 32w + 32) looks at the stream bits [32w + P*SL, + 32).  SL is a multiple of 16, so a tap is
// either word-aligned (class A) or starts at a half word (class B: the odd taps when SL = 16 mod 32).  k2_walk.h forms
// every class-B window with a funnel shift (8 extra instructions per word; the compiler shares them between groups, which
// costs 68 more live registers with the whole row resident: two waves per SIMD instead of three).  Here the class-B
// taps are evaluated on the position blocks they ARE aligned to -- B-block u = positions [32u + 16, 32u + 48): its
// window of tap P is the plain word u + (P*SL >> 5) + 1 -- and the accumulated mask is shifted back once:
//     M[w] = A[w] & alignbit(B[w-1], B[w], 16)
// 4 + 4 v_bitop3 and one v_alignbit per word, no shifted copy of the stream, one register of carry (B[w-1]).
// B[-1] of a row needs own-row words only (index x_P >= 0), so a lane starts its row without its neighbour.
template <int SL>
struct K2RTaps {
    static constexpr bool kHalf = (SL & 31) != 0;                   // class B exists
    static constexpr int kStep = kHalf ? 2 : 1;                     // taps of one class: P0, P0 + kStep, ...
};

// ring slot of stream word s (the row is the ring)
template <int WPB>
__device__ __forceinline__ uint32_t k2r_word(const K2WRing<WPB / 4> &R, int s)
{
    const int i = s % WPB;
    return R.c[i >> 2][i & 3];
}

// the class of taps P, P + STEP, ... < 16 on stream words base + (tap offset): acc &= AND of (W == bit), two taps per
// v_bitop3 (three in the first); FOLD: the last instruction also ANDs `extra` in (the shifted class-B mask)
template <int SL, int WPB, int D, uint32_t BITS, int P, int STEP, bool FIRST, bool FOLD>
__device__ __forceinline__ uint32_t k2r_chain(const K2WRing<WPB / 4> &R, int base, uint32_t acc, uint32_t extra)
{
    constexpr int x0 = (P * SL) >> 5, x1 = ((P + STEP) * SL) >> 5, x2 = ((P + 2 * STEP) * SL) >> 5;
    constexpr uint32_t b0 = (BITS >> P) & 1u, b1 = (BITS >> (P + STEP)) & 1u, b2 = (BITS >> (P + 2 * STEP)) & 1u;
    if constexpr (FIRST) {
        static_assert(P + 2 * STEP < D, "a class has at least three taps");
        constexpr uint32_t tt = 1u << (4 * b0 + 2 * b1 + b2);                       // (x == b0) & (y == b1) & (z == b2)
        acc = __builtin_amdgcn_bitop3_b32(k2r_word<WPB>(R, base + x0), k2r_word<WPB>(R, base + x1), k2r_word<WPB>(R, base + x2), tt);
        return k2r_chain<SL, WPB, D, BITS, P + 3 * STEP, STEP, false, FOLD>(R, base, acc, extra);
    } else if constexpr (P + STEP < D) {                                          // x & (y == b0) & (z == b1)
        constexpr uint32_t tt = 1u << (4 + 2 * b0 + b1);
        acc = __builtin_amdgcn_bitop3_b32(acc, k2r_word<WPB>(R, base + x0), k2r_word<WPB>(R, base + x1), tt);
        return k2r_chain<SL, WPB, D, BITS, P + 2 * STEP, STEP, false, FOLD>(R, base, acc, extra);
    } else if constexpr (P < D) {                                                 // the last tap alone
        if constexpr (FOLD) {                                                     // x & (y == b0) & z
            constexpr uint32_t tt = 1u << (4 + 2 * b0 + 1);
            return __builtin_amdgcn_bitop3_b32(acc, k2r_word<WPB>(R, base + x0), extra, tt);
        } else {                                                                  // x & (y == b0)
            constexpr uint32_t tt = (1u << (4 + 2 * b0)) | (1u << (4 + 2 * b0 + 1));
            const uint32_t W = k2r_word<WPB>(R, base + x0);
            return __builtin_amdgcn_bitop3_b32(acc, W, W, tt);
        }
    } else {
        return FOLD ? (acc & extra) : acc;
    }
}

// group GG: masks of its four words; Bc = class-B mask of the B-block in front of the group (in / out)
template <int SL, int WPB, int D, int GG, uint32_t BITS>
__device__ __forceinline__ void k2r_sweep(const K2WRing<WPB / 4> &R, uint32_t (&M)[4], uint32_t &Bc)
{
    using T = K2RTaps<SL>;
    if constexpr (T::kHalf) {
        uint32_t B[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) B[j] = k2r_chain<SL, WPB, D, BITS, 1, 2, true, false>(R, GG * 4 + j + 1, 0u, 0u);
        asm volatile("" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]));       // keep the two classes apart: fewer live registers
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t Bs = __builtin_amdgcn_alignbit(j ? B[j - 1] : Bc, B[j], 16);
            M[j] = k2r_chain<SL, WPB, D, BITS, 0, 2, true, true>(R, GG * 4 + j, 0u, Bs);
        }
        Bc = B[3];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) M[j] = k2r_chain<SL, WPB, D, BITS, 0, 1, true, false>(R, GG * 4 + j, 0u, 0u);
    }
}

// class-B mask of the B-block in front of word 0 of the row: positions [-16, 16), own-row words only
template <int SL, int WPB, int D, uint32_t BITS>
__device__ __forceinline__ uint32_t k2r_b_first(const K2WRing<WPB / 4> &R)
{
    if constexpr (K2RTaps<SL>::kHalf) return k2r_chain<SL, WPB, D, BITS, 1, 2, true, false>(R, 0, 0u, 0u);
    else return 0u;
}

// slot C of the row: own chunk C (swept, dead in every lane) -> chunk C of the NEXT row: the neighbour lane's register,
// lane 63: chunk C of row 0 of the next tile, which lane C fetched into X
template <int CPRN, int C>
__device__ __forceinline__ void k2r_shift(K2WRing<CPRN> &R, const k2w_v4u &X)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t own = R.c[C][j];
        uint32_t v = (uint32_t)__builtin_amdgcn_update_dpp((int)own, (int)own, 0x130 /* wave_shl:1: lane i <- lane i + 1 */, 0xf, 0xf, false);
        const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)X[j], C);
        asm("v_writelane_b32 %0, %1, 63" : "+v"(v) : "s"(s));          // (this clang has no __builtin_amdgcn_writelane)
        R.c[C][j] = v;
    }
}

// The row's loads, hand-issued: the compiler does not see them as memory operations, so the rare-path branch inside every
// group (k2w_record) no longer makes its waitcnt pass give up counting and wait for ALL loads in flight (k2_walk.h had to
// load unconditionally and over-fetch for that reason).  Chunk K of the wave's tile is one KiB at tile + K KiB; the
// instruction's immediate reaches 4 KiB, so there is one scalar base per four chunks.
template <int CPRN, int K>
__device__ __forceinline__ void k2r_fill(K2WRing<CPRN> &R, const uint8_t *tile, uint32_t voff)
{
    if constexpr (K < CPRN) {
        const uint8_t *base = tile + (size_t)(K / 4) * 4096;          // wave-uniform: an SGPR pair
        // sc1: past L1 -- the rows may have been written by a K1 wave on another compute unit microseconds ago (early search);
        // every word is read once, there is nothing L1 could give back
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 sc1" : "=v"(R.c[K]) : "v"(voff), "s"(base), "n"((K % 4) * 1024) : "memory");
        k2r_fill<CPRN, K + 1>(R, tile, voff);
    }
}

// vmcnt retires in order: once at most CPRN - 1 - C1 loads are outstanding, chunks 0 .. C1 (and X, issued first) have
// landed.  The registers that become valid here are operands of the wait or of an (empty) statement BEHIND it -- volatile
// statements keep their order --, so that no use of them can be scheduled above the wait.
template <int CPRN, int C0, int C1, bool FIRST = true>
__device__ __forceinline__ void k2r_wait(K2WRing<CPRN> &R)
{
    static_assert(C0 <= C1 && C1 < CPRN, "chunk range");
    if constexpr (FIRST) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(R.c[C0]) : "n"(CPRN - 1 - C1) : "memory");
    else asm volatile("" : "+v"(R.c[C0]));
    if constexpr (C0 < C1) k2r_wait<CPRN, C0 + 1, C1, false>(R);
}

// groups GG .. CPR-1 of the row, statically unrolled: every register index, every wait count and the point where a slot
// turns into look-ahead are compile-time facts, and the stream of loads at the top is never interrupted by a branch
template <int SL, int WPB, int KIND, int D, int GG>
__device__ __forceinline__ void k2r_groups(K2WRing<WPB / 4> &R, k2w_v4u &X, uint32_t q, uint32_t w_lo, uint32_t w_hi,
                                           uint32_t lane, uint32_t *mylist, uint32_t &list_n, uint32_t &Bc)
{
    using G = K2RGeom<SL, WPB, D>;
    if constexpr (GG < G::CPR) {
        // the chunks of the own row this group touches for the first time: up to stream word 4 GG + 3 + x_last + 1
        constexpr int hi = (4 * GG + 4 + (((D - 1) * SL) >> 5)) >> 2;
        constexpr int c1 = hi < G::CPR - 1 ? hi : G::CPR - 1;
        constexpr int hp = GG == 0 ? -1 : ((4 * (GG - 1) + 4 + (((D - 1) * SL) >> 5)) >> 2);
        constexpr int c0 = hp < G::CPR - 1 ? hp + 1 : G::CPR;          // first chunk not yet waited for
        if constexpr (c0 <= c1) k2r_wait<G::CPR, c0, c1>(R);
        if constexpr (GG == 0) asm volatile("" : "+v"(X));           // issued first: landed with chunk 0
        uint32_t M[4];
        if constexpr (GG == 0) Bc = k2r_b_first<SL, WPB, D, (uint32_t)kK2WKnownAll[KIND]>(R);
        k2r_sweep<SL, WPB, D, GG, (uint32_t)kK2WKnownAll[KIND]>(R, M, Bc);
        if (__ballot((M[0] | M[1] | M[2] | M[3]) != 0))                 // rare
            k2w_record(M, q, (uint32_t)GG, w_lo, w_hi, lane, mylist, list_n);
        if constexpr (GG < G::NLA) k2r_shift<G::CPR, GG>(R, X);
        k2r_groups<SL, WPB, KIND, D, GG + 1>(R, X, q, w_lo, w_hi, lane, mylist, list_n, Bc);
    }
}

// LPR: lanes per row.  1: a lane holds a whole row of WPB words (rows of up to 128 words).  2 (round 4, rows of 256
// words: idm / netidm / r900 alone at chip lengths 72 .. 96): a row is split over TWO NEIGHBOURING lanes, WPB = 128 words
// each -- lane 2r takes the first half of row r, lane 2r + 1 the second, so the stream of every lane still continues in
// the lane above it and the in-place shift works unchanged; a wave then covers 32 rows, two waves a tile (they sit in
// one workgroup and meet once, for the second wave's rank offset inside the tile's staging slot).  A wave-load of chunk k
// touches the two chunks k and k + 32 of the tile, 512 contiguous bytes of each.
template <int SL, int KIND, int WPB, int LPR = 1>
__global__ __launch_bounds__(64 * kK2WWaves, AMR_K2R_WPE) void k2_search_row(const K2Args a)
{
    constexpr int D = k2r_taps<SL, KIND, WPB>();
    using G = K2RGeom<SL, WPB, D>;
    constexpr int LG_WPB = WPB == 16 ? 4 : WPB == 32 ? 5 : WPB == 64 ? 6 : 7;
    static_assert(WPB == 16 || WPB == 32 || WPB == 64 || WPB == 128, "16, 32, 64 or 128 words per lane");
    static_assert(LPR == 1 || (LPR == 2 && WPB == 128), "two lanes per row: rows of 256 words");
    constexpr int LG_LPR = LPR == 2 ? 1 : 0;
    constexpr int RPW = 64 / LPR;                                    // rows per wave
    constexpr uint32_t lg_rw = LG_WPB + LG_LPR, rw = (uint32_t)WPB * LPR;   // words per ROW
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // workgroup b runs on XCD b % 8: every XCD gets one contiguous run of tiles (the grid is rounded up to 8 equal runs)
    const uint32_t wgT = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint32_t n_wg = (a.n_tiles * LPR + kK2WWaves - 1) / kK2WWaves;   // workgroups that search
    k2_announce(a);
    if (wgT >= n_wg) {
        (void)k2_extra_workgroup(a, a.n_tiles + (wgT - n_wg), lds, 64 * kK2WWaves);   // state update / deferred-block copies
        return;
    }
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wave = wgT * kK2WWaves + v;
    const uint32_t T = wave >> LG_LPR, h2 = wave & (LPR - 1);        // tile; which RPW rows of it
    const bool active = T < a.n_tiles;                               // wave-uniform
    if (LPR == 1 && !active) return;                                 // (LPR 2: the waves of a workgroup meet at a barrier below)
    uint32_t *mylist = lds + v * (kK2WList * 2 + 2 * 4 * 64);        // [kK2WList][2]
    uint32_t *cnts = mylist + kK2WList * 2;                          // [64] hits per lane-row (one preamble)
    uint32_t *bases = cnts + 4 * 64;                                 // [64]
    uint32_t *wtot = lds + kK2WList * 2 + 64;                        // [kK2WWaves] hits per wave (LPR 2; unused words of wave 0's counts)
    const uint32_t lg_bs = lg_rw + 5;
    constexpr uint32_t tile_words = 64u << (LG_WPB + LG_LPR);
    uint32_t total = 0, list_n = 0, n_keep = 0;                      // wave-uniform
    const uint32_t rt = RPW * h2 + (lane >> LG_LPR), hl = lane & (LPR - 1);   // this lane: row of the tile, part of the row
    const uint32_t *tw = a.qt + (size_t)T * tile_words;
    if (active) {
#if AMR_K2W_DBG
    if (a.dbg && lane == 0 && h2 == 0) a.dbg[(size_t)T * 16 + 8] = __builtin_amdgcn_s_memrealtime();
#endif
    K2W_STAMP(0);
    if (a.k1_flags) {
        // early search: tile T was written by K1's wave-tile T - 1, the row behind lane 63 (X) by wave-tile T; lanes 0 and 1
        // wait for one each.  A wait that gives up marks the batch: the host searches it again, in stream order.
        const uint32_t n_wt = a.n_tiles - 1;                          // K1's wave-tiles
        bool ok = true;
        if (lane < 2) {
            const int64_t wt = (int64_t)T - 1 + lane;
            if (wt >= 0 && wt < (int64_t)n_wt) ok = k1_flag_wait(a.k1_flags + wt, a.k1_flag_value, kK1FlagTicks);
        }
        if (__any(!ok) && lane == 0) atomicOr(a.overflow, kOvfGate);
    }

    // ---- the lane's words, all of them, and the head of what follows lane 63 (lane c: its chunk c) ----
    const uint8_t *tile = reinterpret_cast<const uint8_t *>(tw);
    K2WRing<G::CPR> R;
    k2w_v4u X;
    {   // X first (the first shift needs it: in-order retirement then never waits for more than its own chunks);
        // lanes beyond NLA read chunk NLA-1 again (a valid address, never used).  Lane 63's stream continues in the row
        // behind the wave's last one: row RPW (h2 + 1) of this tile, or row 0 of the next tile
        const uint32_t xoff = (lane < (uint32_t)G::NLA ? lane : (uint32_t)G::NLA - 1) * 1024;
        const uint8_t *next = (h2 + 1 < (uint32_t)LPR) ? tile + (size_t)RPW * (h2 + 1) * 16 : tile + (size_t)tile_words * 4;
        asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(X) : "v"(xoff), "s"(next) : "memory");
    }
    k2r_fill<G::CPR, 0>(R, tile, (uint32_t)G::CPR * hl * 1024u + rt * 16u);

    cnts[lane] = 0;

    // ---- the preamble (the taps behind the D-th, stage 2, take their bits from the geometry) ----
    const uint64_t pb = a.g.pre_bits[0];
    const uint32_t pl = a.g.pre_len[0];

    // ---- valid word range of this lane: n_lo <= R*BS + 32 (row word) < n_hi, row word = WPB * hl + w ----
    const int64_t rowbase = ((int64_t)T * 64 + rt - 64) << lg_bs;
    int64_t lo64 = ((a.n_lo - rowbase) >> 5) - (int64_t)(WPB * hl), hi64 = ((a.n_hi - rowbase) >> 5) - (int64_t)(WPB * hl);
    const uint32_t w_lo = (uint32_t)(lo64 < 0 ? 0 : lo64 > (int64_t)WPB ? WPB : lo64);
    const uint32_t w_hi = (uint32_t)(hi64 < 0 ? 0 : hi64 > (int64_t)WPB ? WPB : hi64);

    // ---- stage 1: D taps on every position of the lane's words ----
    uint32_t Bc = 0;
    k2r_groups<SL, WPB, KIND, D, 0>(R, X, 0u, w_lo, w_hi, lane, mylist, list_n, Bc);
    K2W_STAMP(1);

    // ---- stage 2: the taps behind the first D on the list entries (one per lane), words from memory, eight taps (sixteen
    // loads) in flight per round; compaction in place.  Nothing to do when the sweep applied the whole preamble (scm, scm+)
    const uint32_t n_cand = list_n < (uint32_t)kK2WList ? list_n : (uint32_t)kK2WList;
    const uint32_t maxL = pl;
    for (uint32_t e0 = 0; e0 < n_cand; e0 += 64) {
        const uint32_t e = e0 + lane;
        uint32_t key = 0, m = 0;
        if (e < n_cand) { key = mylist[e * 2]; m = mylist[e * 2 + 1]; }
        const uint32_t kl = (key >> 8) & 63;                         // the lane that found it
        const uint32_t l = RPW * h2 + (kl >> LG_LPR), w = (uint32_t)WPB * (kl & (LPR - 1)) + (key & 0xff);   // row of the tile, row word
        if constexpr (D < (int)kK2WKnownLen[KIND]) {
            for (uint32_t p = D; p < maxL; p += 8) {
                if (!__any(m != 0)) break;
                uint32_t Wd[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t pk = p + k < maxL ? p + k : maxL - 1;
                    const uint32_t o = pk * SL;
                    const uint32_t x = w + (o >> 5);
                    // word x of the stream that starts with row l of this tile: tiled row l + x / rw (may be row 0 of the next tile)
                    // (agent-scope loads = sc1: see k2r_fill)
                    const uint32_t A = __hip_atomic_load(&tw[qt_index(l + (x >> lg_rw), x & (rw - 1), lg_rw)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t B = __hip_atomic_load(&tw[qt_index(l + ((x + 1) >> lg_rw), (x + 1) & (rw - 1), lg_rw)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    Wd[k] = (o & 31) ? __builtin_amdgcn_alignbit(A, B, 16) : A;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (p + k < pl) m &= ((pb >> (p + k)) & 1) ? Wd[k] : ~Wd[k];
            }
        }
        const uint64_t b = __ballot(m != 0);
        if (m != 0) {   // survivors move to the front, order preserved (slot <= e, earlier entries already read)
            const uint32_t slot = n_keep + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
            mylist[slot * 2] = key;
            mylist[slot * 2 + 1] = m;
            atomicAdd(&cnts[kl], (uint32_t)__popc(m));
        }
        n_keep += __popcll(b);
    }
    K2W_STAMP(2);

    // ---- ranks: exclusive scan over the lanes in stream order (lane-major = row-major: all of a row before the next) ----
    const uint32_t val = cnts[lane];
    const uint32_t inc = k2w_wave_scan(val);
    total = __builtin_amdgcn_readlane(inc, 63);
    bases[lane] = inc - val;
    }   // active

    // ---- two waves per tile: the second one's hits come behind the first one's in the tile's staging slot ----
    uint32_t tile_base = 0, tile_total = total;
    if constexpr (LPR > 1) {
        if (lane == 0) wtot[v] = active ? total : 0u;
        __syncthreads();
        if (!active) return;
        tile_base = h2 ? wtot[v - 1] : 0u;
        tile_total = wtot[v & ~1u] + wtot[v | 1u];
    }

    // ---- emit.  The list is in walk order: word-major across the lanes, ascending words inside a lane, so the slot of an
    // entry = the hits of the lanes in front of its lane + the hits of the earlier entries of its own lane.  Sixty-four
    // entries at a time live in registers (lane e = entry e): their slots come out of one pass of v_readlane broadcasts,
    // and the positions of an entry are then written by 32 lanes at once (lane b = bit b, MSB first = stream order) from
    // broadcast values -- no LDS or memory latency inside either loop (k2_walk.h re-reads the list entry by entry: 700
    // cycles per entry; a tile with two packets holds eight).
    uint32_t run = 0;                                                // lane l: hits of lane-row l emitted by earlier rounds
    for (uint32_t e0 = 0; e0 < n_keep; e0 += 64) {
        const uint32_t e = e0 + lane;
        uint32_t key = 0, m = 0;
        if (e < n_keep) { key = mylist[e * 2]; m = mylist[e * 2 + 1]; }
        const uint32_t l = (key >> 8) & 63, c = (uint32_t)__popc(m);
        uint32_t slot = tile_base + bases[l] + (uint32_t)__builtin_amdgcn_ds_bpermute((int)(l << 2), (int)run);
        const uint32_t n = n_keep - e0 < 64u ? n_keep - e0 : 64u;    // wave-uniform
        uint32_t add = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t lj = (uint32_t)__builtin_amdgcn_readlane((int)l, (int)j), cj = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)j);
            slot += (lane > j && l == lj) ? cj : 0u;
            add += (lane == lj) ? cj : 0u;
        }
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)j), mj = (uint32_t)__builtin_amdgcn_readlane((int)m, (int)j);
            const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)slot, (int)j);
            if (lane < 32 && ((mj >> (31 - lane)) & 1)) {
                const uint32_t rank = sj + (lane ? __popc(mj >> (32 - lane)) : 0);
                const uint32_t klj = (kj >> 8) & 63;
                const uint32_t row = RPW * h2 + (klj >> LG_LPR), wrow = (uint32_t)WPB * (klj & (LPR - 1)) + (kj & 0xff);
                if (rank < a.cap) a.staging[(size_t)T * a.cap + rank] = (row << lg_bs) + (wrow << 5) + lane;
            }
        }
        run += add;
    }
    K2W_STAMP(3);
#if AMR_K2W_DBG
    if (a.dbg && lane == 0 && h2 == 0) {
        a.dbg[(size_t)T * 16 + 7] = ((unsigned long long)list_n << 32) | n_keep;
        a.dbg[(size_t)T * 16 + 9] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (lane == 0) {
        if (h2 == 0) {
            const uint32_t c = tile_total < a.cap ? tile_total : a.cap;
            a.counts[T] = c;
            if (c) atomicAdd(&a.gcnt[(T >> 6) * kGroupStride], c);
            if (tile_total > a.cap) atomicOr(a.overflow, 1u);
        }
        if (list_n > (uint32_t)kK2WList) atomicOr(a.overflow, 2u);
    }
}

// ---- clean-up behind an in-wave search (k1_search.h): row 63 of every tile, the words whose windows reach into the next tile ----
// One THREAD per tile: its row 63 from word w_cut on (the K1 wave searched the words below it), the look-ahead from row 0 of
// the next tile, which is complete now.  The row is its own ring as in k2_search_row: slots below the first group hold the
// next row's chunks from the start, a slot turns into look-ahead once its group has been swept.  The hits go behind the
// tile's in-wave hits (row 63's last words are the last positions of the tile); count and group sum are brought up to date.
// The history tile and the state update are a launch of k2_search_row with n_tiles = 1 (amr_pipeline.hip).
template <int SL, int WPB, int KIND, int D, int GG, int G0>
__device__ __forceinline__ void k2c_groups(K2WRing<WPB / 4> &R, const k2w_v4u (&N)[(K2RGeom<SL, WPB, D>::NLA)], uint32_t w_lo, uint32_t w_hi,
                                           uint32_t (&M)[WPB], uint32_t &Bc)
{
    using G = K2RGeom<SL, WPB, D>;
    if constexpr (GG < G::CPR) {
        uint32_t m4[4];
        k2r_sweep<SL, WPB, D, GG, (uint32_t)kK2WKnownAll[KIND]>(R, m4, Bc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = GG * 4 + j;
            M[w] = (w >= w_lo && w < w_hi) ? m4[j] : 0u;
        }
        if constexpr (GG < G::NLA) R.c[GG] = N[GG];
        k2c_groups<SL, WPB, KIND, D, GG + 1, G0>(R, N, w_lo, w_hi, M, Bc);
    }
}

template <int SL, int KIND, int WPB>
__global__ __launch_bounds__(256) void k2_row_cleanup(const K2Args a)
{
    constexpr int D = k2r_taps<SL, KIND, WPB>();
    using G = K2RGeom<SL, WPB, D>;
    constexpr int LG_WPB = WPB == 16 ? 4 : WPB == 32 ? 5 : WPB == 64 ? 6 : 7;
    constexpr uint32_t lg_bs = LG_WPB + 5, tile_words = 64u << LG_WPB;
    constexpr int w_cut = WPB - 1 - (((D - 1) * SL) >> 5), G0 = w_cut / 4;    // first word / group this kernel owns (k1s_wcut)
    static_assert(D == (int)kK2WKnownLen[KIND] && w_cut >= 1, "as k1s_ok");
    const uint32_t T = 1u + blockIdx.x * 256u + threadIdx.x;           // tile 0 is the history tile: k2_search_row's
    if (T >= a.n_tiles) return;
    const uint32_t *tw = a.qt + (size_t)T * tile_words;
    // row 63: chunk c at tw + c * 256 + 63 * 4 words; row 0 of the next tile: chunk c at tw + tile_words + c * 256
    K2WRing<G::CPR> R;
    k2w_v4u N[G::NLA];
#pragma unroll
    for (int c = 0; c < G::NLA; ++c) N[c] = *reinterpret_cast<const k2w_v4u *>(tw + tile_words + c * 256);
#pragma unroll
    for (int c = 0; c < G::CPR; ++c) {
        if (c >= G0) R.c[c] = *reinterpret_cast<const k2w_v4u *>(tw + c * 256 + 63 * 4);
        else if (c < G::NLA) R.c[c] = N[c];
        else R.c[c] = k2w_v4u{0u, 0u, 0u, 0u};                        // (swept by nobody, read by nobody)
    }
    const int64_t rowbase = ((int64_t)T * 64 + 63 - 64) << lg_bs;
    const int64_t lo64 = (a.n_lo - rowbase) >> 5, hi64 = (a.n_hi - rowbase) >> 5;
    uint32_t w_lo = (uint32_t)(lo64 < 0 ? 0 : lo64 > (int64_t)WPB ? WPB : lo64);
    const uint32_t w_hi = (uint32_t)(hi64 < 0 ? 0 : hi64 > (int64_t)WPB ? WPB : hi64);
    if (w_lo < (uint32_t)w_cut) w_lo = (uint32_t)w_cut;
    if (w_lo >= w_hi) return;                                          // (the batch's last rows: searched with the next batch)
    uint32_t M[WPB];
#pragma unroll
    for (int w = 0; w < WPB; ++w) M[w] = 0;
    // class-B mask of the B-block in front of the first group: words of the row from 4 G0 on (and look-ahead slots below it)
    uint32_t Bc = 0;
    if constexpr (K2RTaps<SL>::kHalf) Bc = k2r_chain<SL, WPB, D, (uint32_t)kK2WKnownAll[KIND], 1, 2, true, false>(R, 4 * G0, 0u, 0u);
    k2c_groups<SL, WPB, KIND, D, G0, G0>(R, N, w_lo, w_hi, M, Bc);
    uint32_t extra = 0;
#pragma unroll
    for (int w = G0 * 4; w < WPB; ++w) extra += (uint32_t)__popc(M[w]);
    if (!extra) return;
    const uint32_t have = a.counts[T];                                 // the in-wave hits of the tile (clamped to cap)
    uint32_t rank = have;
#pragma unroll
    for (int w = G0 * 4; w < WPB; ++w) {
        uint32_t m = M[w];
        while (m) {
            const uint32_t b = (uint32_t)__clz((int)m);               // bit 31 = the word's first position
            m &= ~(0x80000000u >> b);
            if (rank < a.cap) a.staging[(size_t)T * a.cap + rank] = (63u << lg_bs) + ((uint32_t)w << 5) + b;
            ++rank;
        }
    }
    const uint32_t c = rank < a.cap ? rank : a.cap;
    a.counts[T] = c;
    if (c > have) atomicAdd(&a.gcnt[(T >> 6) * kGroupStride], c - have);
    if (rank > a.cap) atomicOr(a.overflow, 1u);
}

}  // namespace amr
