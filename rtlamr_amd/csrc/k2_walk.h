// K2 (third generation) -- the preamble search of Decoder.Search (protocol/decode.go:255-328: every idx with
// Quantized[idx + p*SymbolLength] == preamble[p] for all p, ascending), as a STREAM WALK: one wave = one tile of 64 rows,
// lane = row = one reference block, every lane walks ITS row from the first word to the last, straight out of global
// memory into a register ring.
//
// Why.  The second generation (round 2's k2_stream.h) staged a tile in LDS, split a row's words over 2 / 4 / 8 waves and met three
// barriers per tile; with rows of 256 words (BlockSize 8192: idm, "all") a workgroup held 79 KB of LDS, two fitted a CU, and
// the kernel took 0.22 ms (one preamble) to 0.54 ms (four) per 4 GiB of IQ for ~30 us of VALU work and 256 MiB of
// reads -- a chain of exposed latencies.  Here nothing is staged and nothing is shared:
//   * the "tiled4" bitstream puts words 4c..4c+3 of all 64 rows into one contiguous KiB, so chunk c of the wave's 64
//     rows is ONE coalesced global_load_dwordx4 (16 bytes per lane) -- no LDS round trip, no transpose;
//   * a lane keeps the words its first D taps can reach in a register ring of RC chunks, refilled PF chunks ahead of
//     their use (ordinary loads: the compiler counts the vmcnt);
//   * the stream of a row continues in the next row: past the row end a lane's loads simply go on at row l+1 (lane 63:
//     row 0 of the next tile) -- the look-ahead is re-read (LOOK words per row), not exchanged;
//   * all preambles share the window words (one SymbolLength); the sixteen tap polarities of rtlamr's four preambles are
//     compile-time constants, so two taps fold into ONE v_bitop3 (k2w_sweep below): 8 instructions per word and
//     preamble, plus one v_perm per window at an odd multiple of 16 bits; several preambles run one after the other on
//     the same ring registers;
//   * D = 16 taps run on every position: 2^-16 of them survive in noise (8 per preamble and wave at BlockSize 8192), so
//     the remaining taps of the longer preambles (scm 21, idm / netidm / r900 32) are rare enough to fetch their words
//     from memory one candidate per lane;
//   * a wave owns its tile alone: hit counts per row, the exclusive scan over the rows (DPP) and the ordered emission into
//     the staging slot need no barrier.  Four waves (four consecutive tiles) make a workgroup only so that the folded
//     state update (K2Args::do_hist) has 256 threads.
// Output (counts, group sums, staging slots, overflow protocol) is that of the other search kernels: K3 and the host see
// no difference.  Used when every registered preamble is one of rtlamr's own (at most four distinct ones exist) and a row
// has 16..256 words; otherwise k2_search_fast / k2_search_dense run.
#pragma once
#include "k2_common.h"

#ifndef AMR_K2W_DBG
#define AMR_K2W_DBG 0     // harness builds: phase stamps per wave in K2Args::dbg
#endif

namespace amr {

template <int SL, int PF>
struct K2WGeom {
    static constexpr int D = kK2WTaps;
    static constexpr int LOOK = ((D - 1) * SL + 31) / 32;           // words beyond w that the taps of word w reach
    static constexpr int NEED = (3 + LOOK) / 4 + 1;                 // chunks (4 words) a group of 4 words needs
    static constexpr int RC = NEED + PF;                            // ring size in chunks
    static constexpr int RW = RC * 4;                               // ring size in words
    // bytes a lane may touch behind the end of the tile after its own: chunk index (NEED - 1) + RC past a row end
    static_assert((size_t)(NEED + RC + 1) * 1024 <= kQtSlackBytes, "the bitstream allocation's slack (kQtSlackBytes) must cover the walk's over-read");
};

#ifndef AMR_K2W_PF1
#define AMR_K2W_PF1 5
#endif
#ifndef AMR_K2W_PFM
#define AMR_K2W_PFM 2
#endif
constexpr int kK2WPrefetch = AMR_K2W_PFM;   // chunks loaded ahead of their use, several preambles: a group takes 2-4 x the time there,
                                            // and 12 VGPRs fewer than with 5 bring the set of four under 168 (three waves per SIMD)
constexpr int kK2WPrefetch1 = AMR_K2W_PF1; // ... one preamble: little arithmetic per group, the loads need more lead


typedef uint32_t k2w_v4u __attribute__((ext_vector_type(4)));

template <int RC>
struct K2WRing { k2w_v4u c[RC]; };

struct K2WCtx {
    const uint8_t *tile;      // wave-uniform: chunk 0 of row 0 of the wave's tile
    uint32_t voff_own;        // byte offset of this lane's row inside a chunk: lane * 16
    uint32_t voff_next;       // ... of the row behind it: (lane + 1) * 16, lane 63: row 0 of the next tile (a whole tile on)
    uint32_t cpr;             // chunks per row
};

// chunk k of the lane's stream (k counts from the row start and runs past the row end into the next row).  k is
// wave-uniform, so the address is a scalar base (tile + chunk) plus one of two per-lane offsets: no 64-bit lane arithmetic.
template <int RC>
__device__ __forceinline__ void k2w_load(K2WRing<RC> &R, const K2WCtx &cx, int slot, uint32_t k)
{
    const bool in_row = k < cx.cpr;                                               // wave-uniform
    const uint8_t *base = cx.tile + (size_t)(in_row ? k : k - cx.cpr) * 1024;     // wave-uniform
    R.c[slot] = *reinterpret_cast<const k2w_v4u *>(base + (in_row ? cx.voff_own : cx.voff_next));
}

// one group of four words (ring slot GG of the current ring turn), all taps, all preambles; then the group's chunk is
// dead and its slot takes the chunk RC further on.  Template recursion: every ring index has to be a constant.
// ---- the sweep, specialised at COMPILE time on the preamble --------------------------------------------------------------
// rtlamr's parsers bring four preambles, fixed in their NewParser functions: scm/scm.go:45, scmplus/scmplus.go:52,
// idm/idm.go:52 (netidm/netidm.go:63 the same string), r900/r900.go:60.  With the first sixteen symbols known at compile
// time the tap polarities move from a scalar operand into v_bitop3's truth table, which leaves the operand slots for
// window words: M & f(W_p) & g(W_p+1) is ONE instruction instead of two (the first one takes three taps, having no M to
// carry): 8 per word and preamble instead of 16.  The multi-preamble search is bound by exactly this count (a wave64
// VALU instruction holds its SIMD for 4 cycles: "all" at 4 GiB is 67 M words x 40 = 2.7 G lane-operations = 70 us of the chip's VALU).
// Any other preamble (a custom protocol entry) sends the whole set to the fallback kernels of k2_search.h.
constexpr uint32_t k2w_bits(const char *s)                                    // bit p = s[p] == '1', first sixteen symbols
{
    uint32_t v = 0;
    for (int p = 0; p < kK2WTaps && s[p]; ++p) v |= (uint32_t)(s[p] == '1') << p;
    return v;
}
constexpr uint32_t kK2WKnown[4] = {k2w_bits("111110010101001100000"),             // scm
                                   k2w_bits("0001011010100011"),                  // scm+
                                   k2w_bits("01010101010101010001011010100011"),  // idm, netidm
                                   k2w_bits("00000000000000001110010101100100")};  // r900
constexpr uint32_t kK2WKnownLen[4] = {21, 16, 32, 32};
constexpr uint64_t k2w_bits64(const char *s)
{
    uint64_t v = 0;
    for (int p = 0; p < 64 && s[p]; ++p) v |= (uint64_t)(s[p] == '1') << p;
    return v;
}
constexpr uint64_t kK2WKnownAll[4] = {k2w_bits64("111110010101001100000"), k2w_bits64("0001011010100011"),
                                      k2w_bits64("01010101010101010001011010100011"), k2w_bits64("00000000000000001110010101100100")};

// which of the four a registered preamble is (-1: none of them)
inline int k2_walk_kind(uint32_t len, uint64_t bits)
{
    for (int k = 0; k < 4; ++k)
        if (len == kK2WKnownLen[k] && bits == kK2WKnownAll[k]) return k;
    return -1;
}

// window word of tap P for word j of ring group GG
template <int SL, int PF, int GG, int P>
__device__ __forceinline__ uint32_t k2w_win(const K2WRing<K2WGeom<SL, PF>::RC> &R, int j)
{
    using G = K2WGeom<SL, PF>;
    constexpr int x = (P * SL) >> 5;
    constexpr bool half = ((P * SL) & 31) != 0;                        // SL is a multiple of 16: the shift is 0 or 16
    const int i0 = (GG * 4 + j + x) % G::RW, i1 = (i0 + 1) % G::RW;
    return half ? __builtin_amdgcn_alignbit(R.c[i0 >> 2][i0 & 3], R.c[i1 >> 2][i1 & 3], 16) : R.c[i0 >> 2][i0 & 3];
}

// taps P, P+1 folded into M: truth table of  x & (y == b_P) & (z == b_P+1)  over index 4x + 2y + z
template <int SL, int PF, int GG, uint32_t BITS, int P>
__device__ __forceinline__ void k2w_pairs(const K2WRing<K2WGeom<SL, PF>::RC> &R, uint32_t (&M)[4])
{
    if constexpr (P + 1 < kK2WTaps) {
        constexpr uint32_t tt = 1u << (4 + 2 * ((BITS >> P) & 1u) + ((BITS >> (P + 1)) & 1u));
#pragma unroll
        for (int j = 0; j < 4; ++j)
            M[j] = __builtin_amdgcn_bitop3_b32(M[j], k2w_win<SL, PF, GG, P>(R, j), k2w_win<SL, PF, GG, P + 1>(R, j), tt);
        k2w_pairs<SL, PF, GG, BITS, P + 2>(R, M);
    } else if constexpr (P < kK2WTaps) {                               // the odd tap out: x & (y == b_P), z ignored
        constexpr uint32_t b = (BITS >> P) & 1u;
        constexpr uint32_t tt = (1u << (4 + 2 * b)) | (1u << (4 + 2 * b + 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t W = k2w_win<SL, PF, GG, P>(R, j);
            M[j] = __builtin_amdgcn_bitop3_b32(M[j], W, W, tt);
        }
    }
}

// all sixteen taps of one preamble on the four words of group GG: 8 instructions per word
template <int SL, int PF, int GG, uint32_t BITS>
__device__ __forceinline__ void k2w_sweep(const K2WRing<K2WGeom<SL, PF>::RC> &R, uint32_t (&M)[4])
{
    constexpr uint32_t tt = 1u << (4 * (BITS & 1u) + 2 * ((BITS >> 1) & 1u) + ((BITS >> 2) & 1u));   // (x == b0) & (y == b1) & (z == b2)
#pragma unroll
    for (int j = 0; j < 4; ++j)
        M[j] = __builtin_amdgcn_bitop3_b32(k2w_win<SL, PF, GG, 0>(R, j), k2w_win<SL, PF, GG, 1>(R, j), k2w_win<SL, PF, GG, 2>(R, j), tt);
    k2w_pairs<SL, PF, GG, BITS, 3>(R, M);
}

// ---- several preambles: tap-major.  Preamble after preamble, the compiler kept the funnel-shifted windows of the first
// sweep alive for the next ones (they are common subexpressions): 186 VGPRs for the set of four, two waves per SIMD, and
// a batch of 2^k + 1 tiles then needs a third round for its last tile.  Here a pair of windows is formed once, applied
// to every preamble's masks and dropped: 16 masks + 8 windows live instead of 4 masks + up to 32 windows. ----
template <int SL, int PF, int GG, int SET, int P, int KIND>
__device__ __forceinline__ void k2w_apply(uint32_t (&M)[4][4], const uint32_t (&Wa)[4], const uint32_t (&Wb)[4], const uint32_t (&Wc)[4])
{
    if constexpr (KIND < 4) {
        if constexpr ((SET >> KIND) & 1) {
            constexpr uint32_t BITS = kK2WKnown[KIND];
            if constexpr (P == 0) {                                     // (x == b0) & (y == b1) & (z == b2)
                constexpr uint32_t tt = 1u << (4 * (BITS & 1u) + 2 * ((BITS >> 1) & 1u) + ((BITS >> 2) & 1u));
#pragma unroll
                for (int j = 0; j < 4; ++j) M[KIND][j] = __builtin_amdgcn_bitop3_b32(Wa[j], Wb[j], Wc[j], tt);
            } else if constexpr (P + 1 < kK2WTaps) {                     // x & (y == b_P) & (z == b_P+1)
                constexpr uint32_t tt = 1u << (4 + 2 * ((BITS >> P) & 1u) + ((BITS >> (P + 1)) & 1u));
#pragma unroll
                for (int j = 0; j < 4; ++j) M[KIND][j] = __builtin_amdgcn_bitop3_b32(M[KIND][j], Wa[j], Wb[j], tt);
            } else {                                                    // the odd tap out: x & (y == b_P)
                constexpr uint32_t b = (BITS >> P) & 1u;
                constexpr uint32_t tt = (1u << (4 + 2 * b)) | (1u << (4 + 2 * b + 1));
#pragma unroll
                for (int j = 0; j < 4; ++j) M[KIND][j] = __builtin_amdgcn_bitop3_b32(M[KIND][j], Wa[j], Wa[j], tt);
            }
        }
        k2w_apply<SL, PF, GG, SET, P, KIND + 1>(M, Wa, Wb, Wc);
    }
}

template <int SL, int PF, int GG, int SET, int P>
__device__ __forceinline__ void k2w_sweep_set(const K2WRing<K2WGeom<SL, PF>::RC> &R, uint32_t (&M)[4][4])
{
    if constexpr (P < kK2WTaps) {
        uint32_t Wa[4], Wb[4], Wc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Wa[j] = k2w_win<SL, PF, GG, P>(R, j);
            Wb[j] = P + 1 < kK2WTaps ? k2w_win<SL, PF, GG, (P + 1 < kK2WTaps ? P + 1 : P)>(R, j) : 0u;
            Wc[j] = P == 0 ? k2w_win<SL, PF, GG, 2>(R, j) : 0u;
        }
        k2w_apply<SL, PF, GG, SET, P, 0>(M, Wa, Wb, Wc);
        // pin the order: without it the scheduler hoists the next pairs' funnel shifts above this pair's bitops again
        asm volatile("" : "+v"(M[0][0]), "+v"(M[1][0]), "+v"(M[2][0]), "+v"(M[3][0]));
        k2w_sweep_set<SL, PF, GG, SET, (P == 0 ? 3 : P + 2)>(R, M);
    }
}

// ---- SymbolLength = 16 mod 32: two classes of taps (round 4) ---------------------------------------------------------------
// Every odd tap then starts at a half word, and the sweeps above form its window with a funnel shift (one more instruction
// per odd tap and word; shared by the preambles of a set, not by one preamble alone).  Evaluated instead on the position
// blocks they ARE aligned to -- B-block u = positions [32u + 16, 32u + 48): the window of odd tap P is the plain ring word
// u + (P*SL >> 5) + 1 -- the odd taps need no shifted windows at all; their mask is shifted back once per word:
//     M[w] = A[w] & alignbit(B[w-1], B[w], 16)        (A: the even taps on the ordinary blocks)
// 4 + 4 v_bitop3 and one v_alignbit per word and preamble: 9 instead of 16 for one preamble, 36 instead of 40 for the
// set of four, and no window temporaries.  One register of carry per preamble (B of the word in front of the group);
// at the start of a row it is computed from the row's own words (indices x_P >= 0).  k2_row.h uses the same scheme.
template <int SL>
constexpr bool k2w_has_half() { return (SL & 31) != 0; }

template <int SL, int PF>
__device__ __forceinline__ uint32_t k2w_ring_word(const K2WRing<K2WGeom<SL, PF>::RC> &R, int s)
{
    const int i = s % K2WGeom<SL, PF>::RW;
    return R.c[i >> 2][i & 3];
}

// taps P, P + 2, ... < 16 of one class on ring words base + (tap offset); FOLD: the last instruction also ANDs `extra` in
template <int SL, int PF, uint32_t BITS, int P, bool FIRST, bool FOLD>
__device__ __forceinline__ uint32_t k2w_chain(const K2WRing<K2WGeom<SL, PF>::RC> &R, int base, uint32_t acc, uint32_t extra)
{
    constexpr int D = kK2WTaps;
    constexpr int x0 = (P * SL) >> 5, x1 = ((P + 2) * SL) >> 5, x2 = ((P + 4) * SL) >> 5;
    constexpr uint32_t b0 = (BITS >> P) & 1u, b1 = (BITS >> (P + 2)) & 1u, b2 = (BITS >> (P + 4)) & 1u;
    if constexpr (FIRST) {
        constexpr uint32_t tt = 1u << (4 * b0 + 2 * b1 + b2);                       // (x == b0) & (y == b1) & (z == b2)
        acc = __builtin_amdgcn_bitop3_b32(k2w_ring_word<SL, PF>(R, base + x0), k2w_ring_word<SL, PF>(R, base + x1),
                                          k2w_ring_word<SL, PF>(R, base + x2), tt);
        return k2w_chain<SL, PF, BITS, P + 6, false, FOLD>(R, base, acc, extra);
    } else if constexpr (P + 2 < D) {                                             // x & (y == b0) & (z == b1)
        constexpr uint32_t tt = 1u << (4 + 2 * b0 + b1);
        acc = __builtin_amdgcn_bitop3_b32(acc, k2w_ring_word<SL, PF>(R, base + x0), k2w_ring_word<SL, PF>(R, base + x1), tt);
        return k2w_chain<SL, PF, BITS, P + 4, false, FOLD>(R, base, acc, extra);
    } else {                                                                      // the last tap of the class alone
        static_assert(P < D, "eight taps per class");
        if constexpr (FOLD) {                                                     // x & (y == b0) & z
            constexpr uint32_t tt = 1u << (4 + 2 * b0 + 1);
            return __builtin_amdgcn_bitop3_b32(acc, k2w_ring_word<SL, PF>(R, base + x0), extra, tt);
        } else {                                                                  // x & (y == b0)
            constexpr uint32_t tt = (1u << (4 + 2 * b0)) | (1u << (4 + 2 * b0 + 1));
            const uint32_t W = k2w_ring_word<SL, PF>(R, base + x0);
            return __builtin_amdgcn_bitop3_b32(acc, W, W, tt);
        }
    }
}

// one preamble, the four words of group GG; Bc: class-B mask of the word in front of the group (in / out)
template <int SL, int PF, int GG, uint32_t BITS>
__device__ __forceinline__ void k2w_sweep_ab(const K2WRing<K2WGeom<SL, PF>::RC> &R, uint32_t (&M)[4], uint32_t &Bc)
{
    uint32_t B[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) B[j] = k2w_chain<SL, PF, BITS, 1, true, false>(R, GG * 4 + j + 1, 0u, 0u);
    asm volatile("" : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t Bs = __builtin_amdgcn_alignbit(j ? B[j - 1] : Bc, B[j], 16);
        M[j] = k2w_chain<SL, PF, BITS, 0, true, true>(R, GG * 4 + j, 0u, Bs);
    }
    Bc = B[3];
}

// class-B mask of the B-block in front of word 0 of a row: own-row words only (ring slots x_P, filled before the walk)
template <int SL, int PF, uint32_t BITS>
__device__ __forceinline__ uint32_t k2w_b_first(const K2WRing<K2WGeom<SL, PF>::RC> &R)
{
    return k2w_chain<SL, PF, BITS, 1, true, false>(R, 0, 0u, 0u);
}

// record the non-zero masks of one group and preamble (rare path)
__device__ __forceinline__ void k2w_record(const uint32_t (&M)[4], uint32_t q, uint32_t g, uint32_t w_lo, uint32_t w_hi, uint32_t lane,
                                           uint32_t *mylist, uint32_t &list_n)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t w = g * 4 + j;
        const uint32_t m = (w >= w_lo && w < w_hi) ? M[j] : 0u;
        const uint64_t b = __ballot(m != 0);
        if (b) {
            const uint32_t idx = list_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
            if (m != 0 && idx < (uint32_t)kK2WList) {
                mylist[idx * 2] = (q << 16) | (lane << 8) | w;
                mylist[idx * 2 + 1] = m;
            }
            list_n += __popcll(b);
        }
    }
}

// one known preamble (KIND, present in the launch's SET) on the four words of group GG
template <int SL, int PF, int GG, int SET, int KIND>
__device__ __forceinline__ void k2w_kind(const K2WRing<K2WGeom<SL, PF>::RC> &R, uint32_t pids, uint32_t g, uint32_t w_lo, uint32_t w_hi,
                                         uint32_t lane, uint32_t *mylist, uint32_t &list_n, uint32_t (&Bc)[4])
{
    if constexpr ((SET >> KIND) & 1) {
        uint32_t M[4];
        if constexpr (k2w_has_half<SL>()) k2w_sweep_ab<SL, PF, GG, kK2WKnown[KIND]>(R, M, Bc[KIND]);
        else k2w_sweep<SL, PF, GG, kK2WKnown[KIND]>(R, M);
        if (__ballot((M[0] | M[1] | M[2] | M[3]) != 0))                 // rare
            k2w_record(M, (pids >> (8 * KIND)) & 0xffu, g, w_lo, w_hi, lane, mylist, list_n);
    }
}

// one group of four words (ring slot GG of the current ring turn), all taps, all preambles of the SET; then the group's
// chunk is dead and its slot takes the chunk RC further on.  Template recursion: every ring index has to be a constant.
// SET: which of rtlamr's four preambles the launch searches, a compile-time mask -- a launch's code holds the sweeps it
// runs and no others.  (One kernel for all sets with a run-time choice per preamble was 75 KB of loop body; the
// instruction cache two CUs share holds 64 KB and every wave walks the whole body: 0.18 ms for what takes 0.08.)
// pids: the preamble id (K2Args order) of each kind, 8 bits each.
template <int SL, int PF, int SET, int GG>
__device__ __forceinline__ void k2w_groups(K2WRing<K2WGeom<SL, PF>::RC> &R, const K2WCtx &cx, uint32_t g0, uint32_t n_groups,
                                           uint32_t pids, uint32_t w_lo, uint32_t w_hi,
                                           uint32_t lane, uint32_t *mylist, uint32_t &list_n, uint32_t (&Bc)[4])
{
    using G = K2WGeom<SL, PF>;
    if constexpr (GG < G::RC) {
        const uint32_t g = g0 + GG;
        if (g >= n_groups) return;                                  // wave-uniform
        if constexpr ((SET & (SET - 1)) == 0) {                     // one preamble
            k2w_kind<SL, PF, GG, SET, 0>(R, pids, g, w_lo, w_hi, lane, mylist, list_n, Bc);
            k2w_kind<SL, PF, GG, SET, 1>(R, pids, g, w_lo, w_hi, lane, mylist, list_n, Bc);
            k2w_kind<SL, PF, GG, SET, 2>(R, pids, g, w_lo, w_hi, lane, mylist, list_n, Bc);
            k2w_kind<SL, PF, GG, SET, 3>(R, pids, g, w_lo, w_hi, lane, mylist, list_n, Bc);
        } else {                                                    // several: tap-major (k2w_sweep_set)
            uint32_t M[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) M[k][j] = 0;
            if constexpr (k2w_has_half<SL>()) {      // preamble after preamble: the classes leave nothing to share between them
                if constexpr (SET & 1) { k2w_sweep_ab<SL, PF, GG, kK2WKnown[0]>(R, M[0], Bc[0]); asm volatile("" : "+v"(M[0][0]), "+v"(M[0][1]), "+v"(M[0][2]), "+v"(M[0][3])); }
                if constexpr (SET & 2) { k2w_sweep_ab<SL, PF, GG, kK2WKnown[1]>(R, M[1], Bc[1]); asm volatile("" : "+v"(M[1][0]), "+v"(M[1][1]), "+v"(M[1][2]), "+v"(M[1][3])); }
                if constexpr (SET & 4) { k2w_sweep_ab<SL, PF, GG, kK2WKnown[2]>(R, M[2], Bc[2]); asm volatile("" : "+v"(M[2][0]), "+v"(M[2][1]), "+v"(M[2][2]), "+v"(M[2][3])); }
                if constexpr (SET & 8) { k2w_sweep_ab<SL, PF, GG, kK2WKnown[3]>(R, M[3], Bc[3]); }
            } else
                k2w_sweep_set<SL, PF, GG, SET, 0>(R, M);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (((SET >> k) & 1) && __ballot((M[k][0] | M[k][1] | M[k][2] | M[k][3]) != 0))   // rare
                    k2w_record(M[k], (pids >> (8 * k)) & 0xffu, g, w_lo, w_hi, lane, mylist, list_n);
        }
        // unconditionally, also past the last chunk the walk needs (PF + 1 chunks of the following row, harmless): a load
        // inside a branch makes the compiler's waitcnt pass give up counting and wait for ALL loads in flight at every
        // group (vmcnt(0): 64 exposed memory latencies per row walk)
        k2w_load<G::RC>(R, cx, GG, g + G::RC);
        k2w_groups<SL, PF, SET, GG + 1>(R, cx, g0, n_groups, pids, w_lo, w_hi, lane, mylist, list_n, Bc);
    }
}

template <int RC, int K>
__device__ __forceinline__ void k2w_fill(K2WRing<RC> &R, const K2WCtx &cx, uint32_t n_chunks)
{
    if constexpr (K < RC) {
        k2w_load<RC>(R, cx, K, K);
        k2w_fill<RC, K + 1>(R, cx, n_chunks);
    }
}

// inclusive prefix sum over the 64 lanes of a wave: four row_shr steps inside the rows of 16, then the two row broadcasts
__device__ __forceinline__ uint32_t k2w_wave_scan(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);    // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);    // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);    // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);    // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return x;
}

template <int SL, int SET>
__global__ __launch_bounds__(64 * kK2WWaves) void k2_search_walk(const K2Args a)
{
    constexpr int PF = (SET & (SET - 1)) == 0 ? kK2WPrefetch1 : kK2WPrefetch;
    using G = K2WGeom<SL, PF>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
#if AMR_K2W_DBG
#define K2W_STAMP(i) do { if (a.dbg && lane == 0) a.dbg[(size_t)T * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define K2W_STAMP(i) do { } while (0)
#endif
    // workgroup b runs on XCD b % 8: every XCD gets one contiguous run of tiles (the grid is rounded up to 8 equal runs)
    const uint32_t wgT = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint32_t n_wg = (a.n_tiles + kK2WWaves - 1) / kK2WWaves;      // workgroups that search
    k2_announce(a);
    if (wgT >= n_wg) {
        (void)k2_extra_workgroup(a, a.n_tiles + (wgT - n_wg), lds, 64 * kK2WWaves);   // state update / deferred-block copies
        return;
    }
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t T = wgT * kK2WWaves + v;
    if (T >= a.n_tiles) return;                                      // the last workgroup may hold fewer tiles (no barrier below)
#if AMR_K2W_DBG
    if (a.dbg && lane == 0) a.dbg[(size_t)T * 16 + 8] = __builtin_amdgcn_s_memrealtime();
#endif
    K2W_STAMP(0);
    const uint32_t wpb = a.g.wpb, lg_wpb = a.g.lg_wpb;
    const uint32_t lg_bs = a.g.lg_block_size;
    const uint32_t tile_words = 64u << lg_wpb;
    const uint32_t cpr = wpb >> 2;                                   // chunks per row
    uint32_t *mylist = lds + v * (kK2WList * 2 + 2 * 4 * 64);        // [kK2WList][2]
    uint32_t *cnts = mylist + kK2WList * 2;                          // [4][64] hits per (preamble, row)
    uint32_t *bases = cnts + 4 * 64;                                 // [4][64]
#pragma unroll
    for (int q = 0; q < 4; ++q) cnts[q * 64 + lane] = 0;

    // ---- the preambles (up to four; the taps behind the sixteenth, stage 2, take their bits from the geometry) ----
    constexpr int NP = 4;
    const uint32_t n_pre = a.g.n_pre;
    uint64_t pbits[NP];
    uint32_t plen[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        pbits[q] = q < (int)n_pre ? a.g.pre_bits[q] : 0;
        plen[q] = q < (int)n_pre ? a.g.pre_len[q] : 0;
    }

    // ---- valid word range of this lane's row: n_lo <= R*BS + 32w < n_hi ----
    const int64_t rowbase = ((int64_t)T * 64 + lane - 64) << lg_bs;
    int64_t lo64 = (a.n_lo - rowbase) >> 5, hi64 = (a.n_hi - rowbase) >> 5;
    const uint32_t w_lo = (uint32_t)(lo64 < 0 ? 0 : lo64 > (int64_t)wpb ? wpb : lo64);
    const uint32_t w_hi = (uint32_t)(hi64 < 0 ? 0 : hi64 > (int64_t)wpb ? wpb : hi64);

    // ---- stage 1: the walk ----
    const uint8_t *tile = reinterpret_cast<const uint8_t *>(a.qt + (size_t)T * tile_words);
    K2WCtx cx;
    cx.cpr = cpr;
    cx.tile = tile;
    cx.voff_own = lane * 16;
    cx.voff_next = lane == 63 ? tile_words * 4 : (lane + 1) * 16;
    const uint32_t n_chunks = cpr + G::NEED - 1;                     // chunks of the stream the walk touches
    K2WRing<G::RC> R;
    k2w_fill<G::RC, 0>(R, cx, n_chunks);
    uint32_t list_n = 0;                                             // wave-uniform
    uint32_t Bc[4] = {0u, 0u, 0u, 0u};                               // class-B carries (SymbolLength = 16 mod 32), per kind
    if constexpr (k2w_has_half<SL>()) {
        if constexpr (SET & 1) Bc[0] = k2w_b_first<SL, PF, kK2WKnown[0]>(R);
        if constexpr (SET & 2) Bc[1] = k2w_b_first<SL, PF, kK2WKnown[1]>(R);
        if constexpr (SET & 4) Bc[2] = k2w_b_first<SL, PF, kK2WKnown[2]>(R);
        if constexpr (SET & 8) Bc[3] = k2w_b_first<SL, PF, kK2WKnown[3]>(R);
    }
    for (uint32_t g0 = 0; g0 < cpr; g0 += G::RC)
        k2w_groups<SL, PF, SET, 0>(R, cx, g0, cpr, a.walk_pids, w_lo, w_hi, lane, mylist, list_n, Bc);
    K2W_STAMP(1);

    // ---- stage 2: the taps behind the first 16 on the list entries (one per lane), words from memory; compaction in place ----
    const uint32_t maxL = a.g.max_pre_len;
    const uint32_t n_cand = list_n < (uint32_t)kK2WList ? list_n : (uint32_t)kK2WList;
    uint32_t n_keep = 0;                                             // wave-uniform
    const uint32_t *tw = a.qt + (size_t)T * tile_words;
    for (uint32_t e0 = 0; e0 < n_cand; e0 += 64) {
        const uint32_t e = e0 + lane;
        uint32_t key = 0, m = 0;
        if (e < n_cand) { key = mylist[e * 2]; m = mylist[e * 2 + 1]; }
        const uint32_t q = key >> 16, l = (key >> 8) & 63, w = key & 0xff;
        uint64_t pb = pbits[0];
        uint32_t pl = plen[0];
#pragma unroll
        for (int qq = 1; qq < NP; ++qq)
            if (q == (uint32_t)qq) { pb = pbits[qq]; pl = plen[qq]; }
        for (uint32_t p = kK2WTaps; p < maxL; p += 4) {              // four taps per round: their eight loads are in flight together
            if (!__any(m != 0)) break;
            uint32_t Wd[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t pk = p + k < maxL ? p + k : maxL - 1;
                const uint32_t o = pk * SL;
                const uint32_t x = w + (o >> 5);
                // word x of the stream that starts with row l of this tile: tiled row l + x / wpb (may be row 0 of the next tile)
                const uint32_t A = tw[qt_index(l + (x >> lg_wpb), x & (wpb - 1), lg_wpb)];
                const uint32_t B = tw[qt_index(l + ((x + 1) >> lg_wpb), (x + 1) & (wpb - 1), lg_wpb)];
                Wd[k] = (o & 31) ? __builtin_amdgcn_alignbit(A, B, 16) : A;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (p + k < pl) m &= ((pb >> (p + k)) & 1) ? Wd[k] : ~Wd[k];
        }
        const uint64_t b = __ballot(m != 0);
        if (m != 0) {   // survivors move to the front, order preserved (slot <= e, earlier entries already read)
            const uint32_t slot = n_keep + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
            mylist[slot * 2] = key;
            mylist[slot * 2 + 1] = m;
            atomicAdd(&cnts[q * 64 + l], (uint32_t)__popc(m));
        }
        n_keep += __popcll(b);
    }
    K2W_STAMP(2);

    // ---- ranks: exclusive scan over the rows in stream order (row-major: all of row l before row l+1) ----
    uint32_t total[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const uint32_t val = cnts[q * 64 + lane];
        const uint32_t inc = k2w_wave_scan(val);
        total[q] = __builtin_amdgcn_readlane(inc, 63);
        bases[q * 64 + lane] = inc - val;
    }

    // ---- emit: every surviving entry by 32 lanes at once, lane b = bit b (MSB first = stream order).  The list is in
    // walk order: word-major across the rows, ascending words inside a row -- which is all the ranks need ----
    uint32_t run[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) run[q] = 0;
    for (uint32_t e = 0; e < n_keep; ++e) {
        const uint32_t key = __builtin_amdgcn_readfirstlane(mylist[e * 2]);
        const uint32_t m = __builtin_amdgcn_readfirstlane(mylist[e * 2 + 1]);
        const uint32_t q = key >> 16, l = (key >> 8) & 63, w = key & 0xff;
        uint32_t r = 0;
#pragma unroll
        for (int qq = 0; qq < NP; ++qq)
            if (q == (uint32_t)qq) r = __builtin_amdgcn_readlane(run[qq], l);
        const uint32_t base = bases[q * 64 + l] + r;
        if (lane < 32 && ((m >> (31 - lane)) & 1)) {
            const uint32_t before = lane ? __popc(m >> (32 - lane)) : 0;
            const uint32_t rank = base + before;
            if (rank < a.cap) a.staging[((size_t)T * n_pre + q) * a.cap + rank] = (l << lg_bs) + (w << 5) + lane;
        }
        const uint32_t add = (lane == l) ? __popc(m) : 0;
#pragma unroll
        for (int qq = 0; qq < NP; ++qq)
            if (q == (uint32_t)qq) run[qq] += add;
    }
    K2W_STAMP(3);
#if AMR_K2W_DBG
    if (a.dbg && lane == 0) {
        a.dbg[(size_t)T * 16 + 7] = ((unsigned long long)n_cand << 32) | n_keep;
        a.dbg[(size_t)T * 16 + 9] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (q >= (int)n_pre) break;
            const uint32_t c = total[q] < a.cap ? total[q] : a.cap;
            a.counts[q * a.n_tiles + T] = c;
            if (c) atomicAdd(&a.gcnt[(q * k2_groups(a.n_tiles) + (T >> 6)) * kGroupStride], c);
            if (total[q] > a.cap) atomicOr(a.overflow, 1u);
        }
        if (list_n > (uint32_t)kK2WList) atomicOr(a.overflow, 2u);
    }
}

}  // namespace amr
