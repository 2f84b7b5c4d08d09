// First-generation K1 (rounds 1-2; chip length 96's kernel until round 5).  Harness only (tools/k1_bench.hip, variant "old"):
// an independently written second implementation of the same arithmetic whose checksum every k1_tile.h variant must reproduce.
// K1 -- fused magnitude LUT + cumulative-sum matched filter + quantize + pack.
//
// Computes, for every reference block k of a batch, exactly what
//   MagLUT.Execute  (protocol/decode.go:219-225)
//   Decoder.Filter  (protocol/decode.go:229-245)
// compute in the k-th Decoder.Decode call (decode.go:163-172), and stores the
// BlockSize new bit decisions (Decoder.Quantized[PacketLength:]) packed 32 per
// word, first sample in the most significant bit.
//
// Bit-exactness: the reference filter is a SEQUENTIAL float32 running sum that
// restarts at zero for every call over BlockSize+SymbolLength samples
// (decode.go:232-236).  float32 addition is not associative, so any tree or
// wavefront scan would round differently and flip bit decisions at the zero
// crossings (hard part 1).  The exact mapping is therefore
//     one lane = one reference block,
// each lane walking its block's samples in the reference's order with the
// reference's restart point; a 64-wide wavefront processes 64 consecutive
// blocks.  The chip holds 256 CUs x 8 waves x 64 lanes = 131072 such lanes,
// i.e. 1 GiB of SCM chip-72 IQ in flight at once.
//
// Data movement: a lane's samples are 2*BlockSize bytes apart from its
// neighbour's, so rows are staged through LDS as a transpose.  Per staging tile
// the wave issues 8 LDS-DMA loads (global_load_lds_dwordx4): 8 lanes cover one
// 128-byte line of one row, so HBM sees whole, aligned cache lines; each lane
// then reads its own row 16 bytes at a time (ds_read_b128).  The 16-byte column
// index is XOR-swizzled on the SOURCE side ((row>>1)&7), which makes the
// row-per-lane ds_read_b128 bank-conflict free.  Two tile buffers per wave:
// tile t+1 is in flight while tile t is consumed.
//
// The csum window the filter needs (c[t-CL] and d[t-CL] = c[t-CL]-c[t-SL]) lives
// in 2*CL VGPRs per lane, addressed statically by unrolling the sample loop CL
// times ("body").  f = d[t-CL] - (c[t] - c[t-CL]) is the reference's
// (csum[i+CL]-csum[i]) - (csum[i+SL]-csum[i+CL]) with the same three roundings.
//
// Output layout ("tiled4"): word w of block b is stored at
//     qt[(b/64 + 1) * 64*WPB + (w/4)*256 + (b%64)*4 + w%4],   WPB = BlockSize/32,
// i.e. a lane stores 16 bytes and a wave 1 KiB per store instruction.  Tile 0 of
// qt holds the previous batch's last rows (history for the preamble search).
#pragma once
#include "k1_common.h"

// Developer diagnostics (never defined in the product build): 1 = no HBM traffic after the first tile
// (times the LDS/VALU side alone), 2 = HBM->LDS staging + row reads only (times the memory side alone),
// 3 = no output stores.
#ifndef AMR_K1_DIAG
#define AMR_K1_DIAG 0
#endif
// 1: LUT gathers one group ahead of their use (32 more VGPRs), 0: gathers, row read and one wait per group
// LDS waits of a group: 0 = hipcc's own (one s_waitcnt per LUT pair as the values are used), 1 = one wait for all 16
// gathers before the arithmetic, 2 = two waits (first half, second half).  Measured on MI355X (chip 72, 1 GiB):
// 0.2226 / 0.2244 / 0.2223 ms -- seven fewer s_waitcnt per group buy nothing, a satisfied s_waitcnt is almost free.
#ifndef AMR_K1_WAIT
#define AMR_K1_WAIT 0
#endif
// 1: magnitude add and the two differences as v_pk_add_f32 (12 packed + 8 plain adds per group instead of 32 plain:
// 11 fewer instructions of 90).  Measured on MI355X, A/B on one box: 0.2296 / 0.2174 ms against 0.2276 / 0.2144 ms for
// the plain build -- no gain: a packed op takes two passes through the FP32 lanes, and those, not the issue slots, are
// what the arithmetic costs.  Bit-exact either way (tests/test_gpu_parity.py, test_gpu_fullsize.py pass with it).
// 1: non-temporal stores for the bitstream.  Measured: K1 0.2305 vs 0.2265 ms, K2 48.4 vs 46.5 us -- worse on both sides
// (K2 finds part of the 64 MiB bitstream in the Infinity Cache when K1 wrote it with the default policy).
#ifndef AMR_K1_NTSTORE
#define AMR_K1_NTSTORE 0
#endif
#ifndef AMR_K1_PK
#define AMR_K1_PK 0
#endif
// (AMR_K1_PIPE, default 0 in k1_common.h: measured on MI355X (SCM chip 72, 1 GiB): 0.231 ms without, 0.239 ms with, and 40 fewer free VGPRs)

namespace amr {

template <int CL>
struct K1Lane {
    // Rings of RING = CL+8 registers, indexed statically (the sample loop is unrolled RING times): step t
    // writes slot t%RING and reads slot (t+8)%RING = the value of step t-CL.  Because a slot is dead for 8
    // steps before it is overwritten, every value is produced straight into its final register and the
    // loop back-edge needs no register rotation (a CL-deep ring costs two v_mov per sample).
    float hc[K1Geom<CL>::RING];  // hc[t%RING] = c[t], the running sum after sample t (decode.go:234)
    float hd[K1Geom<CL>::RING];  // hd[t%RING] = c[t] - c[t-CL]
    uint4 row1;    // the 8 IQ samples of group G+1 (its LUT gathers are issued while group G is computed)
    float li[8], lq[8];  // lut[I], lut[Q] of the 8 samples of group G (gathered one group earlier)
    uint32_t acc;  // sign bits of f, newest in bit 0 (inverted decisions)
    uint32_t prev; // acc at the previous 32-sample boundary
    uint32_t xs;   // AMR_K1_DIAG == 3 only
    // Finished output words not yet written.  The insert index is wave-uniform, which hipcc lowers to one
    // v_mov under s_set_gpr_idx_on (indirect VGPR addressing); plain vector members, no arrays of vectors
    // and no references to them, or the whole struct is demoted to scratch.
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef uint32_t ow0_t __attribute__((ext_vector_type(K1Geom<CL>::NW0)));
    typedef uint32_t ow1_t __attribute__((ext_vector_type(K1Geom<CL>::NW1)));
    ow0_t ow0;
    ow1_t ow1;
};

struct K1Uni {       // wave-uniform state (SGPRs)
    uint32_t G;      // 8-sample groups computed so far
    uint32_t tcur;   // tile of the next row read
    uint32_t roff;   // LDS offset bits of the next row read: (group in tile) * 16 | (tile parity) * kTileBuf
    uint32_t wc;     // steady state: groups until the next output word completes, minus 1
    int32_t og;      // output groups produced so far (negative during warm-up)
    uint32_t ngroups;// total groups per lane
    uint32_t ntiles; // total staging tiles per lane
    uint32_t odd;    // 1 = this wave sits in an odd hardware wave slot of its SIMD (priority swap, see k1_tile.h PRIO)
    uint32_t wi;     // output words buffered in L.ow
    uint32_t wdone;  // output words already written
    uint32_t st;     // store instructions issued since the last tile DMA
};


// Read the 16 bytes (8 IQ samples) of this lane's row for the next group in line (U.F) and advance.  On a tile
// boundary first wait for the tile's DMA (issued one tile-time earlier), then refill the buffer that was
// drained before it.  Groups past the end of the lane's stream read stale LDS bytes that nobody uses.
// The LDS address is one v_xor: rdv = lane*128 | swizzle, U.roff = (group in tile)*16 | (tile parity)*8192.
// Scalar instructions go through the one scalar unit the four SIMDs of a CU share, and trimming them is what paid
// in this loop (25 -> 15 per group: 0.229 -> 0.218 ms), so the common path is 4 scalar ops.
template <int CL, bool TAIL>
__device__ __forceinline__ uint4 k1_fetch_next(K1Uni &U, const K1Args &a, uint32_t tiles_lds, const uint8_t *tiles,
                                               uint32_t wg, uint32_t lane, uint32_t rdv, uint32_t voff_e,
                                               uint32_t voff_o, uint32_t rows_valid)
{
    if ((U.roff & 0x70) == 0) {                   // first group of a tile (rare path; the wrap of roff happens here too)
        U.roff = (U.roff ^ ((U.roff & 0x80) << 6)) & (kTileBuf | 0x70);   // carry out of the group bits toggles the buffer
        const uint32_t t = U.tcur;
        U.tcur = t + 1;
        // vmcnt retires in order on gfx9 (loads and stores alike; checked by tools/dma_bench.hip), so the output
        // stores issued after the DMA of this tile need not be waited for.
        if (U.st == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (U.st == K1Geom<CL>::NW / 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K1Geom<CL>::NW / 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        U.st = 0;
        if (t + 1 < U.ntiles && AMR_K1_DIAG != 1)
            k1_prefetch<CL, TAIL>(a, tiles_lds, wg, t + 1, ((t + 1) & 1) * kTileBuf, lane, voff_e, voff_o, rows_valid);
        // the two waves of a SIMD swap priority every 8 tiles: left alone the arbiter favours the older one, which
        // finishes ~9 % early and leaves the other without a partner to hide its stalls (measured in k1_tile.h)
        if (((t >> 3) ^ U.odd) & 1) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
    }
    const uint4 r = *reinterpret_cast<const uint4 *>(tiles + (rdv ^ U.roff));
    U.roff += 16;                                 // 8 groups per tile; bit 7 set = wrapped, fixed up on the rare path
    return r;
}

template <int CL, int J>
__device__ __forceinline__ void k1_flush_chunks(const K1Lane<CL> &L, typename K1Lane<CL>::v4u *dst, uint32_t count)
{
    using G = K1Geom<CL>;
    if constexpr (4 * J < G::NW) {
        if ((uint32_t)(4 * J) < count) {
            typename K1Lane<CL>::v4u x;
            if constexpr (4 * J < G::NW0) x = {L.ow0[4 * J], L.ow0[4 * J + 1], L.ow0[4 * J + 2], L.ow0[4 * J + 3]};
            else x = {L.ow1[4 * J - G::NW0], L.ow1[4 * J + 1 - G::NW0], L.ow1[4 * J + 2 - G::NW0], L.ow1[4 * J + 3 - G::NW0]};
#if AMR_K1_NTSTORE
            __builtin_nontemporal_store(x, &dst[J * kRows]);
#else
            dst[J * kRows] = x;
#endif
        }
        k1_flush_chunks<CL, J + 1>(L, dst, count);
    }
}

// Write the buffered output words: word W of row `lane` of wave-tile `wg` lives at
//   qt[(wg+1)*64*WPB + (W>>2)*256 + lane*4 + (W&3)]   ("tiled4": a lane stores 16 bytes, a wave 1 KiB, per instruction).
template <int CL>
__device__ __forceinline__ void k1_flush(K1Lane<CL> &L, K1Uni &U, uint32_t *qbase, uint32_t count)
{
    typedef typename K1Lane<CL>::v4u v4u;
    // uniform base (SGPRs) + 32-bit lane offset: no 64-bit per-lane pointer kept alive across the loop
    uint8_t *ub = reinterpret_cast<uint8_t *>(qbase) + (size_t)(U.wdone >> 2) * (kRows * 16);
    v4u *dst = reinterpret_cast<v4u *>(ub + (uint32_t)(threadIdx.x * 16));
    k1_flush_chunks<CL, 0>(L, dst, count);
    U.wdone += count;
    U.wi = 0;
    U.st += count >> 2;
}

// One unrolled "body" = RING samples = RING/8 groups; the csum history registers are indexed statically.
// Two-stage software pipeline per group G, so that a wave never waits for the LDS and both waves of a SIMD
// can issue VALU work back to back (a lone wave issues one VALU op per ~5 cycles, two share ~2.5):
//   1. 16 LUT gathers for group G+1 (its IQ bytes, L.row1, arrived during group G-1),
//   2. the row read of group G+2,
//   3. 8 x (magnitude add, running sum, two differences, sign bit) for group G on the LUT values gathered
//      during group G-1 -- no memory access, no wait.
// sched_barrier keeps hipcc from re-interleaving this into a load-wait-use chain per sample.
// PRO: the zero-magnitude predicate of a fresh Decoder (decode.go:144), needed only by the wave that holds stream
// block 0 while its first SymbolLength samples pass; CHECK: the last, partial body.
// MODE 2 (PRO): the zero-magnitude predicate of a fresh Decoder; MODE 1: warm-up bookkeeping (no output before the
// window is full); MODE 0: steady state -- word completion is a down-counter, nothing else is tracked per group.
template <int CL, int MODE, bool CHECK, bool TAIL>
__device__ __forceinline__ void k1_body(K1Lane<CL> &L, K1Uni &U, const K1Args &a, uint32_t tiles_lds,
                                        const uint8_t *tiles, const float *lut, uint32_t wg, uint32_t lane, uint32_t rdv,
                                        uint32_t zlim, uint32_t voff_e, uint32_t voff_o, uint32_t rows_valid,
                                        uint32_t *qrow)
{
    using G = K1Geom<CL>;
#pragma unroll
    for (int g = 0; g < G::GPB; ++g) {
        constexpr bool PRO = MODE == 2;
        if (CHECK && U.G >= U.ngroups) return;
#if AMR_K1_DIAG == 2
        L.acc ^= L.row1.x ^ L.row1.y ^ L.row1.z ^ L.row1.w;
        L.row1 = k1_fetch_next<CL, TAIL>(U, a, tiles_lds, tiles, wg, lane, rdv, voff_e, voff_o, rows_valid);
#else
        const uint32_t dw[4] = {L.row1.x, L.row1.y, L.row1.z, L.row1.w};
#if AMR_K1_PIPE
        float nli[8], nlq[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t v = dw[k >> 1] >> ((k & 1) * 16);
#if AMR_K1_DIAG == 5   // no LUT traffic: one cheap ALU op per byte instead (wrong values, timing only)
            nli[k] = __uint_as_float(((v & 0xff) << 15) | 0x3c000000u);
            nlq[k] = __uint_as_float((((v >> 8) & 0xff) << 15) | 0x3c000000u);
#else
            nli[k] = lut[v & 0xff];                            // decode.go:222
            nlq[k] = lut[(v >> 8) & 0xff];
#endif
        }
        const uint4 row2 = k1_fetch_next<CL, TAIL>(U, a, tiles_lds, tiles, wg, lane, rdv, voff_e, voff_o, rows_valid);
        __builtin_amdgcn_sched_barrier(0);
#else
        // L.row1 holds group G itself here (the prologue fetched one group less): gather, fetch G+1, one wait
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t v = dw[k >> 1] >> ((k & 1) * 16);
            L.li[k] = lut[v & 0xff];                           // decode.go:222
            L.lq[k] = lut[(v >> 8) & 0xff];
        }
        const uint4 row2 = k1_fetch_next<CL, TAIL>(U, a, tiles_lds, tiles, wg, lane, rdv, voff_e, voff_o, rows_valid);
        __builtin_amdgcn_sched_barrier(0);
#if AMR_K1_WAIT == 1
        __builtin_amdgcn_s_waitcnt(0xC17F);   // lgkmcnt(1): LDS returns in order, only the row read may be outstanding
        __builtin_amdgcn_sched_barrier(0);
#elif AMR_K1_WAIT == 2
        __builtin_amdgcn_s_waitcnt(0xC97F);   // lgkmcnt(9): the first 8 gathers are back
        __builtin_amdgcn_sched_barrier(0);
#endif
#endif
#if AMR_K1_PK
        // Packed form (an experiment, see AMR_K1_PK above: fewer issue slots, same lane passes, no gain):
        // v_pk_add_f32 does two independent IEEE binary32 adds in one issue slot: the magnitude add and the two
        // differences of samples (2j, 2j+1) pair up (ring slots r, r+1 and ro, ro+1 are adjacent registers, RING
        // is even); the running sum itself stays a serial chain of plain v_add_f32 (in asm: left to itself hipcc
        // packs those too, with a wasted half and two v_mov per pair).  Same operations, same roundings.
        // gfx950 needs a wait state between a packed op and a dependent instruction right behind it, so the 28
        // instructions are laid out by hand with independent work in between, and pinned with sched_barrier.
        {
            typedef float v2f __attribute__((ext_vector_type(2)));
            constexpr int R = G::RING;
#define AMR_SB __builtin_amdgcn_sched_barrier(0)
#define AMR_ADD(dst, x, y) asm("v_add_f32 %0, %1, %2" : "=v"(dst) : "v"(x), "v"(y))
#define AMR_RO(j) ((g * 8 + 2 * (j) + 8) % R)
#define AMR_HC(j) v2f{L.hc[AMR_RO(j)], L.hc[AMR_RO(j) + 1]}
#define AMR_HD(j) v2f{L.hd[AMR_RO(j)], L.hd[AMR_RO(j) + 1]}
#define AMR_BIT(x) L.acc = __builtin_amdgcn_alignbit(L.acc, __float_as_uint(x), 31)   /* decode.go:243, inverted */
            // issue order: m0 m1 c0x m2 c0y m3 c1x d0 c1y f0 c2x d1 c2y b c3x f1 c3y b d2 b d3 f2 b f3 b b b b
#define AMR_MAG(j) v2f m##j = v2f{L.li[2 * j], L.li[2 * j + 1]} + v2f{L.lq[2 * j], L.lq[2 * j + 1]};   /* decode.go:222 */ \
            if (PRO) {                                                                    /* zero history, decode.go:144 */ \
                m##j.x = U.G * 8 + 2 * j < zlim ? 0.0f : m##j.x;                                                          \
                m##j.y = U.G * 8 + 2 * j + 1 < zlim ? 0.0f : m##j.y;                                                      \
            }                                                                                                            \
            AMR_SB
            float c0x, c0y, c1x, c1y, c2x, c2y, c3x, c3y;
            AMR_MAG(0);
            AMR_MAG(1);
            AMR_ADD(c0x, L.hc[(g * 8 + R - 1) % R], m0.x); AMR_SB;                        // decode.go:234
            AMR_MAG(2);
            AMR_ADD(c0y, c0x, m0.y); AMR_SB;
            AMR_MAG(3);
            AMR_ADD(c1x, c0y, m1.x); AMR_SB;
            const v2f d0 = v2f{c0x, c0y} - AMR_HC(0); AMR_SB;                             // decode.go:242
            AMR_ADD(c1y, c1x, m1.y); AMR_SB;
            const v2f f0 = AMR_HD(0) - d0; AMR_SB;
            AMR_ADD(c2x, c1y, m2.x); AMR_SB;
            const v2f d1 = v2f{c1x, c1y} - AMR_HC(1); AMR_SB;
            AMR_ADD(c2y, c2x, m2.y); AMR_SB;
            AMR_BIT(f0.x); AMR_SB;
            AMR_ADD(c3x, c2y, m3.x); AMR_SB;
            const v2f f1 = AMR_HD(1) - d1; AMR_SB;
            AMR_ADD(c3y, c3x, m3.y); AMR_SB;
            AMR_BIT(f0.y); AMR_SB;
            const v2f d2 = v2f{c2x, c2y} - AMR_HC(2); AMR_SB;
            AMR_BIT(f1.x); AMR_SB;
            const v2f d3 = v2f{c3x, c3y} - AMR_HC(3); AMR_SB;
            const v2f f2 = AMR_HD(2) - d2; AMR_SB;
            AMR_BIT(f1.y); AMR_SB;
            const v2f f3 = AMR_HD(3) - d3; AMR_SB;
            AMR_BIT(f2.x); AMR_SB;
            AMR_BIT(f2.y); AMR_SB;
            AMR_BIT(f3.x); AMR_SB;
            AMR_BIT(f3.y); AMR_SB;
            const int r0 = g * 8;
            L.hc[r0 + 0] = c0x; L.hc[r0 + 1] = c0y; L.hc[r0 + 2] = c1x; L.hc[r0 + 3] = c1y;
            L.hc[r0 + 4] = c2x; L.hc[r0 + 5] = c2y; L.hc[r0 + 6] = c3x; L.hc[r0 + 7] = c3y;
            L.hd[r0 + 0] = d0.x; L.hd[r0 + 1] = d0.y; L.hd[r0 + 2] = d1.x; L.hd[r0 + 3] = d1.y;
            L.hd[r0 + 4] = d2.x; L.hd[r0 + 5] = d2.y; L.hd[r0 + 6] = d3.x; L.hd[r0 + 7] = d3.y;
#undef AMR_MAG
#undef AMR_SB
#undef AMR_ADD
#undef AMR_RO
#undef AMR_HC
#undef AMR_HD
#undef AMR_BIT
        }
#else
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#if AMR_K1_WAIT == 2
            if (k == 4) { __builtin_amdgcn_s_waitcnt(0xC17F); __builtin_amdgcn_sched_barrier(0); }
#endif
            constexpr int R = G::RING;
            const int r = g * 8 + k, rp = (r + R - 1) % R, ro = (r + 8) % R;
            float m = L.li[k] + L.lq[k];                       // decode.go:222
            if (PRO) m = (U.G * 8 + k < zlim) ? 0.0f : m;      // zero history, decode.go:144
            const float c = L.hc[rp] + m;                      // decode.go:234
            const float d = c - L.hc[ro];                      // csum[i+SL]-csum[i+CL]   (decode.go:242)
            const float f = L.hd[ro] - d;                      // (csum[i+CL]-csum[i]) - d (decode.go:242)
            L.acc = __builtin_amdgcn_alignbit(L.acc, __float_as_uint(f), 31);  // decode.go:243, inverted
            L.hc[r] = c;
            L.hd[r] = d;
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
#if AMR_K1_PIPE
#pragma unroll
        for (int k = 0; k < 8; ++k) { L.li[k] = nli[k]; L.lq[k] = nlq[k]; }
#endif
        L.row1 = row2;
#endif
        bool word_done;
        if (MODE == 0 && !CHECK) {
            word_done = U.wc == 0;
            U.wc = word_done ? 3 : U.wc - 1;
        } else {
            U.G += 1;
            U.og += 1;
            // Output i leaves the filter at step WARM-1+i, i.e. one step before a group boundary: at a
            // boundary acc holds outputs [32m+1 .. 32m+32]; output 32m is bit 0 of acc at the previous boundary.
            // WARM/8 is a multiple of 8, so word boundaries are the groups with og % 4 == 0.
            word_done = (U.og & 3) == 0 && U.og > 0;
            if (U.og == 0) L.prev = L.acc;
        }
        if (word_done) {
            const uint32_t word = ~__builtin_amdgcn_alignbit(L.prev, L.acc, 1);
#if AMR_K1_DIAG == 3
            L.xs ^= word * 0x9e3779b9u + U.wi;   // keep the computation alive without writing the bitstream
#endif
            if (G::NW <= 32 || U.wi < (uint32_t)G::NW0) L.ow0[U.wi] = word;
            else L.ow1[U.wi - G::NW0] = word;
            U.wi += 1;
            if (U.wi == (uint32_t)G::NW && AMR_K1_DIAG != 3) k1_flush<CL>(L, U, qrow, G::NW);
            L.prev = L.acc;
        }
    }
}

template <int CL, bool TAIL>
__global__ __launch_bounds__(64, 2) void k1_demod(const K1Args a)
{
    using G = K1Geom<CL>;
    __shared__ __attribute__((aligned(16))) uint8_t tiles[2 * kTileBuf];
    __shared__ __attribute__((aligned(16))) float lut[256];
    const uint32_t tiles_lds = (uint32_t)(uintptr_t)(lds_ptr_t)tiles;

    const uint32_t lane = threadIdx.x;
    k1_announce(a, lane);
    const uint32_t wg = a.wg_first + blockIdx.x;
    const uint32_t bs2 = a.block_size * 2;
    const uint32_t wpb = a.block_size >> 5;
    const uint32_t b = wg * kRows + lane;
    const uint32_t rows_valid = TAIL ? (a.n_blocks - wg * kRows) : kRows;

#pragma unroll
    for (int i = 0; i < 4; ++i) lut[lane + 64 * i] = a.lut[lane + 64 * i];

    // loader role: lane (rl, c') of load q fetches row 8q+rl, 16-byte column c'^((row>>1)&7)
    const uint32_t rl = lane >> 3;
    const uint32_t colx = (lane & 7) ^ (rl >> 1);
    const uint32_t voff_e = rl * bs2 + colx * 16;
    const uint32_t voff_o = rl * bs2 + (colx ^ 4) * 16;
    // consumer role: lane reads row `lane`, column gt, at the swizzled slot
    const uint32_t rdv = lane * kTileBytes | ((lane >> 1) & 7) * 16;
    // The lane stream starts HBA bytes before the block (whole cache lines); its first SKIP samples are not
    // part of the reference's window at all, so consumption simply starts at group SKIP/8 with all-zero rings.
    // Only stream block 0 of a fresh Decoder needs more: its SymbolLength history samples are magnitude 0.0.
    const bool fresh = a.zero_halo && wg == 0;
    const uint32_t zlim = (fresh && b == 0) ? G::WARM : G::SKIP;
    uint32_t *qrow = a.qt + (size_t)(wg + 1) * kRows * wpb;   // uniform: this wave-tile of the bitstream

    K1Lane<CL> L;
#pragma unroll
    for (int r = 0; r < G::RING; ++r) { L.hc[r] = 0.0f; L.hd[r] = 0.0f; }
    L.acc = 0;
    L.prev = 0;
    L.xs = 0;

    K1Uni U;
    U.G = G::SKIP / 8;
    U.og = -(G::WARM / 8) + G::SKIP / 8;
    U.ngroups = (G::HBA / 2 + a.block_size) / 8;
    U.ntiles = U.ngroups / 8;
    U.wi = 0;
    U.wdone = 0;
    U.st = 0;
    U.odd = __builtin_amdgcn_s_getreg(4 | (3 << 11)) & 1u;   // HW_REG_HW_ID.wave_id: slot of this wave in its SIMD
    if (U.odd) __builtin_amdgcn_s_setprio(1);

    // pipeline prologue: tile 0, the LUT values of group 0, the row of group 1
    U.tcur = 0;
    U.wc = 0;
    U.roff = (G::SKIP / 8) * 16;     // SKIP < 64: still inside tile 0
    k1_prefetch<CL, TAIL>(a, tiles_lds, wg, 0, 0, lane, voff_e, voff_o, rows_valid);
    if (G::SKIP != 0) {   // consumption starts inside tile 0: do here what k1_fetch_next does on a tile boundary
        U.tcur = 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (1 < U.ntiles && AMR_K1_DIAG != 1)
            k1_prefetch<CL, TAIL>(a, tiles_lds, wg, 1, kTileBuf, lane, voff_e, voff_o, rows_valid);
    }
#if !AMR_K1_PIPE
    L.row1 = k1_fetch_next<CL, TAIL>(U, a, tiles_lds, tiles, wg, lane, rdv, voff_e, voff_o, rows_valid);
#else
    {
        const uint4 row0 = k1_fetch_next<CL, TAIL>(U, a, tiles_lds, tiles, wg, lane, rdv, voff_e, voff_o, rows_valid);
        const uint32_t dw[4] = {row0.x, row0.y, row0.z, row0.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t v = dw[k >> 1] >> ((k & 1) * 16);
            L.li[k] = lut[v & 0xff];
            L.lq[k] = lut[(v >> 8) & 0xff];
        }
        L.row1 = k1_fetch_next<CL, TAIL>(U, a, tiles_lds, tiles, wg, lane, rdv, voff_e, voff_o, rows_valid);
    }
#endif

    // bodies: [predicate bodies of a fresh stream's wave 0] [warm-up bodies until output flows] [steady] [partial]
    const uint32_t nfull = (U.ngroups - G::SKIP / 8) / G::GPB;   // whole bodies; the rest goes through the checked body
    uint32_t body = 0;
    if (fresh)
        for (; body < (uint32_t)G::NPB && body < nfull; ++body)
            k1_body<CL, 2, false, TAIL>(L, U, a, tiles_lds, tiles, lut, wg, lane, rdv, zlim, voff_e, voff_o, rows_valid, qrow);
    for (; body < nfull && U.og <= 0; ++body)
        k1_body<CL, 1, false, TAIL>(L, U, a, tiles_lds, tiles, lut, wg, lane, rdv, zlim, voff_e, voff_o, rows_valid, qrow);
    U.wc = 3 - ((uint32_t)U.og & 3);          // og > 0 here (or no steady body runs): next word after 4 - og%4 groups
    for (; body < nfull; ++body)
        k1_body<CL, 0, false, TAIL>(L, U, a, tiles_lds, tiles, lut, wg, lane, rdv, zlim, voff_e, voff_o, rows_valid, qrow);
    {   // the steady bodies did not count groups: re-derive the counters the checked body needs
        const uint32_t done = G::SKIP / 8 + body * G::GPB;
        U.og += (int32_t)(done - U.G);
        U.G = done;
    }
    if (U.G < U.ngroups)
        k1_body<CL, 1, true, TAIL>(L, U, a, tiles_lds, tiles, lut, wg, lane, rdv, zlim, voff_e, voff_o, rows_valid, qrow);
    if (U.wi && AMR_K1_DIAG != 3) k1_flush<CL>(L, U, qrow, U.wi);
    if (AMR_K1_DIAG == 3) qrow[lane * 4] = L.xs;
}

}  // namespace amr
