// K1 -- MagLUT.Execute + Decoder.Filter + pack (protocol/decode.go:219-245; one lane = one reference block, same operation
// order and roundings), organised around what bounded the first-generation kernel (rounds 1-2; in the history) on MI355X: with two waves per SIMD nothing hides a stall, so the kernel
//   1. keeps the staging tile in REGISTERS.  A tile (64 rows x 128 B) that has landed in LDS is drained into 32 VGPRs in
//      one burst at the tile boundary (the same eight ds_read_b128 the old kernel spread over the tile), which frees its
//      LDS buffer a whole tile-time early: the DMA of the tile after next goes out at once and is in flight during the
//      whole computation of the next tile, with ONE 8 KiB buffer per wave instead of two (two buffers with two tiles
//      in flight, DEPTH 2, turned out slower: the memory side alone runs at 0.205 ms with 32 MiB in flight chip-wide
//      against 0.164 ms with 16 MiB).  The LDS saved parks finished output chunks (NLC) for larger store bursts.
//   2. runs a fully static instruction stream in steady state.  One "super-body" = lcm(RING, 64) samples covers a
//      whole number of csum-ring turns AND of staging tiles (5 tiles / 4 turns at chip length 72), so ring slots,
//      tile registers and word boundaries are all compile-time constants: no per-group scalar bookkeeping, no
//      per-group branches, three instructions per DMA piece.
//   3. lets the two waves of a SIMD swap s_setprio every 8 tiles (PRIO below): the arbiter's preference for the older
//      wave made it finish 9 % earlier and leave the other one alone for the rest of the launch.
// Output layout, arguments and the halo/carry conventions: k1_common.h (K1Args, "tiled4" bitstream).
#pragma once
#include "k1_common.h"
#include "k1_search.h"

// Developer diagnostics (K1TCfg::DIAG, 0 in the product): 1 = no HBM traffic after the prologue (arithmetic side
// alone), 2 = staging + drains only (memory side alone), 3 = no output stores, 5 = no LUT gathers (one cheap ALU op per
// byte instead: wrong values, timing only), 6 = gathers but no filter arithmetic, 7 = output stores confined to a 4 MiB
// window (they stay in the L2 / Infinity Cache).
// Tuning knobs are template parameters (struct K1TCfg) so that one binary can hold several variants and compare them on
// the same buffer (tools/k1_bench.hip); the product instantiates K1TDefault only.
// harness only (AMR_K1T_CLK, k1_common.h): lane 0 of workgroup 0 leaves (shader clock ticks, 100 MHz real-time ticks) of its run in qt[0..3]
#ifndef AMR_K1T_LUT_DMA
#define AMR_K1T_LUT_DMA 1     // 0: the round-2 table fill (global load + ds_write in front of the first tile), for A/B builds
#endif

// chip length 80: rings of 88 slots (a super-body of 11 tiles).  Rounds 2-4 used 96 (3 tiles) and paid for the 16 extra slots
// with a third of the hd ring in LDS and four store bursts per block; equal in the harness, 6 % of K1 in the pipelined
// product: 0.210-0.213 -> 0.199-0.201 ms (profiles/r05/chip_80_96/k1_long_chips.txt).
#ifndef AMR_K1T_RING80
#define AMR_K1T_RING80 88
#endif
#ifndef AMR_K1T_RING_OVR
#define AMR_K1T_RING_OVR 0      // harness builds: any multiple of 8 above the chip length, for every chip length of the binary
#endif

namespace amr {

#if AMR_K1T_CLK
// diagnostic builds only: per workgroup of the last kK1TLaunches launches (K1Args::tl_seq picks the slot): 100 MHz start
// tick, end tick, (XCC_ID << 32 | HW_ID)
#ifndef AMR_K1T_WGS
#define AMR_K1T_WGS 2048    // workgroups stamped per launch (launches of several rounds: raise it)
#endif
constexpr int kK1TLaunches = 64, kK1TWgs = AMR_K1T_WGS;
__device__ unsigned long long k1t_timeline[kK1TLaunches][kK1TWgs][3];
#endif

// SCHED  instruction order inside a tile: 0 = per group: 16 gathers, then the 8 samples' arithmetic; 1 = per half
//        group, gathers one half ahead of their use (same 16 value registers).
// DEPTH  staging tiles in flight per wave: 2 = two LDS buffers, every DMA has two tile-times to land; 1 = one LDS
//        buffer (9 KiB of LDS per wave instead of 17), the DMA of tile t+2 goes out as soon as tile t+1 sits in registers.
// XCD    1: workgroup b handles wave-tile (b % 8) * (grid / 8) + b / 8, i.e. every XCD (b % 8) streams one contiguous
//        eighth of the batch instead of every eighth wave-tile.
// NW     output words held in registers between store bursts.
// STPOL  cache policy of the bitstream stores: 0 default (write-back in L2), 1 sc1, 2 sc0 sc1, 3 nt, 4 sc1 nt, 5 sc0
// STORE_AFTER  (depth 1) 1: the store burst is issued behind the boundary's DMA and not waited for
// NLC    finished output chunks (4 words per lane = 1 KiB per wave) parked in LDS between store bursts, on top of the
//        NW words kept in registers and the 4-word staging chunk: a burst is 4 * (NLC + 1) + NW words per lane.
//        Output stores mixed into the read stream cost far more than their bytes, and the cost goes with the number
//        of bursts (tools/sst_bench.hip: 64 MiB in 8 / 4 / 2 / 1 chip-wide bursts: +30 / +23 / +15 / +8 %).
// PRIO   the two waves of a SIMD do not share it fairly: the arbiter prefers the older wave, which then finishes ~9 %
//        earlier and leaves the younger one alone (no latency hiding, half the memory parallelism) for the rest of the
//        launch.  1: the wave in the odd hardware slot runs at priority 1 throughout; 2 / 3 / 4: the two waves swap
//        priority 0 / 1 every tile / 4 tiles / 16 tiles (the wave in the odd slot starts high).  5: priority 2 from the
//        wait for the next tile to the issue of the following DMA (the memory-critical stretch), 0 otherwise; 6 = 5 on
//        top of the per-tile swap of 2 (levels 0 / 1, boundary 3).  10 + k: swap every 2^k tiles.
// HDL    slots of the hd ring that live in LDS instead of registers (chip length 80 .. 96: two rings of 88 .. 104 slots do
//        not fit 248 VGPRs next to the register tile; a value of the ring is written once and read once, chip-length steps
//        later, so part of the ring -- the slots at the ring's upper end -- goes through LDS: two ds_write_b128 when
//        a group of 8 is produced, two ds_read_b128 one ring turn minus a chip later, fetched next to the LUT gathers).
//        Costs parking space: chip 88's 8 KiB of hd leave 3 KiB for output chunks, i.e. four store bursts per 4096-sample
//        block, chip 96's 10 KiB leave one (eight bursts).
// HS     (round 6) halo shift, for chip lengths whose halo is ONE staging tile (4 * chip length <= 128: chip 8 .. 32).  The halo
//        tile of row r is the LAST tile of row r - 1 -- and lane r - 1 of the same wave walks that row: every wave read 64 lines
//        twice, at its start as halos and at its end as last tiles (chip 8: 9 tiles per 1 KiB row, 12.5 % of the traffic).
//        1: a lane keeps its halo tile in 32 registers, and the wave's last tile comes from them -- lane r takes lane r + 1's
//        (v_mov_b32 dpp wave_shl:1), only rows 56 .. 63 are fetched again (one DMA piece of eight) for lane 63.
template <int SCHED_, int DEPTH_, int XCD_, int NW_, int DIAG_ = 0, int STPOL_ = 0, int STORE_AFTER_ = 0, int NLC_ = 0, int PRIO_ = 0, int HDL_ = 0, int HS_ = 0>
struct K1TCfg {
    static constexpr int SCHED = SCHED_, DEPTH = DEPTH_, XCD = XCD_, NW = NW_, DIAG = DIAG_, STPOL = STPOL_, STORE_AFTER = STORE_AFTER_, NLC = NLC_, PRIO = PRIO_, HDL = HDL_, HS = HS_;
    static constexpr int CAP = NLC_ + NW_ / 4 + 1;               // chunks per full burst (LDS, registers, staging)
    static constexpr uint32_t kLut = DEPTH_ * kTileBuf;          // LDS byte offset of the LUT behind the tile buffer(s)
    static constexpr uint32_t kPark = DEPTH_ * kTileBuf + 1024;  // parked output chunks: chunk c of lane l at kPark + c * 1024 + l * 16
    static constexpr uint32_t kHd = kPark + NLC_ * 1024;         // hd slots in LDS: slots 4j..4j+3 of lane l at kHd + j * 1024 + l * 16
    static constexpr uint32_t kLds = kHd + HDL_ * 256;           // dynamic LDS bytes per workgroup
    static constexpr uint32_t kFlip = DEPTH_ == 2 ? kTileBuf : 0; // toggles U.par between the buffers
    static_assert(HDL_ % 8 == 0 && kLds <= 20 * 1024, "20 KiB of LDS per wave at 8 waves per CU");
};
typedef K1TCfg<0, 1, 1, 16, 0, 1, 1, 11, 13> K1TDefault;            // chip length <= 72
typedef K1TCfg<0, 1, 1, 16, 0, 1, 1, 3, 13, 32> K1TLongChip;        // chip length 88: rings of 96, 32 hd slots in LDS
// chip length 96: rings of 104 (a super-body of 13 tiles), 40 hd slots in LDS, 8 output words in registers and one chunk
// parked: 247 registers.  (NW 8, NLC 3, HDL 32 is 3 % faster alone -- 24-word store bursts instead of 16 -- but takes 256
// registers, and a K1 wave of more than 248 leaves no room for the tail's gate kernel next to two of them:
// amr_pipeline.hip, submit.)
#ifndef AMR_K1T_96
#define AMR_K1T_96 8, 0, 1, 1, 1, 13, 40       // NW, DIAG, STPOL, STORE_AFTER, NLC, PRIO, HDL (harness / variant builds override)
#endif
typedef K1TCfg<0, 1, 1, AMR_K1T_96> K1TChip96;
// chip length 80: 8 hd slots in LDS, 9 parked chunks: store bursts of 48 words (241 registers)
#ifndef AMR_K1T_80
#define AMR_K1T_80 8, 0, 1, 1, 9, 13, 8
#endif
typedef K1TCfg<0, 1, 1, AMR_K1T_80> K1TChip80;
#ifndef AMR_K1T_HS
#define AMR_K1T_HS 1      // 0: chip 8 without the halo shift (A/B builds)
#endif
typedef K1TCfg<0, 1, 1, 16, 0, 1, 1, 11, 13, 0, AMR_K1T_HS> K1TChip8;   // chip length 8: rows of 1 KiB, the halo tile is 12.5 % of them
template <int CL> struct K1TCfgFor { typedef K1TDefault type; };
template <> struct K1TCfgFor<8> { typedef K1TChip8 type; };
#ifndef AMR_K1T_HS32
#define AMR_K1T_HS32 1      // 0: chip 32 without the halo shift (A/B builds)
#endif
typedef K1TCfg<0, 1, 1, 16, 0, 1, 1, 11, 13, 0, AMR_K1T_HS32> K1TChip32;   // chip length 32: the halo tile is 3 % of a 4 KiB row
template <> struct K1TCfgFor<32> { typedef K1TChip32 type; };
template <> struct K1TCfgFor<80> { typedef K1TChip80 type; };
template <> struct K1TCfgFor<88> { typedef K1TLongChip type; };
template <> struct K1TCfgFor<96> { typedef K1TChip96 type; };

constexpr int k1t_gcd(int a, int b) { return b == 0 ? a : k1t_gcd(b, a % b); }

template <int CL>
struct K1TGeom {
    static constexpr int SL = 2 * CL;
    static constexpr int HB = 4 * CL;
    static constexpr int HBA = (HB + 127) & ~127;
    static constexpr int SKIP = (HBA - HB) / 2;        // leading stream samples outside the reference's window
    static constexpr int WARM = HBA / 2;               // = SKIP + SL: first sample of the block proper, tile aligned
    static constexpr int NPT = HBA / kTileBytes;       // halo tiles
    // csum rings: slot t % RING is written at step t and holds c[t] / d[t] until step t + CL reads it.  RING is the
    // smallest multiple of 8 above CL (chip 64: 80, not 72 -> a super-body of 5 tiles instead of 9; kept from round 2).
    static constexpr int RING = AMR_K1T_RING_OVR ? AMR_K1T_RING_OVR : (CL == 64) ? 80 : (CL == 80) ? AMR_K1T_RING80 : CL + 8;
    static constexpr int SPB = RING / k1t_gcd(RING, 64) * 64;   // samples per super-body
    static constexpr int TPS = SPB / 64;                         // tiles per super-body
    // (every legal chip length, flags.go:127-132, has a configuration: K1TCfgFor.  Round 2 stopped at 7 tiles per super-body
    // for fear of the instruction footprint; chip 96 runs 13 and is no slower per sample than chip 88 with 3.)
};

typedef const __attribute__((address_space(3))) float *k1t_lds_f;
typedef uint32_t k1t_v4u __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) k1t_v4u *k1t_lds_u4;

template <int CL, class C>
struct K1TLane {
    using G = K1TGeom<CL>;
    float hc[G::RING];   // hc[t % RING] = c[t], running sum after sample t (decode.go:234)
    float hd[G::RING - C::HDL > 0 ? G::RING - C::HDL : 1];   // hd[t % RING] = c[t] - c[t-CL]; slots >= RING - HDL live in LDS
    float hdt[C::HDL ? 8 : 1];   // the eight hd values of the current group when they come from LDS
    uint32_t tl[32];     // the staging tile of this lane's row: 128 B = 64 IQ samples
    uint32_t keep[C::HS ? 32 : 1];   // HS: the row's halo tile = the last tile of the row before it
    float lv[16];        // LUT values of 8 samples (lut[I], lut[Q] interleaved)
    uint32_t acc, prev;  // sign bits of f, newest in bit 0 (inverted decisions); acc at the last word boundary
    uint32_t xs;
    typedef uint32_t ow_t __attribute__((ext_vector_type(C::NW)));
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    ow_t ow;             // finished chunks kept in registers (dynamic, wave-uniform index -> s_set_gpr_idx)
    v4u st4;             // the chunk being filled
};

struct K1TUni {
    uint32_t t;        // tile being computed
    uint32_t ntiles;
    uint32_t par;      // LDS byte offset of the buffer holding tile t+1 (= the one tile t+3 goes to)
    uint32_t wi;       // words in the staging chunk (0..3)
    uint32_t nch;      // finished chunks buffered (LDS first, then registers, the last one stays in staging)
    uint32_t wdone;    // words stored
    uint32_t st;       // depth 1: 1 = a store burst was issued after the DMA in flight
    uint32_t odd;      // PRIO: 1 = this wave sits in an odd hardware wave slot of its SIMD
    uint32_t hold_last; // in-wave search: the burst that completes at the LAST tile boundary stays on the chip (the epilogue searches, then stores)
};

// LUT gathers of `n` samples starting at sample s0 (0..63) of the register tile: lv[2j], lv[2j+1] = lut[I], lut[Q]
template <int CL, class C>
__device__ __forceinline__ void k1t_gather(K1TLane<CL, C> &L, int s0, int n, int lv0)
{
#pragma unroll
    for (int j = 0; j < n; ++j) {
        const int s = s0 + j;
        const uint32_t v = L.tl[s >> 1] >> ((s & 1) * 16);
        if constexpr (C::DIAG == 5) {
            L.lv[lv0 + 2 * j] = __uint_as_float(((v & 0xff) << 15) | 0x3c000000u);
            L.lv[lv0 + 2 * j + 1] = __uint_as_float((((v >> 8) & 0xff) << 15) | 0x3c000000u);
        } else {
            k1t_lds_f lut = (k1t_lds_f)(uintptr_t)C::kLut;
            L.lv[lv0 + 2 * j] = lut[v & 0xff];                 // decode.go:222
            L.lv[lv0 + 2 * j + 1] = lut[(v >> 8) & 0xff];
        }
    }
}

// The filter arithmetic of `n` samples whose LUT values sit in lv[lv0..]; rb = ring slot of the first sample;
// PRED: magnitude forced to 0.0 below stream sample zlim (fresh Decoder, decode.go:144); sidx = stream sample index.
template <int CL, class C, bool PRED>
__device__ __forceinline__ void k1t_arith(K1TLane<CL, C> &L, int rb, int n, int lv0, uint32_t sidx, uint32_t zlim)
{
    constexpr int R = K1TGeom<CL>::RING;
    if constexpr (C::DIAG == 6) {   // keep the gathered values alive, nothing else
#pragma unroll
        for (int j = 0; j < 2 * n; ++j) L.acc ^= __float_as_uint(L.lv[lv0 + j]);
        return;
    }
    constexpr int RR = R - C::HDL;                                     // hd slots below RR are registers, the rest LDS
    float dn[8];                                                       // the group's new d values bound for LDS
#pragma unroll
    for (int j = 0; j < n; ++j) {
        const int r = (rb + j) % R, rp = (r + R - 1) % R, ro = (r + R - CL) % R;
        float m = L.lv[lv0 + 2 * j] + L.lv[lv0 + 2 * j + 1];           // decode.go:222
        if (PRED) m = (sidx + j < zlim) ? 0.0f : m;
        const float c = L.hc[rp] + m;                                  // decode.go:234
        const float d = c - L.hc[ro];                                  // csum[i+SL]-csum[i+CL]   (decode.go:242)
        const float dold = (C::HDL && ro >= RR) ? L.hdt[j] : L.hd[ro < RR ? ro : 0];   // csum[i+CL]-csum[i], one ring turn old
        const float f = dold - d;                                      // (csum[i+CL]-csum[i]) - d
        L.acc = __builtin_amdgcn_alignbit(L.acc, __float_as_uint(f), 31);   // decode.go:243, inverted
        L.hc[r] = c;
        if (C::HDL && r >= RR) dn[j] = d; else L.hd[r < RR ? r : 0] = d;
    }
    if constexpr (C::HDL > 0) {
        // groups are 8-aligned and so is the LDS part of the ring: a group's eight new values go out together
        if (n == 8 && rb % R >= RR) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) v4f *lds_v4f;
            const uint32_t base = C::kHd + (uint32_t)((rb % R - RR) / 4) * 1024 + threadIdx.x * 16;
            *(lds_v4f)(uintptr_t)base = v4f{dn[0], dn[1], dn[2], dn[3]};
            *(lds_v4f)(uintptr_t)(base + 1024) = v4f{dn[4], dn[5], dn[6], dn[7]};
        }
    }
}

// the eight hd values the group at ring slot rb will read (slots rb - CL ...), when they live in LDS: issued next to the
// group's LUT gathers, consumed by k1t_arith
template <int CL, class C>
__device__ __forceinline__ void k1t_hd_fetch(K1TLane<CL, C> &L, int rb)
{
    if constexpr (C::HDL > 0) {
        constexpr int R = K1TGeom<CL>::RING, RR = R - C::HDL;
        const int ro = (rb % R + R - CL) % R;
        if (ro >= RR) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(3))) v4f *lds_v4f;
            const uint32_t base = C::kHd + (uint32_t)((ro - RR) / 4) * 1024 + threadIdx.x * 16;
            const v4f a = *(lds_v4f)(uintptr_t)base, b = *(lds_v4f)(uintptr_t)(base + 1024);
            L.hdt[0] = a.x; L.hdt[1] = a.y; L.hdt[2] = a.z; L.hdt[3] = a.w;
            L.hdt[4] = b.x; L.hdt[5] = b.y; L.hdt[6] = b.z; L.hdt[7] = b.w;
        }
    }
}

// Program-order pins.  Inside one basic block hipcc's instruction selection is free to move pure arithmetic across
// sched_barrier (it hoisted every gather of a tile above all of the tile's arithmetic: 128 values live, spills).
// An empty volatile asm that takes a value of the stage before it and clobbers memory fixes the order: loads cannot
// cross it, and what consumes its operand cannot be issued before it.
__device__ __forceinline__ void k1t_pin(uint32_t &x)
{
    asm volatile("" : "+v"(x) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int CL, class C>
__device__ __forceinline__ void k1t_pin8(K1TLane<CL, C> &L, int lv0)
{
    asm volatile("" : "+v"(L.lv[lv0]), "+v"(L.lv[lv0 + 1]), "+v"(L.lv[lv0 + 2]), "+v"(L.lv[lv0 + 3]), "+v"(L.lv[lv0 + 4]),
                      "+v"(L.lv[lv0 + 5]), "+v"(L.lv[lv0 + 6]), "+v"(L.lv[lv0 + 7]) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Drain the landed tile in the buffer at LDS offset `par` into the tile registers: column c of row `lane` sits in
// slot c ^ ((lane>>1)&7) (source-side swizzle, bank-conflict-free row-per-lane reads).
template <int CL, class C>
__device__ __forceinline__ void k1t_drain(K1TLane<CL, C> &L, uint32_t rdv, uint32_t par)
{
    const uint32_t rdp = rdv + par;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        uint32_t ad;   // asm: computed here, not hoisted into eight address registers kept alive across the loop
        asm volatile("v_xor_b32 %0, %2, %1" : "=v"(ad) : "v"(rdp), "n"(c * 16));
        const k1t_v4u v = *(k1t_lds_u4)(uintptr_t)ad;
        L.tl[4 * c] = v.x; L.tl[4 * c + 1] = v.y; L.tl[4 * c + 2] = v.z; L.tl[4 * c + 3] = v.w;
    }
}

template <class C, class V>
__device__ __forceinline__ void k1t_store(V *p, V x)
{
    if constexpr (C::STPOL == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
    else if constexpr (C::STPOL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(x) : "memory");
    else if constexpr (C::STPOL == 3) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(x) : "memory");
    else if constexpr (C::STPOL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(x) : "memory");
    else if constexpr (C::STPOL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(x) : "memory");
    else *p = x;
}

template <int CL, class C, int J>
__device__ __forceinline__ void k1t_flush_regs(const K1TLane<CL, C> &L, typename K1TLane<CL, C>::v4u *dst, uint32_t n)
{
    if constexpr (4 * J < C::NW) {
        if ((uint32_t)(C::NLC + J) < n) {
            typename K1TLane<CL, C>::v4u x = {L.ow[4 * J], L.ow[4 * J + 1], L.ow[4 * J + 2], L.ow[4 * J + 3]};
            k1t_store<C>(&dst[(C::NLC + J) * kRows], x);
        }
        k1t_flush_regs<CL, C, J + 1>(L, dst, n);
    }
}

// Write the U.nch buffered chunks: chunk k of this burst holds words [wdone + 4k, wdone + 4k + 4) of row `lane`, i.e.
//   qt[(wg+1)*64*WPB + ((wdone>>2) + k)*256 + lane*4 .. +4]   ("tiled4": a lane stores 16 bytes, a wave 1 KiB, per instruction).
template <int CL, class C>
__device__ __forceinline__ void k1t_flush(K1TLane<CL, C> &L, K1TUni &U, uint32_t *qbase)
{
    typedef typename K1TLane<CL, C>::v4u v4u;
    typedef const __attribute__((address_space(3))) v4u *lds_v4;
    const uint32_t n = U.nch;
    uint8_t *ub = reinterpret_cast<uint8_t *>(qbase) + (size_t)(U.wdone >> 2) * (kRows * 16);
    v4u *dst = reinterpret_cast<v4u *>(ub + (uint32_t)(threadIdx.x * 16));
    if constexpr (C::NLC > 0) {
        const uint32_t nl = n < (uint32_t)C::NLC ? n : (uint32_t)C::NLC;
#pragma unroll 1
        for (uint32_t c = 0; c < nl; ++c) {       // rolled: a burst happens twice per block, registers matter more
            const v4u x = *(lds_v4)(uintptr_t)(C::kPark + c * 1024 + threadIdx.x * 16);
            k1t_store<C>(&dst[c * kRows], x);
        }
    }
    k1t_flush_regs<CL, C, 0>(L, dst, n);
    if (n == (uint32_t)C::CAP) k1t_store<C>(&dst[(C::CAP - 1) * kRows], L.st4);
    U.wdone += 4 * n;
    U.nch = 0;
}

// One finished output word: at a word boundary acc holds outputs [32m+1 .. 32m+32] (output i leaves the filter one
// step before a group boundary), output 32m is bit 0 of acc at the previous boundary.  Every fourth word closes a
// chunk, which moves to LDS, then to the register vector; the last chunk of a burst stays in the staging registers.
template <int CL, class C>
__device__ __forceinline__ void k1t_word(K1TLane<CL, C> &L, K1TUni &U)
{
    typedef typename K1TLane<CL, C>::v4u v4u;
    typedef __attribute__((address_space(3))) v4u *lds_v4w;
    const uint32_t word = ~__builtin_amdgcn_alignbit(L.prev, L.acc, 1);
    if (C::DIAG == 3) L.xs ^= word * 0x9e3779b9u + U.wi;
    L.st4[U.wi] = word;
    U.wi += 1;
    L.prev = L.acc;
    if (U.wi == 4) {
        U.wi = 0;
        if (C::NLC > 0 && U.nch < (uint32_t)C::NLC) {
            *(lds_v4w)(uintptr_t)(C::kPark + U.nch * 1024 + threadIdx.x * 16) = L.st4;
        } else if (U.nch < (uint32_t)(C::CAP - 1)) {
            const uint32_t j = (U.nch - C::NLC) * 4;
            L.ow[j] = L.st4.x; L.ow[j + 1] = L.st4.y; L.ow[j + 2] = L.st4.z; L.ow[j + 3] = L.st4.w;
        }
        U.nch += 1;
    }
}

// Steady-state DMA of one tile into the buffer at LDS offset `par`: eight pieces of 8 rows, SGPR base per piece
// (constant per wave), the tile's byte offset inside the lane offsets vt_e / vt_o (the instruction's immediate offset
// would also shift the LDS destination, so it stays 0), M0 = par + piece * 1 KiB.
// Issued from inline asm so that hipcc does not guard later LDS reads with vmcnt(0) (see k1_prefetch, k1_common.h).
__device__ __forceinline__ void k1t_dma_fast(const uint8_t *const (&sb)[8], uint32_t vt_e, uint32_t vt_o, uint32_t par)
{
    par = __builtin_amdgcn_readfirstlane(par);   // wave-uniform by construction; makes it an SGPR for the asm operand
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 " AMR_K1_LDFLAGS
                     :: "v"((q & 1) ? vt_o : vt_e), "s"(sb[q]), "s"(par), "n"(q * 1024)
                     : "memory", "scc");
    }
}

// HS: the wave's LAST tile -- only the piece that holds row 63 (rows 56 .. 63); the other lanes take theirs from a neighbour's registers
__device__ __forceinline__ void k1t_dma_last(const uint8_t *const (&sb)[8], uint32_t vt_o, uint32_t par)
{
    par = __builtin_amdgcn_readfirstlane(par);
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 " AMR_K1_LDFLAGS
                 :: "v"(vt_o), "s"(sb[7]), "s"(par), "n"(7 * 1024)
                 : "memory", "scc");
}

// HS: the last tile out of the halo tiles the lanes kept: lane r <- lane r + 1; lane 63 keeps what it drained from LDS (its own row)
template <int CL, class C>
__device__ __forceinline__ void k1t_halo_shift(K1TLane<CL, C> &L)
{
#pragma unroll
    for (int k = 0; k < 32; ++k)
        L.tl[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)L.tl[k], (int)L.keep[k], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

// Wait for the DMA of tile t+1.  Outstanding VMEM in issue order: DMA(t+1), the store burst of the previous boundary,
// then (depth 2) DMA(t+2): vmcnt(8) leaves at most that tile's eight pieces in flight.  vmcnt retires in order.
// Depth 1: the store burst goes out AFTER the DMA of the same boundary, so the wait skips it (vmcnt(NW/4)): a store is
// acknowledged later than a load returns, and in-order retirement would otherwise make the tile wait for it.
template <class C>
__device__ __forceinline__ void k1t_wait_tile(const K1TUni &U)
{
    if (C::DEPTH == 2 && U.t + 2 < U.ntiles) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (C::DEPTH == 1 && C::STORE_AFTER && U.st) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(C::CAP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// One tile (64 samples) of one lane-row: tile registers -> LUT gathers -> filter arithmetic, and at its end the
// tile boundary: drain tile t+1 (landed two tile-times after its DMA went out), store burst if due, DMA of tile t+3
// into the buffer just drained.  TI = tile index inside the super-body (steady) or the halo tile number (warm-up).
// Returns true when the lane stream is finished.
template <int CL, bool TAIL, class C, bool WARMUP, int TI>
__device__ __forceinline__ bool k1t_tile(K1TLane<CL, C> &L, K1TUni &U, const K1Args &a, const uint8_t *const (&sb)[8],
                                         uint32_t wg, uint32_t lane, uint32_t rdv, uint32_t zlim, uint32_t voff_e,
                                         uint32_t voff_o, uint32_t &vt_e, uint32_t &vt_o, uint32_t rows_valid, uint32_t *qrow)
{
    using G = K1TGeom<CL>;
    constexpr int R = G::RING;
    // ring slot of the tile's first sample: stream sample index mod RING; steady super-bodies start at WARM
    constexpr int RB0 = WARMUP ? (TI * 64) % R : (G::WARM + TI * 64) % R;
    const bool more = U.t + 1 < U.ntiles;
    if constexpr (C::DIAG == 2) {
#pragma unroll
    for (int k = 0; k < 32; ++k) L.acc ^= L.tl[k];
    if (more) {
        k1t_wait_tile<C>(U);
        k1t_drain<CL, C>(L, rdv, U.par);
    }
    } else if constexpr (C::SCHED == 0) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const bool skip = WARMUP && (TI * 64 + g * 8 + 8 <= G::SKIP);     // static: samples before the reference's window
        if (!skip) k1t_gather<CL, C>(L, g * 8, 8, 0);
        if (!skip) k1t_hd_fetch<CL, C>(L, (RB0 + g * 8) % R);
        if (g == 7 && more) {
            // every tile register is dead now.  Outstanding VMEM in issue order: DMA(t+1), the store burst of the
            // previous boundary, DMA(t+2): vmcnt(8) leaves at most DMA(t+2)'s eight pieces in flight.
            if constexpr (C::PRIO == 5) __builtin_amdgcn_s_setprio(2);
            if constexpr (C::PRIO == 6) __builtin_amdgcn_s_setprio(3);
            k1t_wait_tile<C>(U);
            k1t_drain<CL, C>(L, rdv, U.par);
            if (C::HS && !TAIL && U.t + 2 == U.ntiles) k1t_halo_shift<CL, C>(L);      // (wave-uniform)
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!skip) k1t_arith<CL, C, WARMUP>(L, (RB0 + g * 8) % R, 8, 0, TI * 64 + g * 8, zlim);
        k1t_pin(L.acc);
        if (!WARMUP && (g == 3 || g == 7)) k1t_word<CL, C>(L, U);
    }
    } else {
    // half groups, gathers one half ahead: lv[0..7] and lv[8..15] alternate.  Half 0's values were gathered in the
    // last slot of the previous tile (or in the prologue).
#pragma unroll
    for (int h = 0; h < 16; ++h) {
        const bool skip = WARMUP && (TI * 64 + h * 4 + 4 <= G::SKIP);
        const bool skipn = WARMUP && (TI * 64 + h * 4 + 8 <= G::SKIP);
        if (h < 15 && !skipn) k1t_gather<CL, C>(L, (h + 1) * 4, 4, ((h + 1) & 1) * 8);
        if (h == 14 && more) {
            k1t_wait_tile<C>(U);
            k1t_drain<CL, C>(L, rdv, U.par);
        }
        if (h == 15) k1t_gather<CL, C>(L, 0, 4, 0);                          // first half of the next tile
        if (!skip) k1t_pin8<CL, C>(L, (h & 1) * 8);                          // the values gathered one half earlier
        if (!skip) k1t_arith<CL, C, WARMUP>(L, (RB0 + h * 4) % R, 4, (h & 1) * 8, TI * 64 + h * 4, zlim);
        k1t_pin(L.acc);
        if (!WARMUP && (h == 7 || h == 15)) k1t_word<CL, C>(L, U);
    }
    }
    // tile boundary, second part
    if (C::DIAG == 3 && U.nch == (uint32_t)C::CAP) U.nch = 0;
    constexpr bool kStoreAfter = C::DEPTH == 1 && C::STORE_AFTER;
    const bool held = U.hold_last && U.t + 1 >= U.ntiles;              // (wave-uniform; hold_last is a constant 0 without the in-wave search)
    if (!kStoreAfter && !WARMUP && U.nch == (uint32_t)C::CAP && C::DIAG != 3 && !held) k1t_flush<CL, C>(L, U, qrow);
    if (U.t + 1 + C::DEPTH < U.ntiles && C::DIAG != 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the drain has left the buffer
        if (TAIL || WARMUP)
            k1_prefetch<CL, TAIL>(a, 0, wg, U.t + 1 + C::DEPTH, U.par, lane, voff_e, voff_o, rows_valid);
        else if (C::HS && U.t + 2 + C::DEPTH == U.ntiles)
            k1t_dma_last(sb, vt_o, U.par);
        else
            k1t_dma_fast(sb, vt_e, vt_o, U.par);
    }
    if (kStoreAfter) {
        U.st = 0;
        if (!WARMUP && U.nch == (uint32_t)C::CAP && C::DIAG != 3 && !held) { k1t_flush<CL, C>(L, U, qrow); U.st = 1; }
    }
    if constexpr (C::PRIO == 5) __builtin_amdgcn_s_setprio(0);
    if constexpr ((C::PRIO >= 2 && C::PRIO <= 4) || C::PRIO == 6 || C::PRIO >= 10) {
        constexpr int SH = C::PRIO >= 10 ? C::PRIO - 10 : (C::PRIO == 2 || C::PRIO == 6) ? 0 : C::PRIO == 3 ? 2 : 4;
        if ((((U.t + 1) >> SH) ^ U.odd) & 1) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
    }
    vt_e += kTileBytes;   // lane offsets of the tile the next boundary fetches
    vt_o += kTileBytes;
    U.par ^= C::kFlip;
    U.t += 1;
    return U.t >= U.ntiles;
}

template <int CL, bool TAIL, class C, int TI>
__device__ __forceinline__ bool k1t_super(K1TLane<CL, C> &L, K1TUni &U, const K1Args &a, const uint8_t *const (&sb)[8],
                                          uint32_t wg, uint32_t lane, uint32_t rdv, uint32_t voff_e, uint32_t voff_o,
                                          uint32_t &vt_e, uint32_t &vt_o, uint32_t rows_valid, uint32_t *qrow)
{
    if constexpr (TI < K1TGeom<CL>::TPS) {
        if (k1t_tile<CL, TAIL, C, false, TI>(L, U, a, sb, wg, lane, rdv, 0, voff_e, voff_o, vt_e, vt_o, rows_valid, qrow))
            return true;
        return k1t_super<CL, TAIL, C, TI + 1>(L, U, a, sb, wg, lane, rdv, voff_e, voff_o, vt_e, vt_o, rows_valid, qrow);
    } else {
        return false;
    }
}

template <int CL, bool TAIL, class C, int TI>
__device__ __forceinline__ void k1t_warm(K1TLane<CL, C> &L, K1TUni &U, const K1Args &a, const uint8_t *const (&sb)[8],
                                         uint32_t wg, uint32_t lane, uint32_t rdv, uint32_t zlim, uint32_t voff_e,
                                         uint32_t voff_o, uint32_t rows_valid)
{
    if constexpr (TI < K1TGeom<CL>::NPT) {
        uint32_t d0 = 0, d1 = 0;
        k1t_tile<CL, TAIL, C, true, TI>(L, U, a, sb, wg, lane, rdv, zlim, voff_e, voff_o, d0, d1, rows_valid, nullptr);
        k1t_warm<CL, TAIL, C, TI + 1>(L, U, a, sb, wg, lane, rdv, zlim, voff_e, voff_o, rows_valid);
    }
}

// SRCH: 0, or 1 + the kind of the decoder's one preamble: the wave searches its tile before it stores it (k1_search.h)
template <int CL, bool TAIL, class C, int SRCH>
__device__ __forceinline__ void k1t_body(const K1Args &a)
{
    using G = K1TGeom<CL>;
    extern __shared__ __attribute__((aligned(16))) uint8_t k1t_lds[];   // [0,16 KiB) two tile buffers, then the LUT
    // the DMA's M0 values and the LDS reads assume the dynamic segment starts at LDS address 0 (no static LDS here)
    if ((uint32_t)(uintptr_t)(lds_ptr_t)k1t_lds != 0) __builtin_trap();

    const uint32_t lane = threadIdx.x;
    k1_announce(a, lane);
#if AMR_K1T_CLK
    const uint64_t clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    uint32_t wgi = blockIdx.x;
    if (C::XCD && !TAIL && (gridDim.x & 7) == 0) wgi = (wgi & 7) * (gridDim.x >> 3) + (wgi >> 3);
    const uint32_t wg = a.wg_first + wgi;
    const uint32_t bs2 = a.block_size * 2;
    const uint32_t wpb = a.block_size >> 5;
    const uint32_t b = wg * kRows + lane;
    const uint32_t rows_valid = TAIL ? (a.n_blocks - wg * kRows) : kRows;

#if !AMR_K1T_LUT_DMA
    {
        float *lut = reinterpret_cast<float *>(k1t_lds + C::kLut);
#pragma unroll
        for (int i = 0; i < 4; ++i) lut[lane + 64 * i] = a.lut[lane + 64 * i];
    }
#endif

    // loader role: lane (rl, c') of piece q fetches row 8q+rl, 16-byte column c'^((row>>1)&7)
    const uint32_t rl = lane >> 3;
    const uint32_t colx = (lane & 7) ^ (rl >> 1);
    const uint32_t voff_e = rl * bs2 + colx * 16;
    const uint32_t voff_o = rl * bs2 + (colx ^ 4) * 16;
    // consumer role: row `lane`, swizzled slot
    const uint32_t rdv = lane * kTileBytes | ((lane >> 1) & 7) * 16;
    const bool fresh = a.zero_halo && wg == 0;
    const uint32_t zlim = (fresh && b == 0) ? G::WARM : 0;    // only stream block 0 of a fresh Decoder has zero history
    uint32_t *qrow = a.qt + (size_t)((C::DIAG == 7 ? (wg & 127) : wg) + 1) * kRows * wpb;

    // per piece q: start of the (aligned-halo + block) stream of row 8q of this wave-tile
    const uint8_t *sb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) sb[q] = k1_tile_base<G::HBA>(a, wg, bs2) - G::HBA + (int64_t)q * 8 * bs2;

    K1TLane<CL, C> L;
#pragma unroll
    for (int r = 0; r < G::RING; ++r) L.hc[r] = 0.0f;
#pragma unroll
    for (int r = 0; r < G::RING - C::HDL; ++r) L.hd[r] = 0.0f;
    if constexpr (C::HDL > 0) {                       // the ring starts from zeros, its LDS part too
        typedef float v4f __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4f *lds_v4f;
#pragma unroll
        for (int j = 0; j < C::HDL / 4; ++j) *(lds_v4f)(uintptr_t)(C::kHd + j * 1024 + lane * 16) = v4f{0.f, 0.f, 0.f, 0.f};
    }
    L.acc = 0; L.prev = 0; L.xs = 0;

    K1TUni U;
    U.ntiles = (G::HBA / 2 + a.block_size) / 64;
    U.t = 0; U.wi = 0; U.nch = 0; U.wdone = 0; U.st = 0;
    U.hold_last = SRCH > 0 ? 1u : 0u;
    U.odd = C::PRIO ? (__builtin_amdgcn_s_getreg(4 | (3 << 11)) & 1u) : 0u;   // HW_REG_HW_ID, wave_id: slot of this wave in its SIMD
    if (C::PRIO && U.odd) __builtin_amdgcn_s_setprio(1);

    // prologue: tiles 0 and 1 go out, tile 0 is drained, tile 2 follows it into buffer 0
    k1_prefetch<CL, TAIL>(a, 0, wg, 0, 0, lane, voff_e, voff_o, rows_valid);
#if AMR_K1T_LUT_DMA
    // the LUT (NewMagLUT, 1 KiB) follows tile 0 as ONE LDS-DMA instruction, 16 bytes per lane, instead of a global load +
    // ds_write pair in front of the first tile: one memory latency at the start of a wave instead of two.  Measured
    // neutral (harness A/B in one call, chip 8 and 72: inside the run-to-run spread -- with 8 waves per CU out of step
    // after the first round, another wave covers the start of this one); kept because it is less code.  No "nt": every
    // wave of the launch reads the same KiB.
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(lane * 16u), "s"(a.lut), "s"((uint32_t)C::kLut) : "memory");
#endif
    if (C::DEPTH == 2) {
        k1_prefetch<CL, TAIL>(a, 0, wg, 1, kTileBuf, lane, voff_e, voff_o, rows_valid);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    k1t_drain<CL, C>(L, rdv, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (C::HS != 0 && !TAIL) {
        static_assert(C::HS == 0 || (G::NPT == 1 && C::SCHED == 0 && C::DEPTH == 1), "halo shift: one halo tile, one tile in flight");
#pragma unroll
        for (int k = 0; k < 32; ++k) L.keep[k] = L.tl[k];
    }
    if ((uint32_t)C::DEPTH < U.ntiles) k1_prefetch<CL, TAIL>(a, 0, wg, C::DEPTH, 0, lane, voff_e, voff_o, rows_valid);
    U.par = C::kFlip;
    if (C::SCHED == 1 && G::SKIP < 4) k1t_gather<CL, C>(L, 0, 4, 0);

    k1t_warm<CL, TAIL, C, 0>(L, U, a, sb, wg, lane, rdv, zlim, voff_e, voff_o, rows_valid);
    L.prev = L.acc;

    // steady state: super-bodies of TPS tiles; vt_e / vt_o = lane offsets of the tile the next boundary fetches (t + 1 + depth)
    uint32_t vt_e = voff_e + (G::NPT + 1 + C::DEPTH) * kTileBytes, vt_o = voff_o + (G::NPT + 1 + C::DEPTH) * kTileBytes;
    while (!k1t_super<CL, TAIL, C, 0>(L, U, a, sb, wg, lane, rdv, voff_e, voff_o, vt_e, vt_o, rows_valid, qrow)) {}
    if (C::DIAG == 3) U.nch = 0;
    if constexpr (SRCH > 0) {
        // the lane's row, whole and still on the chip: rows of 16 words (chip 8) are four chunks parked in LDS; rows of 64 words
        // (chip 32, 40) are exactly one burst -- 11 chunks parked, 4 in the register vector, the last in the staging registers --
        // which the last tile boundary held back (K1TUni::hold_last)
        constexpr int WPB = CL == 8 ? 16 : 64, CPR = WPB / 4;
        static_assert(!TAIL && (CPR <= C::NLC || CPR == C::CAP), "in-wave search: the whole row must be buffered when the block ends");
        if (a.block_size == 32u * WPB && U.nch == (uint32_t)CPR) {                // (wave-uniform; anything else: the launcher's mistake)
            typedef const __attribute__((address_space(3))) k2w_v4u *lds_v4;
            K2WRing<CPR> R;
#pragma unroll
            for (int c = 0; c < CPR; ++c) {
                if (c < C::NLC) R.c[c] = *(lds_v4)(uintptr_t)(C::kPark + c * 1024 + lane * 16);
                else if (c < C::CAP - 1) R.c[c] = k2w_v4u{L.ow[(c - C::NLC) * 4], L.ow[(c - C::NLC) * 4 + 1], L.ow[(c - C::NLC) * 4 + 2], L.ow[(c - C::NLC) * 4 + 3]};
                else R.c[c] = k2w_v4u{L.st4.x, L.st4.y, L.st4.z, L.st4.w};
            }
            // scratch: the tile buffer at LDS offset 0, which no DMA targets any more
            k1s_search_tile<2 * CL, SRCH - 1, WPB>(a.srch, wg + 1, lane, R, reinterpret_cast<uint32_t *>(k1t_lds), WPB == 16 ? 9u : 11u);
        }
    }
    if (U.nch) k1t_flush<CL, C>(L, U, qrow);
    if (C::DIAG == 3 || C::DIAG == 2) qrow[lane * 4] = L.xs ^ L.acc;
    if (a.done_flags) {
        // early search: the rows of this wave-tile are in memory once every store of the wave has been acknowledged (they are
        // write-through, sc1) -- then, and only then, the flag (sc1 as well; the searching wave polls it with sc1 loads)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            uint32_t *p = a.done_flags + wg;
            const uint32_t v = a.done_value;
            asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
        }
        if (a.carry_out && (wg + 1) * kRows == a.n_blocks && lane < (uint32_t)G::HBA / 16) {   // the batch's last wave-tile
            const uint4 *src = reinterpret_cast<const uint4 *>(a.iq + (size_t)a.n_blocks * bs2 - G::HBA);   // (no head rows in this mode)
            reinterpret_cast<uint4 *>(a.carry_out)[lane] = src[lane];
        }
    }
#if AMR_K1T_CLK
    if (blockIdx.x == 0 && lane == 0) {
        const uint64_t c = __builtin_readcyclecounter() - clk0, r = __builtin_amdgcn_s_memrealtime() - rt0;
        a.qt[0] = (uint32_t)c; a.qt[1] = (uint32_t)(c >> 32); a.qt[2] = (uint32_t)r; a.qt[3] = (uint32_t)(r >> 32);
    }
    if (lane == 0 && blockIdx.x < (uint32_t)kK1TWgs) {
        uint32_t xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned long long *t = k1t_timeline[a.tl_seq % kK1TLaunches][blockIdx.x];
        t[0] = rt0; t[1] = __builtin_amdgcn_s_memrealtime(); t[2] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

template <int CL, bool TAIL, class C>
__global__ __launch_bounds__(64, 2) void k1t_demod(const K1Args a)
{
    k1t_body<CL, TAIL, C, 0>(a);
}

// ... with the in-wave search for the decoder's one preamble (kind KIND), rows of 16 words
template <int CL, class C, int KIND>
__global__ __launch_bounds__(64, 2) void k1t_demod_srch(const K1Args a)
{
    k1t_body<CL, false, C, KIND + 1>(a);
}

}  // namespace amr
