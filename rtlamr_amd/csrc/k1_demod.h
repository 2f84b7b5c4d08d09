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
// Output layout ("tiled"): word w of block b is stored at
//     qt[(b/64 + 1) * 64*WPB + w*64 + b%64],   WPB = BlockSize/32,
// i.e. a wave's 64 lanes store 256 contiguous bytes per 32 samples.  Tile 0 of
// qt holds the previous batch's last rows (history for the preamble search).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Developer diagnostics (never defined in the product build): 1 = no HBM traffic after the first tile
// (times the LDS/VALU side alone), 2 = HBM->LDS staging + row reads only (times the memory side alone),
// 3 = no output stores.
#ifndef AMR_K1_DIAG
#define AMR_K1_DIAG 0
#endif

namespace amr {

constexpr int kRows = 64;                       // block-rows per wave, one per lane
constexpr int kTileBytes = 128;                 // IQ bytes per row per staging tile (one cache line)
constexpr int kTileBuf = kRows * kTileBytes;    // 8 KiB per buffer

struct K1Args {
    const uint8_t *iq;     // batch block 0, byte 0 (device)
    const uint8_t *carry;  // the HBA stream bytes that precede iq (device, always valid memory)
    const float *lut;      // NewMagLUT, 256 floats (device)
    uint32_t *qt;          // tiled bitstream; tile 0 = history tile
    uint32_t n_blocks;     // blocks in the batch
    uint32_t block_size;   // BlockSize in samples (power of two >= 512)
    uint32_t wg_first;     // first wave-tile of this launch
    uint32_t zero_halo;    // 1: magnitudes before batch block 0 are 0.0 (fresh Decoder, decode.go:144)
};

template <int CL>
struct K1Geom {
    static constexpr int SL = 2 * CL;
    static constexpr int HB = 4 * CL;                 // halo bytes a block needs before its first sample
    static constexpr int HBA = (HB + 127) & ~127;     // rounded up to whole cache lines
    static constexpr int SKIP = (HBA - HB) / 2;       // leading samples forced to magnitude 0
    static constexpr int WARM = SKIP + SL;            // steps before the first output bit
    static constexpr int RING = CL + 8;               // csum history ring (registers), one unrolled body = RING samples
    static constexpr int GPB = RING / 8;              // 8-sample groups per unrolled body
    static constexpr int NPB = (WARM + RING - 1) / RING;  // bodies that need the zero-magnitude predicate
    static constexpr int NPT = HBA / kTileBytes;      // staging tiles that lie in the halo
};

// cache policy of the IQ stream: "nt" (aux bit 1).  Every byte is read exactly once, so keeping it out of the
// L2 / Infinity Cache replacement queues is worth 6.1 -> 7.0 TB/s on the bare staging loop (tools/dma_bench2.hip).
constexpr int kAuxNT = 2;
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

template <int CL>
struct K1Lane {
    // Rings of RING = CL+8 registers, indexed statically (the sample loop is unrolled RING times): step t
    // writes slot t%RING and reads slot (t+8)%RING = the value of step t-CL.  Because a slot is dead for 8
    // steps before it is overwritten, every value is produced straight into its final register and the
    // loop back-edge needs no register rotation (a CL-deep ring costs two v_mov per sample).
    float hc[K1Geom<CL>::RING];  // hc[t%RING] = c[t], the running sum after sample t (decode.go:234)
    float hd[K1Geom<CL>::RING];  // hd[t%RING] = c[t] - c[t-CL]
    uint4 row;     // the 8 IQ samples of the group about to be consumed
    uint32_t acc;  // sign bits of f, newest in bit 0 (inverted decisions)
    uint32_t prev; // acc at the previous 32-sample boundary
};

struct K1Uni {       // wave-uniform state (SGPRs)
    uint32_t G;      // 8-sample groups consumed so far
    int32_t og;      // output groups produced so far (negative during warm-up)
    uint32_t ngroups;// total groups per lane
    uint32_t ntiles; // total staging tiles per lane
};

template <int CL, bool TAIL>
__device__ __forceinline__ void k1_prefetch(const K1Args &a, uint8_t *smem, uint32_t wg, uint32_t t, uint32_t buf_off,
                                            uint32_t lane, uint32_t voff_e, uint32_t voff_o, uint32_t rows_valid)
{
    using G = K1Geom<CL>;
    const uint32_t bs2 = a.block_size * 2;
    // uniform: first row of this wave-tile, shifted to staging tile t of the (aligned-halo + block) stream
    const uint8_t *sb = a.iq + (int64_t)wg * kRows * bs2 - G::HBA + (int64_t)t * kTileBytes;
    const uint32_t rl = lane >> 3;
    const bool carry_tile = (wg == 0) && (t < (uint32_t)G::NPT);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint8_t *g;
        if (TAIL) {
            uint32_t row = q * 8 + rl;
            uint32_t colb = ((lane & 7) ^ ((row >> 1) & 7)) * 16;
            uint32_t rc = row < rows_valid ? row : rows_valid - 1;
            g = sb + (size_t)rc * bs2 + colb;
            if (carry_tile && rc == 0) g = a.carry + t * kTileBytes + colb;
        } else {
            uint32_t voff = (q & 1) ? voff_o : voff_e;
            g = sb + (size_t)(q * 8) * bs2 + voff;
            if (carry_tile && q == 0) {
                uint32_t colb = ((lane & 7) ^ ((rl >> 1) & 7)) * 16;
                g = (rl == 0) ? a.carry + t * kTileBytes + colb : g;
            }
        }
        __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(smem + buf_off + q * 1024), 16, 0, kAuxNT);
    }
}

// Read the 16 bytes (8 IQ samples) of one group of this lane's row.  Inline asm on purpose: hipcc
// orders every compiler-visible LDS load behind ALL outstanding LDS-DMA (s_waitcnt vmcnt(0)), which
// would serialise the prefetch of tile t+1 with the consumption of tile t.  The asm load is
// invisible to that logic; correctness comes from the explicit s_waitcnt vmcnt(0) issued when a
// buffer is first read.  The lgkmcnt wait sits in the SAME asm statement: an asynchronous asm load
// whose wait comes later is unsafe, because the compiler may copy the destination registers (phi
// moves, live-range splits) while the load is still in flight.
__device__ __forceinline__ uint4 k1_row_read(uint32_t lds_addr)
{
    uint4 dst;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(dst) : "v"(lds_addr) : "memory");
    return dst;
}

// Make group `grp` readable and read it: on a tile boundary wait for the tile's DMA (issued one
// tile-time earlier), then refill the buffer that was just drained (all its reads have returned).
template <int CL, bool TAIL>
__device__ __forceinline__ uint4 k1_fetch_group(uint32_t grp, const K1Uni &U, const K1Args &a, uint32_t tiles_lds,
                                                uint8_t *tiles, uint32_t wg, uint32_t lane, uint32_t rd_base,
                                                uint32_t rd_xor, uint32_t voff_e, uint32_t voff_o, uint32_t rows_valid)
{
    const uint32_t gt = grp & 7;
    const uint32_t t = grp >> 3;
    if (gt == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t + 1 < U.ntiles && AMR_K1_DIAG != 1)
            k1_prefetch<CL, TAIL>(a, tiles, wg, t + 1, ((t + 1) & 1) * kTileBuf, lane, voff_e, voff_o, rows_valid);
    }
    return k1_row_read(tiles_lds + (t & 1) * kTileBuf + rd_base + ((gt * 16) ^ rd_xor));
}

// One unrolled "body" = CL samples = CL/8 groups; the csum history registers are indexed statically.
// Per group the instruction stream is organised by hand (sched_barrier keeps hipcc from re-interleaving
// it into a load-wait-use chain per sample, which left the wave waiting on ~9 LDS round trips per group):
//   1. 16 LUT gathers of the group whose IQ bytes are already in registers (L.row),
//   2. the row read of the NEXT group (asm, with its lgkmcnt(0)): one wait covers 1. and 2.,
//   3. 8 x (magnitude add, running sum, two differences, sign bit), no memory access.
template <int CL, bool PRO, bool TAIL>
__device__ __forceinline__ void k1_body(K1Lane<CL> &L, K1Uni &U, const K1Args &a, uint32_t tiles_lds,
                                        uint8_t *tiles, const float *lut, uint32_t wg, uint32_t lane, uint32_t rd_base,
                                        uint32_t rd_xor, uint32_t zlim, uint32_t voff_e, uint32_t voff_o,
                                        uint32_t rows_valid, uint32_t *qrow)
{
    using G = K1Geom<CL>;
#pragma unroll
    for (int g = 0; g < G::GPB; ++g) {
        if (U.G >= U.ngroups) return;
        const uint32_t dw[4] = {L.row.x, L.row.y, L.row.z, L.row.w};
#if AMR_K1_DIAG == 2
        L.acc ^= dw[0] ^ dw[1] ^ dw[2] ^ dw[3];
        if (U.G + 1 < U.ngroups)
            L.row = k1_fetch_group<CL, TAIL>(U.G + 1, U, a, tiles_lds, tiles, wg, lane, rd_base, rd_xor, voff_e, voff_o,
                                             rows_valid);
#else
        float li[8], lq[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t v = dw[k >> 1] >> ((k & 1) * 16);
            li[k] = lut[v & 0xff];                             // decode.go:222
            lq[k] = lut[(v >> 8) & 0xff];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (U.G + 1 < U.ngroups)
            L.row = k1_fetch_group<CL, TAIL>(U.G + 1, U, a, tiles_lds, tiles, wg, lane, rd_base, rd_xor, voff_e, voff_o,
                                             rows_valid);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            constexpr int R = G::RING;
            const int r = g * 8 + k, rp = (r + R - 1) % R, ro = (r + 8) % R;
            float m = li[k] + lq[k];                           // decode.go:222
            if (PRO) m = (U.G * 8 + k < zlim) ? 0.0f : m;      // zero history, decode.go:144
            const float c = L.hc[rp] + m;                      // decode.go:234
            const float d = c - L.hc[ro];                      // csum[i+SL]-csum[i+CL]   (decode.go:242)
            const float f = L.hd[ro] - d;                      // (csum[i+CL]-csum[i]) - d (decode.go:242)
            L.acc = __builtin_amdgcn_alignbit(L.acc, __float_as_uint(f), 31);  // decode.go:243, inverted
            L.hc[r] = c;
            L.hd[r] = d;
        }
#endif
        U.G += 1;
        U.og += 1;
        // Output i leaves the filter at step WARM-1+i, i.e. one step before a group boundary: at a
        // boundary acc holds outputs [32m+1 .. 32m+32]; output 32m is bit 0 of acc at the previous boundary.
        if (U.og >= 0 && (U.og & 3) == 0) {
            if (U.og > 0 && AMR_K1_DIAG != 3)   // 64 lanes -> 256 contiguous bytes
                qrow[((U.og >> 2) - 1) * kRows] = ~__builtin_amdgcn_alignbit(L.prev, L.acc, 1);
            L.prev = L.acc;
        }
    }
}

template <int CL, bool TAIL>
__global__ __launch_bounds__(64, 2) void k1_demod(const K1Args a)
{
    using G = K1Geom<CL>;
    // Two LDS objects on purpose: with distinct objects hipcc can prove that the LUT loads do not
    // alias the LDS-DMA destination and does not put s_waitcnt vmcnt(0) in front of them.
    __shared__ __attribute__((aligned(16))) uint8_t tiles[2 * kTileBuf];
    __shared__ __attribute__((aligned(16))) float lut[256];
    const uint32_t tiles_lds = (uint32_t)(uintptr_t)(lds_ptr_t)tiles;

    const uint32_t lane = threadIdx.x;
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
    const uint32_t rd_base = lane * kTileBytes;
    const uint32_t rd_xor = ((lane >> 1) & 7) * 16;
    const uint32_t zlim = (a.zero_halo && b == 0) ? G::WARM : G::SKIP;
    uint32_t *qrow = a.qt + (size_t)(wg + 1) * kRows * wpb + lane;

    K1Lane<CL> L;
#pragma unroll
    for (int r = 0; r < G::RING; ++r) { L.hc[r] = 0.0f; L.hd[r] = 0.0f; }
    L.acc = 0;
    L.prev = 0;

    K1Uni U;
    U.G = 0;
    U.og = -(G::WARM / 8);
    U.ngroups = (G::HBA / 2 + a.block_size) / 8;
    U.ntiles = U.ngroups / 8;

    k1_prefetch<CL, TAIL>(a, tiles, wg, 0, 0, lane, voff_e, voff_o, rows_valid);
    L.row = k1_fetch_group<CL, TAIL>(0, U, a, tiles_lds, tiles, wg, lane, rd_base, rd_xor, voff_e, voff_o, rows_valid);

    const uint32_t nbodies = (U.ngroups + G::GPB - 1) / G::GPB;
    uint32_t body = 0;
    for (; body < (uint32_t)G::NPB && body < nbodies; ++body)
        k1_body<CL, true, TAIL>(L, U, a, tiles_lds, tiles, lut, wg, lane, rd_base, rd_xor, zlim, voff_e, voff_o,
                                rows_valid, qrow);
    for (; body < nbodies; ++body)
        k1_body<CL, false, TAIL>(L, U, a, tiles_lds, tiles, lut, wg, lane, rd_base, rd_xor, zlim, voff_e, voff_o,
                                 rows_valid, qrow);
}

}  // namespace amr
