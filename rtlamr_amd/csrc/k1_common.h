// What the K1 kernels share -- k1_tile.h (whole wave-tiles), k1_coop.h (one wave per block) --: the launch arguments, the
// "tiled4" output layout, where a wave-tile's rows start, the announcement to the gate, and the LDS-DMA of one staging
// tile.  (K1Geom's RING / NW and AMR_K1_PIPE are left from the first-generation kernel of rounds 1-2, which is in the
// history: git log -- tools/k1_demod_gen1.h; the product uses K1Geom's halo arithmetic only.)
//
// Output layout ("tiled4"): word w of block b is stored at
//     qt[(b/64 + 1) * 64*WPB + (w/4)*256 + (b%64)*4 + w%4],   WPB = BlockSize/32,
// i.e. a lane stores 16 bytes and a wave 1 KiB per store instruction.  Tile 0 of
// qt holds the previous batch's last rows (history for the preamble search).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// cache-policy bits of the LDS-DMA loads of the IQ stream.  A/B on one MI355X (K1 ms, 1 GiB): "nt" 0.2255 / 0.2261,
// "sc1 nt" 0.2261 / 0.2258, "sc0 sc1 nt" 0.2271 / 0.2257, "sc0 nt" 0.2256 / 0.2246, "sc1" (temporal) 0.2430 / 0.2447.
#ifndef AMR_K1_LDFLAGS
#define AMR_K1_LDFLAGS "nt"
#endif

// ... and of the halo tiles alone (the first NPT tiles of a row = the last NPT tiles of the row before it, which that row's
// lane reads again at the end of the launch).  Experiment of round 5: left cacheable ("") they could come from the Infinity Cache
// the second time: +1 % in the harness, -1..2 % in the pipelined bench; the default is the stream's policy.
#ifndef AMR_K1_HALO_LDFLAGS
#define AMR_K1_HALO_LDFLAGS AMR_K1_LDFLAGS
#endif

#ifndef AMR_K1_PIPE
#define AMR_K1_PIPE 0   // first-generation kernel only (in the history); K1Geom::NW below depends on it
#endif

// diagnostic builds (make EXTRA=-DAMR_K1T_CLK=1, tools/build_variant.sh): per-workgroup clock stamps of k1t_demod
#ifndef AMR_K1T_CLK
#define AMR_K1T_CLK 0
#endif

namespace amr {

constexpr int kRows = 64;                       // block-rows per wave, one per lane
constexpr int kTileBytes = 128;                 // IQ bytes per row per staging tile (one cache line)
constexpr int kTileBuf = kRows * kTileBytes;    // 8 KiB per buffer

// In-wave search (round 6, k1_search.h): where a lane's whole row of decisions is still on the chip when its block ends (rows of
// 16 words: BlockSize 512), the K1 wave searches its own tile -- lane = row, the look-ahead is the neighbour lane, exactly the
// geometry of k2_search_row -- and writes what that kernel writes: the tile's list in the staging slot, its count, the group
// sum.  Row 63's last words (their windows reach into the next tile, which another wave is still writing) and the history
// tile are left to a clean-up launch (k2_row_cleanup, amr_pipeline.hip).  kind_p1 = 0: off.
struct K1Search {
    uint32_t *counts, *gcnt, *staging, *overflow;   // K2Args of the batch (one preamble)
    int64_t n_lo, n_hi;                             // valid positions, as K2Args
    uint32_t cap, n_tiles;
    uint32_t kind_p1;                               // 1 + the preamble's kind (k2_walk_kind_of)
    uint32_t pad_;
};

struct K1Args {
    const uint8_t *iq;     // row 0 of the launch, byte 0 (device); with head_rows the rows below 64 are never read from here
    const uint8_t *carry;  // the "head" buffer: the HBA stream bytes that precede row 0 of the launch, and behind them
                           // (head_rows) the 64 rows of wave-tile 0, contiguous like a caller's batch
    const float *lut;      // NewMagLUT, 256 floats (device)
    uint32_t *qt;          // tiled bitstream; tile 0 = history tile
    uint32_t n_blocks;     // blocks in the batch
    uint32_t block_size;   // BlockSize in samples (power of two >= 512)
    uint32_t wg_first;     // first wave-tile of this launch
    uint32_t zero_halo;    // 1: magnitudes before batch block 0 are 0.0 (fresh Decoder, decode.go:144)
    // 1: wave-tile 0 of the launch lies in the head buffer (blocks deferred from the previous batch in front, the first
    // blocks of this batch copied in behind them, see submit() in amr_pipeline.hip); wave-tiles >= 1 are at iq + row * bs2
    uint32_t head_rows;
    // Pipelined callers (amr_pipeline.hip, submit): the LAST workgroup of the batch's last K1 launch stores started_value here
    // when it starts -- by then every wave of the launch has its slot.  A gate kernel on the second stream waits for it and
    // lets the previous batch's K3 in: its workgroups then find room only where K1 waves retire, i.e. they fill the ragged
    // end of this launch instead of standing in front of it.  null: no announcement.
    uint64_t *started;
    uint64_t started_value;
    // (round 6) ... when EVERY XCD has placed its last workgroup of the launch: the last min(8, grid) workgroups (one per XCD:
    // workgroups go round the XCDs) count themselves in `started_ctr`, a device word that only ever grows; the one that
    // brings it to `started_target` (the host adds up what each announcing launch contributes) stores the ticket.  The XCDs'
    // dispatchers do not run in step: the last workgroup of the grid alone said nothing about the other seven.
    uint32_t *started_ctr;
    uint32_t started_target;
    // 1: the FIRST min(8, grid) workgroups announce instead (a launch of many rounds that has no lock-step to protect, BlockSize
    // 512: what the gate then says is "everything in front of this launch on its stream has finished", and the tail it lets in
    // takes its slots round by round as K1 waves retire)
    uint32_t started_first;
    // Early search (amr_pipeline.hip): the search of this batch runs off the compute stream, NEXT to this launch,
    // and takes a tile as soon as the waves that wrote it are done.  Every wave, at its end, waits for its stores (sc1:
    // written through, nothing stays dirty in the XCD's L2) and then stores done_value into done_flags[wave-tile]; the wave of
    // the batch's last wave-tile also leaves the IQ halo of the next batch's block 0 in carry_out (decode.go:165), which the
    // state update inside the search can no longer do in front of the next K1 launch.  null: none of this (k1t_demod only).
    uint32_t *done_flags;
    uint8_t *carry_out;
    uint32_t done_value;
    K1Search srch;
#if AMR_K1T_CLK
    uint32_t tl_seq;       // diagnostic builds: slot of this launch in k1t_timeline (k1_launch.inc counts)
#endif
};

__device__ __forceinline__ void k1_announce(const K1Args &a, uint32_t lane)
{
    // A store the optimiser cannot see (no "memory" clobber: it touches nothing this kernel reads), system scope (sc0 sc1):
    // the gate kernel reads the word with agent-scope loads on another XCD.  Written as __hip_atomic_store it makes hipcc give up
    // the scalar loads of the kernel arguments behind it: the DMA's base pointers then arrive in VGPR pairs, which the
    // "s" operands of the inline asm cannot take.
    if (a.started && (a.started_first ? blockIdx.x < 8 : blockIdx.x + 8 >= gridDim.x) && lane == 0) {
        uint32_t *c = a.started_ctr;
        uint32_t old;
        // (device-scope atomic with return, issued before any of the wave's DMA: the counted waits of the tile loop never see it)
        asm volatile("global_atomic_add %0, %1, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(c), "v"(1u));
        if (old + 1u == a.started_target) {
            uint64_t *p = a.started;
            const uint64_t v = a.started_value;
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v));
        }
    }
}

// start of the row stream (block 0 of the wave-tile, byte 0) a wave-tile reads
template <int HBA>
__device__ __forceinline__ const uint8_t *k1_tile_base(const K1Args &a, uint32_t wg, uint32_t bs2)
{
    return (wg == 0 && a.head_rows) ? a.carry + HBA : a.iq + (int64_t)wg * kRows * bs2;
}

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
    // Output words (32 decisions each) a lane keeps in registers between flushes.  Mixing a trickle of writes
    // into the saturated read stream costs far more than the bytes (tools/sst_bench.hip: 64 MiB written per
    // tile step +35 % kernel time, the same bytes in a few chip-wide bursts +3 %), so the bits are held in
    // whatever VGPRs the csum rings leave free (256 per lane at 2 waves per SIMD) and written in bursts.
#ifdef AMR_K1_NW
    static constexpr int NW = AMR_K1_NW;
#else
    static constexpr int NW = AMR_K1_PIPE ? (RING <= 80 ? 8 : 4) : (RING <= 80 ? 32 : RING <= 88 ? 16 : 4);
#endif
    static constexpr int NW0 = NW > 32 ? 32 : NW;      // words in the first register vector
    static constexpr int NW1 = NW > 32 ? NW - 32 : 4;  // words in the second (a 4-word dummy when unused)
};

// cache policy of the IQ stream: "nt" (aux bit 1).  Every byte is read exactly once, so keeping it out of the
// L2 / Infinity Cache replacement queues is worth 6.1 -> 7.0 TB/s on the bare staging loop (tools/dma_bench2.hip).
constexpr int kAuxNT = 2;
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

template <int CL, bool TAIL>
__device__ __forceinline__ void k1_prefetch(const K1Args &a, uint32_t lds_base, uint32_t wg, uint32_t t, uint32_t buf_off,
                                            uint32_t lane, uint32_t voff_e, uint32_t voff_o, uint32_t rows_valid)
{
    using G = K1Geom<CL>;
    const uint32_t bs2 = a.block_size * 2;
    // uniform: first row of this wave-tile, shifted to staging tile t of the (aligned-halo + block) stream
    const uint8_t *sb = k1_tile_base<G::HBA>(a, wg, bs2) - G::HBA + (int64_t)t * kTileBytes;
    const uint32_t rl = lane >> 3;
    // row 0 of the launch takes its halo from the head buffer; with head_rows the whole wave-tile sits behind it anyway
    const bool carry_tile = (wg == 0) && (t < (uint32_t)G::NPT) && !a.head_rows;
    // LDS-DMA: 64 lanes x 16 bytes -> 1 KiB of LDS at M0.  Issued from inline asm on purpose: when hipcc sees an
    // LDS-DMA in flight it guards EVERY later LDS load that may alias its target with s_waitcnt vmcnt(0), which
    // would serialise the prefetch of tile t+1 with the consumption of tile t.  Hidden in asm, the DMA is
    // ordered by the explicit s_waitcnt vmcnt(0) in k1_fetch_next alone, and the row reads stay ordinary loads
    // that the compiler keeps in flight across groups.  "nt": every byte is read once.  SGPR base + 32-bit lane
    // offset addressing keeps the per-lane state at two VGPRs.  The asm overwrites M0 (hipcc does not accept it as a
    // clobber): nothing else here keeps a value in M0 across statements -- gfx9 DS instructions do not read it and
    // hipcc's own s_set_gpr_idx_on/off pairs are self-contained.
    if (!TAIL && !carry_tile) {
        const uint8_t *base = sb;                                   // uniform
        uint32_t m0v = lds_base + buf_off;
        // (Tried: when the first half of tile 0's line holds none of the 4 * CL halo bytes -- chip 8: 32 needed of 128 --
        // let the lanes of that half sit the DMA out.  Bit-exact, and not a microsecond faster at any chip length: a miss
        // brings the whole 128-byte line whatever part of it is asked for.)
        if (t < (uint32_t)G::NPT) {                                 // halo tiles: their own cache policy (see AMR_K1_HALO_LDFLAGS)
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 " AMR_K1_HALO_LDFLAGS
                             :: "v"(voff_e), "s"(base), "s"(m0v) : "memory");
                base += (size_t)8 * bs2;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 " AMR_K1_HALO_LDFLAGS
                             :: "v"(voff_o), "s"(base), "s"(m0v + 1024) : "memory");
                base += (size_t)8 * bs2;
                m0v += 2048;
            }
            return;
        }
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {                               // a rolled loop: this code sits in every group
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 " AMR_K1_LDFLAGS
                         :: "v"(voff_e), "s"(base), "s"(m0v) : "memory");
            base += (size_t)8 * bs2;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 " AMR_K1_LDFLAGS
                         :: "v"(voff_o), "s"(base), "s"(m0v + 1024) : "memory");
            base += (size_t)8 * bs2;
            m0v += 2048;
        }
        return;
    }
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
            if (q == 0) {
                uint32_t colb = ((lane & 7) ^ ((rl >> 1) & 7)) * 16;
                g = (rl == 0) ? a.carry + t * kTileBytes + colb : g;
            }
        }
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off " AMR_K1_LDFLAGS
                     :: "v"(g), "s"(lds_base + buf_off + q * 1024) : "memory");
    }
}

}  // namespace amr
