// K3 (hit slots + Decoder.Slice, protocol/decode.go:353-375) and the small kernels around a batch: the state update
// as a kernel of its own (callers that do not pipeline), the completion ticket, the test helper that untiles the bitstream.
#pragma once
#include "k2_common.h"
#include "k5_validate.h"

namespace amr {

__global__ __launch_bounds__(1024) void k_hist_update(const HistArgs a)
{
    extern __shared__ uint32_t hist_tmp[];  // hr*wpb words
    // one workgroup.  The copies of deferred blocks are NOT part of this kernel (submit runs them ahead of it): the ticket
    // below tells the host that the caller's buffer is free
    hist_body(a, hist_tmp, 1024);
    // every earlier kernel of the batch has completed (same stream); the host polls these words
    if (threadIdx.x == 0) hist_publish(a);
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
    // K5 (k5_validate.h) as the last stage of the workgroup: keep == nullptr = off
    uint8_t *keep;              // [out_cap] 1 = the hit passes its preamble's rule
    uint32_t *listcnt;          // [n_pre][n_tiles] survivors of every list
    uint64_t *listoff;          // [n_pre][n_tiles] slot of every non-empty list in the packed result
    uint32_t *vgcnt;            // [n_pre][n_groups] survivors summed over groups of 64 tiles (atomicAdd; zero before K3 runs)
    uint32_t lds_bytes;         // dynamic LDS of the launch (k3_lds_bytes)
    uint32_t fold;              // 1: grid (n_tiles - 1, n_pre), workgroup 0 also takes the history tile (k3_fold)
    uint32_t prio;              // wave priority (s_setprio) of K3's waves: they share SIMDs with the next batch's search
    ValRule rule[AMR_MAX_PREAMBLES];
};

// K3 slices by bitstream word, not by hit, out of LDS copies of the few rows a run of hits needs.
//
// By word: the hits of a real packet (and most noise hits' neighbours) come in runs of adjacent positions, so slicing hit
// by hit (round 1: one lane = 32 symbols of one hit) read every bitstream word ~20 times and spent ~13 VALU operations
// per (hit, symbol).  The unit of work is a bitstream WORD that holds hits: for symbol p the 32 positions of the word need
// the 32 stream bits starting at word*32 + p*SL -- one window, one or two words (SL is a multiple of 16) -- and the
// packets of all 32 positions are the columns of the bit matrix [symbol][position].  A wave takes 64 symbols at a time,
// lane = symbol (two 32 x 32 blocks), transposes the blocks in five exchange steps (ds_swizzle, no LDS memory), after
// which lane c of a block holds 32 consecutive packet bits of position 31-c: one dword of that packet, already in the
// byte order of Decoder.Slice (decode.go:363-366) because the symbols were dealt to the lanes bit-reversed inside
// every byte.  Positions that are hits store their dword, the others are dropped.
//
// Out of LDS: a packet is PacketSymbols bits at a stride of SymbolLength; IDM-length packets reach 13 rows of 8192 bits
// past the hit.  Straight from the tiled bitstream every symbol's window is another 64-byte line (a row's words sit in
// 16-byte pieces a KiB apart): 70 KB of line fetches per packet run for the 13 KB of stream they lie in -- round 2's K3
// moved more bytes than the search itself (cfg3: 290 MB against 256 MiB).  So the rows [l, l + n_rows) are copied into
// LDS ONCE, in stream order (16-byte pieces, neighbouring rows share their lines), and the windows of every hit that
// starts in row l are taken from there.
//
// One workgroup per (tile, preamble) list, grid (n_tiles, n_pre) (a workgroup that took all lists of its tile and staged
// a row once for all of them measured slower: 188 against 173 us per 4 GiB of the four-preamble decoder).  Its prologue:
// the slot of a list in the packed result = the hits of all lists before it (preamble-major), and the layout needs the
// grand total.  No scan kernel between K2 and K3 (a dispatch costs the stream ~5 us): K2 left sums over groups of 64 tiles; the workgroup scans them (32-bit DPP scan inside a wave -- a wave's 64
// group sums stay below 2^31 -- the few wave totals in 64 bits) and adds the <= 63 counts before it inside its group.
// Tile 0's workgroup publishes the per-preamble bases, the total and K2's overflow word.
// Input: positions in the staging slots, ascending; output: the packed result (K3Args).  Dynamic LDS: k3_lds_bytes().
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

// inclusive 32-bit prefix sum over the 64 lanes of a wave (DPP)
__device__ __forceinline__ uint32_t k3_wave_scan(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);    // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);    // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);    // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);    // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return x;
}

#ifndef AMR_K3_DBG
#define AMR_K3_DBG 0        // diagnostic builds: 100 MHz time stamps of every workgroup's phases, printed by amr_destroy
#endif
#if AMR_K3_DBG
__device__ unsigned long long k3_dbg[4096 * 8];
#define K3_STAMP(i) do { if (tid == 0 && blockIdx.x < 4096) k3_dbg[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define K3_STAMP(i) do { } while (0)
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global store of the wave to be
// acknowledged (vmcnt(0)): behind the slicing that is the drain of all packet stores of the chip's workgroups at once,
// 15 us that a workgroup which only goes on to read LDS does not owe.
__device__ __forceinline__ void k3_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int kK3Batch = 4;    // words (entries) a wave works on together
constexpr int kK3List = 512;   // positions of a list held in LDS at a time (long packets): 2 KiB, eight workgroups still fit a CU

// 64 hits of one list (src[i0 .. i0 + 64) below i_hi) by one wave: their (call, idx) records and the packets of every word
// that holds some of them.  LONG (long packets) and `staged`: the windows come from rows_lds (stream order, bit 31 of word
// 0 = tile-local bit base_bit), else from the tiled bitstream at tbase.  Symbols [p_first, PS) in steps of 128
// (p_step).  pk_lds (short packets of whole dwords): a second copy of the packets of list entries below pk_hits, entry i at pk_lds + i * PB,
// and of their positions in pos_lds.
template <bool LONG>
__device__ __forceinline__ void k3_chunk(const K3Args &a, const SearchGeom &g, uint32_t T, const uint32_t *src, uint32_t i0, uint32_t i_hi,
                                         uint32_t i_lo, uint32_t wv, uint64_t off, uint64_t total, uint32_t (&tab)[kK3Batch][32],
                                         const uint32_t *rows_lds, const uint32_t *__restrict__ tbase, uint32_t base_bit,
                                         uint32_t p_first, bool by_symbols, uint32_t i_out0, uint32_t tid, uint8_t *pk_lds = nullptr, uint32_t pk_hits = 0,
                                         uint32_t *pos_lds = nullptr, bool staged = false)
{
    const uint32_t lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const uint32_t lg_bs = g.lg_block_size, bs_mask = g.block_size - 1, lg_tw = 6 + g.lg_wpb;
    const uint32_t PS = g.packet_symbols, SL = g.symbol_length, PB = g.pkt_bytes;
    const bool dword_ok = (PB & 3) == 0 && (PS & 7) == 0;
    // symbol offset of this lane inside a 64-symbol step: the 32 lanes of a block take the symbols bit-reversed
    // within every byte, so that bit i of the transposed dword is the symbol Decoder.Slice puts into bit i
    const uint32_t sym_lane = half * 32 + ((l32 & ~7u) | (7u - (l32 & 7u)));
    const uint32_t bad = 64u << lg_bs;                 // defensive: never index the bitstream with a bad position
    uint64_t *hit_block = reinterpret_cast<uint64_t *>(a.out);
    uint32_t *hit_idx = reinterpret_cast<uint32_t *>(a.out + total * 8);
    uint8_t *pkt = a.out + total * 12;
    auto word_at = [&](uint32_t v) {                   // bitstream word holding bit v (counted from row 0 of tile T)
        const uint32_t row = v >> lg_bs, w = (v & bs_mask) >> 5;
        return tbase[((row >> 6) << lg_tw) + ((w >> 2) << 8) + ((row & 63) << 2) + (w & 3)];
    };
    const uint32_t i = i0 + lane;
    const bool have = i < i_hi;
    const uint32_t local = have ? src[i] : 0xffffffffu;
    const bool ok = have && local < bad;
    if (!LONG && have && i < pk_hits) pos_lds[i] = local;           // K5's test wants the positions again
    if (ok && (!by_symbols || wv == (((i0 - i_lo) >> 6) & 3u))) {      // by symbols: four waves see the chunk, one writes
        const int64_t n = ((int64_t)T * 64 - 64) * (int64_t)g.block_size + local;
        const uint64_t pos = (uint64_t)(n + g.packet_length);
        hit_block[off + i_out0 + i] = a.block_base + (pos >> lg_bs);
        hit_idx[off + i_out0 + i] = (uint32_t)pos & bs_mask;
    }
    const uint32_t key = ok ? local >> 5 : 0xffffffffu;
    const uint32_t prev = __shfl_up(key, 1);
    uint64_t leaders = __ballot(ok && (lane == 0 || key != prev));
    while (leaders) {
        uint32_t v0[kK3Batch], slot[kK3Batch];
        bool need_a[kK3Batch], need_b[kK3Batch];               // wave-uniform: hits in the first / second half of the word
        int nb = 0;
#pragma unroll
        for (int e = 0; e < kK3Batch; ++e) {
            v0[e] = 0; slot[e] = 0xffffffffu;
            if (leaders) {                                         // wave-uniform
                const uint32_t L = (uint32_t)__ffsll((unsigned long long)leaders) - 1;
                leaders &= leaders - 1;
                const uint32_t key_s = __builtin_amdgcn_readlane(key, L);
                if (lane < 32) tab[e][lane] = 0xffffffffu;
                if (ok && key == key_s) tab[e][local & 31] = i_out0 + i;   // same wave: LDS operations execute in order
                slot[e] = tab[e][31 - l32];                        // lane c of a block ends up with position 31-c
                v0[e] = (key_s << 5) - base_bit;                   // first bit of the word (staged: counted from the staged rows)
                // which halves of the word hold hits (bit c of the ballot = position 31 - c): a window at a half-word offset is
                // the low half of one stream word (positions 0 .. 15) and the high half of the next (16 .. 31), and a lone noise
                // hit -- eight to a tile in "all" -- needs one of the two
                const uint32_t hm = (uint32_t)__ballot(slot[e] != 0xffffffffu);
                need_a[e] = (hm & 0xffff0000u) != 0; need_b[e] = (hm & 0x0000ffffu) != 0;
                nb = e + 1;
            }
        }
        // two 64-symbol steps of up to four words per round (their loads are in flight together)
        for (uint32_t p0 = p_first; p0 < PS; p0 += by_symbols ? 512u : 128u) {
            uint32_t A[kK3Batch][2], B[kK3Batch][2];
#pragma unroll
            for (int e = 0; e < kK3Batch; ++e)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    A[e][k] = 0; B[e][k] = 0;
                    if (e < nb && p0 + 64 * k < PS) {
                        const uint32_t sy = p0 + 64 * k + sym_lane;
                        const uint32_t so = (sy < PS ? sy : PS - 1) * SL;
                        const uint32_t v = v0[e] + so;                             // window = 32 stream bits from bit v
                        // v0 is word-aligned and SL a multiple of 16: the window is one word, or the halves of two.  The second
                        // word only where it is needed -- every lane's window lies in a line of its own (symbols are SL bits
                        // = 4.5 words at chip 72 apart, a row's words in 16-byte pieces a KiB apart), so a load instruction
                        // costs the texture path as many line requests as it has lanes: half of them at SL = 16 mod 32,
                        // none at SL = 0 mod 32 (chip 32, 48, 64, 80, 96)
                        const bool two = (so & 16u) != 0;
                        if (LONG && staged) { A[e][k] = rows_lds[v >> 5]; B[e][k] = two ? rows_lds[(v >> 5) + 1] : 0u; }
                        else {
                            // ... and of the two only the half that holds hits (round 6: "all" at 4 GiB is 32 768 lone scm+ hits of
                            // 736 symbols, 36 M line requests of which 12 M fetched halves nobody read; the lanes that skip
                            // theirs do not load at all)
                            A[e][k] = (!two || need_a[e]) ? word_at(v) : 0u;
                            B[e][k] = (two && need_b[e]) ? word_at(v + 32) : 0u;
                        }
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
                            // K5's test follows in this workgroup: it reads the packets of the list's first pk_hits hits here
                            if (!LONG && slot[e] < pk_hits) *reinterpret_cast<uint32_t *>(pk_lds + slot[e] * PB + b0) = Y;
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

// Grid of K3 / k5_compact.  One workgroup per list = n_tiles per preamble; when that is one more than fills whole rounds
// of the chip's slots (a GiB of scm: 2048 tiles + the history tile, 2048 slots of 256 threads), workgroup 0 takes the
// history tile's short list after its own instead -- the straggler would otherwise wait for the first of the others
// to end.  Small launches keep one list per workgroup (a lone block: two lists side by side, not one after the other).
__host__ __device__ __forceinline__ bool k3_fold(uint32_t n_tiles, uint32_t n_pre, uint32_t slots)
{
    const uint64_t all = (uint64_t)n_tiles * n_pre, less = (uint64_t)(n_tiles - 1) * n_pre;
    return n_tiles > 1 && (all + slots - 1) / slots > (less + slots - 1) / slots;
}

// Where the list (tile T, preamble q) lies among all lists, preamble-major, from per-list counts [n_pre][n_tiles] and
// their sums over groups of 64 tiles [n_pre][n_groups]: S.off[q] = the counts of every list in front of q's group of T
// (S.off[n_pre] = the grand total), S.in[q] = the counts of T's group in front of T, S.cnt[q] = the list's own count,
// for every preamble q.  By a workgroup of 256 threads; ends with a barrier.
struct K3Scan {
    static constexpr uint32_t kRounds = 16;      // rounds whose totals are kept in LDS at a time (1024 sums)
    uint64_t off[kMaxPre + 1];
    uint32_t cnt[kMaxPre], in[kMaxPre], prev[kMaxPre];   // prev: the count of tile T-1's list
    uint32_t round[kRounds], excl[kMaxPre], excl_round[kMaxPre];
};

__device__ __forceinline__ void k3_list_scan(K3Scan &S, const uint32_t *counts, const uint32_t *gcnt, uint32_t n_tiles, uint32_t n_pre, uint32_t T, uint32_t tid)
{
    const uint32_t lane = tid & 63, wv = tid >> 6;
    const uint32_t n_groups = k2_groups(n_tiles), n_sums = n_pre * n_groups;
    const uint32_t v_first = wv * 64 + lane < n_sums ? gcnt[(wv * 64 + lane) * kGroupStride] : 0u;   // this wave's first round, in flight with the counts
    for (uint32_t q = wv; q < n_pre; q += 4) {           // wave q: the counts of preamble q in this tile's group, up to the tile
        const uint32_t t = lane;
        const uint32_t c = t < (T & 63) ? counts[q * n_tiles + (T & ~63u) + t] : 0u;
        const uint32_t mine = counts[q * n_tiles + T], before = T ? counts[q * n_tiles + T - 1] : 0u;
        const uint32_t inc = k3_wave_scan(c);
        if (lane == 63) { S.in[q] = inc; S.cnt[q] = mine; S.prev[q] = before; }
    }
    // exclusive scan over the flattened group sums [n_pre][n_groups] in rounds of 64, dealt to the four waves so that
    // their loads are in flight together; round totals meet in LDS
    const uint32_t n_rounds = (n_sums + 63) >> 6;
    uint64_t carry = 0;                                  // workgroup-uniform: sums of the passes before this one
    if (tid < kMaxPre) S.excl_round[tid] = 0xffffffffu;
    for (uint32_t r0 = 0; r0 < n_rounds; r0 += K3Scan::kRounds) {  // one pass unless a batch has more than 1024 group sums
        __syncthreads();
        for (uint32_t r = r0 + wv; r < n_rounds && r < r0 + K3Scan::kRounds; r += 4) {
            const uint32_t i = r * 64 + lane;
            const uint32_t v = r == wv ? v_first : i < n_sums ? gcnt[i * kGroupStride] : 0u;
            const uint32_t inc = k3_wave_scan(v);
            for (uint32_t q = 0; q < n_pre; ++q)         // the sums of this round before (q, this tile's group)
                if (q * n_groups + (T >> 6) == i) { S.excl[q] = inc - v; S.excl_round[q] = r; }
            if (lane == 63) S.round[r - r0] = inc;
        }
        __syncthreads();
        if (tid <= n_pre) {                      // thread q: the rounds in front of its round, 64 bits from here on
            const uint32_t q = tid;
            const uint32_t my_r = q < n_pre ? S.excl_round[q] : 0xffffffffu;
            uint64_t run = carry;
            for (uint32_t r = r0; r < n_rounds && r < r0 + K3Scan::kRounds; ++r) {
                if (r == my_r) S.off[q] = run + S.excl[q];
                run += S.round[r - r0];
            }
            if (q == n_pre) S.off[n_pre] = run;          // the total so far
        }
        __syncthreads();
        carry = S.off[n_pre];
    }
    __syncthreads();
}

// K5's test (k5_validate.h) of the `cnt` hits of list (T, q), which this workgroup has just written to slots off.. of the
// packed result: keep[] per hit, the number of survivors to listcnt / vgcnt, the list's slot to listoff (k5_compact
// works from those).  tbl: 1 KiB of LDS nobody else uses any more, pk_lds: the launch's dynamic LDS (same).  Ends in no
// barrier; call with all 256 threads.
//
// "The hit right before it in the same (preamble, block) list" of the list's FIRST hit is the last hit of tile T-1's
// list -- written by another workgroup, maybe not yet.  Two hits share a Decode call only if they are less than a block
// apart, so only that one hit can matter, and only when both map to the same call; then the two packets are compared
// where they come from, symbol by symbol in the bitstream (equal first dedupe_bytes bytes = equal first 8 x
// dedupe_bytes symbols: Decoder.Slice fills bytes with consecutive symbols, decode.go:363-366).
__device__ __forceinline__ void k5_flag_list(const K3Args &a, const ValRule &r, const SearchGeom &g, uint32_t T, uint32_t q, uint32_t cnt, uint64_t off,
                                             uint64_t total, uint16_t (*tbl)[256], uint32_t *s_red, uint8_t *pk_lds, const uint32_t *pos_lds, bool direct, bool edge, uint32_t lp, uint32_t lm, uint32_t tid)
{
    const uint32_t lane = tid & 63, wv = tid >> 6;
    const uint32_t PB = g.pkt_bytes, lg_bs = g.lg_block_size, lg_tw = 6 + g.lg_wpb;
    k5_tables(r, tbl, tid);
    int first_same = 0;                                   // workgroup-uniform
    if (edge) {     // lp, lm: the last position of tile T-1's list, the first of this one (loaded before the slicing)
        {
            const uint32_t bad = 64u << lg_bs;
            // same Decode call: (n + PacketLength) >> lg BS agree; n = (64 T - 64) BS + local
            const int64_t pp = ((int64_t)T * 64 - 128) * (int64_t)g.block_size + lp + g.packet_length;
            const int64_t pm = ((int64_t)T * 64 - 64) * (int64_t)g.block_size + lm + g.packet_length;
            if (lp < bad && lm < bad && (pp >> lg_bs) == (pm >> lg_bs)) {
                const uint32_t n_sym = (uint32_t)r.dedupe_bytes * 8 < g.packet_symbols ? (uint32_t)r.dedupe_bytes * 8 : g.packet_symbols;
                auto bit_at = [&](uint32_t tile, uint32_t v) {            // stream bit v counted from row 0 of `tile`
                    const uint32_t *tb = a.qt + ((size_t)tile << lg_tw);
                    const uint32_t row = v >> lg_bs, w = (v & (g.block_size - 1)) >> 5;
                    return (tb[((size_t)(row >> 6) << lg_tw) + ((w >> 2) << 8) + ((row & 63) << 2) + (w & 3)] >> (31 - (v & 31))) & 1u;
                };
                int differ = 0;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
                for (uint32_t p = tid; p < n_sym; p += 256)
                    differ |= (int)(bit_at(T - 1, lp + p * g.symbol_length) ^ bit_at(T, lm + p * g.symbol_length));
                if (tid == 0) s_red[0] = 0;
                k3_lds_barrier();
                if (differ) s_red[0] = 1;
                k3_lds_barrier();
                first_same = !s_red[0];
            }
        }
    }
    K3_STAMP(4);
    // The packets come through LDS, a round of up to 256 hits (and the hit in front of them) at a time: the first round
    // of a short-packet list as the slicing left them there (`direct`), the others in one coalesced copy with every load
    // in flight; the byte reads of the checks then run at LDS latency.
    const uint8_t *pkts = a.out + total * 12;
    const uint32_t *src = a.staging + ((size_t)T * g.n_pre + q) * a.cap;
    const uint32_t room = a.lds_bytes / PB - 1, per = room < 256u ? room : 256u;
    const bool words = (PB & 3) == 0;
    uint32_t kept = 0;
    // hit j of the list, packet at pkt (the one before it at pkt - PB), position sj (the one before it: sjm)
    auto test = [&](uint32_t j, const uint8_t *pkt, uint32_t sj, uint32_t sjm) {
        bool keep = k5_checks(r, tbl, pkt);
        if (keep && r.dedupe_bytes > 0) {
            if (j == 0) keep = !first_same;
            else {
                // same Decode call as the hit before it: (n + PacketLength) >> lg BS agree (the tile's base is a multiple of BS)
                const uint32_t c0 = (sjm + g.packet_length) >> lg_bs, c1 = (sj + g.packet_length) >> lg_bs;
                if (c0 == c1) {
                    const uint8_t *prev = pkt - PB;
                    bool same = true;
#pragma clang loop vectorize(disable) interleave(disable)
                    for (int i = 0; i < r.dedupe_bytes; ++i) same = same && prev[i] == pkt[i];
                    keep = !same;
                }
            }
        }
        a.keep[off + j] = keep ? 1 : 0;
        kept += keep ? 1u : 0u;
    };
    uint32_t j0 = 0;
    if (direct) {
        // No global load anywhere on this path: one would wait (vmcnt counts in order) until every packet store of the
        // wave has been acknowledged -- with all workgroups of the chip storing at once, 5 to 15 us.
        const uint32_t n = cnt < per ? cnt : per;
        k3_lds_barrier();                                 // the tables
        if (tid < n) test(tid, pk_lds + tid * PB, pos_lds[tid], tid ? pos_lds[tid - 1] : 0u);
        j0 = n;
        if (j0 >= cnt) goto counted;
    }
    __syncthreads();                                      // the four waves' packet stores have landed (and: the tables)
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (; j0 < cnt; j0 += per) {
        const uint32_t n = cnt - j0 < per ? cnt - j0 : per, lo = j0 ? j0 - 1 : 0u;
        const uint32_t n_bytes = (j0 + n - lo) * PB;
        const uint8_t *from = pkts + (off + lo) * PB;
        uint32_t sj = 0, sjm = 0;                         // this hit's position and the one before it, in flight during the copy
        if (tid < n && r.dedupe_bytes > 0) { sj = src[j0 + tid]; sjm = j0 + tid ? src[j0 + tid - 1] : 0u; }
        k3_lds_barrier();                                 // the previous round has been read
        if (words) {
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
            for (uint32_t t = tid; t < n_bytes / 4; t += 256) reinterpret_cast<uint32_t *>(pk_lds)[t] = reinterpret_cast<const uint32_t *>(from)[t];
        } else {
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
            for (uint32_t t = tid; t < n_bytes; t += 256) pk_lds[t] = from[t];
        }
        k3_lds_barrier();
        if (tid < n) test(j0 + tid, pk_lds + (j0 + tid - lo) * PB, sj, sjm);
    }
counted:
    for (int d = 32; d; d >>= 1) kept += __shfl_down(kept, d);
    K3_STAMP(5);
    k3_lds_barrier();                                     // first_same has been read
    if (lane == 0) s_red[wv] = kept;
    k3_lds_barrier();
    if (tid == 0) {
        const uint32_t n = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        a.listcnt[q * a.n_tiles + T] = n;
        a.listoff[q * a.n_tiles + T] = off;
        if (n) atomicAdd(&a.vgcnt[(q * k2_groups(a.n_tiles) + (T >> 6)) * kGroupStride], n);
    }
}

// one list = (tile T, preamble q), by a workgroup of 256 threads
__device__ __forceinline__ void k3_one_list(const K3Args &a, uint32_t T, uint32_t q, K3Scan &S, uint32_t (&tab)[4][kK3Batch][32],
                                            uint32_t *s_list, uint32_t *s_red, uint32_t *rows_lds, ValRule *s_rule, uint32_t tid)
{
    const SearchGeom &g = a.g;
    const uint32_t n_pre = g.n_pre;
    const uint32_t wv = tid >> 6;

    // the list is empty (three of four in "all", where only scm+ finds hits in noise): nothing to place, nothing to slice
    // -- leave before the prologue's loads and barriers.  Tile 0's first workgroup stays: it publishes the bases, the
    // total and the overflow word.
    if (a.counts[q * a.n_tiles + T] == 0 && !(T == 0 && q == 0)) {
        if (a.keep && tid == 0) a.listcnt[q * a.n_tiles + T] = 0;
        return;
    }

    K3_STAMP(0);
    // ---- prologue ----
    // K5's rule for this preamble, out of the kernel arguments into LDS NOW: indexed by q the compiler fetches its fields with
    // vector loads, and a vector load behind the slicing waits for every packet store of the wave to drain first
    static_assert(sizeof(ValRule) == 40, "ten words");
    if (a.keep && tid < 10) reinterpret_cast<uint32_t *>(s_rule)[tid] = reinterpret_cast<const uint32_t *>(&a.rule[q])[tid];
    const uint32_t ovf = *a.overflow;
    k3_list_scan(S, a.counts, a.gcnt, a.n_tiles, n_pre, T, tid);
    const uint64_t total = S.off[n_pre];
    if (T == 0 && q == 0 && tid <= n_pre) {
        const uint64_t v = tid < n_pre ? S.off[tid] : total;      // tile 0: nothing of its group in front of it
        a.offs_pre[tid] = v;
        a.h_offs_pre[tid] = v;
        if (tid == 0) *a.h_overflow = ovf;
    }
    // After an overflow the staging slots are incomplete (a wave whose sparse list overflowed counted hits it never
    // emitted): their contents must not be used as positions; the host re-runs the search anyway.
    if (ovf || total > a.out_cap) return;                // ... or grows the buffer and searches again
    K3_STAMP(1);
    const uint32_t cnt = S.cnt[q];
#if AMR_K3_DBG
    if (tid == 0 && blockIdx.x < 4096) k3_dbg[blockIdx.x * 8 + 7] = cnt;
#endif
    if (!cnt) {
        if (a.keep && tid == 0) a.listcnt[q * a.n_tiles + T] = 0;
        return;
    }

    const uint32_t *__restrict__ tbase = a.qt + ((size_t)T << (6 + g.lg_wpb));
    const uint32_t lg_bs = g.lg_block_size, lg_tw = 6 + g.lg_wpb;
    const uint32_t PS = g.packet_symbols, SL = g.symbol_length;
    const uint32_t bad = 64u << lg_bs;                 // defensive: never index the bitstream with a bad position
    const uint32_t wpb = g.wpb, cpr = wpb >> 2;
    const uint32_t n_rows = 1 + ((PS * SL + 32 + g.block_size - 1) >> lg_bs);     // rows a hit-word's windows can touch
    const bool by_symbols = PS > 128;                  // long packets (idm, netidm, "all"): rows staged in LDS
    const uint32_t *src = a.staging + ((size_t)T * n_pre + q) * a.cap;
    const uint64_t off = S.off[q] + S.in[q];
    // K5's look across the tile boundary (k5_flag_list): the two positions it needs are on their way during the slicing
    const bool edge = a.keep && T > 0 && S.prev[q] && s_rule->dedupe_bytes > 0;
    // short packets of whole dwords: the slicing leaves a copy of the first 256 packets in LDS for K5's test
    const uint32_t pk_hits = a.keep && !by_symbols && (g.pkt_bytes & 3) == 0 && (PS & 7) == 0 ? 256u : 0u;
    uint32_t edge_lp = 0, edge_lm = 0;
    if (edge) {
        const uint32_t c_prev = S.prev[q] < a.cap ? S.prev[q] : a.cap;
        edge_lp = a.staging[((size_t)(T - 1) * n_pre + q) * a.cap + c_prev - 1];
        edge_lm = src[0];
    }

    if (!by_symbols) {
        // ---- short packets (scm, scm+, r900 geometries: at most 128 symbols): the windows straight from the tiled
        // bitstream.  A packet's reach is a few KB of stream; staging rows for it costs more than the line fetches it
        // saves (1 GiB of scm: 32 us staged against 16-25 us direct).
        // The list in equal shares for the four waves (at most 64 hits at a time): a wave slices word by word, four words
        // per round of dependent loads, and noise hits sit one to a word -- 142 hits (1 GiB of scm: the average list) dealt
        // as 64 + 64 + 14 + 0 took 16 rounds, as 36 + 36 + 36 + 34 they take 9.
        const uint32_t share = cnt >= 256 ? 64u : (cnt + 3) >> 2;
        for (uint32_t i0 = wv * share; i0 < cnt; i0 += 4 * share)
            k3_chunk<false>(a, g, T, src, i0, i0 + share < cnt ? i0 + share : cnt, 0u, wv, off, total, tab[wv], nullptr, tbase, 0u, 0u, false, 0u, tid,
                            reinterpret_cast<uint8_t *>(rows_lds), pk_hits, s_list);
    } else {
        // ---- long packets, row by row.  The list comes into LDS first, a segment of kK3List positions at a time: what
        // row to stage next and where the hits of that row end are then LDS reads.  (Taken from global memory, every such
        // decision was a dependent load in front of the next row set -- the staging slot's first unsliced entry, then a
        // binary search -- 4 to 5 us per row set next to 2 us of staging and 1 us of slicing; "all" has 8 lone scm+ noise
        // hits per tile, each a row set of its own: K3 184 us per 4 GiB, most of it these loads.) ----
        for (uint32_t seg0 = 0; seg0 < cnt; seg0 += kK3List) {
            const uint32_t seg_n = cnt - seg0 < (uint32_t)kK3List ? cnt - seg0 : (uint32_t)kK3List;
            k3_lds_barrier();                                                    // the previous segment has been consumed
            for (uint32_t t = tid; t < seg_n; t += 256) s_list[t] = src[seg0 + t];
            k3_lds_barrier();
            uint32_t cur = 0;                                                   // workgroup-uniform: first entry not sliced yet
            while (cur < seg_n) {
                const uint32_t first = s_list[cur];
                if (first >= bad) { cur += 1; continue; }                       // defensive
                const uint32_t l0 = first >> lg_bs;                             // the row this round stages from
                // A sparse stretch -- the next (up to) 64 entries lie two or fewer to a row: lone noise hits, eight to a
                // tile in "all" (scm+'s 16-bit preamble), each of which would have a row set of n_rows KiB staged for its one
                // word (4 GiB of "all": 8 x 14 KB per tile, K3 150 us).  Their windows come straight from the bitstream
                // instead, four words to a round of loads, the symbols still shared out over the four waves.
                const uint32_t w_end = cur + 64 < seg_n ? cur + 64 : seg_n;
                const uint32_t w_last = s_list[w_end - 1];
                const bool sparse = w_last < bad && w_end - cur <= 2 * ((w_last >> lg_bs) - l0 + 1);
                uint32_t i_hi = w_end, base_bit = 0;
                if (!sparse) {
                    const uint32_t lim = (l0 + 1) << lg_bs;
                    uint32_t lo = cur + 1, hi = seg_n;                          // end of the hits that start in row l0 (positions ascend)
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_list[mid] < lim) lo = mid + 1; else hi = mid;
                    }
                    i_hi = lo;
                    k3_lds_barrier();                                            // the previous row set has been consumed
                    for (uint32_t t = tid; t < n_rows * cpr; t += 256) {
                        const uint32_t c = t / n_rows, r = t - c * n_rows, row = l0 + r;     // neighbouring threads: neighbouring rows of one chunk
                        const uint4 x = *reinterpret_cast<const uint4 *>(tbase + ((size_t)(row >> 6) << lg_tw) + ((size_t)c << 8) + ((row & 63) << 2));
                        *reinterpret_cast<uint4 *>(rows_lds + r * wpb + c * 4) = x;
                    }
                    k3_lds_barrier();
                    base_bit = l0 << lg_bs;                                     // stream bit (tile-local) of rows_lds[0], bit 31
                }
                // the four waves share a 64-hit chunk by SYMBOLS: the hits of a packet are one run of ~70 positions, i.e. one
                // wave's worth, and 736 symbols in one wave are six rounds one after the other while three waves watch
                for (uint32_t i0 = cur; i0 < i_hi; i0 += 64)
                    k3_chunk<true>(a, g, T, s_list, i0, i_hi, cur, wv, off, total, tab[wv], rows_lds, tbase, base_bit, wv * 128, true, seg0, tid,
                                   nullptr, 0u, nullptr, !sparse);
                cur = i_hi;
            }
        }
    }
    K3_STAMP(2);
    if (!a.keep) return;
    // ---- K5's test of the list, on the packets just written; the CRC tables take the place of the slicing tables
    k3_lds_barrier();
    K3_STAMP(3);
    k5_flag_list(a, *s_rule, g, T, q, cnt, off, total, reinterpret_cast<uint16_t (*)[256]>(&tab[0][0][0]), s_red, reinterpret_cast<uint8_t *>(rows_lds), s_list,
                 pk_hits != 0, edge, edge_lp, edge_lm, tid);
    K3_STAMP(6);
}

// grid (n_tiles, n_pre), one list per workgroup -- or, `fold` (k3_fold): (n_tiles - 1, n_pre), workgroup x takes the list of
// tile x + 1 and workgroup 0 afterwards that of tile 0 as well: the history tile, whose only hits are packets that START
// in the previous batch's last rows, a short list.
__global__ __launch_bounds__(256, 8) void k3_slice_words(const K3Args a)   // 8 workgroups per CU: 64 VGPRs (65 without the hint: 7)
{
    __shared__ K3Scan S;
    __shared__ uint32_t tab[4][kK3Batch][32];   // per wave and entry: staging index of the hit at bit b of the word, or ~0
    __shared__ uint32_t s_list[kK3List];        // long packets: the segment of the list being sliced
    __shared__ uint32_t s_red[4];
    __shared__ ValRule s_rule;
    extern __shared__ __attribute__((aligned(16))) uint32_t rows_lds[];          // [n_rows][wpb] words, stream order; K5: packets
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    const uint32_t passes = a.fold && blockIdx.x == 0 ? 2u : 1u;
#pragma clang loop unroll(disable)
    for (uint32_t pass = 0; pass < passes; ++pass) {
        if (pass) __syncthreads();
        // the thread index through an opaque move: whatever the list code derives from it is computed per pass, in place,
        // instead of being hoisted out of this loop to live (and spill) across all of it
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        k3_one_list(a, pass ? 0u : blockIdx.x + a.fold, blockIdx.y, S, tab, s_list, s_red, rows_lds, &s_rule, tid);
    }
}

// K5, second half: the survivors of every list move to their slot in a second packed buffer of K3's layout, in order.
// Same grid as K3, one workgroup per list (workgroup 0: two); the slot = the survivors of all lists in front (k3_list_scan over listcnt /
// vgcnt, as K3 places the hits themselves).
struct K5Args {
    const uint8_t *in;          // K3's packed result [hit_block u64 x n | hit_idx u32 x n | pkt x n]
    uint8_t *out;               // same layout, n' = surviving hits
    const uint64_t *offs_pre;   // [n_pre+1] from k3_slice
    uint64_t *offs_val;         // [n_pre+1] offsets into the validated list (device)
    uint64_t *h_offs_val;       // the same in pinned host memory
    const uint8_t *keep;        // [cap] from K3's last stage
    const uint32_t *counts;     // [n_pre][n_tiles] hits per list (K2)
    const uint32_t *listcnt;    // [n_pre][n_tiles] survivors per list
    const uint64_t *listoff;    // [n_pre][n_tiles] slot of the list in `in`
    const uint32_t *vgcnt;      // [n_pre][n_groups] survivors per group of 64 tiles
    const uint32_t *overflow;   // K2's overflow word: the host searches again, nothing here is used
    uint64_t cap;               // hits the buffers hold
    uint32_t n_pre, n_tiles, pkt_bytes;
    uint32_t fold;              // as K3Args::fold
};

__device__ __forceinline__ void k5_compact_list(const K5Args &a, uint32_t T, uint32_t q, K3Scan &S, uint32_t *wcnt)
{
    const uint32_t n_pre = a.n_pre;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool first = T == 0 && q == 0;
    const uint64_t total = a.offs_pre[n_pre];
    if (*a.overflow != 0 || total > a.cap || total == 0) {
        if (first && threadIdx.x <= n_pre) { a.offs_val[threadIdx.x] = 0; a.h_offs_val[threadIdx.x] = 0; }
        return;
    }
    const uint32_t cnt = a.counts[q * a.n_tiles + T];
    if (cnt == 0 && !first) return;
    k3_list_scan(S, a.listcnt, a.vgcnt, a.n_tiles, n_pre, T, threadIdx.x);
    const uint64_t kept = S.off[n_pre];
    if (first && threadIdx.x <= n_pre) {                 // tile 0: nothing of its group in front of it
        const uint64_t v = threadIdx.x < n_pre ? S.off[threadIdx.x] : kept;
        a.offs_val[threadIdx.x] = v;
        a.h_offs_val[threadIdx.x] = v;
    }
    if (cnt == 0 || S.cnt[q] == 0) return;
    const uint64_t off = a.listoff[q * a.n_tiles + T];
    uint64_t slot = S.off[q] + S.in[q];                  // of the list's next survivor
    const uint64_t *ib = reinterpret_cast<const uint64_t *>(a.in);
    const uint32_t *ii = reinterpret_cast<const uint32_t *>(a.in + total * 8);
    uint64_t *ob = reinterpret_cast<uint64_t *>(a.out);
    uint32_t *oi = reinterpret_cast<uint32_t *>(a.out + kept * 8);
    for (uint32_t j0 = 0; j0 < cnt; j0 += 256) {
        const uint32_t j = j0 + threadIdx.x;
        const bool keep = j < cnt && a.keep[off + j] != 0;
        const uint64_t m = __ballot(keep);
        __syncthreads();                                  // wcnt of the previous round has been read
        if (lane == 0) wcnt[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t i = 0; i < 4; ++i) { before += i < wv ? wcnt[i] : 0u; all += wcnt[i]; }
        if (keep) {
            const uint64_t rank = slot + before + (uint32_t)__popcll(m & ((1ull << lane) - 1));
            const uint64_t gi = off + j;
            const uint8_t *ip = a.in + total * 12 + gi * a.pkt_bytes;
            uint8_t *op = a.out + kept * 12 + rank * a.pkt_bytes;
            ob[rank] = ib[gi];
            oi[rank] = ii[gi];
            for (uint32_t i = 0; i < a.pkt_bytes; ++i) op[i] = ip[i];
        }
        slot += all;
    }
}

// grid and tiles as k3_slice_words
__global__ __launch_bounds__(256) void k5_compact(const K5Args a)
{
    __shared__ K3Scan S;
    __shared__ uint32_t wcnt[4];
    const uint32_t passes = a.fold && blockIdx.x == 0 ? 2u : 1u;
#pragma clang loop unroll(disable)
    for (uint32_t pass = 0; pass < passes; ++pass) {
        if (pass) __syncthreads();
        k5_compact_list(a, pass ? 0u : blockIdx.x + a.fold, blockIdx.y, S, wcnt);
    }
}

// last kernel of a batch whose K3 (K4, K5) ran on the second stream: publishes the batch ticket
__global__ void k_done(uint64_t *flag, uint64_t value, uint64_t *dev_flag)
{
    __hip_atomic_store(dev_flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // k_hist_update of the next batch waits here
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// First kernel of a tail that was enqueued AHEAD of time (submit of the following batch): holds the second stream until
// that batch's K1 has all its waves on the chip (K1Args::started), plus `delay` ticks of the 100 MHz clock for the other
// XCDs' dispatchers.  One lane; it sleeps between polls and needs no LDS and 8 registers, so it fits next to a full K1.
//
// The wait is bounded (`timeout` ticks): the K1 launch it waits for sits behind this batch's search on the compute stream,
// and that stream may be held up for any length of time (foreign work on a caller's stream, amr_set_stream; a
// serialising tool).  A gate that gives up must NOT let the tail through as if nothing had happened -- nothing else
// orders K3 against this batch's K2, which may still be running: it sets bit 2 (value 4) of the batch's overflow word.
// K3 / K4 / K5 then leave the result alone (they do for any non-zero overflow word), the word reaches the host with the
// batch ticket, and amr_collect searches the batch again on the COMPUTE stream, in order behind its K2.
#ifndef AMR_GATE_CLK
#define AMR_GATE_CLK 0      // diagnostic builds: the gate measures the shader clock while it waits (cycles per 100 MHz tick)
#endif
#if AMR_GATE_CLK
__device__ unsigned long long k_gate_clk[4096];
#endif
__global__ void k_gate(const uint64_t *flag, uint64_t value, uint32_t delay, uint64_t timeout, uint32_t *overflow)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
#if AMR_GATE_CLK
    const uint64_t c0 = __builtin_readcyclecounter();
#endif
    bool open = false;                   // timeout 0 (test hook AMR_GATE_TIMEOUT_US=0): give up without looking
    while (timeout) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= value) { open = true; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 >= timeout) break;
        __builtin_amdgcn_s_sleep(16);
    }
    if (!open) { atomicOr(overflow, kOvfGate); return; }
    const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
#if AMR_GATE_CLK
    k_gate_clk[(value & 2047) * 2] = __builtin_readcyclecounter() - c0;
    k_gate_clk[(value & 2047) * 2 + 1] = t1 - t0;
#endif
    while (__builtin_amdgcn_s_memrealtime() - t1 < delay) __builtin_amdgcn_s_sleep(8);
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
