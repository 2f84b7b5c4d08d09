// Shared by the search kernels (k2_walk.h, k2_search.h) and K3 (k3_slice.h): geometry, arguments, the tiled bitstream
// layout, the state update that rides inside a search launch.
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
constexpr int kR900Digits = 42;   // PayloadSymbols, r900.go:30 (K4, k4_r900.h)

// Every group sum in a cache line of its own: sum i lives at word i * kGroupStride.  They are the targets of one
// atomicAdd per list from workgroups that all finish at about the same time; packed (round 3: the 33 sums of 1 GiB of
// scm in two lines) the 2048 additions queue up at one memory channel -- 12 us during which the stores of everybody
// else's packets wait behind them (measured on K3's survivor sums, profiles/r04/k3_phases.txt).
#ifndef AMR_GSTRIDE
#define AMR_GSTRIDE 32
#endif
constexpr uint32_t kGroupStride = AMR_GSTRIDE;

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
    // early search (K2Args::k1_flags): the batch's last rows may still be on their way when this workgroup starts -- it waits
    // for the flag of the wave-tile that holds them and reads them past its caches (sc1)
    const uint32_t *k1_flag;     // &done_flags[last wave-tile], or null
    uint32_t k1_flag_value;
};

__device__ __forceinline__ size_t qt_index_fwd(uint64_t R, uint32_t w, uint32_t lg_wpb)   // = qt_index, defined below
{
    return ((R >> 6) << (6 + lg_wpb)) + ((size_t)(w >> 2) << 8) + ((R & 63) << 2) + (w & 3);
}

// the work, by a workgroup of `nt` threads with hr * wpb words of LDS at tmp
// Spin (one lane sleeps and polls with agent-scope loads, i.e. past L1) until *flag >= value; bounded: `ticks` of the
// 100 MHz clock.  false: gave up.
__device__ __forceinline__ bool k1_flag_wait(const uint32_t *flag, uint32_t value, uint64_t ticks)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) return false;
        __builtin_amdgcn_s_sleep(8);
    }
    return true;
}
constexpr uint64_t kK1FlagTicks = 50000000ull;      // 0.5 s: a K1 launch takes 0.2 ms; beyond this something is broken
constexpr uint32_t kOvfGate = 4u;                   // overflow word, bit 2: a device-side wait gave up -- the host searches the
                                                    // batch again on the compute stream, in order (bits 0 / 1: K2's capacities)

__device__ __forceinline__ void hist_body(const HistArgs &a, uint32_t *tmp, uint32_t nt)
{
    const uint32_t tid = threadIdx.x;
    if (a.k1_flag) {     // early search: the rows below are ready when the wave-tile that holds them has said so
        if (tid == 0) (void)k1_flag_wait(a.k1_flag, a.k1_flag_value, kK1FlagTicks);   // (giving up is reported by the searching
        __syncthreads();                                                              // waves, which wait for the same launch)
    }
    for (uint32_t i = tid; i < a.carry_bytes / 16; i += nt)
        reinterpret_cast<uint4 *>(a.carry_dst)[i] = reinterpret_cast<const uint4 *>(a.carry_src)[i];
    if (tid == nt - 1) *a.ovf_next = 0;
    for (uint32_t i = tid; i < a.gcnt_words / kGroupStride; i += nt) a.gcnt_next[i * kGroupStride] = 0;   // the sums, not the padding
    const uint32_t n = a.hr << a.lg_wpb;
    for (uint32_t i = tid; i < n; i += nt) {
        const uint32_t j = i >> a.lg_wpb, w = i & (a.wpb - 1);
        // new history row j = stream row (n_blocks - hr + j) of the batch; negative -> old history
        const int64_t srow = (int64_t)64 + a.n_blocks - a.hr + j;  // tiled row index (tile 0 rows 0..63 = old history)
        const uint32_t *src = a.qt + qt_index_fwd((uint64_t)srow, w, a.lg_wpb);
        tmp[i] = a.k1_flag ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
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

struct K2Args {
    const uint32_t *qt;    // tiled bitstream, tile 0 = history tile
    uint32_t *counts;      // [n_pre][n_tiles]
    uint32_t *gcnt;        // [n_pre][n_groups] sums of counts over groups of 64 tiles, kGroupStride words apart (atomicAdd; zero before K2 runs)
    uint32_t *staging;     // [n_tiles][n_pre][cap] tile-local positions (row*BS + bit), ascending
    uint32_t *overflow;    // set to 1 when a tile found more than cap hits for a preamble
    uint32_t n_tiles;      // tiles searched: ceil(n_blocks/64) + 1 (history tile first)
    uint32_t cap;
    int64_t n_lo, n_hi;    // valid positions: n_lo <= n < n_hi, n relative to batch sample 0
    unsigned long long *dbg;   // harness builds only (AMR_K2W_DBG): 16 words of timestamps per workgroup, or null
    // pinned host word that receives `started_value` when the search starts, i.e. when everything before it on the
    // stream (this batch's K1) has finished: the host then launches the previous batch's K3 on the second stream
    uint64_t *started;
    uint64_t started_value;
    // pipelined callers: the state update rides along as one more workgroup (tile index n_tiles; the hist.defer_wgs
    // workgroups behind it copy the deferred blocks) instead of a 5 us kernel of its own behind the search.  It carries
    // no completion ticket (the search is still running
    // when it is done; a ticket from inside the kernel would also need every workgroup to release its writes, an L2
    // write-back each): the host takes "the next search has started" or "the stream is idle" as the signal instead.
    // Early search: this launch runs off the compute stream, NEXT to the batch's K1 (behind a gate: every K1 wave is on
    // the chip by then); the wave of tile T waits for done_flags[T - 1] (its rows) and done_flags[T] (the head of the tile
    // behind it) of K1Args before it loads anything, and loads past its caches (sc1).  null: the launch follows K1 in
    // stream order, as always.
    const uint32_t *k1_flags;
    uint32_t k1_flag_value;
    uint32_t do_hist;
    uint32_t walk_pids;        // k2_walk.h: the preamble id of each of rtlamr's four preambles (scm, scm+, idm, r900), 8 bits each
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

// k2_walk.h
constexpr int kK2WTaps = 16;              // taps applied to every position
constexpr int kK2WList = 192;             // (key, mask) entries per wave
constexpr int kK2WWaves = 4;              // waves (tiles) per workgroup
// The walk loads its ring chunks unconditionally (k2_walk.h: a load inside a branch costs a vmcnt(0) per group), so at the
// end of a row it fetches up to NEED + PF chunks of 1 KiB that nobody uses -- for the last row of the last searched tile
// they lie behind the tile that follows it.  The bitstream allocation carries this many bytes of slack behind its
// tiles (ensure_qt); k2_search_walk asserts that its over-read fits.
constexpr size_t kQtSlackBytes = 64 * 1024;

inline size_t k2_walk_lds_bytes(uint32_t hist_words)
{
    const size_t per_wave = (size_t)kK2WList * 2 + 2 * 4 * 64;      // list + counts + bases (four preambles)
    const size_t need = per_wave * kK2WWaves;
    return (need > hist_words ? need : hist_words) * 4;
}

// K3: dynamic LDS = the rows of the bitstream the windows of one hit-word can touch (its own row + the packet's reach
// + one word of funnel shift), in words
// + (validation on) room for the packets of 256 hits and the one before them, which K5's test reads from LDS
inline size_t k3_lds_bytes(const SearchGeom &g, bool validate = false)
{
    const size_t n_rows = 1 + (((size_t)g.packet_symbols * g.symbol_length + 32 + g.block_size - 1) >> g.lg_block_size);
    const size_t rows = (n_rows * g.wpb + 4) * 4, pk = validate && g.packet_symbols <= 128 ? 257 * (size_t)g.pkt_bytes : 0;
    return rows > pk ? rows : pk;
}

constexpr int kListCap = 448;  // k2_search_fast: (key, mask) entries per wave

inline size_t k2_fast_lds_bytes(uint32_t wpb, int npre, int nwv)
{
    return ((size_t)wpb * 65 + 4 * kListCap * 2 + 2 * (size_t)npre * 64 * nwv + 8) * 4;
}

}  // namespace amr
