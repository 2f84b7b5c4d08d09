// amr_pipeline.hip -- the batch pipeline behind the hot-path entry points of include/amrdemod.h: submit / collect over
// four slots and three streams, wave-tile deferral (4c), priming of a shard (5).  Everything that
// touches samples runs on the GPU; there is NO CPU fallback.
#include <sched.h>

#include "amr_host.h"
#include "launch.h"
#include "k3_slice.h"
#include "k4_r900.h"
#include "k1_single.h"
#include "k3_stale.h"

using namespace amr_host;

namespace {

// Timing events ride on the kernel dispatches themselves (hipExtLaunchKernelGGL start/stop events): a separate
// hipEventRecord costs a ~6 us bubble on the stream each, four of them per batch were 6 % of a 1 GiB step.
// The kernels are launched through launch.h: one translation unit per kernel family.

// Make room for `tiles` tiles in the bitstream of slot `s` (the slot being submitted: nothing of it is in flight),
// keeping its tile 0 = the history the previous batch left there.  The next slot in the ring (`other`, never one with a
// batch in flight) only has to EXIST here, because this batch's state update writes the next history tile into it; it
// is grown when its own batch is submitted.
amr_status ensure_qt(amr_handle *h, Slot &s, Slot &other, size_t tiles)
{
    const size_t tile_words = (size_t)64 * h->sg.wpb;
    // stream-ordered copies / memsets on purpose: the handle's stream is non-blocking, so a null-stream hipMemcpy /
    // hipMemset (asynchronous to the host for device memory) would race with the kernels enqueued right after
    if (tiles > s.qt_tiles) {
        uint32_t *nq = nullptr;
        hipError_t e = hipMalloc((void **)&nq, tiles * tile_words * 4 + amr::kQtSlackBytes);   // slack: see kQtSlackBytes
        if (e != hipSuccess) return fail(AMR_ENOMEM, "hipMalloc(qt)", e);
        if (s.d_qt) {
            HIP_TRY(hipMemcpyAsync(nq, s.d_qt, tile_words * 4, hipMemcpyDeviceToDevice, h->stream));
            AMR_TRY(sync_compute(h));
            HIP_TRY(hipFree(s.d_qt));
        } else {
            HIP_TRY(hipMemsetAsync(nq, 0, tile_words * 4, h->stream));
        }
        s.d_qt = nq;
        s.qt_tiles = tiles;
    }
    if (!other.d_qt) {
        hipError_t e = hipMalloc((void **)&other.d_qt, tiles * tile_words * 4 + amr::kQtSlackBytes);
        if (e != hipSuccess) { other.d_qt = nullptr; return fail(AMR_ENOMEM, "hipMalloc(qt)", e); }
        HIP_TRY(hipMemsetAsync(other.d_qt, 0, tile_words * 4, h->stream));
        other.qt_tiles = tiles;
    }
    return AMR_OK;
}

// (Re)allocate everything sized by the hit capacity of a slot.
amr_status alloc_hit_buffers(amr_handle *h, Slot &s)
{
    AMR_TRY(dev_realloc(s.d_out, s.out_cap * (12 + h->sg.pkt_bytes)));
    if (h->r900_pid >= 0) AMR_TRY(dev_realloc(s.d_r900, s.out_cap * amr::kR900Digits));
    if (h->validate) {
        AMR_TRY(dev_realloc(s.d_val, s.out_cap * (12 + h->sg.pkt_bytes)));
        AMR_TRY(dev_realloc(s.d_keep, s.out_cap));
    }
    return AMR_OK;
}

amr_status ensure_capacity(amr_handle *h, Slot &s, Slot &other, size_t n_blocks)
{
    const size_t bt = (n_blocks + 63) / 64;   // batch tiles
    const size_t st = bt + 1;                 // tiles searched
    {   // hipMalloc / hipFree wait for the whole device: with batches in flight, launch their pending K3.. first (see
        // sync_compute) -- this happens on the first use of each slot and when a batch is larger than any before
        const uint32_t gw0 = 2 * amr::kGroupStride * (uint32_t)(amr::k2_groups((uint32_t)st) * h->sg.n_pre);
        const bool grows = bt + 2 > s.qt_tiles || !other.d_qt || st > s.cnt_tiles || gw0 > s.gcnt_words || gw0 > other.gcnt_words ||
                           st > s.staging_tiles || !s.d_out || (h->validate && !s.d_val);
        if (grows && h->n_pending) AMR_TRY(sync_compute(h));
    }
    AMR_TRY(ensure_qt(h, s, other, bt + 2));
    if (st > s.cnt_tiles) {     // one "done" word per K1 wave-tile (early search), zero: below every ticket
        AMR_TRY(dev_realloc(s.d_k1flags, st));
        HIP_TRY(hipMemsetAsync(s.d_k1flags, 0, st * 4, h->stream));
    }
    if (st > s.cnt_tiles) {     // per list: hits (K2), then survivors of K5's test and the list's slot (K3)
        AMR_TRY(dev_realloc(s.d_counts, 2 * st * h->sg.n_pre));
        AMR_TRY(dev_realloc(s.d_listoff, st * h->sg.n_pre));
        s.cnt_tiles = st;
    }
    // group sums: this slot and the next one (the hist kernel of this batch zeroes those of the next), kept zero between
    // uses.  Neither holds a batch in flight; the slots that do keep what their own batch was sized for.
    // (two halves: the hits K2 counts, the survivors K3's last stage counts when validation is on)
    const uint32_t gw = 2 * amr::kGroupStride * (uint32_t)(amr::k2_groups((uint32_t)st) * h->sg.n_pre);
    Slot *both[2] = {&s, &other};
    for (Slot *slp : both) {
        Slot &sl = *slp;
        if (gw <= sl.gcnt_words) continue;
        AMR_TRY(sync_compute(h));
        AMR_TRY(dev_realloc(sl.d_gcnt, gw));
        HIP_TRY(hipMemsetAsync(sl.d_gcnt, 0, (size_t)gw * 4, h->stream));   // ordered before the K2 that adds into it
        sl.gcnt_words = gw;
    }
    if (st > s.staging_tiles) {
        AMR_TRY(dev_realloc(s.d_staging, st * h->sg.n_pre * s.stage_cap));
        s.staging_tiles = st;
    }
    if (s.out_cap == 0) s.out_cap = h->init_hit_cap;
    if (!s.d_out || (h->validate && !s.d_val)) AMR_TRY(alloc_hit_buffers(h, s));
    return AMR_OK;
}

// The search of the batch held by slot s in two parts: K2 on stream `st`, then K3 (+ K4, K5) -- the "tail" -- on the
// same stream at once (enqueue_search; also every re-run after a capacity overflow) or later on the second stream
// (pipelined callers: collect() launches it when the next batch's K1 has finished, so that it runs next to that
// batch's K2 instead of in front of its K1).  `split`: K2 gets a stop event of its own for timing level 2.
// The bits Decoder.Slice never clears in a last byte of fewer than 8 symbols (k3_stale.h), for the hits of slot s.
amr_status enqueue_stale(amr_handle *h, Slot &s, hipStream_t st)
{
    amr::StaleArgs sa{};
    sa.out = s.d_out; sa.offs_pre = s.d_offs_pre; sa.overflow = s.d_overflow; sa.cap = s.out_cap;
    sa.carry_in = h->d_pkt_carry + s.carry_in_slot; sa.carry_out = h->d_pkt_carry + (&s - h->slot);
    sa.n_pre = h->sg.n_pre; sa.pkt_bytes = h->sg.pkt_bytes; sa.r = h->sg.packet_symbols & 7u;
    hipLaunchKernelGGL(amr::k_stale_bits, dim3((unsigned)((s.out_cap + 255) / 256)), dim3(256), 0, st, sa);
    HIP_TRY(hipGetLastError());
    AMR_DBG(st, "k_stale_bits");
    return AMR_OK;
}

amr_status enqueue_k2(amr_handle *h, Slot &s, hipStream_t st, bool rerun, bool dense, bool split,
                      const amr::HistArgs *fold = nullptr, bool *folded = nullptr, bool early = false)
{
    if (folded) *folded = false;
    const uint32_t n_pre = h->sg.n_pre;
    const uint32_t bs = (uint32_t)h->geom.block_size;
    amr::K2Args k2{};
    k2.qt = s.d_qt;
    k2.counts = s.d_counts;
    k2.gcnt = s.d_gcnt;
    k2.staging = s.d_staging;
    k2.overflow = s.d_overflow;
    k2.n_tiles = s.n_tiles;
    k2.cap = s.stage_cap;
    k2.n_lo = -(int64_t)h->geom.packet_length;
    k2.n_hi = (int64_t)s.n_blocks * bs - (int64_t)h->geom.packet_length;
    k2.g = h->sg;
    const bool t2 = s.timed >= 2;
    hipEvent_t k2stop = (t2 && split) ? s.ev_k2 : nullptr;
    k2.started = rerun ? nullptr : &h->h_flags[0];
    if (early) { k2.k1_flags = s.d_k1flags; k2.k1_flag_value = (uint32_t)s.ticket; }   // next to K1, tile by tile (submit)
    k2.started_value = s.ticket;
    // the overflow word is zeroed by the previous batch's k_hist_update; only a re-run has to do it here
    if (rerun) {
        HIP_TRY(hipMemsetAsync(s.d_overflow, 0, 4, st));
        HIP_TRY(hipMemsetAsync(s.d_gcnt, 0, (size_t)s.gcnt_words * 4, st));
    }
    // the batch's state update as workgroup number n_tiles of the search launch, the copies of deferred blocks as the
    // workgroups behind it (see K2Args::do_hist)
    uint32_t extra = 0;
    if (fold) {
        k2.do_hist = 1;
        k2.hist = *fold;
        k2.hist.adone_flag = nullptr;     // no ticket from inside the search (see K2Args::do_hist)
        k2.hist.done_flag = nullptr;
        extra = 1u + fold->defer_wgs;
        if (folded) *folded = true;
    }
    const uint32_t wgs = s.n_tiles + extra;
    hipEvent_t k2start = t2 ? s.ev_s : nullptr;
    hipError_t le = hipSuccess;
    // the walk search (k2_walk.h): one wave walks a whole tile out of global memory; every set of rtlamr's own preambles
    // (scm, scm+, idm / netidm, r900: their first sixteen symbols are compile-time constants there) at every BlockSize
    // from 512 to 8192
    bool walk_ok = !h->dense_search && !dense && n_pre <= 4 && h->sg.wpb >= 16 && h->sg.wpb <= 256;
    uint32_t walk_set = 0;
    int last_kind = -1;
    for (uint32_t q = 0; q < n_pre && walk_ok; ++q) {
        const int kind = amr::k2_walk_kind_of(h->sg.pre_len[q], h->sg.pre_bits[q]);
        walk_ok = kind >= 0;
        if (kind >= 0) { walk_set |= 1u << kind; k2.walk_pids |= q << (8 * kind); last_kind = kind; }
    }
    if (walk_ok && s.inwave && !rerun && !dense && !early) {
        // K1 searched every tile itself (k1_search.h) except row 63's last words and the history tile: the history tile + the
        // state update as a launch of the row kernel over ONE tile, then the clean-up of the rows 63
        amr::K2Args kh = k2;
        kh.n_tiles = 1;
        const size_t lds = amr::k2_walk_lds_bytes(h->hist_rows * h->sg.wpb);
        if (!amr::launch_k2_row(h->sg.symbol_length, (uint32_t)last_kind, extra, lds, st, k2start, nullptr, kh, &le) ||
            !amr::launch_k2_cleanup(h->sg.symbol_length, (uint32_t)last_kind, st, nullptr, k2stop, k2))
            return fail(AMR_EHIP, "in-wave search: no clean-up kernel for this geometry");
        HIP_TRY(le);
        HIP_TRY(hipGetLastError());
        AMR_DBG(st, "k2_cleanup");
        return AMR_OK;
    }
    if (walk_ok) {
        const uint32_t n_wg = (s.n_tiles + amr::kK2WWaves - 1) / amr::kK2WWaves + extra;
        const uint32_t grid = 8u * ((n_wg + 7u) / 8u);   // XCD-contiguous tile order: 8 equal runs
        size_t lds = amr::k2_walk_lds_bytes(h->hist_rows * h->sg.wpb);
        // several preambles: the walk holds 168 registers, three waves of it leave a SIMD nothing -- and K3.. of the previous
        // batch, which run next to this search, wait for its waves to retire.  LDS the walk does not need limits it to
        // fewer workgroups per compute unit (hook AMR_K2W_LDS_KB; 0 = off)
        if (n_pre > 1 && h->k2w_lds_min > lds) lds = h->k2w_lds_min;
        // one preamble: the whole row in registers (rows of 256 words: two lanes per row), the look-ahead from the
        // neighbour lane (k2_row.h); it sizes its own grid (one or two waves per tile) around the `extra` workgroups
        walk_ok = (n_pre == 1 && amr::launch_k2_row(h->sg.symbol_length, (uint32_t)last_kind, extra, lds, st, k2start, k2stop, k2, &le)) ||
                  amr::launch_k2_walk(h->sg.symbol_length, walk_set, grid, lds, st, k2start, k2stop, k2, &le);
    }
    // fallbacks: the list-based kernel splits a row's words over 4 or 8 waves, 4 or 8 words per step: rows of fewer
    // than 16 words (BlockSize 256: scm+ alone at chip length 8) and more than four preambles go through the dense kernel
    if (walk_ok) {
    } else if (!h->dense_search && !dense && n_pre <= 4 && h->sg.wpb >= 16) {
        const int nwv = h->sg.wpb >= 64 ? 8 : 4;   // a wave needs at least JW words of a row: 8 x 8 or 4 x 4
        (void)amr::launch_k2_fast(n_pre, nwv, wgs, amr::k2_fast_lds_bytes(h->sg.wpb, (int)n_pre, nwv), st, k2start, k2stop, k2, &le);
    } else {
        amr::launch_k2_dense(wgs, ((size_t)h->sg.wpb * 65 + 8) * 4, st, k2start, k2stop, k2, &le);
    }
    HIP_TRY(le);
    HIP_TRY(hipGetLastError());
    AMR_DBG(st, "k2_search");
    return AMR_OK;
}

amr_status enqueue_tail(amr_handle *h, Slot &s, hipStream_t st, bool split)
{
    const uint32_t n_pre = h->sg.n_pre;
    const uint32_t bs = (uint32_t)h->geom.block_size;
    const bool t2 = s.timed >= 2;
    if (s.pack_pending) {   // a multi-GPU gather's pack kernel may still be reading the result this tail overwrites
        HIP_TRY(hipStreamWaitEvent(st, s.ev_pack, 0));
        s.pack_pending = false;
    }
    amr::K3Args k3{};
    k3.qt = s.d_qt; k3.counts = s.d_counts; k3.gcnt = s.d_gcnt; k3.staging = s.d_staging;
    k3.out = s.d_out; k3.offs_pre = s.d_offs_pre; k3.h_offs_pre = s.h_off; k3.h_overflow = s.h_ovf;
    k3.out_cap = s.out_cap; k3.overflow = s.d_overflow;
    k3.block_base = s.calls_base; k3.n_tiles = s.n_tiles; k3.cap = s.stage_cap; k3.g = h->sg;
    if (h->validate) {   // the checksum test + repeated-packet removal of every hit, as the last stage of K3's workgroups
        k3.keep = s.d_keep;
        k3.listcnt = s.d_counts + s.cnt_tiles * n_pre;
        k3.listoff = s.d_listoff;
        k3.vgcnt = s.d_gcnt + s.gcnt_words / 2;
        for (uint32_t q = 0; q < n_pre; ++q) k3.rule[q] = h->rules[q];
    }
    hipEvent_t k3e0 = (t2 && split) ? s.ev_t : nullptr, k3e1 = t2 ? s.ev2 : nullptr;
    size_t k3lds = amr::k3_lds_bytes(h->sg, h->validate);
    if (h->k3_lds_min > k3lds) k3lds = h->k3_lds_min;     // A/B hook AMR_K3_LDS_KB: fewer K3 workgroups per compute unit
    k3.lds_bytes = (uint32_t)k3lds;
    HIP_TRY(hipFuncSetAttribute((const void *)amr::k3_slice_words, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k3lds));
    // one workgroup per (tile, preamble) list (every list of a tile in one workgroup with shared row staging measured slower
    // on the four-preamble decoder: 188 against 173 us per 4 GiB); k3_fold: when the history tile's workgroup would be the
    // one too many for whole rounds of the chip, workgroup 0 takes its list as well
    k3.fold = amr::k3_fold(s.n_tiles, n_pre, (uint32_t)h->n_cus * 8u) ? 1u : 0u;
    k3.prio = h->k3_prio;
    hipExtLaunchKernelGGL(amr::k3_slice_words, dim3(s.n_tiles - k3.fold, n_pre), dim3(256), k3lds, st, k3e0, k3e1, 0, k3);
    HIP_TRY(hipGetLastError());
    AMR_DBG(st, "k3_slice");
    if (h->sg.packet_symbols & 7u) AMR_TRY(enqueue_stale(h, s, st));
    if (h->r900_pid >= 0) {
        amr::K4Args k4{};
        k4.iq = s.d_iq; k4.hist = h->d_iqhist[s.iqhist_buf]; k4.lut = h->d_lut; k4.out_packed = s.d_out;
        k4.offs_pre = s.d_offs_pre; k4.overflow = s.d_overflow; k4.digits = s.d_r900; k4.cap = s.out_cap; k4.block_base = s.calls_base;
        k4.n_pre = n_pre; k4.pid = (uint32_t)h->r900_pid; k4.hist_valid = s.iqhist_valid;
        k4.block_size = bs; k4.lg_block_size = h->sg.lg_block_size; k4.packet_length = (uint32_t)h->geom.packet_length;
        k4.preamble_length = (uint32_t)h->geom.preamble_length; k4.symbol_length = (uint32_t)h->geom.symbol_length;
        k4.chip_length = (uint32_t)h->geom.chip_length;
        // the hit count is only known on the device: one 64-lane block per 64 possible hits, the surplus exits at once
        hipLaunchKernelGGL(amr::k4_r900_digits, dim3((unsigned)((s.out_cap + 63) / 64), amr::kK4Split), dim3(64), 0, st, k4);
        HIP_TRY(hipGetLastError());
        AMR_DBG(st, "k4_r900_digits");
    }
    if (h->validate) {   // ordered compaction of the hits K3's last stage kept into d_val
        amr::K5Args k5{};
        k5.in = s.d_out; k5.out = s.d_val; k5.offs_pre = s.d_offs_pre; k5.offs_val = s.d_offs_val; k5.h_offs_val = s.h_offv;
        k5.keep = s.d_keep; k5.counts = s.d_counts; k5.listcnt = k3.listcnt; k5.listoff = s.d_listoff; k5.vgcnt = k3.vgcnt;
        k5.overflow = s.d_overflow; k5.cap = s.out_cap;
        k5.n_pre = n_pre; k5.n_tiles = s.n_tiles; k5.pkt_bytes = h->sg.pkt_bytes;
        k5.fold = k3.fold;
        hipLaunchKernelGGL(amr::k5_compact, dim3(s.n_tiles - k5.fold, n_pre), dim3(256), 0, st, k5);
        HIP_TRY(hipGetLastError());
        AMR_DBG(st, "k5_compact");
    }
    return AMR_OK;
}

amr_status enqueue_search(amr_handle *h, Slot &s, bool rerun = false, bool dense = false)
{
    AMR_TRY(enqueue_k2(h, s, h->stream, rerun, dense, false));
    return enqueue_tail(h, s, h->stream, false);
}

amr_status launch_ready_tails(amr_handle *h, bool last_too = false);
amr_status launch_tail(amr_handle *h, Slot &t);

__global__ void k_copy16(const uint4 *src, uint4 *dst, uint32_t n16)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// Enqueue one batch on the compute stream: K1, (search), history + carry update.  Returns at once.
//
// Wave quantisation (amr_set_deferral): K1 works in wave-tiles of 64 blocks, and a batch that does not end on one would
// end in a lone wave that takes as long as a whole chip-filling launch (every wave walks its BlockSize + SymbolLength
// samples in order, whatever the others do).  With `may_defer` the launch stops at the last whole wave-tile; the up to
// 63 blocks behind it are copied into the head buffer (by workgroups of the search launch) and become the first rows of
// the NEXT launch's wave-tile 0, completed with that batch's first blocks.  The stream position of a launch never
// depended on batch boundaries (the carry / history mechanism below), so nothing else changes: hits keep their call
// index, they just arrive with the following batch's result (or with amr_flush).
amr_status submit(amr_handle *h, const uint8_t *d_iq, size_t n_blocks, bool search, bool may_defer = false)
{
    HIP_TRY(hipSetDevice(h->device));
    const uint32_t n_head = h->n_head;                  // blocks deferred by the previous batch, waiting in the head buffer
    const size_t total = n_head + n_blocks;
    if (total == 0 || total > 0x7fffffffull) return fail(AMR_EINVAL, "n_blocks out of range");
    if (h->n_pending >= kMaxPending) return fail(AMR_EINVAL, "three batches already in flight: call amr_collect first");
    AMR_TRY(launch_ready_tails(h));
    const bool defer = may_defer && h->defer_on && search && h->r900_pid < 0 && total >= 64;
    const size_t rows = defer ? (total & ~(size_t)63) : total;   // rows (blocks) this launch processes
    const uint32_t new_head = (uint32_t)(total - rows);
    Slot &s = h->slot[h->next_slot];
    Slot &other = h->slot[(h->next_slot + 1) % kSlots];   // the slot the next batch will use: never one in flight
    Slot &prev = h->slot[(h->next_slot + kSlots - 1) % kSlots];   // the batch submitted before this one (if still in flight)
    AMR_TRY(ensure_capacity(h, s, other, rows));
    hipStream_t st = h->stream;
    const uint32_t bs = (uint32_t)h->geom.block_size;
    const size_t bs2 = (size_t)h->geom.block_size2;
    const uint32_t full = (uint32_t)(rows / 64), rem = (uint32_t)(rows % 64);

    s.ticket = h->next_ticket++;
    s.d_iq = d_iq;
    s.n_blocks = rows;
    s.n_tiles = (uint32_t)((rows + 63) / 64) + 1;
    s.search = search;
    s.calls_base = h->calls_done + h->block_base;
    s.iqhist_valid = h->iqhist_valid;
    s.iqhist_buf = h->iqhist_cur;
    s.force_rerun = false;
    if (search) {                                      // k3_stale.h: this batch's first hits continue the previous search batch's last
        s.carry_in_slot = h->carry_slot;
        if (s.carry_in_slot == h->next_slot) {         // (three batches without a search in between, amr_prime: the ring has come round)
            HIP_TRY(hipMemcpyAsync(h->d_pkt_carry + kSlots, h->d_pkt_carry + h->next_slot, 1, hipMemcpyDeviceToDevice, h->stream));
            s.carry_in_slot = kSlots;                  // never the byte this batch writes
        }
        h->carry_slot = h->next_slot;
    }

    // wave-tile 0 of a launch that starts with deferred blocks: completed in the head buffer with this batch's first blocks
    uint8_t *head_rows = h->d_head + h->halo_bytes;
    if (n_head && n_blocks) {
        const size_t c = std::min<size_t>(n_blocks, 64 - n_head);
        const uint32_t n16 = (uint32_t)(c * bs2 / 16);
        hipLaunchKernelGGL(k_copy16, dim3(std::min<uint32_t>(256, (n16 + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const uint4 *>(d_iq), reinterpret_cast<uint4 *>(head_rows + n_head * bs2), n16);
        HIP_TRY(hipGetLastError());
    }

    amr::K1Args k1{};
    k1.iq = d_iq - n_head * bs2;      // row r >= 64 of the launch is block r - n_head of the caller's batch
    k1.carry = n_head ? h->d_head : h->d_carry_cur;   // (deferred rows sit behind the carry in the head buffer: its writer left it current)
    k1.lut = h->d_lut;
    k1.qt = s.d_qt;
    k1.n_blocks = (uint32_t)rows;
    k1.block_size = bs;
    k1.zero_halo = h->zero_halo ? 1u : 0u;
    k1.head_rows = n_head ? 1u : 0u;

    s.timed = h->timing_level;
    hipEvent_t e0 = s.timed ? s.ev0 : nullptr, e1 = s.timed ? s.ev1 : nullptr;
    // A caller that keeps batches in flight gets K3 (K4, K5) of a batch on the second stream, next to the END of the
    // following batch's K1 and its search (see below, "the tail of the previous batch").
    const bool lazy = search && (h->lazy_tail || h->n_pending >= 1);
    if (lazy) h->lazy_tail = true;
    const bool all_coop = rows > 0 && rows <= h->k1_coop_max;   // see below
    const bool gate_prev = prev.pending && prev.search && prev.tail_split && !prev.tail_enqueued;
    s.dense = h->dense_hold > 0;
    if (s.dense) h->dense_hold--;
    // EARLY SEARCH: the search of this batch off the compute stream (on the tail stream), next to K1 -- the wave of a tile starts
    // as soon as the K1 waves that wrote it are done -- so that the next K1 launch follows this one directly.  For batches of
    // whole wave-tiles through the tile kernel, one preamble with a row kernel, nothing deferred; everything else keeps the
    // search between two K1 launches in stream order.
    // IN-WAVE SEARCH (k1_search.h): rows of 16 words (BlockSize 512), one of rtlamr's preambles whose taps all fit the row: the K1
    // wave searches its own tile before it stores it; the search launch shrinks to the history tile and the rows 63.
    bool inwave = false;
    if (search && h->inwave_mode && !all_coop && rem == 0 && full > 0 && h->r900_pid < 0 && !s.dense && !h->dense_search &&
        h->sg.n_pre == 1 && ((h->sg.wpb == 16 && h->geom.chip_length == 8) ||
                             (h->inwave_mode >= 2 && h->sg.wpb == 64 && (h->geom.chip_length == 32 || h->geom.chip_length == 40)))) {
        const int kind = amr::k2_walk_kind_of(h->sg.pre_len[0], h->sg.pre_bits[0]);
        amr::K2Args q{};                              // qt null: a question, not a launch
        q.g = h->sg; q.n_tiles = s.n_tiles;
        if (kind >= 0 && amr::launch_k2_cleanup(h->sg.symbol_length, (uint32_t)kind, nullptr, nullptr, nullptr, q)) {
            inwave = true;
            k1.srch.counts = s.d_counts; k1.srch.gcnt = s.d_gcnt; k1.srch.staging = s.d_staging; k1.srch.overflow = s.d_overflow;
            k1.srch.cap = s.stage_cap; k1.srch.n_tiles = s.n_tiles;
            k1.srch.n_lo = -(int64_t)h->geom.packet_length;
            k1.srch.n_hi = (int64_t)s.n_blocks * bs - (int64_t)h->geom.packet_length;
            k1.srch.kind_p1 = (uint32_t)kind + 1u;
        }
    }
    s.inwave = inwave;
    if (inwave) h->inwave_batches++;
    bool early = false;
    // Measured (profiles/r05/early_search_ab.txt): at BlockSize >= 2048 the next K1 launch, no longer held back by the search,
    // starts while the last searching waves (and the previous batch's K3) still hold wave slots; some of its waves start
    // late, the launch loses its lock-step and K1 takes 0.25-0.27 ms instead of 0.187: 20 % slower at chip 72, 1-6 % at chip
    // 32 .. 48.  At BlockSize 512 (chip 8) K1 is one launch of eight rounds, out of step anyway: 3-5 % faster.
    const bool early_here = h->early_mode > 0 || (h->early_mode < 0 && bs <= 512);
    if (lazy && early_here && !inwave && !all_coop && rem == 0 && full > 0 && n_head == 0 && new_head == 0 && h->r900_pid < 0 && !s.dense &&
        !h->dense_search && h->sg.n_pre == 1) {
        const int kind = amr::k2_walk_kind_of(h->sg.pre_len[0], h->sg.pre_bits[0]);
        amr::K2Args q{};                              // qt null: a question, not a launch
        q.g = h->sg; q.n_tiles = s.n_tiles;
        hipError_t qe = hipSuccess;
        early = kind >= 0 && amr::launch_k2_row(h->sg.symbol_length, (uint32_t)kind, 0u, 0, nullptr, nullptr, nullptr, q, &qe);
    }
    // the two orders must not overtake each other where they change hands: the state update (history rows, reset words)
    // rides in the search, on whichever stream that runs
    if (early && !h->k2_on_search) {
        HIP_TRY(hipEventRecord(h->ev_switch, st));
        HIP_TRY(hipStreamWaitEvent(h->search_stream, h->ev_switch, 0));
    } else if (!early && h->k2_on_search) {
        HIP_TRY(hipStreamWaitEvent(st, h->ev_last_early, 0));
    }
    h->k2_on_search = early;
    uint8_t *carry_next = h->d_head;                  // where this batch leaves the IQ halo of the next one
    if (early) {
        carry_next = h->d_carry_cur == h->d_head ? h->d_carry_alt : h->d_head;   // never the buffer wave-tile 0 is reading
        k1.done_flags = s.d_k1flags; k1.done_value = (uint32_t)s.ticket; k1.carry_out = carry_next;
    }
    amr::K1Args k1_last = k1;                     // the launch that announces itself to the gate: the batch's last one
    if (gate_prev || early) { k1_last.started = h->d_k1_started; k1_last.started_value = s.ticket; }
    // in-wave search (BlockSize 512: one launch of eight rounds): the previous batch's tail comes in at this launch's START and
    // shares its rounds; nothing is held back for it (below)
    if (inwave) k1_last.started_first = 1u;
    // One launch per "round" for long blocks: K1 holds 8 waves per CU, and a launch that exactly fills the chip keeps its
    // waves in step -- all of them read together and write their output bursts together.  A larger grid runs the later
    // rounds out of step (output stores trickle into the read stream all the time): BlockSize 4096, 4 GiB: 0.895 ms in
    // one launch, 4 x 0.179 ms in four; IDM (BlockSize 8192, 4 GiB) 0.860 -> 0.804 ms.  Short blocks (a round lasts
    // under 0.1 ms) lose more at the extra launch boundaries than they gain: BlockSize 2048 0.182 -> 0.256 ms, so they
    // keep the single launch.
    const uint32_t round = h->k1_round_tiles ? h->k1_round_tiles : bs >= 4096 ? (uint32_t)h->n_cus * 8u : full;
    // Small batches entirely as one wave per block (k1_coop.h): a wave-tile costs a whole wave life (150-175 us) however few
    // tiles there are; a wave per block finishes in ~50 us as long as the waves fit the chip side by side (all_coop).
    // What the previous batch's gate kernel waits for before it comes onto the chip (see below): the end of the K1 round in
    // front of this batch's LAST K1 launch, as the stop event of that dispatch (its completion signal: an event recorded
    // between the two launches is a packet of its own and costs the compute stream 5 us).  Batches of one launch: nothing --
    // the gate is gone 6 us after the launch's last workgroup has started, a workgroup it displaced starts 7 us late, and
    // the search's end as a stop event costs more than that (1 us of the search, 2 us of the K1 behind it:
    // profiles/r05/k1_gate_fragmentation.txt).
    const bool gate_ev = gate_prev && h->gate_event;
    hipEvent_t gate_wait = nullptr;
    const uint32_t n_launches = all_coop ? 1u : (full ? (full + round - 1) / round : 0u) + (rem ? 1u : 0u);
    // A batch whose ONE K1 launch runs several rounds (BlockSize <= 2048: 4096 .. 16384 wave-tiles on 2048 slots): its last
    // workgroup starts with the last round, so a gate kernel would sit on the chip through all the rounds before it -- a
    // one-wave kernel somewhere in a SIMD's register file while the dispatcher places K1 workgroups round after round (the
    // fragmentation above, every round) -- and K3 would come in next to the last round's waves, which at these chip lengths
    // leave it registers.  Measured at chip 40 (profiles/r06/bs2048/): K1 191 .. 254 us in the pipeline for 178 alone.  Here the
    // tail waits for the END of that launch instead (its stop event, no gate kernel): K3.. then run next to this batch's search.
    const bool gate_end = gate_prev && !all_coop && n_launches == 1 && !rem &&
                          (h->gate_end_mode > 0 || (h->gate_end_mode < 0 && full > (uint32_t)h->n_cus * 8u));
    // the announcing launch's share of the counter its last workgroups meet at (K1Args::started_ctr)
    auto arm = [&](uint32_t grid) {
        if (!k1_last.started) return;
        h->k1_ctr_total += std::min(8u, grid);
        k1_last.started_ctr = h->d_k1_ctr;
        k1_last.started_target = h->k1_ctr_total;
    };
    if (all_coop) {
        arm((uint32_t)rows);
        amr::launch_k1_coop(h->geom.chip_length, 0u, (uint32_t)rows, st, k1_last, e0, e1);
    } else {
        uint32_t li = 0;
        for (uint32_t w0 = 0; w0 < full; w0 += round, ++li) {
            const uint32_t n = std::min(round, full - w0);
            const bool last = w0 + n == full && !rem;
            if (last) arm(n);
            amr::K1Args &kk = last ? k1_last : k1;
            kk.wg_first = w0;
            hipEvent_t stop = last ? e1 : nullptr;
            if (li + 2 == n_launches && gate_ev) { stop = s.ev_gate; gate_wait = s.ev_gate; }   // the launch in front of the announcing one
            if (gate_end) { if (!stop) stop = s.ev_gate; gate_wait = stop; }                    // (one launch: this one)
            amr::launch_k1(h->geom.chip_length, dim3(n), st, kk, w0 == 0 ? e0 : nullptr, stop);
        }
        if (rem) arm(rem);
        if (rem)     // the blocks behind the last whole wave-tile (sync callers, flush): a wave each
            amr::launch_k1_coop(h->geom.chip_length, full * 64u, rem, st, k1_last, full ? nullptr : e0, e1);
    }
    HIP_TRY(hipGetLastError());
    AMR_DBG(st, "k1");
    // The tail of the previous batch (its K3, K4, K5 and the kernel that publishes its ticket), enqueued NOW on the second
    // stream behind a gate that opens when this batch's K1 has every wave on the chip.  K1 holds all LDS and all but 16
    // registers per SIMD, so the tail's workgroups get on the chip only where K1 waves retire: they fill the ragged end
    // of the K1 launch and the start of the search, and nobody waits for the host to notice anything (round 3 launched
    // the tail from the host when it saw the search start: 13 us later, and the state update of that search -- the last
    // thing in front of the next K1 -- waited for K3 to finish: K1-to-K1 232 us for K1 185 + K2 25).
    // (Measured on one box, profiles/r04/k2_tail_ab/: host-launched tail 0.262-0.265 ms per step, gated 0.249-0.256, everything
    // behind K2 on the compute stream 0.289; the gate's extra delay -- 0, 6 or 20 us --, whether it is enqueued before
    // or behind K2, and stream priorities make no difference that survives the run-to-run noise.)
    if (gate_prev) {
        // The gate's one wave must not come onto the chip while waves of OTHER kernels are there.  It needs 8 registers and K1
        // leaves 16 per SIMD free -- but a SIMD's registers are handed out as contiguous ranges, and a gate that started
        // next to search or K3 waves (whenever the tail stream got to it: usually during this slot's previous search) may sit
        // in the MIDDLE of its SIMD's file, where no two ranges of 248 fit around it.  One K1 workgroup then finds no slot
        // and, the dispatcher placing workgroups in order, holds up the ones behind it on its XCD until the first K1 wave of
        // that shader engine retires: 4 of 2048 workgroups start 330 us late and run alone at the end (IDM geometry with
        // the two-lanes-per-row search: first K1 round 530 us instead of 400; profiles/r05/k1_gate_fragmentation.txt).  So the
        // gate waits, as an event in front of it, for the end of what precedes the K1 launch it is about (gate_wait above):
        // by then the chip holds nothing but K1 waves (ranges at 0, 248 and 496), or nothing.
        if (gate_wait) HIP_TRY(hipStreamWaitEvent(h->tail_stream, gate_wait, 0));
        // (Round 6 tried hipStreamWaitValue64 on a signal word in its place, hoping for a wait the queue's packet processor does
        // with nothing on the chip: this runtime implements it as a polling KERNEL of its own, __amd_rocclr_streamOpsWait --
        // the same one wave, without this gate's timeout; cfg3 0.87 -> 0.99 ms per step without the event above.
        // tools/waitvalue_probe.hip, profiles/r06/gate/.)
        if (!gate_end) {
            hipLaunchKernelGGL(amr::k_gate, dim3(1), dim3(1), 0, h->tail_stream, h->d_k1_started, s.ticket, h->gate_delay_ticks,
                               h->gate_timeout_ticks, prev.d_overflow);
            HIP_TRY(hipGetLastError());
        }
        // (an early search ran on the search stream -- today the tail stream itself, then this wait is a no-op)
        if (prev.early) HIP_TRY(hipStreamWaitEvent(h->tail_stream, prev.ev_k2done, 0));
        AMR_TRY(launch_tail(h, prev));
        prev.tail_gated = true;
    }
    // state carried to the next batch (decode.go:165-166): the last rows of this slot's bitstream become the history
    // tile of the NEXT slot, the last HBA bytes of IQ (and the deferred blocks behind them) go to the head buffer, the
    // next slot's search words are reset.  Whatever does it is also the last thing in front of the next K1 launch, which
    // must not meet the previous batch's K3.. (it needs every wave slot): it waits for them on a device word.
    const uint8_t *launch_end = rows > n_head ? d_iq + (rows - n_head) * bs2 : head_rows + rows * bs2;
    amr::HistArgs ha{s.d_qt, other.d_qt, (uint32_t)rows, h->hist_rows, h->sg.wpb, h->sg.lg_wpb,
                     launch_end - h->halo_bytes, h->d_head, early ? 0u : h->halo_bytes,     // (early: K1's last wave-tile left the halo)
                     (uint32_t)(new_head * bs2), new_head ? 16u : 0u, other.d_overflow,
                     other.d_gcnt, other.gcnt_words,
                     lazy ? nullptr : s.h_done, s.ticket, &h->h_flags[1],
                     // (early: the next K1 launch no longer waits for this kernel, so there is nothing to hold back)
                     (!early && !inwave && prev.pending && prev.search && prev.tail_split) ? h->d_tail_done : nullptr, prev.ticket,
                     early ? s.d_k1flags + (full - 1) : nullptr, (uint32_t)s.ticket};
    // pipelined callers: the update rides along with the search as more workgroups of its launch instead of following it
    // as a 5 us kernel
    bool folded = false;
    if (search) {
        if (early) {
            // behind a gate of its own: the searching waves wait for K1 waves, so they must not get on the chip before every
            // one of those has its slot (they would be holding what K1 is waiting for)
            hipLaunchKernelGGL(amr::k_gate, dim3(1), dim3(1), 0, h->search_stream, h->d_k1_started, s.ticket, 0u, h->gate_timeout_ticks, s.d_overflow);
            HIP_TRY(hipGetLastError());
            AMR_TRY(enqueue_k2(h, s, h->search_stream, false, false, true, &ha, &folded, true));
            HIP_TRY(hipEventRecord(s.ev_k2done, h->search_stream));
            h->ev_last_early = s.ev_k2done;
        } else if (lazy) AMR_TRY(enqueue_k2(h, s, st, false, s.dense, true, &ha, &folded));
        else AMR_TRY(enqueue_search(h, s, false, s.dense));
    }
    s.early = early;
    h->d_carry_cur = carry_next;
    s.single = false;
    s.tail_enqueued = !lazy;
    s.tail_split = lazy;
    s.tail_gated = false;
    s.folded = folded;

    if (h->r900_pid >= 0) {   // the PL samples that precede the next batch (r900.go:168-170 keeps them as magnitudes)
        const uint64_t n_batch = (uint64_t)n_blocks * bs;
        const int nxt = (h->iqhist_cur + 1) % kIqHist;
        amr::IqHistArgs ih{d_iq, h->d_iqhist[h->iqhist_cur], h->d_iqhist[nxt], n_batch, (uint32_t)h->geom.packet_length};
        hipLaunchKernelGGL(amr::k_iqhist_update, dim3(32), dim3(256), 0, st, ih);
        HIP_TRY(hipGetLastError());
        h->iqhist_cur = nxt;
        const uint64_t v = (uint64_t)h->iqhist_valid + n_batch;
        h->iqhist_valid = (uint32_t)std::min<uint64_t>(v, (uint64_t)h->geom.packet_length);
    }
    if (!folded) {
        // The copies of the deferred blocks run AHEAD of the kernel that publishes the batch ticket: amr_collect may return
        // as soon as the ticket is there, and the caller may then overwrite the buffer the copies read
        // (include/amrdemod.h: "the caller's buffer is free after the collect").
        if (ha.defer_bytes) {
            const uint32_t n16 = ha.defer_bytes / 16;
            hipLaunchKernelGGL(k_copy16, dim3(std::min<uint32_t>(256, (n16 + 255) / 256)), dim3(256), 0, st,
                               reinterpret_cast<const uint4 *>(ha.carry_src + ha.carry_bytes),
                               reinterpret_cast<uint4 *>(ha.carry_dst + ha.carry_bytes), n16);
            HIP_TRY(hipGetLastError());
            ha.defer_bytes = 0;
            ha.defer_wgs = 0;
        }
        hipLaunchKernelGGL(amr::k_hist_update, dim3(1), dim3(1024), (size_t)h->hist_rows * h->sg.wpb * 4, st, ha);
        HIP_TRY(hipGetLastError());
        AMR_DBG(st, "k_hist_update");
    }
    h->zero_halo = false;
    h->n_head = new_head;
    if (search) h->calls_done += rows;
    s.pending = true;
    h->n_pending++;
    h->next_slot = (h->next_slot + 1) % kSlots;
    return AMR_OK;
}

// One block, nothing in flight: the whole Decode call as ONE launch (k1_single.h) -- demodulation, search, slice, state
// update and ticket by one workgroup, the result written straight into the slot's pinned host mirror.  `iq`: the block
// in device memory, or in pinned host memory (the kernel reads it over the link: 8 to 16 KiB, one round trip).
// Bookkeeping as submit(): the batch is left "in flight" for collect(), which finds nothing to copy.
bool single_block_ok(const amr_handle *h, size_t n_blocks, const void *iq)
{
    return n_blocks == 1 && !h->no_single && h->n_pending == 0 && h->n_head == 0 && !h->validate && h->r900_pid < 0 &&
           !h->dense_search && h->dense_hold == 0 && (reinterpret_cast<uintptr_t>(iq) & 15u) == 0 &&
           (h->sg.packet_symbols & 7u) == 0 &&        // (a last byte of fewer than 8 symbols carries bits of earlier hits: k3_stale.h)
           amr::k_single_lds_bytes(h->sg) <= 160 * 1024 - 512;
}

amr_status submit_single(amr_handle *h, const uint8_t *iq)
{
    HIP_TRY(hipSetDevice(h->device));
    Slot &s = h->slot[h->next_slot];
    Slot &other = h->slot[(h->next_slot + 1) % kSlots];
    AMR_TRY(ensure_capacity(h, s, other, 1));
    const size_t rec = 12 + h->sg.pkt_bytes;
    if (s.host_cap < s.out_cap) {             // the kernel writes the pinned mirror itself: it must hold what the device buffer holds
        AMR_TRY(host_realloc(s.h_out, s.out_cap * rec));
        s.host_cap = s.out_cap;
    }
    const size_t lds = amr::k_single_lds_bytes(h->sg);
    if (!h->single_ready) {
        HIP_TRY(hipFuncSetAttribute((const void *)amr::k_single_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        h->single_ready = true;
    }
    s.ticket = h->next_ticket++;
    s.d_iq = iq;
    s.n_blocks = 1;
    s.n_tiles = 2;
    s.search = true;
    s.calls_base = h->calls_done + h->block_base;
    s.iqhist_valid = h->iqhist_valid;
    s.iqhist_buf = h->iqhist_cur;
    s.timed = h->timing_level;
    s.dense = false;
    amr::SingleArgs a{};
    a.iq = iq; a.carry = h->d_carry_cur; a.carry_out = h->d_head; a.lut = h->d_lut;
    h->d_carry_cur = h->d_head;
    s.early = false;
    a.qt = s.d_qt; a.qt_next = other.d_qt; a.ovf_next = other.d_overflow;
    a.gcnt_next = other.d_gcnt; a.gcnt_words = other.gcnt_words;
    a.out = s.d_out; a.h_out = s.h_out; a.cap = s.out_cap;
    a.offs_pre = s.d_offs_pre; a.h_offs_pre = s.h_off; a.h_overflow = s.h_ovf;
    a.block_base = s.calls_base;
    a.done_flag = s.h_done; a.adone_flag = &h->h_flags[1]; a.done_value = s.ticket;
    a.chip_length = (uint32_t)h->geom.chip_length; a.halo_bytes = h->halo_bytes; a.hist_rows = h->hist_rows;
    a.zero_halo = h->zero_halo ? 1u : 0u;
    a.g = h->sg;
    a.dbg = h->d_single_dbg;
    hipExtLaunchKernelGGL(amr::k_single_block, dim3(1), dim3(amr::kSingleThreads), lds, h->stream,
                          s.timed ? s.ev0 : nullptr, s.timed ? s.ev1 : nullptr, 0, a);
    HIP_TRY(hipGetLastError());
    AMR_DBG(h->stream, "k_single_block");
    s.tail_enqueued = true; s.tail_split = false; s.tail_gated = false; s.folded = false;
    s.single = true;
    s.force_rerun = false;
    h->zero_halo = false;
    h->calls_done += 1;
    s.pending = true;
    h->n_pending++;
    h->next_slot = (h->next_slot + 1) % kSlots;
    return AMR_OK;
}

// Completion of a batch = its last kernel stored the batch ticket into pinned host memory.  No event on the
// stream (each costs a ~5 us bubble); the stream is polled now and then so that a device fault ends the wait.
static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}

// Spin until the pinned word `flag` reaches `value`; `st` is the stream whose completion guarantees it.
amr_status wait_flag(const uint64_t *flag, uint64_t value, hipStream_t st)
{
    // three stages: a short busy spin (a batch in steady state completes within tens of microseconds of the call),
    // then spinning with sched_yield so that parser threads and the other ranks' hosts get the core, and after ~2 ms a
    // blocking hipStreamSynchronize (which also surfaces a device fault).  No hipStreamQuery in between: on a stream
    // that is still busy it makes the runtime put a marker packet behind the kernels already enqueued, and the next
    // batch's first kernel then starts 5-9 us after this batch's last one instead of at once (round 4 kernel traces).
    for (uint64_t spin = 0;; ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) >= value) return AMR_OK;
        if (spin < 4096) { cpu_relax(); continue; }
        if (spin > 4096 + 20000) {
            HIP_TRY(hipStreamSynchronize(st));
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) >= value) return AMR_OK;
            return fail(AMR_EHIP, "batch finished without publishing its ticket");
        }
        sched_yield();
    }
}

amr_status wait_done(amr_handle *h, Slot &s)
{
    return wait_flag(s.h_done, s.ticket, s.tail_split ? h->tail_stream : h->stream);
}

// Has the compute-stream part (K1, search, state update) of the k-th batch in flight finished -- in particular its search,
// whose output K3 reads?  The signals, all without an event on the stream: the NEXT batch's search has announced its
// start (pinned word 0; the stream is in order), or the batch's own state-update kernel has published its ticket (pinned
// word 1), or -- when that update rode along inside the search kernel and the batch is the youngest -- the stream is idle.
amr_status search_finished(amr_handle *h, int k, bool wait, bool *yes)
{
    auto pending = [&](int i) -> Slot & { return h->slot[(h->next_slot - h->n_pending + i + 2 * kSlots) % kSlots]; };
    const Slot &t = pending(k);
    const uint64_t *flag = nullptr;
    uint64_t value = 0;
    if (k + 1 < h->n_pending) {
        const Slot &nx = pending(k + 1);
        flag = nx.search ? &h->h_flags[0] : &h->h_flags[1];    // a batch without a search always has the kernel
        value = nx.ticket;
    } else if (!t.folded) {
        flag = &h->h_flags[1];
        value = t.ticket;
    }
    if (flag) {
        if (wait) AMR_TRY(wait_flag(flag, value, h->stream));
        *yes = __atomic_load_n(flag, __ATOMIC_ACQUIRE) >= value;
        return AMR_OK;
    }
    hipStream_t sst = t.early ? h->search_stream : h->stream;       // where the batch's search ran
    if (wait) { HIP_TRY(hipStreamSynchronize(sst)); *yes = true; return AMR_OK; }
    const hipError_t e = hipStreamQuery(sst);
    if (e != hipSuccess && e != hipErrorNotReady) return fail(AMR_EHIP, "hipStreamQuery", e);
    *yes = e == hipSuccess;
    return AMR_OK;
}

amr_status launch_tail(amr_handle *h, Slot &t)
{
    AMR_TRY(enqueue_tail(h, t, h->tail_stream, true));
    hipLaunchKernelGGL(amr::k_done, dim3(1), dim3(1), 0, h->tail_stream, t.h_done, t.ticket, h->d_tail_done);
    HIP_TRY(hipGetLastError());
    t.tail_enqueued = true;
    return AMR_OK;
}

// Launch, without waiting for anything, the second-stream part (K3..) of every batch in flight whose successor's search
// has started (= the successor's K1 has finished), oldest first.  Called wherever the host passes by: submit, collect
// and the wait for the read-back, so that a host that is busy copying results does not hold the GPU up.
// last_too: also the youngest batch's, once its own search has finished (the caller is waiting for a read-back and
// submits nothing meanwhile; otherwise it waits for the K1 of a successor that may be on its way).
amr_status launch_ready_tails(amr_handle *h, bool last_too)
{
    for (int k = 0; k < h->n_pending; ++k) {
        Slot &t = h->slot[(h->next_slot - h->n_pending + k + 2 * kSlots) % kSlots];
        if (!t.search || t.tail_enqueued) continue;
        if (k + 1 == h->n_pending && !last_too) break;
        bool ready = false;
        AMR_TRY(search_finished(h, k, false, &ready));
        if (!ready) break;                       // in order: the tickets on the second stream rise
        AMR_TRY(launch_tail(h, t));
    }
    return AMR_OK;
}

}  // namespace

amr_status amr_host::sync_compute(amr_handle *h)
{
    for (int k = 0; k < h->n_pending; ++k) {
        Slot &t = h->slot[(h->next_slot - h->n_pending + k + 2 * kSlots) % kSlots];
        if (!t.search || t.tail_enqueued) continue;
        bool ready = false;
        AMR_TRY(search_finished(h, k, true, &ready));
        AMR_TRY(launch_tail(h, t));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipStreamSynchronize(h->search_stream));
    HIP_TRY(hipStreamSynchronize(h->tail_stream));
    return AMR_OK;
}

namespace {

// Wait for the oldest batch in flight, grow capacities / re-run the search if it overflowed, read back hits.
amr_status collect(amr_handle *h, amr_result *res)
{
    HIP_TRY(hipSetDevice(h->device));
    if (h->n_pending == 0) return fail(AMR_EINVAL, "amr_collect: nothing in flight");
    const int si = (h->next_slot - h->n_pending + kSlots) % kSlots;
    Slot &s = h->slot[si];
    const uint32_t n_pre = h->sg.n_pre;
    AMR_TRY(launch_ready_tails(h));
    if (s.search && !s.tail_enqueued) {
        // K3 (K4, K5) of this batch, on the second stream.  They need the batch's K2 to have finished; they are held
        // back until the NEXT batch's K1 has finished as well (its search announces itself): next to a K1 launch,
        // which fills every wave slot of the chip, they would only delay some of its waves.
        bool ready = false;
        AMR_TRY(search_finished(h, 0, true, &ready));
        AMR_TRY(launch_tail(h, s));
    }
    AMR_TRY(wait_done(h, s));
    if (h->n_pending == 1) h->lazy_tail = false;   // nothing else in flight: the caller is not pipelining (any more)
    uint64_t total = 0, searched = 0;
    bool use_dense = s.dense;
    bool stale_redone = false;    // the result on the device is newer than what a one-launch batch wrote to its pinned mirror
    if (s.search) {
        bool searched_again = false;
        for (int attempt = 0;; ++attempt) {
            const uint32_t ovf = *s.h_ovf;
            total = s.h_off[n_pre];
            if (attempt > 8) return fail(AMR_EOVERFLOW, "hit capacity could not be grown");
            bool rerun = false;
            // sparse hit list overflowed (e.g. the zero history of a fresh stream matches r900's 16 leading zeros):
            // this batch is searched again with the dense kernel; the next one starts sparse again unless
            // overflows keep coming
            if (ovf & 2u) { use_dense = true; rerun = true; }
            // the gate in front of this batch's K3 gave up waiting for the following K1 (k_gate): K3.. did not touch the
            // result; searched again here, on the compute stream, in order behind the batch's own K2
            if (ovf & amr::kOvfGate) { h->gate_timeouts++; rerun = true; }
            if (ovf & 1u) {   // a tile found more hits than its staging slot holds
                s.stage_cap *= 8;
                const uint64_t lim = (uint64_t)64 * h->geom.block_size;
                if (s.stage_cap > lim) s.stage_cap = (uint32_t)lim;
                AMR_TRY(sync_compute(h));
                AMR_TRY(dev_realloc(s.d_staging, s.staging_tiles * n_pre * (size_t)s.stage_cap));
                rerun = true;
            } else if (!rerun && total > s.out_cap) {
                uint64_t nc = s.out_cap;
                while (nc < total) nc *= 2;
                s.out_cap = nc;
                AMR_TRY(sync_compute(h));
                AMR_TRY(alloc_hit_buffers(h, s));
                rerun = true;
            }
            if (!rerun) {
                // the hist kernel of the batch that followed zeroed this slot's group sums before the re-run added
                // to them again: leave them zero for the slot's next batch
                if (attempt) HIP_TRY(hipMemsetAsync(s.d_gcnt, 0, (size_t)s.gcnt_words * 4, h->stream));
                break;
            }
            // The slot's bitstream, its history rows included, is intact until the slot is reused (four slots, three batches
            // in flight: the state update that overwrites this slot's history tile belongs to a batch that cannot be
            // submitted before this one has been collected), so the search can simply run again.
            AMR_TRY(enqueue_search(h, s, true, use_dense));
            AMR_TRY(sync_compute(h));
            searched_again = true;
        }
        // k3_stale.h across batches in flight.  A batch that was searched again wrote its carry byte (the last byte of its
        // last hit) only now; a younger batch whose tail had ALREADY been enqueued read the byte before that.  Such a batch
        // gets k_stale_bits alone once more when it is collected (the pass is idempotent: it only reads the fresh low bits),
        // on the compute stream behind the re-search -- not a second search -- and hands the duty on only if its own carry
        // byte really changed.  Batches whose tail is not enqueued yet read the corrected byte anyway.
        bool carry_changed = searched_again;
        if (s.force_rerun && !searched_again) {
            uint8_t before = 0, after = 0;
            uint8_t *cb = h->d_pkt_carry + (&s - h->slot);
            AMR_TRY(sync_compute(h));
            HIP_TRY(hipMemcpy(&before, cb, 1, hipMemcpyDeviceToHost));
            AMR_TRY(enqueue_stale(h, s, h->stream));
            AMR_TRY(sync_compute(h));
            HIP_TRY(hipMemcpy(&after, cb, 1, hipMemcpyDeviceToHost));
            carry_changed = before != after;
            stale_redone = true;
            h->stale_reruns++;
        }
        if (searched_again) h->researches++;
        s.force_rerun = false;
        if (carry_changed && (h->sg.packet_symbols & 7u))
            for (Slot &o : h->slot)
                if (&o != &s && o.pending && o.search && o.tail_enqueued) o.force_rerun = true;
        if (use_dense && !s.dense) {
            if (++h->dense_streak >= 4) { h->dense_hold = 32; h->dense_streak = 0; }
        } else if (!use_dense) {
            h->dense_streak = 0;
        }
        searched = total;
        if (h->validate) total = s.h_offv[n_pre];   // what is read back is the validated list
        if (total > s.host_cap) {
            uint64_t nc = s.host_cap ? s.host_cap : (1 << 16);
            while (nc < total) nc *= 2;
            AMR_TRY(host_realloc(s.h_out, nc * (12 + h->sg.pkt_bytes)));
            s.host_cap = nc;
        }
        // (the one-launch path for a single block wrote the pinned mirror itself, k1_single.h)
        if (total && !(s.single && !searched_again && !stale_redone)) {   // on the copy stream: overlaps the next batch's kernels
            HIP_TRY(hipMemcpyAsync(s.h_out, h->validate ? s.d_val : s.d_out, total * (12 + h->sg.pkt_bytes),
                                   hipMemcpyDeviceToHost, h->copy_stream));
            if (h->r900_pid >= 0) {
                const uint64_t nr = s.h_off[h->r900_pid + 1] - s.h_off[h->r900_pid];
                if (nr > s.r900_host_cap) {
                    uint64_t nc = s.r900_host_cap ? s.r900_host_cap : 1024;
                    while (nc < nr) nc *= 2;
                    AMR_TRY(host_realloc(s.h_r900, nc * amr::kR900Digits));
                    s.r900_host_cap = nc;
                }
                if (nr) HIP_TRY(hipMemcpyAsync(s.h_r900, s.d_r900, nr * amr::kR900Digits, hipMemcpyDeviceToHost, h->copy_stream));
            }
            // the read-back takes as long as a K1 launch: keep an eye on the batches behind this one meanwhile.  Only
            // through the pinned flags (launch_ready_tails without last_too): asking the runtime about the COMPUTE stream
            // (hipStreamQuery) puts a marker packet behind the youngest batch's search, right in front of the next K1.
            // A caller that is draining its pipeline (this batch and at most one more in flight: the end of a run, or a
            // caller that never keeps three in flight) submits nothing while it waits here: then the youngest batch's tail
            // is launched as soon as its search has finished, next to this read-back, instead of behind it -- asked of the
            // runtime now and then only, because that is the query that costs a marker packet.
            const bool draining = h->n_pending <= 2;
            for (uint32_t spin = 0;; ++spin) {
                const hipError_t qe = hipStreamQuery(h->copy_stream);
                if (qe == hipSuccess) break;
                if (qe != hipErrorNotReady) return fail(AMR_EHIP, "hipStreamQuery(copy stream)", qe);
                AMR_TRY(launch_ready_tails(h, draining && (spin & 63u) == 63u));
                cpu_relax();
            }
        }
    }
    float a = 0, b = 0, c = 0;
    h->timing_valid = false;
    if (s.timed && s.single && hipEventSynchronize(s.ev1) == hipSuccess && hipEventElapsedTime(&a, s.ev0, s.ev1) == hipSuccess) {
        h->timing = amr_timing{a, 0.f, a};     // one launch: demodulation, search and slice are not separable
        h->timing_valid = true;
    } else if (s.timed && hipEventSynchronize(s.ev1) == hipSuccess && hipEventElapsedTime(&a, s.ev0, s.ev1) == hipSuccess) {
        if (s.timed >= 2 && s.search && s.early && hipEventSynchronize(s.ev_k2) == hipSuccess && hipEventElapsedTime(&b, s.ev1, s.ev_k2) == hipSuccess) {
            // early search: K2 ran next to K1 from its first retiring wave on; what it cost is what stuck out behind K1's end
            b = b > 0.f ? b : 0.f;
            h->timing = amr_timing{a, b, a + b};
        } else if (s.timed >= 2 && s.search && s.tail_split && hipEventSynchronize(s.ev_k2) == hipSuccess &&
            hipEventElapsedTime(&b, s.ev_s, s.ev_k2) == hipSuccess)
            // K2 and the tail ran apart (pipelined callers): what the batch cost the compute stream besides K1 is its K2.
            // The tail (K3, K4, K5) runs on the second stream next to the following batch's kernels -- let in behind a gate
            // at that K1's start, its workgroups trickle in where K1 waves retire and its dispatch spans that whole K1;
            // launched by the host, it runs next to that batch's search -- and has no duration that could be added to
            // the step: the same three numbers in either case (include/amrdemod.h, amr_timing)
            h->timing = amr_timing{a, b, a + b};
        else if (s.timed >= 2 && s.search && !s.tail_split && hipEventSynchronize(s.ev2) == hipSuccess &&
            hipEventElapsedTime(&b, s.ev_s, s.ev2) == hipSuccess && hipEventElapsedTime(&c, s.ev0, s.ev2) == hipSuccess)
            h->timing = amr_timing{a, b, c};
        else
            h->timing = amr_timing{a, 0.f, a};
        h->timing_valid = true;
    }
    s.pending = false;
    h->n_pending--;
    if (s.search) {
        h->last_slot = si;
        h->last_empty = false;
        h->last_n_blocks = s.n_blocks;
        const uint64_t *offs = h->validate ? s.h_offv : s.h_off;
        h->r_off.assign(offs, offs + n_pre + 1);
        h->last_total = total;
        h->last_searched = searched;
        if (res) {
            res->n_preambles = n_pre;
            res->pkt_bytes = h->sg.pkt_bytes;
            res->n_hits = total;
            res->preamble_offset = h->r_off.data();
            res->hit_block = reinterpret_cast<const uint64_t *>(s.h_out);
            res->hit_idx = reinterpret_cast<const uint32_t *>(s.h_out + total * 8);
            res->pkt = s.h_out + total * 12;
            res->r900_preamble = h->r900_pid;
            res->r900_digits = h->r900_pid >= 0 ? s.h_r900 : nullptr;
            res->n_hits_searched = searched;
            res->first_block = s.calls_base;
            res->n_blocks = s.n_blocks;
        }
    }
    return AMR_OK;
}

amr_status stage_host_input(amr_handle *h, const uint8_t *iq, size_t bytes)
{
    if (bytes > h->iq_cap) {
        AMR_TRY(sync_compute(h));
        AMR_TRY(dev_realloc(h->d_iq, bytes));
        h->iq_cap = bytes;
    }
    HIP_TRY(hipMemcpyAsync(h->d_iq, iq, bytes, hipMemcpyHostToDevice, h->stream));
    return AMR_OK;
}

}  // namespace

amr_status amr_host::drain(amr_handle *h)
{
    while (h->n_pending) AMR_TRY(collect(h, nullptr));
    return AMR_OK;
}

#if AMR_K1T_CLK
namespace amr { void k1t_dump_timeline(const char *path); }   // k1_launch.inc, diagnostic builds
#endif

void amr_host::dump_diagnostics(amr_handle *h)
{
    (void)h;
#if AMR_K1T_CLK
    if (const char *fn = getenv("AMR_K1_TIMELINE")) amr::k1t_dump_timeline(fn);
#endif
#if AMR_K3_DBG
    {   // diagnostic build: phases of the last K3 launch's workgroups
        (void)hipDeviceSynchronize();
        static unsigned long long hc[4096 * 8];
        if (hipMemcpyFromSymbol(hc, HIP_SYMBOL(amr::k3_dbg), sizeof hc) == hipSuccess) {
            if (const char *fn = getenv("AMR_K3_DBG_FILE")) { if (FILE *f = fopen(fn, "wb")) { fwrite(hc, 1, sizeof hc, f); fclose(f); } }
            unsigned long long t0 = ~0ull, t1 = 0; int n = 0;
            for (int i = 1; i < 4096; ++i) if (hc[8 * i]) { t0 = std::min(t0, hc[8 * i]); for (int k = 0; k < 7; ++k) t1 = std::max(t1, hc[8 * i + k]); ++n; }
            double ph[7] = {}, mx[7] = {}, st_mx = 0, st_sum = 0;
            for (int i = 1; i < 4096; ++i) if (hc[8 * i]) {
                st_sum += (double)(hc[8 * i] - t0); st_mx = std::max(st_mx, (double)(hc[8 * i] - t0));
                for (int k = 1; k < 7; ++k) if (hc[8 * i + k] >= hc[8 * i + k - 1]) { const double d = (double)(hc[8 * i + k] - hc[8 * i + k - 1]); ph[k] += d; mx[k] = std::max(mx[k], d); }
            }
            if (n) {
                fprintf(stderr, "AMR_K3_DBG: %d workgroups, span %.2f us, start mean %.2f max %.2f us;", n, (double)(t1 - t0) * 0.01, st_sum / n * 0.01, st_mx * 0.01);
                const char *nm[7] = {"", "prologue", "slice", "barrier", "tables+edge", "rounds", "reduce"};
                for (int k = 1; k < 7; ++k) fprintf(stderr, " %s %.2f/%.2f", nm[k], ph[k] / n * 0.01, mx[k] * 0.01);
                fprintf(stderr, " (mean/max us)\n");
                for (int rep = 0; rep < 8; ++rep) {      // the slowest workgroups
                    int best = -1; unsigned long long bt = 0;
                    for (int i = 1; i < 4096; ++i) if (hc[8 * i]) { unsigned long long e = 0; for (int k = 0; k < 7; ++k) e = std::max(e, hc[8 * i + k]); if (e - hc[8 * i] > bt) { bt = e - hc[8 * i]; best = i; } }
                    if (best < 0) break;
                    fprintf(stderr, "  wg %4d hits %4llu start %.2f:", best, hc[8 * best + 7], (double)(hc[8 * best] - t0) * 0.01);
                    for (int k = 1; k < 7; ++k) fprintf(stderr, " %.2f", hc[8 * best + k] >= hc[8 * best + k - 1] ? (double)(hc[8 * best + k] - hc[8 * best + k - 1]) * 0.01 : -1.0);
                    fprintf(stderr, "\n");
                    hc[8 * best] = 0;
                }
            }
        }
    }
#endif
#if AMR_GATE_CLK
    {   // diagnostic build: shader clock seen by the gate kernels (they sleep through the first rounds of the following K1)
        (void)hipDeviceSynchronize();
        static unsigned long long hc[4096];
        if (hipMemcpyFromSymbol(hc, HIP_SYMBOL(amr::k_gate_clk), sizeof hc) == hipSuccess) {
            double cyc = 0, tick = 0; int n = 0;
            for (int i = 0; i < 2048; ++i) if (hc[2 * i + 1] > 1000) { cyc += (double)hc[2 * i]; tick += (double)hc[2 * i + 1]; ++n; }
            if (n) fprintf(stderr, "AMR_GATE_CLK: %d gates, mean wait %.1f us, shader clock while waiting %.3f GHz\n", n, tick / n * 0.01, cyc / tick * 0.1);
        }
    }
#endif
}

extern "C" {

amr_status amr_decode_batch(amr_handle *h, const uint8_t *iq, size_t iq_bytes, size_t n_blocks, amr_result *res)
{
    if (!h || !iq) return fail(AMR_EINVAL, "null argument");
    const size_t need = n_blocks * (size_t)h->geom.block_size2;
    if (iq_bytes < need) return fail(AMR_EINVAL, "short input (the Go decoder panics here, decode.go:222)");
    HIP_TRY(hipSetDevice(h->device));
    AMR_TRY(drain(h));   // the host staging buffer is single: finish what is in flight first
    if (n_blocks == 1) {
        // the unchanged main.go loop (main.go:235): the block goes into a pinned buffer of the handle's (8 to 16 KiB: a
        // memcpy) and ONE launch reads it from there -- no host-to-device copy, no device-to-host copy
        if (!h->h_iq1) HIP_TRY(hipHostMalloc((void **)&h->h_iq1, (size_t)h->geom.block_size2, hipHostMallocDefault));
        if (single_block_ok(h, n_blocks, h->h_iq1)) {
            memcpy(h->h_iq1, iq, need);
            AMR_TRY(submit_single(h, h->h_iq1));
            return collect(h, res);
        }
    }
    AMR_TRY(stage_host_input(h, iq, need));
    AMR_TRY(submit(h, h->d_iq, n_blocks, true));
    return collect(h, res);
}

amr_status amr_decode_batch_device(amr_handle *h, const void *d_iq, size_t n_blocks, amr_result *res)
{
    if (!h || !d_iq) return fail(AMR_EINVAL, "null argument");
    AMR_TRY(drain(h));
    if (single_block_ok(h, n_blocks, d_iq)) AMR_TRY(submit_single(h, (const uint8_t *)d_iq));
    else AMR_TRY(submit(h, (const uint8_t *)d_iq, n_blocks, true));
    return collect(h, res);
}

amr_status amr_submit_device(amr_handle *h, const void *d_iq, size_t n_blocks)
{
    if (!h || !d_iq) return fail(AMR_EINVAL, "null argument");
    if (n_blocks == 0) return fail(AMR_EINVAL, "n_blocks out of range");
    return submit(h, (const uint8_t *)d_iq, n_blocks, true, true);
}

amr_status amr_set_deferral(amr_handle *h, int32_t on)
{
    if (!h) return fail(AMR_EINVAL, "null handle");
    if (on && h->r900_pid >= 0) return fail(AMR_EINVAL, "amr_set_deferral: not available with amr_r900_enable (its second stage reads the batch's IQ by block)");
    if (!on && h->n_head) return fail(AMR_EINVAL, "amr_set_deferral: blocks are deferred: amr_flush first");
    h->defer_on = on != 0;
    return AMR_OK;
}

amr_status amr_flush(amr_handle *h, amr_result *res)
{
    if (!h) return fail(AMR_EINVAL, "null handle");
    if (h->n_pending) return fail(AMR_EINVAL, "amr_flush: batches in flight: collect them first");
    if (h->n_head == 0) {          // nothing deferred: an empty result
        h->r_off.assign(h->sg.n_pre + 1, 0);
        h->last_total = 0;
        h->last_searched = 0;
        h->last_empty = true;      // a gather posted for this result sends zero records, not the previous batch's again
        if (res) {
            *res = amr_result{};
            res->n_preambles = h->sg.n_pre;
            res->pkt_bytes = h->sg.pkt_bytes;
            res->preamble_offset = h->r_off.data();
            res->r900_preamble = h->r900_pid;
            res->first_block = h->calls_done + h->block_base;
        }
        return AMR_OK;
    }
    AMR_TRY(submit(h, h->d_head + h->halo_bytes, 0, true));   // the deferred blocks alone: one partial wave-tile
    return collect(h, res);
}

amr_status amr_submit_host(amr_handle *h, const uint8_t *iq, size_t iq_bytes, size_t n_blocks)
{
    if (!h || !iq) return fail(AMR_EINVAL, "null argument");
    const size_t need = n_blocks * (size_t)h->geom.block_size2;
    if (iq_bytes < need) return fail(AMR_EINVAL, "short input (the Go decoder panics here, decode.go:222)");
    if (n_blocks == 0) return fail(AMR_EINVAL, "n_blocks out of range");
    if (h->n_pending >= kMaxPending) return fail(AMR_EINVAL, "three batches already in flight: call amr_collect first");
    HIP_TRY(hipSetDevice(h->device));
    Slot &s = h->slot[h->next_slot];   // the slot submit() is about to use; its previous batch has been collected
    if (need > s.iq_stage_cap) {
        AMR_TRY(dev_realloc(s.d_iq_stage, need));
        s.iq_stage_cap = need;
    }
    HIP_TRY(hipMemcpyAsync(s.d_iq_stage, iq, need, hipMemcpyHostToDevice, h->h2d_stream));
    HIP_TRY(hipEventRecord(s.ev_h2d, h->h2d_stream));
    HIP_TRY(hipStreamWaitEvent(h->stream, s.ev_h2d, 0));
    return submit(h, s.d_iq_stage, n_blocks, true, true);
}

amr_status amr_collect(amr_handle *h, amr_result *res)
{
    if (!h) return fail(AMR_EINVAL, "null argument");
    return collect(h, res);
}

amr_status amr_result_device(const amr_handle *h, const void **d_packed, uint64_t *n_hits)
{
    if (!h || !d_packed || !n_hits) return fail(AMR_EINVAL, "null argument");
    if (h->last_slot < 0 && !h->last_empty) return fail(AMR_EINVAL, "no batch collected yet");
    if (h->last_empty) { *d_packed = nullptr; *n_hits = 0; return AMR_OK; }   // amr_flush with nothing deferred
    *d_packed = h->validate ? h->slot[h->last_slot].d_val : h->slot[h->last_slot].d_out;
    *n_hits = h->last_total;
    return AMR_OK;
}

amr_status amr_prime(amr_handle *h, const uint8_t *lead, const uint8_t *halo_iq, size_t n_blocks, int on_device)
{
    if (!h || !halo_iq) return fail(AMR_EINVAL, "null argument");
    // a launch without a search would demodulate the deferred blocks, drop their hits and leave every later call index
    // short by their number
    if (h->n_head) return fail(AMR_EINVAL, "amr_prime: blocks are deferred: amr_flush first");
    HIP_TRY(hipSetDevice(h->device));
    if (lead) {
        HIP_TRY(hipMemcpyAsync(h->d_head, lead, h->halo_bytes,
                               on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
        h->zero_halo = false;
        h->d_carry_cur = h->d_head;
    }
    AMR_TRY(drain(h));
    const uint8_t *src = halo_iq;
    if (!on_device) {
        AMR_TRY(stage_host_input(h, halo_iq, n_blocks * (size_t)h->geom.block_size2));
        src = h->d_iq;
    }
    AMR_TRY(submit(h, src, n_blocks, false));
    return collect(h, nullptr);
}

// The byte Decoder.Slice's never-cleared d.pkt carries from one hit to the next (k3_stale.h), across a shard boundary.
amr_status amr_get_stale_carry(amr_handle *h, uint8_t *last_byte)
{
    if (!h || !last_byte) return fail(AMR_EINVAL, "null argument");
    if (h->n_pending) return fail(AMR_EINVAL, "amr_get_stale_carry: batches in flight: collect them first");
    HIP_TRY(hipSetDevice(h->device));
    AMR_TRY(drain(h));
    HIP_TRY(hipMemcpy(last_byte, h->d_pkt_carry + h->carry_slot, 1, hipMemcpyDeviceToHost));
    return AMR_OK;
}

amr_status amr_set_stale_carry(amr_handle *h, uint8_t last_byte)
{
    if (!h) return fail(AMR_EINVAL, "null argument");
    if (h->n_pending) return fail(AMR_EINVAL, "amr_set_stale_carry: batches in flight: collect them first");
    HIP_TRY(hipSetDevice(h->device));
    AMR_TRY(drain(h));
    HIP_TRY(hipMemcpy(h->d_pkt_carry + h->carry_slot, &last_byte, 1, hipMemcpyHostToDevice));
    return AMR_OK;
}

amr_status amr_copy_quantized(amr_handle *h, uint8_t *out, size_t out_bytes)
{
    if (!h || !out) return fail(AMR_EINVAL, "null argument");
    const size_t words = h->last_n_blocks * h->sg.wpb;
    if (out_bytes < words * 4) return fail(AMR_EINVAL, "output buffer too small");
    if (words == 0 || h->last_slot < 0) return AMR_OK;
    HIP_TRY(hipSetDevice(h->device));
    AMR_TRY(drain(h));
    if (words > h->untile_words) {
        AMR_TRY(dev_realloc(h->d_untile, words));
        h->untile_words = words;
    }
    hipLaunchKernelGGL(amr::k_untile, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, h->stream, h->slot[h->last_slot].d_qt, h->d_untile,
                       (uint32_t)h->last_n_blocks, h->sg.lg_wpb);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, h->d_untile, words * 4, hipMemcpyDeviceToHost, h->stream));
    AMR_TRY(sync_compute(h));
    return AMR_OK;
}

}  // extern "C"
