// amrdemod.hip -- C ABI (include/amrdemod.h) over the gfx950 kernels.
//
// Host side of the drop-in boundary for rtlamr's protocol.Decoder
// (protocol/decode.go).  Geometry and registration follow the reference
// line by line (citations inline); everything that touches samples runs on the
// GPU.  There is NO CPU fallback: without a gfx950 device amr_create fails
// with AMR_ENODEV.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>
#include <sched.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "amrdemod.h"
#include "launch.h"
#include "k3_slice.h"
#include "k4_r900.h"
#include "synth.h"

namespace {

thread_local std::string g_last_error;

// AMR_DEBUG_SYNC=1: synchronise and log after every kernel (localises a faulting kernel)
bool debug_sync() { static const bool on = getenv("AMR_DEBUG_SYNC") != nullptr; return on; }
#define AMR_DBG(st, what)                                                                  \
    do {                                                                                   \
        if (debug_sync()) { fprintf(stderr, "[amr] %s ...", what); fflush(stderr);         \
            hipError_t e_ = hipStreamSynchronize(st); fprintf(stderr, " %s\n", hipGetErrorString(e_)); } \
    } while (0)

amr_status fail(amr_status s, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    g_last_error = buf;
    return s;
}

#define HIP_TRY(expr)                                            \
    do {                                                         \
        hipError_t e_ = (expr);                                  \
        if (e_ != hipSuccess) return fail(AMR_EHIP, #expr, e_);  \
    } while (0)

uint32_t ilog2(uint32_t v)
{
    uint32_t l = 0;
    while ((1u << l) < v) ++l;
    return l;
}

bool legal_chip_length(int cl)
{
    // flags.go:127-132
    switch (cl) {
    case 8: case 32: case 40: case 48: case 56: case 64: case 72: case 80: case 88: case 96: return true;
    default: return false;
    }
}

}  // namespace

// One in-flight batch.  Four slots, up to three batches in flight: the host reads back batch i (copy stream) while the
// GPU runs batches i+1 and i+2; the quantized history flows slot -> next slot (see "4b" in DESIGN.md).
struct Slot {
    uint32_t *d_qt = nullptr;     size_t qt_tiles = 0;     // tiled bitstream, tile 0 = history tile
    uint32_t *d_counts = nullptr; size_t cnt_tiles = 0;
    uint32_t *d_gcnt = nullptr; uint32_t gcnt_words = 0;     // hit counts summed over groups of 64 tiles (K2 -> K3)
    uint64_t *d_offs_pre = nullptr;                        // [n_pre+1] + overflow word behind it
    uint32_t *d_overflow = nullptr;
    uint32_t *d_staging = nullptr; size_t staging_tiles = 0; uint32_t stage_cap = 1024;
    // result of a batch, packed: [hit_block u64 x n | hit_idx u32 x n | pkt bytes x n], n = total hits, so that
    // ONE device-to-host copy of (12 + pkt_bytes) * n bytes brings it over
    uint8_t *d_out = nullptr; uint64_t out_cap = 0;
    // pinned host mirrors
    uint64_t *h_off = nullptr;    // [AMR_MAX_PREAMBLES+1]
    uint32_t *h_ovf = nullptr;
    uint8_t *h_out = nullptr; uint64_t host_cap = 0;
    uint8_t *d_r900 = nullptr; uint8_t *h_r900 = nullptr; uint64_t r900_host_cap = 0;   // [out_cap][42] digits (r900 enabled)
    // validation (amr_set_validation): the surviving hits, packed like d_out, and the scratch of the compaction
    uint8_t *d_val = nullptr; uint8_t *d_keep = nullptr; uint64_t *d_listoff = nullptr;   // K5: see k5_validate.h
    uint64_t *d_offs_val = nullptr; uint64_t *h_offv = nullptr;   // [AMR_MAX_PREAMBLES+1] each
    uint8_t *d_iq_stage = nullptr; size_t iq_stage_cap = 0;   // device copy of a host-resident batch (amr_submit_host)
    hipEvent_t ev_h2d = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_s = nullptr, ev2 = nullptr;   // K1 start/stop, K2 start, K3 stop (timing levels 1/2)
    uint64_t *h_done = nullptr;   // pinned, coherent: the batch's last kernel stores the batch ticket here
    uint64_t ticket = 0;          // value that marks the batch in flight as complete
    int timed = 0;                // timing level the batch in flight was submitted with
    bool tail_enqueued = true;    // K3.. of the batch in flight have been launched (false: collect launches them)
    bool tail_split = false;      // ... on the second stream
    bool tail_gated = false;      // ... enqueued ahead, behind k_gate: K3 then runs next to the following K1's end and search
    bool folded = false;          // the state update ran inside the search kernel: no stream-A ticket for this batch
    hipEvent_t ev_k2 = nullptr, ev_t = nullptr;   // K2 stop, K3 start (timing level 2 with the tail on the second stream)
    hipEvent_t ev_pack = nullptr; bool pack_pending = false;   // multi-GPU gather: its pack kernel still reads d_out / d_val of this slot
    // the batch in flight
    bool pending = false, search = false;
    const uint8_t *d_iq = nullptr;
    size_t n_blocks = 0;
    uint32_t n_tiles = 0;
    uint64_t calls_base = 0;
    bool dense = false;           // searched with the dense kernel from the start (dense_hold)
    uint32_t iqhist_valid = 0;    // real samples in the IQ history this batch sees (r900)
    int iqhist_buf = 0;           // which history buffer it reads
};

struct Comm;

// Up to three batches in flight over four slots: the state update of batch i writes the history rows into the slot
// batch i+1 will use, which must not belong to a batch that is still in flight.
constexpr int kSlots = 4;
// Batches of up to this many samples (and at most 8192 blocks: the waves then sit on the chip side by side) run K1 as one
// wave per block throughout.  BlockSize 4096 on an idle MI355X: 49 us up to 512 blocks, 57 at 2048, 75 at 4096, 125 at 8192,
// against 100 us for any number of wave-tiles up to a chip-filling 2048 (tools/coop_sweep.py).
constexpr uint64_t kK1CoopMaxSamples = 1ull << 24;
constexpr uint64_t kK1CoopMaxBlocks = 8192;
constexpr int kMaxPending = 3;
constexpr int kIqHist = 5;   // r900 IQ history buffers, rotating: a batch in flight keeps its own until it is collected

struct amr_handle {
    int device = 0;
    int n_cus = 256;            // compute units of the device (K1 launches one chip-filling round at a time)
    amr_geometry geom{};
    amr::SearchGeom sg{};
    std::vector<int> proto_pid;
    float lut[256];
    uint32_t halo_bytes = 0;   // HBA: aligned halo K1 reads before a block
    uint32_t hist_rows = 0;    // ceil(PL/BS)

    hipStream_t own_stream = nullptr, stream = nullptr, copy_stream = nullptr, h2d_stream = nullptr;
    // K3 / K4 / K5 of batch i run here, next to the search of batch i+1, once the caller pipelines (lazy_tail)
    hipStream_t tail_stream = nullptr;
    // The host launches the tail when it sees the next batch's search start (a pinned flag; no event on the compute
    // stream: stream dependencies were tried and cost ~10 us of bubbles per batch, cfg2 0.237 ms per step against 0.227).
    bool lazy_tail = false;
    uint64_t *d_tail_done = nullptr;   // device word: ticket of the last batch whose second-stream part has finished
    uint64_t *d_k1_started = nullptr;  // device word: ticket of the last batch whose K1 has all its waves on the chip (k_gate)
    uint64_t *h_flags = nullptr;  // pinned: [0] ticket of the last batch whose search has started, [1] whose stream-A part is done
    bool timing_valid = false;
    amr_timing timing{};
    int timing_level = 0;
    uint64_t next_ticket = 1;

    float *d_lut = nullptr;
    // head buffer: [the HBA stream bytes in front of the next launch's row 0 (the IQ halo of that block) | 64 rows:
    // blocks deferred from the last batch, completed by the next submit with its first blocks]
    uint8_t *d_head = nullptr;
    bool defer_on = false;       // amr_set_deferral
    uint32_t n_head = 0;         // deferred blocks waiting in the head buffer
    bool zero_halo = true;
    bool dense_search = false;   // test hook (AMR_DENSE_SEARCH): always use the fallback search kernel
    uint64_t k1_coop_max = 0;    // batches of up to this many blocks run K1 as one wave per block throughout (k1_coop.h):
                                 // from kK1CoopMaxSamples / kK1CoopMaxBlocks; test hook AMR_K1_COOP_MAX (0: only the blocks
                                 // behind the last whole wave-tile -- keeps the tile kernels under the small-batch tests)
    uint64_t init_hit_cap = 1 << 16;   // hits the result buffers hold at first (test hook AMR_HIT_CAP: exercise the growth)
    int dense_streak = 0;        // consecutive batches whose sparse lists overflowed; >= 4: stay dense for a while
    int dense_hold = 0;          // batches left in which the dense kernel is used straight away
    uint8_t *d_iq = nullptr;      size_t iq_cap = 0;       // staging for host input
    uint32_t *d_untile = nullptr; size_t untile_words = 0;

    struct Comm *comm = nullptr;   // multi-GPU hit gather (amr_comm_init), see the section at the end of this file

    Slot slot[kSlots];
    int next_slot = 0;           // slot the next submit uses
    int n_pending = 0;           // submitted, not yet collected (oldest = next_slot - n_pending)
    int last_slot = -1;          // slot of the last collected batch (amr_copy_quantized, result storage)
    bool last_empty = false;     // the last result was the empty one of an amr_flush with nothing deferred: amr_gather_hits /
                                 // amr_result_device then report zero records instead of the previous batch's
    uint64_t calls_done = 0, block_base = 0;
    size_t last_n_blocks = 0;
    std::vector<uint64_t> r_off;
    uint64_t last_total = 0;
    // r900 second stage: the preamble id; the PL samples of IQ that precede the next batch live in d_iqhist below
    int r900_pid = -1;
    // per-hit validation on the device
    bool validate = false;
    amr::ValRule rules[AMR_MAX_PREAMBLES] = {};
    uint64_t last_searched = 0;   // hits the search of the last collected batch found (before validation)
    uint8_t *d_iqhist[kIqHist] = {};   // rotating: a batch in flight keeps its own for K4 and for a re-run
    int iqhist_cur = 0;
    uint32_t iqhist_valid = 0;
};

namespace {

template <typename T>
amr_status dev_realloc(T *&p, size_t count)
{
    if (p) { hipError_t e = hipFree(p); p = nullptr; if (e != hipSuccess) return fail(AMR_EHIP, "hipFree", e); }
    if (count == 0) return AMR_OK;
    hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
    if (e != hipSuccess) { p = nullptr; return fail(AMR_ENOMEM, "hipMalloc", e); }
    return AMR_OK;
}

template <typename T>
amr_status host_realloc(T *&p, size_t count)
{
    if (p) { hipError_t e = hipHostFree(p); p = nullptr; if (e != hipSuccess) return fail(AMR_EHIP, "hipHostFree", e); }
    if (count == 0) return AMR_OK;
    hipError_t e = hipHostMalloc((void **)&p, count * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) { p = nullptr; return fail(AMR_ENOMEM, "hipHostMalloc", e); }
    return AMR_OK;
}

#define AMR_TRY(expr)                          \
    do {                                       \
        amr_status s_ = (expr);                \
        if (s_ != AMR_OK) return s_;           \
    } while (0)

// Timing events ride on the kernel dispatches themselves (hipExtLaunchKernelGGL start/stop events): a separate
// hipEventRecord costs a ~6 us bubble on the stream each, four of them per batch were 6 % of a 1 GiB step.
// The kernels are launched through launch.h: one translation unit per kernel family.

// Wait for the compute stream.  With K3.. of some batches still unlaunched (pipelined callers), a k_hist_update on the
// stream may be waiting for one of them: launch them first (each as soon as its own search has finished), or the wait
// would only end at that kernel's 2 ms time-out.
amr_status sync_compute(amr_handle *h);

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
amr_status enqueue_k2(amr_handle *h, Slot &s, hipStream_t st, bool rerun, bool dense, bool split,
                      const amr::HistArgs *fold = nullptr, bool *folded = nullptr)
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
    if (walk_ok) {
        const uint32_t n_wg = (s.n_tiles + amr::kK2WWaves - 1) / amr::kK2WWaves + extra;
        const uint32_t grid = 8u * ((n_wg + 7u) / 8u);   // XCD-contiguous tile order: 8 equal runs
        const size_t lds = amr::k2_walk_lds_bytes(h->hist_rows * h->sg.wpb);
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
    const size_t k3lds = amr::k3_lds_bytes(h->sg, h->validate);
    k3.lds_bytes = (uint32_t)k3lds;
    HIP_TRY(hipFuncSetAttribute((const void *)amr::k3_slice_words, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k3lds));
    // one workgroup per (tile, preamble) list (every list of a tile in one workgroup with shared row staging measured slower
    // on the four-preamble decoder: 188 against 173 us per 4 GiB); k3_fold: when the history tile's workgroup would be the
    // one too many for whole rounds of the chip, workgroup 0 takes its list as well
    k3.fold = amr::k3_fold(s.n_tiles, n_pre, (uint32_t)h->n_cus * 8u) ? 1u : 0u;
    hipExtLaunchKernelGGL(amr::k3_slice_words, dim3(s.n_tiles - k3.fold, n_pre), dim3(256), k3lds, st, k3e0, k3e1, 0, k3);
    HIP_TRY(hipGetLastError());
    AMR_DBG(st, "k3_slice");
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
    k1.carry = h->d_head;
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
    amr::K1Args k1_last = k1;                     // the launch that announces itself to the gate: the batch's last one
    if (gate_prev) { k1_last.started = h->d_k1_started; k1_last.started_value = s.ticket; }
    // One launch per "round" for long blocks: K1 holds 8 waves per CU, and a launch that exactly fills the chip keeps its
    // waves in step -- all of them read together and write their output bursts together.  A larger grid runs the later
    // rounds out of step (output stores trickle into the read stream all the time): BlockSize 4096, 4 GiB: 0.895 ms in
    // one launch, 4 x 0.179 ms in four; IDM (BlockSize 8192, 4 GiB) 0.860 -> 0.804 ms.  Short blocks (a round lasts
    // under 0.1 ms) lose more at the extra launch boundaries than they gain: BlockSize 2048 0.182 -> 0.256 ms, so they
    // keep the single launch.
    const uint32_t round = bs >= 4096 ? (uint32_t)h->n_cus * 8u : full;
    // Small batches entirely as one wave per block (k1_coop.h): a wave-tile costs a whole wave life (150-175 us) however few
    // tiles there are; a wave per block finishes in ~50 us as long as the waves fit the chip side by side (all_coop).
    if (all_coop) {
        amr::launch_k1_coop(h->geom.chip_length, 0u, (uint32_t)rows, st, k1_last, e0, e1);
    } else {
        for (uint32_t w0 = 0; w0 < full; w0 += round) {
            const uint32_t n = std::min(round, full - w0);
            const bool last = w0 + n == full && !rem;
            amr::K1Args &kk = last ? k1_last : k1;
            kk.wg_first = w0;
            amr::launch_k1(h->geom.chip_length, dim3(n), st, kk, w0 == 0 ? e0 : nullptr, last ? e1 : nullptr);
        }
        if (rem)     // the blocks behind the last whole wave-tile (sync callers, flush): a wave each
            amr::launch_k1_coop(h->geom.chip_length, full * 64u, rem, st, k1_last, full ? nullptr : e0, e1);
    }
    HIP_TRY(hipGetLastError());
    AMR_DBG(st, "k1_demod");
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
        hipLaunchKernelGGL(amr::k_gate, dim3(1), dim3(1), 0, h->tail_stream, h->d_k1_started, s.ticket, 600u /* 6 us */);
        HIP_TRY(hipGetLastError());
        AMR_TRY(launch_tail(h, prev));
        prev.tail_gated = true;
    }
    s.dense = h->dense_hold > 0;
    if (s.dense) h->dense_hold--;
    // state carried to the next batch (decode.go:165-166): the last rows of this slot's bitstream become the history
    // tile of the NEXT slot, the last HBA bytes of IQ (and the deferred blocks behind them) go to the head buffer, the
    // next slot's search words are reset.  Whatever does it is also the last thing in front of the next K1 launch, which
    // must not meet the previous batch's K3.. (it needs every wave slot): it waits for them on a device word.
    const uint8_t *launch_end = rows > n_head ? d_iq + (rows - n_head) * bs2 : head_rows + rows * bs2;
    amr::HistArgs ha{s.d_qt, other.d_qt, (uint32_t)rows, h->hist_rows, h->sg.wpb, h->sg.lg_wpb,
                     launch_end - h->halo_bytes, h->d_head, h->halo_bytes,
                     (uint32_t)(new_head * bs2), new_head ? 16u : 0u, other.d_overflow,
                     other.d_gcnt, other.gcnt_words,
                     lazy ? nullptr : s.h_done, s.ticket, &h->h_flags[1],
                     (prev.pending && prev.search && prev.tail_split) ? h->d_tail_done : nullptr, prev.ticket};
    // pipelined callers: the update rides along with the search as more workgroups of its launch instead of following it
    // as a 5 us kernel
    bool folded = false;
    if (search) {
        if (lazy) AMR_TRY(enqueue_k2(h, s, st, false, s.dense, true, &ha, &folded));
        else AMR_TRY(enqueue_search(h, s, false, s.dense));
    }
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
    if (wait) { HIP_TRY(hipStreamSynchronize(h->stream)); *yes = true; return AMR_OK; }
    const hipError_t e = hipStreamQuery(h->stream);
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

amr_status sync_compute(amr_handle *h)
{
    for (int k = 0; k < h->n_pending; ++k) {
        Slot &t = h->slot[(h->next_slot - h->n_pending + k + 2 * kSlots) % kSlots];
        if (!t.search || t.tail_enqueued) continue;
        bool ready = false;
        AMR_TRY(search_finished(h, k, true, &ready));
        AMR_TRY(launch_tail(h, t));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipStreamSynchronize(h->tail_stream));
    return AMR_OK;
}

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
    if (s.search) {
        for (int attempt = 0;; ++attempt) {
            const uint32_t ovf = *s.h_ovf;
            total = s.h_off[n_pre];
            if (attempt > 8) return fail(AMR_EOVERFLOW, "hit capacity could not be grown");
            bool rerun = false;
            // sparse hit list overflowed (e.g. the zero history of a fresh stream matches r900's 16 leading zeros):
            // this batch is searched again with the dense kernel; the next one starts sparse again unless
            // overflows keep coming
            if (ovf & 2u) { use_dense = true; rerun = true; }
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
        }
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
        if (total) {   // on the copy stream: overlaps the next batch's kernels
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
            for (;;) {
                const hipError_t qe = hipStreamQuery(h->copy_stream);
                if (qe == hipSuccess) break;
                if (qe != hipErrorNotReady) return fail(AMR_EHIP, "hipStreamQuery(copy stream)", qe);
                AMR_TRY(launch_ready_tails(h, false));
                cpu_relax();
            }
        }
    }
    float a = 0, b = 0, c = 0;
    h->timing_valid = false;
    if (s.timed && hipEventSynchronize(s.ev1) == hipSuccess && hipEventElapsedTime(&a, s.ev0, s.ev1) == hipSuccess) {
        float b2 = 0;
        if (s.timed >= 2 && s.search && s.tail_split && hipEventSynchronize(s.ev2) == hipSuccess &&
            hipEventElapsedTime(&b, s.ev_s, s.ev_k2) == hipSuccess && hipEventElapsedTime(&b2, s.ev_t, s.ev2) == hipSuccess)
            // K2 and the tail ran apart: their durations, added up -- unless the tail was let in at the following K1's start
            // (tail_gated): its workgroups then trickle in where K1 waves retire and its "duration" spans that whole K1;
            // what the batch cost the compute stream besides K1 is its K2
            h->timing = s.tail_gated ? amr_timing{a, b, a + b} : amr_timing{a, b + b2, a + b + b2};
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

amr_status drain(amr_handle *h)
{
    while (h->n_pending) AMR_TRY(collect(h, nullptr));
    return AMR_OK;
}

// RegisterProtocol for every entry + the arithmetic of Allocate (decode.go:100-141): no device involved.
amr_status plan_geometry(const amr_protocol *protos, int32_t n_protos, amr_geometry &g, amr::SearchGeom &sg,
                         std::vector<int> &proto_pid, uint32_t &halo_bytes, uint32_t &hist_rows)
{
    g = amr_geometry{};
    sg = amr::SearchGeom{};
    proto_pid.clear();
    // RegisterProtocol, decode.go:100-128: field-wise max, preambles grouped by value
    for (int i = 0; i < n_protos; ++i) {
        const amr_protocol &p = protos[i];
        if (!p.preamble || !legal_chip_length(p.chip_length) || p.preamble_symbols <= 0 || p.packet_symbols <= 0 ||
            (g.chip_length && p.chip_length != g.chip_length)) {
            return fail(AMR_EINVAL, "amr_create: bad protocol entry (chip length must be one of flags.go:127-132)");
        }
        const size_t len = strlen(p.preamble);
        if (len == 0 || len > AMR_MAX_PREAMBLE_BITS) { return fail(AMR_EINVAL, "preamble length"); }
        g.data_rate = std::max(g.data_rate, p.data_rate);
        g.chip_length = std::max(g.chip_length, p.chip_length);
        g.preamble_symbols = std::max(g.preamble_symbols, p.preamble_symbols);
        g.packet_symbols = std::max(g.packet_symbols, p.packet_symbols);
        uint64_t bits = 0;
        for (size_t b = 0; b < len; ++b) {
            if (p.preamble[b] != '0' && p.preamble[b] != '1') { return fail(AMR_EINVAL, "preamble must be 0/1"); }
            if (p.preamble[b] == '1') bits |= 1ull << b;
        }
        int pid = -1;
        for (uint32_t q = 0; q < sg.n_pre; ++q)
            if (sg.pre_len[q] == len && sg.pre_bits[q] == bits) pid = (int)q;
        if (pid < 0) {
            if (sg.n_pre == AMR_MAX_PREAMBLES) { return fail(AMR_EINVAL, "too many distinct preambles"); }
            pid = (int)sg.n_pre++;
            sg.pre_len[pid] = (uint32_t)len;
            sg.pre_bits[pid] = bits;
        }
        proto_pid.push_back(pid);
    }
    // Allocate, decode.go:131-141
    g.symbol_length = g.chip_length << 1;
    g.sample_rate = g.data_rate * g.chip_length;
    g.preamble_length = g.preamble_symbols * g.symbol_length;
    g.packet_length = g.packet_symbols * g.symbol_length;
    g.block_size = 1 << (unsigned)std::ceil(std::log2((double)g.preamble_length));  // NextPowerOf2, decode.go:377-379
    g.block_size2 = g.block_size << 1;
    g.buffer_length = g.packet_length + g.block_size;
    g.n_preambles = (int32_t)sg.n_pre;
    g.pkt_bytes = (g.packet_symbols + 7) >> 3;

    sg.block_size = (uint32_t)g.block_size;
    sg.lg_block_size = ilog2(sg.block_size);
    sg.wpb = sg.block_size >> 5;
    sg.lg_wpb = sg.lg_block_size - 5;
    sg.symbol_length = (uint32_t)g.symbol_length;
    sg.packet_length = (uint32_t)g.packet_length;
    sg.packet_symbols = (uint32_t)g.packet_symbols;
    sg.pkt_bytes = (uint32_t)g.pkt_bytes;
    sg.max_pre_len = 0;
    for (uint32_t q = 0; q < sg.n_pre; ++q) sg.max_pre_len = std::max(sg.max_pre_len, sg.pre_len[q]);
    halo_bytes = (uint32_t)((4 * g.chip_length + 127) & ~127);
    hist_rows = (uint32_t)((g.packet_length + g.block_size - 1) / g.block_size);
    // every preamble must fit the search window the geometry provides (true for all rtlamr parsers,
    // where PreambleSymbols >= len(Preamble)); the tiled search needs <= 63 history rows and
    // word-aligned PacketLength
    for (uint32_t q = 0; q < sg.n_pre; ++q)
        if ((int)sg.pre_len[q] > g.preamble_symbols) { return fail(AMR_EINVAL, "preamble longer than PreambleSymbols"); }
    // the kernels index a row's words with 8 bits and stage whole rows in LDS: BlockSize <= 8192
    if (hist_rows > 63 || (g.packet_length & 63) || g.block_size < 256 || g.block_size > 8192 || g.packet_symbols < g.preamble_symbols) {
        return fail(AMR_EINVAL, "geometry outside the supported range");
    }

    return AMR_OK;
}

}  // namespace

extern "C" {

amr_status amr_create(const amr_protocol *protos, int32_t n_protos, int32_t device_id, amr_handle **out)
{
    if (!protos || n_protos <= 0 || !out) return fail(AMR_EINVAL, "amr_create: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(AMR_ENODEV, "no HIP device visible: amrdemod has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(AMR_ENODEV, "device_id out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(AMR_ENODEV, "device is not gfx950 (MI355X): kernels are built for gfx950 only");

    amr_handle *h = new (std::nothrow) amr_handle();
    if (!h) return fail(AMR_ENOMEM, "new amr_handle");
    h->device = device_id;
    h->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->dense_search = getenv("AMR_DENSE_SEARCH") != nullptr;   // test hook: force the fallback search kernel
    if (const char *hc = getenv("AMR_HIT_CAP")) h->init_hit_cap = std::max<uint64_t>(256, strtoull(hc, nullptr, 10));

    {
        amr_status ps = plan_geometry(protos, n_protos, h->geom, h->sg, h->proto_pid, h->halo_bytes, h->hist_rows);
        if (ps != AMR_OK) { delete h; return ps; }
    }
    h->k1_coop_max = std::min<uint64_t>(kK1CoopMaxBlocks, kK1CoopMaxSamples / (uint64_t)h->geom.block_size);
    if (const char *cm = getenv("AMR_K1_COOP_MAX")) h->k1_coop_max = strtoull(cm, nullptr, 10);   // test hook / A-B
    // NewMagLUT, decode.go:209-216: float32 divide then float32 square, two roundings per entry.
    for (int i = 0; i < 256; ++i) {
        volatile float q = (127.5f - (float)i) / 127.5f;
        volatile float sq = q * q;
        h->lut[i] = sq;
    }

    hipError_t e = hipSetDevice(device_id);
    // (Queue priorities -- compute stream highest, tail stream lowest -- change nothing measurable; without the wait for the
    // previous batch's K3 in front of a K1 launch, priorities or not, K1 takes 0.27 ms instead of 0.18.)
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->h2d_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->tail_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->h_flags, 16, hipHostMallocCoherent);
    if (e == hipSuccess) { h->h_flags[0] = 0; h->h_flags[1] = 0; }
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_tail_done, 16);
    if (e == hipSuccess) e = hipMemset(h->d_tail_done, 0, 16);
    if (e == hipSuccess) h->d_k1_started = h->d_tail_done + 1;
    h->stream = h->own_stream;
    for (Slot &sl : h->slot) {
        if (e == hipSuccess) e = hipEventCreate(&sl.ev0);
        if (e == hipSuccess) e = hipEventCreate(&sl.ev1);
        if (e == hipSuccess) e = hipEventCreate(&sl.ev2);
        if (e == hipSuccess) e = hipEventCreate(&sl.ev_s);
        if (e == hipSuccess) e = hipEventCreate(&sl.ev_k2);
        if (e == hipSuccess) e = hipEventCreate(&sl.ev_t);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_pack, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_h2d, hipEventDisableTiming);
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_done, 8, hipHostMallocCoherent);
        if (e == hipSuccess) *sl.h_done = 0;
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_offs_pre, (AMR_MAX_PREAMBLES + 1) * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_overflow, 4);
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_offs_val, (AMR_MAX_PREAMBLES + 1) * 8);
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_offv, (AMR_MAX_PREAMBLES + 1) * 8, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMemset(sl.d_overflow, 0, 4);
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_off, (AMR_MAX_PREAMBLES + 1) * 8, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_ovf, 4, hipHostMallocDefault);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_lut, 1024);
    const size_t head_bytes = h->halo_bytes + (size_t)64 * h->geom.block_size2;
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_head, head_bytes);
    if (e == hipSuccess) e = hipMemcpy(h->d_lut, h->lut, 1024, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(h->d_head, 0, head_bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();   // the memsets above ran on the null stream; ours is non-blocking
    if (e != hipSuccess) { amr_destroy(h); return fail(AMR_EHIP, "amr_create: device setup", e); }
    *out = h;
    return AMR_OK;
}

amr_status amr_plan(const amr_protocol *protos, int32_t n_protos, amr_geometry *geom, int32_t *preamble_ids)
{
    if (!protos || n_protos <= 0 || !geom) return fail(AMR_EINVAL, "amr_plan: null argument");
    amr::SearchGeom sg;
    std::vector<int> pid;
    uint32_t halo = 0, hist = 0;
    AMR_TRY(plan_geometry(protos, n_protos, *geom, sg, pid, halo, hist));
    if (preamble_ids)
        for (int32_t i = 0; i < n_protos; ++i) preamble_ids[i] = pid[(size_t)i];
    return AMR_OK;
}

amr_status amr_destroy(amr_handle *h)
{
    if (!h) return AMR_OK;
    (void)hipSetDevice(h->device);
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
    // the communicator first: its stream may still hold a pack kernel that reads the slots' result buffers
    if (h->comm) (void)amr_comm_destroy(h);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->tail_stream) (void)hipStreamSynchronize(h->tail_stream);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    void *ptrs[] = {h->d_lut, h->d_head, h->d_iq, h->d_untile, h->d_tail_done};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (uint8_t *p : h->d_iqhist) if (p) (void)hipFree(p);
    if (h->h_flags) (void)hipHostFree(h->h_flags);
    for (Slot &sl : h->slot) {
        void *dp[] = {sl.d_qt, sl.d_counts, sl.d_gcnt, sl.d_offs_pre, sl.d_overflow, sl.d_staging, sl.d_out, sl.d_iq_stage, sl.d_r900,
                      sl.d_val, sl.d_keep, sl.d_listoff, sl.d_offs_val};
        if (sl.h_r900) (void)hipHostFree(sl.h_r900);
        if (sl.ev_h2d) (void)hipEventDestroy(sl.ev_h2d);
        for (void *p : dp) if (p) (void)hipFree(p);
        void *hp[] = {sl.h_off, sl.h_ovf, sl.h_out, sl.h_offv};
        for (void *p : hp) if (p) (void)hipHostFree(p);
        hipEvent_t evs[] = {sl.ev0, sl.ev1, sl.ev_s, sl.ev2, sl.ev_k2, sl.ev_t, sl.ev_pack};
        if (sl.h_done) (void)hipHostFree(sl.h_done);
        for (hipEvent_t ev : evs) if (ev) (void)hipEventDestroy(ev);
    }
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->h2d_stream) (void)hipStreamDestroy(h->h2d_stream);
    if (h->tail_stream) (void)hipStreamDestroy(h->tail_stream);
    delete h;
    return AMR_OK;
}

amr_status amr_reset(amr_handle *h)
{
    if (!h) return fail(AMR_EINVAL, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    AMR_TRY(drain(h));
    HIP_TRY(hipStreamSynchronize(h->tail_stream));
    for (Slot &sl : h->slot)
        if (sl.d_qt) HIP_TRY(hipMemsetAsync(sl.d_qt, 0, (size_t)64 * h->sg.wpb * 4, h->stream));
    AMR_TRY(sync_compute(h));
    h->zero_halo = true;
    h->calls_done = 0;
    h->last_n_blocks = 0;
    h->iqhist_valid = 0;
    h->n_head = 0;            // deferred blocks belong to the stream that is forgotten
    return AMR_OK;
}

amr_status amr_get_geometry(const amr_handle *h, amr_geometry *out)
{
    if (!h || !out) return fail(AMR_EINVAL, "null argument");
    *out = h->geom;
    return AMR_OK;
}

int32_t amr_preamble_id(const amr_handle *h, int32_t proto_index)
{
    if (!h || proto_index < 0 || (size_t)proto_index >= h->proto_pid.size()) return -1;
    return h->proto_pid[(size_t)proto_index];
}

amr_status amr_get_mag_lut(const amr_handle *h, float *out256)
{
    if (!h || !out256) return fail(AMR_EINVAL, "null argument");
    memcpy(out256, h->lut, sizeof h->lut);
    return AMR_OK;
}

amr_status amr_r900_enable(amr_handle *h, int32_t proto_index)
{
    if (!h || proto_index < 0 || (size_t)proto_index >= h->proto_pid.size()) return fail(AMR_EINVAL, "bad protocol index");
    // zero_halo is cleared by every submit, also by amr_prime (which does not advance calls_done): enabling afterwards
    // would zero the IQ history the primed blocks left
    if (h->calls_done != 0 || h->n_pending != 0 || !h->zero_halo) return fail(AMR_EINVAL, "amr_r900_enable: call before the first batch");
    if (h->defer_on) return fail(AMR_EINVAL, "amr_r900_enable: not available with amr_set_deferral");
    HIP_TRY(hipSetDevice(h->device));
    h->r900_pid = h->proto_pid[(size_t)proto_index];
    h->rules[h->r900_pid] = amr::ValRule{};   // its hits carry digits by position: never filtered
    const size_t bytes = 2 * (size_t)h->geom.packet_length;
    for (uint8_t *&p : h->d_iqhist) {
        if (!p) AMR_TRY(dev_realloc(p, bytes));
        HIP_TRY(hipMemsetAsync(p, 0, bytes, h->stream));
    }
    for (Slot &sl : h->slot)
        if (sl.out_cap && !sl.d_r900) AMR_TRY(dev_realloc(sl.d_r900, sl.out_cap * amr::kR900Digits));
    h->iqhist_valid = 0;
    return AMR_OK;
}

amr_status amr_set_validation(amr_handle *h, int32_t preamble_id, const amr_validator *v)
{
    if (!h || preamble_id < 0 || (uint32_t)preamble_id >= h->sg.n_pre) return fail(AMR_EINVAL, "bad preamble id");
    if (h->n_pending != 0) return fail(AMR_EINVAL, "amr_set_validation: batches in flight");
    amr::ValRule r{};
    if (v) {
        if (preamble_id == h->r900_pid)
            return fail(AMR_EINVAL, "amr_set_validation: the r900 preamble's hits carry digits and are always kept");
        if (v->n_checks < 0 || v->n_checks > 2 || v->dedupe_bytes < 0 || (uint32_t)v->dedupe_bytes > h->sg.pkt_bytes)
            return fail(AMR_EINVAL, "amr_set_validation: n_checks must be 0..2, dedupe_bytes 0..pkt_bytes");
        r.n_checks = v->n_checks;
        r.dedupe_bytes = v->dedupe_bytes;
        for (int c = 0; c < v->n_checks; ++c) {
            const amr_crc_check &k = v->checks[c];
            if (k.n_spans < 1 || k.n_spans > 2) return fail(AMR_EINVAL, "amr_set_validation: 1 or 2 spans per check");
            r.chk[c].init = k.init; r.chk[c].poly = k.poly; r.chk[c].residue = k.residue; r.chk[c].n_spans = k.n_spans;
            for (int sp = 0; sp < k.n_spans; ++sp) {
                if ((uint32_t)k.span_off[sp] + k.span_len[sp] > h->sg.pkt_bytes)
                    return fail(AMR_EINVAL, "amr_set_validation: span outside the packet");
                r.chk[c].off[sp] = k.span_off[sp];
                r.chk[c].len[sp] = k.span_len[sp];
            }
        }
    }
    h->rules[preamble_id] = r;
    bool any = false;
    for (uint32_t q = 0; q < h->sg.n_pre; ++q) any = any || h->rules[q].n_checks > 0 || h->rules[q].dedupe_bytes > 0;
    h->validate = any;
    return AMR_OK;
}

amr_status amr_set_stream(amr_handle *h, void *hip_stream)
{
    if (!h) return fail(AMR_EINVAL, "null handle");
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return AMR_OK;
}

amr_status amr_set_block_base(amr_handle *h, uint64_t base)
{
    if (!h) return fail(AMR_EINVAL, "null handle");
    // deferred blocks keep the call indices they were submitted under: a new base in front of them would renumber them
    if (h->n_head) return fail(AMR_EINVAL, "amr_set_block_base: blocks are deferred: amr_flush first");
    h->block_base = base;
    return AMR_OK;
}

amr_status amr_decode_batch(amr_handle *h, const uint8_t *iq, size_t iq_bytes, size_t n_blocks, amr_result *res)
{
    if (!h || !iq) return fail(AMR_EINVAL, "null argument");
    const size_t need = n_blocks * (size_t)h->geom.block_size2;
    if (iq_bytes < need) return fail(AMR_EINVAL, "short input (the Go decoder panics here, decode.go:222)");
    HIP_TRY(hipSetDevice(h->device));
    AMR_TRY(drain(h));   // the host staging buffer is single: finish what is in flight first
    AMR_TRY(stage_host_input(h, iq, need));
    AMR_TRY(submit(h, h->d_iq, n_blocks, true));
    return collect(h, res);
}

amr_status amr_decode_batch_device(amr_handle *h, const void *d_iq, size_t n_blocks, amr_result *res)
{
    if (!h || !d_iq) return fail(AMR_EINVAL, "null argument");
    AMR_TRY(drain(h));
    AMR_TRY(submit(h, (const uint8_t *)d_iq, n_blocks, true));
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

amr_status amr_host_alloc(size_t bytes, void **ptr)
{
    if (!ptr || bytes == 0) return fail(AMR_EINVAL, "null argument");
    hipError_t e = hipHostMalloc(ptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(AMR_ENOMEM, "hipHostMalloc", e);
    return AMR_OK;
}

amr_status amr_host_free(void *ptr)
{
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return AMR_OK;
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

size_t amr_halo_bytes(const amr_handle *h) { return h ? h->halo_bytes : 0; }
size_t amr_prime_blocks(const amr_handle *h) { return h ? (size_t)h->hist_rows + 1 : 0; }

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

amr_status amr_set_timing(amr_handle *h, int32_t level)
{
    if (!h || level < 0 || level > 2) return fail(AMR_EINVAL, "timing level must be 0, 1 or 2");
    h->timing_level = level;
    return AMR_OK;
}

amr_status amr_get_timing(const amr_handle *h, amr_timing *out)
{
    if (!h || !out) return fail(AMR_EINVAL, "null argument");
    if (!h->timing_valid) return fail(AMR_EINVAL, "no batch timed yet");
    *out = h->timing;
    return AMR_OK;
}

const char *amr_strerror(amr_status s)
{
    switch (s) {
    case AMR_OK: return "ok";
    case AMR_EINVAL: return "invalid argument";
    case AMR_ENOMEM: return "out of memory";
    case AMR_EHIP: return "HIP runtime error";
    case AMR_ENODEV: return "no gfx950 device (no CPU fallback)";
    case AMR_EOVERFLOW: return "capacity overflow";
    default: return "unknown status";
    }
}

const char *amr_last_error(void) { return g_last_error.c_str(); }

amr_status amr_describe(const amr_handle *h, char *buf, size_t buf_bytes)
{
    if (!h || !buf || buf_bytes == 0) return fail(AMR_EINVAL, "null argument");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, h->device));
    snprintf(buf, buf_bytes, "amrdemod 0.1 %s %d CUs clock %d kHz chip %d BS %d PL %d preambles %d", prop.gcnArchName,
             prop.multiProcessorCount, prop.clockRate, h->geom.chip_length, h->geom.block_size, h->geom.packet_length,
             h->geom.n_preambles);
    return AMR_OK;
}

/* ---- device utilities ---- */

amr_status amr_dev_alloc(int32_t device_id, size_t bytes, void **d_ptr)
{
    if (!d_ptr) return fail(AMR_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(device_id));
    hipError_t e = hipMalloc(d_ptr, bytes);
    if (e != hipSuccess) return fail(AMR_ENOMEM, "hipMalloc", e);
    return AMR_OK;
}
amr_status amr_dev_free(int32_t device_id, void *d_ptr)
{
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(hipFree(d_ptr));
    return AMR_OK;
}
amr_status amr_dev_upload(int32_t device_id, void *d_dst, const void *src, size_t bytes)
{
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return AMR_OK;
}
amr_status amr_dev_download(int32_t device_id, void *dst, const void *d_src, size_t bytes)
{
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return AMR_OK;
}
amr_status amr_dev_sync(int32_t device_id)
{
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(hipDeviceSynchronize());
    return AMR_OK;
}

static amr_status synth_fill(bool uniform, int32_t device_id, void *d_iq, uint64_t n_samples, uint64_t seed, uint64_t first_sample)
{
    if (!d_iq || (n_samples & 7)) return fail(AMR_EINVAL, "n_samples must be a multiple of 8");
    HIP_TRY(hipSetDevice(device_id));
    const uint64_t threads = n_samples / 8;
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (uniform) hipLaunchKernelGGL(amr::k_synth_noise<true>, grid, dim3(256), 0, 0, (uint8_t *)d_iq, n_samples, seed, first_sample);
    else hipLaunchKernelGGL(amr::k_synth_noise<false>, grid, dim3(256), 0, 0, (uint8_t *)d_iq, n_samples, seed, first_sample);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return AMR_OK;
}

amr_status amr_synth_noise(int32_t device_id, void *d_iq, uint64_t n_samples, uint64_t seed, uint64_t first_sample)
{
    return synth_fill(false, device_id, d_iq, n_samples, seed, first_sample);
}

amr_status amr_synth_uniform(int32_t device_id, void *d_iq, uint64_t n_samples, uint64_t seed, uint64_t first_sample)
{
    return synth_fill(true, device_id, d_iq, n_samples, seed, first_sample);
}

amr_status amr_synth_plant(int32_t device_id, void *d_iq, uint64_t n_samples, uint64_t first_sample, int32_t chip_length,
                           uint32_t n_packets, const uint64_t *start, const uint8_t *bits, uint32_t n_bits,
                           uint32_t stride, const int8_t *d_i, const int8_t *d_q)
{
    if (!d_iq || !start || !bits || !d_i || !d_q || chip_length <= 0) return fail(AMR_EINVAL, "null argument");
    if (n_packets == 0) return AMR_OK;
    HIP_TRY(hipSetDevice(device_id));
    uint64_t *ds = nullptr; uint8_t *db = nullptr; int8_t *di = nullptr, *dq = nullptr;
    HIP_TRY(hipMalloc((void **)&ds, n_packets * 8ull));
    HIP_TRY(hipMalloc((void **)&db, (size_t)n_packets * stride));
    HIP_TRY(hipMalloc((void **)&di, n_packets));
    HIP_TRY(hipMalloc((void **)&dq, n_packets));
    HIP_TRY(hipMemcpy(ds, start, n_packets * 8ull, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, bits, (size_t)n_packets * stride, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(di, d_i, n_packets, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dq, d_q, n_packets, hipMemcpyHostToDevice));
    amr::PlantArgs a{(uint8_t *)d_iq, n_samples, first_sample, ds, db, di, dq, n_packets, n_bits, stride,
                     (uint32_t)chip_length};
    const uint32_t per = n_bits * 2u * (uint32_t)chip_length;
    hipLaunchKernelGGL(amr::k_synth_plant, dim3((per + 255) / 256, n_packets), dim3(256), 0, 0, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(ds); (void)hipFree(db); (void)hipFree(di); (void)hipFree(dq);
    return AMR_OK;
}

}  // extern "C"


// =====================================================================================================================
// Multi-GPU: gather of the hit records on one rank.  One process per GPU; independent shards of whole
// blocks need no data-path collective, the only exchange is this gather.  It runs on its own stream through RCCL
// point-to-point calls (every peer sends its records to the root over its own xGMI link; no ring) and is enqueued from
// the host without any synchronisation, so that it overlaps the kernels of the following batches.
//
// Ordering.  amr_collect has seen the batch complete, so its packed result is there; it stays there until K3 / K5 of the
// batch that REUSES the slot (the fourth submit after this one) overwrite it.  The pack kernel that reads it runs on the
// communicator's stream, behind the previous gather's send -- which completes only when the root has posted its
// receive, i.e. a lagging root or peer can hold it back for any length of time.  So the pack kernel is followed by an
// event (Slot::ev_pack) and enqueue_tail() makes the stream that is about to overwrite the slot wait for it: back-pressure
// instead of a timing assumption.  The send buffer of set k is reused by the pack of gather seq + 2 on the same stream,
// i.e. in order behind the send that read it.
//
// What travels is sized by the hit count, not by the capacity, once the capacity is large (round 4; a fixed 1.5 x
// capacity slot was 5.2 MB per rank and step for raw hits whatever the batch held; slots of up to kGatherWholeSlotMax
// -- validated hits -- still travel whole in one message, with no host wait at all).  For the large ones, two phases
// per gather, both on the communicator's stream:
//   1. every rank sends its 128-byte slot header (true count, records sent, per-preamble offsets, sequence number);
//   2. every rank with records sends exactly gather_wire_bytes(n_sent) = 12 * n_sent bytes rounded up to 4 KiB.
// A sender knows its count on the host (amr_collect returned it) and never waits.  The ROOT has to know every peer's
// count before it can post the receives of phase 2 (RCCL point-to-point needs matching sizes): it copies the received
// headers to pinned memory and waits for that copy -- the one host wait of the protocol, 128 bytes per rank, on the
// root only, and only as long as the slowest peer takes to post the same gather.
//
// Root side.  Behind the receives of a gather, on the same stream, one kernel mirrors every rank's slot (header + the
// records it holds) into pinned host memory and an event marks its arrival:
// amr_gather_fetch(seq, rank) waits for that event only -- no stream synchronisation, no blocking copy -- and returns
// pointers into the mirror.  Two sets alternate: the records of gather `seq` stay valid until gather seq + 2 is posted.
//
// RCCL is bound at run time (dlopen): libamrdemod.so has no link-time dependency on it, and a process that already
// carries a copy (PyTorch ships one) keeps using that one.
// =====================================================================================================================
#include <mutex>

namespace {

struct Id128 { char b[128]; };   // ncclUniqueId (rccl.h: char internal[128]), passed by value

struct Rccl {
    void *so = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy already in the process first (torch's librccl.so), then the ROCm one
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *n : names) if (!r.so) r.so = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char *n : names) if (!r.so) r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.so) r.so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.so) return;
#define AMR_SYM(field, name) *(void **)(&r.field) = dlsym(r.so, name)
        AMR_SYM(GetUniqueId, "ncclGetUniqueId"); AMR_SYM(CommInitRank, "ncclCommInitRank"); AMR_SYM(CommDestroy, "ncclCommDestroy");
        AMR_SYM(CommCount, "ncclCommCount");
        AMR_SYM(GroupStart, "ncclGroupStart"); AMR_SYM(GroupEnd, "ncclGroupEnd"); AMR_SYM(Send, "ncclSend"); AMR_SYM(Recv, "ncclRecv");
        AMR_SYM(GetErrorString, "ncclGetErrorString");
#undef AMR_SYM
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv) r.so = nullptr;
    });
    return r.so ? &r : nullptr;
}

constexpr int kNcclUint8 = 1;              // ncclDataType_t: ncclInt8 0, ncclUint8 1 (rccl.h)

amr_status nccl_fail(const char *what, int rc)
{
    Rccl *r = rccl();
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, (r && r->GetErrorString) ? r->GetErrorString(rc) : "RCCL error");
    return fail(AMR_EHIP, buf);
}
#define NCCL_TRY(expr) do { int rc_ = (expr); if (rc_ != 0) return nccl_fail(#expr, rc_); } while (0)

// ---- the gather slot: ONE description of its layout, used by the device pack kernel, by amr_gather_pack_host (CPU
// hosts and the gloo tests) and by amr_gather_unpack / amr_gather_fetch --------------------------------------------------
//   u64 words [0] n_true  [1] n_sent = min(n_true, cap)  [2] n_pre  [3 .. 3+n_pre] per-preamble offsets into the
//   source rank's (untruncated) hit arrays  [12] gather sequence number  -- header of kGatherHdr words, then
//   n_sent call indices (u64), then n_sent idx (u32).
constexpr uint32_t kGatherHdr = AMR_GATHER_HEADER_BYTES / 8;
static_assert(3 + AMR_MAX_PREAMBLES + 1 <= 12 && kGatherHdr >= 13, "gather header layout");

// bytes of records that travel for n_sent of them: [n_sent call indices u64 | n_sent idx u32], rounded up to 4 KiB
__host__ __device__ inline size_t gather_wire_bytes(uint64_t n_sent)
{
    return ((size_t)n_sent * 12 + 4095) & ~(size_t)4095;
}

// Slots up to this size travel whole, in ONE message per rank and gather, and nobody waits for anybody (round 3's
// protocol): at 12 bytes per record that is a capacity of 21 000 validated hits -- what `bench.py --gpus N` and any
// deployment with amr_set_validation gather.  Only larger slots (raw hit lists: MBs) are worth the two phases, whose
// price is the root's wait for the headers.
constexpr size_t kGatherWholeSlotMax = 256 * 1024;
__host__ __device__ inline bool gather_two_phase(size_t slot_bytes) { return slot_bytes > kGatherWholeSlotMax; }

// a slot in memory: header + room for the wire bytes of `cap` records
__host__ __device__ inline size_t gather_slot_bytes(uint64_t cap)
{
    return ((size_t)kGatherHdr * 8 + gather_wire_bytes(cap) + 255) & ~(size_t)255;
}

// element i of `stride` workers: header words and records of a packed result [blk u64 x n | idx u32 x n | ...]
__host__ __device__ inline void gather_pack_part(const uint64_t *blk, const uint32_t *idx, const uint64_t *offs, uint32_t n_pre,
                                                 uint64_t cap, uint64_t seq, uint64_t *slot, uint64_t t, uint64_t stride)
{
    const uint64_t n = offs[n_pre], m = n < cap ? n : cap;
    uint64_t *rb = slot + kGatherHdr;
    uint32_t *ri = reinterpret_cast<uint32_t *>(rb + m);
    if (t == 0) { slot[0] = n; slot[1] = m; slot[2] = n_pre; slot[12] = seq; }
    for (uint64_t i = t; i <= n_pre; i += stride) slot[3 + i] = offs[i];
    for (uint64_t i = t; i < m; i += stride) { rb[i] = blk[i]; ri[i] = idx[i]; }
}

__global__ void k_gather_pack(const uint8_t *packed, const uint64_t *offs, uint32_t n_pre, uint64_t cap, uint64_t seq, uint64_t *slot)
{
    const uint64_t n = offs[n_pre];
    gather_pack_part(reinterpret_cast<const uint64_t *>(packed), reinterpret_cast<const uint32_t *>(packed + n * 8), offs, n_pre,
                     cap, seq, slot, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

// Root: headers (received contiguously, phase 1) and records (phase 2, in place behind each rank's header slot) of all
// ranks -> the pinned host mirror, laid out as slots again.  grid (x, world): the x blocks of rank p share its records.
// (d_hdr null: the slots arrived whole, every header sits in front of its records)
__global__ void k_gather_mirror(const uint8_t *d_hdr, const uint8_t *d_recv, uint8_t *h_recv, size_t slot_bytes)
{
    const uint32_t p = blockIdx.y;
    const uint4 *hdr = reinterpret_cast<const uint4 *>(d_hdr ? d_hdr + (size_t)p * kGatherHdr * 8 : d_recv + (size_t)p * slot_bytes);
    const uint64_t m = reinterpret_cast<const uint64_t *>(hdr)[1];
    uint4 *dst = reinterpret_cast<uint4 *>(h_recv + (size_t)p * slot_bytes);
    const uint4 *src = reinterpret_cast<const uint4 *>(d_recv + (size_t)p * slot_bytes);
    const uint64_t n16 = kGatherHdr * 8 / 16 + (m * 12 + 15) / 16;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] = i < kGatherHdr * 8 / 16 ? hdr[i] : src[i];
}

amr_status gather_unpack(const void *slot, size_t slot_bytes, amr_gathered *out)
{
    const uint64_t *hdr = reinterpret_cast<const uint64_t *>(slot);
    if (slot_bytes < (size_t)kGatherHdr * 8) return fail(AMR_EINVAL, "gather slot shorter than its header");
    if (hdr[2] > AMR_MAX_PREAMBLES || hdr[1] > hdr[0] || (size_t)kGatherHdr * 8 + hdr[1] * 12 > slot_bytes)
        return fail(AMR_EINVAL, "gather slot header inconsistent");
    out->n_true = hdr[0];
    out->n_hits = hdr[1];
    out->n_preambles = (uint32_t)hdr[2];
    out->seq = hdr[12];
    out->preamble_offset = hdr + 3;
    out->hit_block = hdr + kGatherHdr;
    out->hit_idx = reinterpret_cast<const uint32_t *>(hdr + kGatherHdr + hdr[1]);
    return AMR_OK;
}

}  // namespace

struct Comm {
    void *comm = nullptr;
    int rank = 0, world = 1, root = 0;
    uint64_t cap = 0;            // records a slot holds
    size_t slot_bytes = 0;
    hipStream_t stream = nullptr;
    uint8_t *d_send[2] = {nullptr, nullptr};
    uint8_t *d_recv[2] = {nullptr, nullptr};   // root: world slots each (records land behind each slot's header bytes)
    uint8_t *h_recv[2] = {nullptr, nullptr};   // root: pinned mirror of d_recv
    uint8_t *d_hdr[2] = {nullptr, nullptr};    // root: world headers, contiguous (phase 1)
    uint8_t *h_hdr[2] = {nullptr, nullptr};    // root: pinned copy of d_hdr -- the counts that size phase 2
    uint64_t *d_zero = nullptr;                // AMR_MAX_PREAMBLES + 1 zero offsets: the packed form of an empty result
    hipEvent_t ev_hdr = nullptr;               // root: the headers of the gather being posted are in h_hdr
    hipEvent_t ev_host[2] = {nullptr, nullptr};   // root: the mirror of set k has arrived
    uint64_t seq_of[2] = {~0ull, ~0ull};       // gather sequence number each set holds
    uint64_t next_seq = 0;
};

extern "C" {

size_t amr_gather_slot_bytes(uint64_t cap_hits) { return gather_slot_bytes(cap_hits); }
size_t amr_gather_wire_bytes(uint64_t n_sent) { return gather_wire_bytes(n_sent); }
int32_t amr_gather_two_phase(uint64_t cap_hits) { return gather_two_phase(gather_slot_bytes(cap_hits)) ? 1 : 0; }

amr_status amr_device_count(int32_t *n_devices)
{
    if (!n_devices) return fail(AMR_EINVAL, "null argument");
    *n_devices = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return AMR_OK;    // none: not an error, the count is the answer
    int n = 0;
    for (int d = 0; d < ndev; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++n;
    }
    *n_devices = n;
    return AMR_OK;
}

amr_status amr_gather_pack_host(const amr_result *res, uint64_t cap_hits, uint64_t seq, void *slot, size_t slot_bytes)
{
    if (!res || !slot || !res->preamble_offset || res->n_preambles > AMR_MAX_PREAMBLES) return fail(AMR_EINVAL, "null argument");
    if (slot_bytes < gather_slot_bytes(cap_hits)) return fail(AMR_EINVAL, "amr_gather_pack_host: slot too small for the capacity");
    if (res->preamble_offset[res->n_preambles] != res->n_hits) return fail(AMR_EINVAL, "amr_gather_pack_host: offsets do not end at n_hits");
    gather_pack_part(res->hit_block, res->hit_idx, res->preamble_offset, res->n_preambles, cap_hits, seq,
                     reinterpret_cast<uint64_t *>(slot), 0, 1);
    return AMR_OK;
}

amr_status amr_gather_unpack(const void *slot, size_t slot_bytes, amr_gathered *out)
{
    if (!slot || !out) return fail(AMR_EINVAL, "null argument");
    return gather_unpack(slot, slot_bytes, out);
}

amr_status amr_comm_unique_id(void *id128)
{
    if (!id128) return fail(AMR_EINVAL, "null argument");
    Rccl *r = rccl();
    if (!r) return fail(AMR_ENODEV, "RCCL (librccl.so) not found");
    NCCL_TRY(r->GetUniqueId(id128));
    return AMR_OK;
}

amr_status amr_comm_init(amr_handle *h, const void *id128, int32_t rank, int32_t world, int32_t root, uint64_t cap_hits)
{
    if (!h || !id128) return fail(AMR_EINVAL, "null argument");
    if (world < 1 || world > 65535 || rank < 0 || rank >= world || root < 0 || root >= world || cap_hits == 0) return fail(AMR_EINVAL, "amr_comm_init: bad rank / world / capacity");
    if (h->comm) return fail(AMR_EINVAL, "amr_comm_init: communicator exists already");
    Rccl *r = rccl();
    if (!r) return fail(AMR_ENODEV, "RCCL (librccl.so) not found");
    HIP_TRY(hipSetDevice(h->device));
    Comm *c = new (std::nothrow) Comm();
    if (!c) return fail(AMR_ENOMEM, "Comm");
    c->rank = rank; c->world = world; c->root = root; c->cap = cap_hits;
    c->slot_bytes = gather_slot_bytes(cap_hits);
    Id128 id;
    memcpy(id.b, id128, 128);
    int rc = r->CommInitRank(&c->comm, world, id, rank);
    if (rc != 0) { delete c; return nccl_fail("ncclCommInitRank", rc); }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipMalloc((void **)&c->d_send[k], c->slot_bytes);
        if (e == hipSuccess && rank == root) e = hipMalloc((void **)&c->d_recv[k], c->slot_bytes * (size_t)world);
        if (e == hipSuccess && rank == root) e = hipHostMalloc((void **)&c->h_recv[k], c->slot_bytes * (size_t)world, hipHostMallocDefault);
        if (e == hipSuccess && rank == root) e = hipEventCreateWithFlags(&c->ev_host[k], hipEventDisableTiming);
        if (e == hipSuccess && rank == root) e = hipMalloc((void **)&c->d_hdr[k], (size_t)world * kGatherHdr * 8);
        if (e == hipSuccess && rank == root) e = hipHostMalloc((void **)&c->h_hdr[k], (size_t)world * kGatherHdr * 8, hipHostMallocDefault);
    }
    if (e == hipSuccess && rank == root) e = hipEventCreateWithFlags(&c->ev_hdr, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_zero, (AMR_MAX_PREAMBLES + 1) * 8);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_zero, 0, (AMR_MAX_PREAMBLES + 1) * 8, c->stream);
    h->comm = c;
    if (e != hipSuccess) { (void)amr_comm_destroy(h); return fail(AMR_ENOMEM, "amr_comm_init: buffers", e); }
    return AMR_OK;
}

amr_status amr_comm_ranks(const amr_handle *h, int32_t *n_ranks)
{
    if (!h || !h->comm || !n_ranks) return fail(AMR_EINVAL, "amr_comm_ranks: amr_comm_init first");
    Rccl *r = rccl();
    if (!r || !r->CommCount) return fail(AMR_ENODEV, "ncclCommCount not available");
    int n = 0;
    NCCL_TRY(r->CommCount(h->comm->comm, &n));
    *n_ranks = n;
    return AMR_OK;
}

amr_status amr_comm_destroy(amr_handle *h)
{
    if (!h || !h->comm) return AMR_OK;
    Comm *c = h->comm;
    (void)hipSetDevice(h->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (Slot &sl : h->slot) sl.pack_pending = false;
    Rccl *r = rccl();
    if (r && c->comm) (void)r->CommDestroy(c->comm);
    for (int k = 0; k < 2; ++k) {
        if (c->d_send[k]) (void)hipFree(c->d_send[k]);
        if (c->d_recv[k]) (void)hipFree(c->d_recv[k]);
        if (c->h_recv[k]) (void)hipHostFree(c->h_recv[k]);
        if (c->ev_host[k]) (void)hipEventDestroy(c->ev_host[k]);
        if (c->d_hdr[k]) (void)hipFree(c->d_hdr[k]);
        if (c->h_hdr[k]) (void)hipHostFree(c->h_hdr[k]);
    }
    if (c->ev_hdr) (void)hipEventDestroy(c->ev_hdr);
    if (c->d_zero) (void)hipFree(c->d_zero);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    h->comm = nullptr;
    return AMR_OK;
}

amr_status amr_gather_hits(amr_handle *h, uint64_t *seq_out)
{
    if (!h || !h->comm) return fail(AMR_EINVAL, "amr_gather_hits: amr_comm_init first");
    if (h->last_slot < 0 && !h->last_empty) return fail(AMR_EINVAL, "amr_gather_hits: no batch collected yet");
    Rccl *r = rccl();
    Comm *c = h->comm;
    HIP_TRY(hipSetDevice(h->device));
    // the result amr_collect / amr_flush returned last; an amr_flush with nothing deferred returned an EMPTY one: zero
    // records travel (the slot of the batch before it still holds that batch's hits)
    const bool empty = h->last_empty;
    Slot *s = empty ? nullptr : &h->slot[h->last_slot];
    const uint8_t *packed = empty ? reinterpret_cast<const uint8_t *>(c->d_zero) : (h->validate ? s->d_val : s->d_out);
    const uint64_t *offs = empty ? c->d_zero : (h->validate ? s->d_offs_val : s->d_offs_pre);
    const uint64_t n_host = empty ? 0 : h->last_total;                 // = offs[n_pre] on the device
    const uint64_t m_host = n_host < c->cap ? n_host : c->cap;         // records this rank sends
    const uint64_t seq = c->next_seq++;
    const int k = (int)(seq & 1);
    // on the communicator's stream: behind the sends (and the root's mirror kernel) that last used buffer set k
    hipLaunchKernelGGL(k_gather_pack, dim3(64), dim3(256), 0, c->stream, packed, offs, h->sg.n_pre, c->cap, seq,
                       reinterpret_cast<uint64_t *>(c->d_send[k]));
    HIP_TRY(hipGetLastError());
    if (s) {   // whoever overwrites this slot's result next waits for the pack kernel (enqueue_tail)
        HIP_TRY(hipEventRecord(s->ev_pack, c->stream));
        s->pack_pending = true;
    }
    const size_t hdr_bytes = (size_t)kGatherHdr * 8;
    if (!gather_two_phase(c->slot_bytes)) {
        // ---- small slots: the whole slot in one message, no host wait anywhere ----
        NCCL_TRY(r->GroupStart());
        NCCL_TRY(r->Send(c->d_send[k], c->slot_bytes, kNcclUint8, c->root, c->comm, c->stream));
        if (c->rank == c->root)
            for (int p = 0; p < c->world; ++p)
                NCCL_TRY(r->Recv(c->d_recv[k] + (size_t)p * c->slot_bytes, c->slot_bytes, kNcclUint8, p, c->comm, c->stream));
        NCCL_TRY(r->GroupEnd());
        if (c->rank == c->root) {
            hipLaunchKernelGGL(k_gather_mirror, dim3(8, (unsigned)c->world), dim3(256), 0, c->stream, (const uint8_t *)nullptr, c->d_recv[k], c->h_recv[k], c->slot_bytes);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(c->ev_host[k], c->stream));
        }
        c->seq_of[k] = seq;
        if (seq_out) *seq_out = seq;
        return AMR_OK;
    }
    // ---- phase 1: the headers ----
    NCCL_TRY(r->GroupStart());
    NCCL_TRY(r->Send(c->d_send[k], hdr_bytes, kNcclUint8, c->root, c->comm, c->stream));
    if (c->rank == c->root)
        for (int p = 0; p < c->world; ++p)
            NCCL_TRY(r->Recv(c->d_hdr[k] + (size_t)p * hdr_bytes, hdr_bytes, kNcclUint8, p, c->comm, c->stream));
    NCCL_TRY(r->GroupEnd());
    // ---- phase 2: the records, sized by their count ----
    if (c->rank != c->root) {
        if (m_host) NCCL_TRY(r->Send(c->d_send[k] + hdr_bytes, gather_wire_bytes(m_host), kNcclUint8, c->root, c->comm, c->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(c->h_hdr[k], c->d_hdr[k], (size_t)c->world * hdr_bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(c->ev_hdr, c->stream));
        HIP_TRY(hipEventSynchronize(c->ev_hdr));        // every rank has posted this gather; 128 bytes each
        const uint64_t *hh = reinterpret_cast<const uint64_t *>(c->h_hdr[k]);
        for (int p = 0; p < c->world; ++p) {
            const uint64_t *hp = hh + (size_t)p * kGatherHdr;
            if (hp[1] > c->cap || hp[1] > hp[0] || hp[12] != seq)
                return fail(AMR_EHIP, "amr_gather_hits: a rank's header is inconsistent (ranks out of step, or capacities differ)");
        }
        NCCL_TRY(r->GroupStart());
        if (m_host) NCCL_TRY(r->Send(c->d_send[k] + hdr_bytes, gather_wire_bytes(m_host), kNcclUint8, c->root, c->comm, c->stream));
        for (int p = 0; p < c->world; ++p) {
            const uint64_t m_p = hh[(size_t)p * kGatherHdr + 1];
            if (m_p) NCCL_TRY(r->Recv(c->d_recv[k] + (size_t)p * c->slot_bytes + hdr_bytes, gather_wire_bytes(m_p), kNcclUint8, p, c->comm, c->stream));
        }
        NCCL_TRY(r->GroupEnd());
        hipLaunchKernelGGL(k_gather_mirror, dim3(8, (unsigned)c->world), dim3(256), 0, c->stream, c->d_hdr[k], c->d_recv[k], c->h_recv[k], c->slot_bytes);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev_host[k], c->stream));
    }
    c->seq_of[k] = seq;
    if (seq_out) *seq_out = seq;
    return AMR_OK;
}

amr_status amr_gather_wait(amr_handle *h)
{
    if (!h || !h->comm) return fail(AMR_EINVAL, "amr_gather_wait: amr_comm_init first");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->comm->stream));
    return AMR_OK;
}

amr_status amr_gather_fetch(amr_handle *h, uint64_t seq, int32_t src_rank, amr_gathered *out)
{
    if (!h || !h->comm || !out) return fail(AMR_EINVAL, "amr_gather_fetch: null argument / no communicator");
    Comm *c = h->comm;
    if (c->rank != c->root) return fail(AMR_EINVAL, "amr_gather_fetch: only the root holds the gathered records");
    if (src_rank < 0 || src_rank >= c->world) return fail(AMR_EINVAL, "amr_gather_fetch: bad rank");
    const int k = (int)(seq & 1);
    if (c->seq_of[k] != seq) return fail(AMR_EINVAL, "amr_gather_fetch: that gather was never posted or its records have been overwritten (two sets)");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventSynchronize(c->ev_host[k]));      // the mirror copy of this gather, nothing else
    AMR_TRY(gather_unpack(c->h_recv[k] + (size_t)src_rank * c->slot_bytes, c->slot_bytes, out));
    if (out->seq != seq) return fail(AMR_EHIP, "amr_gather_fetch: a rank's slot carries another gather's sequence number (ranks out of step)");
    return AMR_OK;
}

}  // extern "C"
