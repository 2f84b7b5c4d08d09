// amr_host.h -- what the host-side translation units of libamrdemod.so share: status helpers, the per-batch Slot, the
// handle, and the few functions that cross from one unit to another.  Internal: nothing here is part of the C ABI
// (include/amrdemod.h).
//   amrdemod.hip      lifecycle and configuration: RegisterProtocol / Allocate arithmetic, amr_create .. amr_destroy, setters
//   amr_pipeline.hip  the batch pipeline: submit / collect over four slots and three streams, the hot-path entry points
//   amr_gather.hip    multi-GPU: the RCCL gather of hit records (amr_comm_*, amr_gather_*)
//   amr_util.hip      device utilities and the synthetic IQ generator for bench and tests
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "amrdemod.h"
#include "k2_common.h"
#include "k5_validate.h"

namespace amr_host {

// AMR_DEBUG_SYNC=1: synchronise and log after every kernel (localises a faulting kernel)
inline bool debug_sync() { static const bool on = getenv("AMR_DEBUG_SYNC") != nullptr; return on; }
#define AMR_DBG(st, what)                                                                  \
    do {                                                                                   \
        if (debug_sync()) { fprintf(stderr, "[amr] %s ...", what); fflush(stderr);         \
            hipError_t e_ = hipStreamSynchronize(st); fprintf(stderr, " %s\n", hipGetErrorString(e_)); } \
    } while (0)

// records the text amr_last_error() returns (thread-local) and hands the status back
amr_status fail(amr_status s, const char *what, hipError_t e = hipSuccess);

#define HIP_TRY(expr)                                            \
    do {                                                         \
        hipError_t e_ = (expr);                                  \
        if (e_ != hipSuccess) return fail(AMR_EHIP, #expr, e_);  \
    } while (0)

#define AMR_TRY(expr)                          \
    do {                                       \
        amr_status s_ = (expr);                \
        if (s_ != AMR_OK) return s_;           \
    } while (0)

}  // namespace amr_host

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
    bool inwave = false;          // K1 searched the batch's tiles itself (k1_search.h): the search launch is the clean-up
    bool tail_gated = false;      // ... enqueued ahead, behind k_gate: K3 then runs next to the following K1's end and search
    bool folded = false;          // the state update ran inside the search kernel: no stream-A ticket for this batch
    int carry_in_slot = 0;        // k3_stale.h: the slot whose carry byte precedes this batch's first hit
    bool force_rerun = false;     // ... and "search this batch's tail again": an older batch's re-search changed that byte
    bool early = false;           // the batch's search ran on the search stream, next to its K1 (early search)
    uint32_t *d_k1flags = nullptr;   // [cnt_tiles] one "done" word per K1 wave-tile: the batch ticket when its rows are in memory
    // stop event of this batch's second-to-last K1 round (batches of several launches): the PREVIOUS batch's gate kernel comes
    // onto the chip behind it (amr_pipeline.hip, submit)
    hipEvent_t ev_gate = nullptr;
    hipEvent_t ev_k2done = nullptr;  // early search: recorded behind K2 on the search stream (K3 on the tail stream waits for it)
    bool single = false;          // the batch was one block through the one-launch path (k1_single.h): h_out holds its result already
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
    // early search: K2 of a batch next to its K1, tile by tile
    hipStream_t search_stream = nullptr;
    int early_mode = -1;              // -1: where it was measured to pay (BlockSize <= 512: K1 runs several rounds in one launch
                                      // there, out of step anyway); AMR_EARLY_SEARCH=1: wherever it applies; 0: never
    bool k2_on_search = false;        // the last batch's search ran on the search stream
    hipEvent_t ev_switch = nullptr, ev_last_early = nullptr;   // hand-over between the two orders (ev_last_early: a slot's ev_k2done)
    uint8_t *d_carry_alt = nullptr, *d_carry_cur = nullptr;    // the IQ halo of the next batch's block 0: in d_head or here
    // The host launches the tail when it sees the next batch's search start (a pinned flag; no event on the compute
    // stream: stream dependencies were tried and cost ~10 us of bubbles per batch, cfg2 0.237 ms per step against 0.227).
    bool lazy_tail = false;
    uint64_t *d_tail_done = nullptr;   // device word: ticket of the last batch whose second-stream part has finished
    uint64_t *d_k1_started = nullptr;  // device word: ticket of the last batch whose K1 has all its waves on the chip (k_gate)
    uint32_t *d_k1_ctr = nullptr;      // K1Args::started_ctr
    uint32_t k1_ctr_total = 0;         // what the announcing launches so far add up to (wraps with the device word)
    uint64_t gate_timeout_ticks = 400000000ull;   // k_gate gives up after this many 100 MHz ticks (4 s; test hook AMR_GATE_TIMEOUT_US)
    uint64_t gate_timeouts = 0;   // batches searched again because their gate gave up (amr_describe)
    uint64_t researches = 0;      // batches searched again at collect, for any reason (amr_describe)
    uint64_t inwave_batches = 0;  // batches whose tiles the K1 waves searched themselves (k1_search.h; amr_describe)
    uint64_t stale_reruns = 0;    // batches whose k_stale_bits pass alone ran again behind an older batch's re-search
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
    // the tail's gate kernel (submit): ticks of the 100 MHz clock it stays after it has seen the K1 launch's last workgroup
    // start, for the other XCDs' dispatchers to place theirs (hook AMR_GATE_DELAY_TICKS); whether, in batches of several K1
    // launches, it comes onto the chip behind the stop event of the round before the last (AMR_GATE_EVENT=0: as in round 4,
    // as soon as the tail stream reaches it)
    uint32_t gate_delay_ticks = 600;
    bool gate_event = true;
    uint32_t k3_prio = 0;        // hook AMR_K3_PRIO: s_setprio level of K3's waves (0..3)
    int inwave_mode = 1;         // AMR_INWAVE: 0 never search inside the K1 wave (BlockSize 512 then runs the early search), 1 rows of 16 words
                                 // (chip 8), 2 rows of 64 words as well (chip 32 / 40, scm: experiment)
    size_t k3_lds_min = 0;       // hook AMR_K3_LDS_KB: dynamic LDS of K3 at least this (bytes)
    size_t k2w_lds_min = 0;      // hook AMR_K2W_LDS_KB: dynamic LDS of the multi-preamble walk at least this (bytes)
    int gate_end_mode = 0;       // A/B hook AMR_GATE_END (round 6, lost: profiles/r06/bs2048/): the tail behind the END of a one-launch K1 instead of
                                 // behind a gate: 1 every one-launch batch, -1 those with more wave-tiles than the chip has slots, 0 never
    uint32_t k1_round_tiles = 0; // test hook AMR_K1_ROUND_TILES: wave-tiles per K1 launch (0: a chip's worth at BlockSize >= 4096, else one launch)
    uint64_t k1_coop_max = 0;    // batches of up to this many blocks run K1 as one wave per block throughout (k1_coop.h):
                                 // from kK1CoopMaxSamples / kK1CoopMaxBlocks; test hook AMR_K1_COOP_MAX (0: only the blocks
                                 // behind the last whole wave-tile -- keeps the tile kernels under the small-batch tests)
    uint64_t init_hit_cap = 1 << 16;   // hits the result buffers hold at first (test hook AMR_HIT_CAP: exercise the growth)
    int dense_streak = 0;        // consecutive batches whose sparse lists overflowed; >= 4: stay dense for a while
    int dense_hold = 0;          // batches left in which the dense kernel is used straight away
    uint8_t *d_iq = nullptr;      size_t iq_cap = 0;       // staging for host input
    uint8_t *d_pkt_carry = nullptr;   // [kSlots + 1] k3_stale.h: final last packet byte of each slot's last hit (PacketSymbols % 8 != 0)
    int carry_slot = 4;               // the slot of the last batch that was searched; kSlots: none yet (a byte of its own, zero: a
                                      // batch must never read and write the same byte -- its first and its last hit do so at once)
    uint8_t *h_iq1 = nullptr;    // pinned: the block of a one-block amr_decode_batch (read by k_single_block over the link)
    bool no_single = false;      // test hook AMR_NO_SINGLE: one-block calls take the regular kernels
    unsigned long long *d_single_dbg = nullptr;   // AMR_SINGLE_DBG: 8 phase time stamps of the last k_single_block (pinned)
    bool single_ready = false;   // k_single_block's dynamic LDS limit has been raised
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

namespace amr_host {

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

// ---- amr_pipeline.hip ----
// Wait for the compute stream.  With K3.. of some batches still unlaunched (pipelined callers), a k_hist_update on the
// stream may be waiting for one of them: launch them first (each as soon as its own search has finished), or the wait
// would only end at that kernel's 2 ms time-out.
amr_status sync_compute(amr_handle *h);
// collect every batch in flight, dropping the results
amr_status drain(amr_handle *h);
// phases of the last K3 launch / the gates' shader clock: diagnostic builds only (AMR_K3_DBG, AMR_GATE_CLK), from amr_destroy
void dump_diagnostics(amr_handle *h);

// ---- amrdemod.hip ----
// RegisterProtocol for every entry + the arithmetic of Allocate (decode.go:100-141): no device involved.
amr_status plan_geometry(const amr_protocol *protos, int32_t n_protos, amr_geometry &g, amr::SearchGeom &sg,
                         std::vector<int> &proto_pid, uint32_t &halo_bytes, uint32_t &hist_rows);

}  // namespace amr_host
