// amr_gather.hip -- multi-GPU: the RCCL gather of hit records (include/amrdemod.h, amr_comm_* / amr_gather_*).
#include <dlfcn.h>

#include "amr_host.h"

using namespace amr_host;

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
#include <algorithm>
#include <deque>
#include <mutex>
#include <vector>

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

// ---------------------------------------------------------------------------------------------------------------------
// TEST HOOK (amr_comm_test_loopback): an in-process stand-in for the RCCL entry points above, so that the logic AROUND
// the transport -- the root's n receives, the two-phase header wait, truncation, the inconsistent-header path, the
// mirror kernel -- runs with n = 2, 3 ranks on a ONE-GPU box (first contact with n > 1 must not be the
// 8-GPU lease).  Semantics kept: point-to-point, FIFO matching per (source, destination) pair, a send completes on its
// stream only after the matching receive has been posted, sizes must match (a mismatch is an error, like RCCL's), calls
// inside ncclGroupStart/End take effect at the outermost ncclGroupEnd.  A matched pair is one hipMemcpyAsync on the
// receiver's stream between two cross-stream events.  Every "rank" may sit on the same device: nothing else is relaxed.
namespace loop {
struct World;
struct LComm { World *w; int rank; };
struct Op { bool send; void *buf; size_t bytes; int peer; LComm *c; hipStream_t st; };
struct World { Id128 id; int n; std::deque<Op> sends, recvs; int members = 0; };
std::mutex mu;
std::vector<World *> worlds;
std::vector<Op> group_ops;
int depth = 0;
uint64_t next_id = 1;
uint64_t mismatches = 0;

int get_unique_id(void *out)
{
    std::lock_guard<std::mutex> lk(mu);
    memset(out, 0, 128);
    memcpy(out, &next_id, 8);
    memcpy((char *)out + 8, "amr-loopback", 12);
    ++next_id;
    return 0;
}
int comm_init_rank(void **comm, int n, Id128 id, int rank)
{
    std::lock_guard<std::mutex> lk(mu);
    World *w = nullptr;
    for (World *x : worlds) if (memcmp(x->id.b, id.b, 128) == 0) w = x;
    if (!w) { w = new World(); w->id = id; w->n = n; worlds.push_back(w); }
    if (w->n != n || rank < 0 || rank >= n) return 4;   // ncclInvalidArgument
    w->members++;
    *comm = new LComm{w, rank};
    return 0;
}
int comm_destroy(void *comm)
{
    std::lock_guard<std::mutex> lk(mu);
    LComm *c = (LComm *)comm;
    if (--c->w->members == 0) {
        worlds.erase(std::remove(worlds.begin(), worlds.end(), c->w), worlds.end());
        delete c->w;
    }
    delete c;
    return 0;
}
int comm_count(void *comm, int *n) { *n = ((LComm *)comm)->w->n; return 0; }
// match what can be matched: receives in posting order, each with the oldest send of its (source -> destination) pair
int match(World *w)
{
    int rc = 0;
    for (auto r = w->recvs.begin(); r != w->recvs.end();) {
        auto s = w->sends.begin();
        for (; s != w->sends.end(); ++s) if (s->c->rank == r->peer && s->peer == r->c->rank) break;
        if (s == w->sends.end()) { ++r; continue; }
        size_t bytes = s->bytes;
        if (s->bytes != r->bytes) { ++mismatches; rc = 4; bytes = std::min(s->bytes, r->bytes); }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const bool cross = s->st != r->st;
        if (cross) {
            if (hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) return 1;
            if (hipEventRecord(e0, s->st) != hipSuccess || hipStreamWaitEvent(r->st, e0, 0) != hipSuccess) return 1;
        }
        if (bytes && hipMemcpyAsync(r->buf, s->buf, bytes, hipMemcpyDeviceToDevice, r->st) != hipSuccess) return 1;
        if (cross) {   // the sender's stream goes on (and may reuse the buffer) only behind the copy
            if (hipEventRecord(e1, r->st) != hipSuccess || hipStreamWaitEvent(s->st, e1, 0) != hipSuccess) return 1;
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
        w->sends.erase(s);
        r = w->recvs.erase(r);
    }
    return rc;
}
int flush()
{
    int rc = 0;
    std::vector<World *> touched;
    for (const Op &o : group_ops) {
        (o.send ? o.c->w->sends : o.c->w->recvs).push_back(o);
        if (std::find(touched.begin(), touched.end(), o.c->w) == touched.end()) touched.push_back(o.c->w);
    }
    group_ops.clear();
    for (World *w : touched) { const int r = match(w); if (r) rc = r; }
    return rc;
}
int group_start() { std::lock_guard<std::mutex> lk(mu); ++depth; return 0; }
int group_end() { std::lock_guard<std::mutex> lk(mu); if (depth > 0 && --depth == 0) return flush(); return 0; }
int post(bool send, void *buf, size_t bytes, int peer, void *comm, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(mu);
    LComm *c = (LComm *)comm;
    if (peer < 0 || peer >= c->w->n) return 4;
    group_ops.push_back(Op{send, buf, bytes, peer, c, st});
    return depth == 0 ? flush() : 0;
}
int send(const void *buf, size_t n, int, int peer, void *comm, hipStream_t st) { return post(true, const_cast<void *>(buf), n, peer, comm, st); }
int recv(void *buf, size_t n, int, int peer, void *comm, hipStream_t st) { return post(false, buf, n, peer, comm, st); }
const char *error_string(int rc) { return rc == 4 ? "loopback transport: invalid argument / send and receive sizes differ" : "loopback transport: HIP call failed"; }
bool enabled = false;
}  // namespace loop

Rccl *rccl()
{
    if (loop::enabled) {
        static Rccl l;
        l.so = &l;
        l.GetUniqueId = loop::get_unique_id; l.CommInitRank = loop::comm_init_rank; l.CommDestroy = loop::comm_destroy;
        l.CommCount = loop::comm_count; l.GroupStart = loop::group_start; l.GroupEnd = loop::group_end;
        l.Send = loop::send; l.Recv = loop::recv; l.GetErrorString = loop::error_string;
        return &l;
    }
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
// `cap`: records a slot holds -- a header that claims more (a rank out of step, another capacity) must not make this kernel
// write past the slot in the mirror: the count is clamped, amr_gather_fetch rejects the header itself.
__global__ void k_gather_mirror(const uint8_t *d_hdr, const uint8_t *d_recv, uint8_t *h_recv, size_t slot_bytes, uint64_t cap)
{
    const uint32_t p = blockIdx.y;
    const uint4 *hdr = reinterpret_cast<const uint4 *>(d_hdr ? d_hdr + (size_t)p * kGatherHdr * 8 : d_recv + (size_t)p * slot_bytes);
    const uint64_t m_hdr = reinterpret_cast<const uint64_t *>(hdr)[1];
    const uint64_t m = m_hdr < cap ? m_hdr : cap;
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
    bool local_world = false;                  // every rank of the communicator lives in THIS process (amr_comm_init_all, n > 1)
    bool failed[2] = {false, false};           // root: that gather's headers did not fit; amr_gather_fetch refuses it
    uint64_t next_seq = 0;
};

extern "C" {

size_t amr_gather_slot_bytes(uint64_t cap_hits) { return gather_slot_bytes(cap_hits); }
size_t amr_gather_wire_bytes(uint64_t n_sent) { return gather_wire_bytes(n_sent); }
int32_t amr_gather_two_phase(uint64_t cap_hits) { return gather_two_phase(gather_slot_bytes(cap_hits)) ? 1 : 0; }


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

amr_status amr_comm_test_loopback(int32_t enable)
{
    loop::enabled = enable != 0;
    return AMR_OK;
}

amr_status amr_comm_unique_id(void *id128)
{
    if (!id128) return fail(AMR_EINVAL, "null argument");
    Rccl *r = rccl();
    if (!r) return fail(AMR_ENODEV, "RCCL (librccl.so) not found");
    NCCL_TRY(r->GetUniqueId(id128));
    return AMR_OK;
}

// the buffers of a communicator that exists (c->comm, rank, world, root, cap set); on failure the caller destroys h->comm
static amr_status comm_alloc(amr_handle *h, Comm *c)
{
    const int rank = c->rank, root = c->root, world = c->world;
    c->slot_bytes = gather_slot_bytes(c->cap);
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
    Id128 id;
    memcpy(id.b, id128, 128);
    int rc = r->CommInitRank(&c->comm, world, id, rank);
    if (rc != 0) { delete c; return nccl_fail("ncclCommInitRank", rc); }
    return comm_alloc(h, c);
}

// The argument rules of the single-process form, checked before any device or RCCL call (tests drive this on CPU).
amr_status amr_comm_check_all(amr_handle *const *hs, const int32_t *devices, int32_t n, int32_t root, uint64_t cap_hits)
{
    if (n < 1 || n > 65535 || root < 0 || root >= n || cap_hits == 0) return fail(AMR_EINVAL, "amr_comm_init_all: bad handle count / root / capacity");
    if (!hs && !devices) return fail(AMR_EINVAL, "amr_comm_init_all: null argument");
    for (int32_t i = 0; i < n; ++i) {
        if (hs && !hs[i]) return fail(AMR_EINVAL, "amr_comm_init_all: null handle");
        const int32_t di = devices ? devices[i] : hs[i]->device;
        if (di < 0) return fail(AMR_EINVAL, "amr_comm_init_all: negative device ordinal");
        for (int32_t j = 0; j < i; ++j) {
            if (hs && hs[j] == hs[i]) return fail(AMR_EINVAL, "amr_comm_init_all: the same handle twice");
            // one rank per device: two ranks of one communicator on one GPU deadlock in RCCL's point-to-point kernels
            if (!loop::enabled && (devices ? devices[j] : hs[j]->device) == di) return fail(AMR_EINVAL, "amr_comm_init_all: two handles on one device");
        }
        if (hs && hs[i]->comm) return fail(AMR_EINVAL, "amr_comm_init_all: a handle has a communicator already");
    }
    return AMR_OK;
}

amr_status amr_comm_init_all(amr_handle **hs, int32_t n, int32_t root, uint64_t cap_hits)
{
    if (!hs) return fail(AMR_EINVAL, "amr_comm_init_all: null argument");
    AMR_TRY(amr_comm_check_all(hs, nullptr, n, root, cap_hits));
    Rccl *r = rccl();
    if (!r) return fail(AMR_ENODEV, "RCCL (librccl.so) not found");
    Id128 id;
    NCCL_TRY(r->GetUniqueId(&id));
    std::vector<Comm *> cs((size_t)n, nullptr);
    for (int32_t i = 0; i < n; ++i) {
        cs[(size_t)i] = new (std::nothrow) Comm();
        if (!cs[(size_t)i]) { for (Comm *c : cs) delete c; return fail(AMR_ENOMEM, "Comm"); }
        cs[(size_t)i]->rank = i; cs[(size_t)i]->world = n; cs[(size_t)i]->root = root; cs[(size_t)i]->cap = cap_hits;
        cs[(size_t)i]->local_world = n > 1;
    }
    // ONE thread initialises every rank: the calls must sit in one group, or the first ncclCommInitRank waits for ever
    // for peers this thread has not got round to yet (the reference caller is one process, main.go:59-128)
    int rc = r->GroupStart();
    for (int32_t i = 0; i < n && rc == 0; ++i) {
        if (hipSetDevice(hs[i]->device) != hipSuccess) { rc = -1; break; }
        rc = r->CommInitRank(&cs[(size_t)i]->comm, n, id, i);
    }
    const int rc_end = r->GroupEnd();
    if (rc == 0) rc = rc_end;
    if (rc != 0) {
        for (Comm *c : cs) { if (c->comm) (void)r->CommDestroy(c->comm); delete c; }
        return rc == -1 ? fail(AMR_EHIP, "amr_comm_init_all: hipSetDevice") : nccl_fail("ncclCommInitRank (group)", rc);
    }
    for (int32_t i = 0; i < n; ++i) {
        hipError_t e = hipSetDevice(hs[i]->device);
        amr_status st = e == hipSuccess ? comm_alloc(hs[i], cs[(size_t)i]) : fail(AMR_EHIP, "hipSetDevice", e);
        if (st != AMR_OK) {
            if (e != hipSuccess) { (void)r->CommDestroy(cs[(size_t)i]->comm); delete cs[(size_t)i]; }
            for (int32_t j = 0; j < i; ++j) (void)amr_comm_destroy(hs[j]);
            for (int32_t j = i + 1; j < n; ++j) { (void)r->CommDestroy(cs[(size_t)j]->comm); delete cs[(size_t)j]; }
            return st;
        }
    }
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

}  // extern "C"

namespace {

struct GatherStep { uint64_t seq = 0, m_host = 0; int k = 0; };

// One gather for the n handles of `hs` -- n == 1: the rank of a one-process-per-GPU job; n > 1: every rank of a
// communicator made by amr_comm_init_all, driven by ONE thread, which therefore must post the matching sends and
// receives of all ranks inside one RCCL group (a lone ncclRecv on the root may wait on the host for a peer this thread
// has not served yet).  Same wire protocol in either case.
amr_status gather_many(amr_handle *const *hs, int n, uint64_t *seq_out)
{
    Rccl *r = rccl();
    std::vector<GatherStep> g((size_t)n);
    const size_t hdr_bytes = (size_t)kGatherHdr * 8;
    amr_handle *root_h = nullptr;
    // nothing is touched before every handle is known to be in step (a refusal half way through left the
    // sequence numbers of the first handles advanced for good)
    for (int i = 1; i < n; ++i)
        if (hs[i]->comm->next_seq != hs[0]->comm->next_seq) return fail(AMR_EINVAL, "amr_gather_hits_all: the handles' gathers are out of step");
    // an RCCL group that was opened is closed on every path: an error inside it would otherwise leave this thread inside an
    // open group, and every later RCCL call of the process would be queued into it
    struct Group {
        Rccl *r; bool open = false;
        explicit Group(Rccl *rr) : r(rr) {}
        int start() { const int rc = r->GroupStart(); open = rc == 0; return rc; }
        int end() { open = false; return r->GroupEnd(); }
        ~Group() { if (open) (void)r->GroupEnd(); }
    };
    // ---- every rank: the pack kernel on its communicator's stream, behind the sends (and the root's mirror kernel) that
    // last used buffer set k ----
    for (int i = 0; i < n; ++i) {
        amr_handle *h = hs[i];
        Comm *c = h->comm;
        HIP_TRY(hipSetDevice(h->device));
        // the result amr_collect / amr_flush returned last; an amr_flush with nothing deferred returned an EMPTY one: zero
        // records travel (the slot of the batch before it still holds that batch's hits)
        const bool empty = h->last_empty;
        Slot *s = empty ? nullptr : &h->slot[h->last_slot];
        const uint8_t *packed = empty ? reinterpret_cast<const uint8_t *>(c->d_zero) : (h->validate ? s->d_val : s->d_out);
        const uint64_t *offs = empty ? c->d_zero : (h->validate ? s->d_offs_val : s->d_offs_pre);
        const uint64_t n_host = empty ? 0 : h->last_total;                 // = offs[n_pre] on the device
        g[(size_t)i].m_host = n_host < c->cap ? n_host : c->cap;           // records this rank sends
        g[(size_t)i].seq = c->next_seq++;
        g[(size_t)i].k = (int)(g[(size_t)i].seq & 1);
        c->failed[g[(size_t)i].k] = false;
        hipLaunchKernelGGL(k_gather_pack, dim3(64), dim3(256), 0, c->stream, packed, offs, h->sg.n_pre, c->cap, g[(size_t)i].seq,
                           reinterpret_cast<uint64_t *>(c->d_send[g[(size_t)i].k]));
        HIP_TRY(hipGetLastError());
        if (s) {   // whoever overwrites this slot's result next waits for the pack kernel (enqueue_tail)
            HIP_TRY(hipEventRecord(s->ev_pack, c->stream));
            s->pack_pending = true;
        }
        if (c->rank == c->root) root_h = h;
    }
    const uint64_t seq = g[0].seq;
    const int k = g[0].k;
    Comm *rc = root_h ? root_h->comm : nullptr;                            // the root's communicator, when the root is ours
    auto finish_root = [&](const uint8_t *d_hdr) -> amr_status {
        HIP_TRY(hipSetDevice(root_h->device));
        hipLaunchKernelGGL(k_gather_mirror, dim3(8, (unsigned)rc->world), dim3(256), 0, rc->stream, d_hdr, rc->d_recv[k], rc->h_recv[k], rc->slot_bytes, rc->cap);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(rc->ev_host[k], rc->stream));
        return AMR_OK;
    };
    if (!gather_two_phase(hs[0]->comm->slot_bytes)) {
        // ---- small slots: the whole slot in one message, no host wait anywhere ----
        Group grp(r);
        NCCL_TRY(grp.start());
        for (int i = 0; i < n; ++i) {
            Comm *c = hs[i]->comm;
            HIP_TRY(hipSetDevice(hs[i]->device));
            NCCL_TRY(r->Send(c->d_send[k], c->slot_bytes, kNcclUint8, c->root, c->comm, c->stream));
            if (c->rank == c->root)
                for (int p = 0; p < c->world; ++p)
                    NCCL_TRY(r->Recv(c->d_recv[k] + (size_t)p * c->slot_bytes, c->slot_bytes, kNcclUint8, p, c->comm, c->stream));
        }
        NCCL_TRY(grp.end());
        if (rc) AMR_TRY(finish_root(nullptr));
        for (int i = 0; i < n; ++i) hs[i]->comm->seq_of[k] = seq;
        if (seq_out) *seq_out = seq;
        return AMR_OK;
    }
    // ---- phase 1: the headers ----
    Group grp(r);
    NCCL_TRY(grp.start());
    for (int i = 0; i < n; ++i) {
        Comm *c = hs[i]->comm;
        HIP_TRY(hipSetDevice(hs[i]->device));
        NCCL_TRY(r->Send(c->d_send[k], hdr_bytes, kNcclUint8, c->root, c->comm, c->stream));
        if (c->rank == c->root)
            for (int p = 0; p < c->world; ++p)
                NCCL_TRY(r->Recv(c->d_hdr[k] + (size_t)p * hdr_bytes, hdr_bytes, kNcclUint8, p, c->comm, c->stream));
    }
    NCCL_TRY(grp.end());
    // ---- phase 2: the records, sized by their count.  A rank that is not the root knows its count on the host and never
    // waits; the root needs every peer's count before it can post its receives (RCCL point-to-point wants matching
    // sizes) and waits for the 128-byte headers -- with all ranks in one thread, that wait comes first for everybody ----
    bool consistent = true;
    const uint64_t *hh = nullptr;
    if (rc) {
        HIP_TRY(hipSetDevice(root_h->device));
        HIP_TRY(hipMemcpyAsync(rc->h_hdr[k], rc->d_hdr[k], (size_t)rc->world * hdr_bytes, hipMemcpyDeviceToHost, rc->stream));
        HIP_TRY(hipEventRecord(rc->ev_hdr, rc->stream));
        HIP_TRY(hipEventSynchronize(rc->ev_hdr));        // every rank has posted this gather; 128 bytes each
        hh = reinterpret_cast<const uint64_t *>(rc->h_hdr[k]);
        // A header that does not fit (a rank out of step, or capacities that differ) fails this gather -- but only AFTER the
        // receives of phase 2 have been posted: every peer with records has its send enqueued already and would otherwise
        // block on the communicator's stream for ever.  The sizes it advertised are taken at their word up to
        // the slot's capacity (a sender cannot have more in its own slot); amr_gather_fetch refuses the gather's records.
        for (int p = 0; p < rc->world; ++p) {
            const uint64_t *hp = hh + (size_t)p * kGatherHdr;
            if (hp[1] > rc->cap || hp[1] > hp[0] || hp[12] != seq) consistent = false;
        }
    }
    NCCL_TRY(grp.start());
    for (int i = 0; i < n; ++i) {
        Comm *c = hs[i]->comm;
        HIP_TRY(hipSetDevice(hs[i]->device));
        if (g[(size_t)i].m_host)
            NCCL_TRY(r->Send(c->d_send[k] + hdr_bytes, gather_wire_bytes(g[(size_t)i].m_host), kNcclUint8, c->root, c->comm, c->stream));
        if (c->rank == c->root)
            for (int p = 0; p < c->world; ++p) {
                const uint64_t m_adv = hh[(size_t)p * kGatherHdr + 1], m_p = m_adv < c->cap ? m_adv : c->cap;
                if (m_p) NCCL_TRY(r->Recv(c->d_recv[k] + (size_t)p * c->slot_bytes + hdr_bytes, gather_wire_bytes(m_p), kNcclUint8, p, c->comm, c->stream));
            }
    }
    // (a receive sized by a clamped count against a sender that believes in a larger capacity: the transport may refuse the
    // pair -- the gather is failed below either way, which is the error the caller must see)
    const int rc_p2 = grp.end();
    for (int i = 0; i < n; ++i) hs[i]->comm->seq_of[k] = seq;
    if (seq_out) *seq_out = seq;
    if (rc_p2 != 0 && !(rc && !consistent)) return nccl_fail("ncclGroupEnd (records)", rc_p2);
    if (rc && !consistent) {
        rc->failed[k] = true;
        HIP_TRY(hipSetDevice(root_h->device));
        HIP_TRY(hipEventRecord(rc->ev_host[k], rc->stream));
        return fail(AMR_EHIP, "amr_gather_hits: a rank's header is inconsistent (ranks out of step, or capacities differ); the gather's records are refused");
    }
    if (rc) AMR_TRY(finish_root(rc->d_hdr[k]));
    return AMR_OK;
}

}  // namespace

extern "C" {

amr_status amr_gather_hits(amr_handle *h, uint64_t *seq_out)
{
    if (!h || !h->comm) return fail(AMR_EINVAL, "amr_gather_hits: amr_comm_init first");
    if (h->last_slot < 0 && !h->last_empty) return fail(AMR_EINVAL, "amr_gather_hits: no batch collected yet");
    if (h->comm->local_world) return fail(AMR_EINVAL, "amr_gather_hits: this communicator's ranks share one process (amr_comm_init_all): use amr_gather_hits_all");
    return gather_many(&h, 1, seq_out);
}

amr_status amr_gather_hits_all(amr_handle **hs, int32_t n, uint64_t *seq_out)
{
    if (!hs || n < 1) return fail(AMR_EINVAL, "amr_gather_hits_all: null argument");
    for (int32_t i = 0; i < n; ++i) {
        if (!hs[i] || !hs[i]->comm) return fail(AMR_EINVAL, "amr_gather_hits_all: amr_comm_init_all first");
        if (hs[i]->comm->world != n || hs[i]->comm->rank != i) return fail(AMR_EINVAL, "amr_gather_hits_all: pass the handles of amr_comm_init_all, in its order");
        if (hs[i]->last_slot < 0 && !hs[i]->last_empty) return fail(AMR_EINVAL, "amr_gather_hits_all: a handle has no collected batch yet");
    }
    return gather_many(hs, n, seq_out);
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
    if (c->failed[k]) return fail(AMR_EHIP, "amr_gather_fetch: that gather failed (a rank's header was inconsistent)");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventSynchronize(c->ev_host[k]));      // the mirror copy of this gather, nothing else
    AMR_TRY(gather_unpack(c->h_recv[k] + (size_t)src_rank * c->slot_bytes, c->slot_bytes, out));
    if (out->seq != seq) return fail(AMR_EHIP, "amr_gather_fetch: a rank's slot carries another gather's sequence number (ranks out of step)");
    return AMR_OK;
}

}  // extern "C"
