// amrdemod.hip -- C ABI (include/amrdemod.h) over the gfx950 kernels.
//
// Host side of the drop-in boundary for rtlamr's protocol.Decoder
// (protocol/decode.go).  Geometry and registration follow the reference
// line by line (citations inline); everything that touches samples runs on the
// GPU.  There is NO CPU fallback: without a gfx950 device amr_create fails
// with AMR_ENODEV.
#include "amr_host.h"
#include "launch.h"

namespace {

thread_local std::string g_last_error;

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

amr_status amr_host::fail(amr_status s, const char *what, hipError_t e)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    g_last_error = buf;
    return s;
}

using namespace amr_host;

// RegisterProtocol for every entry + the arithmetic of Allocate (decode.go:100-141): no device involved.
amr_status amr_host::plan_geometry(const amr_protocol *protos, int32_t n_protos, amr_geometry &g, amr::SearchGeom &sg,
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
    if (const char *gt = getenv("AMR_GATE_TIMEOUT_US")) h->gate_timeout_ticks = strtoull(gt, nullptr, 10) * 100ull;   // test hook
    if (getenv("AMR_SINGLE_DBG")) (void)hipHostMalloc((void **)&h->d_single_dbg, 64, hipHostMallocCoherent);   // diagnostic
    h->no_single = getenv("AMR_NO_SINGLE") != nullptr;             // test hook: keep the regular kernels under the one-block tests
    if (const char *hc = getenv("AMR_HIT_CAP")) h->init_hit_cap = std::max<uint64_t>(64, strtoull(hc, nullptr, 10));

    {
        amr_status ps = plan_geometry(protos, n_protos, h->geom, h->sg, h->proto_pid, h->halo_bytes, h->hist_rows);
        if (ps != AMR_OK) { delete h; return ps; }
    }
    h->k1_coop_max = std::min<uint64_t>(kK1CoopMaxBlocks, kK1CoopMaxSamples / (uint64_t)h->geom.block_size);
    if (const char *cm = getenv("AMR_K1_COOP_MAX")) h->k1_coop_max = strtoull(cm, nullptr, 10);   // test hook / A-B
    if (const char *gd = getenv("AMR_GATE_DELAY_TICKS")) h->gate_delay_ticks = (uint32_t)strtoul(gd, nullptr, 10);   // A/B runs
    if (const char *ge = getenv("AMR_GATE_EVENT")) h->gate_event = ge[0] != '0';
    if (const char *ge = getenv("AMR_GATE_END")) h->gate_end_mode = atoi(ge);   // A/B runs
    if (const char *iw = getenv("AMR_INWAVE")) h->inwave_mode = atoi(iw);   // A/B runs
    if (const char *kp = getenv("AMR_K3_PRIO")) h->k3_prio = (uint32_t)atoi(kp) & 3u;   // A/B runs
    if (const char *kl = getenv("AMR_K3_LDS_KB")) h->k3_lds_min = (size_t)strtoul(kl, nullptr, 10) * 1024;   // A/B runs
    if (const char *kl = getenv("AMR_K2W_LDS_KB")) h->k2w_lds_min = (size_t)strtoul(kl, nullptr, 10) * 1024;   // A/B runs
    if (const char *rt = getenv("AMR_K1_ROUND_TILES")) h->k1_round_tiles = (uint32_t)strtoul(rt, nullptr, 10);   // test hook: batches of several K1 launches at test sizes
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
    // The search stream of the early search IS the tail stream: K2 of batch i runs behind K3 of batch i-1 there.  As a
    // stream with a hardware queue of its own (a stream of another priority gets one; a fifth plain stream shares the tail
    // stream's, which is how this form was found) the searching waves really sit next to K1 for its whole ragged end and K1
    // gets slower in every geometry (profiles/r05/early_timelines_*; chip 8: 0.269 against 0.249 ms per step).
    h->search_stream = h->tail_stream;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_switch, hipEventDisableTiming);
    if (const char *es = getenv("AMR_EARLY_SEARCH")) h->early_mode = atoi(es) != 0 ? 1 : 0;
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->h_flags, 16, hipHostMallocCoherent);
    if (e == hipSuccess) { h->h_flags[0] = 0; h->h_flags[1] = 0; }
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_tail_done, 32);
    if (e == hipSuccess) e = hipMemset(h->d_tail_done, 0, 32);
    if (e == hipSuccess) h->d_k1_started = h->d_tail_done + 1;
    if (e == hipSuccess) h->d_k1_ctr = reinterpret_cast<uint32_t *>(h->d_tail_done + 2);
    h->stream = h->own_stream;
    for (Slot &sl : h->slot) {
        if (e == hipSuccess) e = hipEventCreate(&sl.ev0);
        if (e == hipSuccess) e = hipEventCreate(&sl.ev1);
        if (e == hipSuccess) e = hipEventCreate(&sl.ev2);
        if (e == hipSuccess) e = hipEventCreate(&sl.ev_s);
        if (e == hipSuccess) e = hipEventCreate(&sl.ev_k2);
        if (e == hipSuccess) e = hipEventCreate(&sl.ev_t);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_pack, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_k2done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_gate, hipEventDisableTiming);
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
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_carry_alt, h->halo_bytes);
    if (e == hipSuccess) e = hipMemset(h->d_carry_alt, 0, h->halo_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_pkt_carry, 16);
    if (e == hipSuccess) e = hipMemset(h->d_pkt_carry, 0, 16);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_lut, 1024);
    const size_t head_bytes = h->halo_bytes + (size_t)64 * h->geom.block_size2;
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_head, head_bytes);
    if (e == hipSuccess) e = hipMemcpy(h->d_lut, h->lut, 1024, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(h->d_head, 0, head_bytes);
    h->d_carry_cur = h->d_head;
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
    dump_diagnostics(h);
    // the communicator first: its stream may still hold a pack kernel that reads the slots' result buffers
    if (h->comm) (void)amr_comm_destroy(h);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->search_stream) (void)hipStreamSynchronize(h->search_stream);
    if (h->tail_stream) (void)hipStreamSynchronize(h->tail_stream);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    void *ptrs[] = {h->d_lut, h->d_head, h->d_iq, h->d_untile, h->d_tail_done, h->d_pkt_carry, h->d_carry_alt};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (uint8_t *p : h->d_iqhist) if (p) (void)hipFree(p);
    if (h->h_flags) (void)hipHostFree(h->h_flags);
    if (h->h_iq1) (void)hipHostFree(h->h_iq1);
    if (h->d_single_dbg) {
        const unsigned long long *t = h->d_single_dbg;
        fprintf(stderr, "AMR_SINGLE_DBG (last call, us): stage+hist %.2f, mags %.2f, chain %.2f, filter %.2f, state+search %.2f, slots+slice %.2f, fence %.2f\n",
                (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01, (t[5] - t[4]) * 0.01, (t[6] - t[5]) * 0.01, (t[7] - t[6]) * 0.01);
        (void)hipHostFree(h->d_single_dbg);
    }
    for (Slot &sl : h->slot) {
        if (sl.ev_k2done) (void)hipEventDestroy(sl.ev_k2done);
        if (sl.ev_gate) (void)hipEventDestroy(sl.ev_gate);
        if (sl.d_k1flags) (void)hipFree(sl.d_k1flags);
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
    if (h->ev_switch) (void)hipEventDestroy(h->ev_switch);
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
    HIP_TRY(hipMemsetAsync(h->d_pkt_carry, 0, 16, h->stream));      // a fresh Decoder's pkt is zero (decode.go:151)
    AMR_TRY(sync_compute(h));
    h->carry_slot = 4;
    h->d_carry_cur = h->d_head;
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

size_t amr_halo_bytes(const amr_handle *h) { return h ? h->halo_bytes : 0; }
size_t amr_prime_blocks(const amr_handle *h) { return h ? (size_t)h->hist_rows + 1 : 0; }

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
    int n = snprintf(buf, buf_bytes, "amrdemod 0.1 %s %d CUs clock %d kHz chip %d BS %d PL %d preambles %d", prop.gcnArchName,
                     prop.multiProcessorCount, prop.clockRate, h->geom.chip_length, h->geom.block_size, h->geom.packet_length,
                     h->geom.n_preambles);
    {   // the demodulation kernel whole wave-tiles of this chip length run (bench.py: the committed PMC figure must be this kernel's)
        char name[160] = "";
        amr::K1Args q{};
        q.iq = reinterpret_cast<const uint8_t *>(name); q.n_blocks = sizeof name;
        if (amr::launch_k1(h->geom.chip_length, dim3(0), nullptr, q, nullptr, nullptr) && n > 0 && (size_t)n < buf_bytes)
            n += snprintf(buf + n, buf_bytes - (size_t)n, " | K1 %s |", name);
    }
    if (h->gate_timeouts && n > 0 && (size_t)n < buf_bytes)
        n += snprintf(buf + n, buf_bytes - (size_t)n, " gate-timeouts %llu", (unsigned long long)h->gate_timeouts);
    if (h->inwave_batches && n > 0 && (size_t)n < buf_bytes)
        n += snprintf(buf + n, buf_bytes - (size_t)n, " in-wave-searches %llu", (unsigned long long)h->inwave_batches);
    if ((h->researches || h->stale_reruns) && n > 0 && (size_t)n < buf_bytes)
        snprintf(buf + n, buf_bytes - (size_t)n, " re-searches %llu stale-reruns %llu", (unsigned long long)h->researches,
                 (unsigned long long)h->stale_reruns);
    return AMR_OK;
}

}  // extern "C"
