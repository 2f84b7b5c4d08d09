// amrdemod.hip -- C ABI (include/amrdemod.h) over the gfx950 kernels.
//
// Host side of the drop-in boundary for rtlamr's protocol.Decoder
// (protocol/decode.go).  Geometry and registration follow the reference
// line by line (citations inline); everything that touches samples runs on the
// GPU.  There is NO CPU fallback: without a gfx950 device amr_create fails
// with AMR_ENODEV.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "amrdemod.h"
#include "k1_demod.h"
#include "k2_search.h"
#include "synth.h"

namespace {

thread_local std::string g_last_error;

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

struct amr_handle {
    int device = 0;
    amr_geometry geom{};
    amr::SearchGeom sg{};
    std::vector<int> proto_pid;
    float lut[256];
    uint32_t halo_bytes = 0;   // HBA: aligned halo K1 reads before a block
    uint32_t hist_rows = 0;    // ceil(PL/BS)

    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    bool timing_valid = false;
    amr_timing timing{};

    float *d_lut = nullptr;
    uint8_t *d_carry = nullptr;
    bool zero_halo = true;
    uint8_t *d_iq = nullptr;      size_t iq_cap = 0;       // staging for host input
    uint32_t *d_qt = nullptr;     size_t qt_tiles = 0;     // tiles allocated
    uint32_t *d_counts = nullptr; uint64_t *d_offsets = nullptr; size_t cnt_tiles = 0;
    uint64_t *d_offs_pre = nullptr;
    uint32_t *d_overflow = nullptr;
    uint32_t *d_staging = nullptr; size_t staging_tiles = 0; uint32_t stage_cap = 1024;
    uint64_t *d_hit_pos = nullptr; uint8_t *d_pkt = nullptr; uint64_t out_cap = 0;
    uint32_t *d_untile = nullptr; size_t untile_words = 0;

    uint64_t calls_done = 0, block_base = 0;
    size_t last_n_blocks = 0;

    // result storage (valid until the next call)
    std::vector<uint64_t> r_off, r_block, r_pos;
    std::vector<uint32_t> r_idx;
    std::vector<uint8_t> r_pkt;
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

#define AMR_TRY(expr)                          \
    do {                                       \
        amr_status s_ = (expr);                \
        if (s_ != AMR_OK) return s_;           \
    } while (0)

template <bool TAIL>
void launch_k1(int cl, dim3 grid, hipStream_t st, const amr::K1Args &a)
{
    switch (cl) {
#define AMR_K1_CASE(N) case N: hipLaunchKernelGGL((amr::k1_demod<N, TAIL>), grid, dim3(64), 0, st, a); break;
        AMR_K1_CASE(8) AMR_K1_CASE(32) AMR_K1_CASE(40) AMR_K1_CASE(48) AMR_K1_CASE(56)
        AMR_K1_CASE(64) AMR_K1_CASE(72) AMR_K1_CASE(80) AMR_K1_CASE(88) AMR_K1_CASE(96)
#undef AMR_K1_CASE
    default: break;
    }
}

amr_status ensure_capacity(amr_handle *h, size_t n_blocks)
{
    const size_t bt = (n_blocks + 63) / 64;   // batch tiles
    const size_t tile_words = (size_t)64 * h->sg.wpb;
    if (bt + 2 > h->qt_tiles) {
        // keep the history tile across a regrow
        uint32_t *nq = nullptr;
        const size_t nt = bt + 2;
        hipError_t e = hipMalloc((void **)&nq, nt * tile_words * 4);
        if (e != hipSuccess) return fail(AMR_ENOMEM, "hipMalloc(qt)", e);
        if (h->d_qt) {
            HIP_TRY(hipMemcpyAsync(nq, h->d_qt, tile_words * 4, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            HIP_TRY(hipFree(h->d_qt));
        } else {
            HIP_TRY(hipMemsetAsync(nq, 0, tile_words * 4, h->stream));
        }
        h->d_qt = nq;
        h->qt_tiles = nt;
    }
    const size_t st = bt + 1;                 // tiles searched
    if (st > h->cnt_tiles) {
        AMR_TRY(dev_realloc(h->d_counts, st * h->sg.n_pre));
        AMR_TRY(dev_realloc(h->d_offsets, st * h->sg.n_pre));
        h->cnt_tiles = st;
    }
    if (st > h->staging_tiles) {
        AMR_TRY(dev_realloc(h->d_staging, st * h->sg.n_pre * h->stage_cap));
        h->staging_tiles = st;
    }
    if (h->out_cap == 0) {
        h->out_cap = 1 << 16;
        AMR_TRY(dev_realloc(h->d_hit_pos, h->out_cap));
        AMR_TRY(dev_realloc(h->d_pkt, h->out_cap * h->sg.pkt_bytes));
    }
    return AMR_OK;
}

// K1 for n_blocks rows starting at d_iq, then (optionally) search; history/carry update last.
amr_status run_batch(amr_handle *h, const uint8_t *d_iq, size_t n_blocks, bool search, amr_result *res)
{
    HIP_TRY(hipSetDevice(h->device));
    if (n_blocks == 0 || n_blocks > 0x7fffffffull) return fail(AMR_EINVAL, "n_blocks out of range");
    AMR_TRY(ensure_capacity(h, n_blocks));
    hipStream_t st = h->stream;
    const uint32_t bs = (uint32_t)h->geom.block_size;
    const uint32_t full = (uint32_t)(n_blocks / 64), rem = (uint32_t)(n_blocks % 64);

    amr::K1Args k1{};
    k1.iq = d_iq;
    k1.carry = h->d_carry;
    k1.lut = h->d_lut;
    k1.qt = h->d_qt;
    k1.n_blocks = (uint32_t)n_blocks;
    k1.block_size = bs;
    k1.zero_halo = h->zero_halo ? 1u : 0u;

    HIP_TRY(hipEventRecord(h->ev0, st));
    if (full) { k1.wg_first = 0; launch_k1<false>(h->geom.chip_length, dim3(full), st, k1); }
    if (rem) { k1.wg_first = full; launch_k1<true>(h->geom.chip_length, dim3(1), st, k1); }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev1, st));

    const uint32_t n_tiles = (uint32_t)((n_blocks + 63) / 64) + 1;
    uint64_t total = 0;
    if (search) {
        const uint32_t n_pre = h->sg.n_pre;
        for (int attempt = 0;; ++attempt) {
            amr::K2Args k2{};
            k2.qt = h->d_qt;
            k2.counts = h->d_counts;
            k2.staging = h->d_staging;
            k2.overflow = h->d_overflow;
            k2.n_tiles = n_tiles;
            k2.cap = h->stage_cap;
            k2.n_lo = -(int64_t)h->geom.packet_length;
            k2.n_hi = (int64_t)n_blocks * bs - (int64_t)h->geom.packet_length;
            k2.g = h->sg;
            HIP_TRY(hipMemsetAsync(h->d_overflow, 0, 4, st));
            const size_t lds2 = ((size_t)h->sg.wpb * 65 + 8) * 4;
            hipLaunchKernelGGL(amr::k2_search, dim3(n_tiles), dim3(256), lds2, st, k2);
            amr::ScanArgs sc{h->d_counts, h->d_offsets, h->d_offs_pre, n_tiles, n_pre};
            hipLaunchKernelGGL(amr::k2s_scan, dim3(1), dim3(1024), 0, st, sc);
            amr::K3Args k3{};
            k3.qt = h->d_qt; k3.counts = h->d_counts; k3.offsets = h->d_offsets; k3.staging = h->d_staging;
            k3.hit_pos = h->d_hit_pos; k3.pkt = h->d_pkt; k3.out_cap = h->out_cap;
            k3.n_tiles = n_tiles; k3.cap = h->stage_cap; k3.g = h->sg;
            hipLaunchKernelGGL(amr::k3_slice, dim3(n_tiles, n_pre), dim3(256), 0, st, k3);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(h->ev2, st));

            h->r_off.assign(n_pre + 1, 0);
            uint32_t ovf = 0;
            HIP_TRY(hipMemcpyAsync(h->r_off.data(), h->d_offs_pre, (n_pre + 1) * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&ovf, h->d_overflow, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            total = h->r_off[n_pre];
            if (attempt > 8) return fail(AMR_EOVERFLOW, "hit capacity could not be grown");
            if (ovf) {  // some tile found more hits than its staging slot holds: grow and redo the search
                h->stage_cap *= 8;
                const uint64_t lim = (uint64_t)64 * bs;
                if (h->stage_cap > lim) h->stage_cap = (uint32_t)lim;
                AMR_TRY(dev_realloc(h->d_staging, h->staging_tiles * n_pre * (size_t)h->stage_cap));
                continue;
            }
            if (total > h->out_cap) {
                uint64_t nc = h->out_cap;
                while (nc < total) nc *= 2;
                h->out_cap = nc;
                AMR_TRY(dev_realloc(h->d_hit_pos, h->out_cap));
                AMR_TRY(dev_realloc(h->d_pkt, h->out_cap * h->sg.pkt_bytes));
                continue;
            }
            break;
        }
    } else {
        HIP_TRY(hipEventRecord(h->ev2, st));
    }

    // state carried to the next batch: quantized history rows and the IQ halo (decode.go:165-166)
    amr::HistArgs ha{h->d_qt, (uint32_t)n_blocks, h->hist_rows, h->sg.wpb, h->sg.lg_wpb};
    hipLaunchKernelGGL(amr::k_hist_update, dim3(1), dim3(1024), (size_t)h->hist_rows * h->sg.wpb * 4, st, ha);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h->d_carry, d_iq + n_blocks * (size_t)h->geom.block_size2 - h->halo_bytes, h->halo_bytes,
                           hipMemcpyDeviceToDevice, st));
    h->zero_halo = false;

    if (search) {
        h->r_pos.resize(total);
        h->r_pkt.resize(total * h->sg.pkt_bytes);
        if (total) {
            HIP_TRY(hipMemcpyAsync(h->r_pos.data(), h->d_hit_pos, total * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(h->r_pkt.data(), h->d_pkt, total * h->sg.pkt_bytes, hipMemcpyDeviceToHost, st));
        }
    }
    HIP_TRY(hipStreamSynchronize(st));

    float a = 0, b = 0, c = 0;
    if (hipEventElapsedTime(&a, h->ev0, h->ev1) == hipSuccess && hipEventElapsedTime(&b, h->ev1, h->ev2) == hipSuccess &&
        hipEventElapsedTime(&c, h->ev0, h->ev2) == hipSuccess) {
        h->timing = amr_timing{a, b, c};
        h->timing_valid = true;
    }

    if (search) {
        h->r_block.resize(total);
        h->r_idx.resize(total);
        const uint32_t lg = h->sg.lg_block_size;
        for (uint64_t i = 0; i < total; ++i) {
            const uint64_t pos = h->r_pos[i];
            h->r_block[i] = (pos >> lg) + h->calls_done + h->block_base;
            h->r_idx[i] = (uint32_t)(pos & (bs - 1));
        }
        h->calls_done += n_blocks;
        h->last_n_blocks = n_blocks;
        if (res) {
            res->n_preambles = h->sg.n_pre;
            res->pkt_bytes = h->sg.pkt_bytes;
            res->n_hits = total;
            res->preamble_offset = h->r_off.data();
            res->hit_block = h->r_block.data();
            res->hit_idx = h->r_idx.data();
            res->pkt = h->r_pkt.data();
        }
    }
    return AMR_OK;
}

amr_status stage_host_input(amr_handle *h, const uint8_t *iq, size_t bytes)
{
    if (bytes > h->iq_cap) {
        AMR_TRY(dev_realloc(h->d_iq, bytes));
        h->iq_cap = bytes;
    }
    HIP_TRY(hipMemcpyAsync(h->d_iq, iq, bytes, hipMemcpyHostToDevice, h->stream));
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

    // RegisterProtocol, decode.go:100-128: field-wise max, preambles grouped by value
    amr_geometry &g = h->geom;
    amr::SearchGeom &sg = h->sg;
    for (int i = 0; i < n_protos; ++i) {
        const amr_protocol &p = protos[i];
        if (!p.preamble || !legal_chip_length(p.chip_length) || p.preamble_symbols <= 0 || p.packet_symbols <= 0 ||
            (g.chip_length && p.chip_length != g.chip_length)) {
            delete h;
            return fail(AMR_EINVAL, "amr_create: bad protocol entry (chip length must be one of flags.go:127-132)");
        }
        const size_t len = strlen(p.preamble);
        if (len == 0 || len > AMR_MAX_PREAMBLE_BITS) { delete h; return fail(AMR_EINVAL, "preamble length"); }
        g.data_rate = std::max(g.data_rate, p.data_rate);
        g.chip_length = std::max(g.chip_length, p.chip_length);
        g.preamble_symbols = std::max(g.preamble_symbols, p.preamble_symbols);
        g.packet_symbols = std::max(g.packet_symbols, p.packet_symbols);
        uint64_t bits = 0;
        for (size_t b = 0; b < len; ++b) {
            if (p.preamble[b] != '0' && p.preamble[b] != '1') { delete h; return fail(AMR_EINVAL, "preamble must be 0/1"); }
            if (p.preamble[b] == '1') bits |= 1ull << b;
        }
        int pid = -1;
        for (uint32_t q = 0; q < sg.n_pre; ++q)
            if (sg.pre_len[q] == len && sg.pre_bits[q] == bits) pid = (int)q;
        if (pid < 0) {
            if (sg.n_pre == AMR_MAX_PREAMBLES) { delete h; return fail(AMR_EINVAL, "too many distinct preambles"); }
            pid = (int)sg.n_pre++;
            sg.pre_len[pid] = (uint32_t)len;
            sg.pre_bits[pid] = bits;
        }
        h->proto_pid.push_back(pid);
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
    h->halo_bytes = (uint32_t)((4 * g.chip_length + 127) & ~127);
    h->hist_rows = (uint32_t)((g.packet_length + g.block_size - 1) / g.block_size);
    // every preamble must fit the search window the geometry provides (true for all rtlamr parsers,
    // where PreambleSymbols >= len(Preamble)); the tiled search needs <= 63 history rows and
    // word-aligned PacketLength
    for (uint32_t q = 0; q < sg.n_pre; ++q)
        if ((int)sg.pre_len[q] > g.preamble_symbols) { delete h; return fail(AMR_EINVAL, "preamble longer than PreambleSymbols"); }
    if (h->hist_rows > 63 || (g.packet_length & 63) || g.block_size < 512 || g.packet_symbols < g.preamble_symbols) {
        delete h;
        return fail(AMR_EINVAL, "geometry outside the supported range");
    }

    // NewMagLUT, decode.go:209-216: float32 divide then float32 square, two roundings per entry.
    for (int i = 0; i < 256; ++i) {
        volatile float q = (127.5f - (float)i) / 127.5f;
        volatile float sq = q * q;
        h->lut[i] = sq;
    }

    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    h->stream = h->own_stream;
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e == hipSuccess) e = hipEventCreate(&h->ev2);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_lut, 1024);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_carry, h->halo_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_offs_pre, (AMR_MAX_PREAMBLES + 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_overflow, 4);
    if (e == hipSuccess) e = hipMemcpy(h->d_lut, h->lut, 1024, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(h->d_carry, 0, h->halo_bytes);
    if (e != hipSuccess) { amr_destroy(h); return fail(AMR_EHIP, "amr_create: device setup", e); }
    *out = h;
    return AMR_OK;
}

amr_status amr_destroy(amr_handle *h)
{
    if (!h) return AMR_OK;
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    void *ptrs[] = {h->d_lut, h->d_carry, h->d_iq, h->d_qt, h->d_counts, h->d_offsets, h->d_offs_pre,
                    h->d_overflow, h->d_staging, h->d_hit_pos, h->d_pkt, h->d_untile};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev2) (void)hipEventDestroy(h->ev2);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return AMR_OK;
}

amr_status amr_reset(amr_handle *h)
{
    if (!h) return fail(AMR_EINVAL, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    if (h->d_qt) HIP_TRY(hipMemsetAsync(h->d_qt, 0, (size_t)64 * h->sg.wpb * 4, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->zero_halo = true;
    h->calls_done = 0;
    h->last_n_blocks = 0;
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

amr_status amr_set_stream(amr_handle *h, void *hip_stream)
{
    if (!h) return fail(AMR_EINVAL, "null handle");
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return AMR_OK;
}

amr_status amr_set_block_base(amr_handle *h, uint64_t base)
{
    if (!h) return fail(AMR_EINVAL, "null handle");
    h->block_base = base;
    return AMR_OK;
}

amr_status amr_decode_batch(amr_handle *h, const uint8_t *iq, size_t iq_bytes, size_t n_blocks, amr_result *res)
{
    if (!h || !iq) return fail(AMR_EINVAL, "null argument");
    const size_t need = n_blocks * (size_t)h->geom.block_size2;
    if (iq_bytes < need) return fail(AMR_EINVAL, "short input (the Go decoder panics here, decode.go:222)");
    HIP_TRY(hipSetDevice(h->device));
    AMR_TRY(stage_host_input(h, iq, need));
    return run_batch(h, h->d_iq, n_blocks, true, res);
}

amr_status amr_decode_batch_device(amr_handle *h, const void *d_iq, size_t n_blocks, amr_result *res)
{
    if (!h || !d_iq) return fail(AMR_EINVAL, "null argument");
    return run_batch(h, (const uint8_t *)d_iq, n_blocks, true, res);
}

size_t amr_halo_bytes(const amr_handle *h) { return h ? h->halo_bytes : 0; }
size_t amr_prime_blocks(const amr_handle *h) { return h ? (size_t)h->hist_rows + 1 : 0; }

amr_status amr_prime(amr_handle *h, const uint8_t *lead, const uint8_t *halo_iq, size_t n_blocks, int on_device)
{
    if (!h || !halo_iq) return fail(AMR_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (lead) {
        HIP_TRY(hipMemcpyAsync(h->d_carry, lead, h->halo_bytes,
                               on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
        h->zero_halo = false;
    }
    const uint8_t *src = halo_iq;
    if (!on_device) {
        AMR_TRY(stage_host_input(h, halo_iq, n_blocks * (size_t)h->geom.block_size2));
        src = h->d_iq;
    }
    return run_batch(h, src, n_blocks, false, nullptr);
}

amr_status amr_copy_quantized(amr_handle *h, uint8_t *out, size_t out_bytes)
{
    if (!h || !out) return fail(AMR_EINVAL, "null argument");
    const size_t words = h->last_n_blocks * h->sg.wpb;
    if (out_bytes < words * 4) return fail(AMR_EINVAL, "output buffer too small");
    if (words == 0) return AMR_OK;
    HIP_TRY(hipSetDevice(h->device));
    if (words > h->untile_words) {
        AMR_TRY(dev_realloc(h->d_untile, words));
        h->untile_words = words;
    }
    hipLaunchKernelGGL(amr::k_untile, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, h->stream, h->d_qt, h->d_untile,
                       (uint32_t)h->last_n_blocks, h->sg.lg_wpb);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, h->d_untile, words * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
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

amr_status amr_synth_noise(int32_t device_id, void *d_iq, uint64_t n_samples, uint64_t seed, uint64_t first_sample)
{
    if (!d_iq || (n_samples & 7)) return fail(AMR_EINVAL, "n_samples must be a multiple of 8");
    HIP_TRY(hipSetDevice(device_id));
    const uint64_t threads = n_samples / 8;
    hipLaunchKernelGGL(amr::k_synth_noise, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, (uint8_t *)d_iq,
                       n_samples, seed, first_sample);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return AMR_OK;
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
