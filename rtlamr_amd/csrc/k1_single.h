// One Decode call in one launch -- the unchanged main.go loop (main.go:235: one block per Decoder.Decode, results back
// before the next call).  amr_decode_batch(n_blocks = 1) used to be five dependent launches (k1c_demod, the search, K3,
// the state update, ...) behind a host-to-device copy: 74-81 us per call against 10 us for the CPU port, 48 us of it the
// running sum as a DPP chain across the lanes of one wave.  Here ONE workgroup does the whole call:
//
//   A  magnitudes of the Signal (SL history samples + the block, decode.go:163-170) in parallel: two LUT gathers and one
//      float add per sample (decode.go:222); the quantized history (PacketLength bits, decode.go:166) into LDS as a bit array
//   B  the running sum (decode.go:232-236), sequential float32 in the reference, reproduced EXACTLY by a parallel scan over
//      integer transducers, one binade of the sum at a time (ks_running_sum below): 45 us -> a few us
//   C  the matched filter and the sign (decode.go:239-244), parallel over the outputs, a ballot per 64 of them: the new
//      BlockSize bits behind the history in LDS and as row 64 of the slot's tiled bitstream (amr_copy_quantized, a
//      re-search by the regular kernels)
//   D  the state the Decoder carries to the next call (decode.go:165-166): the next slot's history rows, the IQ halo
//   E  Search (decode.go:255-328), every preamble: 32 positions per lane-step as AND of the taps' windows -- the set the
//      byte-prefiltered two-pass search returns for every legal chip length
//   F  the hits in Search's order (preamble, idx ascending) and Slice (decode.go:353-375): PacketSymbols bits at a stride
//      of SymbolLength, MSB first -- written to the slot's packed result AND straight into the pinned host mirror, with the
//      per-preamble offsets and the batch ticket: amr_collect finds everything there, no copy follows.
//
// Output conventions are the regular kernels' (K3Args: packed result, offsets, overflow word), so that anything this
// kernel cannot finish -- more hits than the result buffers hold -- is taken over by the regular search on the same slot.
#pragma once
#include "exact_sum.h"
#include "k2_common.h"

namespace amr {

struct SingleArgs {
    const uint8_t *iq;          // the block: 2 * BlockSize bytes (device memory, or pinned host memory read over the link)
    const uint8_t *carry;       // head buffer: the HBA stream bytes in front of the block
    uint8_t *carry_out;         // ... and where this block's last HBA bytes go (the same buffer)
    const float *lut;           // NewMagLUT
    uint32_t *qt;               // the slot's tiled bitstream: tile 0 = history rows, row 64 = this block
    uint32_t *qt_next;          // the next slot's: receives the new history rows
    uint32_t *ovf_next;         // the next slot's overflow word: reset
    uint32_t *gcnt_next;        // the next slot's group sums (K2 -> K3 of the regular kernels): reset, as hist_body does
    uint32_t gcnt_words;
    uint8_t *out;               // packed result [hit_block u64 x n | hit_idx u32 x n | pkt x n] (device)
    uint8_t *h_out;             // the same in pinned host memory
    uint64_t cap;               // hits either buffer holds
    uint64_t *offs_pre;         // [n_pre + 1] (device)
    uint64_t *h_offs_pre;       // [n_pre + 1] (pinned)
    uint32_t *h_overflow;       // pinned
    uint64_t block_base;        // call index of the block
    uint64_t *done_flag;        // pinned: receives done_value when everything above has been written
    uint64_t *adone_flag;       // pinned: "the compute-stream part of the batch is done" (the same moment here)
    uint64_t done_value;
    uint32_t chip_length;
    uint32_t halo_bytes;        // HBA
    uint32_t hist_rows;         // HR = ceil(PL / BS)
    uint32_t zero_halo;         // fresh Decoder: the SL history magnitudes are 0.0 (decode.go:144)
    unsigned long long *dbg;    // null, or 8 words: 100 MHz time stamps of the phases (AMR_SINGLE_DBG, tools/single_block_rate.py)
    SearchGeom g;
};
#define KS_STAMP(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)

constexpr int kSingleThreads = 512;

// dynamic LDS of k_single_block for a geometry: magnitudes, sums, LUT, bit array, hit masks, scan scratch
inline size_t k_single_lds_bytes(const SearchGeom &g)
{
    const size_t n_sig = (size_t)g.block_size + g.symbol_length;
    const size_t qwords = ((size_t)g.packet_length + g.block_size) / 32 + 4;
    return (2 * n_sig + 256 + qwords + (size_t)g.n_pre * g.wpb + 2 * kSingleThreads + 64) * 4;
}

// sequential float32 sums of mag[i0 .. i1) by the calling lane, c = the sum in front of them; returns the last sum
__device__ __forceinline__ float ks_sum_seq(const float *mag, float *cs, uint32_t i0, uint32_t i1, float c)
{
    for (uint32_t i = i0; i < i1; ++i) { c = c + mag[i]; cs[i] = c; }      // decode.go:234
    return c;
}

// the same for mag[0 .. 4 * n4), n4 a multiple of 16, software-pipelined by hand: the 32 operands of the NEXT stretch are
// requested before the 32 dependent additions of this one start (left to itself hipcc waits for every ds_read right in
// front of its four additions and for the ds_write behind them: 35 cycles per sample instead of 15)
__device__ __forceinline__ float ks_sum_seq_fast(const float *mag, float *cs, uint32_t n4)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f *__restrict__ src = reinterpret_cast<const v4f *>(mag);
    v4f *__restrict__ dst = reinterpret_cast<v4f *>(cs);
    constexpr uint32_t U = 8;                                         // v4f per stretch
    float c = 0.0f;                                                   // csum[0]
    v4f A[U], B[U];                                                   // two stretches, ping-pong: no register copies
    auto fetch = [&](v4f (&x)[U], uint32_t i0) {
#pragma unroll
        for (uint32_t k = 0; k < U; ++k) x[k] = src[i0 + k < n4 ? i0 + k : 0];         // past the end: a harmless re-read
    };
    auto sum_store = [&](v4f (&x)[U], uint32_t i0) {
#pragma unroll
        for (uint32_t k = 0; k < U; ++k) {
            v4f m = x[k];
            c = c + m.x; m.x = c;                                     // decode.go:234
            c = c + m.y; m.y = c;
            c = c + m.z; m.z = c;
            c = c + m.w; m.w = c;
            x[k] = m;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (uint32_t k = 0; k < U; ++k) dst[i0 + k] = x[k];
    };
    fetch(A, 0);
    for (uint32_t i = 0; i < n4; i += 2 * U) {
        fetch(B, i + U);
        __builtin_amdgcn_sched_barrier(0);
        sum_store(A, i);
        fetch(A, i + 2 * U);
        __builtin_amdgcn_sched_barrier(0);
        sum_store(B, i + U);
    }
    return c;
}

// cs[j] = csum[j + 1] for j < n, by the whole workgroup (kSingleThreads threads); scratch: 64 words.
// A phase takes a WINDOW of terms, twice as many as have been summed so far (stationary noise doubles its sum when the
// count doubles): thread t its run of `per` consecutive terms -> the run's transducer (the terms' (a, tie) parked in the
// very words of cs that will receive their sums) -> scan inside the wave (DPP), the 8 wave totals through LDS -> every
// thread walks its run again from its exclusive prefix, writes the sums and reports the first term that leaves the
// binade -> everybody reads that index, performs the one float32 addition and moves on.  Two barriers per phase.
__device__ __forceinline__ void ks_running_sum(const float *mag, float *cs, uint32_t n, uint32_t *scratch, uint32_t tid)
{
    constexpr uint32_t NT = kSingleThreads, NW = NT / 64;
    const uint32_t lane = tid & 63, wv = tid >> 6;
    KsPair *wtot = reinterpret_cast<KsPair *>(scratch);     // [NW] inclusive total of every wave
    uint32_t *first = scratch + 2 * NW;                     // [3] first term that leaves the binade, rotating by phase
    uint32_t *seq = scratch + 2 * NW + 3;                   // [2] the sequential prologue's (position, sum)
    if (tid == 0) {
        const float c = ks_sum_seq_fast(mag, cs, kSumSeq / 4);       // n >= BlockSize + SymbolLength >= 272
        seq[0] = kSumSeq; seq[1] = __float_as_uint(c);
        first[0] = 0xffffffffu; first[1] = 0xffffffffu; first[2] = 0xffffffffu;
    }
    __syncthreads();
    uint32_t pos = seq[0], cbits = seq[1];                  // workgroup-uniform from here on
    for (uint32_t phase = 0; pos < n; ++phase) {
        if (cbits == 0 || phase >= kSumMaxPhases) {         // a sum still zero behind 256 terms, or far too many binades: sequentially
            if (tid == 0) ks_sum_seq(mag, cs, pos, n, __uint_as_float(cbits));
            break;
        }
        const uint32_t E = cbits >> 23;                     // biased exponent of the sum (positive, normal: >= 2^-15)
        const float ulp = __uint_as_float((E - 23u) << 23), inv_ulp = __uint_as_float((277u - E) << 23);
        const uint32_t n0 = (cbits & 0x7fffffu) | 0x800000u;
        const uint32_t rem = n - pos, want = 2 * pos > NT ? 2 * pos : NT, win = want < rem ? want : rem;
        uint32_t per = (win + NT - 1) / NT;
        per |= 1u;                                          // odd: lane-to-lane stride in LDS words without bank conflicts
        const uint32_t end = pos + win;
        const uint32_t i0 = pos + tid * per < end ? pos + tid * per : end, i1 = i0 + per < end ? i0 + per : end;
        uint32_t *park = reinterpret_cast<uint32_t *>(cs);
        // 1. this thread's run of terms as one transducer
        KsPair f{0u, 0u};
        for (uint32_t i = i0; i < i1; ++i) {
            uint32_t a; bool tie;
            ks_term(mag[i], inv_ulp, a, tie);
            park[i] = a | (tie ? 0x80000000u : 0u);         // a <= 2^26
            KsPair g{ks_sat(a + (tie ? (a & 1u) : 0u)), ks_sat(a + (tie ? ((a + 1u) & 1u) : 0u))};
            f = ks_compose(f, g);
        }
        // 2. exclusive prefix: inside the wave by DPP, the waves in front through LDS
        const KsPair inc = ks_wave_scan(f);
        KsPair ex;
        ex.d0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc.d0, 0x138, 0xf, 0xf, true);   // wave_shr:1, lane 0: identity
        ex.d1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc.d1, 0x138, 0xf, 0xf, true);
        if (lane == 63) wtot[wv] = inc;
        // the next phase's word (three rotate: the one being reset was last read two phases ago, two barriers back; with
        // two, a wave still reading the previous phase's result would race with this reset)
        if (tid == 0) first[(phase + 1) % 3] = 0xffffffffu;
        __syncthreads();
        KsPair base{0u, 0u};
        for (uint32_t w = 0; w < wv; ++w) base = ks_compose(base, wtot[w]);
        const KsPair e = ks_compose(base, ex);
        // 3. the sums of this thread's terms, up to the first one that leaves the binade
        {
            uint32_t nn = ks_sat(n0 + ((n0 & 1u) ? e.d1 : e.d0));
            uint32_t ev = 0xffffffffu;
            if (nn >= (1u << 24)) ev = i0;                  // left the binade in front of this thread's run (a smaller index wins)
            else
                for (uint32_t i = i0; i < i1; ++i) {
                    const uint32_t t = park[i];
                    nn = ks_step(nn, t & 0x7fffffffu, (t >> 31) != 0);
                    if (nn >= (1u << 24)) { ev = i; break; }
                    cs[i] = (float)nn * ulp;                // exact: nn < 2^24, ulp a power of two
                }
            if (ev < end) atomicMin(&first[phase % 3], ev);
        }
        __syncthreads();
        // 4. the term that leaves the binade: that one addition in float32 (everybody, the same value); none in this window:
        //    on with the next window in the same binade
        const uint32_t j = first[phase % 3];
        if (j < end) {
            const float cp = j == pos ? __uint_as_float(cbits) : cs[j - 1];
            const float c = cp + mag[j];                    // decode.go:234
            if (tid == 0) cs[j] = c;
            pos = j + 1; cbits = __float_as_uint(c);
        } else {
            cbits = __float_as_uint(cs[end - 1]);
            pos = end;
        }
        // (cs[j] is written by thread 0 after everybody has read cs[j - 1]; cs[j] itself is read in the next phase only
        // behind its first barrier)
    }
    __syncthreads();
}

__global__ __launch_bounds__(kSingleThreads) void k_single_block(const SingleArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t ks_lds[];
    const SearchGeom &g = a.g;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr uint32_t NT = kSingleThreads, NW = kSingleThreads / 64;
    const uint32_t BS = g.block_size, SL = g.symbol_length, CL = a.chip_length, PL = g.packet_length;
    const uint32_t wpb = g.wpb, lg_wpb = g.lg_wpb, HR = a.hist_rows, HBA = a.halo_bytes, HB = 2 * SL;
    const uint32_t n_sig = BS + SL, qw_hist = PL >> 5, qwords = (PL + BS) >> 5, n_pre = g.n_pre;
    float *mag = reinterpret_cast<float *>(ks_lds);                 // [n_sig] |z|^2
    float *cs = mag + n_sig;                                          // [n_sig] cs[s] = csum[s + 1]
    float *lut = cs + n_sig;                                          // [256]
    uint32_t *qb = reinterpret_cast<uint32_t *>(lut + 256);           // [qwords + 4] Quantized, first sample in bit 31 of word 0
    uint32_t *hm = qb + qwords + 4;                                   // [n_pre][wpb] hit masks, position 32w + t in bit 31 - t
    uint32_t *part = hm + n_pre * wpb;                                // [2 NT] scan scratch
    uint32_t *misc = part + 2 * NT;                                   // [0..n_pre] per-preamble bases, [n_pre] = total; [16..] the sum's control words

    KS_STAMP(0);
    // ---- A: LUT, quantized history, magnitudes ----
    if (tid < 256) lut[tid] = a.lut[tid];
    {   // Quantized[0 .. PL) = the last PL bits of the HR history rows (rows 64 - HR .. 63 of tile 0); PL and BS are
        // multiples of 32: whole words
        const uint32_t skip = ((HR << g.lg_block_size) - PL) >> 5;
        for (uint32_t x = tid; x < qw_hist; x += NT) {
            const uint32_t hw = skip + x, j = hw >> lg_wpb, w = hw & (wpb - 1);
            qb[x] = a.qt[qt_index(64 - HR + j, w, lg_wpb)];
        }
        if (tid < 4) qb[qwords + tid] = 0;
    }
    // the Signal's bytes into LDS first (in the space the sums will take), 16 bytes per lane and all requests in flight
    // together: the block may sit in pinned HOST memory, where a dependent round of loads costs a link round trip
    uint8_t *raw = reinterpret_cast<uint8_t *>(cs);                   // [HB + 2 BS]: SL history samples, then the block
    for (uint32_t i = tid; i < (2 * BS) / 16; i += NT)
        reinterpret_cast<uint4 *>(raw + HB)[i] = reinterpret_cast<const uint4 *>(a.iq)[i];
    for (uint32_t i = tid; i < HB / 4; i += NT)
        reinterpret_cast<uint32_t *>(raw)[i] = reinterpret_cast<const uint32_t *>(a.carry + HBA - HB)[i];
    __syncthreads();                                                  // the LUT, the bytes; everybody has read the old halo
    KS_STAMP(1);
    for (uint32_t s = tid; s < n_sig; s += NT) {
        const uint32_t v = reinterpret_cast<const uint16_t *>(raw)[s];
        float m = lut[v & 0xff] + lut[v >> 8];                        // decode.go:222
        if (a.zero_halo && s < SL) m = 0.0f;                          // decode.go:144
        mag[s] = m;
    }
    // this block's last HBA bytes: the IQ halo of the next call (decode.go:165); the other state of the next slot
    for (uint32_t i = tid; i < HBA / 4; i += NT)
        reinterpret_cast<uint32_t *>(a.carry_out)[i] = reinterpret_cast<const uint32_t *>(raw + HB + 2 * BS - HBA)[i];
    if (tid == NT - 1) *a.ovf_next = 0;
    for (uint32_t i = tid; i < a.gcnt_words / kGroupStride; i += NT) a.gcnt_next[i * kGroupStride] = 0;
    __syncthreads();

    KS_STAMP(2);
    // ---- B: the running sum (decode.go:232-236): the reference's sequential float32 additions, reproduced exactly ----
    ks_running_sum(mag, cs, n_sig, misc + 16, tid);
    __syncthreads();
    KS_STAMP(3);

    // ---- C: matched filter, sign, pack (decode.go:239-244) ----
    for (uint32_t q = wv; q < BS / 64; q += NW) {
        const uint32_t i = q * 64 + lane;
        const float c0 = i ? cs[i - 1] : 0.0f, c1 = cs[i + CL - 1], c2 = cs[i + SL - 1];
        const float lo = c1 - c0;                                     // decode.go:241
        const float up = c2 - c1;                                     // decode.go:242
        const float f = lo - up;                                      // decode.go:243
        const uint64_t neg = __ballot(__float_as_uint(f) >> 31);
        // Quantized = 1 - signbit (decode.go:244); first sample in bit 31 of its word
        const uint32_t w0 = __builtin_bitreverse32(~(uint32_t)neg), w1 = __builtin_bitreverse32(~(uint32_t)(neg >> 32));
        if (lane == 0) {
            const uint32_t w = 2 * q;
            qb[qw_hist + w] = w0; qb[qw_hist + w + 1] = w1;
            *reinterpret_cast<uint2 *>(a.qt + qt_index(64, w, lg_wpb)) = make_uint2(w0, w1);
        }
    }
    __syncthreads();

    KS_STAMP(4);
    // ---- D: the next slot's history rows = the last HR rows of (old history, this block) ----
    for (uint32_t x = tid; x < (HR << lg_wpb); x += NT) {
        const uint32_t j = x >> lg_wpb, w = x & (wpb - 1);
        const uint32_t v = j + 1 < HR ? a.qt[qt_index(64 - HR + j + 1, w, lg_wpb)] : qb[qw_hist + w];
        a.qt_next[qt_index(64 - HR + j, w, lg_wpb)] = v;
    }

    // ---- E: Search, every preamble (decode.go:255-328): position 32w + t matches iff every tap p has
    // Quantized[32w + t + p * SL] == preamble[p] ----
    for (uint32_t e = tid; e < n_pre * wpb; e += NT) {
        const uint32_t q = e >> lg_wpb, w = e & (wpb - 1);
        const uint64_t bits = g.pre_bits[q];
        const uint32_t L = g.pre_len[q];
        uint32_t M = 0xffffffffu;
        for (uint32_t p = 0; p < L; ++p) {
            const uint32_t o = (w << 5) + p * SL, x = o >> 5, sh = o & 31;
            const uint32_t W = sh ? (qb[x] << sh) | (qb[x + 1] >> (32 - sh)) : qb[x];
            M &= ((bits >> p) & 1) ? W : ~W;
        }
        hm[e] = M;
    }
    __syncthreads();

    KS_STAMP(5);
    // ---- F: slots.  Entries e = (q, w) in Search's order; thread t owns entries [t * per, t * per + per) ----
    const uint32_t n_ent = n_pre * wpb, per = (n_ent + NT - 1) / NT;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t e = tid * per + k;
        if (e < n_ent) mine += (uint32_t)__popc(hm[e]);
    }
    part[tid] = mine;
    __syncthreads();
    if (wv == 0) {   // exclusive scan of the NT partial counts by one wave: NT / 64 each, then across the lanes
        uint32_t v[NT / 64], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < NT / 64; ++k) { v[k] = part[lane * (NT / 64) + k]; sum += v[k]; }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d); if ((int)lane >= d) inc += y; }
        uint32_t run = inc - sum;
#pragma unroll
        for (uint32_t k = 0; k < NT / 64; ++k) { part[lane * (NT / 64) + k] = run; run += v[k]; }
        if (lane == 63) misc[n_pre] = inc;                            // the total
    }
    __syncthreads();
    const uint32_t total = misc[n_pre];
    // per-preamble bases: the slot of entry (q, 0) = the scan value of the thread that owns it + its entries before it
    if (tid < n_pre) {
        const uint32_t e0 = tid * wpb, t0 = e0 / per;
        uint32_t b = part[t0];
        for (uint32_t e = t0 * per; e < e0; ++e) b += (uint32_t)__popc(hm[e]);
        misc[tid] = b;
    }
    __syncthreads();
    if (tid <= n_pre) {
        const uint64_t v = misc[tid];
        a.offs_pre[tid] = v;
        a.h_offs_pre[tid] = v;
    }
    if (tid == 0) *a.h_overflow = 0;
    if (total <= a.cap) {    // otherwise the host grows the buffers and searches the slot again with the regular kernels
        const uint32_t PS = g.packet_symbols, PB = g.pkt_bytes;
        uint64_t *hb_d = reinterpret_cast<uint64_t *>(a.out), *hb_h = reinterpret_cast<uint64_t *>(a.h_out);
        uint32_t *hi_d = reinterpret_cast<uint32_t *>(a.out + (size_t)total * 8), *hi_h = reinterpret_cast<uint32_t *>(a.h_out + (size_t)total * 8);
        uint32_t slot = part[tid];
        uint32_t *pos = reinterpret_cast<uint32_t *>(mag);            // the magnitudes are dead: positions of the hits, by slot
        for (uint32_t k = 0; k < per; ++k) {
            const uint32_t e = tid * per + k;
            if (e >= n_ent) break;
            uint32_t M = hm[e];
            const uint32_t w = e & (wpb - 1);
            while (M) {
                const uint32_t t = (uint32_t)__clz((int)M);
                M &= ~(0x80000000u >> t);
                const uint32_t idx = (w << 5) + t;
                hb_d[slot] = a.block_base; hb_h[slot] = a.block_base;   // Decode call of the hit
                hi_d[slot] = idx; hi_h[slot] = idx;                     // Data.Idx (decode.go:371)
                if (slot < 2 * n_sig) pos[slot] = idx;
                ++slot;
            }
        }
        __syncthreads();
        // Slice (decode.go:363-366): byte b of hit j = symbols 8b .. 8b + 7, first symbol in the MSB; a last byte of fewer
        // than 8 symbols is right-aligned.  Consecutive threads write consecutive bytes.
        uint8_t *pk_d = a.out + (size_t)total * 12, *pk_h = a.h_out + (size_t)total * 12;
        const uint32_t n_bytes = total * PB;
        const bool in_lds = total <= 2 * n_sig;                       // always, unless a block is nearly all hits
        for (uint32_t i = tid; i < n_bytes; i += NT) {
            const uint32_t j = i / PB, b = i - j * PB;
            const uint32_t idx = in_lds ? pos[j] : hi_d[j];
            const uint32_t nsym = PS - 8 * b < 8 ? PS - 8 * b : 8;
            uint32_t byte = 0;
            for (uint32_t k = 0; k < nsym; ++k) {
                const uint32_t o = idx + (8 * b + k) * SL;
                byte = (byte << 1) | ((qb[o >> 5] >> (31 - (o & 31))) & 1u);
            }
            pk_d[i] = (uint8_t)byte; pk_h[i] = (uint8_t)byte;
        }
    }
    // ---- the ticket: everything above is visible to the host first ----
    KS_STAMP(6);
    __threadfence_system();
    __syncthreads();
    KS_STAMP(7);
    if (tid == 0) {
        if (a.adone_flag) __hip_atomic_store(a.adone_flag, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.done_flag, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace amr
