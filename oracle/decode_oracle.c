/*
 * decode_oracle.c -- CPU restatement of rtlamr's protocol.Decoder hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this.  The product path
 * (rtlamr_amd/, libamrdemod.so) never links, imports or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE'S OWN TESTS: the reference ships no test,
 * golden vector or known-answer fixture for protocol/decode.go (SURVEY.md
 * section 4), and there is no Go toolchain in the build image, so the Go binary
 * itself cannot be run.  This restatement is pinned instead against
 *   (o)  oracle/_ref: the reference's OWN Go sources translated mechanically to
 *        C++ (oracle/go2cxx) -- every committed golden and 1 040 random streams,
 *        tests/test_ref_translated.py (round 6; not the Go toolchain: DESIGN.md 2),
 *   (i)  the derived golden hashes of SURVEY.md section 8c (two independent
 *        restatements, numpy and C, agreed on them during the survey),
 *   (ii) an independent numpy restatement (oracle/np_oracle.py), and
 *   (iii) self-validating decodes: CRC-valid SCM packets recovered from the
 *        reference's own capture assets/sample.bin at its true chip length.
 *
 * Every function cites the reference lines (into /root/reference) it follows.
 * All float arithmetic is IEEE binary32, round-to-nearest-even, one rounding
 * per operation: build with -O2 -ffp-contract=off (see oracle/Makefile).
 *
 * Deliberate differences from the Go code, none of which change results:
 *   - preambles are visited in registration order (Go ranges over a map in
 *     random order, decode.go:177);
 *   - parsers are not run here: a call returns, per preamble, the ascending
 *     hit indices and the sliced packet bytes that Go hands to Parser.Parse.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_PREAMBLES 8
#define ORC_MAX_PREAMBLE_BITS 64

typedef struct orc_decoder {
    /* PacketConfig, decode.go:27-42 */
    int data_rate;
    int block_size, block_size2;
    int chip_length, symbol_length;
    int sample_rate;
    int preamble_symbols, packet_symbols;
    int preamble_length, packet_length;
    int buffer_length;

    /* Decoder buffers, decode.go:45-63 */
    float *signal;      /* BlockSize + SymbolLength */
    uint8_t *quantized; /* BufferLength, one bit per byte */
    float *csum;        /* len(signal) + 1 */
    float lut[256];
    uint8_t *pkt;       /* (PacketSymbols+7)>>3 */
    uint8_t *packed;    /* (BlockSize+PreambleLength+7)>>3 */
    int *idx_a, *idx_b; /* capacity BlockSize */
    int pkt_bytes;
    int packed_len;

    int n_preambles;
    int preamble_len[ORC_MAX_PREAMBLES];
    uint8_t preamble[ORC_MAX_PREAMBLES][ORC_MAX_PREAMBLE_BITS];
    int allocated;
} orc_decoder;

/* decode.go:65-71: a fresh decoder has an all-zero config. */
orc_decoder *orc_new(void) { return (orc_decoder *)calloc(1, sizeof(orc_decoder)); }

void orc_free(orc_decoder *d)
{
    if (!d) return;
    free(d->signal); free(d->quantized); free(d->csum); free(d->pkt);
    free(d->packed); free(d->idx_a); free(d->idx_b); free(d);
}

static int imax(int a, int b) { return a > b ? a : b; }

/*
 * decode.go:100-128 RegisterProtocol: field-wise max over the registered
 * parsers' configs; the preamble string becomes 0/1 bytes; parsers sharing a
 * preamble share one Search.  Returns the preamble id (existing id if the same
 * preamble string was registered before), or -1 on error.
 */
int orc_register(orc_decoder *d, const char *preamble, int data_rate, int chip_length,
                 int preamble_symbols, int packet_symbols)
{
    int len = (int)strlen(preamble);
    if (len <= 0 || len > ORC_MAX_PREAMBLE_BITS) return -1;
    d->data_rate = imax(d->data_rate, data_rate);
    d->chip_length = imax(d->chip_length, chip_length);
    d->preamble_symbols = imax(d->preamble_symbols, preamble_symbols);
    d->packet_symbols = imax(d->packet_symbols, packet_symbols);

    uint8_t bits[ORC_MAX_PREAMBLE_BITS];
    for (int i = 0; i < len; i++) bits[i] = preamble[i] == '1';
    for (int p = 0; p < d->n_preambles; p++)
        if (d->preamble_len[p] == len && memcmp(d->preamble[p], bits, (size_t)len) == 0) return p;
    if (d->n_preambles == ORC_MAX_PREAMBLES) return -1;
    memcpy(d->preamble[d->n_preambles], bits, (size_t)len);
    d->preamble_len[d->n_preambles] = len;
    return d->n_preambles++;
}

/* decode.go:377-379 NextPowerOf2: 1 << ceil(log2(v)), evaluated in float64. */
int orc_next_power_of_2(int v) { return 1 << (unsigned)ceil(log2((double)v)); }

/*
 * decode.go:209-216 NewMagLUT.  Go evaluates (127.5 - float32(idx)) / 127.5 in
 * float32 (the untyped constants convert to float32), stores it, then squares
 * it in float32: two roundings per entry.
 */
void orc_mag_lut(float lut[256])
{
    for (int i = 0; i < 256; i++) {
        float q = (127.5f - (float)i) / 127.5f;
        lut[i] = q * q;
    }
}

/* decode.go:131-160 Allocate: geometry, zeroed buffers, LUT. */
int orc_allocate(orc_decoder *d)
{
    if (d->chip_length <= 0 || d->preamble_symbols <= 0 || d->packet_symbols <= 0) return -1;
    d->symbol_length = d->chip_length << 1;
    d->sample_rate = d->data_rate * d->chip_length;
    d->preamble_length = d->preamble_symbols * d->symbol_length;
    d->packet_length = d->packet_symbols * d->symbol_length;
    d->block_size = orc_next_power_of_2(d->preamble_length);
    d->block_size2 = d->block_size << 1;
    d->buffer_length = d->packet_length + d->block_size;

    int sig_len = d->block_size + d->symbol_length;
    d->signal = (float *)calloc((size_t)sig_len, sizeof(float));
    d->quantized = (uint8_t *)calloc((size_t)d->buffer_length, 1);
    d->csum = (float *)calloc((size_t)sig_len + 1, sizeof(float));
    orc_mag_lut(d->lut);
    d->pkt_bytes = (d->packet_symbols + 7) >> 3;
    d->pkt = (uint8_t *)calloc((size_t)d->pkt_bytes, 1);
    d->idx_a = (int *)calloc((size_t)d->block_size, sizeof(int));
    d->idx_b = (int *)calloc((size_t)d->block_size, sizeof(int));
    d->packed_len = (d->block_size + d->preamble_length + 7) >> 3;
    d->packed = (uint8_t *)calloc((size_t)d->packed_len, 1);
    d->allocated = 1;
    return 0;
}

/* geometry read-back for the tests: order documented in oracle/oracle.py */
void orc_geometry(const orc_decoder *d, int out[12])
{
    out[0] = d->data_rate; out[1] = d->chip_length; out[2] = d->symbol_length;
    out[3] = d->sample_rate; out[4] = d->preamble_symbols; out[5] = d->packet_symbols;
    out[6] = d->preamble_length; out[7] = d->packet_length; out[8] = d->block_size;
    out[9] = d->block_size2; out[10] = d->buffer_length; out[11] = d->n_preambles;
}
const float *orc_signal(const orc_decoder *d) { return d->signal; }
const uint8_t *orc_quantized(const orc_decoder *d) { return d->quantized; }
const float *orc_lut(const orc_decoder *d) { return d->lut; }
int orc_pkt_bytes(const orc_decoder *d) { return d->pkt_bytes; }

/* decode.go:219-225 MagLUT.Execute: out[i] = lut[in[2i]] + lut[in[2i+1]]. */
static void mag_execute(const float *lut, const uint8_t *in, float *out, int n)
{
    for (int i = 0; i < n; i++) out[i] = lut[in[2 * i]] + lut[in[2 * i + 1]];
}

/*
 * decode.go:229-245 Filter: sequential float32 running sum restarted at zero
 * (csum[0] = 0), then f = (csum[i+CL]-csum[i]) - (csum[i+SL]-csum[i+CL]) and
 * out[i] = 1 - signbit(f), n_out outputs.
 */
static void filter(orc_decoder *d, const float *in, int n_in, uint8_t *out, int n_out)
{
    float sum = 0.0f;
    d->csum[0] = 0.0f;
    for (int i = 0; i < n_in; i++) {
        sum = sum + in[i];
        d->csum[i + 1] = sum;
    }
    const float *lower = d->csum + d->chip_length;
    const float *upper = d->csum + d->symbol_length;
    for (int i = 0; i < n_out; i++) {
        float a = lower[i] - d->csum[i];
        float b = upper[i] - lower[i];
        float f = a - b;
        uint32_t bits;
        memcpy(&bits, &f, 4);
        out[i] = (uint8_t)(1u - (bits >> 31));
    }
}

/*
 * decode.go:255-328 Search, literally: pack, byte-granular prefilter at offsets
 * pIdx*(SymbolLength>>3), expand x8, exact pass at stride SymbolLength.
 * Returns the number of surviving indices (ascending) written to out.
 */
static int search_literal(orc_decoder *d, const uint8_t *pre, int plen, int *out)
{
    int sym_len_byte = d->symbol_length >> 3;
    /* :259-265 pack MSB first */
    for (int b = 0; b < d->packed_len; b++) {
        uint8_t v = 0;
        for (int k = 0; k < 8; k++) v = (uint8_t)((v << 1) | d->quantized[(b << 3) + k]);
        d->packed[b] = v;
    }
    int *a = d->idx_a, *bb = d->idx_b;
    int na = 0, nb = 0;
    /* :268-294 */
    for (int p = 0; p < plen; p++) {
        uint8_t skip = (uint8_t)((pre[p] ^ 1) * 0xFF);
        int offset = p * sym_len_byte;
        if (p == 0) {
            na = 0;
            for (int q = 0; q < (d->block_size >> 3); q++)
                if (d->packed[q] != skip) a[na++] = q;
        } else {
            nb = 0;
            for (int i = 0; i < na; i++)
                if (d->packed[offset + a[i]] != skip) bb[nb++] = a[i];
            int *t = a; a = bb; bb = t;
            na = nb;
            if (na == 0) return 0;
        }
    }
    /* :299-310 expand byte indices to bit indices */
    nb = 0;
    for (int i = 0; i < na; i++)
        for (int k = 0; k < 8; k++) bb[nb++] = (a[i] << 3) + k;
    { int *t = a; a = bb; bb = t; na = nb; }
    /* :313-325 exact pass */
    for (int p = 0; p < plen; p++) {
        const uint8_t *sig = d->quantized + p * d->symbol_length;
        nb = 0;
        for (int i = 0; i < na; i++)
            if (sig[a[i]] == pre[p]) bb[nb++] = a[i];
        int *t = a; a = bb; bb = t;
        na = nb;
        if (na == 0) return 0;
    }
    memcpy(out, a, (size_t)na * sizeof(int));
    return na;
}

/* Semantic search: idx in [0,BlockSize) with Quantized[idx+p*SL]==pre[p] for all p.
 * Equal to search_literal whenever SymbolLength is a multiple of 8 (all legal
 * -symbollength values); used to cross-check the literal version. */
static int search_semantic(const orc_decoder *d, const uint8_t *pre, int plen, int *out)
{
    int n = 0;
    for (int idx = 0; idx < d->block_size; idx++) {
        int ok = 1;
        for (int p = 0; p < plen && ok; p++) ok = d->quantized[idx + p * d->symbol_length] == pre[p];
        if (ok) out[n++] = idx;
    }
    return n;
}

/*
 * decode.go:353-375 Slice + parse.go:61-69 NewData.  d->pkt is shifted, never
 * cleared (as in Go), so when PacketSymbols%8 != 0 the last byte keeps stale
 * high bits from earlier hits.  The guard idx > BlockSize (:358) is kept.
 */
static int slice(orc_decoder *d, const int *idx, int n, uint8_t *out_bytes)
{
    int m = 0;
    for (int i = 0; i < n; i++) {
        if (idx[i] > d->block_size) continue;
        for (int p = 0; p < d->packet_symbols; p++) {
            d->pkt[p >> 3] = (uint8_t)(d->pkt[p >> 3] << 1);
            d->pkt[p >> 3] |= d->quantized[idx[i] + p * d->symbol_length];
        }
        memcpy(out_bytes + (size_t)m * d->pkt_bytes, d->pkt, (size_t)d->pkt_bytes);
        m++;
    }
    return m;
}

/*
 * decode.go:163-197 Decode for one block of BlockSize2 bytes.
 *   hit_count[p]           number of hits of preamble p
 *   hit_idx[p*cap + i]     ascending indices (cap = hit_cap ints per preamble)
 *   hit_bytes[(p*cap+i)*pkt_bytes ...]  sliced packet bytes
 * mode 0 = literal Search, 1 = semantic search.
 * Returns 0, or -1 when a preamble produced more than hit_cap hits.
 */
int orc_decode(orc_decoder *d, const uint8_t *input, int mode, int hit_cap,
               int *hit_count, int *hit_idx, uint8_t *hit_bytes)
{
    int bs = d->block_size, sl = d->symbol_length, pl = d->packet_length;
    /* :165-166 slide histories */
    memmove(d->signal, d->signal + bs, (size_t)sl * sizeof(float));
    memmove(d->quantized, d->quantized + bs, (size_t)pl);
    /* :169 magnitude of the new block */
    mag_execute(d->lut, input, d->signal + sl, bs);
    /* :172 matched filter into Quantized[PacketLength:] */
    filter(d, d->signal, bs + sl, d->quantized + pl, bs);

    int *tmp = (int *)malloc((size_t)bs * sizeof(int));
    int rc = 0;
    for (int p = 0; p < d->n_preambles; p++) {
        int n = mode ? search_semantic(d, d->preamble[p], d->preamble_len[p], tmp)
                     : search_literal(d, d->preamble[p], d->preamble_len[p], tmp);
        if (n > hit_cap) { rc = -1; n = hit_cap; }
        hit_count[p] = n;
        if (hit_idx) memcpy(hit_idx + (size_t)p * hit_cap, tmp, (size_t)n * sizeof(int));
        if (hit_bytes)
            slice(d, tmp, n, hit_bytes + (size_t)p * hit_cap * d->pkt_bytes);
    }
    free(tmp);
    return rc;
}

/*
 * Convenience for tests and the CPU baseline: run n_blocks consecutive Decode
 * calls over a contiguous IQ buffer.
 *   qpacked (optional): the new quantized bits of every call, Quantized[PL:],
 *       packed MSB-first, n_blocks*BlockSize/8 bytes.
 *   hits (optional): records {block, preamble id, idx}, 3 ints each, in
 *       (block, preamble id, idx) order, at most hits_cap records;
 *   hit_bytes (optional): pkt_bytes per record.
 * Returns the total number of hits found (may exceed hits_cap).
 */
long orc_decode_stream(orc_decoder *d, const uint8_t *iq, long n_blocks, int mode,
                       uint8_t *qpacked, int *hits, uint8_t *hit_bytes, long hits_cap)
{
    int bs = d->block_size, pl = d->packet_length;
    int cap = bs;
    int *cnt = (int *)malloc(sizeof(int) * ORC_MAX_PREAMBLES);
    int *idx = (int *)malloc(sizeof(int) * (size_t)cap * ORC_MAX_PREAMBLES);
    uint8_t *pb = (uint8_t *)malloc((size_t)cap * ORC_MAX_PREAMBLES * (size_t)d->pkt_bytes);
    long total = 0;
    for (long k = 0; k < n_blocks; k++) {
        orc_decode(d, iq + (size_t)k * d->block_size2, mode, cap, cnt, idx, pb);
        if (qpacked) {
            uint8_t *dst = qpacked + (size_t)k * (bs >> 3);
            const uint8_t *q = d->quantized + pl;
            for (int b = 0; b < (bs >> 3); b++) {
                uint8_t v = 0;
                for (int j = 0; j < 8; j++) v = (uint8_t)((v << 1) | q[(b << 3) + j]);
                dst[b] = v;
            }
        }
        for (int p = 0; p < d->n_preambles; p++) {
            for (int i = 0; i < cnt[p]; i++) {
                if (total < hits_cap) {
                    if (hits) {
                        hits[total * 3 + 0] = (int)k;
                        hits[total * 3 + 1] = p;
                        hits[total * 3 + 2] = idx[(size_t)p * cap + i];
                    }
                    if (hit_bytes)
                        memcpy(hit_bytes + (size_t)total * d->pkt_bytes,
                               pb + ((size_t)p * cap + i) * d->pkt_bytes, (size_t)d->pkt_bytes);
                }
                total++;
            }
        }
    }
    free(cnt); free(idx); free(pb);
    return total;
}

/* ------------------------------------------------------------------------------------------------
 * r900 second stage -- literal restatement of r900/r900.go:82-150 (Parser.filter) and of the buffer
 * handling at the top of Parser.Parse (r900.go:168-170).  State lives in the caller-provided arrays:
 *   signal[BufferLength], csum[BufferLength+1], quantized[BufferLength]  (r900.go:162-165, zero-initialised).
 * Call once per Decode call, after orc_decode, like Parse is (decode.go:185-187).
 */
static float absf32(float x) { return x < 0 ? -x : x; }   /* r900.go:152-157 */

void orc_r900_filter(const orc_decoder *d, float *signal, float *csum, uint8_t *quantized)
{
    int bs = d->block_size, pl = d->packet_length, sl = d->symbol_length, bl = d->buffer_length;
    memmove(signal, signal + bs, (size_t)(bl - bs) * sizeof(float));          /* :168 copy(p.signal, p.signal[BlockSize:]) */
    memcpy(signal + pl, d->signal + sl, (size_t)bs * sizeof(float));          /* :169-170 */

    float sum = 0;                                                            /* :96-100 */
    for (int idx = 0; idx < bl; idx++) {
        sum += signal[idx];
        csum[idx + 1] = sum;
    }
    int cl = d->chip_length, cl2 = cl * 2, cl3 = cl * 3, cl4 = cl * 4;        /* :112-116 */
    int limit = bl - cl4;
    for (int idx = 0; idx < limit; idx++) {                                   /* :121-149 */
        float c0 = csum[idx];
        float c1 = csum[idx + cl] + csum[idx + cl];
        float c2 = csum[idx + cl2] + csum[idx + cl2];
        float c3 = csum[idx + cl3] + csum[idx + cl3];
        float c4 = csum[idx + cl4];
        float a0 = c2 - c4 - c0;
        float a1 = c1 - c2 + c3 - c4 - c0;
        float a2 = c1 - c3 + c4 - c0;
        float max_abs = absf32(a0);
        uint8_t argmax = 0;
        if (absf32(a1) > max_abs) { max_abs = absf32(a1); argmax = 1; }
        if (absf32(a2) > max_abs) { max_abs = absf32(a2); argmax = 2; }
        float v[3] = {a0, a1, a2};
        quantized[idx] = argmax;
        if (v[argmax] > 0) quantized[idx] += 3;
    }
}
