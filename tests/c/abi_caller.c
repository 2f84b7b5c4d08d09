/*
 * A plain C caller of include/amrdemod.h -- what a cgo binding is underneath.  Built with gcc
 * (no HIP, no C++): proves the header is C, the library links from C, and the entry points behave through raw
 * pointers.  tests/test_gpu_c_caller.py runs it on the GPU box and compares the printed result with the Python
 * mirror on the same synthetic stream.
 *
 *   abi_caller <chip_length> <n_blocks> <seed>
 * prints: geometry, hit count, FNV-1a of (block, idx, packet bytes) over all hits, FNV-1a of the packed bitstream.
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "amrdemod.h"

#define CHECK(expr)                                                                        \
    do {                                                                                   \
        amr_status s_ = (expr);                                                            \
        if (s_ != AMR_OK) {                                                                \
            fprintf(stderr, "%s -> %d (%s: %s)\n", #expr, s_, amr_strerror(s_), amr_last_error()); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

static uint64_t fnv(uint64_t h, const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s chip n_blocks seed\n", argv[0]); return 2; }
    const int chip = atoi(argv[1]);
    const size_t n_blocks = (size_t)atoll(argv[2]);
    const uint64_t seed = (uint64_t)atoll(argv[3]);

    /* scm.NewParser's PacketConfig, scm/scm.go:42-50 */
    amr_protocol scm = {"111110010101001100000", 32768, chip, 21, 96};
    amr_handle *h = NULL;
    CHECK(amr_create(&scm, 1, 0, &h));
    amr_geometry g;
    CHECK(amr_get_geometry(h, &g));
    printf("geometry bs=%d pl=%d buf=%d pkt_bytes=%d\n", g.block_size, g.packet_length, g.buffer_length, g.pkt_bytes);

    /* synthetic IQ on the device, one planted packet (bytes given by the caller side of the test) */
    const uint64_t n_samples = (uint64_t)n_blocks * (uint64_t)g.block_size;
    void *d_iq = NULL;
    CHECK(amr_dev_alloc(0, n_samples * 2, &d_iq));
    CHECK(amr_synth_noise(0, d_iq, n_samples, seed, 0));
    const uint8_t pkt[12] = {0xf9, 0x53, 0x02, 0x61, 0x01, 0xb3, 0x36, 0x0c, 0x41, 0x05, 0xd0, 0x05};  /* SURVEY.md 8c */
    const uint64_t start[2] = {(uint64_t)g.block_size * 3 + 17, (uint64_t)g.block_size * (n_blocks / 2) - 500};
    uint8_t bits[24];
    memcpy(bits, pkt, 12);
    memcpy(bits + 12, pkt, 12);
    const int8_t di[2] = {30, -30}, dq[2] = {-26, 26};
    CHECK(amr_synth_plant(0, d_iq, n_samples, 0, chip, 2, start, bits, 96, 12, di, dq));

    /* the same stream twice: one batch, then host memory in two uneven batches after a reset */
    amr_result r;
    CHECK(amr_decode_batch_device(h, d_iq, n_blocks, &r));
    uint64_t hh = 14695981039346656037ull;
    for (uint64_t i = 0; i < r.n_hits; ++i) {
        hh = fnv(hh, &r.hit_block[i], 8);
        hh = fnv(hh, &r.hit_idx[i], 4);
        hh = fnv(hh, r.pkt + i * r.pkt_bytes, r.pkt_bytes);
    }
    const uint64_t n1 = r.n_hits;
    size_t qbytes = (size_t)n_samples / 8;
    uint8_t *q = (uint8_t *)malloc(qbytes);
    CHECK(amr_copy_quantized(h, q, qbytes));
    printf("device hits=%" PRIu64 " hit_hash=%016" PRIx64 " q_hash=%016" PRIx64 "\n", n1, hh, fnv(14695981039346656037ull, q, qbytes));

    uint8_t *host = (uint8_t *)malloc(n_samples * 2);
    CHECK(amr_dev_download(0, host, d_iq, n_samples * 2));
    CHECK(amr_reset(h));
    const size_t first = n_blocks / 3;
    uint64_t hh2 = 14695981039346656037ull, n2 = 0;
    for (int part = 0; part < 2; ++part) {
        const size_t b0 = part ? first : 0, nb = part ? n_blocks - first : first;
        CHECK(amr_decode_batch(h, host + b0 * (size_t)g.block_size2, nb * (size_t)g.block_size2, nb, &r));
        for (uint64_t i = 0; i < r.n_hits; ++i) {
            hh2 = fnv(hh2, &r.hit_block[i], 8);
            hh2 = fnv(hh2, &r.hit_idx[i], 4);
            hh2 = fnv(hh2, r.pkt + i * r.pkt_bytes, r.pkt_bytes);
        }
        n2 += r.n_hits;
    }
    printf("host   hits=%" PRIu64 " hit_hash=%016" PRIx64 "\n", n2, hh2);

    /* error behaviour across the boundary: status codes, never a crash */
    printf("short_input=%d null_handle=%d\n", amr_decode_batch(h, host, 10, 1, &r), amr_reset(NULL));
    free(host);
    free(q);
    CHECK(amr_dev_free(0, d_iq));
    CHECK(amr_destroy(h));
    return (n1 == n2 && hh == hh2) ? 0 : 3;
}
