/*
 * amrdemod.h -- C ABI of the MI355X (gfx950) implementation of rtlamr's
 * IQ-to-bits hot path, protocol.Decoder.Decode (reference: protocol/decode.go).
 *
 * The reference has no FFI boundary of its own: its boundary is the Go API of
 * package protocol.  This header is what a cgo binding of that package binds
 * (INTEGRATION.md shows the Go side).  Each entry point names the reference
 * interface it replaces (file:line into the rtlamr tree).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ or torch types.
 *   - every function returns an amr_status (0 = ok, negative = error); no
 *     exception and no Go-style panic crosses the boundary.
 *   - the library never keeps a caller HOST pointer after the call returns (cgo
 *     pointer-passing rule); result arrays are owned by the handle (pinned host
 *     memory) and stay valid until the next amr_decode_* / second amr_submit_device /
 *     amr_reset / amr_destroy on it.
 *   - one handle = one reference Decoder = one GPU; like the Go Decoder
 *     (decode.go:163, shared slices) a handle is not re-entrant: one caller
 *     thread at a time.
 *
 * Batch semantics: amr_decode_batch(h, iq, n) is exactly n consecutive calls
 * Decoder.Decode(iq[k*BlockSize2 : (k+1)*BlockSize2]) (main.go:235), including
 * the SymbolLength-sample magnitude history and the PacketLength-bit quantized
 * history the Go decoder carries between calls (decode.go:165-166).
 */
#ifndef AMRDEMOD_H
#define AMRDEMOD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int amr_status;
#define AMR_OK 0
#define AMR_EINVAL (-1)   /* bad argument / illegal chip length / bad preamble */
#define AMR_ENOMEM (-2)   /* host or device allocation failed */
#define AMR_EHIP (-3)     /* a HIP runtime call failed; amr_last_error() has the text */
#define AMR_ENODEV (-4)   /* no usable gfx950 device: the product path has NO CPU fallback */
#define AMR_EOVERFLOW (-5)/* internal capacity exceeded and could not be grown */

#define AMR_MAX_PREAMBLES 8
#define AMR_MAX_PREAMBLE_BITS 64

typedef struct amr_handle amr_handle;

/*
 * The fields of protocol.PacketConfig that Decoder.RegisterProtocol reads from
 * a parser's Cfg() (decode.go:27-42, 100-128).  One per registered parser.
 */
typedef struct amr_protocol {
    const char *preamble;  /* ASCII '0'/'1', PacketConfig.Preamble */
    int32_t data_rate;     /* PacketConfig.DataRate (32768 for every rtlamr parser) */
    int32_t chip_length;   /* PacketConfig.ChipLength = -symbollength flag (main.go:77) */
    int32_t preamble_symbols;
    int32_t packet_symbols;
} amr_protocol;

/* Geometry computed exactly as Decoder.Allocate (decode.go:131-141). */
typedef struct amr_geometry {
    int32_t data_rate, chip_length, symbol_length, sample_rate;
    int32_t preamble_symbols, packet_symbols;  /* field-wise max, decode.go:105-109 */
    int32_t preamble_length, packet_length;
    int32_t block_size, block_size2, buffer_length;
    int32_t n_preambles;   /* distinct preambles = number of Search calls per Decode */
    int32_t pkt_bytes;     /* (PacketSymbols+7)>>3, decode.go:151 */
} amr_geometry;

/*
 * Result of one batch.  Hits are grouped by preamble id (registration order of
 * first appearance), and inside a preamble sorted by (block, idx) ascending --
 * the order Decoder.Search returns per call (decode.go:327).
 *   hit_block[i]  index of the Decode call that reports the hit, counted from
 *                 the last amr_reset (plus amr_set_block_base)
 *   hit_idx[i]    Data.Idx, the index into Decoder.Quantized (decode.go:371)
 *   pkt[i*pkt_bytes .. ]  Data.Bytes as built by Decoder.Slice (decode.go:363-366),
 *                 every bit of it: when PacketSymbols%8 != 0 (r900 alone or with
 *                 scm) Go never clears the high bits of the last byte, which then
 *                 hold symbols of the hits sliced before -- reproduced here in the
 *                 order call, preamble id, idx, across calls and batches, and across
 *                 shards when the caller hands the byte on (amr_get_stale_carry /
 *                 amr_set_stale_carry below).
 */
typedef struct amr_result {
    uint32_t n_preambles;
    uint32_t pkt_bytes;
    uint64_t n_hits;
    const uint64_t *preamble_offset;  /* [n_preambles+1]: hits of preamble p are [off[p], off[p+1]) */
    const uint64_t *hit_block;        /* [n_hits] */
    const uint32_t *hit_idx;          /* [n_hits] */
    const uint8_t *pkt;               /* [n_hits * pkt_bytes] */
    /*
     * r900 second stage (amr_r900_enable): for every hit of the r900 preamble, in the order of that preamble's
     * hits, the 42 base-6 digits r900.Parser.Parse reads from its quantized buffer (r900.go:187-193).
     */
    int32_t r900_preamble;            /* preamble id the digits belong to, -1 = not enabled */
    const uint8_t *r900_digits;       /* [(off[r900_preamble+1]-off[r900_preamble]) * 42] */
    /* hits the search found; n_hits is smaller only when amr_set_validation dropped some on the device */
    uint64_t n_hits_searched;
    /*
     * The Decode calls this result covers: [first_block, first_block + n_blocks), call indices as in hit_block.
     * Equal to the batch that was submitted unless amr_set_deferral is on (see there).
     */
    uint64_t first_block;
    uint64_t n_blocks;
} amr_result;

/*
 * Timing of the last batch, measured with HIP events on the kernel dispatches.
 *   Synchronous callers (amr_decode_batch[_device]; one stream): the whole batch -- demod_ms = K1, search_ms = K2 through
 *   the last kernel of the tail (K3 slice, K4 r900 digits, K5 validation), total_ms = first kernel start to last kernel end.
 *   Pipelined callers (two or more batches in flight): the tail of a batch runs on a second stream next to
 *   the FOLLOWING batch's kernels and has no duration that could be added to a step.  The fields then describe what the
 *   batch cost the compute stream: demod_ms = K1, search_ms = K2 only, total_ms = demod_ms + search_ms -- whether the
 *   tail was enqueued ahead behind its gate or launched by the host.  A whole-path figure comes from wall-clock time over
 *   many batches (bench.py: steady_ms_per_step), not from these.
 */
typedef struct amr_timing {
    float demod_ms;   /* K1: magnitude + csum matched filter + quantize + pack */
    float search_ms;  /* synchronous: K2 + K3 (+ K4, K5); pipelined: K2 only (see above) */
    float total_ms;   /* synchronous: first kernel start to last kernel end; pipelined: demod_ms + search_ms */
} amr_timing;

/* ---- lifecycle -------------------------------------------------------- */

/*
 * NewDecoder + RegisterProtocol for each entry + Allocate
 * (decode.go:65, 100-128, 131-160).  chip_length must be one of the values
 * flags.go:127-132 accepts (8,32,40,...,96) and equal for all entries.
 * device_id: HIP device ordinal.  Fails with AMR_ENODEV when the device is
 * missing or is not gfx950.
 */
amr_status amr_create(const amr_protocol *protos, int32_t n_protos, int32_t device_id, amr_handle **out);
amr_status amr_destroy(amr_handle *h);
/* gfx950 devices visible to this process (0 is an answer, not an error): what a host that starts one process per GPU
 *  asks before it spawns its ranks; main.go has no counterpart (one rtl_tcp stream, one Decoder). */
amr_status amr_device_count(int32_t *n_devices);

/*
 * The arithmetic of RegisterProtocol + Allocate alone (decode.go:100-141), without a device: the geometry amr_create
 * would arrive at, and (preamble_ids != NULL, n_protos entries) the preamble group of every entry.  Lets a binding
 * validate a protocol set and size its buffers before a GPU is touched; same argument checks as amr_create.
 */
amr_status amr_plan(const amr_protocol *protos, int32_t n_protos, amr_geometry *geom, int32_t *preamble_ids);

/* Forget all history: equivalent to a freshly allocated Decoder (decode.go:144-145 zero buffers). */
amr_status amr_reset(amr_handle *h);

/* PacketConfig read-back (Decoder.Cfg, decode.go:46). */
amr_status amr_get_geometry(const amr_handle *h, amr_geometry *out);
/* Preamble id a registered protocol (by registration index) was grouped under (decode.go:124). */
int32_t amr_preamble_id(const amr_handle *h, int32_t proto_index);
/* The float32 magnitude table, NewMagLUT (decode.go:209-216): 256 floats. */
amr_status amr_get_mag_lut(const amr_handle *h, float *out256);

/*
 * The r900 parser's second matched filter (r900/r900.go:82-150) on the GPU, evaluated at the preamble hits only.
 * proto_index = registration index of the r900 parser's entry; its preamble's hits then carry 42 digits each
 * (amr_result.r900_digits).  Call after amr_create, before the first batch.
 */
amr_status amr_r900_enable(amr_handle *h, int32_t proto_index);

/*
 * Per-hit validation on the GPU (optional): what every rtlamr parser does first with a packet, done before the
 * hits leave the device, so that the read-back (and the multi-GPU hit gather) carries only hits a parser can
 * turn into a message.  A hit of preamble_id is dropped when one of the CRC checks fails, or when its first
 * dedupe_bytes packet bytes equal those of the hit right before it in the same block (Parse's `seen` map drops
 * those too).  The parsers run unchanged on what is left and emit the same messages.
 *   check = crc.Checksum (crc/crc.go:49-55) over one or two byte spans of the packet, compared with residue:
 *     SCM     scm/scm.go:77           {0x0000, 0x6F63, 0x0000, spans {2,10}}
 *     SCM+    scmplus/scmplus.go:77   {0xFFFF, 0x1021, 0x1D0F, spans {2,14}}
 *     IDM, NetIDM  idm/idm.go:77-87, netidm/netidm.go:88-98
 *                                     {0xFFFF, 0x1021, 0x1D0F, spans {4,88}} and {.., spans {9,4},{88,2}}
 * v == NULL switches the preamble's validation off.  Not allowed on the r900 preamble (its hits carry digits).
 * Call with no batch in flight.
 */
typedef struct amr_crc_check {
    uint16_t init, poly, residue;
    uint16_t n_spans;                 /* 1 or 2 */
    uint16_t span_off[2], span_len[2];
} amr_crc_check;
typedef struct amr_validator {
    int32_t n_checks;                 /* 0..2, all must pass */
    int32_t dedupe_bytes;             /* 0 = keep repeats; else the parser's own packet length in bytes */
    amr_crc_check checks[2];
} amr_validator;
amr_status amr_set_validation(amr_handle *h, int32_t preamble_id, const amr_validator *v);

/* Run on a caller-owned HIP stream (hipStream_t passed as void*); NULL = the handle's own stream. */
amr_status amr_set_stream(amr_handle *h, void *hip_stream);
/* Add a constant to reported hit_block (multi-GPU: first call index owned by this shard). */
amr_status amr_set_block_base(amr_handle *h, uint64_t base);

/* ---- the hot path ------------------------------------------------------ */

/*
 * Decoder.Decode (decode.go:163-197) for n_blocks consecutive blocks held in
 * HOST memory (n_blocks * BlockSize2 bytes).  Copies the batch to the device,
 * runs the kernels, returns hits and packets.  Parsers are NOT run here: the
 * binding hands each preamble's packets to its parsers as decode.go:177-187.
 * iq_bytes < n_blocks*BlockSize2 is AMR_EINVAL (Go panics at decode.go:222);
 * extra bytes are ignored, as in Go.
 */
amr_status amr_decode_batch(amr_handle *h, const uint8_t *iq, size_t iq_bytes, size_t n_blocks, amr_result *res);

/* Same, input already resident in device memory (hipMalloc'd or a torch tensor's data_ptr). */
amr_status amr_decode_batch_device(amr_handle *h, const void *d_iq, size_t n_blocks, amr_result *res);

/*
 * Pipelined form for throughput-oriented callers (file replay, many-SDR aggregation): submit
 * enqueues a batch and returns immediately, collect waits for the OLDEST submitted batch and returns
 * its result.  At most three batches may be in flight; the GPU then runs batches i+1 and i+2 while the host
 * reads back and parses batch i, and with two or more in flight the slicing of batch i (K3) shares the GPU with the
 * search of batch i+1 instead of waiting in front of its demodulation.  amr_decode_batch_device == submit + collect.  A result stays valid until
 * the second submit after the collect that returned it.  d_iq must stay untouched until collected.
 */
amr_status amr_submit_device(amr_handle *h, const void *d_iq, size_t n_blocks);
amr_status amr_collect(amr_handle *h, amr_result *res);

/*
 * Wave quantisation inside the library (main.go:166,235: the caller chooses the block count, not the library).
 * The demodulation kernel works in wave-tiles of 64 blocks; a batch that does not end on one would end in a lone
 * wavefront that takes as long as a whole chip-filling launch (100 000 blocks of SCM: 0.405 ms per step against 0.20
 * for 98 304).  With deferral ON, amr_submit_device / amr_submit_host process a batch up to its last whole wave-tile
 * and carry the (at most 63) blocks behind it into the next submit's launch; the library keeps its own copy of those
 * blocks, so the caller's buffer is still free after the collect.  Nothing is lost or reordered: every hit carries the
 * index of its Decode call, and amr_result.first_block / n_blocks say which calls a result covers -- the tail of
 * batch i simply arrives with batch i+1.  amr_flush processes what is still deferred at the end of the stream (call it
 * with nothing in flight) and returns its hits.  amr_decode_batch[_device] never defer: they process every block
 * handed over so far, deferred ones included.  Off by default; not available with amr_r900_enable.
 */
amr_status amr_set_deferral(amr_handle *h, int32_t on);
amr_status amr_flush(amr_handle *h, amr_result *res);

/*
 * The same pipeline for input in HOST memory (file replay, many-SDR aggregation; the caller the reference
 * has is Receiver.Run, main.go:156-235): the batch is copied to a per-slot device buffer on a transfer
 * stream, the kernels wait for that copy only, so the host-to-device transfer of batch i+1 overlaps the
 * kernels of batch i.  iq must stay untouched until the batch is collected; memory from amr_host_alloc
 * (pinned) makes the copy a true DMA -- pageable memory works but is staged by the runtime.
 */
amr_status amr_submit_host(amr_handle *h, const uint8_t *iq, size_t iq_bytes, size_t n_blocks);
amr_status amr_host_alloc(size_t bytes, void **ptr);
amr_status amr_host_free(void *ptr);

/*
 * Device-side view of the result amr_collect / amr_decode_* returned last: the packed buffer
 * [hit_block u64 x n | hit_idx u32 x n | pkt x n] in device memory (the validated list when
 * amr_set_validation is active), for consumers that stay on the GPU
 * (the multi-GPU hit gather sends the first 12*n bytes over RCCL without a host round trip).
 * Valid until the second amr_submit_device after that collect.
 */
amr_status amr_result_device(const amr_handle *h, const void **d_packed, uint64_t *n_hits);

/*
 * Multi-GPU sharding: feed the ceil(PacketLength/BlockSize)+1
 * blocks that precede a shard so its magnitude / quantized history equals what
 * a single decoder would hold, without reporting hits or advancing the block
 * counter.  halo_iq points at the first warm-up block; lead (may be NULL = zeros)
 * at the aligned-halo bytes (amr_halo_bytes) that precede it in the stream.
 */
amr_status amr_prime(amr_handle *h, const uint8_t *lead, const uint8_t *halo_iq, size_t n_blocks, int on_device);
/*
 * The one piece of Decoder state a block-range split cannot rebuild from the blocks in front of a shard: d.pkt, which
 * Decoder.Slice (decode.go:353-375) never clears.  With PacketSymbols % 8 = r != 0 (r900 alone or with scm: r = 4) the
 * last byte of a packet holds, above its r fresh bits, bits of the hits sliced before it -- possibly found long before
 * the shard begins.  amr_get_stale_carry returns the last byte of the last hit this decoder has sliced so far (0 for a
 * fresh one); amr_set_stale_carry makes it the predecessor of the next batch's first hit.  A caller that runs its shards
 * one after another passes the byte from shard r-1 to shard r (after amr_prime) and gets the single decoder's bytes,
 * every bit.  Shards that run at the same time cannot know it in advance: they start from zero and the caller patches
 * afterwards -- only the first ceil(8/r) - 1 hits of a shard (in slicing order: call, preamble id, idx) are affected,
 * hit i of them by   last_byte |= (carry << (r * (i + 1))) & 0xff   (rtlamr_amd/dist.py: patch_stale_carry).
 * No batch may be in flight.  With PacketSymbols % 8 == 0 the byte is never used.
 */
amr_status amr_get_stale_carry(amr_handle *h, uint8_t *last_byte);
amr_status amr_set_stale_carry(amr_handle *h, uint8_t last_byte);
size_t amr_halo_bytes(const amr_handle *h);
size_t amr_prime_blocks(const amr_handle *h);

/* ---- introspection used by the parity tests ---------------------------- */

/*
 * The new quantized bits of the last batch -- Decoder.Quantized[PacketLength:]
 * after each call (decode.go:172) -- packed MSB-first as Search packs them
 * (decode.go:259-265): n_blocks*BlockSize/8 bytes.
 */
amr_status amr_copy_quantized(amr_handle *h, uint8_t *out, size_t out_bytes);
/*
 * Device-side timing.  A HIP event on the stream costs a ~5 us bubble, so none is recorded unless asked for:
 * level 0 = none (default), 1 = K1 start/stop (demod_ms), 2 = also the search (search_ms, total_ms).
 * amr_get_timing returns the numbers of the last collected batch (AMR_EINVAL when the level was 0).
 */
amr_status amr_set_timing(amr_handle *h, int32_t level);
amr_status amr_get_timing(const amr_handle *h, amr_timing *out);
const char *amr_strerror(amr_status s);
const char *amr_last_error(void);
/* Library / device description for logs: "amrdemod <ver> gfx950 <n> CUs ..." */
amr_status amr_describe(const amr_handle *h, char *buf, size_t buf_bytes);

/* ---- device utilities for bench and tests (not part of the decode path) -- */

amr_status amr_dev_alloc(int32_t device_id, size_t bytes, void **d_ptr);
amr_status amr_dev_free(int32_t device_id, void *d_ptr);
amr_status amr_dev_upload(int32_t device_id, void *d_dst, const void *src, size_t bytes);
amr_status amr_dev_download(int32_t device_id, void *dst, const void *d_src, size_t bytes);
amr_status amr_dev_sync(int32_t device_id);

/*
 * Deterministic integer-only synthetic IQ: sample n of the
 * stream gets h = splitmix64(seed ^ n), I = 119 + popcount(h & 0xFFFF),
 * Q = 120 + popcount((h >> 16) & 0xFFFF).  first_sample = stream index of d_iq[0].
 */
amr_status amr_synth_noise(int32_t device_id, void *d_iq, uint64_t n_samples, uint64_t seed, uint64_t first_sample);
/* The second distribution of SURVEY.md 8d: uniform random bytes, I = bits 32..39, Q = bits 40..47 of the same hash
 * (every magnitude-LUT entry equally likely: the worst case for the LUT gathers of the demodulation kernel). */
amr_status amr_synth_uniform(int32_t device_id, void *d_iq, uint64_t n_samples, uint64_t seed, uint64_t first_sample);

/*
 * Add Manchester-OOK bursts: packet j starts at stream sample start[j], carries
 * n_bits bits (bits[j*stride .. ], MSB first inside each byte); bit 1 = chip
 * high then low, bit 0 = low then high, each chip chip_length samples; "high"
 * adds (d_i[j], d_q[j]) to (I,Q) with clamping to [0,255].  Packets must not
 * overlap.  Samples outside [first_sample, first_sample+n_samples) are skipped.
 */
amr_status amr_synth_plant(int32_t device_id, void *d_iq, uint64_t n_samples, uint64_t first_sample,
                           int32_t chip_length, uint32_t n_packets, const uint64_t *start,
                           const uint8_t *bits, uint32_t n_bits, uint32_t stride,
                           const int8_t *d_i, const int8_t *d_q);

/* ---- multi-GPU: gather of the hit records on one rank -------------------------------------------------
 * One process per GPU, each decoding a contiguous range of whole blocks (amr_prime rebuilds the history the reference
 * Decoder would carry into the range, decode.go:165-166; amr_set_block_base makes the call indices global).  No
 * data-path collective exists; the only exchange is this gather of (call index, idx) records, through RCCL
 * point-to-point calls on a stream of its own, enqueued without any host synchronisation and overlapping the next
 * batches.  RCCL is bound with dlopen at amr_comm_init: a host that does not call these needs no librccl.
 * A cgo host: rank 0 calls amr_comm_unique_id and hands the 128 bytes to the other processes (any transport), every
 * rank calls amr_comm_init, then after each amr_collect one amr_gather_hits. */
#define AMR_COMM_ID_BYTES 128
#define AMR_GATHER_HEADER_BYTES 128
typedef struct amr_gathered {
    uint64_t n_true;                  /* hits the source rank had */
    uint64_t n_hits;                  /* records received = min(n_true, capacity); n_true > n_hits: raise the capacity */
    uint32_t n_preambles;
    uint64_t seq;                     /* sequence number of the gather the records belong to (0, 1, 2 ... per communicator) */
    const uint64_t *preamble_offset;  /* [n_preambles+1] into the source rank's (untruncated) hit arrays */
    const uint64_t *hit_block;        /* [n_hits] global call indices */
    const uint32_t *hit_idx;          /* [n_hits] */
} amr_gathered;
amr_status amr_comm_unique_id(void *id128);   /* ncclGetUniqueId */
amr_status amr_comm_init(amr_handle *h, const void *id128, int32_t rank, int32_t world, int32_t root, uint64_t cap_hits);
amr_status amr_comm_destroy(amr_handle *h);
/*
 * The single-process form (SURVEY.md 8b "amr_create_multi"): the reference caller is ONE process (main.go:59-128), and
 * a host that holds a handle per GPU in one thread cannot call amr_comm_init n times -- the first ncclCommInitRank
 * would wait for ever for ranks the same thread has not initialised yet.  amr_comm_init_all makes the communicator of
 * hs[0..n) in one RCCL group (ncclGroupStart, n x ncclCommInitRank, ncclGroupEnd): handle i is rank i, one handle per
 * device, hs[root] receives.  amr_gather_hits_all then posts ONE gather for all n handles -- the result each handle's
 * amr_collect / amr_flush returned last -- with every rank's sends and the root's receives inside one group, so that a
 * single thread can drive it; amr_gather_fetch / amr_gather_wait / amr_comm_ranks / amr_comm_destroy work per handle as
 * before (fetch on hs[root]).  amr_gather_hits on a handle of such a communicator is AMR_EINVAL when n > 1.
 * amr_comm_check_all is the argument check alone -- handles, or (hs == NULL) the device ordinals they would sit on --
 * and needs neither a device nor RCCL.
 */
amr_status amr_comm_init_all(amr_handle **hs, int32_t n, int32_t root, uint64_t cap_hits);
amr_status amr_gather_hits_all(amr_handle **hs, int32_t n, uint64_t *seq);
amr_status amr_comm_check_all(amr_handle *const *hs, const int32_t *devices, int32_t n, int32_t root, uint64_t cap_hits);
/*
 * TEST HOOK, not for production: enable != 0 replaces RCCL by an in-process loopback transport (same point-to-point
 * semantics: FIFO matching per rank pair, sizes must match, group calls; a matched pair is one device copy) for every
 * communicator made afterwards, and lets amr_comm_check_all / amr_comm_init_all accept several handles on ONE device.
 * It exists so that the code around the transport -- the root's n receives, the two-phase header wait, truncation, the
 * inconsistent-header path, the mirror kernel -- runs with 2 and 3 ranks on a one-GPU box (tests/test_gpu_comm.py).
 * Off (the default) nothing is relaxed: two handles on one device stay AMR_EINVAL.
 */
amr_status amr_comm_test_loopback(int32_t enable);
/* ranks the RCCL communicator spans (ncclCommCount): lets a bench line prove the gather ran over N ranks */
amr_status amr_comm_ranks(const amr_handle *h, int32_t *n_ranks);
/* Enqueue the gather of the result amr_collect / amr_flush returned last (with amr_set_validation: of its surviving
 * hits -- the natural companion: 70x fewer records; an amr_flush that had nothing deferred returned an empty result:
 * zero records travel).  Collective: every rank calls it once per result, in the same order.  *seq (may be NULL)
 * receives the gather's sequence number.  Slots of up to 256 KiB (amr_gather_two_phase(cap) == 0: the capacities of
 * validated hit lists) travel whole in one message and every rank returns at once.  Larger ones (raw hit lists) are
 * sized by the hit count: a 128-byte header from every rank, then amr_gather_wire_bytes(records) from every rank that
 * has any; non-root ranks return at once, the root returns when every rank's header has arrived (it needs the counts to
 * post its receives), i.e. it runs at most one gather ahead of the slowest rank.  The library orders the kernel that
 * reads the batch's result against the later reuse of its slot by itself. */
amr_status amr_gather_hits(amr_handle *h, uint64_t *seq);
amr_status amr_gather_wait(amr_handle *h);    /* block until every gather enqueued so far has completed */
/* Root only: the records rank src_rank contributed to gather `seq`.  Waits for the arrival of that gather's records in
 * the handle's pinned host mirror (an event; no stream synchronisation, no copy) and returns pointers into it, valid
 * until gather seq + 2 is posted. */
amr_status amr_gather_fetch(amr_handle *h, uint64_t seq, int32_t src_rank, amr_gathered *out);
/* The slot every rank sends, as ONE description shared by the device pack kernel, CPU hosts and the tests:
 * [header AMR_GATHER_HEADER_BYTES | n_hits call indices u64 | n_hits idx u32], amr_gather_slot_bytes(cap) bytes in
 * memory; on the wire the whole slot, or (amr_gather_two_phase) the header and amr_gather_wire_bytes(n_hits) bytes of
 * records as two messages per rank.
 * amr_gather_pack_host builds it from a host-side result (a transport other than RCCL, e.g. the gloo tests),
 * amr_gather_unpack reads one (pointers into `slot`).  Neither needs a device. */
size_t amr_gather_slot_bytes(uint64_t cap_hits);
/* bytes of records that travel behind the header for n_sent records: 12 * n_sent rounded up to 4 KiB (0 for none) */
size_t amr_gather_wire_bytes(uint64_t n_sent);
/* 1: a communicator of this capacity sends header and count-sized records separately (slots above 256 KiB: raw hit
 * lists); 0: its slots are small enough to travel whole in one message and nobody waits for anybody (validated hits) */
int32_t amr_gather_two_phase(uint64_t cap_hits);
amr_status amr_gather_pack_host(const amr_result *res, uint64_t cap_hits, uint64_t seq, void *slot, size_t slot_bytes);
amr_status amr_gather_unpack(const void *slot, size_t slot_bytes, amr_gathered *out);

#ifdef __cplusplus
}
#endif
#endif /* AMRDEMOD_H */
