"""Wave quantisation absorbed inside the library (amr_set_deferral, include/amrdemod.h; main.go:166,235: the caller picks
the block count).  With deferral on, a pipelined batch is processed up to its last whole 64-block wave-tile and the rest
rides in front of the next batch's launch; hits keep their call index, results say which calls they cover, amr_flush
brings in the end of the stream.  Everything against the oracle: hit lists, packet bytes and -- through a synchronous
decode behind deferred blocks -- the quantized bits of a launch that starts in the head buffer."""
import ctypes as C

import numpy as np
import pytest

from rtlamr_amd import _lib
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["wave-per-block", "wave-tiles"])
def _k1_policy(request, monkeypatch):
    """Small batches run K1 as one wave per block throughout (k1_coop.h); deferral is about wave-tiles.  Every test here runs
    under both policies (test hook AMR_K1_COOP_MAX, read at amr_create: 0 = wave-tiles + one wave per block for the
    remainder only)."""
    if request.param == "wave-tiles":
        monkeypatch.setenv("AMR_K1_COOP_MAX", "0")
    else:
        monkeypatch.delenv("AMR_K1_COOP_MAX", raising=False)


def _rows(dec, br):
    rows, pk = [], []
    for pid in range(dec.n_preambles):
        blk, idx, p = br.for_preamble(pid)
        rows.append(np.stack([np.full(len(blk), pid, np.int64), blk.astype(np.int64), idx.astype(np.int64)], axis=1))
        pk.append(p)
    return np.concatenate(rows), np.concatenate(pk)


def _sorted(rows, pkt):
    o = np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0]))
    return rows[o], pkt[o]


@pytest.mark.parametrize("protos,chip,sizes,depth,host", [
    (["scm"], 72, [100, 70, 1, 63, 64, 65, 130, 7, 200], 3, False),
    (["scm"], 72, [10, 20, 5, 40, 3, 100, 64], 2, False),          # launches under 64 blocks: nothing to defer to
    (["idm"], 72, [90, 90, 90, 31], 3, True),                      # host input, BlockSize 8192
    (["scm", "scm+", "idm"], 72, [129, 127, 66, 62], 3, False),
    (["scm"], 8, [1000, 1001, 37, 999], 3, True),                  # BlockSize 512, the list-based search kernel
    (["scm"], 96, [70, 70, 70], 2, False),                         # first-generation K1
    (["scm"], 8, [300, 301, 37, 259, 64], 3, False),               # tile kernels at test size (see below): deferred head rows + the in-wave search
])
def test_deferred_pipeline_equals_oracle(protos, chip, sizes, depth, host, monkeypatch):
    L = _lib.lib()
    tile_kernels = chip == 8 and not host
    if tile_kernels:      # read at amr_create: no batch is small enough for the one-wave-per-block K1, so whole wave-tiles run the
        monkeypatch.setenv("AMR_K1_COOP_MAX", "0")       # tile kernel -- at chip 8 with its in-wave search (k1_search.h)
    dec = util.make_decoder(protos, chip)
    bufs, parts = [], []
    try:
        dec.SetDeferral(True)
        bs, bs2 = dec.Cfg.BlockSize, dec.Cfg.BlockSize2
        n_blocks = sum(sizes)
        longest = max(util.PKT_BUILDERS[p][1] for p in protos) * 2 * chip
        iq, _ = util.synth_stream(protos, chip, n_blocks, bs, seed=41, n_packets=int(max(2, min(12, n_blocks * bs // (3 * longest)))))
        want = util.oracle_run(protos, chip, iq)
        got, covered, pos, inflight = [], [], 0, 0

        def take(br):
            covered.append((br.first_block, br.n_blocks))
            got.append(_rows(dec, br))
        for nb in sizes:
            part = np.ascontiguousarray(iq[pos * bs2:(pos + nb) * bs2])
            parts.append(part)
            if host:
                dec.submit_host(part)
            else:
                d = C.c_void_p()
                _lib.check(L.amr_dev_alloc(0, part.size, C.byref(d)), "alloc")
                _lib.check(L.amr_dev_upload(0, d, part.ctypes.data, part.size), "upload")
                bufs.append(d)
                dec.submit_device(d.value, nb)
            pos += nb
            inflight += 1
            if inflight == depth:
                take(dec.collect())
                inflight -= 1
        while inflight:
            take(dec.collect())
            inflight -= 1
        take(dec.flush())
        take(dec.flush())                                   # nothing left: an empty result
        assert covered[-1] == (n_blocks, 0)
        # whole wave-tiles per launch wherever a launch had 64 blocks to work with; the results tile the stream
        edges = [c[0] for c in covered] + [n_blocks]
        assert edges[0] == 0 and all(a + n == b for (a, n), b in zip(covered, edges[1:]))
        assert all(n % 64 == 0 or n < 64 for _, n in covered[:-2]), covered
        rows, pkt = _sorted(np.concatenate([g[0] for g in got]), np.concatenate([g[1] for g in got]))
        assert len(want[2]) > 0 and np.array_equal(rows, want[2]), f"hits differ: gpu {len(rows)} oracle {len(want[2])}"
        nfull = dec.Cfg.PacketSymbols // 8
        assert np.array_equal(pkt[:, :nfull], want[3][:, :nfull])
        if tile_kernels:
            assert "in-wave-searches" in dec.describe()
    finally:
        dec.close()
        for d in bufs:
            L.amr_dev_free(0, d)


def test_sync_decode_behind_deferred_blocks_and_quantized_bits():
    """amr_decode_batch never defers: it processes the deferred blocks together with its own (a launch whose wave-tile 0
    sits in the head buffer, ending in a partial wave-tile).  Its quantized bits and hits equal the oracle's."""
    protos, chip = ["scm"], 72
    dec = util.make_decoder(protos, chip)
    try:
        dec.SetDeferral(True)
        bs, bs2 = dec.Cfg.BlockSize, dec.Cfg.BlockSize2
        n_blocks = 100 + 70 + 50
        iq, _ = util.synth_stream(protos, chip, n_blocks, bs, seed=43, n_packets=8)
        o, q_ref, h_ref, p_ref = util.oracle_run(protos, chip, iq)
        a, b = np.ascontiguousarray(iq[:100 * bs2]), np.ascontiguousarray(iq[100 * bs2:170 * bs2])
        dec.submit_host(a)           # launches 64, defers 36
        dec.submit_host(b)           # 36 + 70 = 106: launches 64, defers 42
        r1, r2 = dec.collect(), dec.collect()
        assert (r1.first_block, r1.n_blocks, r2.first_block, r2.n_blocks) == (0, 64, 64, 64)
        r3 = dec.decode_batch(iq[170 * bs2:])            # 42 deferred + 50: one launch of 92 rows
        assert (r3.first_block, r3.n_blocks) == (128, 92)
        q = dec.quantized_packed()
        assert np.array_equal(q, q_ref[128 * bs // 8:]), "quantized bits of the launch behind deferred blocks differ"
        rows, pkt = _sorted(np.concatenate([_rows(dec, r)[0] for r in (r1, r2, r3)]),
                            np.concatenate([_rows(dec, r)[1] for r in (r1, r2, r3)]))
        assert len(h_ref) > 0 and np.array_equal(rows, h_ref) and np.array_equal(pkt, p_ref)
        assert dec.flush().n_blocks == 0
    finally:
        dec.close()


def test_deferral_rules():
    """Switching deferral off with blocks deferred is refused until they are flushed; amr_flush needs an empty pipeline;
    amr_reset forgets deferred blocks with the rest of the stream; r900's second stage and deferral exclude each other."""
    import rtlamr_amd as ra
    L = _lib.lib()
    dec = util.make_decoder(["scm"], 72)
    try:
        dec.SetDeferral(True)
        bs2 = dec.Cfg.BlockSize2
        iq, _ = util.synth_stream(["scm"], 72, 100, dec.Cfg.BlockSize, seed=44, n_packets=3)
        dec.submit_host(iq)
        res = _lib.AmrResult()
        assert L.amr_flush(dec._require(), C.byref(res)) == _lib.AMR_EINVAL           # a batch is in flight
        assert dec.collect().n_blocks == 64
        assert L.amr_set_deferral(dec._require(), 0) == _lib.AMR_EINVAL               # 36 blocks are deferred
        # ... and neither a priming launch (no search: their hits would be dropped, later call indices short by 36) nor a
        # new block base (it would renumber them) is accepted in front of them
        assert L.amr_prime(dec._require(), None, iq.ctypes.data, 1, 0) == _lib.AMR_EINVAL
        assert b"amr_flush first" in L.amr_last_error()
        assert L.amr_set_block_base(dec._require(), 1000) == _lib.AMR_EINVAL
        dec.reset()
        assert dec.flush().n_blocks == 0                                              # reset forgot them
        dec.SetDeferral(False)
        want = util.oracle_run(["scm"], 72, iq)
        got = util.gpu_run(dec, iq)                                                   # and the decoder is fresh
        util.assert_same(want, got, dec.Cfg.PacketSymbols)
    finally:
        dec.close()
    # a replay closed early hands the decoder back clean: no block of the abandoned stream in the head buffer, deferral off
    from rtlamr_amd import replay
    import io
    dec = util.make_decoder(["scm"], 72)
    try:
        iq, _ = util.synth_stream(["scm"], 72, 300, dec.Cfg.BlockSize, seed=45, n_packets=6)
        gen = replay.replay(dec, io.BytesIO(iq.tobytes()), batch_blocks=100)
        next(gen, None)
        gen.close()                                       # up to two batches in flight, 36 + ... blocks deferred
        assert L.amr_set_deferral(dec._require(), 0) == _lib.AMR_OK                   # nothing deferred, already off
        dec.reset()
        want = util.oracle_run(["scm"], 72, iq[: 70 * dec.Cfg.BlockSize2])
        got = util.gpu_run(dec, iq[: 70 * dec.Cfg.BlockSize2])                        # no stale block in front of this stream
        util.assert_same(want, got, dec.Cfg.PacketSymbols)
    finally:
        dec.close()
    d9 = util.make_decoder(["scm", "r900"], 72)           # make_decoder enables the r900 second stage
    try:
        assert L.amr_set_deferral(d9._require(), 1) == _lib.AMR_EINVAL
    finally:
        d9.close()
