"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Bit-exact: quantized
bitstream, hit (preamble, block, idx) lists and packet bytes.  Runs on the MI355X only."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _check(protos, chip, n_blocks, seed, n_packets, batches=None, iq=None):
    dec = util.make_decoder(protos, chip)
    try:
        if iq is None:
            iq, _ = util.synth_stream(protos, chip, n_blocks, dec.Cfg.BlockSize, seed, n_packets)
        o = util.oracle_run(protos, chip, iq)
        assert o[0].geom.block_size == dec.Cfg.BlockSize and o[0].geom.packet_length == dec.Cfg.PacketLength
        g = util.gpu_run(dec, iq, batches)
        util.assert_same(o, g, dec.Cfg.PacketSymbols)
        return o, g, dec
    finally:
        dec.close()


def test_scm72_single_batch_with_tail():
    o, g, _ = _check(["scm"], 72, 200, seed=1, n_packets=12)
    assert len(g[1]) > 100  # planted packets produce runs of hits


def test_scm72_streaming_uneven_batches():
    # 1-block calls (the unchanged main.go loop), sub-history batches, tile-boundary sizes
    _check(["scm"], 72, 1 + 1 + 2 + 3 + 64 + 63 + 65 + 1 + 128, seed=2, n_packets=16,
           batches=[1, 1, 2, 3, 64, 63, 65, 1, 128])


def test_scm72_uniform_random_bytes():
    rng = np.random.default_rng(3)
    iq = rng.integers(0, 256, 130 * 8192, dtype=np.uint8)
    _check(["scm"], 72, 130, seed=0, n_packets=0, iq=iq)


@pytest.mark.parametrize("chip", [8, 32, 40, 48, 56, 64, 72, 80, 88, 96])
def test_scm_all_legal_chip_lengths(chip):
    _check(["scm"], chip, 150, seed=10 + chip, n_packets=10, batches=[70, 80])


def test_scmplus_alone_at_chip_8_has_256_sample_blocks():
    """The smallest geometry a legal command line produces (-msgtype=scm+ -symbollength=8): PreambleLength 256 =
    BlockSize, 8 words per row."""
    dec = util.make_decoder(["scm+"], 8)
    try:
        assert dec.Cfg.BlockSize == 256
        iq, _ = util.synth_stream(["scm+"], 8, 333, 256, seed=21, n_packets=10)
        for split in ([333], [1, 2, 3, 64, 65, 198]):
            dec.reset()
            util.assert_same(util.oracle_run(["scm+"], 8, iq), util.gpu_run(dec, iq, split), dec.Cfg.PacketSymbols)
    finally:
        dec.close()


@pytest.mark.parametrize("chip,bs", [(32, 1024), (40, 2048), (64, 2048), (72, 4096)])
def test_scmplus_alone_geometries(chip, bs):
    """scm+ alone has the shortest preamble window (16 symbols): BlockSize 1024 at chip 32 (32 words per row: the
    4-wave search variant), 2048 at chip 40..64."""
    dec = util.make_decoder(["scm+"], chip)
    try:
        assert dec.Cfg.BlockSize == bs
        iq, _ = util.synth_stream(["scm+"], chip, 200, bs, seed=31, n_packets=8)
        want = util.oracle_run(["scm+"], chip, iq)
        assert len(want[2]) > 50
        util.assert_same(want, util.gpu_run(dec, iq, [3, 64, 133]), dec.Cfg.PacketSymbols)
    finally:
        dec.close()


def test_idm72():
    o, g, _ = _check(["idm"], 72, 140, seed=4, n_packets=6, batches=[5, 135])
    assert len(g[1]) > 50


def test_all_protocols_72():
    o, g, dec = _check(["scm", "scm+", "idm", "r900"], 72, 140, seed=5, n_packets=9, batches=[64, 76])
    assert dec.n_preambles == 4


def test_end_to_end_messages_scm():
    protos, chip = ["scm"], 72
    dec = util.make_decoder(protos, chip)
    try:
        iq, pkts = util.synth_stream(protos, chip, 96, dec.Cfg.BlockSize, seed=6, n_packets=8)
        msgs = dec.Decode(iq)
        ids = {m.ID for m in msgs}
        want = {int.from_bytes(p.data, "big") for p in pkts}
        from rtlamr_amd.contrib.parsers.scm import SCM
        import rtlamr_amd as ra
        want_ids = {SCM.from_data(ra.new_data(p.data)).ID for p in pkts}
        assert want_ids <= ids, f"planted meters not all recovered: missing {want_ids - ids}"
    finally:
        dec.close()


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_committed_golden_fixtures():
    """tests/golden/synth.json (made by tests/golden/make_golden.py from the oracle): the HIP path
    reproduces the committed hashes without the oracle in the loop."""
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "synth.json")))
    for c in fx["cases"]:
        dec = util.make_decoder(c["protocols"], c["chip"])
        try:
            iq, _ = util.synth_stream(c["protocols"], c["chip"], c["blocks"], dec.Cfg.BlockSize, c["seed"], c["packets"])
            assert _sha(iq) == c["iq_sha"]
            q, h, p = util.gpu_run(dec, iq, [c["blocks"] // 3, c["blocks"] - c["blocks"] // 3])
            assert _sha(q) == c["qsha"], c["name"]
            assert len(h) == c["n_hits"] and _sha(h.astype("<i8")) == c["hits_sha"], c["name"]
            assert _sha(p[:, : dec.Cfg.PacketSymbols // 8]) == c["pkt_sha"], c["name"]
        finally:
            dec.close()


def test_device_generator_matches_numpy_twin():
    import ctypes as C
    from rtlamr_amd import _lib, synth
    from rtlamr_amd.contrib.parsers.scm import build_packet
    L = _lib.lib()
    n = 1 << 18
    first = 123456
    pk = [synth.Packet(first + 12000 + i * 20000, build_packet(42 + i, 5, i), 96, 25 - 50 * (i % 2), 31 - 9 * i) for i in range(8)]
    pk.append(synth.Packet(first - 5000, build_packet(7, 7, 7), 96, 40, 40))          # partly before the buffer
    pk.append(synth.Packet(first + n - 3000, build_packet(8, 8, 8), 96, -40, 127))    # partly after, clamps
    host = synth.noise(n, seed=9, first_sample=first)
    synth.plant(host, pk, 72, first_sample=first)
    d = C.c_void_p()
    _lib.check(L.amr_dev_alloc(0, 2 * n, C.byref(d)), "alloc")
    try:
        synth.device_fill(0, d.value, n, 9, first, pk, 72)
        dev = np.empty(2 * n, np.uint8)
        _lib.check(L.amr_dev_download(0, dev.ctypes.data, d, dev.size), "download")
        synth.device_fill(0, d.value, n, 9, first, [], 72, uniform_bytes=True)     # the second distribution (--data uniform)
        dev_u = np.empty(2 * n, np.uint8)
        _lib.check(L.amr_dev_download(0, dev_u.ctypes.data, d, dev_u.size), "download")
    finally:
        L.amr_dev_free(0, d)
    assert np.array_equal(host, dev)
    assert np.array_equal(synth.uniform(n, seed=9, first_sample=first), dev_u)


@pytest.mark.parametrize("protos,chip,n_blocks,npk", [(["scm"], 72, 300, 14), (["idm"], 72, 150, 6),
                                                          (["scm", "r900"], 8, 400, 14)])
def test_sharded_decode_with_priming_equals_single_decoder(protos, chip, n_blocks, npk):
    """SURVEY 8e on one GPU: three handles play three ranks; each primes with the blocks preceding its
    range (amr_prime), decodes its range, and the union of hit lists must equal the single-decoder result."""
    from rtlamr_amd import dist, synth
    one = util.make_decoder(protos, chip)
    bs, bs2 = one.Cfg.BlockSize, one.Cfg.BlockSize2
    iq, _ = util.synth_stream(protos, chip, n_blocks, bs, seed=21, n_packets=npk, edge_every=2)
    # packets across both shard edges
    kind = [p for p in protos if p in util.PKT_BUILDERS][0]
    fn, nbits = util.PKT_BUILDERS[kind]
    for e in (n_blocks // 3, 2 * (n_blocks // 3)):
        synth.plant(iq, [synth.Packet(e * bs - nbits * chip, fn(900 + e), nbits, 35, -30)], chip)
    want = util.gpu_run(one, iq)
    o = util.oracle_run(protos, chip, iq)
    util.assert_same(o, want, one.Cfg.PacketSymbols)
    PS = one.Cfg.PacketSymbols
    one.close()
    world = 3
    got_h, got_p = [], []
    for rank in range(world):
        k0, k1 = dist.shard_range(n_blocks, world, rank)
        dec = util.make_decoder(protos, chip)
        try:
            p0, _ = dist.prime_range(k0, dec.prime_blocks())
            if k0 > 0:
                lead = None
                if p0 > 0:   # the aligned halo bytes preceding the first primed block
                    lead = iq[p0 * bs2 - dec.halo_bytes(): p0 * bs2]
                dec.prime(iq[p0 * bs2: k0 * bs2], lead)
                dec.set_block_base(k0)
            _, h, p = util.gpu_run(dec, iq[k0 * bs2: k1 * bs2])
            got_h.append(h)
            got_p.append(p)
        finally:
            dec.close()
    h = np.concatenate(got_h)
    p = np.concatenate(got_p)
    order = np.lexsort((h[:, 2], h[:, 1], h[:, 0]))
    assert np.array_equal(h[order], want[1])
    p, wp = p[order].copy(), want[2].copy()
    ps = PS
    if ps % 8:
        # the one thing a block-range split cannot know: the bits Decoder.Slice never clears above the last byte's fresh
        # symbols come from the hit sliced before -- for a shard's first hit that is another shard's last (include/amrdemod.h)
        p[:, -1] &= (1 << (ps % 8)) - 1
        wp[:, -1] &= (1 << (ps % 8)) - 1
    assert np.array_equal(p, wp)


def test_dense_search_fallback_kernel(monkeypatch):
    """k2_search_dense (used when a wave's sparse hit list overflows) must give the same results."""
    monkeypatch.setenv("AMR_DENSE_SEARCH", "1")
    _check(["scm", "idm"], 72, 130, seed=31, n_packets=8, batches=[64, 66])


def test_pathological_hit_density_grows_capacities():
    """A 1-bit preamble matches ~half of all positions: overflows the sparse lists (-> dense kernel), the
    per-tile staging slots and the output arrays; all must grow transparently and still match the oracle."""
    import rtlamr_amd as ra

    class OneBit(ra.Parser):
        def __init__(self):
            self.cfg = ra.PacketConfig(Protocol="onebit", Preamble="1", DataRate=32768, ChipLength=72,
                                       PreambleSymbols=21, PacketSymbols=96)

        def Cfg(self):
            return self.cfg

        def Parse(self, pkts):
            return []

    dec = ra.new_decoder()
    dec.RegisterProtocol(OneBit())
    dec.Allocate()
    try:
        iq, _ = util.synth_stream(["scm"], 72, 70, dec.Cfg.BlockSize, seed=8, n_packets=3)
        o = util.oracle_run([("1", 21, 96)], 72, iq, hits_cap=iq.size // 2)
        g = util.gpu_run(dec, iq, [30, 40])
        util.assert_same(o, g, 96)
        assert len(g[1]) > 100000
    finally:
        dec.close()


def test_replay_file_equals_block_by_block_loop(tmp_path):
    """rtlamr_amd.replay (large host batches through amr_submit_host, three rotating pinned buffers) must print what
    the unchanged main.go loop prints: Decode one block at a time, drop a message whose digest the previous block
    already produced (main.go:235-292)."""
    from rtlamr_amd import replay
    protos, chip = ["scm", "idm"], 72
    one = util.make_decoder(protos, chip)
    bs, bs2 = one.Cfg.BlockSize, one.Cfg.BlockSize2
    n_blocks = 150
    iq, pkts = util.synth_stream(protos, chip, n_blocks, bs, seed=41, n_packets=10, edge_every=2)
    path = tmp_path / "capture.bin"
    with open(path, "wb") as f:
        f.write(iq.tobytes())
        f.write(b"\x7f" * 1000)            # a partial trailing block is dropped
    want, prev = [], set()
    for k in range(n_blocks):
        cur = set()
        for m in one.Decode(iq[k * bs2:(k + 1) * bs2]):
            key = (m.MsgType(), m.MeterType(), m.MeterID(), bytes(m.Checksum()))
            cur.add(key)
            if key not in prev:
                want.append((k, key))
        prev = cur
    one.close()
    assert len(want) >= len(pkts) - 1
    dec = util.make_decoder(protos, chip)
    try:
        with open(path, "rb") as f:
            got = [(k, (m.MsgType(), m.MeterType(), m.MeterID(), bytes(m.Checksum())))
                   for k, m in replay.replay(dec, f, batch_blocks=37)]
    finally:
        dec.close()
    assert got == want


def test_torch_tensor_input_on_torch_stream():
    """PyTorch as plumbing: IQ lives in a torch uint8 CUDA tensor, the decoder runs on a torch stream
    (amr_set_stream) and consumes tensor.data_ptr() without a copy; result == the host-input path."""
    import ctypes as C
    import torch
    from rtlamr_amd import _lib
    protos, chip = ["scm"], 72
    dec = util.make_decoder(protos, chip)
    try:
        iq, _ = util.synth_stream(protos, chip, 200, dec.Cfg.BlockSize, seed=77, n_packets=9)
        want = util.gpu_run(dec, iq)
        dec.reset()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            t = torch.from_numpy(iq).to("cuda", non_blocking=True)      # produced on `st` ...
            _lib.check(_lib.lib().amr_set_stream(dec._require(), C.c_void_p(st.cuda_stream)), "amr_set_stream")
            br = dec.decode_batch_device(t.data_ptr(), 200)              # ... and consumed on `st`: ordered, no sync
        q = dec.quantized_packed()
        h = np.concatenate([np.stack([np.full(len(b), p, np.int64), b.astype(np.int64), i.astype(np.int64)], axis=1)
                            for p in range(dec.n_preambles) for b, i, _ in [br.for_preamble(p)]])
        assert np.array_equal(q, want[0]) and np.array_equal(h, want[1])
        _lib.check(_lib.lib().amr_set_stream(dec._require(), None), "amr_set_stream")
    finally:
        dec.close()


def test_error_paths_return_status_not_crash():
    """Short input is AMR_EINVAL (the Go decoder panics, decode.go:222); a third submit without collect is refused;
    illegal chip lengths are refused at create (flags.go:127-132)."""
    import rtlamr_amd as ra
    from rtlamr_amd import _lib
    dec = util.make_decoder(["scm"], 72)
    try:
        with pytest.raises(IndexError):
            dec.decode_batch(np.zeros(100, np.uint8))
        res = _lib.AmrResult()
        rc = _lib.lib().amr_decode_batch(dec._require(), np.zeros(8192, np.uint8).ctypes.data, 8192, 2, res)
        assert rc == _lib.AMR_EINVAL
        buf = ra.protocol.PinnedBuffer(4 * dec.Cfg.BlockSize2)
        buf.array[:] = 127
        dec.submit_host(buf.array)
        dec.submit_host(buf.array)
        dec.submit_host(buf.array)
        with pytest.raises(_lib.AmrError):   # a fourth batch in flight
            dec.submit_host(buf.array)
        dec.collect(); dec.collect(); dec.collect()
        buf.free()
    finally:
        dec.close()
    with pytest.raises(Exception):
        util.make_decoder(["scm"], 78)


def test_ragged_and_empty_inputs():
    """Extra input beyond whole blocks is ignored (as in Go: Decode reads exactly BlockSize2 bytes, decode.go:169);
    zero blocks, a collect without a submit and a timing query without timing are AMR_EINVAL, not crashes."""
    import ctypes as C
    from rtlamr_amd import _lib
    L = _lib.lib()
    dec = util.make_decoder(["scm"], 72)
    try:
        bs2 = dec.Cfg.BlockSize2
        iq, _ = util.synth_stream(["scm"], 72, 40, dec.Cfg.BlockSize, seed=12, n_packets=3)
        want = util.gpu_run(dec, iq)
        dec.reset()
        ragged = np.concatenate([iq, np.full(bs2 - 1, 200, np.uint8)])       # 40 blocks + almost one more
        got = util.gpu_run(dec, ragged[: 40 * bs2])                              # the mirror passes whole blocks ...
        assert all(np.array_equal(a, b) for a, b in zip(want, got))
        dec.reset()
        res = _lib.AmrResult()                                                   # ... the ABI itself ignores the surplus
        _lib.check(L.amr_decode_batch(dec._require(), ragged.ctypes.data, ragged.size, 40, C.byref(res)), "ragged")
        assert int(res.n_hits) == len(want[1]) > 0
        assert L.amr_decode_batch(dec._require(), ragged.ctypes.data, ragged.size, 0, C.byref(res)) == _lib.AMR_EINVAL
        assert L.amr_collect(dec._require(), C.byref(res)) == _lib.AMR_EINVAL
        t = _lib.AmrTiming()
        fresh = util.make_decoder(["scm"], 72)
        try:
            assert L.amr_get_timing(fresh._require(), C.byref(t)) == _lib.AMR_EINVAL
        finally:
            fresh.close()
    finally:
        dec.close()


def test_three_batches_in_flight_and_the_fourth_is_refused():
    """amr_submit_device: up to three batches in flight; with two or more in flight K3 of a batch runs on the second
    stream next to the search of the batch behind it.  Results in submission order, equal to the oracle's."""
    import ctypes as C
    from rtlamr_amd import _lib
    L = _lib.lib()
    dec = util.make_decoder(["scm"], 72)
    bufs = []
    try:
        bs2 = dec.Cfg.BlockSize2
        sizes = [70, 64, 1, 130, 65, 3, 128]
        iq, _ = util.synth_stream(["scm"], 72, sum(sizes), dec.Cfg.BlockSize, seed=41, n_packets=14)
        want = util.oracle_run(["scm"], 72, iq)
        got, pos, inflight = [], 0, 0
        for k, nb in enumerate(sizes):
            part = np.ascontiguousarray(iq[pos * bs2:(pos + nb) * bs2])
            d = C.c_void_p()
            _lib.check(L.amr_dev_alloc(0, part.size, C.byref(d)), "alloc")
            _lib.check(L.amr_dev_upload(0, d, part.ctypes.data, part.size), "upload")
            bufs.append(d)
            dec.submit_device(d.value, nb)
            inflight += 1
            pos += nb
            if inflight == 3:
                if k == 2:   # a fourth submit is an argument error and leaves the three in flight alone
                    assert L.amr_submit_device(dec._require(), d, nb) == _lib.AMR_EINVAL
                got.append(dec.collect()); inflight -= 1
        while inflight:
            got.append(dec.collect()); inflight -= 1
        hs, ps = [], []
        for br in got:
            blk, idx, pk = br.for_preamble(0)
            hs.append(np.stack([np.zeros(len(blk), np.int64), blk.astype(np.int64), idx.astype(np.int64)], axis=1))
            ps.append(pk)
        h, p = np.concatenate(hs), np.concatenate(ps)
        assert np.array_equal(h, want[2]), "hits of the pipelined run differ from the oracle's"
        nfull = dec.Cfg.PacketSymbols // 8
        assert np.array_equal(p[:, :nfull], want[3][:, :nfull])
    finally:
        dec.close()
        for d in bufs:
            L.amr_dev_free(0, d)


@pytest.mark.parametrize("protos,chip", [(["scm"], 72), (["scm", "r900"], 72), (["scm+"], 32)])
def test_exported_signal_and_quantized_buffers(protos, chip):
    """Decoder.Signal / Decoder.Quantized (decode.go:46-50; r900 reads Signal, r900/r900.go:162-170): with KeepSignal /
    KeepQuantized the mirror holds after every batch exactly what the reference's buffers hold after the batch's last
    Decode call -- integration level (A) of INTEGRATION.md, the unchanged r900 parser on top of the GPU decoder."""
    from oracle.oracle import OracleDecoder
    dec = util.make_decoder(protos, chip)
    try:
        dec.KeepSignal = dec.KeepQuantized = True
        bs2 = dec.Cfg.BlockSize2
        sizes = [1, 1, 3, 5, 27]
        iq, _ = util.synth_stream(protos, chip, sum(sizes), dec.Cfg.BlockSize, seed=61, n_packets=3)
        o = OracleDecoder(list(protos), chip)
        pos = 0
        for nb in sizes:
            dec.decode_batch(iq[pos * bs2:(pos + nb) * bs2])
            for k in range(pos, pos + nb):
                o.decode(iq[k * bs2:(k + 1) * bs2])
            pos += nb
            assert dec.Signal.dtype == np.float32 and dec.Signal.shape == o.signal.shape
            assert np.array_equal(dec.Signal.view(np.uint32), o.signal.view(np.uint32)), f"Signal differs after call {pos}"
            assert np.array_equal(dec.Quantized, o.quantized), f"Quantized differs after call {pos}"
        dec.reset()
        assert dec.Signal is None and dec.Quantized is None
    finally:
        dec.close()
