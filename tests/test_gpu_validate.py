"""Per-hit validation on the GPU: with Decoder.EnableValidation the hits read back must be
exactly the oracle's hits filtered by oracle/validate_oracle.py (checksum tests of the Parse loops + adjacent-repeat
removal), in the same order, with the same packet bytes -- and the parsers must emit the same messages as without."""
import ctypes as C

import numpy as np
import pytest

import rtlamr_amd as ra
from oracle import r900_oracle, validate_oracle as vo
from oracle.oracle import PROTOCOLS
from rtlamr_amd import _lib, synth
from rtlamr_amd.contrib.parsers import r900
from tests import util

pytestmark = pytest.mark.gpu


def _expected(protos, chip, iq, dec):
    """oracle hits -> per preamble filter -> (pid, block, idx) rows + packets, sorted like the GPU result"""
    _, _, oh, op = util.oracle_run(protos, chip, iq)
    H, P, done = [], [], set()
    for name in protos:                      # registration order = preamble id order
        pid = dec._pid_of_preamble[PROTOCOLS[name][0]]
        if pid in done:                      # idm + netidm share one preamble (and one rule)
            continue
        done.add(pid)
        sel = np.flatnonzero(oh[:, 0] == pid)
        keep = sel[vo.filter_hits(name, oh[sel, 1], op[sel])] if name in vo.RULES else sel
        H.append(oh[keep]); P.append(op[keep])
    return np.concatenate(H), np.concatenate(P), len(oh)


def _msgs(dec, br):
    return sorted((m.MsgType(), m.MeterID(), m.MeterType(), bytes(m.Checksum())) for b in dec.run_parsers(br) for m in b)


@pytest.mark.parametrize("protos,chip,n_blocks,n_packets,batches", [
    (["scm"], 72, 96, 24, [96]),
    (["scm"], 32, 150, 30, [7, 64, 79]),
    (["idm"], 72, 80, 5, [33, 47]),
    (["scm", "scm+", "idm"], 72, 120, 8, [120]),
    (["idm", "netidm"], 72, 64, 4, [64]),          # two parsers behind one preamble, same rule
])
def test_validated_hits_equal_filtered_oracle(protos, chip, n_blocks, n_packets, batches):
    plain = util.make_decoder(protos, chip)
    val = util.make_decoder(protos, chip)
    try:
        assert len(val.EnableValidation()) == len({PROTOCOLS[p][0] for p in protos})
        bs, bs2 = val.Cfg.BlockSize, val.Cfg.BlockSize2
        iq, pkts = util.synth_stream(protos, chip, n_blocks, bs, seed=23, n_packets=n_packets)
        # flip one payload bit of some planted packets: their hits must be searched but not reported
        for i, p in enumerate(pkts):
            if i % 4 == 3:
                b = bytearray(p.data); b[6] ^= 0x10
                synth.plant(iq, [synth.Packet(p.start, p.data, p.n_bits, -p.d_i, -p.d_q)], chip)   # undo (nothing clips)
                synth.plant(iq, [synth.Packet(p.start, bytes(b), p.n_bits, p.d_i, p.d_q)], chip)
        want_h, want_p, n_searched = _expected(protos, chip, iq, val)
        assert 0 < len(want_h) < n_searched
        got_h, got_p, searched, m_val, m_plain = [], [], 0, [], []
        pos = 0
        for nb in batches:
            chunk = iq[pos * bs2:(pos + nb) * bs2]
            br = val.decode_batch(chunk)
            searched += br.n_hits_searched
            for pid in range(val.n_preambles):
                blk, idx, pk = br.for_preamble(pid)
                got_h.append(np.stack([np.full(len(blk), pid, np.int64), blk.astype(np.int64), idx.astype(np.int64)], axis=1))
                got_p.append(pk)
            m_val += _msgs(val, br)
            m_plain += _msgs(plain, plain.decode_batch(chunk))
            pos += nb
        got_h, got_p = np.concatenate(got_h), np.concatenate(got_p)
        order = np.lexsort((got_h[:, 2], got_h[:, 1], got_h[:, 0]))
        assert np.array_equal(got_h[order], got_h), "validated hits are not in (preamble, block, idx) order"
        assert searched == n_searched
        assert np.array_equal(got_h, want_h), f"{len(got_h)} validated hits, oracle filter keeps {len(want_h)}"
        assert np.array_equal(got_p, want_p)
        assert m_val == m_plain and len(m_val) > 0
    finally:
        plain.close()
        val.close()


@pytest.mark.parametrize("protos,kind,chip,batches", [
    (["scm"], "scm", 72, [192]),
    (["scm"], "scm", 72, [70, 122]),              # the boundary of the second batch's tiles lies elsewhere in the stream
    (["idm"], "idm", 72, [192]),
    (["scm", "scm+", "idm"], "scm+", 72, [192]),
])
def test_repeats_across_a_wave_tile_boundary(protos, kind, chip, batches):
    """K5's test runs per (wave-tile, preamble) list; the hit before a list's first one belongs to the previous tile's
    list.  Packets planted so that the run of hits of ONE packet (same bytes, same Decode call) straddles the
    boundary between two wave-tiles of the launch: the repeats on the far side must go like all the others."""
    val = util.make_decoder(protos, chip)
    try:
        val.EnableValidation()
        bs, bs2 = val.Cfg.BlockSize, val.Cfg.BlockSize2
        n_blocks = sum(batches)
        iq = synth.noise(n_blocks * bs, seed=41)
        fn, nbits = util.PKT_BUILDERS[kind]
        # wave-tile t of a launch starts at block 64 t of that launch (the history tile in front of block 0)
        bounds, first = [], 0
        for nb in batches:
            bounds += [first + 64 * t for t in range(1, (nb + 63) // 64)]
            first += nb
        assert len(bounds) >= 2
        # hit position n (bitstream index = Signal index, SymbolLength of history in front) = packet start + SymbolLength,
        # +- about half a chip of neighbours that slice to the same bytes
        pkts = [synth.Packet(b * bs - 2 * chip + (10 if i % 2 else -10), fn(50 + i), nbits, 31 if i % 2 else -31, -27 if i % 2 else 27)
                for i, b in enumerate(bounds)]
        synth.plant(iq, pkts, chip)
        want_h, want_p, n_searched = _expected(protos, chip, iq, val)
        # the premise: of every planted packet there are hits on both sides of its boundary with the same bytes
        _, _, oh, op = util.oracle_run(protos, chip, iq)
        PL = val.Cfg.PacketLength
        n_of = oh[:, 1] * bs + oh[:, 2] - PL
        pid = val._pid_of_preamble[PROTOCOLS[kind][0]]
        for b, p in zip(bounds, pkts):
            nb_ = (len(p.data))
            same = (oh[:, 0] == pid) & (np.abs(n_of - b * bs) < 4 * chip) & np.all(op[:, :nb_] == np.frombuffer(p.data, np.uint8), axis=1)
            assert np.any(same & (n_of < b * bs)) and np.any(same & (n_of >= b * bs)), "the run does not straddle the tile boundary"
        got_h, got_p, pos = [], [], 0
        per_batch = []
        for nb in batches:
            per_batch.append(val.decode_batch(iq[pos * bs2:(pos + nb) * bs2]))
            pos += nb
        for q in range(val.n_preambles):
            for br in per_batch:
                blk, idx, pk = br.for_preamble(q)
                got_h.append(np.stack([np.full(len(blk), q, np.int64), blk.astype(np.int64), idx.astype(np.int64)], axis=1))
                got_p.append(pk)
        got_h, got_p = np.concatenate(got_h), np.concatenate(got_p)
        assert sum(br.n_hits_searched for br in per_batch) == n_searched
        assert np.array_equal(got_h, want_h), f"{len(got_h)} validated hits, oracle filter keeps {len(want_h)}"
        assert np.array_equal(got_p, want_p)
    finally:
        val.close()


def test_validation_with_r900_digits_and_pipeline():
    """'all' geometry: scm / scm+ / idm validated, r900 hits all kept and still aligned with their digits;
    submit/collect pipeline; result_device describes the validated list."""
    protos, chip = ["scm", "scm+", "idm", "r900"], 72
    plain = util.make_decoder(protos, chip)
    val = util.make_decoder(protos, chip)
    try:
        assert len(val.EnableValidation()) == 3
        bs, bs2 = val.Cfg.BlockSize, val.Cfg.BlockSize2
        n_blocks = 96
        iq, _ = util.synth_stream(["scm", "scm+", "idm"], chip, n_blocks, bs, seed=5, n_packets=6)
        pre = r900_oracle.PROTOCOLS["r900"][0]
        for i, (mid, s) in enumerate(((4242, 30 * bs + 100), (777001, 70 * bs - 5000))):
            chips = synth.r900_chips(pre, r900.build_r900_symbols(mid, consumption=mid * 3))
            synth.plant_chips(iq, s, chips, chip, 33 if i else -33, -27 if i else 27)
        halves = [40, 56]
        d = ra._lib.lib()
        bufs = []
        for k, nb in enumerate(halves):
            ptr = C.c_void_p()
            _lib.check(d.amr_dev_alloc(0, nb * bs2, C.byref(ptr)), "alloc")
            part = np.ascontiguousarray(iq[sum(halves[:k]) * bs2:(sum(halves[:k]) + nb) * bs2])
            _lib.check(d.amr_dev_upload(0, ptr, part.ctypes.data, part.size), "upload")
            bufs.append((ptr, nb))
        for ptr, nb in bufs:
            val.submit_device(ptr.value, nb)
        m_val, kept, searched = [], 0, 0
        rp = val._pid_of_preamble[pre]
        for k in range(2):
            br = val.collect()
            p, n = val.result_device()
            assert n == len(br.hit_idx) and (p != 0 or n == 0)
            blk, idx, _ = br.for_preamble(rp)
            assert br.r900_digits.shape == (len(blk), 42)
            kept += len(br.hit_idx); searched += br.n_hits_searched
            m_val += _msgs(val, br)
        m_plain = []
        for k, nb in enumerate(halves):
            m_plain += _msgs(plain, plain.decode_batch(iq[sum(halves[:k]) * bs2:(sum(halves[:k]) + nb) * bs2]))
        assert m_val == m_plain
        assert {"SCM", "SCM+", "IDM", "R900"} <= {m[0] for m in m_val}
        assert kept < searched
        for ptr, _ in bufs:
            d.amr_dev_free(0, ptr)
    finally:
        plain.close()
        val.close()


def test_validation_survives_capacity_growth(monkeypatch):
    """More hits than the result buffers hold (AMR_HIT_CAP test hook: 256): the search is re-run with larger
    buffers, the validation buffers grow with them."""
    monkeypatch.setenv("AMR_HIT_CAP", "256")
    val = util.make_decoder(["scm"], 72)
    monkeypatch.delenv("AMR_HIT_CAP")
    plain = util.make_decoder(["scm"], 72)
    try:
        val.EnableValidation()
        bs = val.Cfg.BlockSize
        iq, _ = util.synth_stream(["scm"], 72, 200, bs, seed=3, n_packets=40)
        brp = plain.decode_batch(iq)
        br = val.decode_batch(iq)
        assert br.n_hits_searched == len(brp.hit_idx) > 1024
        keep = vo.filter_hits("scm", brp.hit_block, brp.pkt)
        assert 0 < len(keep) < len(brp.hit_idx)
        assert np.array_equal(br.hit_block, brp.hit_block[keep]) and np.array_equal(br.hit_idx, brp.hit_idx[keep])
        assert np.array_equal(br.pkt, brp.pkt[keep])
        br2 = val.decode_batch(iq[: 50 * val.Cfg.BlockSize2])       # and the grown buffers keep working
        assert 0 < len(br2.hit_idx) < br2.n_hits_searched
    finally:
        val.close()
        plain.close()


def test_validation_argument_checks():
    dec = util.make_decoder(["scm", "r900"], 72)
    try:
        L, h = _lib.lib(), dec._handle
        v = _lib.AmrValidator()
        v.n_checks, v.dedupe_bytes = 1, 12
        v.checks[0].init, v.checks[0].poly, v.checks[0].residue, v.checks[0].n_spans = 0, 0x6F63, 0, 1
        v.checks[0].span_off[0], v.checks[0].span_len[0] = 2, 10
        rp = dec._pid_of_preamble[PROTOCOLS["r900"][0]]
        assert L.amr_set_validation(h, rp, C.byref(v)) == _lib.AMR_EINVAL          # r900 hits carry digits
        assert L.amr_set_validation(h, 7, C.byref(v)) == _lib.AMR_EINVAL           # no such preamble
        v.checks[0].span_len[0] = 200
        assert L.amr_set_validation(h, 0, C.byref(v)) == _lib.AMR_EINVAL           # span outside the packet
        v.checks[0].span_len[0] = 10
        v.n_checks = 3
        assert L.amr_set_validation(h, 0, C.byref(v)) == _lib.AMR_EINVAL
        v.n_checks = 1
        assert L.amr_set_validation(h, 0, C.byref(v)) == _lib.AMR_OK
        assert L.amr_set_validation(h, 0, None) == _lib.AMR_OK                      # off again
        assert dec.EnableValidation() == [PROTOCOLS["scm"][0]]                     # r900 has no VALIDATOR
    finally:
        dec.close()
