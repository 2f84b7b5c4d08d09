"""Shared helpers for the parity tests: run the same IQ through the CPU oracle and the HIP path."""
from __future__ import annotations

import numpy as np

import rtlamr_amd as ra
from oracle.oracle import PROTOCOLS, OracleDecoder
from rtlamr_amd import synth
from rtlamr_amd.contrib.parsers.idm import build_idm_packet, build_scmplus_packet
from rtlamr_amd.contrib.parsers.scm import build_packet


def load_capture() -> np.ndarray:
    """The reference's capture assets/sample.bin (raw uint8 IQ, 572 160 bytes) from the committed fixture
    tests/golden/capture_iq.xz; its sha256 is pinned in tests/golden/sample_bin.json and SURVEY.md 8c."""
    import lzma
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "capture_iq.xz"), "rb") as f:
        return np.frombuffer(lzma.decompress(f.read()), dtype=np.uint8).copy()


def make_decoder(protos, chip) -> ra.Decoder:
    d = ra.new_decoder()
    for name in protos:
        d.RegisterProtocol(ra.new_parser(name, chip))
    d.Allocate()
    return d


def oracle_run(protos, chip, iq, mode=0, hits_cap=None):
    """-> (qpacked, hits sorted by (pid, block, idx) as int64[n,3], pkt[n,B])"""
    o = OracleDecoder(list(protos), chip)
    q, hits, hb = o.decode_stream(iq, mode=mode, hits_cap=hits_cap or max(1 << 16, iq.size // 64))
    if len(hits):
        order = np.lexsort((hits[:, 2], hits[:, 0], hits[:, 1]))
        hits, hb = hits[order], hb[order]
    h = np.stack([hits[:, 1], hits[:, 0], hits[:, 2]], axis=1).astype(np.int64) if len(hits) else np.zeros((0, 3), np.int64)
    return o, q, h, hb


def gpu_run(dec: ra.Decoder, iq, batches=None):
    """Feed iq in the given batch sizes (in blocks).  -> (qpacked, hits[n,3] (pid, block, idx), pkt[n,B])"""
    bs2 = dec.Cfg.BlockSize2
    n_blocks = iq.size // bs2
    if batches is None:
        batches = [n_blocks]
    assert sum(batches) == n_blocks
    qs, hs, ps = [], [], []
    pos = 0
    for nb in batches:
        br = dec.decode_batch(iq[pos * bs2:(pos + nb) * bs2])
        qs.append(dec.quantized_packed())
        for pid in range(dec.n_preambles):
            blk, idx, pkt = br.for_preamble(pid)
            hs.append(np.stack([np.full(len(blk), pid, np.int64), blk.astype(np.int64), idx.astype(np.int64)], axis=1))
            ps.append(pkt)
        pos += nb
    h = np.concatenate(hs) if hs else np.zeros((0, 3), np.int64)
    p = np.concatenate(ps) if ps else np.zeros((0, dec.pkt_bytes), np.uint8)
    order = np.lexsort((h[:, 2], h[:, 1], h[:, 0]))
    return np.concatenate(qs), h[order], p[order]


def assert_same(o_res, g_res, packet_symbols=None):
    _, oq, oh, op = o_res
    gq, gh, gp = g_res
    assert oq.shape == gq.shape
    if not np.array_equal(oq, gq):
        bad = np.flatnonzero(oq != gq)
        raise AssertionError(f"quantized bitstream differs in {len(bad)} bytes, first at byte {bad[0]} "
                             f"(oracle {oq[bad[0]]:08b} gpu {gq[bad[0]]:08b})")
    assert oh.shape == gh.shape, f"hit count differs: oracle {len(oh)} gpu {len(gh)}"
    assert np.array_equal(oh, gh), "hit (preamble, block, idx) lists differ"
    # every byte, also a last byte of fewer than 8 symbols with the bits Decoder.Slice never clears above them
    # (decode.go:363-366: r900 alone or with scm; csrc/k3_stale.h reproduces them in the oracle's slicing order)
    assert np.array_equal(op, gp), "packet bytes differ"


PKT_BUILDERS = {
    "scm": (lambda i: build_packet(1000 + i * 7919, (i % 12) + 1, (i * 104729) & 0xFFFFFF), 96),
    "idm": (lambda i: build_idm_packet(2000 + i * 7919, consumption=i * 31), 736),
    "netidm": (lambda i: build_idm_packet(3000 + i * 7919, consumption=i * 17), 736),
    "scm+": (lambda i: build_scmplus_packet(4000 + i * 7919, consumption=i * 13), 128),
}


def synth_stream(protos, chip, n_blocks, block_size, seed, n_packets, edge_every=4, amp=(30, -26)):
    """Noise + planted CRC-valid packets of the given protocols (round robin), host side."""
    n_samples = n_blocks * block_size
    iq = synth.noise(n_samples, seed)
    kinds = [p for p in protos if p in PKT_BUILDERS]
    pkts = []
    if n_packets and kinds:
        longest = max(PKT_BUILDERS[k][1] for k in kinds) * 2 * chip
        starts = synth.packet_schedule(n_packets, n_samples, longest, seed, edge_every, block_size)
        for i, s in enumerate(starts):
            k = kinds[i % len(kinds)]
            fn, nbits = PKT_BUILDERS[k]
            sign = 1 if i % 2 else -1
            pkts.append(synth.Packet(int(s), fn(i), nbits, sign * amp[0], -sign * amp[1] + (i % 5)))
        # packets of different lengths: plant one by one (plant() handles per-packet n_bits)
        synth.plant(iq, pkts, chip)
    return iq, pkts
