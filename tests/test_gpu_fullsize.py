"""BASELINE.json configurations at their full single-GPU sizes (1 GiB SCM, 4 GiB IDM, 1 GiB per chip length,
multi-protocol), checked through size-independent properties -- the oracle cannot chew GiBs in a test:

  1. every planted CRC-valid packet is found at the position its start sample dictates, with its exact bytes;
  2. on sampled windows of blocks (stream start, stream end, around block-straddling packets, random) the hit
     lists, packet bytes and quantized bits equal the CPU oracle's, bit for bit.  The oracle is fed the window
     plus enough preceding blocks that its histories equal the single-stream ones (decode.go:165-166);
  3. decoding the stream in two device-resident batches gives the same hits as one batch (history carry).
IQ is generated in HBM (SURVEY.md 8d generator); nothing here reads /root/reference.
"""
import ctypes as C

import numpy as np
import pytest

import rtlamr_amd as ra
from oracle.oracle import OracleDecoder
from rtlamr_amd import _lib, synth
from tests import util

pytestmark = pytest.mark.gpu

GIB = 1 << 30


def _packets(kind, chip, n_packets, n_samples, bs, seed):
    fn, nbits = util.PKT_BUILDERS[kind]
    starts = synth.packet_schedule(n_packets, n_samples, nbits * 2 * chip, seed=seed, edge_every=16, block_size=bs)
    pk = []
    for i, s in enumerate(starts):
        sign = 1 if i % 2 else -1
        pk.append(synth.Packet(int(s), fn(i), nbits, sign * (24 + i % 15), -sign * (23 + i % 11)))
    return pk


def _hits(dec, br):
    out = []
    for pid in range(dec.n_preambles):
        blk, idx, pkt = br.for_preamble(pid)
        out.append((blk.astype(np.int64), idx.astype(np.int64), pkt))
    return out


def _oracle_window(protos, chip, L, d_iq, bs2, k0, w, warm, n_blocks):
    """Oracle hits for calls [k0, k0+w) of the device-resident stream -> (pid, block, idx) rows, packets, q bytes."""
    lo = max(0, k0 - warm)
    buf = np.empty((k0 + w - lo) * bs2, np.uint8)
    _lib.check(L.amr_dev_download(0, buf.ctypes.data, C.c_void_p(d_iq + lo * bs2), buf.size), "download")
    o = OracleDecoder(list(protos), chip)
    q, hits, hb = o.decode_stream(buf, hits_cap=max(1 << 16, buf.size // 64))
    hits = hits.astype(np.int64)
    keep = hits[:, 0] >= (k0 - lo)
    hits, hb = hits[keep], hb[keep]
    rows = np.stack([hits[:, 1], hits[:, 0] + lo, hits[:, 2]], axis=1)
    order = np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0]))
    bs = bs2 // 2
    return rows[order], hb[order], q[(k0 - lo) * bs // 8:]


CASES = [
    # (name, protocols, chip, bytes, planted kind, packets)
    ("cfg2_scm72_1GiB", ["scm"], 72, 1 * GIB, "scm", 4096),
    ("cfg3_idm72_4GiB", ["idm"], 72, 4 * GIB, "idm", 4096),
    ("cfg5_all72_4GiB", ["scm", "scm+", "idm", "r900"], 72, 4 * GIB, "scm+", 4096),
] + [(f"cfg4_scm{c}_1GiB", ["scm"], c, 1 * GIB, "scm", 4096) for c in (8, 32, 40, 48, 56, 64)]


@pytest.mark.parametrize("name,protos,chip,nbytes,kind,npk", CASES, ids=[c[0] for c in CASES])
def test_full_size_properties(name, protos, chip, nbytes, kind, npk):
    L = _lib.lib()
    dec = util.make_decoder(protos, chip)
    d = C.c_void_p()
    try:
        bs, bs2, pl = dec.Cfg.BlockSize, dec.Cfg.BlockSize2, dec.Cfg.PacketLength
        n_blocks = nbytes // bs2
        n_samples = n_blocks * bs
        pk = _packets(kind, chip, npk, n_samples, bs, seed=3)
        _lib.check(L.amr_dev_alloc(0, nbytes, C.byref(d)), "alloc")
        synth.device_fill(0, d.value, n_samples, seed=5, first_sample=0, packets=pk, chip_length=chip)

        br = dec.decode_batch_device(d.value, n_blocks)
        q_all = dec.quantized_packed()
        hits = _hits(dec, br)
        pid = dec._pid_of_preamble[ra.new_parser(kind, chip).Cfg().Preamble]
        blk, idx, pkt = hits[pid]
        pos = blk * bs + idx                       # = first-tap bit position + PacketLength, ascending

        # 1. every planted packet: a hit carrying exactly its bytes within a chip of where its start sample puts it.
        #    Bit n of the stream is the matched filter over samples [n - SymbolLength, n) (decode.go:239-244 on
        #    Signal = SymbolLength history + block), so a symbol that starts at sample s is decided at n = s + SL.
        nb = pk[0].n_bits // 8
        missing = checked = 0
        for p in pk:
            want = p.start + 2 * chip + pl
            if want + bs > n_blocks * bs:          # its last call lies beyond the stream
                continue
            checked += 1
            a, b = np.searchsorted(pos, [want - chip, want + chip + 1])
            ref = np.frombuffer(p.data[:nb], np.uint8)
            if not (b > a and (pkt[a:b, :nb] == ref).all(axis=1).any()):
                missing += 1
        assert checked >= len(pk) - 2 and len(pos) > checked      # not vacuous
        assert missing == 0, f"{missing} of {checked} planted packets not recovered"

        # 2. sampled windows against the oracle
        warm = dec.prime_blocks() + 1
        w = 6
        edge_pk = [p for p in pk if (p.start // bs) != ((p.start + p.n_bits * 2 * chip) // bs)][:2]
        k0s = {0, n_blocks - w, n_blocks // 3, (n_blocks // 7) * 5}
        k0s |= {min(max(0, (p.start + 2 * chip + pl) // bs - 2), n_blocks - w) for p in edge_pk}
        n_window_hits = 0
        for k0 in sorted(k0s):
            rows, opkt, oq = _oracle_window(protos, chip, L, d.value, bs2, k0, w, warm, n_blocks)
            got_rows, got_pkt = [], []
            for q_id, (gb, gi, gp) in enumerate(hits):
                a, b = np.searchsorted(gb, [k0, k0 + w])
                got_rows.append(np.stack([np.full(b - a, q_id, np.int64), gb[a:b], gi[a:b]], axis=1))
                got_pkt.append(gp[a:b])
            got_rows, got_pkt = np.concatenate(got_rows), np.concatenate(got_pkt)
            n_window_hits += len(rows)
            assert np.array_equal(rows, got_rows), f"{name}: hit list differs in calls [{k0},{k0 + w})"
            nfull = dec.Cfg.PacketSymbols // 8
            assert np.array_equal(opkt[:, :nfull], got_pkt[:, :nfull]), f"{name}: packet bytes differ at call {k0}"
            assert np.array_equal(oq, q_all[k0 * bs // 8:(k0 + w) * bs // 8]), f"{name}: quantized bits differ at call {k0}"

        assert n_window_hits > 0, "sampled windows contained no hit at all"

        # 3. two batches == one batch (state carried across batches)
        dec.reset()
        cut = (n_blocks // 2) | 1                   # odd: not a multiple of the 64-block wave tile
        br1 = dec.decode_batch_device(d.value, cut)
        h1 = _hits(dec, br1)
        br2 = dec.decode_batch_device(d.value + cut * bs2, n_blocks - cut)
        h2 = _hits(dec, br2)
        for q_id in range(dec.n_preambles):
            for j in range(3):
                assert np.array_equal(np.concatenate([h1[q_id][j], h2[q_id][j]]), hits[q_id][j]), \
                    f"{name}: split decode differs (preamble {q_id}, field {j})"
    finally:
        if d.value:
            L.amr_dev_free(0, d)
        dec.close()
