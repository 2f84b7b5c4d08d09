"""One pipeline mode, several search kernels.  The library has no run-time switches between kernel generations any more
(round 2's AMR_K1_IMPL / AMR_K2_IMPL / AMR_K3_IMPL / AMR_TAIL_MODE / AMR_TAIL_OVERLAP / AMR_HIST_FOLD are gone); which
kernel runs follows from the geometry alone:
  K1   whole wave-tiles: k1t_demod (register tile; part of the hd ring in LDS for chip 80 .. 96); the blocks behind the last
       whole wave-tile, and small batches throughout: k1c_demod (one wave per block)
  K2   k2_search_row for one known preamble (two lanes per row at BlockSize 8192), k2_search_walk<SymbolLength, set> for every set of rtlamr's own preambles (scm, scm+, idm / netidm, r900) at every
       BlockSize from 512 to 8192; k2_search_fast when a set holds any other preamble (a custom protocol entry), up to
       four; k2_search_dense for more than four preambles, rows under 16 words, and as the overflow fallback (test hook
       AMR_DENSE_SEARCH, read at amr_create)
and the state update rides inside whichever search kernel a pipelined batch uses.  Each combination, three batches in
flight with ragged sizes, against the oracle."""
import ctypes as C

import numpy as np
import pytest

import rtlamr_amd as ra
from oracle.oracle import OracleDecoder
from rtlamr_amd import _lib, synth
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _tile_kernels(monkeypatch):
    """This file is about which kernel family runs: keep K1 on the tile kernels for whole wave-tiles (small batches would
    otherwise run one wave per block throughout, k1_coop.h; test hook AMR_K1_COOP_MAX, read at amr_create)."""
    monkeypatch.setenv("AMR_K1_COOP_MAX", "0")

SIZES = [66, 64, 3, 129, 70, 1, 65]


def _pipeline(dec, iq, sizes, depth=3):
    L = _lib.lib()
    bs2 = dec.Cfg.BlockSize2
    bufs, got, pos, inflight = [], [], 0, 0
    try:
        for nb in sizes:
            part = np.ascontiguousarray(iq[pos * bs2:(pos + nb) * bs2])
            d = C.c_void_p()
            _lib.check(L.amr_dev_alloc(0, part.size, C.byref(d)), "alloc")
            _lib.check(L.amr_dev_upload(0, d, part.ctypes.data, part.size), "upload")
            bufs.append(d)
            dec.submit_device(d.value, nb)
            inflight += 1
            pos += nb
            if inflight == depth:
                got.append(dec.collect())
                inflight -= 1
        while inflight:
            got.append(dec.collect())
            inflight -= 1
    finally:
        for d in bufs:
            L.amr_dev_free(0, d)
    hs, ps = [], []
    for br in got:
        for pid in range(dec.n_preambles):
            blk, idx, pk = br.for_preamble(pid)
            hs.append(np.stack([np.full(len(blk), pid, np.int64), blk.astype(np.int64), idx.astype(np.int64)], axis=1))
            ps.append(pk)
    h, p = np.concatenate(hs), np.concatenate(ps)
    o = np.lexsort((h[:, 2], h[:, 1], h[:, 0]))
    return h[o], p[o]


@pytest.mark.parametrize("protos,chip,dense", [
    (["scm"], 72, False),                            # k1t_demod + k2_search_walk<144, scm>
    (["scm", "scm+", "idm", "r900"], 72, False),     # all four preambles, rows of 256 words
    (["r900", "scm+"], 56, False),                   # a two-preamble set, registration order != the kernel's kind order
    (["scm"], 8, False),                             # rows of 16 words
    (["scm"], 88, False),                            # hd ring partly in LDS, rings of 96
    (["scm"], 96, False),                            # rings of 104: a super-body of 13 tiles
    (["scm"], 80, False),                            # rings of 88: 11 tiles
    (["idm"], 72, False),                            # k2_search_row, two lanes per row
    (["scm", "idm"], 72, True),                      # k2_search_dense everywhere (AMR_DENSE_SEARCH)
], ids=["walk", "walk-4pre", "walk-2pre-order", "walk-chip8", "k1-chip88", "k1-chip96", "k1-chip80", "row-two-lanes", "dense"])
def test_three_deep_pipeline_with_every_search_kernel(protos, chip, dense, monkeypatch):
    if dense:
        monkeypatch.setenv("AMR_DENSE_SEARCH", "1")
    dec = util.make_decoder(protos, chip)
    try:
        scale = 8 if chip == 8 else 1
        sizes = [s * scale for s in SIZES]
        iq, _ = util.synth_stream(protos, chip, sum(sizes), dec.Cfg.BlockSize, 91, 10)
        want = util.oracle_run(protos, chip, iq)
        h, p = _pipeline(dec, iq, sizes)
        assert len(h) > 0 and np.array_equal(h, want[2]), f"hit lists differ: gpu {len(h)} oracle {len(want[2])}"
        nfull = dec.Cfg.PacketSymbols // 8
        assert np.array_equal(p[:, :nfull], want[3][:, :nfull]), "packet bytes differ"
    finally:
        dec.close()


@pytest.mark.parametrize("extra", [["1100110011110000"], ["1100110011110000", "101100111000111100001"]], ids=["fast", "dense"])
def test_custom_preambles_go_through_the_fallback_kernels(extra):
    """Custom protocol entries next to the rtlamr ones: the walk kernel knows rtlamr's four preambles only.  One custom
    preamble (four in all) -> the list kernel k2_search_fast; two (five in all) -> k2_search_dense.  Pipelined, so the
    state update rides inside either."""
    from rtlamr_amd.protocol import PacketConfig, Parser

    class Custom(Parser):
        def __init__(self, pre, chip):
            self._c = PacketConfig(Protocol="x" + pre[:4], Preamble=pre, DataRate=32768, ChipLength=chip,
                                   PreambleSymbols=len(pre), PacketSymbols=96)
        def Cfg(self): return self._c
        def Parse(self, pkts): return []

    chip = 72
    dec = ra.new_decoder()
    for name in ("scm", "scm+", "idm"):
        dec.RegisterProtocol(ra.new_parser(name, chip))
    for pre in extra:
        dec.RegisterProtocol(Custom(pre, chip))
    dec.Allocate()
    try:
        assert dec.n_preambles == 3 + len(extra)
        protos = ["scm", "scm+", "idm"] + [(pre, len(pre), 96) for pre in extra]
        iq, _ = util.synth_stream(["scm", "scm+", "idm"], chip, sum(SIZES), dec.Cfg.BlockSize, 17, 8)
        o = OracleDecoder(protos, chip)
        _, hits, hb = o.decode_stream(iq, hits_cap=1 << 18)
        order = np.lexsort((hits[:, 2], hits[:, 0], hits[:, 1]))
        want = np.stack([hits[order, 1], hits[order, 0], hits[order, 2]], axis=1).astype(np.int64)
        h, p = _pipeline(dec, iq, SIZES)
        assert len(want) > 0 and np.array_equal(h, want)
        assert np.array_equal(p, hb[order])
    finally:
        dec.close()


@pytest.mark.parametrize("protos,chip", [(["scm+"], 72), (["scm", "scm+"], 32), (["scm+", "idm"], 72)])
def test_walk_candidate_list_overflow_reruns_dense(protos, chip):
    """A carrier that repeats scm+'s 16-symbol preamble without pause: every repetition is a run of adjacent hits three
    bitstream words long, two runs per row -- several hundred candidate words per 64-row tile, more than the 192 entries
    a wave of k2_search_walk keeps (K2Args::overflow bit 1).  The host must notice, re-run the batch's search with
    k2_search_dense and return exactly the oracle's hits; the batches after it go back to the walk kernel."""
    dec = util.make_decoder(protos, chip)
    try:
        bs = dec.Cfg.BlockSize
        n_blocks = 200
        iq = synth.noise(n_blocks * bs, 77)
        rep = bytes([0x16, 0xA3] * 8)                                   # scmplus/scmplus.go:52, eight times = 128 symbols
        span = 128 * 2 * chip                                            # samples one planted "packet" covers
        first = 70 * bs                                                  # quiet blocks first: the walk kernel runs
        pk = [synth.Packet(first + k * span, rep, 128, 34, -30) for k in range((60 * bs) // span)]
        synth.plant(iq, pk, chip)
        want = util.oracle_run(protos, chip, iq, hits_cap=1 << 20)
        got = util.gpu_run(dec, iq, [64, 64, 72])
        util.assert_same(want, got, dec.Cfg.PacketSymbols)
        pid = dec._pid_of_preamble[ra.new_parser("scm+", chip).Cfg().Preamble]
        hh = got[1][got[1][:, 0] == pid]
        words = np.unique(np.stack([hh[:, 1], hh[:, 2] // 32], axis=1), axis=0)           # bitstream words that hold hits
        per_tile = np.bincount(words[:, 0] // 64, minlength=4)
        assert per_tile.max() > 192, f"vacuous: the carrier did not fill a wave's candidate list ({per_tile.tolist()})"
    finally:
        dec.close()
