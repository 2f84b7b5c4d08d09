"""One block per Decode call -- the unchanged main.go loop (main.go:235) -- through the one-launch path (csrc/k1_single.h:
demodulation, search, slice, state update and ticket by ONE workgroup, the result written straight into pinned host
memory), against the oracle: every quantized bit, every hit, every packet byte; the state handed back and forth between
single-block calls and batches of the regular kernels; the take-over by the regular search when the result buffers are
too small; host and device input."""
import ctypes as C

import numpy as np
import pytest

from rtlamr_amd import _lib
from tests import util

pytestmark = pytest.mark.gpu

SETS = [(["scm"], 72), (["scm"], 8), (["scm"], 96), (["idm"], 72), (["scm+"], 32), (["scm", "scm+", "idm"], 56),
        (["scm", "idm", "netidm", "scm+"], 40), (["idm"], 88)]


@pytest.mark.parametrize("protos,chip", SETS, ids=[f"{'+'.join(p)}@{c}" for p, c in SETS])
def test_block_by_block_equals_oracle(protos, chip):
    dec = util.make_decoder(protos, chip)
    try:
        npk = 4
        longest = max(util.PKT_BUILDERS[p][1] for p in protos) * 2 * chip          # samples of the longest planted packet
        n = max(40, -(-(npk + 2) * 3 * longest // (2 * dec.Cfg.BlockSize)))      # room for the schedule's spacing
        iq, pk = util.synth_stream(protos, chip, n, dec.Cfg.BlockSize, seed=900 + chip, n_packets=npk, edge_every=2)
        want = util.oracle_run(protos, chip, iq)
        assert len(want[2]) > 0
        got = util.gpu_run(dec, iq, batches=[1] * n)
        util.assert_same(want, got, dec.Cfg.PacketSymbols)
        t = None
        dec.set_timing(1)
        dec.decode_batch(iq[: dec.Cfg.BlockSize2])
        t = dec.timing()
        assert t["demod_ms"] > 0 and t["search_ms"] == 0 and abs(t["total_ms"] - t["demod_ms"]) < 1e-6   # one launch
    finally:
        dec.close()


@pytest.mark.parametrize("protos,chip", [(["scm"], 72), (["idm"], 72), (["scm", "scm+"], 48)])
def test_single_blocks_and_batches_hand_the_state_to_each_other(protos, chip):
    dec = util.make_decoder(protos, chip)
    try:
        sizes = [1, 1, 70, 1, 64, 1, 1, 3, 1, 129, 1]
        iq, _ = util.synth_stream(protos, chip, sum(sizes), dec.Cfg.BlockSize, seed=77, n_packets=12, edge_every=2)
        want = util.oracle_run(protos, chip, iq)
        got = util.gpu_run(dec, iq, batches=sizes)
        util.assert_same(want, got, dec.Cfg.PacketSymbols)
    finally:
        dec.close()


def test_single_block_path_equals_the_regular_kernels(monkeypatch):
    """The same stream block by block with and without the one-launch path (test hook AMR_NO_SINGLE): identical results,
    and the hook really switches (the regular path reports a search time)."""
    protos, chip, n = ["scm", "idm"], 72, 60
    res = []
    for hook in (False, True):
        if hook:
            monkeypatch.setenv("AMR_NO_SINGLE", "1")
        dec = util.make_decoder(protos, chip)
        try:
            iq, _ = util.synth_stream(protos, chip, n, dec.Cfg.BlockSize, seed=5, n_packets=3, edge_every=1)
            res.append(util.gpu_run(dec, iq, batches=[1] * n))
            dec.set_timing(2)
            dec.decode_batch(iq[: dec.Cfg.BlockSize2])
            assert (dec.timing()["search_ms"] > 0) == hook
        finally:
            dec.close()
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


def test_device_input_and_unaligned_device_pointer():
    """amr_decode_batch_device with one block: the one-launch path reads the block with 16-byte loads; a pointer that is
    not 16-byte aligned takes the regular kernels -- same results either way."""
    L = _lib.lib()
    dec = util.make_decoder(["scm"], 72)
    d = C.c_void_p()
    try:
        bs2, n = dec.Cfg.BlockSize2, 12
        iq, _ = util.synth_stream(["scm"], 72, n, dec.Cfg.BlockSize, seed=31, n_packets=3, edge_every=1)
        want = util.oracle_run(["scm"], 72, iq)
        _lib.check(L.amr_dev_alloc(0, iq.size + 64, C.byref(d)), "alloc")
        for shift in (0, 4):
            dec.reset()
            _lib.check(L.amr_dev_upload(0, C.c_void_p(d.value + shift), iq.ctypes.data, iq.size), "upload")
            hs, ps = [], []
            for k in range(n):
                br = dec.decode_batch_device(d.value + shift + k * bs2, 1)
                blk, idx, pkt = br.for_preamble(0)
                hs.append(np.stack([np.zeros(len(blk), np.int64), blk.astype(np.int64), idx.astype(np.int64)], axis=1))
                ps.append(pkt)
            assert np.array_equal(np.concatenate(hs), want[2]) and np.array_equal(np.concatenate(ps), want[3])
    finally:
        dec.close()
        if d.value:
            L.amr_dev_free(0, d)


def test_more_hits_than_the_result_buffers_hold_is_taken_over_by_the_regular_search(monkeypatch):
    """AMR_HIT_CAP=64: a planted packet's run of adjacent hits (about one chip length of positions) exceeds 64 records in
    one block; the one-launch kernel then publishes only the count, amr_collect grows the buffers and searches the slot
    again with the regular kernels.  Every slot of the ring starts small, so the take-over happens several times."""
    monkeypatch.setenv("AMR_HIT_CAP", "64")
    protos, chip, n = ["scm", "idm"], 72, 72
    dec = util.make_decoder(protos, chip)
    try:
        iq, _ = util.synth_stream(protos, chip, n, dec.Cfg.BlockSize, seed=11, n_packets=4, edge_every=1)
        want = util.oracle_run(protos, chip, iq)
        per_call = np.bincount(want[2][:, 1].astype(np.int64), minlength=n)
        assert (per_call > 64).sum() >= 3, "no call with more than 64 hits: the test tests nothing"
        got = util.gpu_run(dec, iq, batches=[1] * n)
        util.assert_same(want, got, dec.Cfg.PacketSymbols)
    finally:
        dec.close()
