"""r900 second-stage matched filter on the GPU: the 42 base-6 digits of every r900 preamble hit
must equal, bit for bit, what r900.Parser.Parse reads from its quantized buffer (r900.go:82-150, 183-193) -- checked
against oracle/r900_oracle.py -- and planted Reed-Solomon-valid bursts must come out as R900 messages."""
import numpy as np
import pytest

from oracle import r900_oracle
from rtlamr_amd import synth
from rtlamr_amd.contrib.parsers import r900
from tests import util

pytestmark = pytest.mark.gpu


def _stream(protos, chip, n_blocks, bs, mids, starts, seed):
    iq = synth.noise(n_blocks * bs, seed)
    pre = r900_oracle.PROTOCOLS["r900"][0]
    for i, (mid, s) in enumerate(zip(mids, starts)):
        chips = synth.r900_chips(pre, r900.build_r900_symbols(mid, consumption=(mid * 7) & 0xFFFFFF, leak=i & 15))
        sign = 1 if i % 2 else -1
        synth.plant_chips(iq, s, chips, chip, sign * 33, -sign * 27)
    return iq


def _gpu_digits(dec, iq, batches):
    bs2 = dec.Cfg.BlockSize2
    pid = dec._pid_of_preamble[r900_oracle.PROTOCOLS["r900"][0]]
    hits, digits, msgs = [], [], []
    pos = 0
    for nb in batches:
        br = dec.decode_batch(iq[pos * bs2:(pos + nb) * bs2])
        assert br.r900_preamble == pid
        blk, idx, _ = br.for_preamble(pid)
        hits.append(np.stack([blk.astype(np.int64), idx.astype(np.int64)], axis=1))
        digits.append(br.r900_digits)
        msgs += [m for b in dec.run_parsers(br) for m in b]
        pos += nb
    return np.concatenate(hits), np.concatenate(digits), msgs


@pytest.mark.parametrize("protos,chip,n_blocks,batches", [
    (["r900"], 72, 30, [30]),
    (["r900"], 72, 30, [1, 2, 3, 7, 17]),                 # history shorter than PacketLength at first, then carried
    (["scm", "scm+", "idm", "r900"], 72, 40, [13, 27]),   # "all" geometry: BufferLength 114176
    (["r900"], 32, 40, [9, 31]),
])
def test_r900_digits_equal_oracle_and_messages_recovered(protos, chip, n_blocks, batches):
    dec = util.make_decoder(protos, chip)
    try:
        bs = dec.Cfg.BlockSize
        burst = (64 + 168) * chip
        mids = [1001, 20002, 300003, 4000004]
        total, pl = n_blocks * bs, dec.Cfg.PacketLength
        usable = total - pl - 2 * bs - burst          # a burst that starts later is reported by a call beyond the stream
        s2 = sum(batches[:-1]) * bs - burst // 3 if len(batches) > 1 else usable // 2   # straddles the last batch boundary
        anchor = usable * 5 // 8 if s2 < usable // 2 else usable // 4
        s1 = (anchor // bs + 1) * bs - burst // 2                                      # straddles a block boundary
        starts = sorted([40, s1, s2, usable])
        assert all(b - a > burst + 2 * chip for a, b in zip(starts, starts[1:])), starts
        iq = _stream(protos, chip, n_blocks, bs, mids, starts, seed=17)
        want_hits, want_digits = r900_oracle.digits_for_stream(protos, chip, iq)
        assert len(want_hits) > 100
        hits, digits, msgs = _gpu_digits(dec, iq, batches)
        assert np.array_equal(hits, want_hits)
        bad = np.flatnonzero((digits != want_digits).any(axis=1))
        assert len(bad) == 0, f"{len(bad)} of {len(hits)} hits differ, first: call {hits[bad[0]]} gpu {digits[bad[0]]} oracle {want_digits[bad[0]]}"
        ids = {m.ID for m in msgs if m.MsgType() == "R900"}
        assert ids == set(mids)
    finally:
        dec.close()
