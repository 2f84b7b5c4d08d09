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
        from rtlamr_amd.parsers.scm import SCM
        import rtlamr_amd as ra
        want_ids = {SCM.from_data(ra.new_data(p.data)).ID for p in pkts}
        assert want_ids <= ids, f"planted meters not all recovered: missing {want_ids - ids}"
    finally:
        dec.close()
