"""HIP path against oracle/_ref -- the reference's own Go sources, translated mechanically (oracle/go2cxx) -- directly,
without the hand restatement in between: quantized bitstream, hit lists from the translated literal Search, packet
bytes incl. the never-cleared bits of Decoder.Slice.  libref.so is built in the container that has /root/reference and
travels to the GPU box as a prebuilt file; where it is missing these tests skip (test_gpu_parity.py etc. still hold the
HIP path to decode_oracle.c, which tests/test_ref_translated.py holds to _ref on the CPU)."""
import numpy as np
import pytest

from oracle import ref
from tests import util

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libref.so not present on this box")]


def ref_run(protos, chip, iq):
    r = ref.RefDecoder(list(protos), chip)
    q, hits, pb, msgs = r.decode_stream(iq, hits_cap=1 << 20)
    order = np.lexsort((hits[:, 2], hits[:, 0], hits[:, 1]))
    h = np.stack([hits[:, 1], hits[:, 0], hits[:, 2]], axis=1).astype(np.int64)[order]
    return r, q, h, pb[order], msgs


@pytest.mark.parametrize("protos,chip,n_blocks,batches", [
    (["scm"], 72, 300, None),
    (["scm"], 8, 700, [1] * 5 + [695]),
    (["idm"], 72, 200, [64, 136]),
    (["scm", "scm+", "idm", "r900"], 72, 200, None),
    (["r900"], 32, 200, [100, 1, 99]),
    (["scm+", "netidm"], 96, 160, None),
])
def test_hip_equals_translated_reference(protos, chip, n_blocks, batches):
    dec = util.make_decoder(protos, chip)
    try:
        iq, _ = util.synth_stream(protos, chip, n_blocks, dec.Cfg.BlockSize, seed=31 + chip, n_packets=5, edge_every=2)
        a = iq.size // 3
        iq[a:a + 60_000] = np.random.default_rng(chip).integers(0, 256, 60_000, dtype=np.uint8)
        if "r900" in protos:          # bursts with the r900 preamble (the packet builders above cover the other protocols)
            from oracle.oracle import PROTOCOLS
            from rtlamr_amd import synth
            from rtlamr_amd.contrib.parsers import r900 as pr900
            for j in range(4):
                chips = synth.r900_chips(PROTOCOLS["r900"][0], pr900.build_r900_symbols(300 + j, consumption=j))
                synth.plant_chips(iq, (7 + 45 * j) * dec.Cfg.BlockSize + 5 * j, chips, chip, 34, -29)
        _, q, h, p, msgs = ref_run(protos, chip, iq)
        gq, gh, gp = util.gpu_run(dec, iq, batches)
        assert np.array_equal(q, gq), "quantized bitstream differs from the translated reference"
        assert h.shape == gh.shape and np.array_equal(h, gh), "hit lists differ from the translated literal Search"
        assert np.array_equal(p, gp), "packet bytes differ from the translated Slice"
        assert len(h) > 20
    finally:
        dec.close()


@pytest.mark.parametrize("protos,chip,n_blocks,batches", [
    (["scm"], 8, 704, [256, 448]),        # halo-shift K1 + the in-wave search + k2_row_cleanup (k1_search.h)
    (["scm"], 32, 256, None),             # halo-shift K1, rows of 64 words
    (["scm"], 40, 320, [64, 256]),        # two halo tiles; the all-XCD announcement is scheduling only
    (["scm"], 72, 192, [128, 64]),
])
def test_tile_kernels_equal_translated_reference(protos, chip, n_blocks, batches, monkeypatch):
    """The same comparison with the whole-wave-tile K1 kernels at test size (AMR_K1_COOP_MAX=0, read at amr_create: by default
    batches this small run one wave per block): round 6's halo shift and in-wave search straight against the translated Go."""
    monkeypatch.setenv("AMR_K1_COOP_MAX", "0")
    dec = util.make_decoder(protos, chip)
    try:
        iq, _ = util.synth_stream(protos, chip, n_blocks, dec.Cfg.BlockSize, seed=77 + chip, n_packets=6, edge_every=2)
        a = iq.size // 3
        iq[a:a + 40_000] = np.random.default_rng(chip).integers(0, 256, 40_000, dtype=np.uint8)
        _, q, h, p, _ = ref_run(protos, chip, iq)
        gq, gh, gp = util.gpu_run(dec, iq, batches)
        assert np.array_equal(q, gq), "quantized bitstream differs from the translated reference"
        assert h.shape == gh.shape and np.array_equal(h, gh), "hit lists differ from the translated literal Search"
        assert np.array_equal(p, gp), "packet bytes differ from the translated Slice"
        assert len(h) > 20
        assert ("in-wave-searches" in dec.describe()) == (chip == 8)
    finally:
        dec.close()


def test_capture_through_hip_equals_translated_reference():
    raw = util.load_capture()
    for protos, chip in ((["scm"], 72), (["scm"], 80), (["idm"], 72)):
        dec = util.make_decoder(protos, chip)
        try:
            nb = raw.size // dec.Cfg.BlockSize2
            iq = raw[: nb * dec.Cfg.BlockSize2]
            _, q, h, p, _ = ref_run(protos, chip, iq)
            gq, gh, gp = util.gpu_run(dec, iq, [1] * nb)          # one block per call, as main.go:235
            assert np.array_equal(q, gq) and np.array_equal(h, gh) and np.array_equal(p, gp)
        finally:
            dec.close()
