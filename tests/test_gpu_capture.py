"""BASELINE config 1 on the GPU: the reference's own capture (assets/sample.bin, committed as tests/golden/capture_iq.xz)
through the HIP path, against the golden vectors of tests/golden/sample_bin.json -- the hashes SURVEY.md 8c publishes
(two independent restatements) and tests/test_oracle_golden.py pins the oracle to."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
FX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_bin.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("case", FX["cases"], ids=[c["name"] for c in FX["cases"]])
@pytest.mark.parametrize("batches", ["one", "per_block", "uneven"])
def test_capture_matches_golden(case, batches):
    raw = util.load_capture()
    assert sha(raw) == FX["file_sha256"]
    dec = util.make_decoder(case["protocols"], case["chip"])
    try:
        assert dec.Cfg.BlockSize == case["block_size"]
        nb = case["blocks"]
        iq = raw[: nb * dec.Cfg.BlockSize2]
        split = {"one": [nb], "per_block": [1] * nb, "uneven": [1, 2, 7, nb - 10]}[batches]   # main.go:235 is per_block
        q, h, p = util.gpu_run(dec, iq, split)
        assert int(np.unpackbits(q).sum()) == case["ones"]
        assert sha(q) == case["qsha"], "quantized bitstream differs from the golden hash"
        assert h[:, 1:].tolist() == case["hits"]
        assert sha(p) == case["pkt_sha"]
        assert sha(dec.mag_lut().astype("<f4")) == FX["lut_sha256"]
    finally:
        dec.close()
