"""The first-generation kernels (k1_demod for chip <= 72, k2_search_fast/dense for the stream kernel's geometries, the per-hit k3_slice) stay in
the library as fallbacks (AMR_K1_IMPL=old, AMR_K2_IMPL=old, AMR_K3_IMPL=old) and as the A side of A/B measurements: keep them exact.
The switch is read once per process, hence the child processes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import numpy as np
from tests import util
for protos, chip, n_blocks, split in ((["scm"], 72, 200, [3, 64, 133]), (["scm"], 32, 150, [70, 80]),
                                      (["idm"], 72, 140, [5, 135]), (["scm", "scm+", "idm"], 72, 150, [150])):
    dec = util.make_decoder(protos, chip)
    iq, _ = util.synth_stream(protos, chip, n_blocks, dec.Cfg.BlockSize, 77, 8)
    want = util.oracle_run(protos, chip, iq)
    got = util.gpu_run(dec, iq, split)
    util.assert_same(want, got, dec.Cfg.PacketSymbols)
    assert len(want[2]) > 0
    dec.close()
print("fallbacks exact")
"""


@pytest.mark.parametrize("env", [{"AMR_K1_IMPL": "old"}, {"AMR_K2_IMPL": "old"}, {"AMR_K3_IMPL": "old"},
                                 {"AMR_K1_IMPL": "old", "AMR_K2_IMPL": "old", "AMR_K3_IMPL": "old"}],
                         ids=["k1-old", "k2-old", "k3-old", "all-old"])
def test_first_generation_kernels_stay_exact(env):
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env={**os.environ, **env}, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "fallbacks exact" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
