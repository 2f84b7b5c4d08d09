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


PIPE_CHILD = r"""
import ctypes as C
import numpy as np
from tests import util
from rtlamr_amd import _lib
L = _lib.lib()
for protos, chip in ((["scm"], 72), (["scm", "scm+", "idm", "r900"], 72)):
    dec = util.make_decoder(protos, chip)
    bs2 = dec.Cfg.BlockSize2
    sizes = [66, 64, 3, 129, 70, 1, 65]
    iq, _ = util.synth_stream(protos, chip, sum(sizes), dec.Cfg.BlockSize, 91, 10)
    want = util.oracle_run(protos, chip, iq)
    bufs, got, pos, inflight = [], [], 0, 0
    for nb in sizes:
        part = np.ascontiguousarray(iq[pos * bs2:(pos + nb) * bs2])
        d = C.c_void_p()
        _lib.check(L.amr_dev_alloc(0, part.size, C.byref(d)), "alloc")
        _lib.check(L.amr_dev_upload(0, d, part.ctypes.data, part.size), "upload")
        bufs.append(d)
        dec.submit_device(d.value, nb)
        inflight += 1; pos += nb
        if inflight == 3:
            got.append(dec.collect()); inflight -= 1
    while inflight:
        got.append(dec.collect()); inflight -= 1
    hs, ps = [], []
    for br in got:
        for pid in range(dec.n_preambles):
            blk, idx, pk = br.for_preamble(pid)
            hs.append(np.stack([np.full(len(blk), pid, np.int64), blk.astype(np.int64), idx.astype(np.int64)], axis=1))
            ps.append(pk)
    h, p = np.concatenate(hs), np.concatenate(ps)
    o = np.lexsort((h[:, 2], h[:, 1], h[:, 0]))
    h, p = h[o], p[o]
    assert np.array_equal(h, want[2]), "hit lists differ"
    nfull = dec.Cfg.PacketSymbols // 8
    assert np.array_equal(p[:, :nfull], want[3][:, :nfull]), "packet bytes differ"
    assert len(h) > 0
    dec.close()
    for d in bufs:
        L.amr_dev_free(0, d)
print("pipeline exact")
"""


@pytest.mark.parametrize("env", [{}, {"AMR_TAIL_MODE": "event"}, {"AMR_TAIL_OVERLAP": "0"}],
                         ids=["host-launched-tail", "event-driven-tail", "single-stream"])
def test_three_deep_pipeline_in_every_tail_mode(env):
    """K3.. of a pipelined batch run on the second stream, launched by the host (default) or behind stream events, or on
    the compute stream as before: same results as the oracle in all three."""
    r = subprocess.run([sys.executable, "-c", PIPE_CHILD], cwd=ROOT, env={**os.environ, **env}, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "pipeline exact" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
