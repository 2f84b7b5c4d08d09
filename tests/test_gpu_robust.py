"""The pipelined decoder where its cross-stream hand-offs are stressed:

* a gate that gives up (k_gate's bounded wait, forced with AMR_GATE_TIMEOUT_US=0) must not let K3 read a search that may
  still be running: the batch is searched again on the compute stream and the results are the oracle's;
* the whole pipeline under a serialising runtime (AMD_SERIALIZE_KERNEL=3: every launch waits for the kernel before it
  and after it; HIP_LAUNCH_BLOCKING=1) gives the same results and never sits out a device-side time-out.

Each run is a process of its own (tests/pipelined_probe.py): the switches are read when the HIP runtime / the handle
is created."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def probe(protos, chip, per, n_batches, env=None, validate=False):
    e = dict(os.environ)
    e.update(env or {})
    cmd = [sys.executable, os.path.join(ROOT, "tests", "pipelined_probe.py"), ",".join(protos), str(chip), str(per), str(n_batches)]
    r = subprocess.run(cmd + (["validate"] if validate else []), env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def oracle_digest(protos, chip, per, n_batches):
    dec = util.make_decoder(protos, chip)
    try:
        bs, ps = dec.Cfg.BlockSize, dec.Cfg.PacketSymbols
    finally:
        dec.close()
    iq, _ = util.synth_stream(protos, chip, per * n_batches, bs, seed=123, n_packets=3 * n_batches)
    _, _, h, p = util.oracle_run(protos, chip, iq)
    assert ps % 8 == 0
    return {"n_hits": int(len(h)), "hits_sha256": hashlib.sha256(np.ascontiguousarray(h).tobytes()).hexdigest(),
            "pkt_sha256": hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest()}


CASES = [(["scm"], 72, 256, 7), (["idm"], 72, 128, 6), (["scm", "scm+", "idm"], 32, 192, 5)]


@pytest.mark.parametrize("protos,chip,per,n", CASES)
def test_a_gate_that_gives_up_is_recovered_on_the_compute_stream(protos, chip, per, n):
    want = oracle_digest(protos, chip, per, n)
    assert want["n_hits"] > 0
    base = probe(protos, chip, per, n, env={"AMR_K1_COOP_MAX": "0"})
    assert {k: base[k] for k in want} == want
    assert "gate-timeouts" not in base["describe"]
    got = probe(protos, chip, per, n, env={"AMR_K1_COOP_MAX": "0", "AMR_GATE_TIMEOUT_US": "0"})
    assert {k: got[k] for k in want} == want, "a batch whose gate gave up came back with other hits"
    assert "gate-timeouts" in got["describe"], "the forced time-out never fired: the test tests nothing"


def test_a_gate_that_gives_up_with_validation_on():
    protos, chip, per, n = ["scm"], 72, 256, 6
    base = probe(protos, chip, per, n, env={"AMR_K1_COOP_MAX": "0"}, validate=True)
    got = probe(protos, chip, per, n, env={"AMR_K1_COOP_MAX": "0", "AMR_GATE_TIMEOUT_US": "0"}, validate=True)
    assert base["n_hits"] > 0 and {k: got[k] for k in ("n_hits", "hits_sha256", "pkt_sha256")} == {k: base[k] for k in ("n_hits", "hits_sha256", "pkt_sha256")}
    assert "gate-timeouts" in got["describe"]


@pytest.mark.parametrize("env", [{"AMD_SERIALIZE_KERNEL": "3"}, {"HIP_LAUNCH_BLOCKING": "1"}, {"AMD_SERIALIZE_KERNEL": "3", "AMD_SERIALIZE_COPY": "3"}],
                         ids=["serialize-kernel", "launch-blocking", "serialize-kernel+copy"])
@pytest.mark.parametrize("protos,chip,per,n", CASES[:2])
def test_pipelined_decoder_under_a_serialising_runtime(protos, chip, per, n, env):
    """No multi-second stall: the device-side waits (the gate, the state update's wait for the previous tail) are bounded at
    4 s / 2 ms; a run that sat one out would take seconds for work that takes milliseconds."""
    want = oracle_digest(protos, chip, per, n)
    got = probe(protos, chip, per, n, env=dict(env, AMR_K1_COOP_MAX="0"))
    assert {k: got[k] for k in want} == want
    assert got["seconds"] < 2.0, f"the pipelined part took {got['seconds']:.2f} s: a device-side wait ran into its time-out"
    assert "gate-timeouts" not in got["describe"]


@pytest.mark.parametrize("chip", [72, 8, 32])
def test_early_search_next_to_k1_equals_oracle(chip):
    """The search of a batch off the compute stream, next to the batch's K1, every tile as soon as the K1 waves that wrote it have
    said so (per-wave-tile flags, write-through stores, loads past L1), the next K1 launch no longer behind it: forced with
    AMR_EARLY_SEARCH=1 (by default only BlockSize <= 512 runs this way, where it was measured to pay); also with a gate that
    gives up at once, which must end in a re-search, not in other hits."""
    protos, per, n = ["scm"], 256, 7
    want = oracle_digest(protos, chip, per, n)
    assert want["n_hits"] > 0
    for env in ({"AMR_EARLY_SEARCH": "1"}, {"AMR_EARLY_SEARCH": "0"}, {"AMR_EARLY_SEARCH": "1", "AMR_GATE_TIMEOUT_US": "0"}):
        got = probe(protos, chip, per, n, env=dict(env, AMR_K1_COOP_MAX="0"))
        assert {k: got[k] for k in want} == want, f"{env}: other hits than the oracle's"
        assert ("gate-timeouts" in got["describe"]) == ("AMR_GATE_TIMEOUT_US" in env)
        assert got["seconds"] < 2.0


@pytest.mark.parametrize("env", [{}, {"AMD_SERIALIZE_KERNEL": "3"}, {"HIP_LAUNCH_BLOCKING": "1"}, {"AMR_GATE_TIMEOUT_US": "0"}, {"AMR_GATE_EVENT": "0"}],
                         ids=["plain", "serialize-kernel", "launch-blocking", "gate-gives-up", "gate-without-event"])
@pytest.mark.parametrize("protos,chip", [(["idm"], 72), (["scm"], 72), (["scm", "scm+", "idm", "r900"], 72)], ids=["idm-two-lanes-per-row", "scm", "all"])
def test_batches_of_several_k1_launches(protos, chip, env):
    """At full size a batch above a chip's worth of wave-tiles runs K1 in rounds (BlockSize >= 4096), and the previous batch's
    gate kernel comes onto the chip behind the stop event of the round before the last (amr_pipeline.hip, submit).  Here at
    test size: AMR_K1_ROUND_TILES=1 makes every wave-tile a launch of its own -- batches of 256 blocks are four launches --,
    plain, under a serialising runtime, with a gate that gives up, and with round 4's gate placement."""
    per, n = 256, 6
    want = oracle_digest(protos, chip, per, n)
    assert want["n_hits"] > 0
    got = probe(protos, chip, per, n, env=dict(env, AMR_K1_COOP_MAX="0", AMR_K1_ROUND_TILES="1"))
    assert {k: got[k] for k in want} == want, f"{env}: other hits than the oracle's"
    assert ("gate-timeouts" in got["describe"]) == ("AMR_GATE_TIMEOUT_US" in env)
    assert got["seconds"] < 2.0, f"the pipelined part took {got['seconds']:.2f} s: a device-side wait ran into its time-out"
