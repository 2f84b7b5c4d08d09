"""Pinning to the Go reference.

go/cmd/amdgolden -- built inside an rtlamr checkout, linking the UNMODIFIED reference packages -- writes
tests/golden/go_sample_bin.json and go_synth.json in the schema of sample_bin.json / synth.json.  When those files
exist, the oracle's golden vectors must equal them field by field (oracle == Go), and on the GPU box the HIP path is
run against them directly (HIP == Go).  No Go toolchain exists in the build image, so until someone runs the generator
the comparison itself is exercised on a stand-in file (a copy of the oracle's vectors, and a corrupted copy that must
be rejected): the machinery is tested, the pin is not claimed.
"""
import copy
import hashlib
import json
import os

import numpy as np
import pytest

from tests import util

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
SYNTH_FIELDS = ("protocols", "chip", "blocks", "seed", "packets", "iq_sha", "qsha", "n_hits", "hits_sha", "pkt_sha")
SAMPLE_FIELDS = ("protocols", "chip", "blocks", "block_size", "ones", "qsha", "hits", "pkt_sha")
R900_FIELDS = ("protocols", "chip", "input", "calls", "qsha", "hist")      # r900.Parser.filter, r900/r900.go:82-150


def _load(name):
    p = os.path.join(G, name)
    return json.load(open(p)) if os.path.exists(p) else None


def compare_goldens(ours: dict, go: dict, fields) -> list:
    """-> list of human-readable differences between two golden files (empty = equal on every case and field)."""
    diffs = []
    by_name = {c["name"]: c for c in go.get("cases", [])}
    for c in ours["cases"]:
        g = by_name.get(c["name"])
        if g is None:
            diffs.append(f"{c['name']}: missing from the Go file")
            continue
        for f in fields:
            if f in c and c[f] != g.get(f):
                diffs.append(f"{c['name']}.{f}: oracle {str(c[f])[:80]} != Go {str(g.get(f))[:80]}")
    for k in ("file_sha256", "file_bytes", "lut_sha256"):
        if k in ours and ours[k] != go.get(k):
            diffs.append(f"{k}: oracle {ours[k]} != Go {go.get(k)}")
    return diffs


def test_comparison_machinery_on_a_stand_in():
    """Schema round trip: a Go-shaped copy of the oracle's vectors compares equal; one flipped hash, one dropped hit
    and one missing case are each reported."""
    for name, fields in (("synth.json", SYNTH_FIELDS), ("sample_bin.json", SAMPLE_FIELDS), ("r900_filter.json", R900_FIELDS)):
        ours = _load(name)
        fake = json.loads(json.dumps({"generator": "stand-in", **copy.deepcopy(ours)}))   # through JSON, like the real file
        assert compare_goldens(ours, fake, fields) == []
        bad = copy.deepcopy(fake)
        bad["cases"][0]["qsha"] = "0" * 64
        assert any("qsha" in d for d in compare_goldens(ours, bad, fields))
        bad = copy.deepcopy(fake)
        del bad["cases"][-1]
        assert any("missing" in d for d in compare_goldens(ours, bad, fields))
    ours = _load("sample_bin.json")
    bad = copy.deepcopy(ours)
    hit_case = next(c for c in bad["cases"] if c["hits"])
    hit_case["hits"] = hit_case["hits"][:-1]
    assert any(".hits" in d for d in compare_goldens(ours, bad, SAMPLE_FIELDS))


def test_r900_filter_vectors_reproduce_from_the_oracle():
    """tests/golden/r900_filter.json (what go/cmd/amdgolden -tags amdgolden re-computes with the reference's own
    r900.Parser.filter) is what the oracle's literal C restatement gives today, and the numpy restatement
    (oracle/r900_oracle.py: the checker of the HIP digits) agrees with it at every position a hit could use."""
    import importlib.util
    from oracle import r900_oracle as r9
    from oracle.oracle import OracleDecoder, R900Filter
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    raw = util.load_capture()
    gold = {c["name"]: c for c in _load("r900_filter.json")["cases"]}
    c = gold["capture_r900_72"]
    qsha, hist, n = mg.r900_filter_run(c["protocols"], c["chip"], raw)
    assert (qsha, hist, n) == (c["qsha"], c["hist"], c["calls"])
    # numpy twin == C restatement on the last call's buffer, wherever filter() writes (idx < BufferLength - 4*CL)
    d = OracleDecoder(c["protocols"], c["chip"])
    f = R900Filter(d)
    bs2 = d.geom.block_size2
    for k in range(3):
        d.decode(raw[k * bs2:(k + 1) * bs2])
        q = f.step().copy()
    csum = np.concatenate([np.zeros(1, np.float32), np.cumsum(f.signal, dtype=np.float32)])
    pos = np.arange(d.geom.buffer_length - 4 * d.geom.chip_length)
    assert np.array_equal(r9.quantize_at(csum, pos, d.geom.chip_length), q[: len(pos)])


@pytest.mark.parametrize("name,fields", [("synth.json", SYNTH_FIELDS), ("sample_bin.json", SAMPLE_FIELDS),
                                         ("r900_filter.json", R900_FIELDS)])
def test_oracle_equals_go_reference(name, fields):
    go = _load("go_" + name)
    if go is None:
        pytest.skip(f"tests/golden/go_{name} not generated yet (go/README.md: one command, needs a Go toolchain): parity unpinned")
    diffs = compare_goldens(_load(name), go, fields)
    assert not diffs, "oracle golden vectors differ from the Go reference:\n" + "\n".join(diffs)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.gpu
def test_hip_equals_go_reference_synth():
    go = _load("go_synth.json")
    if go is None:
        pytest.skip("tests/golden/go_synth.json not generated yet")
    for c in go["cases"]:
        dec = util.make_decoder(c["protocols"], c["chip"])
        try:
            iq, _ = util.synth_stream(c["protocols"], c["chip"], c["blocks"], dec.Cfg.BlockSize, c["seed"], c["packets"])
            assert _sha(iq) == c["iq_sha"]
            q, h, p = util.gpu_run(dec, iq, [1] * c["blocks"])          # one block per call, as main.go:235
            assert _sha(q) == c["qsha"], f"{c['name']}: quantized bits differ from the Go reference"
            assert len(h) == c["n_hits"] and _sha(h.astype("<i8")) == c["hits_sha"], f"{c['name']}: hit list differs"
            assert _sha(p[:, : dec.Cfg.PacketSymbols // 8]) == c["pkt_sha"], f"{c['name']}: packet bytes differ"
        finally:
            dec.close()


@pytest.mark.gpu
def test_hip_equals_go_reference_capture():
    go = _load("go_sample_bin.json")
    if go is None:
        pytest.skip("tests/golden/go_sample_bin.json not generated yet")
    raw = util.load_capture()
    assert _sha(raw) == go["file_sha256"]
    for c in go["cases"]:
        dec = util.make_decoder(c["protocols"], c["chip"])
        try:
            nb = c["blocks"]
            q, h, p = util.gpu_run(dec, raw[: nb * dec.Cfg.BlockSize2], [1] * nb)
            assert _sha(q) == c["qsha"] and int(np.unpackbits(q).sum()) == c["ones"]
            assert h[:, 1:].tolist() == c["hits"]
            assert _sha(p) == c["pkt_sha"]
            assert _sha(dec.mag_lut().astype("<f4")) == go["lut_sha256"]
        finally:
            dec.close()
