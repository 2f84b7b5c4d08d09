#!/usr/bin/env python
"""Regenerates tests/golden/*.json and capture_iq.xz.  Run in the BUILD container (needs /root/reference for the
sample.bin vectors and the capture fixture; the synthetic vectors need nothing).  The JSON files are committed; tests on the
GPU box compare the HIP path against them without touching /root/reference.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.oracle import OracleDecoder  # noqa: E402
from oracle import np_oracle as npo      # noqa: E402
from tests import util                   # noqa: E402

SAMPLE = "/root/reference/assets/sample.bin"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sample_bin_vectors():
    raw = np.fromfile(SAMPLE, dtype=np.uint8)
    out = {"file_sha256": sha(raw), "file_bytes": int(raw.size), "lut_sha256": sha(npo.mag_lut().astype("<f4")),
           "cases": []}
    for name, protos, chip, nbytes in [("cfg1_first_512KiB_scm72", ["scm"], 72, 524288),
                                       ("whole_scm72", ["scm"], 72, None), ("whole_idm72", ["idm"], 72, None),
                                       ("whole_scm80", ["scm"], 80, None)]:
        d = OracleDecoder(protos, chip)
        nb = (nbytes or raw.size) // d.geom.block_size2
        q, hits, hb = d.decode_stream(raw[: nb * d.geom.block_size2])
        out["cases"].append({"name": name, "protocols": protos, "chip": chip, "blocks": nb,
                             "block_size": d.geom.block_size, "ones": int(np.unpackbits(q).sum()), "qsha": sha(q),
                             "hits": hits[:, [0, 2]].tolist(), "pkt_sha": sha(hb)})
    # the capture's true chip length: semantic search recovers CRC-valid SCM packets
    d = OracleDecoder(["scm"], 78)
    nb = raw.size // d.geom.block_size2
    q, hits, hb = d.decode_stream(raw[: nb * d.geom.block_size2], mode=1)
    from rtlamr_amd.contrib.parsers.crc import CRC
    bch = CRC("BCH", 0, 0x6F63, 0)
    valid = sorted({bytes(b).hex() for b in hb if bch.Checksum(bytes(b[2:12])) == 0})
    out["chip78_semantic"] = {"ones": int(np.unpackbits(q).sum()), "qsha": sha(q), "n_hits": int(len(hits)),
                              "crc_valid_packets": valid}
    return out


def synth_vectors():
    """Seeded synthetic streams: expected quantized-bit hash, hit list hash and hit count."""
    cases = []
    for name, protos, chip, blocks, seed, npk in [
            ("scm72_200", ["scm"], 72, 200, 1, 12), ("scm8_150", ["scm"], 8, 150, 18, 10),
            ("scm32_150", ["scm"], 32, 150, 42, 10), ("scm96_150", ["scm"], 96, 150, 106, 10),
            ("idm72_140", ["idm"], 72, 140, 4, 6), ("all72_140", ["scm", "scm+", "idm", "r900"], 72, 140, 5, 9)]:
        d = OracleDecoder(protos, chip)
        iq, _ = util.synth_stream(protos, chip, blocks, d.geom.block_size, seed, npk)
        o, q, h, p = util.oracle_run(protos, chip, iq)
        cases.append({"name": name, "protocols": protos, "chip": chip, "blocks": blocks, "seed": seed,
                      "packets": npk, "iq_sha": sha(iq), "qsha": sha(q), "n_hits": int(len(h)),
                      "hits_sha": sha(h.astype("<i8")), "pkt_sha": sha(p[:, : o.geom.packet_symbols // 8])})
    return {"cases": cases}


def r900_filter_run(protos, chip, iq):
    """The r900 parser's second stage over a stream: p.quantized after EVERY Decode call (r900/r900.go:160-172, 82-150),
    all calls concatenated -> (sha256, histogram of the six symbol values, calls)."""
    from oracle.oracle import R900Filter
    d = OracleDecoder(protos, chip)
    f = R900Filter(d)
    bs2 = d.geom.block_size2
    h = hashlib.sha256()
    hist = np.zeros(6, np.int64)
    n = iq.size // bs2
    for k in range(n):
        d.decode(iq[k * bs2:(k + 1) * bs2])
        q = f.step()
        h.update(q.tobytes())
        hist += np.bincount(q, minlength=6)[:6]
    return h.hexdigest(), hist.tolist(), n


def r900_filter_vectors(raw):
    """What go/cmd/amdgolden -tags amdgolden reproduces with the reference's own r900.Parser.filter (go/r900/amd_golden.go)."""
    cases = []
    for name, protos, chip in [("capture_r900_72", ["r900"], 72), ("capture_r900_32", ["r900"], 32),
                               ("capture_all_72", ["scm", "scm+", "idm", "r900"], 72)]:
        qsha, hist, n = r900_filter_run(protos, chip, raw)
        cases.append({"name": name, "protocols": protos, "chip": chip, "input": "capture", "calls": n, "qsha": qsha, "hist": hist})
    for c in json.load(open(os.path.join(HERE, "synth.json")))["cases"]:
        if "r900" not in c["protocols"]:
            continue
        d = OracleDecoder(c["protocols"], c["chip"])
        iq, _ = util.synth_stream(c["protocols"], c["chip"], c["blocks"], d.geom.block_size, c["seed"], c["packets"])
        assert sha(iq) == c["iq_sha"]
        qsha, hist, n = r900_filter_run(c["protocols"], c["chip"], iq)
        cases.append({"name": c["name"], "protocols": c["protocols"], "chip": c["chip"], "input": "synth", "calls": n,
                      "qsha": qsha, "hist": hist})
    return {"cases": cases}


def capture_fixture():
    """The reference's only real-signal input, assets/sample.bin (raw uint8 IQ -- data, not code), xz-compressed, so
    that the -m gpu tests can run BASELINE config 1 on the GPU box, where /root/reference does not exist."""
    import lzma
    raw = open(SAMPLE, "rb").read()
    with open(os.path.join(HERE, "capture_iq.xz"), "wb") as f:
        f.write(lzma.compress(raw, preset=9 | lzma.PRESET_EXTREME))


if __name__ == "__main__":
    if os.path.exists(SAMPLE):
        json.dump(sample_bin_vectors(), open(os.path.join(HERE, "sample_bin.json"), "w"), indent=1)
        capture_fixture()
    json.dump(synth_vectors(), open(os.path.join(HERE, "synth.json"), "w"), indent=1)
    json.dump(r900_filter_vectors(util.load_capture()), open(os.path.join(HERE, "r900_filter.json"), "w"), indent=1)
    print("golden vectors written")
