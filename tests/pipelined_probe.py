"""Helper of tests/test_gpu_robust.py: one pipelined run (three batches in flight, deferral optional) of a seeded synthetic
stream in a process of its own -- so that environment that must be set before the HIP runtime starts
(AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING) or before amr_create (AMR_GATE_TIMEOUT_US) can be -- printing one JSON line:
digests of the hit list and the packet bytes, the wall time of the pipelined part, and amr_describe's text.

    python tests/pipelined_probe.py <protos,comma> <chip> <blocks per batch> <batches> [validate]
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    import numpy as np
    from rtlamr_amd import _lib
    from tests import util
    protos, chip, per, n_batches = sys.argv[1].split(","), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    validate = len(sys.argv) > 5 and sys.argv[5] == "validate"
    L = _lib.lib()
    dec = util.make_decoder(protos, chip)
    bs2 = dec.Cfg.BlockSize2
    iq, _ = util.synth_stream(protos, chip, per * n_batches, dec.Cfg.BlockSize, seed=123, n_packets=3 * n_batches)
    if validate:
        dec.EnableValidation()
    d = C.c_void_p()
    _lib.check(L.amr_dev_alloc(0, iq.size, C.byref(d)), "alloc")
    _lib.check(L.amr_dev_upload(0, d, iq.ctypes.data, iq.size), "upload")
    got, inflight = [], 0
    t0 = time.perf_counter()
    for k in range(n_batches):
        dec.submit_device(d.value + k * per * bs2, per)
        inflight += 1
        if inflight == 3:
            got.append(dec.collect()); inflight -= 1
    while inflight:
        got.append(dec.collect()); inflight -= 1
    dt = time.perf_counter() - t0
    rows, pkts = [], []
    for br in got:
        for pid in range(dec.n_preambles):
            blk, idx, pk = br.for_preamble(pid)
            rows.append(np.stack([np.full(len(blk), pid, np.int64), blk.astype(np.int64), idx.astype(np.int64)], axis=1))
            pkts.append(np.ascontiguousarray(pk))
    rows, pkts = np.concatenate(rows), np.concatenate(pkts)
    order = np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0]))        # the oracle's order: preamble, call, idx
    rows, pkts = np.ascontiguousarray(rows[order]), np.ascontiguousarray(pkts[order])
    out = {"n_hits": int(len(rows)), "hits_sha256": hashlib.sha256(rows.tobytes()).hexdigest(),
           "pkt_sha256": hashlib.sha256(pkts.tobytes()).hexdigest(), "seconds": dt, "describe": dec.describe()}
    dec.close()
    L.amr_dev_free(0, d)
    print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
