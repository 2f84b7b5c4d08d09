#!/usr/bin/env python
"""Golden results of bench.py's workloads, computed by the CPU ORACLE (never by the HIP path).

For every workload of bench.py (BASELINE.json configs 2-5) and every shard 0..7 of its 8-GPU form, the stream bench.py
generates on the device (SURVEY.md 8d generator, the packets of bench.build_packets) is generated here on the host by
oracle/synth_gen.c, decoded by oracle/decode_oracle.c (the literal restatement of protocol/decode.go) as ONE logical
Decoder, and the result is recorded as a hit count plus sha256 fingerprints of the hit list (pid, call index, idx), of
the sliced packet bytes and of the packed quantized bitstream -- once from the state a single Decoder carries into the
shard ("first") and once with the shard's own tail as history ("steady": what bench.py's timed steps see, which replay
one buffer), each also after the parsers' checksum / repeat filter ("validated").  bench.py compares what the GPU
produced with these, tests/test_gpu_fullsize.py does the same for the single-GPU shards.

    python tests/golden/make_bench_golden.py                 # everything (a few minutes on 8 cores)
    python tests/golden/make_bench_golden.py --workloads cfg2 cfg4:8 --shards 0 1

Writes tests/golden/bench_golden.json ("source": "oracle").  Runs on CPU only; reads nothing under /root/reference.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import bench  # noqa: E402  (workload table and packet schedule: the same code the GPU run uses)
from oracle import oracle as orc  # noqa: E402
from oracle import validate_oracle as vo  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "bench_golden.json")
ALL = ["cfg2", "cfg3", "cfg5"] + [f"cfg4:{c}" for c in bench.LEGAL_CHIPS]


def _digest(wl, g, rows, pkt, q) -> dict:
    """digests of the full result + of what Decoder.EnableValidation would leave of it (oracle/validate_oracle.py:
    checksum tests and adjacent-repeat removal per validated preamble; the r900 preamble is never filtered)."""
    d = orc.result_digest(rows, pkt, q)
    d["per_preamble"] = [int((rows[:, 0] == p).sum()) for p in range(g.n_preambles)]
    keep = []
    done = set()
    probe = orc.OracleDecoder(wl["protos"], wl["chip"])
    for name, pid in zip(wl["protos"], probe.preamble_ids):
        if pid in done:
            continue
        done.add(pid)
        sel = np.flatnonzero(rows[:, 0] == pid)
        keep.append(sel[vo.filter_hits_np(name, rows[sel, 1], pkt[sel])] if name in vo.RULES else sel)
    keep = np.concatenate(keep)
    d["validated"] = orc.result_digest(rows[keep], pkt[keep], None)
    return d


def golden_for(spec: str, shard: int, n_blocks: int = 0, threads: int = 0, uniform: bool = False) -> dict:
    """{"first": the shard decoded from the state a single Decoder carries into it (stream start, or the blocks
    bench.device_workload primes with), "steady": the shard decoded with its OWN tail as history -- what every
    timed step of bench.py after the first sees, since the steps replay one buffer}."""
    wl = bench.workload(spec)
    probe = orc.OracleDecoder(wl["protos"], wl["chip"])
    g = probe.geom
    bs, bs2 = g.block_size, g.block_size2
    n_blocks = n_blocks or wl["nbytes"] // bs2
    n_samples = n_blocks * bs
    pk = [] if uniform else bench.build_packets(wl, shard, bs, n_samples)     # --data uniform: random bytes, nothing planted
    iq = orc.synth_stream(n_samples, 1, shard * n_samples, pk, wl["chip"], threads, uniform=uniform)
    hb = (g.packet_length + bs - 1) // bs + 2
    out = {}
    for state in ("first", "steady"):
        if state == "steady":
            head = iq[-hb * bs2:]
        elif shard > 0:      # the blocks bench.py primes a shard's decoder with: same noise, same packets
            prev = [] if uniform else bench.build_packets(wl, shard - 1, bs, n_samples)[-8:]
            head = orc.synth_stream(hb * bs, 1, shard * n_samples - hb * bs, prev + pk[:1], wl["chip"], threads, uniform=uniform)
        else:
            head = iq[:0]
        nh = head.size // bs2
        q, rows, pkt = orc.decode_sharded(wl["protos"], wl["chip"], np.concatenate([head, iq]) if nh else iq, threads,
                                          first_block=nh)
        rows[:, 1] += shard * n_blocks - nh
        out[state] = _digest(wl, g, rows, pkt, q)
    return out


def key(spec: str, n_blocks: int, shard: int, uniform: bool = False) -> str:
    return f"{spec}|blocks={n_blocks}|shard={shard}" + ("|data=uniform" if uniform else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", nargs="*", default=ALL)
    ap.add_argument("--shards", nargs="*", type=int, default=list(range(8)))
    ap.add_argument("--blocks", type=int, default=0, help="blocks per shard (default: the workload's size)")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--uniform", action="store_true", help="bench.py --data uniform: uniform random bytes, nothing planted")
    args = ap.parse_args()
    try:
        gold = json.load(open(OUT))
    except Exception:
        gold = {}
    gold["source"] = ("oracle: oracle/decode_oracle.c on streams from oracle/synth_gen.c, "
                      "made by tests/golden/make_bench_golden.py on CPU")
    for spec in args.workloads:
        wl = bench.workload(spec)
        probe = orc.OracleDecoder(wl["protos"], wl["chip"])
        nb = args.blocks or wl["nbytes"] // probe.geom.block_size2
        for shard in args.shards:
            t0 = time.time()
            k = key(wl["name"], nb, shard, args.uniform)
            gold[k] = golden_for(spec, shard, nb, args.threads, args.uniform)
            print(f"{k}: {gold[k]['first']['n_hits']} hits "
                  f"({time.time() - t0:.1f} s)", flush=True)
            json.dump(gold, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
