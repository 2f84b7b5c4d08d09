#!/usr/bin/env python
"""bench.py -- IQ Msamples/s through Decoder.Decode semantics on N MI355X (BASELINE.json metric and configs).

One "step" = one pass of the whole hot path (K1 demod + K2 search + K3 slice + hit read-back) over one batch of
synthetic uint8 IQ per GPU, resident in HBM before the timed region starts.  Weak scaling: every rank owns one
batch-sized shard of an N-shard stream; ranks > 0 prime their decoder with the blocks preceding their shard; hits are
gathered on rank 0 over RCCL every step (N > 1).

Workloads (--workload, BASELINE.json configs 2-5; cfg1 is the CPU-only plumbing case, covered by the tests):
  cfg2          SCM, chip 72, 1 GiB per GPU, 4096 planted CRC-valid SCM packets            (default: the headline metric)
  cfg3          IDM, chip 72, 4 GiB per GPU, 4096 planted IDM packets
  cfg4:<chip>   SCM at chip length <chip> (8 32 40 48 56 64 72 80 88 96), 1 GiB per GPU
  cfg5          scm + scm+ + idm + r900 ("all" geometry), chip 72, 4 GiB per GPU, 4096 planted packets of the three
                Manchester protocols (parsers and the r900 second stage are outside the timed region)

Before the warm-up the GPU is spun up with untimed passes for --spinup-ms: the shader clock of an idle MI355X needs
tens of milliseconds of load to reach its steady value (K1 measured at 1.6 GHz in the first 5 ms of a fresh process,
2.4 GHz after 60 ms), and a 20-step timed region is only 6 ms long.  After the timed region the last step's complete
result (hit list, packet bytes, quantized bitstream) is compared by sha256 with what the CPU oracle computed for the
workload (tests/golden/bench_golden.json, made by tests/golden/make_bench_golden.py) and, on rank 0, a validating
decoder must turn the batch into exactly the planted messages; a mismatch makes the exit code non-zero.

N > 1: every rank's hits that pass the parsers' checksum tests on the GPU (K5; --gather raw: every hit) travel to rank 0
each step through the library's RCCL gather, and rank 0 reads every step's records inside the timed loop.

Launch:  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Without a launcher `--gpus N` (N > 1) starts the N ranks itself, one per device (rtlamr_amd/launch.py); `--gpus` that
disagrees with the launcher's WORLD_SIZE, or asks for more ranks than there are gfx950 devices, exits non-zero with a
one-line reason -- a request for N ranks never turns into a run of one.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GIB = 1 << 30
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
ALG_BYTES_PER_SAMPLE = 2.0   # K1 reads one (I,Q) uint8 pair per sample
GOLDEN = os.path.join(ROOT, "tests", "golden", "bench_golden.json")
LEGAL_CHIPS = (8, 32, 40, 48, 56, 64, 72, 80, 88, 96)


def workload(spec: str) -> dict:
    if spec == "cfg2":
        return dict(name="cfg2", protos=["scm"], chip=72, nbytes=1 * GIB, kinds=["scm"], n_packets=4096)
    if spec == "cfg3":
        return dict(name="cfg3", protos=["idm"], chip=72, nbytes=4 * GIB, kinds=["idm"], n_packets=4096)
    if spec.startswith("cfg4:"):
        chip = int(spec.split(":", 1)[1])
        if chip not in LEGAL_CHIPS:
            raise SystemExit(f"cfg4: chip length {chip} is not a legal -symbollength (flags.go:127-132)")
        return dict(name=f"cfg4:{chip}", protos=["scm"], chip=chip, nbytes=1 * GIB, kinds=["scm"], n_packets=4096)
    if spec == "cfg5":
        return dict(name="cfg5", protos=["scm", "scm+", "idm", "r900"], chip=72, nbytes=4 * GIB,
                    kinds=["scm", "scm+", "idm"], n_packets=4096)
    raise SystemExit(f"unknown workload {spec!r} (cfg2, cfg3, cfg4:<chip>, cfg5)")


def packet_builders():
    from rtlamr_amd.contrib.parsers.idm import build_idm_packet, build_scmplus_packet
    from rtlamr_amd.contrib.parsers.scm import build_packet
    return {
        "scm": (lambda i: build_packet(100000 + i, (i % 12) + 1, i * 37), 96),
        "idm": (lambda i: build_idm_packet(200000 + i, consumption=i * 31), 736),
        "scm+": (lambda i: build_scmplus_packet(300000 + i, consumption=i * 13), 128),
    }


def build_packets(wl: dict, shard: int, bs: int, n_samples: int):
    """The planted packets of one shard: CRC-valid, hash-chosen offsets, every 64th straddling a block edge."""
    from rtlamr_amd import synth
    B = packet_builders()
    kinds = wl["kinds"]
    longest = max(B[k][1] for k in kinds) * 2 * wl["chip"]
    base = shard * n_samples
    n_packets = max(1, min(wl["n_packets"], n_samples // (4 * longest)))      # --blocks N: fewer packets in a short stream
    starts = synth.packet_schedule(n_packets, n_samples, longest, seed=1 + shard, edge_every=64, block_size=bs)
    pk = []
    for i, s in enumerate(starts):
        fn, nbits = B[kinds[i % len(kinds)]]
        sign = 1 if i % 2 else -1
        pk.append(synth.Packet(int(base + s), fn(shard * wl["n_packets"] + i), nbits, sign * (22 + i % 17), -sign * (21 + i % 13)))
    return pk


def golden_key(wl: dict, n_blocks: int, shard: int) -> str:
    return f"{wl['name']}|blocks={n_blocks}|shard={shard}" + ("|data=uniform" if wl.get("data") == "uniform" else "")


def device_workload(dec, wl: dict, shard_idx: int, n_blocks: int, local_rank: int = 0):
    """Shard `shard_idx` of the workload's stream, generated in HBM (SURVEY.md 8d generator, K0), and -- for shards
    behind the first -- the decoder primed with the blocks in front of it (amr_prime), as a rank of a multi-GPU run
    does.  -> (device pointer as ctypes.c_void_p, the shard's planted packets).  tests/golden/make_bench_golden.py
    builds the same stream on the host for the oracle."""
    from rtlamr_amd import _lib, synth
    L = _lib.lib()
    chip = wl["chip"]
    bs, bs2 = dec.Cfg.BlockSize, dec.Cfg.BlockSize2
    n_samples = n_blocks * bs
    d_iq = C.c_void_p()
    _lib.check(L.amr_dev_alloc(local_rank, n_blocks * bs2, C.byref(d_iq)), "amr_dev_alloc")
    uni = wl.get("data") == "uniform"      # SURVEY.md 8d's second distribution: uniform random bytes, nothing planted
    pk = [] if uni else build_packets(wl, shard_idx, bs, n_samples)
    synth.device_fill(local_rank, d_iq.value, n_samples, seed=1, first_sample=shard_idx * n_samples, packets=pk,
                      chip_length=chip, uniform_bytes=uni)
    if shard_idx > 0:   # rebuild the history a single decoder would carry into this shard
        pb = dec.prime_blocks()
        hb = pb + 1
        d_h = C.c_void_p()
        _lib.check(L.amr_dev_alloc(local_rank, hb * bs2, C.byref(d_h)), "amr_dev_alloc")
        prev = [] if uni else build_packets(wl, shard_idx - 1, bs, n_samples)
        synth.device_fill(local_rank, d_h.value, hb * bs, seed=1, first_sample=shard_idx * n_samples - hb * bs,
                          packets=prev[-8:] + pk[:1], chip_length=chip, uniform_bytes=uni)
        dec.prime_device(d_h.value + bs2, pb, d_lead=d_h.value + bs2 - dec.halo_bytes())
        dec.set_block_base(shard_idx * n_blocks)
        _lib.check(L.amr_dev_free(local_rank, d_h), "amr_dev_free")
    return d_iq, pk


def result_digest(rows, pkt, q) -> dict:
    """sha256 fingerprints of a decode result in the canonical form of oracle.result_digest (restated here so that the
    timed path of bench.py imports nothing from oracle/): rows int64[n,3] (pid, call, idx) preamble-major, packet
    bytes in the same order, packed bitstream MSB first."""
    import hashlib
    import numpy as np
    return {"n_hits": int(len(rows)),
            "hits_sha256": hashlib.sha256(np.ascontiguousarray(rows, np.int64).tobytes()).hexdigest(),
            "pkt_sha256": hashlib.sha256(np.ascontiguousarray(pkt, np.uint8).tobytes()).hexdigest(),
            "q_sha256": hashlib.sha256(np.ascontiguousarray(q, np.uint8).tobytes()).hexdigest()}


def cpu_baseline(wl, dec, d_iq, bs2, seconds=12.0):
    """The C restatement of the reference Go path (oracle/decode_oracle.c, kind "port") timed on this
    host: same IQ bytes (first 64 MiB of rank 0's shard), one independent stream per thread."""
    import numpy as np
    from oracle.oracle import OracleDecoder
    from rtlamr_amd import _lib
    nblk = (64 << 20) // bs2
    sample = np.empty(nblk * bs2, np.uint8)
    _lib.check(_lib.lib().amr_dev_download(dec.device_id, sample.ctypes.data, C.c_void_p(d_iq), sample.size), "download")
    ncores = os.cpu_count() or 1

    def run(nthreads, budget):
        done = [0] * nthreads
        decs = [OracleDecoder(wl["protos"], wl["chip"]) for _ in range(nthreads)]
        t_end = time.perf_counter() + budget

        def work(i):
            while time.perf_counter() < t_end:
                decs[i].decode_stream(sample, want_q=False)   # ctypes releases the GIL
                done[i] += nblk
        th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.perf_counter() - t0
        return sum(done) * (bs2 // 2) / dt / 1e6

    one = run(1, seconds * 0.4)
    allc = run(ncores, seconds * 0.6) if ncores > 1 else one
    return {"value": round(allc, 1), "unit": "Msamples/s", "cores": ncores, "kind": "port",
            "single_thread": round(one, 1),
            "sample": f"first 64 MiB ({nblk} blocks) of the bench IQ, replayed per thread for ~{seconds:.0f} s; "
                      "C restatement of protocol/decode.go (gcc -O2 -ffp-contract=off), one stream per thread"}


def measure_traffic(spec: str, blocks: int, k1_full: str):
    """K1's HBM bytes per launch, measured now: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (TCC slots) over a
    short run of this workload with one batch in flight (with counters the profiler runs one kernel at a time), units and
    the gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md prescribes: KiB, wide coalesced reads reported at half.
    -> (bytes per launch, description) or (None, why not)."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3")
    if not rp:
        return None, "rocprofv3 not on this box"
    norm = lambda t: "".join(str(t).split())
    vals = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        env = dict(os.environ, TMPDIR="/tmp")
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [rp, "--kernel-trace", "--pmc", ctr, "-d", out, "-o", "pmc", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__), "--workload", spec, "--steps", "3", "--warmup", "1", "--depth", "1", "--k1-events", "0",
                   "--no-cpu-baseline", "--no-verify", "--spinup-ms", "0", "--device-state", "off", "--no-measure-traffic"] + (["--blocks", str(blocks)] if blocks else [])
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, timeout=150)
            except Exception as e:
                return None, f"rocprofv3 --pmc {ctr} failed: {e}"
            got = collections.defaultdict(list)
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == ctr:
                        got[norm(r["Kernel_Name"].split("(")[0])].append(float(r["Counter_Value"]))
            v = got.get(norm(k1_full))
            if not v:
                return None, f"no {ctr} rows for {k1_full} in the rocprofv3 output"
            vals[ctr] = (sum(v) / len(v), len(v))
    rd, wr = vals["FETCH_SIZE"][0] * 1024 * 2, vals["WRITE_SIZE"][0] * 1024
    return rd + wr, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over {vals['FETCH_SIZE'][1]} launches of "
                     f"{k1_full} (--depth 1); KiB x 1024, FETCH_SIZE x 2 (gfx950 reports half of wide coalesced reads): "
                     f"{rd:.0f} B read + {wr:.0f} B written per launch")


def single_block_latency(ra, wl, local_rank, d_iq, bs, bs2, n=300, warm=50):
    """The latency case, the unchanged main.go loop (main.go:235): ONE block per Decode call, from host memory, results back
    on the host before the next call.  Not what the GPU is for -- reported so that nobody has to guess (us per call)."""
    import numpy as np
    from rtlamr_amd import _lib
    iq = np.empty((n + warm) * bs2, np.uint8)
    _lib.check(_lib.lib().amr_dev_download(local_rank, iq.ctypes.data, C.c_void_p(d_iq), iq.size), "download")
    dec = ra.new_decoder(local_rank)
    try:
        for p in wl["protos"]:
            dec.RegisterProtocol(ra.new_parser(p, wl["chip"]))
        dec.Allocate()
        for k in range(warm):
            dec.decode_batch(iq[k * bs2:(k + 1) * bs2])
        t0 = time.perf_counter()
        for k in range(warm, warm + n):
            dec.decode_batch(iq[k * bs2:(k + 1) * bs2])
        return (time.perf_counter() - t0) / n * 1e6
    finally:
        dec.close()


def verify_batch(ra, wl, local_rank, d_iq, n_blocks, pk, bs, n_samples):
    """A second, validating decoder turns the batch into messages: exactly the planted meters must come out."""
    dec = ra.new_decoder(local_rank)
    try:
        for p in wl["protos"]:
            dec.RegisterProtocol(ra.new_parser(p, wl["chip"]))
        dec.Allocate()
        dec.EnableValidation()
        br = dec.decode_batch_device(d_iq, n_blocks)
        got = set()
        for msgs in dec.run_parsers(br):
            for m in msgs:
                got.add((m.MsgType(), m.MeterID()))
        pl = dec.Cfg.PacketLength
        want = set()
        ids = {"scm": "SCM", "idm": "IDM", "scm+": "SCM+"}
        B = packet_builders()
        for i, p in enumerate(pk):
            if p.start + 2 * wl["chip"] + pl + bs > n_samples:   # its last call lies beyond the batch
                continue
            kind = wl["kinds"][i % len(wl["kinds"])]
            base = {"scm": 100000, "idm": 200000, "scm+": 300000}[kind]
            want.add((ids[kind], base + i))
        missing = want - got
        extra = {g for g in got if g not in want and g[0] in ids.values()}
        return len(want), len(missing), len(extra)
    finally:
        dec.close()


def sharded_gather_check(ra, shard, wl, local_rank, rank, world, n_blocks=256):
    """Every rank decodes its block range of one small stream (primed with the blocks before it), the hits are gathered
    through the C ABI, and rank 0 compares them with its own single-decoder result for the whole stream."""
    import numpy as np
    import torch.distributed as dist
    from rtlamr_amd import synth

    def mk():
        d = ra.new_decoder(local_rank)
        for p in wl["protos"]:
            d.RegisterProtocol(ra.new_parser(p, wl["chip"]))
        d.Allocate()
        return d
    dec = mk()
    try:
        bs, bs2 = dec.Cfg.BlockSize, dec.Cfg.BlockSize2
        iq = synth.noise(n_blocks * bs, seed=77)
        B = packet_builders()
        kinds = wl["kinds"]
        longest = max(B[k][1] for k in kinds) * 2 * wl["chip"]
        starts = synth.packet_schedule(12, n_blocks * bs, longest, seed=5, edge_every=3, block_size=bs)
        pk = []
        for i, s in enumerate(starts):
            fn, nbits = B[kinds[i % len(kinds)]]
            pk.append(synth.Packet(int(s), fn(900 + i), nbits, 27 if i % 2 else -27, -25 if i % 2 else 25))
        synth.plant(iq, pk, wl["chip"])
        g = shard.CommGatherer(dec, cap_hits=1 << 16)
        k0, k1 = shard.shard_range(n_blocks, world, rank)
        p0, _ = shard.prime_range(k0, dec.prime_blocks())
        if k0 > p0:
            dec.prime(iq[p0 * bs2: k0 * bs2], iq[p0 * bs2 - dec.halo_bytes(): p0 * bs2] if p0 > 0 else None)
        dec.set_block_base(k0)
        dec.decode_batch(iq[k0 * bs2: k1 * bs2])
        g.post()
        got = g.result()
        g.wait()
        ok, detail = True, f"{n_blocks}-block stream over {world} rank(s), C ABI gather: gathered hits == single decoder"
        if rank == 0:
            one = mk()
            try:
                br = one.decode_batch(iq)
                want = shard.batch_hits_array(br, one.n_preambles)
            finally:
                one.close()
            order = np.lexsort((got[:, 2], got[:, 1], got[:, 0]))
            ok = len(want) > 0 and np.array_equal(got[order], want)
            if not ok:
                detail = f"MISMATCH (C ABI gather): gathered {len(got)} hit records, single decoder {len(want)}"
        flag = [ok]
        dist.broadcast_object_list(flag, src=0)
        return bool(flag[0]), detail
    finally:
        dec.close()


def _hip_pci_bus_id(index: int):
    """PCI address of HIP device `index` ("0000:05:00.0") through the runtime that is loaded anyway, or None."""
    try:
        hip = C.CDLL("libamdhip64.so")
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0:
            return buf.value.decode().lower()
    except Exception:   # noqa: BLE001
        pass
    return None


def device_state(index: int) -> dict:
    """What the kernel driver says about device `index` right now: current shader / memory clock, socket power and its
    cap, the DPM tables with the active level, compute / memory partition mode -- read from sysfs
    (/sys/class/drm/card*/device: hwmon freq1_input, freq2_input, power1_input, power1_cap; pp_dpm_*; current_*_partition).
    Recorded while the pipeline is busy, before and after the timed region, never inside it, so that a slow line can be
    told from a regression (one binary measured 0.62 ... 0.72 of the roofline on different boxes).
    Plain file reads on purpose: `rocm-smi` as a child process costs the parent 0.45 ms inside the NEXT timed region
    (copy-on-write faults after the fork; measured A/B, profiles/r06/README.md)."""
    import glob
    try:
        want = _hip_pci_bus_id(index)
        cards = []
        for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device"), key=lambda p: int(p.split("card")[1].split("/")[0])):
            if os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                cards.append(d)
        if not cards:
            return {"unavailable": "no amdgpu card with pp_dpm_sclk under /sys/class/drm"}
        dev = None
        for d in cards:
            if want and os.path.basename(os.path.realpath(d)).lower() == want:
                dev = d
        how = "matched by PCI address" if dev else "by ordinal"
        dev = dev or cards[min(index, len(cards) - 1)]

        def rd(rel):
            try:
                with open(os.path.join(dev, rel)) as f:
                    return f.read().strip()
            except OSError:
                return None
        hw = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
        hwd = os.path.relpath(hw[0], dev) if hw else None

        def num(rel, scale):
            v = rd(os.path.join(hwd, rel)) if hwd else None
            return round(int(v) / scale, 1) if v and v.lstrip("-").isdigit() else None

        def active(table):
            for line in (table or "").splitlines():
                if line.rstrip().endswith("*"):
                    return line.replace("*", "").strip()
            return None
        return {"card": os.path.basename(os.path.dirname(dev)), "selected": how, "pci": os.path.basename(os.path.realpath(dev)),
                "sclk_mhz": num("freq1_input", 1e6), "mclk_mhz": num("freq2_input", 1e6),
                "power_w": num("power1_input", 1e6), "power_cap_w": num("power1_cap", 1e6),
                "sclk_level": active(rd("pp_dpm_sclk")), "sclk_levels": (rd("pp_dpm_sclk") or "").replace("\n", " | "),
                "fclk_level": active(rd("pp_dpm_fclk")), "socclk_level": active(rd("pp_dpm_socclk")),
                "compute_partition": rd("current_compute_partition"), "memory_partition": rd("current_memory_partition")}
    except Exception as e:   # noqa: BLE001 -- a missing file must not fail the bench
        return {"unavailable": f"{type(e).__name__}: {str(e)[:120]}"}


def n_cus_of(describe: str):
    """compute units from amr_describe's line ("... gfx950:sramecc+:xnack- 256 CUs clock ...")."""
    import re
    m = re.search(r"(\d+) CUs", describe)
    return int(m.group(1)) if m else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", help="cfg2 (default), cfg3, cfg4:<chip>, cfg5")
    ap.add_argument("--blocks", type=int, default=0, help="blocks per GPU per step (default: the workload's size)")
    ap.add_argument("--data", choices=["synthetic", "uniform"], default="synthetic",
                    help="synthetic: 2.4 Msps-shaped noise (binomial around 127/128) + planted CRC-valid packets (default, the "
                         "headline); uniform: uniform random bytes, nothing planted -- SURVEY.md 8d's second distribution, the "
                         "worst case for the magnitude LUT's LDS gathers")
    ap.add_argument("--spinup-ms", type=float, default=250.0, help="untimed passes before the warm-up (shader clock ramp)")
    ap.add_argument("--depth", type=int, default=3, help="batches in flight (1..3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--device-state", choices=["load", "off"], default="load",
                    help="sysfs snapshots (clocks, power, partition) under load before and after the timed region")
    ap.add_argument("--no-verify", action="store_true", help="skip the golden hit count / planted message check")
    ap.add_argument("--k1-events", type=int, default=4,
                    help="HIP events around the K1 dispatch of every N-th timed step (0 = none: roofline fields are NaN)")
    ap.add_argument("--k1-level", type=int, default=1, choices=[1, 2],
                    help="what those timed steps carry: 1 = K1's start / stop (search_ms then comes from the warm-up steps), 2 = K2 and K3.. as well")
    ap.add_argument("--measure-traffic", action="store_true",
                    help="N = 1: after the timed region, measure K1's HBM bytes per launch in THIS run: two rocprofv3 --pmc passes "
                         "(FETCH_SIZE, WRITE_SIZE; kernel trace only) of a short --depth 1 run of the same workload; needs rocprofv3 "
                         "on the box and takes about a minute.  On by default for the full line (N = 1, CPU baseline not skipped); "
                         "when it cannot be measured roofline.traffic falls back to the committed figure of the same kernel and says so")
    ap.add_argument("--no-measure-traffic", action="store_true", help="never start the rocprofv3 passes")
    ap.add_argument("--validate", action="store_true",
                    help="also run the parsers' checksum tests + repeat removal on the GPU (K5); only surviving hits are read back")
    ap.add_argument("--gather", choices=["validated", "raw"], default="validated",
                    help="N > 1: what travels to rank 0 every step -- the hits that pass the parsers' checksum tests on the "
                         "GPU (default; turns --validate on) or every hit the search found")
    args = ap.parse_args()
    wl = workload(args.workload)
    wl["data"] = args.data

    # ---- who am I: a rank of a launcher's job, the only rank, or the process that has to start the ranks ----
    from rtlamr_amd import launch
    launched = "WORLD_SIZE" in os.environ
    torch = None
    if launched or os.environ.get("AMR_BENCH_FORCE_DIST") == "1":
        import torch  # noqa: F811  (first, so its HIP runtime is the one in the process)
        n_devices = torch.cuda.device_count()
    else:
        from rtlamr_amd import _lib as _l0
        n_devices = _l0.device_count()
    try:
        mode, envs = launch.rank_plan(args.gpus, os.environ, n_devices)
    except launch.LaunchError as e:
        print(f"bench.py: {e}", file=sys.stderr)
        return 2
    if mode == "spawn":
        return launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], envs)
    if args.gather == "raw" and args.gpus > 2:
        print("bench.py: --gather raw is refused above 2 ranks: every step's raw hit list of every rank (3.5 MB each at cfg2) "
              "would pass through rank 0's host inside the timed loop; the validated records (K5) are what a deployment gathers",
              file=sys.stderr)
        return 2

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("AMR_BENCH_FORCE_DIST") == "1"   # the env: exercise the RCCL path with one rank
    if distributed:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import numpy as np
    import rtlamr_amd as ra
    from rtlamr_amd import _lib, dist as shard, synth

    L = _lib.lib()
    chip = wl["chip"]
    dec = ra.new_decoder(local_rank)
    for p in wl["protos"]:
        dec.RegisterProtocol(ra.new_parser(p, chip))
    dec.Allocate()
    bs, bs2 = dec.Cfg.BlockSize, dec.Cfg.BlockSize2
    n_blocks = args.blocks or wl["nbytes"] // bs2
    n_samples = n_blocks * bs
    nbytes = n_blocks * bs2

    # developer hook: build the workload of shard AMR_BENCH_SHARD on a single GPU (exercises the priming path of
    # ranks > 0 without a second GPU); the process stays rank 0 of a world of 1
    shard_idx = int(os.environ.get("AMR_BENCH_SHARD", rank))
    d_iq, pk = device_workload(dec, wl, shard_idx, n_blocks, local_rank)

    dev = torch.device("cuda", local_rank) if distributed else None
    gatherer = None
    gather_kind = "none (single GPU)"
    check = {}
    rc = 0
    try:
        gold = json.load(open(GOLDEN))
    except Exception:
        gold = {}
    key = golden_key(wl, n_blocks, shard_idx)
    verify = not args.no_verify
    aligned = n_blocks % 64 == 0       # otherwise deferral shifts the calls a result covers from step to step

    def sync_all():
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
        _lib.check(L.amr_dev_sync(local_rank), "amr_dev_sync")
        if distributed:
            dist.barrier()

    def rows_pkt(br):
        """canonical form of a result: int64 rows (pid, call index within the N-shard stream, idx), packet bytes"""
        rows, pkts = [], []
        for pid in range(dec.n_preambles):
            blk, idx, pkt = br.for_preamble(pid)
            rows.append(np.stack([np.full(len(blk), pid, np.int64), blk.astype(np.int64) - (br.first_block - shard_idx * n_blocks),
                                  idx.astype(np.int64)], axis=1))
            pkts.append(pkt)
        return np.concatenate(rows), np.concatenate(pkts)

    def against_golden(br, state, label):
        """sha256 of the complete result of one batch -- hit list, packet bytes, packed quantized bitstream -- against
        what the CPU oracle computed for this (workload, size, shard, decoder state); tests/golden/make_bench_golden.py,
        nothing in that file ever came from a GPU."""
        nonlocal rc
        if not verify:
            return
        if key not in gold:
            check[label] = f"no oracle golden for {key} (tests/golden/make_bench_golden.py --blocks {n_blocks})"
            return
        g = gold[key][state]
        rows, pkts = rows_pkt(br)
        got = result_digest(rows, pkts, dec.quantized_packed())
        want = dict(g["validated"], q_sha256=g["q_sha256"]) if validating else {k: g[k] for k in got}
        bad = [k for k in got if got[k] != want[k]] + ([] if br.n_hits_searched == g["n_hits"] else ["n_hits_searched"])
        what = "validated hit list + packet bytes" if validating else "hit list + packet bytes"
        check[label] = (f"{br.n_hits_searched} hits searched, {what} + the whole quantized bitstream: sha256 == oracle golden "
                        f"['{state}']" if not bad else f"MISMATCH against the oracle golden ['{state}'] in {bad}")
        rc = rc or (3 if bad else 0)

    # ---- N > 1: the hit gather.  Validated hits by default (the parsers' checksum + repeat filter on the GPU, K5): a
    # step's 290 131 raw hits are 4 169 records after it, and the root consumes EVERY step's records inside the timed loop
    validating = args.validate or (distributed and args.gather != "raw")
    if validating:
        dec.EnableValidation()
    try:
        dec.SetDeferral(True)      # any block count per batch at the full rate (--blocks N); no-op for multiples of 64
        deferral = True
    except _lib.AmrError:
        deferral = False           # the r900 second stage reads the batch's IQ by block (cfg5)

    # the first batch, from the state a single Decoder carries into this shard: against the oracle golden ["first"]
    first = dec.decode_batch_device(d_iq.value, n_blocks)
    against_golden(first, "first", "first_batch")
    n_first = len(first.hit_idx)
    gstat = {"consumed": 0, "records": 0, "bad": [], "checksum": 0, "local": {}}
    if distributed:
        t = torch.tensor([n_first], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        cap = max(1024, int(int(t.item()) * 1.5))

        # Every decision below is taken by all ranks together (all_reduce MIN), so that no rank waits in a collective
        # the others never enter.
        def all_agree(flag: bool) -> bool:
            t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))
        try:
            shard.comm_unique_id()            # loads librccl through the library: fails here, not inside a collective
            loadable = True
        except Exception as e:
            loadable = False
            check["sharded_gather"] = f"C-ABI gather unavailable ({e})"
        ok = all_agree(loadable)
        if ok:
            # Before anything is timed: a 256-block stream sharded over the ranks (HIP engine + amr_prime + the gather)
            # must give exactly what rank 0's single decoder gives for the whole stream.
            try:
                ok, detail = sharded_gather_check(ra, shard, wl, local_rank, rank, world)
            except Exception as e:
                ok, detail = False, f"C-ABI gather failed ({e})"
            check["sharded_gather"] = detail
            ok = all_agree(ok)
        if not ok:
            rc = 5
            print(f"bench.py: the multi-GPU hit gather is not usable: {check}", file=sys.stderr)
            dist.destroy_process_group()
            return rc
        gatherer = shard.CommGatherer(dec, cap_hits=cap)
        gather_kind = ("C ABI amr_gather_hits: RCCL send/recv of (call index, idx) records to rank 0 on the library's own stream, "
                       "one per step, no host synchronisation; rank 0 reads every step's records from the pinned mirror "
                       "(amr_gather_fetch) inside the timed loop, one gather behind")

    def consume(seq):
        """Rank 0, inside the timed loop: the records every rank contributed to gather `seq`, read from the library's
        pinned mirror -- header checks, every record touched, rank 0's own records compared with its local result."""
        for r in range(world):
            n_true, off, blk, idx = gatherer.fetch(seq, r, copy=False)
            if n_true != len(blk):
                gstat["bad"].append(f"gather {seq} rank {r}: truncated ({n_true} > {len(blk)})")
            gstat["checksum"] += int(blk.sum(dtype=np.uint64)) + int(idx.sum(dtype=np.uint64))
            gstat["records"] += len(blk)
            if r == rank and seq in gstat["local"]:
                lb, li = gstat["local"].pop(seq)
                if not (np.array_equal(lb, blk) and np.array_equal(li, idx)):
                    gstat["bad"].append(f"gather {seq}: rank 0's gathered records differ from its local result")
        gstat["consumed"] += 1

    def finish():
        """Collect the oldest batch: read back its hits and (N > 1) gather them on rank 0 over RCCL."""
        br = dec.collect(copy=False)
        if distributed:   # records go device -> RCCL -> rank 0, asynchronously, behind the kernels of the next batches
            seq = gatherer.post()                  # C ABI: amr_gather_hits, no host synchronisation
            if rank == 0:
                if validating:
                    gstat["local"][seq] = (np.array(br.hit_block, np.uint64), np.array(br.hit_idx, np.uint32))
                if seq >= 1:
                    consume(seq - 1)               # posted one step ago: normally there already
        return br

    def run(n, level, every=1, at_half=None):
        """n steps through the pipeline, --depth batches in flight (3: the GPU runs batches i+1 and i+2 while the host
        reads back batch i, and K3 of batch i runs next to the search of batch i+1).
        Steps 0, every, 2*every, ... carry timing events of the given level (every=0: none)."""
        out, timed = [], []

        def submit(i):
            t = every > 0 and i % every == 0
            dec.set_timing(level if t else 0)
            dec.submit_device(d_iq.value, n_blocks)
            timed.append(t)

        depth = max(1, min(args.depth, 3))
        for i in range(min(depth - 1, n)):
            submit(i)
        done = 0
        for i in range(depth - 1, n):
            submit(i)
            if at_half is not None and i == n // 2:
                at_half()
            out.append((finish(), dec.timing() if timed[done] else None))
            done += 1
        while done < n:
            out.append((finish(), dec.timing() if timed[done] else None))
            done += 1
        return out

    # ---- spin-up: untimed passes until the shader clock has ramped; they also give the steady-state step time ----
    spin_steps, steady_ms = 0, float("nan")
    if args.spinup_ms > 0:
        chunk = max(8, int(8e-3 / max(1e-6, 0.28e-3 * nbytes / GIB)))     # ~8 ms of batches per chunk
        t_spin = time.perf_counter()
        pairs = []                       # (ms for c steps, ms for 2c steps): the slope is the steady step, fill excluded
        def more_spin():
            go = (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms
            if distributed:          # one decision for all ranks: every step posts a collective gather
                tg = torch.tensor([1 if go else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(tg, op=dist.ReduceOp.MAX)
                go = bool(tg.item())
            return go
        while more_spin():
            t_pair = []
            for n in (chunk, 2 * chunk):
                sync_all()
                t1 = time.perf_counter()
                run(n, 0, 0)
                if distributed:
                    gatherer.wait()
                sync_all()
                t_pair.append((time.perf_counter() - t1) * 1e3)
                spin_steps += n
            pairs.append((t_pair[1] - t_pair[0]) / chunk)
        steady_ms = float(np.median(pairs[-3:])) if pairs else float("nan")

    # Timing events ride on the kernel dispatches (a separate event record costs a ~5 us stream bubble each).  The warm-up
    # steps carry the full set (K1, K2 and K3..: search_ms); every --k1-events-th TIMED step carries K1's start / stop alone
    # (level 1: the roofline figure).  Measured on the driver's command (profiles/r06/fill.txt): the full set on every 4th of
    # 20 timed steps costs 2 % of the line (a stop event on the search is a completion signal in front of the next K1
    # launch), K1's pair alone 0.4 %; K1's average is the same either way.
    def state_under_load():
        """device_state() of this rank's GPU read WHILE the pipeline is busy: 64 untimed steps, the files are read when half of
        them have been submitted (an idle device drops its shader clock within milliseconds: a snapshot of an idle device
        says nothing about the timed steps)."""
        if args.device_state == "off":
            return None
        box = {}
        run(64, 0, 0, at_half=lambda: box.update(device_state(local_rank)))
        if distributed:
            gatherer.wait()
        sync_all()
        return box

    state_before = state_under_load()
    warm = run(max(args.warmup, 1), 2) if args.warmup else []
    sync_all()
    c0, r0 = gstat["consumed"], gstat["records"]
    t0 = time.perf_counter()
    res = run(args.steps, args.k1_level, args.k1_events)
    if distributed:
        gatherer.wait()
    sync_all()
    dt = time.perf_counter() - t0
    consumed_timed, records_timed = gstat["consumed"] - c0, gstat["records"] - r0
    tms = [t for _, t in res if t is not None] or [t for _, t in warm]
    demod_ms = [t["demod_ms"] for t in tms] or [float("nan")]
    full = [t for _, t in (res if args.k1_level >= 2 else warm) if t is not None] or tms     # the steps that timed K2 as well
    search_ms = [t["search_ms"] for t in full] or [float("nan")]
    last = res[-1][0]
    n_hits = len(last.hit_idx)
    n_searched = last.n_hits_searched
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- what was timed is checked ----
    # the last timed step saw the shard's own tail as history (the steps replay one buffer): oracle golden ["steady"]
    if aligned:
        against_golden(last, "steady", "last_timed_step")
    else:
        check["last_timed_step"] = f"{n_searched} hits searched (no digest: with {n_blocks} blocks per batch the calls a result covers shift from step to step)"
    if distributed and rank == 0:
        # the last gather, complete: every rank's records against the oracle golden of ITS shard, by sha256
        seq = gatherer.last_seq
        consume(seq)
        for r in range(world):
            n_true, off, blk, idx = gatherer.fetch(seq, r)
            rows = shard.rows_from_gathered(off, blk, idx)
            rows[:, 1] -= last.first_block - shard_idx * n_blocks   # every rank has made the same number of calls
            kr = golden_key(wl, n_blocks, r if world > 1 else shard_idx)
            if verify and aligned and kr in gold:
                g = gold[kr]["steady"]
                want = (g["validated"] if validating else g)["hits_sha256"]
                import hashlib
                if hashlib.sha256(np.ascontiguousarray(rows, np.int64).tobytes()).hexdigest() != want:
                    gstat["bad"].append(f"last gather: rank {r}'s records differ from the oracle golden of its shard")
        check["gather"] = (f"{gstat['consumed']} gathers consumed on rank 0 ({consumed_timed} inside the timed region, "
                           f"{records_timed} records), rank 0's records == its local result every step, the last gather's "
                           f"records of all {world} rank(s) == oracle golden" if not gstat["bad"] else
                           "MISMATCH: " + "; ".join(gstat["bad"][:4]))
        rc = rc or (6 if gstat["bad"] else 0)
    if verify and rank == 0 and shard_idx == 0 and not validating and wl["data"] != "uniform":
        n_want, n_missing, n_extra = verify_batch(ra, wl, local_rank, d_iq.value, n_blocks, pk, bs, n_samples)
        ok = n_missing == 0 and n_extra == 0 and n_want > 0
        check["planted"] = (f"{n_want} planted messages recovered, none missing, none unexpected" if ok else
                            f"MISMATCH: {n_missing} of {n_want} planted messages missing, {n_extra} unexpected")
        rc = rc or (0 if ok else 4)

    state_after = state_under_load()      # (after every check of the timed steps' results: it reuses their buffers)

    if rank == 0:
        total_samples = float(world) * args.steps * n_samples
        k1_ms = float(np.mean(demod_ms))
        alg_bytes = ALG_BYTES_PER_SAMPLE * n_samples
        achieved = alg_bytes / (k1_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        k1_full = dec.k1_kernel()                      # what the library launches for this chip length, as rocprofv3 prints it
        k1_name = k1_full.replace("void amr::", "").split("<")[0] + f"<{chip}>"
        tf = os.path.join(ROOT, "profiles", "k1_hbm_traffic.json")
        if wl["name"] == "cfg2" and n_blocks == GIB // bs2 and os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                traffic = tj.get("bytes_per_launch")
                traffic_src = f"profiles/k1_hbm_traffic.json (static: rocprofv3 --pmc passes of this command, {tj.get('tag', 'see profiles/README.md')}); not measured in this run"
                # the committed figure belongs to ONE kernel: a K1 that has changed since (another template configuration,
                # another generation) must not go on quoting it
                norm = lambda t: "".join(str(t).split())
                if norm(tj.get("kernel")) != norm(k1_full):
                    # (not an error of the run: the stale figure is simply not quoted; a full line measures its own below)
                    check["traffic"] = f"profiles/k1_hbm_traffic.json is stale (measured on {tj.get('kernel')!r}, the library launches {k1_full!r}): not quoted"
                    traffic, traffic_src = None, None
            except Exception:
                traffic = None
        # (never from inside a profiler: a nested rocprofv3 inherits the outer one's tool libraries)
        profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
        full_line = world == 1 and not args.no_cpu_baseline and not distributed and not profiled
        if (args.measure_traffic or full_line) and not args.no_measure_traffic and world == 1:
            t_now, how = measure_traffic(args.workload, args.blocks, k1_full)
            if t_now is not None:
                traffic, traffic_src = t_now, how
                check.pop("traffic", None)             # (a stale committed figure no longer matters)
            else:
                traffic_src = (traffic_src or "none") + f" ({how})"
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "IQ Msamples/s through Decoder.Decode (SCM, 72 sym/len)" if wl["name"] == "cfg2" else
                      f"IQ Msamples/s through Decoder.Decode ({'+'.join(wl['protos'])}, {chip} sym/len)",
            "value": round(total_samples / dt / 1e6, 1),
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "steady_ms_per_step": round(steady_ms, 4),
            "pipeline_fill_ms": round(dt * 1e3 - args.steps * steady_ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if wl["data"] != "uniform" else "synthetic (uniform random bytes)",
            "config": {"workload": f"{wl['name']}: {'+'.join(wl['protos'])} chip {chip}, {n_blocks} blocks of {bs2} B per GPU",
                       "protocols": wl["protos"], "chip_length": chip,
                       "block_size": bs, "bytes_per_gpu_per_step": nbytes, "planted_packets_per_gpu": len(pk),
                       "hits_per_step_rank0": n_hits, "hits_searched_per_step_rank0": n_searched,
                       "gpu_validation": bool(validating), "deferral": deferral,
                       "spin_up": f"{spin_steps} untimed steps (~{args.spinup_ms:.0f} ms) before the warm-up: shader clock ramp; 64 more untimed steps on either side of warm-up + timed region while the driver's sysfs files are read (device.before / after)",
                       "checks": check,
                       "iq_buffer": "first device allocation", "parallelism": f"block-range shards x{world}",
                       "hit_gather": gather_kind},
            "roofline": {"bound": "hbm", "kernel": k1_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         # the whole path, not only its dominant kernel: algorithmic bytes over the steady step / the timed step
                         "whole_path_frac": round(alg_bytes / (steady_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "whole_path_frac_timed": round(alg_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "k1_ms": round(k1_ms, 4), "search_ms": round(float(np.mean(search_ms)), 4),
                         "kernel_timing": (f"HIP events on the dispatches of every {args.k1_events}th timed step ({len(demod_ms)} steps): "
                                           "K1 duration; search_ms = K2 duration" + (f", from the {len(search_ms)} warm-up steps" if args.k1_level < 2 else "") +
                                           " (with batches in flight K3.. of a batch run on a "
                                           "second stream, let in when the following batch's K1 has all its waves on the chip: they "
                                           "fill its ragged end and the start of that batch's K2; steady_ms_per_step - k1_ms is "
                                           "everything a step costs besides K1)"),
                         "algorithmic_bytes_per_launch": alg_bytes},
            # the box this line was measured on (the pool's boxes differ by +-7 % with one binary: profiles/r05/fresh_runs*.txt)
            "device": {"library": dec.describe(), "compute_units": n_cus_of(dec.describe()),
                       "xcds": (n_cus_of(dec.describe()) or 0) // 32 or None,
                       "before_timed_region": state_before, "after_timed_region": state_after},
        }
        if distributed:
            # what amr_gather_hits put on the wire for the last step: small slots (validated hits) whole, large ones as a
            # 128-byte header + the records sized by their count
            two_phase = dec.gather_two_phase(gatherer.cap)
            sent = 128 + dec.gather_wire_bytes(n_hits) if two_phase else int(gatherer.slot_bytes)
            ranks = dec.comm_ranks()
            if not (ranks == world == args.gpus or os.environ.get("AMR_BENCH_FORCE_DIST") == "1"):
                check["ranks"] = f"MISMATCH: --gpus {args.gpus}, WORLD_SIZE {world}, RCCL communicator spans {ranks}"
                rc = rc or 7
            out["config"].update({"rccl_ranks": ranks,
                                  "gather_bytes_per_step": {"sent_per_rank": sent, "payload_per_rank": 128 + 12 * n_hits,
                                                            "slot_capacity_bytes": int(gatherer.slot_bytes),
                                                            "into_root": sent * world,
                                                            "rule": ("128-byte header + 12 B per record rounded up to 4 KiB (amr_gather_wire_bytes): "
                                                                     "two messages, the root waits for the headers" if two_phase else
                                                                     "slot of up to 256 KiB: one message per rank, nobody waits")},
                                  "gather_records": "validated hits (K5)" if validating else "raw hits (--gather raw)"})
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, dec, d_iq.value, bs2)
            # the other end of the scale: one block per call on the GPU, next to the CPU port's time per block
            nlat = min(300, n_blocks - 50)
            out["latency_us_per_block"] = round(single_block_latency(ra, wl, local_rank, d_iq.value, bs, bs2, n=nlat), 1) if nlat > 0 else None
            out["cpu_baseline"]["single_thread_us_per_block"] = round(bs / out["cpu_baseline"]["single_thread"], 2)
    else:
        out = None
    if distributed:
        rcs = torch.tensor([rc], dtype=torch.int32, device=dev)
        dist.all_reduce(rcs, op=dist.ReduceOp.MAX)
        rc = int(rcs.item())
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        # RCCL writes its version banner to the C stdout buffer; push that out first so that the JSON line is the
        # last line this job prints
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if rc:
        print(f"bench.py: verification failed (rc {rc}): {check}", file=sys.stderr)
    return rc


if __name__ == "__main__":
    raise SystemExit(main())
