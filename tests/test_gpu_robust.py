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


@pytest.mark.parametrize("env", [{"AMR_GATE_END": "1"}, {"AMR_GATE_END": "-1"}, {"AMR_K3_LDS_KB": "40", "AMR_K2W_LDS_KB": "40"}, {"AMR_K3_PRIO": "3"}],
                         ids=["tail-behind-k1-end", "tail-behind-k1-end-auto", "lds-padding", "k3-priority"])
@pytest.mark.parametrize("protos,chip", [(["scm"], 40), (["scm"], 8), (["scm", "scm+", "idm", "r900"], 72)], ids=["scm-40", "scm-8", "all"])
def test_round6_ab_hooks_change_no_result(protos, chip, env):
    """The A/B hooks of round 6 (amr_host.h: the tail behind the END of a one-launch K1 instead of behind the gate kernel; LDS
    padding of the multi-preamble search and of K3; K3's wave priority) are scheduling choices: every one of them must leave
    hit lists and packet bytes alone.  Chip 8 and 40 also run the all-XCD announcement of K1's last eight workgroups and
    (chip 8) the halo-shift kernel through the pipelined path."""
    per, n = 256, 6
    want = oracle_digest(protos, chip, per, n)
    assert want["n_hits"] > 0
    got = probe(protos, chip, per, n, env=dict(env, AMR_K1_COOP_MAX="0"))
    assert {k: got[k] for k in want} == want, f"{env}: other hits than the oracle's"
    assert "gate-timeouts" not in got["describe"]
    assert got["seconds"] < 2.0


@pytest.mark.parametrize("env", [{}, {"AMR_K1_ROUND_TILES": "1"}, {"AMD_SERIALIZE_KERNEL": "3"}, {"AMR_HIT_CAP": "64"}],
                         ids=["plain", "a-launch-per-wave-tile", "serialize-kernel", "capacity-growth"])
@pytest.mark.parametrize("protos,per,n", [(["scm"], 256, 7), (["scm+"], 256, 7), (["scm"], 128, 9), (["scm"], 100, 9)],
                         ids=["scm", "scm+-rows-of-8-words", "scm-128", "scm-ragged-batches"])
def test_in_wave_search_at_chip_8_equals_oracle(protos, per, n, env):
    """Rows of 16 words (chip length 8, one preamble): the K1 wave searches its own tile before it stores it (k1_search.h), the
    search launch is the history tile + the rows 63 (k2_row_cleanup).  Hits across tile and batch boundaries (row 63's last
    words, the history rows), with batches that are no multiple of 64 blocks (the remainder runs the one-wave-per-block K1 and
    the ordinary search), with every wave-tile a launch of its own, with a result that outgrows its buffers (re-search by the
    ordinary kernel) -- against the oracle's literal Search; and AMR_INWAVE=0 must give the same."""
    chip = 8
    want = oracle_digest(protos, chip, per, n)
    assert want["n_hits"] > 0
    on = probe(protos, chip, per, n, env=dict(env, AMR_K1_COOP_MAX="0"))
    off = probe(protos, chip, per, n, env=dict(env, AMR_K1_COOP_MAX="0", AMR_INWAVE="0"))
    assert {k: on[k] for k in want} == want, "in-wave search: other hits than the oracle's"
    assert {k: off[k] for k in want} == want
    rows16 = protos == ["scm"]            # scm+ alone at chip 8 has rows of 8 words: the fallback search, never in-wave
    assert ("in-wave-searches" in on["describe"]) == (rows16 and per % 64 == 0), on["describe"]
    assert "in-wave-searches" not in off["describe"]


@pytest.mark.parametrize("chip", [32, 40])
def test_in_wave_search_at_rows_of_64_words_equals_oracle(chip):
    """AMR_INWAVE=2 (an experiment that lost on speed, profiles/r06/inwave/: off by default): at chip 32 / 40 a row of 64 words
    is exactly the burst the last tile boundary completes; K1 holds it back, searches it and stores it.  Same hits as the
    oracle's, and the path must really have run."""
    protos, per, n = ["scm"], 256, 7
    want = oracle_digest(protos, chip, per, n)
    assert want["n_hits"] > 0
    got = probe(protos, chip, per, n, env={"AMR_K1_COOP_MAX": "0", "AMR_INWAVE": "2"})
    assert {k: got[k] for k in want} == want
    assert "in-wave-searches" in got["describe"]


@pytest.mark.gpu
@pytest.mark.parametrize("protos,depth", [(["r900"], 3), (["scm", "r900"], 3), (["r900"], 2)])
def test_pipelined_r900_searches_every_batch_once(protos, depth):
    """ADVICE r05 (amr_pipeline.hip collect): with PacketSymbols % 8 != 0 a re-search of one batch (a fresh r900 stream
    overflows its first batch's sparse list: the zero history matches the preamble's 16 leading zeros) used to mark every
    other batch in flight for a full second search, which marked the next ones, for the rest of the stream.  Now a younger
    batch whose tail had already run gets k_stale_bits alone once more, and hands the duty on only when its carry byte
    changed.  Every byte equals the oracle's, and the counters of amr_describe stay small over 40 pipelined batches."""
    import ctypes as C
    import re
    from rtlamr_amd import _lib, synth
    from rtlamr_amd.contrib.parsers import r900 as pr900
    from oracle.oracle import PROTOCOLS
    L = _lib.lib()
    chip = 72
    dec = util.make_decoder(protos, chip)
    bufs = []
    try:
        bs, bs2 = dec.Cfg.BlockSize, dec.Cfg.BlockSize2
        n_batches, nb = 40, 64
        iq, _ = util.synth_stream(["scm"], chip, n_batches * nb, bs, seed=77, n_packets=30, edge_every=3)
        for j in range(12):            # r900 bursts all along the stream: hits (and carry bytes) in most batches
            chips = synth.r900_chips(PROTOCOLS["r900"][0], pr900.build_r900_symbols(1000 + j, consumption=j))
            synth.plant_chips(iq, (3 + 200 * j) * bs + 31 * j, chips, chip, 34, -29)
        want = util.oracle_run(protos, chip, iq)
        got, inflight = [], 0
        for k in range(n_batches):
            part = np.ascontiguousarray(iq[k * nb * bs2:(k + 1) * nb * bs2])
            d = C.c_void_p()
            _lib.check(L.amr_dev_alloc(0, part.size, C.byref(d)), "alloc")
            _lib.check(L.amr_dev_upload(0, d, part.ctypes.data, part.size), "upload")
            bufs.append(d)
            dec.submit_device(d.value, nb)
            inflight += 1
            if inflight == depth:
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
        order = np.lexsort((h[:, 2], h[:, 1], h[:, 0]))
        assert np.array_equal(h[order], want[2]) and len(h) > 200
        assert np.array_equal(p[order], want[3]), "packet bytes (stale high bits included) differ from the oracle's"
        m = re.search(r"re-searches (\d+) stale-reruns (\d+)", dec.describe())
        researches, stale = (int(m.group(1)), int(m.group(2))) if m else (0, 0)
        assert researches <= 3, f"{researches} of {n_batches} pipelined batches were searched twice"
        assert stale <= 2 * researches + 2, f"{stale} stale-bit passes re-run for {researches} re-searches"
    finally:
        dec.close()
        for d in bufs:
            L.amr_dev_free(0, d)


@pytest.mark.gpu
@pytest.mark.parametrize("protos,chip", [(["r900"], 72), (["scm", "r900"], 32)])
def test_sharded_run_equals_single_decoder_every_byte(protos, chip):
    """the last asterisk on "bit-exact bytes".  A block-range shard primed with amr_prime cannot know the
    byte Decoder.Slice's never-cleared d.pkt carries into its first hit (decode.go:363-366, PacketSymbols % 8 = 4 here).
    (a) shards run one after another hand it on with amr_get_stale_carry / amr_set_stale_carry; (b) shards run at the
    same time start from zero and are patched afterwards (dist.patch_stale_carry).  Both: every byte of every packet
    equals the single decoder's, i.e. the oracle's."""
    from rtlamr_amd import dist as shard
    from rtlamr_amd import synth
    from rtlamr_amd.contrib.parsers import r900 as pr900
    from oracle.oracle import PROTOCOLS
    probe = util.make_decoder(protos, chip)
    bs, bs2, ps = probe.Cfg.BlockSize, probe.Cfg.BlockSize2, probe.Cfg.PacketSymbols
    pb, hb = probe.prime_blocks(), probe.halo_bytes()
    probe.close()
    n_blocks, world = 256, 4
    iq, _ = util.synth_stream(["scm"], chip, n_blocks, bs, seed=19, n_packets=10, edge_every=2)
    for j in range(6):
        chips = synth.r900_chips(PROTOCOLS["r900"][0], pr900.build_r900_symbols(500 + j, consumption=j))
        synth.plant_chips(iq, (9 + 40 * j) * bs + 13 * j, chips, chip, 34, -29)
    want = util.oracle_run(protos, chip, iq)
    assert (want[3][:, -1] & 0xF0).any()

    def run_shard(r, carry_in):
        k0, k1 = shard.shard_range(n_blocks, world, r)
        p0, _ = shard.prime_range(k0, pb)
        dec = util.make_decoder(protos, chip)
        try:
            if k0 > 0:
                lead = iq[p0 * bs2 - hb:p0 * bs2] if p0 * bs2 >= hb else None
                dec.prime(iq[p0 * bs2:k0 * bs2], lead)
            dec.set_block_base(k0)
            if carry_in is not None:
                dec.set_stale_carry(carry_in)
            q, h, p = util.gpu_run(dec, iq[k0 * bs2:k1 * bs2])
            return h, p, dec.stale_carry()
        finally:
            dec.close()

    # (a) sequential hand-over
    hs, ps_, carry = [], [], 0
    for r in range(world):
        h, p, carry = run_shard(r, carry)
        hs.append(h); ps_.append(p)
    h, p = np.concatenate(hs), np.concatenate(ps_)
    order = np.lexsort((h[:, 2], h[:, 1], h[:, 0]))
    assert np.array_equal(h[order], want[2]) and np.array_equal(p[order], want[3])
    # (b) concurrent shards (zero start), patched afterwards in shard order
    hs, ps_, carry, differed = [], [], 0, False
    for r in range(world):
        h, p, own = run_shard(r, None)
        fixed = shard.patch_stale_carry(h, p, ps, carry)
        differed |= not np.array_equal(fixed, p)
        if len(h):                                    # what this shard hands on: its last hit's byte, patched
            last = np.lexsort((h[:, 2], h[:, 0], h[:, 1]))[-1]
            carry = int(fixed[last, -1])
        hs.append(h); ps_.append(fixed)
    h, p = np.concatenate(hs), np.concatenate(ps_)
    order = np.lexsort((h[:, 2], h[:, 1], h[:, 0]))
    assert np.array_equal(h[order], want[2]) and np.array_equal(p[order], want[3])
    assert differed, "no shard boundary carried a stale bit: the test tests nothing"
