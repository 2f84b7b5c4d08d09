"""oracle/_ref -- the reference's own Go sources translated mechanically to C++ (oracle/go2cxx, `make -C oracle _ref`) --
against oracle/decode_oracle.c, the hand restatement every HIP parity test is held to.

What runs on the `_ref` side is code derived from the TEXT of /root/reference/protocol/decode.go, parse.go,
scm/scm.go, scmplus/scmplus.go, idm/idm.go, netidm/netidm.go, r900/r900.go, r900/gf/gf.go and crc/crc.go by a
translator that knows Go syntax only (tests/test_go2cxx.py holds it to the language specification); the caller
(oracle/go2cxx/ref_harness.cpp) uses the public API the way main.go does.  So these tests replace "three people read
decode.go the same way" with "the restatement equals what the source text computes", on

  * every committed golden: SURVEY 8c hashes of assets/sample.bin, tests/golden/sample_bin.json, synth.json,
    r900_filter.json, and the reduced-size keys of bench_golden.json;
  * more than 1 000 random streams over all 31 protocol sets and all 10 legal chip lengths: quantized bits, hit lists
    from the TRANSLATED literal Search (prefilter + exact pass), packet bytes incl. the never-cleared bits of
    Decoder.Slice, Data.Bits, the exported Signal / Quantized buffers, r900's whole filter() output;
  * the messages the translated parsers emit, against the Python mirrors the GPU tests and bench.py use.

It is still not the Go toolchain (oracle/go2cxx/README.md: what the translation guarantees and what it does not), which
is why DESIGN.md words the claim as "pinned to a mechanical translation of the reference source".

The library is built where /root/reference exists and travels as a prebuilt file; without either, the tests skip."""
import hashlib
import itertools
import json
import os

import numpy as np
import pytest

from oracle import ref
from oracle.oracle import PROTOCOLS, OracleDecoder, R900Filter
from tests import util

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libref.so not built and no reference tree to build it from")

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
NAMES = ["scm", "scm+", "idm", "netidm", "r900"]
CHIPS = [8, 32, 40, 48, 56, 64, 72, 80, 88, 96]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(name):
    return json.load(open(os.path.join(G, name)))


def both(protos, chip, iq, hits_cap=1 << 18):
    """-> (oracle (q, hits, bytes), translated reference (q, hits, bytes, messages)), same shapes and order."""
    o = OracleDecoder(list(protos), chip)
    r = ref.RefDecoder(list(protos), chip)
    return o, r, o.decode_stream(iq, hits_cap=hits_cap), r.decode_stream(iq, hits_cap=hits_cap)


def assert_equal_results(o_res, r_res, what=""):
    oq, oh, ob = o_res
    rq, rh, rb, _ = r_res
    assert np.array_equal(oq, rq), f"{what}: quantized bits differ"
    assert oh.shape == rh.shape and np.array_equal(oh, rh), f"{what}: hit lists differ ({len(oh)} vs {len(rh)})"
    assert np.array_equal(ob, rb), f"{what}: packet bytes differ"


# ---------------------------------------------------------------------------------------------------- setup-time rows
def test_lut_geometry_next_power_of_2_crc():
    """a2-a4: NewMagLUT (decode.go:209-216) bit for bit incl. the SURVEY sha256; Allocate's geometry (decode.go:131-160)
    for every protocol set x chip length; NextPowerOf2 (decode.go:377-379) on every value Allocate can feed it; the CRC
    table code (crc/crc.go:34-55) on random messages."""
    lut = ref.mag_lut()
    assert sha(lut.astype("<f4")) == "43571608069d3339e99f1f5c727336b46b81c68431785a43ecb477afaa6c5a73"
    assert np.array_equal(lut.view(np.uint32), OracleDecoder(["scm"], 72).lut.view(np.uint32))
    n = 0
    for k in range(1, len(NAMES) + 1):
        for protos in itertools.combinations(NAMES, k):
            for chip in CHIPS:
                g = ref.RefDecoder(list(protos), chip).geom
                og = OracleDecoder(list(protos), chip).geom
                assert [g[f] for f in ref.GEOM_FIELDS[:12]] == [og.data_rate, og.chip_length, og.symbol_length, og.sample_rate,
                                                                og.preamble_symbols, og.packet_symbols, og.preamble_length,
                                                                og.packet_length, og.block_size, og.block_size2,
                                                                og.buffer_length, og.n_preambles], (protos, chip)
                assert g["signal_len"] == og.block_size + og.symbol_length and g["quantized_len"] == og.buffer_length
                n += 1
    assert n == 31 * 10
    from oracle.oracle import next_power_of_2
    for v in list(range(1, 5000)) + [2 ** k + d for k in range(12, 30) for d in (-1, 0, 1)]:
        assert ref.next_power_of_2(v) == next_power_of_2(v), v
    from oracle import validate_oracle as vo
    from rtlamr_amd.contrib.parsers.crc import CRC
    rng = np.random.default_rng(1)
    for init, poly in ((0, 0x6F63), (0xFFFF, 0x1021)):
        for _ in range(200):
            msg = rng.integers(0, 256, int(rng.integers(1, 100)), dtype=np.uint8).tobytes()
            want = ref.crc16(init, poly, msg)
            assert want == vo.checksum(init, poly, msg) == CRC("x", init, poly, 0).Checksum(msg)


# ---------------------------------------------------------------------------------------------------- committed goldens
def test_survey_and_sample_bin_goldens():
    """SURVEY.md 8c / tests/golden/sample_bin.json: the reference's only fixture through the translated Decoder."""
    raw = util.load_capture()
    fx = load("sample_bin.json")
    assert sha(raw) == fx["file_sha256"]
    for c in fx["cases"]:
        r = ref.RefDecoder(c["protocols"], c["chip"])
        assert r.geom["block_size"] == c["block_size"]
        q, hits, hb, _ = r.decode_stream(raw[: c["blocks"] * r.geom["block_size2"]])
        assert sha(q) == c["qsha"] and int(np.unpackbits(q).sum()) == c["ones"], c["name"]
        assert hits[:, [0, 2]].tolist() == c["hits"], c["name"]
        assert sha(hb) == c["pkt_sha"], c["name"]
    # the 36 chip-80 hits SURVEY 8c lists come out of the TRANSLATED literal Search
    by = {c["name"]: c for c in fx["cases"]}
    assert len(by["whole_scm80"]["hits"]) == 36


def test_synth_goldens():
    for c in load("synth.json")["cases"]:
        o = OracleDecoder(c["protocols"], c["chip"])
        iq, _ = util.synth_stream(c["protocols"], c["chip"], c["blocks"], o.geom.block_size, c["seed"], c["packets"])
        assert sha(iq) == c["iq_sha"]
        r = ref.RefDecoder(c["protocols"], c["chip"])
        q, hits, hb, msgs = r.decode_stream(iq)
        order = np.lexsort((hits[:, 2], hits[:, 0], hits[:, 1]))
        h = np.stack([hits[:, 1], hits[:, 0], hits[:, 2]], axis=1).astype(np.int64)[order]
        assert sha(q) == c["qsha"], c["name"]
        assert len(h) == c["n_hits"] and sha(h.astype("<i8")) == c["hits_sha"], c["name"]
        assert sha(hb[order][:, : r.geom["packet_symbols"] // 8]) == c["pkt_sha"], c["name"]
        assert len(msgs) > 0, "planted CRC-valid packets must come out of the translated parsers as messages"


def test_r900_filter_goldens():
    """tests/golden/r900_filter.json: p.quantized of r900.Parser after EVERY call (r900.go:82-150, 160-172)."""
    raw = util.load_capture()
    synth = {c["name"]: c for c in load("synth.json")["cases"]}
    for c in load("r900_filter.json")["cases"]:
        r = ref.RefDecoder(c["protocols"], c["chip"])
        if c["input"] == "capture":
            iq = raw
        else:
            s = synth[c["input"]] if c["input"] in synth else next(x for x in synth.values() if x["name"] in c["name"])
            iq, _ = util.synth_stream(s["protocols"], s["chip"], s["blocks"], r.geom["block_size"], s["seed"], s["packets"])
        bs2 = r.geom["block_size2"]
        h = hashlib.sha256()
        hist = np.zeros(6, np.int64)
        n = iq.size // bs2
        for k in range(n):
            r.decode(iq[k * bs2:(k + 1) * bs2])
            q = r.r900_quantized()
            h.update(q.tobytes())
            hist += np.bincount(q, minlength=6)[:6]
        assert (h.hexdigest(), hist.tolist(), n) == (c["qsha"], c["hist"], c["calls"]), c["name"]


@pytest.mark.parametrize("key", ["cfg2|blocks=4096|shard=0", "cfg2|blocks=4096|shard=1", "cfg3|blocks=2048|shard=0",
                                 "cfg4:8|blocks=4096|shard=1", "cfg5|blocks=2048|shard=2"])
def test_bench_goldens_reduced_size(key):
    """The reduced-size entries of tests/golden/bench_golden.json (what bench.py --blocks N checks itself against),
    recomputed with the translated reference as ONE Decoder instead of the oracle: "first" and "steady" digests."""
    import bench
    from oracle import oracle as orc
    from tests.golden import make_bench_golden as mk
    gold = load("bench_golden.json")[key]
    spec, nb, shard = key.split("|")
    n_blocks, shard = int(nb.split("=")[1]), int(shard.split("=")[1])
    wl = bench.workload(spec)
    probe = OracleDecoder(wl["protos"], wl["chip"])
    g = probe.geom
    bs, bs2 = g.block_size, g.block_size2
    n_samples = n_blocks * bs
    pk = bench.build_packets(wl, shard, bs, n_samples)
    iq = orc.synth_stream(n_samples, 1, shard * n_samples, pk, wl["chip"], 4)
    hb = (g.packet_length + bs - 1) // bs + 2
    for state in ("first", "steady"):
        if state == "steady":
            head = iq[-hb * bs2:]
        elif shard > 0:
            prev = bench.build_packets(wl, shard - 1, bs, n_samples)[-8:]
            head = orc.synth_stream(hb * bs, 1, shard * n_samples - hb * bs, prev + pk[:1], wl["chip"], 4)
        else:
            head = iq[:0]
        nh = head.size // bs2
        r = ref.RefDecoder(wl["protos"], wl["chip"])
        if nh:
            r.decode_stream(head)
        q, hits, pb, _ = r.decode_stream(iq, hits_cap=1 << 20)
        order = np.lexsort((hits[:, 2], hits[:, 0], hits[:, 1]))
        rows = np.stack([hits[:, 1], hits[:, 0], hits[:, 2]], axis=1).astype(np.int64)[order]
        rows[:, 1] += shard * n_blocks
        got = mk._digest(wl, g, rows, pb[order], q)
        assert got == gold[state], (key, state)


# ---------------------------------------------------------------------------------------------------- random streams
def random_case(rng):
    k = int(rng.integers(1, 6))
    protos = [NAMES[i] for i in sorted(rng.choice(5, k, replace=False))]
    rng.shuffle(protos)
    chip = int(rng.choice(CHIPS, p=[.28, .12, .1, .1, .08, .08, .08, .06, .05, .05]))
    g = OracleDecoder(protos, chip).geom
    npk = int(rng.integers(1, 4))
    longest = max([util.PKT_BUILDERS[p][1] for p in protos if p in util.PKT_BUILDERS] + [0]) * 2 * chip
    need = max((2 * g.packet_length) // g.block_size + 4, npk * (longest // g.block_size + 2) + 1)
    n_blocks = int(rng.integers(need, need + 12))
    iq, _ = util.synth_stream(protos, chip, n_blocks, g.block_size, seed=int(rng.integers(1 << 30)), n_packets=npk,
                              edge_every=int(rng.integers(1, 4)))
    if rng.integers(3) == 0:                      # a stretch of uniform random bytes: dense candidates for the prefilter
        a = int(rng.integers(0, iq.size // 2))
        iq[a:a + 30_000] = rng.integers(0, 256, min(30_000, iq.size - a), dtype=np.uint8)
    return protos, chip, iq


@pytest.mark.parametrize("chunk", range(8))
def test_random_streams(chunk):
    """8 x 130 = 1 040 random streams: protocol set (any of the 31, in any registration order), chip length, length,
    planted CRC-valid packets of the set's protocols (block-edge straddlers included), optional uniform stretch."""
    rng = np.random.default_rng(9000 + chunk)
    seen_sets, total_hits = set(), 0
    for i in range(130):
        protos, chip, iq = random_case(rng)
        o, r, o_res, r_res = both(protos, chip, iq)
        assert_equal_results(o_res, r_res, f"chunk {chunk} case {i} {protos} chip {chip}")
        # the exported buffers after the last call (decode.go:46-50)
        assert np.array_equal(o.signal.view(np.uint32), r.signal.view(np.uint32))
        assert np.array_equal(o.quantized, r.quantized)
        seen_sets.add(tuple(sorted(protos)))
        total_hits += len(o_res[1])
    assert total_hits > 1000 and len(seen_sets) >= 25


def test_stale_bits_of_slice_come_from_the_translated_source():
    """PacketSymbols % 8 != 0 (r900 alone / with scm): Decoder.Slice never clears d.pkt (decode.go:363-366), so the last
    byte of a packet carries bits of earlier hits.  The translated Slice must show the effect, and the oracle -- and
    with it the GPU's k_stale_bits -- must reproduce every one of those bytes."""
    from rtlamr_amd import synth
    from rtlamr_amd.contrib.parsers import r900 as pr900
    for protos, chip in ((["r900"], 8), (["scm", "r900"], 32), (["r900", "scm"], 72)):
        g = OracleDecoder(protos, chip).geom
        assert g.packet_symbols % 8 == 4
        n_blocks = max(60, (4 * g.packet_length) // g.block_size + 8)
        iq, _ = util.synth_stream(["scm"], chip, n_blocks, g.block_size, seed=3 + chip, n_packets=3, edge_every=2)
        for j, mid in enumerate((77, 4242)):        # r900 bursts: hits of the r900 preamble itself
            chips = synth.r900_chips(PROTOCOLS["r900"][0], pr900.build_r900_symbols(mid, consumption=mid))
            synth.plant_chips(iq, (5 + 20 * j) * g.block_size + 17, chips, chip, 34, -29)
        o, r, o_res, r_res = both(protos, chip, iq)
        assert_equal_results(o_res, r_res, str(protos))
        assert len(r_res[1]) >= 10
        assert (r_res[2][:, -1] & 0xF0).any(), "no stale bit anywhere: the test tests nothing"


def test_data_bits_and_per_call_form():
    """One Decode call at a time like main.go:235: hit lists per preamble, Data.Bits (parse.go:61-69), Signal."""
    protos, chip = ["scm", "idm"], 72
    o = OracleDecoder(protos, chip)
    r = ref.RefDecoder(protos, chip)
    g = o.geom
    iq, _ = util.synth_stream(protos, chip, 40, g.block_size, seed=11, n_packets=3)
    n_hits = 0
    for k in range(40):
        blk = iq[k * g.block_size2:(k + 1) * g.block_size2]
        ores = o.decode(blk)
        rres, _ = r.decode(blk)
        for pid in range(g.n_preambles):
            assert r.preambles[pid] == PROTOCOLS[protos[pid]][0]
            assert np.array_equal(ores[pid][0], rres[pid][0]) and np.array_equal(ores[pid][1], rres[pid][1])
            for j in range(min(3, len(rres[pid][0]))):
                assert r.hit_bits(pid, j) == "".join(f"{b:08b}" for b in rres[pid][1][j])
            n_hits += len(rres[pid][0])
        assert np.array_equal(o.signal.view(np.uint32), r.signal.view(np.uint32))
    assert n_hits > 100


def test_r900_filter_every_call_and_signal_history():
    """r900.Parser.Parse's buffer handling (r900.go:160-170) and filter() (r900.go:82-150) on planted r900 bursts +
    noise, every call, every position: the literal C restatement (oracle.R900Filter) against the translated source; and
    the R900 messages of the translated parser (digits, base-6 symbols, Reed-Solomon syndromes: r900.go:174-246,
    gf/gf.go) against the Python mirror fed with the numpy oracle's digits (what the GPU's K4 is checked with)."""
    from oracle import r900_oracle
    from rtlamr_amd import synth
    from rtlamr_amd.contrib.parsers import gf
    from rtlamr_amd.contrib.parsers import r900 as pr900
    field = gf.Field(32, 37, 2)
    mids = (11111, 22222222, 3333333333)
    for protos, chip in ((["r900"], 72), (["scm", "r900"], 8), (["r900", "idm"], 32)):
        o = OracleDecoder(protos, chip)
        r = ref.RefDecoder(protos, chip)
        g = o.geom
        burst = (64 + 168) * chip
        n_blocks = (10 * burst + 2 * g.packet_length) // g.block_size + 4
        iq = synth.noise(n_blocks * g.block_size, 9 + chip)
        for j, mid in enumerate(mids):
            chips = synth.r900_chips(PROTOCOLS["r900"][0], pr900.build_r900_symbols(mid, consumption=mid & 0xFFFFFF))
            synth.plant_chips(iq, burst // 2 + 3 * j * burst + 11 * j, chips, chip, 34, -29)
        flt = R900Filter(o)
        got = []
        for k in range(n_blocks):
            blk = iq[k * g.block_size2:(k + 1) * g.block_size2]
            o.decode(blk)
            _, msgs = r.decode(blk)
            want = flt.step()
            assert np.array_equal(r.r900_quantized(), want), (protos, chip, k)
            assert np.array_equal(r.r900_signal().view(np.uint32), flt.signal.view(np.uint32))
            got += [(k, m[2], tuple(m[5])) for m in msgs if m[1] == "R900"]
        hits, digits = r900_oracle.digits_for_stream(protos, chip, iq)
        want_msgs, seen, last = [], set(), -1
        for (k, _), d in zip(hits.tolist(), digits):
            if k != last:
                seen, last = set(), k              # `seen` lives for one Parse call (r900.go:176)
            m = pr900.parse_digits(field, d, seen)
            if m is not None:
                want_msgs.append((k, m.MeterID(), tuple(m.Record())))
        assert got == want_msgs, (protos, chip)
        assert {m[1] for m in got} == set(mids), "the translated r900 parser must recover the planted messages"


def test_messages_of_translated_parsers_equal_the_python_mirrors():
    """a13 + the parsers behind it: what the translated scm / scm+ / idm / netidm parsers send on the message channel
    (MsgType, MeterID, MeterType, Checksum, Record()) against rtlamr_amd/contrib/parsers run over the ORACLE's hit lists --
    the mirrors every GPU test and every bench.py run use to recover their planted packets."""
    import rtlamr_amd as ra
    for protos, chip in ((["scm"], 72), (["scm+", "scm"], 32), (["idm", "netidm"], 72), (["scm", "scm+", "idm", "r900"], 8)):
        o = OracleDecoder(protos, chip)
        r = ref.RefDecoder(protos, chip)
        g = o.geom
        longest = max(util.PKT_BUILDERS[p][1] for p in protos if p in util.PKT_BUILDERS) * 2 * chip
        n_blocks = max(60, 6 * (longest // g.block_size + 2) + 2)
        iq, _ = util.synth_stream(protos, chip, n_blocks, g.block_size, seed=5, n_packets=6)
        parsers = [ra.new_parser(p, chip) for p in protos]
        n_msgs = 0
        for k in range(n_blocks):
            blk = iq[k * g.block_size2:(k + 1) * g.block_size2]
            ores = o.decode(blk)
            _, msgs = r.decode(blk)
            got = sorted((m[1], m[2], m[3], m[4], tuple(m[5])) for m in msgs)
            want = []
            for name, p, pid in zip(protos, parsers, o.preamble_ids):
                if name == "r900":
                    continue
                data = []
                for idx, pb in zip(*ores[pid]):
                    d = ra.new_data(bytes(pb))
                    d.Idx = int(idx)
                    data.append(d)
                for m in p.Parse(data):
                    want.append((m.MsgType(), m.MeterID(), m.MeterType(), bytes(m.Checksum()).hex(), tuple(m.Record())))
            assert got == sorted(want), (protos, chip, k)
            n_msgs += len(got)
        assert n_msgs >= 6


def test_r900bcd_wrapper_parser():
    """r900bcd/r900bcd.go: a parser that embeds the protocol.Parser interface value of r900.NewParser, runs it on a channel
    of its own in two goroutines, asserts its messages back to r900.R900 and re-reads Consumption's hex digits as decimal
    (a hex letter: ParseUint fails and Go keeps 0).  Translated source vs the Python mirror fed with the numpy oracle's digits."""
    from oracle import r900_oracle
    from rtlamr_amd import synth
    from rtlamr_amd.contrib.parsers import gf
    from rtlamr_amd.contrib.parsers import r900 as pr900
    import rtlamr_amd as ra
    chip = 72
    r = ref.RefDecoder(["r900bcd"], chip)
    g = OracleDecoder(["r900"], chip).geom
    assert r.geom["packet_symbols"] == 116 and r.preambles == [PROTOCOLS["r900"][0]]      # it reports r900's configuration
    burst = (64 + 168) * chip
    cons = (0x123456, 0x2b67, 0x000999)                                                  # decimal digits / a hex letter / leading zeros
    n_blocks = (10 * burst + 2 * g.packet_length) // g.block_size + 4
    iq = synth.noise(n_blocks * g.block_size, 77)
    for j, c in enumerate(cons):
        chips = synth.r900_chips(PROTOCOLS["r900"][0], pr900.build_r900_symbols(9000 + j, consumption=c))
        synth.plant_chips(iq, burst // 2 + 3 * j * burst + 11 * j, chips, chip, 34, -29)
    _, _, _, msgs = r.decode_stream(iq)
    got = [(m[0], m[1], m[2], tuple(m[5])) for m in msgs]
    hits, digits = r900_oracle.digits_for_stream(["r900"], chip, iq)
    p = ra.new_parser("r900bcd", chip)
    want, last, batch = [], -1, []

    def flush(k, batch):
        for m in p.Parse(batch):
            want.append((k, m.MsgType(), m.MeterID(), tuple(m.Record())))
    for (k, _), d in zip(hits.tolist(), digits):
        if k != last and batch:
            flush(last, batch)
            batch = []
        last = k
        pkt = ra.new_data(bytes(15))
        pkt.Digits = d
        batch.append(pkt)
    if batch:
        flush(last, batch)
    assert got == want and {m[1] for m in got} == {"R900BCD"}
    by_id = {m[2]: m[3][4] for m in got}                                                 # Record()[4] = Consumption
    assert by_id == {9000: "123456", 9001: "0", 9002: "999"}
