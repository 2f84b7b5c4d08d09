"""Pins the CPU oracle (tests-only infrastructure).  The reference has no golden vectors for
protocol/decode.go, so these are the derived vectors of SURVEY.md section 8c (hard-coded below,
produced during the survey by two independent restatements) plus the committed fixtures of
tests/golden/ and the self-validating CRC decodes of the reference's own capture."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import np_oracle as npo
from oracle.oracle import PROTOCOLS, OracleDecoder, next_power_of_2
from tests import util

HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLE = "/root/reference/assets/sample.bin"
need_reference = pytest.mark.skipif(not os.path.exists(SAMPLE), reason="the reference tree only exists in the build container")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# SURVEY.md section 8c (derived golden vectors)
SURVEY_LUT_SHA = "43571608069d3339e99f1f5c727336b46b81c68431785a43ecb477afaa6c5a73"
SURVEY_SAMPLE_SHA = "541668790710efd3610b3fb1f1c6e6c4e6e4c1704590413517d402c7c083860c"
SURVEY_CASES = [
    (["scm"], 72, 524288, 64, 131671, "c5da0773649a70477e2fec8b01285a1694f46eef4fadc6361e07242c495a2261", 0),
    (["scm"], 72, None, 69, 141952, "17ecf593593261d1ad439000caf569fcfe381cae0e42dc3b959ee960f98cfa3b", 0),
    (["idm"], 72, None, 34, 139911, "bb9ed89851a4cca9583bc1031d2d75dec528c90b688132bf1602dbccac4a8ef6", 0),
    (["scm"], 80, None, 69, 141654, "76c3b9e08a007c1203b06dbb94197d82c21b0614dc833e8e564fbbab3125ad0f", 36),
]
SURVEY_CHIP80_HITS = ([(8, i) for i in range(1817, 1821)] + [(13, 683), (23, 612), (23, 613), (27, 3305)] +
                      [(32, i) for i in range(420, 432)] + [(37, i) for i in range(812, 817)] + [(46, 1898)] +
                      [(51, i) for i in range(394, 398)] + [(55, 2406)] + [(55, i) for i in range(2408, 2413)])


def test_mag_lut_matches_survey_vector():
    lut = OracleDecoder(["scm"], 72).lut
    assert sha(lut.astype("<f4")) == SURVEY_LUT_SHA
    assert lut[0] == 1.0 and lut[255] == 1.0
    assert lut[127:129].view(np.uint32).tolist() == [0x37810183, 0x37810183]
    assert np.array_equal(lut, npo.mag_lut())


@pytest.mark.parametrize("chip,protos,exp", [
    (72, ["scm"], dict(SL=144, PreL=3024, PL=13824, BS=4096, BS2=8192, Buf=17920)),
    (72, ["idm"], dict(SL=144, PreL=4608, PL=105984, BS=8192, BS2=16384, Buf=114176)),
    (8, ["scm"], dict(SL=16, PreL=336, PL=1536, BS=512, BS2=1024, Buf=2048)),
    (72, ["scm", "scm+", "idm", "r900"], dict(SL=144, PreL=4608, PL=105984, BS=8192, BS2=16384, Buf=114176)),
    # the smallest geometries a legal command line produces: scm+ alone (16 preamble symbols, scmplus.go:48-57)
    (8, ["scm+"], dict(SL=16, PreL=256, PL=2048, BS=256, BS2=512, Buf=2304)),
    (32, ["scm+"], dict(SL=64, PreL=1024, PL=8192, BS=1024, BS2=2048, Buf=9216)),
])
def test_geometry_table(chip, protos, exp):
    """SURVEY.md section 8 geometry table (Allocate, decode.go:131-141)."""
    g = OracleDecoder(protos, chip).geom
    assert (g.symbol_length, g.preamble_length, g.packet_length, g.block_size, g.block_size2, g.buffer_length) == (
        exp["SL"], exp["PreL"], exp["PL"], exp["BS"], exp["BS2"], exp["Buf"])
    assert g.sample_rate == 32768 * chip
    assert next_power_of_2(exp["PreL"]) == exp["BS"]


@need_reference
def test_capture_fixture_is_the_reference_capture():
    """tests/golden/capture_iq.xz (made by make_golden.py) decompresses to assets/sample.bin byte for byte."""
    assert np.array_equal(util.load_capture(), np.fromfile(SAMPLE, dtype=np.uint8))


@pytest.mark.parametrize("protos,chip,nbytes,calls,ones,qsha,nhits", SURVEY_CASES)
def test_sample_bin_survey_vectors(protos, chip, nbytes, calls, ones, qsha, nhits):
    raw = util.load_capture()
    assert sha(raw) == SURVEY_SAMPLE_SHA
    d = OracleDecoder(protos, chip)
    nb = (nbytes or raw.size) // d.geom.block_size2
    assert nb == calls
    q, hits, hb = d.decode_stream(raw[: nb * d.geom.block_size2])
    assert int(np.unpackbits(q).sum()) == ones
    assert sha(q) == qsha
    assert len(hits) == nhits
    if chip == 80:
        assert [tuple(x) for x in hits[:, [0, 2]].tolist()] == SURVEY_CHIP80_HITS
    # independent numpy restatement agrees bit for bit
    assert np.array_equal(np.packbits(npo.quantize_stream(raw[: nb * d.geom.block_size2], chip, d.geom.block_size)), q)


def test_sample_bin_true_chip_length_decodes_crc_valid_packets():
    """SURVEY 8c self-check: chip 78 semantic search -> 853 hits, 14 distinct CRC-valid SCM packets."""
    raw = util.load_capture()
    d = OracleDecoder(["scm"], 78)
    nb = raw.size // d.geom.block_size2
    q, hits, hb = d.decode_stream(raw[: nb * d.geom.block_size2], mode=1)
    assert sha(q) == "fd3c816bf2dad42c8566115024f5430a76ee83fad93fccd95b36f26f884303bf"
    assert len(hits) == 853
    from rtlamr_amd.contrib.parsers.crc import CRC
    bch = CRC("BCH", 0, 0x6F63, 0)
    valid = {bytes(b).hex() for b in hb if bch.Checksum(bytes(b[2:12])) == 0}
    assert len(valid) == 14
    assert {"f953026101b3360c4105d005", "f953036003b5e30c3a08f6bb", "f95303600c30220ab87c8069"} <= valid
    # literal Search (misaligned prefilter at SL%8 != 0) drops matches, exactly as the survey measured
    d2 = OracleDecoder(["scm"], 78)
    _, hits_lit, _ = d2.decode_stream(raw[: nb * d2.geom.block_size2], mode=0)
    assert [tuple(x) for x in hits_lit[:, [0, 2]].tolist()] == [(13, 376), (23, 304), (23, 305), (23, 306), (46, 1592), (51, 88)]


def test_committed_sample_fixture_is_current():
    fx = json.load(open(os.path.join(HERE, "golden", "sample_bin.json")))
    assert fx["file_sha256"] == SURVEY_SAMPLE_SHA and fx["lut_sha256"] == SURVEY_LUT_SHA
    by = {c["name"]: c for c in fx["cases"]}
    assert by["cfg1_first_512KiB_scm72"]["qsha"] == SURVEY_CASES[0][5]
    assert by["whole_scm80"]["hits"] == [list(x) for x in SURVEY_CHIP80_HITS]


def test_synth_fixtures_oracle_and_numpy_agree():
    """Committed synthetic golden vectors (also used by the -m gpu tests): the C oracle reproduces
    them, its literal and semantic searches agree (legal chip lengths), and numpy agrees on q."""
    fx = json.load(open(os.path.join(HERE, "golden", "synth.json")))
    for c in fx["cases"]:
        d = OracleDecoder(c["protocols"], c["chip"])
        iq, _ = util.synth_stream(c["protocols"], c["chip"], c["blocks"], d.geom.block_size, c["seed"], c["packets"])
        assert sha(iq) == c["iq_sha"], c["name"]
        o, q, h, p = util.oracle_run(c["protocols"], c["chip"], iq)
        assert sha(q) == c["qsha"] and len(h) == c["n_hits"] and sha(h.astype("<i8")) == c["hits_sha"], c["name"]
        _, q2, h2, p2 = util.oracle_run(c["protocols"], c["chip"], iq, mode=1)
        assert np.array_equal(h, h2) and np.array_equal(q, q2)
        assert np.array_equal(np.packbits(npo.quantize_stream(iq, c["chip"], d.geom.block_size)), q)
        assert c["n_hits"] > 0


def test_numpy_search_and_slice_agree_with_c_oracle():
    protos, chip = ["scm"], 72
    d = OracleDecoder(protos, chip)
    iq, _ = util.synth_stream(protos, chip, 40, d.geom.block_size, seed=77, n_packets=4)
    o, q, h, p = util.oracle_run(protos, chip, iq)
    g = npo.geometry(chip, 21, 96)
    qn = npo.quantize_stream(iq, chip, g["BS"])
    hits, qq = npo.search_stream(qn, PROTOCOLS["scm"][0], g)
    assert np.array_equal(hits, h[:, 1:3])
    assert np.array_equal(npo.slice_packets(qq, hits, g, 96), p)


def test_stream_start_zero_history():
    """decode.go:144-145: fresh buffers are zero.  First output bit is f=+0 -> 1; r900's preamble starts
    with 16 zero bits, so hits that reach into the zero history exist and must be reproduced."""
    d = OracleDecoder(["scm"], 72)
    iq = np.full(2 * d.geom.block_size2, 127, np.uint8)
    q, hits, _ = d.decode_stream(iq)
    assert q[0] & 0x80
    d = OracleDecoder(["r900"], 72)
    bs = d.geom.block_size
    from rtlamr_amd import synth
    iq = synth.noise(4 * bs, seed=3)
    # plant the non-zero tail of the r900 preamble so that its 16 leading zeros fall before the stream
    tail = "1110010101100100"
    pk = synth.Packet(start=0, data=int(tail, 2).to_bytes(2, "big"), n_bits=16, d_i=40, d_q=-40)
    synth.plant(iq, [pk], 72)
    _, hits, _ = d.decode_stream(iq)
    pl = d.geom.packet_length
    early = [h for h in hits.tolist() if h[0] * bs + h[2] - pl < 0]
    assert early, "expected hits whose first taps lie in the zero history"


@pytest.mark.parametrize("seed", range(24))
def test_c_and_numpy_restatements_agree_on_random_configurations(seed):
    """The two independent restatements (literal per-call C, stream-global numpy) against each other on random
    protocol/chip/length/byte-distribution draws: quantized bits, literal == semantic search, search and slice."""
    rng = np.random.default_rng(700 + seed)
    name = ["scm", "scm+", "idm", "r900"][int(rng.integers(4))]
    chip = int(rng.choice([8, 32, 40, 48, 56, 64, 72, 80, 88, 96]))
    d = OracleDecoder([name], chip)
    g = d.geom
    n_blocks = int(np.clip(rng.integers(20, 120), (3 * g.packet_length) // g.block_size + 3, 600_000 // g.block_size + 20))
    npk = 0 if name == "r900" else int(rng.integers(1, 4))
    iq, _ = util.synth_stream([name], chip, n_blocks, g.block_size, seed=int(rng.integers(1 << 30)), n_packets=npk)
    if rng.integers(2):
        a = int(rng.integers(0, iq.size // 2))
        iq[a:a + 50_000] = rng.integers(0, 256, min(50_000, iq.size - a), dtype=np.uint8)
    o, q, h, p = util.oracle_run([name], chip, iq)
    _, q2, h2, p2 = util.oracle_run([name], chip, iq, mode=1)
    assert np.array_equal(q, q2) and np.array_equal(h, h2) and np.array_equal(p, p2)
    qn = npo.quantize_stream(iq, chip, g.block_size)
    assert np.array_equal(np.packbits(qn), q)
    pre, pre_sym, pkt_sym = PROTOCOLS[name][0], PROTOCOLS[name][1], PROTOCOLS[name][2]
    gg = npo.geometry(chip, pre_sym, pkt_sym)
    hits, qq = npo.search_stream(qn, pre, gg)
    assert np.array_equal(hits, h[:, 1:3])
    if len(hits):
        nfull = pkt_sym // 8
        assert np.array_equal(npo.slice_packets(qq, hits, gg, pkt_sym)[:, :nfull], p[:, :nfull])


@pytest.mark.parametrize("protos,chip", [(["scm", "r900"], 72), (["r900", "scm"], 8), (["scm", "r900"], 32)])
def test_stale_bits_of_the_last_packet_byte_agree_between_the_restatements(protos, chip):
    """PacketSymbols % 8 != 0 (116: r900 with or without scm): the literal C restatement never clears d.pkt, like
    decode.go:353-375; the numpy restatement slices clean packets and applies the recurrence B(j) = (B(j-1) << r | fresh(j))
    over the hits in slicing order (call, preamble id, idx).  Every byte of every packet must agree -- this is what the GPU's
    k_stale_bits is held to (tests/util.py compares every byte)."""
    d = OracleDecoder(protos, chip)
    g = d.geom
    assert g.packet_symbols % 8 == 4
    n_blocks = max(60, (4 * g.packet_length) // g.block_size + 8)
    iq, _ = util.synth_stream(["scm"], chip, n_blocks, g.block_size, seed=3 + chip, n_packets=3, edge_every=2)
    _, q, h, p = util.oracle_run(protos, chip, iq)                 # h rows: (preamble id, call, idx)
    assert len(h) > 15
    qn = npo.quantize_stream(iq, chip, g.block_size)
    gg = npo.geometry(chip, g.preamble_symbols, g.packet_symbols)
    rows, clean = [], []
    for pid, name in enumerate(protos):
        hits, qq = npo.search_stream(qn, PROTOCOLS[name][0], gg)
        rows.append(np.concatenate([np.full((len(hits), 1), pid, np.int64), hits], axis=1))
        clean.append(npo.slice_packets(qq, hits, gg, g.packet_symbols))
    rows, clean = np.concatenate(rows), np.concatenate(clean)
    assert np.array_equal(rows, h)
    assert np.array_equal(clean[:, :-1], p[:, :-1]) and np.array_equal(clean[:, -1] & 15, p[:, -1] & 15)
    assert not np.array_equal(clean, p), "no stale bit anywhere: the test tests nothing"
    assert np.array_equal(npo.stale_last_bytes(clean, rows, g.packet_symbols), p)


# ---- full-size machinery: the C generator, the threaded sharded decode, the bench goldens ---------------------------

def test_c_generator_equals_numpy_generator():
    """oracle/synth_gen.c == rtlamr_amd.synth (the numpy twin of the device generator), noise and planted bursts,
    at a stream offset, with packets clipped by both ends of the buffer."""
    from oracle import oracle as orc
    from rtlamr_amd import synth
    n, first = 1 << 16, 777_000
    pk = [synth.Packet(first - 3000, bytes(range(12)), 96, 30, -26),            # starts before the buffer
          synth.Packet(first + 20_000, bytes(range(50, 62)), 96, -30, 26),
          synth.Packet(first + 40_000, bytes(range(100, 192)), 736, 127, -128),   # saturates
          synth.Packet(first + n - 3000, bytes(range(12)), 96, 22, 23)]         # runs past the end
    for chip in (8, 72):
        a = synth.noise(n, 5, first)
        synth.plant(a, pk, chip, first_sample=first)
        b = orc.synth_stream(n, 5, first, pk, chip, n_threads=3)
        assert np.array_equal(a, b)
    # the second distribution (uniform random bytes, bench.py --data uniform): same twins, and it IS uniform
    a = synth.uniform(n, 5, first)
    b = orc.synth_stream(n, 5, first, [], 72, n_threads=3, uniform=True)
    assert np.array_equal(a, b)
    hist = np.bincount(a, minlength=256)
    assert hist.min() > 0.7 * hist.mean() and hist.max() < 1.3 * hist.mean()


@pytest.mark.parametrize("protos,chip,n_blocks", [(["scm"], 72, 700), (["idm"], 72, 320),
                                                 (["scm", "scm+", "idm", "r900"], 72, 330), (["scm"], 8, 3000)])
def test_sharded_oracle_equals_single_decoder(protos, chip, n_blocks):
    """oracle.decode_sharded (what the full-size GPU tests and the bench goldens use): block ranges on threads, each
    primed with ceil(PL/BS)+1 blocks, give exactly the single decoder's bitstream, hits and packets."""
    from oracle import oracle as orc
    o = OracleDecoder(protos, chip)
    bs = o.geom.block_size
    iq, _ = util.synth_stream(protos, chip, n_blocks, bs, seed=3, n_packets=24)
    _, q1, h1, p1 = util.oracle_run(protos, chip, iq)
    q2, h2, p2 = orc.decode_sharded(protos, chip, iq, n_threads=6)
    assert len(h1) > 0 and np.array_equal(q1, q2) and np.array_equal(h1, h2) and np.array_equal(p1, p2)
    q3, h3, p3 = orc.decode_sharded(protos, chip, iq, n_threads=6, first_block=9)
    keep = h1[:, 1] >= 9
    assert np.array_equal(q1[9 * bs // 8:], q3) and np.array_equal(h1[keep], h3) and np.array_equal(p1[keep], p3)


def test_bench_goldens_come_from_the_oracle_and_reproduce():
    """tests/golden/bench_golden.json: declared oracle-made, complete (every workload x shard 0..7 at full size), and
    its small-size entries recompute to the same digests here (same schedule code, same oracle)."""
    import bench
    from tests.golden import make_bench_golden as mk
    gold = json.load(open(os.path.join(HERE, "golden", "bench_golden.json")))
    assert gold["source"].startswith("oracle")
    for spec in mk.ALL:
        wl = bench.workload(spec)
        nb = wl["nbytes"] // OracleDecoder(wl["protos"], wl["chip"]).geom.block_size2
        for shard in range(8):
            for state in ("first", "steady"):
                e = gold[mk.key(wl["name"], nb, shard)][state]
                assert e["n_hits"] > 4096 and len(e["hits_sha256"]) == len(e["q_sha256"]) == len(e["pkt_sha256"]) == 64
                assert 0 < e["validated"]["n_hits"] < e["n_hits"]
    small = [k for k in gold if "|blocks=4096|" in k or "|blocks=2048|" in k]
    assert len(small) >= 4
    for k in small:
        spec, nb, shard = k.split("|")
        got = mk.golden_for(spec, int(shard.split("=")[1]), int(nb.split("=")[1]), threads=4)
        assert got == gold[k], k
