"""Host-side mirror of package protocol + parsers + synthetic generator (CPU only)."""
import numpy as np
import pytest

import rtlamr_amd as ra
from rtlamr_amd import dist, synth
from rtlamr_amd.contrib.parsers.crc import CRC
from rtlamr_amd.contrib.parsers.idm import IdmParser, NetIdmParser, ScmPlusParser, build_idm_packet, build_scmplus_packet
from rtlamr_amd.contrib.parsers.scm import SCM, ScmParser, build_packet


def test_new_data_bits_and_copy():  # parse.go:61-69
    src = bytearray(b"\xf9\x53\x00")
    d = ra.new_data(src)
    src[0] = 0
    assert d.Bytes == b"\xf9\x53\x00" and d.Bits == "111110010101001100000000" and d.Idx == 0


def test_parser_registry():  # parse.go:28-51
    assert isinstance(ra.new_parser("scm", 72), ScmParser)
    with pytest.raises(ValueError, match="invalid message type"):
        ra.new_parser("nope", 72)
    with pytest.raises(RuntimeError, match="already registered"):
        ra.register_parser("scm", ScmParser)
    with pytest.raises(RuntimeError, match="nil"):
        ra.register_parser("x", None)


def test_register_protocol_takes_field_wise_max():  # decode.go:100-128
    d = ra.new_decoder()
    for name in ["scm", "scm+", "idm", "r900"]:   # "all", main.go:67-73
        d.RegisterProtocol(ra.new_parser(name, 72))
    c = d.Cfg
    assert (c.DataRate, c.ChipLength, c.PreambleSymbols, c.PacketSymbols) == (32768, 72, 32, 736)
    assert c.CenterFreq == 912380000  # last registered wins (decode.go:105)
    assert len(d._preamble_strs) == 4
    d2 = ra.new_decoder()
    d2.RegisterProtocol(ra.new_parser("idm", 72))
    d2.RegisterProtocol(ra.new_parser("netidm", 72))
    assert len(d2._preamble_strs) == 1 and len(d2._preambles[d2._preamble_strs[0]]) == 2  # shared Search


def test_next_power_of_2():  # decode.go:377-379
    assert [ra.next_power_of_2(v) for v in (336, 1344, 3024, 4096, 4608)] == [512, 2048, 4096, 4096, 8192]


@pytest.mark.parametrize("name,init,poly,residue", [("IBM", 0, 0x8005, 0), ("BCH", 0, 0x6F63, 0),
                                                    ("CCITT", 0xFFFF, 0x1021, 0x1D0F)])
def test_crc_identity(name, init, poly, residue):
    """crc/crc_test.go:16-37: appending the checksum (complemented for CCITT) leaves the residue."""
    rng = np.random.default_rng(1)
    crc = CRC(name, init, poly, residue)
    for _ in range(64):
        msg = rng.integers(0, 256, rng.integers(1, 64), dtype=np.uint8).tobytes()
        cs = crc.Checksum(msg)
        if name == "CCITT":
            cs ^= 0xFFFF
        assert crc.Checksum(msg + cs.to_bytes(2, "big")) == residue


def test_scm_packet_roundtrip_and_rejects():
    p = ScmParser(72)
    pkt = build_packet(0x2ABCDEF, 7, 123456, tamper_phy=2, tamper_enc=1)
    (m,) = p.Parse([ra.new_data(pkt)])
    assert (m.ID, m.Type, m.Consumption, m.TamperPhy, m.TamperEnc) == (0x2ABCDEF, 7, 123456, 2, 1)
    assert m.MsgType() == "SCM" and m.MeterID() == m.ID and m.Checksum() == pkt[10:12]
    bad = bytearray(pkt)
    bad[5] ^= 1
    assert p.Parse([ra.new_data(bytes(bad))]) == []           # CRC fails (scm.go:76)
    assert len(p.Parse([ra.new_data(pkt), ra.new_data(pkt)])) == 1   # dedupe by bytes (scm.go:69-73)
    assert p.Parse([ra.new_data(build_packet(0, 7, 1))]) == []  # ID 0 rejected (scm.go:83)


def test_idm_scmplus_roundtrip():
    (m,) = IdmParser(72).Parse([ra.new_data(build_idm_packet(123456789, ert_type=8, consumption=4242))])
    assert (m.ERTSerialNumber, m.ERTType, m.LastConsumptionCount, m.Preamble) == (123456789, 8, 4242, 0x555516A3)
    assert len(NetIdmParser(72).Parse([ra.new_data(build_idm_packet(5))])) == 1
    (s,) = ScmPlusParser(72).Parse([ra.new_data(build_scmplus_packet(777, consumption=99))])
    assert (s.EndpointID, s.Consumption, s.FrameSync) == (777, 99, 0x16A3)


def test_synth_generator_is_deterministic_and_position_based():
    a = synth.noise(4096, seed=5, first_sample=0)
    b = synth.noise(1024, seed=5, first_sample=1000)
    assert np.array_equal(a[2000:2000 + 2048], b)
    assert 126.5 < a[0::2].mean() < 127.5 and 127.5 < a[1::2].mean() < 128.5
    assert a.min() >= 119 and a.max() <= 136
    assert int(synth.splitmix64(np.array([0], np.uint64))[0]) == 0xE220A8397B1DCDAF  # published splitmix64 vector


def test_plant_is_clamped_and_manchester():
    iq = np.full(2 * 1000, 250, np.uint8)
    synth.plant(iq, [synth.Packet(100, b"\x80", 2, 20, -20)], 8)   # bits 1,0
    I = iq[0::2]
    assert I[100:108].tolist() == [255] * 8 and I[108:116].tolist() == [250] * 8   # bit 1: high, low
    assert I[116:124].tolist() == [250] * 8 and I[124:132].tolist() == [255] * 8   # bit 0: low, high
    assert iq[1::2][100:108].tolist() == [230] * 8


def test_shard_ranges_cover_stream_exactly():
    for total, world in [(131072, 8), (1000, 3), (7, 8)]:
        spans = [dist.shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert dist.prime_range(100, 5) == (95, 100) and dist.prime_range(3, 5) == (0, 3)


def test_decode_short_input_mirrors_go_panic():
    d = ra.new_decoder()
    d.RegisterProtocol(ra.new_parser("scm", 72))
    with pytest.raises(RuntimeError, match="Allocate"):
        d.decode_batch(np.zeros(8192, np.uint8))


def test_run_parsers_groups_hits_per_block_and_preamble():
    """decode.go:177-187 in the mirror, without a device: hits of a batch are handed to the parsers block by block and
    preamble by preamble (idm and netidm share one preamble: both parsers see the same packets), with Data.Idx set."""
    import numpy as np
    from rtlamr_amd.protocol import BatchResult
    d = ra.new_decoder()
    for n in ("scm", "idm", "netidm"):
        d.RegisterProtocol(ra.new_parser(n, 72))
    pre_scm, pre_idm = ra.new_parser("scm", 72).Cfg().Preamble, ra.new_parser("idm", 72).Cfg().Preamble
    d._pid_of_preamble = {pre_scm: 0, pre_idm: 1}      # what Allocate reads back from amr_preamble_id
    d.n_preambles = 2
    scm_a, scm_b = build_packet(111, 4, 5), build_packet(222, 7, 9)
    idm_a = build_idm_packet(333, consumption=1)
    pkt = np.zeros((5, 92), np.uint8)
    for i, b in enumerate((scm_a, scm_a, scm_b, idm_a, idm_a)):
        pkt[i, : len(b)] = np.frombuffer(b, np.uint8)
    br = BatchResult(n_blocks=3, first_block=10, preamble_offset=np.array([0, 3, 5], np.uint64),
                     hit_block=np.array([10, 10, 12, 11, 11], np.uint64), hit_idx=np.array([5, 6, 9, 70, 71], np.uint32),
                     pkt=pkt)
    per_block = d.run_parsers(br)
    assert [sorted((m.MsgType(), m.MeterID()) for m in b) for b in per_block] == [
        [("SCM", 111)],                              # block 10: two hits, identical bytes -> one message (seen map)
        [("IDM", 333), ("NetIDM", 333)],             # block 11: the shared preamble feeds both parsers
        [("SCM", 222)]]                              # block 12


def test_message_records_have_the_reference_columns():
    """Message.Record / String (parse.go:78-84): idm.IDM (idm/idm.go:176-221): 16 scalar columns + 47
    intervals; netidm (netidm/netidm.go:186-235): 15 + 27; hex fields zero-padded upper case, byte-slice fields as plain
    hex, the serial right-aligned in String; scm (scm/scm.go:139-154), scm+ (scmplus/scmplus.go:129-150), r900
    (r900/r900.go:278-302): Record uses strconv's lower-case unpadded hex, String fmt's padded upper case."""
    from rtlamr_amd.contrib.parsers.idm import IDM, NetIDM, SCMPlus
    from rtlamr_amd.contrib.parsers.r900 import R900
    from rtlamr_amd.contrib.parsers.scm import SCM
    m = IDM(0x555516A3, 0x1C, 0x5C, 0xC6, 4, 7, 12345678, 3, 0xBC, bytes([1, 2, 3, 4, 5, 6]), 0x12, bytes(6), 99,
            list(range(47)), 17, 0xABCD, 0x1D0F)
    r = m.Record()
    assert len(r) == 16 + 47 and r[0] == "0x555516A3" and r[4] == "0x04" and r[6] == "12345678"
    assert r[9] == "010203040506" and r[10] == "0x12" and r[13:13 + 47] == [str(i) for i in range(47)]
    assert r[-3:] == ["17", "0xABCD", "0x1D0F"]
    s = str(m)
    assert s.startswith("{Preamble:0x555516A3 PacketTypeID:0x1C ") and "ERTSerialNumber:  12345678 " in s
    assert "DifferentialConsumptionIntervals:[0 1 2 " in s and s.endswith("PacketCRC:0x1D0F}")
    n = NetIDM(0x555516A3, 0x1C, 0x5C, 0xC6, 4, 7, 123, 3, 0xBC, 5, 6, 7, list(range(27)), 17, 0xABCD, 0x1D0F)
    assert len(n.Record()) == 15 + 27 and n.Record()[9:12] == ["5", "6", "7"]
    assert "LastGeneration:5 LastConsumption:6 LastConsumptionNet:7 " in str(n)
    c = SCM(ID=17581447, Type=12, TamperPhy=2, TamperEnc=1, Consumption=5810, ChecksumVal=0x0B7E)
    assert c.Record() == ["17581447", "12", "0x2", "0x1", "5810", "0xb7e"]
    assert str(c) == "{ID:17581447 Type:12 Tamper:{Phy:02 Enc:01} Consumption:    5810 CRC:0x0B7E}"
    p = SCMPlus(0x16A3, 0x1E, 0xAB, 42, 1234, 0x0102, 0x00FF)
    assert p.Record() == ["0x16a3", "0x1e", "0xab", "42", "1234", "0x102", "0xff"]
    assert str(p) == "{ProtocolID:0x1E EndpointType:0xAB EndpointID:        42 Consumption:      1234 Tamper:0x0102 PacketCRC:0x00FF}"
    w = R900(ID=1234, Unkn1=0xA3, NoUse=5, BackFlow=1, Consumption=99, Unkn3=2, Leak=3, LeakNow=0, checksum=b"\x01\x02")
    assert w.Record() == ["1234", "163", "5", "1", "99", "2", "3", "0"]
    assert str(w) == "{ID:      1234 Unkn1:0xA3 NoUse: 5 BackFlow:1 Consumption:      99 Unkn3:0x02 Leak: 3 LeakNow:0}"
    from rtlamr_amd.contrib.parsers.r900 import R900BCD
    b = R900BCD(ID=1234, Unkn1=0xA3, NoUse=5, BackFlow=1, Consumption=99, Unkn3=2, Leak=3, LeakNow=0, checksum=b"\x01\x02")
    assert b.MsgType() == "R900BCD" and b.Record() == w.Record() and str(b) == str(w)     # r900bcd.go:39-45: embedded R900


def test_k1_timeline_report_finds_late_workgroups(tmp_path):
    """tools/k1_timeline_report.py (the diagnostic of profiles/r05/k1_gate_fragmentation.txt) on a synthetic dump: two launches per
    batch, in every first one four workgroups of one XCC start 330 us late."""
    import os, subprocess, sys
    rows = []
    t = 1_000_000
    for launch in range(6):
        for w in range(2048):
            late = 33_000 if (launch % 2 == 0 and w in (2019, 2027, 2035, 2043)) else 0
            st = t + (w % 8) * 3 + late
            rows.append(f"{launch} {w} {st} {st + 38_000 + (0 if late else w % 50)} {w % 8} 0")
        t += 41_000
    p = tmp_path / "tl.txt"
    p.write_text("# slot wg start end xcc hw\n" + "\n".join(rows) + "\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "k1_timeline_report.py"), str(p), "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    summary = [l for l in out.stdout.splitlines() if l.startswith("# launch ") and " per batch:" in l]
    assert len(summary) == 2
    assert "late>2us 4" in summary[0] and "late>2us 0" in summary[1], summary
    first = float(summary[0].split("span mean")[1].split()[0]); second = float(summary[1].split("span mean")[1].split()[0])
    assert first > second + 300


def test_patch_stale_carry_rule():
    """dist.patch_stale_carry against a literal replay of Decoder.Slice's d.pkt (decode.go:363-366) over two shards."""
    from rtlamr_amd import dist as shard
    rng = np.random.default_rng(4)
    for psym in (116, 117, 99, 96):
        r, nb = psym % 8, (psym + 7) // 8
        n = 9
        bits = rng.integers(0, 2, (n, psym), dtype=np.uint8)
        rows = np.stack([rng.integers(0, 2, n), np.sort(rng.integers(0, 5, n)), rng.integers(0, 4096, n)], axis=1).astype(np.int64)
        order = np.lexsort((rows[:, 2], rows[:, 0], rows[:, 1]))

        def slice_all(idxs, pkt):
            out = {}
            for j in idxs:
                for p in range(psym):
                    pkt[p >> 3] = ((pkt[p >> 3] << 1) | bits[j, p]) & 0xFF
                out[j] = pkt.copy()
            return out
        whole = slice_all(order, np.zeros(nb, np.int64))
        first, second = order[:4], order[4:]
        a = slice_all(first, np.zeros(nb, np.int64))
        b = slice_all(second, np.zeros(nb, np.int64))                 # a concurrent shard: zero start
        pk = np.array([b[j] for j in second], np.uint8)
        fixed = shard.patch_stale_carry(rows[second], pk, psym, int(a[first[-1]][-1]))
        assert np.array_equal(fixed, np.array([whole[j] for j in second], np.uint8)), psym
