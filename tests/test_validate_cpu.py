"""CPU side of the per-hit validation row: the oracle's CRC restatement against the reference's
own test property and known values, the host-side table CRC against it, and the filter semantics."""
import numpy as np

from oracle import validate_oracle as vo
from rtlamr_amd.contrib.parsers.crc import CRC
from rtlamr_amd.contrib.parsers.idm import build_idm_packet, build_scmplus_packet
from rtlamr_amd.contrib.parsers.scm import build_packet


def test_identity_property_of_reference_crc_test():
    """crc/crc_test.go:24-44 (TestIdentity): appending the checksum big-endian gives checksum 0, for its three CRCs."""
    rng = np.random.default_rng(7)
    for init, poly in ((0, 0x8005), (0, 0x6F63), (0xFFFF, 0x1021)):
        for _ in range(512):
            length = (int(rng.integers(0, 32)) & 0xFE) + 8
            buf = bytearray(rng.integers(0, 256, length, dtype=np.uint8).tobytes())
            inter = vo.checksum(init, poly, buf[:length - 2])
            buf[length - 2:] = inter.to_bytes(2, "big")
            assert vo.checksum(init, poly, buf) == 0


def test_known_values():
    assert vo.checksum(0xFFFF, 0x1021, b"123456789") == 0x29B1      # CRC-16/CCITT-FALSE catalogue check value
    # the CRC-valid SCM packets of assets/sample.bin at chip 78
    for hexpkt in ("f953026101b3360c4105d005", "f953036003b5e30c3a08f6bb", "f95303600c30220ab87c8069"):
        assert vo.passes("scm", bytes.fromhex(hexpkt))
    assert not vo.passes("scm", bytes.fromhex("f953026101b3360c4105d004"))


def test_host_table_crc_equals_bitwise_restatement():
    rng = np.random.default_rng(11)
    for name, init, poly, res in (("BCH", 0, 0x6F63, 0), ("CCITT", 0xFFFF, 0x1021, 0x1D0F)):
        c = CRC(name, init, poly, res)
        for _ in range(200):
            data = rng.integers(0, 256, int(rng.integers(1, 100)), dtype=np.uint8).tobytes()
            assert c.Checksum(data) == vo.checksum(init, poly, data)


def test_built_packets_pass_and_corrupted_fail():
    assert vo.passes("scm", build_packet(1234567, 7, 4242))
    assert vo.passes("idm", build_idm_packet(987654, consumption=99))
    assert vo.passes("netidm", build_idm_packet(55, consumption=1))
    assert vo.passes("scm+", build_scmplus_packet(777, consumption=5))
    bad = bytearray(build_idm_packet(987654)); bad[10] ^= 1      # serial number: both checks see it
    assert not vo.passes("idm", bytes(bad))
    bad = bytearray(build_idm_packet(987654)); bad[40] ^= 0x80    # payload: packet check only
    assert not vo.passes("idm", bytes(bad))


def test_filter_drops_failed_and_adjacent_repeats_only():
    good, other = build_packet(1, 1, 1), build_packet(2, 2, 2)
    bad = bytearray(good); bad[5] ^= 4
    pk = np.frombuffer(b"".join([good, good, bytes(bad), good, other, other, good]), np.uint8).reshape(-1, 12)
    blocks = np.array([3, 3, 3, 3, 3, 4, 4])
    # hit 1 repeats hit 0; hit 2 fails; hit 3 follows a different byte string (kept: only adjacent repeats go);
    # hit 5 is in another block than hit 4 (kept); hit 6 differs from hit 5
    assert vo.filter_hits("scm", blocks, pk).tolist() == [0, 3, 4, 5, 6]


def test_parser_validators_state_the_same_rules_as_the_oracle():
    """rtlamr_amd/contrib/parsers/*.VALIDATOR (what Decoder.EnableValidation hands to amr_set_validation) against the
    independently written rule table of oracle/validate_oracle.py."""
    import rtlamr_amd as ra
    for name, (nbytes, checks) in vo.RULES.items():
        v = ra.new_parser(name, 72).VALIDATOR
        assert v["dedupe_bytes"] == nbytes
        assert [(c[0], c[1], c[2], list(c[3])) for c in v["checks"]] == [(i, p, r, list(sp)) for (i, p, r), sp in checks]
    assert getattr(ra.new_parser("r900", 72), "VALIDATOR", None) is None     # r900 hits carry digits: never filtered


def test_netidm_message_layout():
    """netidm.NewNetIDM (netidm/netidm.go:133-161) on a packet with known field bytes."""
    import rtlamr_amd as ra
    from rtlamr_amd.contrib.parsers.idm import NetIdmParser
    pkt = bytearray(build_idm_packet(0x01020304, ert_type=7))
    pkt[25:28] = (0x0A0B0C).to_bytes(3, "big")     # LastConsumption
    pkt[28:31] = (0x0D0E0F).to_bytes(3, "big")     # LastGeneration
    pkt[34:38] = (0x11223344).to_bytes(4, "big")   # LastConsumptionNet
    crc = CRC("CCITT", 0xFFFF, 0x1021, 0x1D0F)
    pkt[90:92] = (crc.Checksum(bytes(pkt[4:90])) ^ 0xFFFF).to_bytes(2, "big")
    msgs = NetIdmParser(72).Parse([ra.new_data(bytes(pkt))])
    assert len(msgs) == 1
    m = msgs[0]
    assert (m.MsgType(), m.MeterID(), m.MeterType()) == ("NetIDM", 0x01020304, 7)
    assert (m.LastConsumption, m.LastGeneration, m.LastConsumptionNet) == (0x0A0B0C, 0x0D0E0F, 0x11223344)
    assert len(m.DifferentialConsumptionIntervals) == 27
    bad = bytearray(pkt); bad[9] ^= 1              # serial number: the serial-number CRC must reject it (netidm.go:93-98)
    bad[90:92] = (crc.Checksum(bytes(bad[4:90])) ^ 0xFFFF).to_bytes(2, "big")
    assert NetIdmParser(72).Parse([ra.new_data(bytes(bad))]) == []
