"""IDM / NetIDM / SCM+ parsers: host-side mirrors of rtlamr's idm, netidm and scmplus packages
(idm/idm.go, netidm/netidm.go, scmplus/scmplus.go) -- CRC gating and the identifying fields."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

from ...protocol import Data, Message, PacketConfig, Parser, register_parser
from .crc import CRC

IDM_PREAMBLE = "01010101010101010001011010100011"


@dataclass
class IDM(Message):
    """idm.IDM (idm/idm.go:101-119), fields of NewIDM (idm/idm.go:121-148)."""
    Preamble: int = 0
    PacketTypeID: int = 0
    PacketLength: int = 0
    HammingCode: int = 0
    ApplicationVersion: int = 0
    ERTType: int = 0
    ERTSerialNumber: int = 0
    ConsumptionIntervalCount: int = 0
    ModuleProgrammingState: int = 0
    TamperCounters: bytes = b""
    AsynchronousCounters: int = 0
    PowerOutageFlags: bytes = b""
    LastConsumptionCount: int = 0
    DifferentialConsumptionIntervals: List[int] = field(default_factory=list)
    TransmitTimeOffset: int = 0
    SerialNumberCRC: int = 0
    PacketCRC: int = 0

    @staticmethod
    def from_data(d: Data) -> "IDM":
        B, bits = d.Bytes, d.Bits
        be = lambda b: int.from_bytes(b, "big")
        iv = [int(bits[264 + 9 * i: 273 + 9 * i], 2) for i in range(47)]
        return IDM(be(B[0:4]), B[4], B[5], B[6], B[7], B[8] & 0x0F, be(B[9:13]), B[13], B[14], B[15:21],
                   be(B[21:23]), B[23:29], be(B[29:33]), iv, be(B[86:88]), be(B[88:90]), be(B[90:92]))

    def MsgType(self): return "IDM"
    def MeterID(self): return self.ERTSerialNumber
    def MeterType(self): return self.ERTType
    def Checksum(self): return self.PacketCRC.to_bytes(2, "big")


class IdmParser(Parser):
    """idm.Parser (idm/idm.go:34-98): packet CRC over bytes 4..91 and serial-number CRC."""
    NAME, MSG = "idm", "IDM"
    # what Parse tests before anything else, for Decoder.EnableValidation (idm.go:68-87; netidm.go:79-98 is the same)
    VALIDATOR = {"dedupe_bytes": 92, "checks": [(0xFFFF, 0x1021, 0x1D0F, [(4, 88)]),
                                                (0xFFFF, 0x1021, 0x1D0F, [(9, 4), (88, 2)])]}

    def __init__(self, chip_length: int):
        self.crc = CRC("CCITT", 0xFFFF, 0x1021, 0x1D0F)
        self.cfg = PacketConfig(Protocol=self.NAME, CenterFreq=912600155, DataRate=32768, ChipLength=chip_length,
                                PreambleSymbols=32, PacketSymbols=92 * 8, Preamble=IDM_PREAMBLE)

    def Cfg(self): return self.cfg

    def Parse(self, pkts: List[Data]) -> List[Message]:
        seen, out = set(), []
        for pkt in pkts:
            data = Data(Idx=pkt.Idx, Bits=pkt.Bits[0:self.cfg.PacketSymbols], Bytes=pkt.Bytes[:92])
            if data.Bytes in seen:
                continue
            seen.add(data.Bytes)
            if self.crc.Checksum(data.Bytes[4:92]) != self.crc.Residue:      # idm.go:77
                continue
            if self.crc.Checksum(data.Bytes[9:13] + data.Bytes[88:90]) != self.crc.Residue:   # idm.go:82-87
                continue
            m = IDM.from_data(data)
            if m.ERTSerialNumber == 0:
                continue
            out.append(m)
        return out


@dataclass
class NetIDM(Message):
    """netidm.NetIDM (netidm/netidm.go:114-131), fields of NewNetIDM (netidm/netidm.go:133-161)."""
    Preamble: int = 0
    ProtocolID: int = 0
    PacketLength: int = 0
    HammingCode: int = 0
    ApplicationVersion: int = 0
    ERTType: int = 0
    ERTSerialNumber: int = 0
    ConsumptionIntervalCount: int = 0
    ProgrammingState: int = 0
    LastGeneration: int = 0
    LastConsumption: int = 0
    LastConsumptionNet: int = 0
    DifferentialConsumptionIntervals: List[int] = field(default_factory=list)
    TransmitTimeOffset: int = 0
    SerialNumberCRC: int = 0
    PacketCRC: int = 0

    @staticmethod
    def from_data(d: Data) -> "NetIDM":
        B, bits = d.Bytes, d.Bits
        be = lambda b: int.from_bytes(b, "big")
        iv = [int(bits[304 + 14 * i: 318 + 14 * i], 2) for i in range(27)]     # netidm.go:149-155
        return NetIDM(be(B[0:4]), B[4], B[5], B[6], B[7], B[8] & 0x0F, be(B[9:13]), B[13], B[14],
                      be(B[28:31]), be(B[25:28]), be(B[34:38]), iv, be(B[86:88]), be(B[88:90]), be(B[90:92]))

    def MsgType(self): return "NetIDM"
    def MeterID(self): return self.ERTSerialNumber
    def MeterType(self): return self.ERTType
    def Checksum(self): return self.PacketCRC.to_bytes(2, "big")


class NetIdmParser(IdmParser):
    """netidm.Parser (netidm/netidm.go:57-111): same preamble, length and both CRC checks as IDM (one shared
    Search, decode.go:124); only the message layout differs."""
    NAME, MSG = "netidm", "NetIDM"

    def Parse(self, pkts: List[Data]) -> List[Message]:
        seen, out = set(), []
        for pkt in pkts:
            data = Data(Idx=pkt.Idx, Bits=pkt.Bits[0:self.cfg.PacketSymbols], Bytes=pkt.Bytes[:92])
            if data.Bytes in seen:
                continue
            seen.add(data.Bytes)
            if self.crc.Checksum(data.Bytes[4:92]) != self.crc.Residue:      # netidm.go:88
                continue
            if self.crc.Checksum(data.Bytes[9:13] + data.Bytes[88:90]) != self.crc.Residue:   # netidm.go:93-98
                continue
            m = NetIDM.from_data(data)
            if m.ERTSerialNumber == 0:
                continue
            out.append(m)
        return out


@dataclass
class SCMPlus(Message):
    """scmplus.SCM (scmplus/scmplus.go:95-104)."""
    FrameSync: int = 0
    ProtocolID: int = 0
    EndpointType: int = 0
    EndpointID: int = 0
    Consumption: int = 0
    Tamper: int = 0
    PacketCRC: int = 0

    def MsgType(self): return "SCM+"
    def MeterID(self): return self.EndpointID
    def MeterType(self): return self.EndpointType
    def Checksum(self): return self.PacketCRC.to_bytes(2, "big")


class ScmPlusParser(Parser):
    """scmplus.Parser (scmplus/scmplus.go:40-92): CCITT over bytes 2..15."""
    VALIDATOR = {"dedupe_bytes": 16, "checks": [(0xFFFF, 0x1021, 0x1D0F, [(2, 14)])]}   # scmplus.go:68-79

    def __init__(self, chip_length: int):
        self.crc = CRC("CCITT", 0xFFFF, 0x1021, 0x1D0F)
        self.cfg = PacketConfig(Protocol="scm+", CenterFreq=912600155, DataRate=32768, ChipLength=chip_length,
                                PreambleSymbols=16, PacketSymbols=16 * 8, Preamble="0001011010100011")

    def Cfg(self): return self.cfg

    def Parse(self, pkts: List[Data]) -> List[Message]:
        seen, out = set(), []
        for pkt in pkts:
            B = pkt.Bytes[:16]
            if B in seen:
                continue
            seen.add(B)
            if self.crc.Checksum(B[2:]) != self.crc.Residue:
                continue
            be = lambda b: int.from_bytes(b, "big")
            m = SCMPlus(be(B[0:2]), B[2], B[3], be(B[4:8]), be(B[8:12]), be(B[12:14]), be(B[14:16]))
            if m.EndpointID == 0 or m.ProtocolID != 0x1E:   # scmplus.go:84
                continue
            out.append(m)
        return out


def build_idm_packet(serial: int, ert_type: int = 7, consumption: int = 0, fill: int = 0x5A) -> bytes:
    """A 92-byte IDM packet that passes both CRC checks of idm.Parser.Parse."""
    crc = CRC("CCITT", 0xFFFF, 0x1021, 0x1D0F)
    B = bytearray([fill] * 92)
    B[0:4] = bytes([0x55, 0x55, 0x16, 0xA3])
    B[4], B[5], B[6], B[7], B[8] = 0x1C, 0x5C, 0xC6, 0x04, ert_type & 0x0F
    B[9:13] = serial.to_bytes(4, "big")
    B[29:33] = consumption.to_bytes(4, "big")
    B[88:90] = (crc.Checksum(bytes(B[9:13])) ^ 0xFFFF).to_bytes(2, "big")
    B[90:92] = (crc.Checksum(bytes(B[4:90])) ^ 0xFFFF).to_bytes(2, "big")
    return bytes(B)


def build_scmplus_packet(endpoint_id: int, endpoint_type: int = 0x9C, consumption: int = 0) -> bytes:
    crc = CRC("CCITT", 0xFFFF, 0x1021, 0x1D0F)
    B = bytearray(16)
    B[0:2] = bytes([0x16, 0xA3])
    B[2], B[3] = 0x1E, endpoint_type & 0xFF
    B[4:8] = endpoint_id.to_bytes(4, "big")
    B[8:12] = consumption.to_bytes(4, "big")
    B[12:14] = (0x0248).to_bytes(2, "big")
    B[14:16] = (crc.Checksum(bytes(B[2:14])) ^ 0xFFFF).to_bytes(2, "big")
    return bytes(B)


register_parser("idm", IdmParser)
register_parser("netidm", NetIdmParser)
register_parser("scm+", ScmPlusParser)
