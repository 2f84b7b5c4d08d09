"""SCM parser: host-side mirror of rtlamr's scm package (scm/scm.go).  Per-packet CPU work that
sits downstream of the GPU hot path; kept so planted packets can be recovered end to end."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

from ...protocol import Data, Message, PacketConfig, Parser, register_parser
from .crc import CRC


@dataclass
class SCM(Message):
    """scm.SCM (scm/scm.go:94-101); field extraction of NewSCM (scm/scm.go:103-119)."""
    ID: int = 0
    Type: int = 0
    TamperPhy: int = 0
    TamperEnc: int = 0
    Consumption: int = 0
    ChecksumVal: int = 0

    @staticmethod
    def from_data(d: Data) -> "SCM":
        b = d.Bits
        return SCM(ID=int(b[21:23] + b[56:80], 2), Type=int(b[26:30], 2), TamperPhy=int(b[24:26], 2),
                   TamperEnc=int(b[30:32], 2), Consumption=int(b[32:56], 2), ChecksumVal=int(b[80:96], 2))

    def MsgType(self): return "SCM"
    def MeterID(self): return self.ID
    def MeterType(self): return self.Type
    def Checksum(self): return self.ChecksumVal.to_bytes(2, "big")


class ScmParser(Parser):
    """scm.Parser (scm/scm.go:33-91)."""
    VALIDATOR = {"dedupe_bytes": 12, "checks": [(0x0000, 0x6F63, 0x0000, [(2, 10)])]}   # scm.go:68-79

    def __init__(self, chip_length: int):
        self.crc = CRC("BCH", 0, 0x6F63, 0)
        self.cfg = PacketConfig(Protocol="scm", CenterFreq=912600155, DataRate=32768, ChipLength=chip_length,
                                PreambleSymbols=21, PacketSymbols=96, Preamble="111110010101001100000")

    def Cfg(self) -> PacketConfig:
        return self.cfg

    def Parse(self, pkts: List[Data]) -> List[Message]:
        seen = set()
        out: List[Message] = []
        for pkt in pkts:
            data = Data(Idx=pkt.Idx, Bits=pkt.Bits[0:self.cfg.PacketSymbols], Bytes=pkt.Bytes[:12])
            if data.Bytes in seen:        # scm.go:69-73
                continue
            seen.add(data.Bytes)
            if self.crc.Checksum(data.Bytes[2:12]) != 0:   # scm.go:76
                continue
            m = SCM.from_data(data)
            if m.ID == 0:                 # scm.go:83
                continue
            out.append(m)
        return out


def build_packet(meter_id: int, meter_type: int, consumption: int, tamper_phy: int = 0, tamper_enc: int = 0) -> bytes:
    """A CRC-valid 96-bit SCM packet (inverse of NewSCM; BCH residue 0 over bytes 2..11)."""
    bits = "111110010101001100000"
    bits += f"{(meter_id >> 24) & 3:02b}" + "0" + f"{tamper_phy & 3:02b}" + f"{meter_type & 15:04b}" + f"{tamper_enc & 3:02b}"
    bits += f"{consumption & 0xFFFFFF:024b}" + f"{meter_id & 0xFFFFFF:024b}"
    body = int(bits, 2).to_bytes(10, "big")
    crc = CRC("BCH", 0, 0x6F63, 0).Checksum(body[2:])
    return body + crc.to_bytes(2, "big")


register_parser("scm", ScmParser)
