"""R900: configuration mirror of rtlamr's r900 package (r900/r900.go:54-71).

Only the PacketConfig is mirrored in this round: it is what Decoder.RegisterProtocol needs, so the
r900 preamble is searched on the GPU with the same geometry the reference would use ("all" =
scm, scm+, idm, r900; main.go:67-73).  The parser's second-stage 6-ary matched filter
(r900/r900.go:82-150, 160-248) is SURVEY.md section 8f row 1 ("next") and is not implemented yet:
Parse returns no messages.
"""
from __future__ import annotations

from typing import List

from ..protocol import Data, Message, PacketConfig, Parser, register_parser

PAYLOAD_SYMBOLS = 42  # r900/r900.go:30


class R900Parser(Parser):
    ALWAYS_PARSE = True   # the Go parser filters every block, even without hits (r900.go:160-172)

    def __init__(self, chip_length: int):
        self.cfg = PacketConfig(Protocol="r900", CenterFreq=912380000, DataRate=32768, ChipLength=chip_length,
                                PreambleSymbols=32, PacketSymbols=116, Preamble="00000000000000001110010101100100")
        self.decoder = None

    def SetDecoder(self, d) -> None:   # r900.go:73-75
        self.decoder = d

    def Cfg(self) -> PacketConfig:
        return self.cfg

    def Parse(self, pkts: List[Data]) -> List[Message]:
        return []


register_parser("r900", R900Parser)
