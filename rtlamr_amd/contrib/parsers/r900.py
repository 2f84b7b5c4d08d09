"""R900: host-side mirror of rtlamr's r900 package (r900/r900.go).

The r900 parser is the one parser that is not cheap in the reference: on EVERY block it runs a second matched
filter over its own BufferLength-long history of Decoder.Signal -- a fresh sequential float32 running sum plus a
6-ary symbol quantizer (r900.go:82-150) -- and then reads 42 of those symbols per preamble hit (r900.go:187-193).
Here the 42 base-6 digits of every r900 preamble hit are computed on the GPU (kernel k4_r900_digits, bit-exact
with the reference's running sum) and arrive with the hit as Data.Digits; this module does the rest of
Parser.Parse unchanged: digit pairs -> 5-bit symbols, dedupe, Reed-Solomon syndrome check, field extraction.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence

from ...protocol import Data, Message, PacketConfig, Parser, register_parser
from .gf import Field

PAYLOAD_SYMBOLS = 42  # r900/r900.go:30


@dataclass
class R900(Message):
    """r900.R900 (r900.go:250-261)."""
    ID: int = 0
    Unkn1: int = 0
    NoUse: int = 0
    BackFlow: int = 0
    Consumption: int = 0
    Unkn3: int = 0
    Leak: int = 0
    LeakNow: int = 0
    checksum: bytes = b""

    def MsgType(self): return "R900"
    def MeterID(self): return self.ID
    def MeterType(self): return self.Unkn1
    def Checksum(self): return self.checksum


class R900Parser(Parser):
    ALWAYS_PARSE = False   # nothing to do for a block without r900 hits: the per-block filter has moved to the GPU
    NEEDS_R900_DIGITS = True

    def __init__(self, chip_length: int):
        self.cfg = PacketConfig(Protocol="r900", CenterFreq=912380000, DataRate=32768, ChipLength=chip_length,
                                PreambleSymbols=32, PacketSymbols=116, Preamble="00000000000000001110010101100100")
        self.field = Field(32, 37, 2)   # r900.go:67-68: GF of order 32, polynomial 37, generator 2
        self.decoder = None

    def SetDecoder(self, d) -> None:   # r900.go:73-75
        self.decoder = d

    def Cfg(self) -> PacketConfig:
        return self.cfg

    def Parse(self, pkts: List[Data]) -> List[Message]:
        """r900.go:160-248 from the point where the digits are known (r900.go:187-193 reads them from the
        parser's quantized buffer; here they come with the hit)."""
        out: List[Message] = []
        seen = set()
        for pkt in pkts:
            digits = getattr(pkt, "Digits", None)
            if digits is None:
                raise RuntimeError("r900: hit without digits (the decoder was not told about the r900 parser)")
            msg = parse_digits(self.field, digits, seen)
            if msg is not None:
                out.append(msg)
        return out


def parse_digits(field: Field, digits: Sequence[int], seen: set):
    """r900.go:195-246 for one hit."""
    symbols = [0] * 21
    bits = ""
    for i in range(0, PAYLOAD_SYMBOLS, 2):
        symbol = int(digits[i]) * 6 + int(digits[i + 1])      # strconv.ParseInt(digits[idx:idx+2], 6, 32)
        if symbol > 31:
            return None                                       # badSymbol
        symbols[i >> 1] = symbol
        bits += format(symbol, "05b")
    if bits in seen:
        return None
    seen.add(bits)
    rs = [0] * 31
    rs[:16] = symbols[:16]
    rs[26:] = symbols[16:]
    if any(field.Syndrome(rs, 5, 29)):
        return None
    return R900(ID=int(bits[:32], 2), Unkn1=int(bits[32:40], 2), NoUse=int(bits[40:46], 2), BackFlow=int(bits[46:48], 2),
                Consumption=int(bits[48:72], 2), Unkn3=int(bits[72:74], 2), Leak=int(bits[74:78], 2),
                LeakNow=int(bits[78:80], 2), checksum=bytes(symbols[16:]))


def build_r900_symbols(meter_id: int, unkn1: int = 0x5A, nouse: int = 3, backflow: int = 1, consumption: int = 123456,
                       unkn3: int = 2, leak: int = 5, leaknow: int = 1) -> List[int]:
    """Test/bench support: the 21 five-bit symbols of a packet that passes r900.Parser.Parse -- 80 payload bits in
    16 symbols plus 5 parity symbols solved from Syndrome(rsBuf, 5, 29) == 0 (linear over GF(32))."""
    f = Field(32, 37, 2)
    bits = (format(meter_id & 0xFFFFFFFF, "032b") + format(unkn1 & 0xFF, "08b") + format(nouse & 0x3F, "06b") +
            format(backflow & 3, "02b") + format(consumption & 0xFFFFFF, "024b") + format(unkn3 & 3, "02b") +
            format(leak & 0xF, "04b") + format(leaknow & 3, "02b"))
    data = [int(bits[i:i + 5], 2) for i in range(0, 80, 5)]
    # syndrome i = sum_j rs[j] * a_i^(30-j), a_i = alpha^(29+i); unknowns: rs[26..30]
    A = [[f.Exp((29 + i) * (30 - j)) for j in range(26, 31)] for i in range(5)]
    rhs = []
    for i in range(5):
        s = 0
        for j in range(16):
            s ^= f.Mul(data[j], f.Exp((29 + i) * (30 - j)))
        rhs.append(s)
    # Gaussian elimination over GF(32)
    M = [A[i] + [rhs[i]] for i in range(5)]
    for c in range(5):
        p = next(r for r in range(c, 5) if M[r][c])
        M[c], M[p] = M[p], M[c]
        inv = f.Inv(M[c][c])
        M[c] = [f.Mul(v, inv) for v in M[c]]
        for r in range(5):
            if r != c and M[r][c]:
                k = M[r][c]
                M[r] = [a ^ f.Mul(k, b) for a, b in zip(M[r], M[c])]
    parity = [M[i][5] for i in range(5)]
    return data + parity


# digit -> the four chips of the symbol kernel whose correlation is largest and positive/negative
# (r900.go:104-142: 1100/1010/1001 give argmax 0/1/2 with a positive value -> +3; their inversions stay 0..2)
DIGIT_CHIPS = {0: (0, 0, 1, 1), 1: (0, 1, 0, 1), 2: (0, 1, 1, 0), 3: (1, 1, 0, 0), 4: (1, 0, 1, 0), 5: (1, 0, 0, 1)}


def symbols_to_chips(symbols: Sequence[int]) -> List[int]:
    """21 symbols -> 42 base-6 digits -> 168 chips (1 = carrier on)."""
    chips: List[int] = []
    for s in symbols:
        for d in (s // 6, s % 6):
            chips.extend(DIGIT_CHIPS[d])
    return chips


class R900BCD(R900):
    """r900bcd.R900BCD (r900bcd/r900bcd.go:39-45): an R900 whose consumption was transmitted as BCD."""

    def MsgType(self): return "R900BCD"


class R900BcdParser(R900Parser):
    """r900bcd.Parser (r900bcd/r900bcd.go:31-72): the r900 parser (same PacketConfig, so Cfg().Protocol stays "r900"
    and the second stage on the GPU is the same), consumption re-read as decimal digits of its hex form."""

    def Parse(self, pkts: List[Data]) -> List[Message]:
        out: List[Message] = []
        for m in super().Parse(pkts):
            try:
                consumption = int(format(m.Consumption, "x"), 10) & 0xFFFFFFFF   # strconv.FormatUint(.., 16) -> ParseUint(.., 10, 32)
            except ValueError:                                                     # a hex letter: ParseUint fails, Go keeps 0
                consumption = 0
            out.append(R900BCD(m.ID, m.Unkn1, m.NoUse, m.BackFlow, consumption, m.Unkn3, m.Leak, m.LeakNow, m.checksum))
        return out


register_parser("r900", R900Parser)
register_parser("r900bcd", R900BcdParser)
