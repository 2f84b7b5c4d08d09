"""GF(2^k) tables and Reed-Solomon syndromes: host-side mirror of rtlamr's r900/gf package (r900/gf/gf.go),
used by the r900 parser downstream of the GPU path (per-packet CPU work)."""
from __future__ import annotations

from typing import List


def _mul(x: int, y: int, order: int, poly: int) -> int:
    """gf.go:81-94: carry-less multiply modulo poly."""
    z = 0
    while x > 0:
        if x & 1:
            z ^= y
        x >>= 1
        y <<= 1
        if y & order:
            y ^= poly
    return z


class Field:
    """gf.NewField (gf.go:20-57): log/exp tables for GF(order) with generator alpha."""

    def __init__(self, order: int, poly: int, alpha: int):
        self.order = order - 1
        self.log = [0] * order
        self.exp = [0] * ((order - 1) << 1)
        x = 1
        for i in range(self.order):
            if x == 1 and i != 0:
                raise ValueError("gf: invalid generator")
            self.exp[i] = x
            self.exp[i + self.order] = x
            self.log[x] = i
            x = _mul(x, alpha, order, poly)
        self.log[0] = self.order

    def Exp(self, e: int) -> int:   # gf.go:115-120
        return 0 if e < 0 else self.exp[e % self.order]

    def Inv(self, x: int) -> int:   # gf.go:133-138
        return 0 if x == 0 else self.exp[self.order - self.log[x]]

    def Mul(self, x: int, y: int) -> int:   # gf.go:141-146
        if x == 0 or y == 0:
            return 0
        return self.exp[self.log[x] + self.log[y]]

    def Syndrome(self, message: List[int], parity_symbol_count: int, offset: int) -> List[int]:
        """gf.go:150-172: Horner evaluation of the message polynomial at alpha^(offset+i)."""
        out = []
        for i in range(parity_symbol_count):
            syn = message[0]
            a = self.Exp(offset + i)
            for v in message[1:]:
                syn = self.Mul(syn, a) ^ v
            out.append(syn)
        return out
