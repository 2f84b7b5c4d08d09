"""Table-driven MSB-first CRC-16, as rtlamr's crc package (crc/crc.go:14-55)."""
from __future__ import annotations

from typing import List


def new_table(poly: int) -> List[int]:
    """crc.NewTable (crc/crc.go:34-47)."""
    table = []
    for t in range(256):
        crc = (t << 8) & 0xFFFF
        for _ in range(8):
            crc = ((crc << 1) ^ poly) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
        table.append(crc)
    return table


class CRC:
    """crc.CRC (crc/crc.go:5-22)."""

    def __init__(self, name: str, init: int, poly: int, residue: int):
        self.Name, self.Init, self.Poly, self.Residue = name, init, poly, residue
        self._tbl = new_table(poly)

    def Checksum(self, data: bytes) -> int:
        """crc.Checksum (crc/crc.go:49-55)."""
        crc = self.Init
        for v in data:
            crc = ((crc << 8) & 0xFFFF) ^ self._tbl[(crc >> 8) ^ v]
        return crc
