"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the first steps of every rtlamr Parse loop, used to check the
GPU's per-hit validation (rtlamr_amd/csrc/k5_validate.h).  Never imported by the product path.

What is restated (bitwise CRC, no table, so it shares nothing with the kernel or with rtlamr_amd/contrib/parsers/crc.py):
  crc.Checksum / crc.NewTable   crc/crc.go:34-55   MSB-first CRC-16, no reflection, no final xor
  scm.Parser.Parse              scm/scm.go:61-90       seen[string(Bytes)], Checksum(Bytes[2:12]) != 0
  scmplus.Parser.Parse          scmplus/scmplus.go:62-92   Checksum(Bytes[2:16]) != Residue
  idm / netidm Parser.Parse     idm/idm.go:62-98, netidm/netidm.go:73-110   Checksum(Bytes[4:92]), Checksum(Bytes[9:13]+Bytes[88:90])

Pinned (tests/test_validate_cpu.py) against: the identity property the reference's own crc_test.go checks
(crc/crc_test.go:24-44, all three of its CRCs), the CRC-16/CCITT-FALSE catalogue value of "123456789", and the three
CRC-valid SCM packets SURVEY.md 8c lists from assets/sample.bin.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

BCH = (0x0000, 0x6F63, 0x0000)      # scm/scm.go:36   crc.NewCRC("BCH", 0, 0x6F63, 0)
CCITT = (0xFFFF, 0x1021, 0x1D0F)    # idm/idm.go:41   crc.NewCRC("CCITT", 0xFFFF, 0x1021, 0x1D0F)

# preamble -> (bytes the parser keeps of a packet, checks); a check = (crc, spans of (offset, length))
RULES: Dict[str, Tuple[int, list]] = {
    "scm": (12, [(BCH, [(2, 10)])]),
    "scm+": (16, [(CCITT, [(2, 14)])]),
    "idm": (92, [(CCITT, [(4, 88)]), (CCITT, [(9, 4), (88, 2)])]),
    "netidm": (92, [(CCITT, [(4, 88)]), (CCITT, [(9, 4), (88, 2)])]),
}


def checksum(init: int, poly: int, data: Sequence[int]) -> int:
    """crc.Checksum (crc/crc.go:49-55) with the table of crc/crc.go:34-47 unrolled into its bit steps."""
    crc = init
    for v in data:
        crc ^= (int(v) & 0xFF) << 8
        for _ in range(8):
            crc = ((crc << 1) ^ poly) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc


def passes(proto: str, pkt: Sequence[int]) -> bool:
    """True when Parse of `proto` would get past its checksum tests with this packet."""
    _, checks = RULES[proto]
    for (init, poly, residue), spans in checks:
        buf: List[int] = []
        for off, ln in spans:
            buf.extend(int(x) for x in pkt[off:off + ln])
        if checksum(init, poly, buf) != residue:
            return False
    return True


def filter_hits(proto: str, blocks: np.ndarray, pkts: np.ndarray) -> np.ndarray:
    """Indices of the hits of one preamble (sorted by (block, idx)) that survive the GPU-side validation as
    amrdemod.h defines it: every check passes, and the parser's bytes differ from those of the hit right before
    it in the same block.  (Parse's `seen` drops every repeat inside a block; the device drops the adjacent ones.)"""
    nkeep, _ = RULES[proto]
    keep = []
    for i in range(len(blocks)):
        if not passes(proto, pkts[i]):
            continue
        if i > 0 and blocks[i - 1] == blocks[i] and np.array_equal(pkts[i - 1, :nkeep], pkts[i, :nkeep]):
            continue
        keep.append(i)
    return np.asarray(keep, dtype=np.int64)


# ---- the same rule, vectorised over all hits of a preamble (full-size goldens: 300 000 hits per GiB) -----------------

def checksum_np(init: int, poly: int, data: np.ndarray) -> np.ndarray:
    """crc.Checksum (crc/crc.go:49-55) of every row of data uint8[n, k] at once: the bit steps of `checksum` above,
    each applied to all rows."""
    crc = np.full(len(data), init, np.uint32)
    for j in range(data.shape[1]):
        crc ^= data[:, j].astype(np.uint32) << 8
        for _ in range(8):
            top = (crc & 0x8000) != 0
            crc = (crc << 1) & 0xFFFF
            crc[top] ^= poly
    return crc


def filter_hits_np(proto: str, blocks: np.ndarray, pkts: np.ndarray) -> np.ndarray:
    """filter_hits for arrays: identical result (tests/test_validate_cpu.py compares the two)."""
    nkeep, checks = RULES[proto]
    ok = np.ones(len(blocks), bool)
    for (init, poly, residue), spans in checks:
        buf = np.concatenate([pkts[:, off:off + ln] for off, ln in spans], axis=1)
        ok &= checksum_np(init, poly, buf) == residue
    # like filter_hits: a hit is a repeat of the hit right before it in the FULL list, whether or not that one passed
    rep = np.zeros(len(blocks), bool)
    if len(blocks) > 1:
        rep[1:] = (blocks[1:] == blocks[:-1]) & (pkts[1:, :nkeep] == pkts[:-1, :nkeep]).all(axis=1)
    return np.flatnonzero(ok & ~rep).astype(np.int64)
