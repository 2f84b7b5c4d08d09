"""numpy restatement of the r900 parser's second stage (r900/r900.go:82-150, 160-193).

TEST INFRASTRUCTURE ONLY, like everything under oracle/.  Parity status: UNPINNED by the reference (no test or
vector exists for r900 either); what pins this file is (i) a pure-Python float32 loop over the same lines on small
inputs (tests/test_r900_cpu.py), (ii) recovery of planted Reed-Solomon-valid packets.

`digits_for_stream` replays Parser.Parse's buffer handling call by call on top of the C oracle's Decoder
(oracle.decode_oracle.c): slide p.signal by BlockSize, append Decoder.Signal[SymbolLength:] (r900.go:168-170),
running float32 sum from zero over the whole buffer (r900.go:96-100; numpy's cumsum is the same sequential
accumulation), then for every r900 preamble hit the 42 quantized symbols at payloadIdx + j*4*ChipLength
(r900.go:183-193) with the a0/a1/a2 arithmetic of r900.go:119-148 in float32, operation for operation.
"""
from __future__ import annotations

import numpy as np

from .oracle import OracleDecoder, PROTOCOLS

F = np.float32
PAYLOAD_SYMBOLS = 42


def quantize_at(csum: np.ndarray, q: np.ndarray, cl: int) -> np.ndarray:
    """r900.go:119-148 at the positions q (vectorised over q, every operation in float32)."""
    c0 = csum[q]
    c1 = csum[q + cl] + csum[q + cl]
    c2 = csum[q + 2 * cl] + csum[q + 2 * cl]
    c3 = csum[q + 3 * cl] + csum[q + 3 * cl]
    c4 = csum[q + 4 * cl]
    a0 = (c2 - c4) - c0                      # 1100
    a1 = (((c1 - c2) + c3) - c4) - c0        # 1010
    a2 = ((c1 - c3) + c4) - c0               # 1001
    a = np.stack([a0, a1, a2])
    max_abs = np.abs(a0)
    arg = np.zeros(len(q), np.int64)
    m1 = np.abs(a1) > max_abs
    max_abs = np.where(m1, np.abs(a1), max_abs)
    arg[m1] = 1
    m2 = np.abs(a2) > max_abs
    arg[m2] = 2
    val = a[arg, np.arange(len(q))]
    return (arg + 3 * (val > 0)).astype(np.uint8)


def digits_for_stream(protocols, chip_length: int, iq: np.ndarray):
    """-> (hits int64[n,2] (call, idx) of the r900 preamble in Search order, digits uint8[n,42])."""
    o = OracleDecoder(list(protocols), chip_length)
    g = o.geom
    names = [p for p in protocols]
    pid = o.preamble_ids[names.index("r900")]
    bs, pl, sl, cl = g.block_size, g.packet_length, g.symbol_length, g.chip_length
    signal = np.zeros(g.buffer_length, F)             # r900.go:163
    hits, digits = [], []
    n_blocks = iq.size // g.block_size2
    for k in range(n_blocks):
        res = o.decode(iq[k * g.block_size2:(k + 1) * g.block_size2])
        signal[:-bs] = signal[bs:].copy()             # r900.go:168
        signal[pl:] = o.signal[sl:]                   # r900.go:169-170
        idxs = res[pid][0]
        if len(idxs) == 0:
            continue
        csum = np.concatenate([np.zeros(1, F), np.cumsum(signal, dtype=F)])   # r900.go:96-100
        for idx in idxs:
            payload = int(idx) + g.preamble_length - sl                        # r900.go:183
            q = payload + np.arange(PAYLOAD_SYMBOLS) * 4 * cl                  # r900.go:187-189
            hits.append((k, int(idx)))
            digits.append(quantize_at(csum, q, cl))
    if not hits:
        return np.zeros((0, 2), np.int64), np.zeros((0, PAYLOAD_SYMBOLS), np.uint8)
    return np.array(hits, np.int64), np.stack(digits)


def quantize_literal(signal, cl: int, positions):
    """Pure-Python float32 restatement of r900.go:96-148 (slow; pins the vectorised version on small inputs)."""
    csum = [F(0)]
    s = F(0)
    for v in signal:
        s = F(s + F(v))
        csum.append(s)
    out = []
    for idx in positions:
        c0 = csum[idx]
        c1 = F(csum[idx + cl] + csum[idx + cl])
        c2 = F(csum[idx + 2 * cl] + csum[idx + 2 * cl])
        c3 = F(csum[idx + 3 * cl] + csum[idx + 3 * cl])
        c4 = csum[idx + 4 * cl]
        a0 = F(F(c2 - c4) - c0)
        a1 = F(F(F(F(c1 - c2) + c3) - c4) - c0)
        a2 = F(F(F(c1 - c3) + c4) - c0)
        max_abs, arg = abs(a0), 0
        if abs(a1) > max_abs:
            max_abs, arg = abs(a1), 1
        if abs(a2) > max_abs:
            max_abs, arg = abs(a2), 2
        if (a0, a1, a2)[arg] > 0:
            arg += 3
        out.append(arg)
    return np.array(out, np.uint8)
