"""Independent numpy restatement of protocol/decode.go in STREAM-GLOBAL form.

TEST INFRASTRUCTURE ONLY (same rules as decode_oracle.c).  Written from
SURVEY.md section 8a's "semantics in stream-global form" rather than from the
per-call buffer shuffling of the Go code, so that agreement with
decode_oracle.c (which follows the Go control flow literally) is evidence that
both are right.  Parity status: unpinned by the reference (see oracle.py).

    mag[n] = lut[I_n] + lut[Q_n]  (n >= 0), 0.0 for n < 0        decode.go:219-225,144
    block k: c[0] = 0; c[j+1] = fl32(c[j] + mag[k*BS - SL + j])    decode.go:232-236
    q[k*BS+i] = 1 - signbit(fl32(fl32(c[i+CL]-c[i]) - fl32(c[i+SL]-c[i+CL])))   :239-244
    q[n] = 0 for n < 0                                             decode.go:145
    call k, preamble P: { idx in [0,BS) : q[k*BS - PL + idx + p*SL] == P[p] for all p }  :255-328
    packet bit p of a hit = q[k*BS - PL + idx + p*SL], MSB first    decode.go:353-375
"""
from __future__ import annotations

import math

import numpy as np


def mag_lut() -> np.ndarray:
    """decode.go:209-216: float32 division, then float32 square (two roundings)."""
    i = np.arange(256, dtype=np.float32)
    q = (np.float32(127.5) - i) / np.float32(127.5)
    assert q.dtype == np.float32
    return (q * q).astype(np.float32)


def geometry(chip_length: int, preamble_symbols: int, packet_symbols: int) -> dict:
    """decode.go:131-141 with the max-merged symbols of decode.go:105-109."""
    sl = chip_length * 2
    prel = preamble_symbols * sl
    pl = packet_symbols * sl
    bs = 1 << int(math.ceil(math.log2(float(prel))))
    return dict(CL=chip_length, SL=sl, PreL=prel, PL=pl, BS=bs, BS2=2 * bs, BufLen=pl + bs)


def quantize_stream(iq: np.ndarray, chip_length: int, block_size: int) -> np.ndarray:
    """q[n] for n in [0, n_blocks*BS) as uint8 0/1, whole blocks only."""
    lut = mag_lut()
    iq = np.asarray(iq, dtype=np.uint8)
    cl, sl, bs = chip_length, 2 * chip_length, block_size
    n_blocks = iq.size // (2 * bs)
    mag = lut[iq[0:2 * n_blocks * bs:2]] + lut[iq[1:2 * n_blocks * bs:2]]
    assert mag.dtype == np.float32
    mag = np.concatenate([np.zeros(sl, np.float32), mag])  # zero history before the stream
    q = np.empty(n_blocks * bs, np.uint8)
    for k in range(n_blocks):
        seg = mag[k * bs: k * bs + bs + sl]
        # np.add.accumulate on float32 is a strictly sequential float32 running sum
        c = np.concatenate([np.zeros(1, np.float32), np.add.accumulate(seg, dtype=np.float32)])
        lower = c[cl: cl + bs]
        f = (lower - c[0:bs]) - (c[sl: sl + bs] - lower)
        q[k * bs: (k + 1) * bs] = 1 - (f.view(np.uint32) >> 31).astype(np.uint8)
    return q


def search_stream(q: np.ndarray, preamble: str, g: dict):
    """All (block, idx) with the preamble present, (block, idx) ascending."""
    bs, sl, pl = g["BS"], g["SL"], g["PL"]
    n = q.size
    qq = np.concatenate([np.zeros(pl, np.uint8), q])  # q[n]=0 for n<0; qq[j] = q[j-pl]
    # position g_pos = k*BS - PL + idx  <->  qq index k*BS + idx = any j in [0, n)
    ok = np.ones(n, bool)
    for p, ch in enumerate(preamble):
        ok &= qq[p * sl: p * sl + n] == (1 if ch == "1" else 0)
    j = np.nonzero(ok)[0]
    return np.stack([j // bs, j % bs], axis=1).astype(np.int64), qq


def slice_packets(qq: np.ndarray, hits: np.ndarray, g: dict, packet_symbols: int) -> np.ndarray:
    """Packet bytes per hit (clean last byte: high bits zero when PacketSymbols%8 != 0)."""
    nb = (packet_symbols + 7) >> 3
    out = np.zeros((len(hits), nb), np.uint8)
    for h, (k, idx) in enumerate(hits):
        base = int(k) * g["BS"] + int(idx)
        bits = qq[base: base + packet_symbols * g["SL"]: g["SL"]]
        for p, b in enumerate(bits):
            out[h, p >> 3] = ((int(out[h, p >> 3]) << 1) | int(b)) & 0xFF
    return out


def stale_last_bytes(pkts: np.ndarray, rows: np.ndarray, packet_symbols: int) -> np.ndarray:
    """The bits Decoder.Slice never clears (decode.go:353-375): `pkt` is shifted per symbol and lives as long as the Decoder,
    so with r = PacketSymbols % 8 != 0 the last byte of hit j is  B(j) = (B(j-1) << r | fresh(j)) & 0xff  over the hits in
    the order they are sliced -- call, preamble id (registration order: the literal C restatement's order), idx.
    pkts: the packets with CLEAN last bytes (slice_packets), rows: int64[n,3] (preamble id, call, idx) in any order.
    -> a copy of pkts with the last bytes as the Decoder leaves them."""
    r = packet_symbols % 8
    out = pkts.copy()
    if r == 0 or not len(pkts):
        return out
    order = np.lexsort((rows[:, 2], rows[:, 0], rows[:, 1]))       # call, then preamble id, then idx
    b = 0
    for j in order:
        b = ((b << r) | int(pkts[j, -1])) & 0xFF
        out[j, -1] = b
    return out
