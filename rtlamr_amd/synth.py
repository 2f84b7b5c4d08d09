"""Deterministic integer-only synthetic IQ -- numpy twin of csrc/synth.h.

    h = splitmix64(seed ^ n);  I = 119 + popcount(h & 0xFFFF);  Q = 120 + popcount((h >> 16) & 0xFFFF)
Noise is binomial around 127/128 with sigma 2, like the noise floor of the reference capture
assets/sample.bin.  Packets are Manchester-OOK bursts: bit 1 = chip high then low, bit 0 = low then
high (the sign convention of Decoder.Filter, decode.go:239-244); "high" adds (dI, dQ) with clamping.
Bench and test support only -- not part of the decode path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Sequence

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _popcount16(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.uint32)
    v = v - ((v >> 1) & 0x5555)
    v = (v & 0x3333) + ((v >> 2) & 0x3333)
    v = (v + (v >> 4)) & 0x0F0F
    return (v + (v >> 8)) & 0x1F


def noise(n_samples: int, seed: int, first_sample: int = 0) -> np.ndarray:
    """uint8[2*n_samples] interleaved I,Q."""
    n = np.arange(first_sample, first_sample + n_samples, dtype=np.uint64)
    h = splitmix64(np.uint64(seed) ^ n)
    out = np.empty(2 * n_samples, np.uint8)
    out[0::2] = 119 + _popcount16(h & np.uint64(0xFFFF))
    out[1::2] = 120 + _popcount16((h >> np.uint64(16)) & np.uint64(0xFFFF))
    return out


def uniform(n_samples: int, seed: int, first_sample: int = 0) -> np.ndarray:
    """uint8[2*n_samples] interleaved I,Q of the second distribution of SURVEY.md 8d: uniform random bytes (I = bits
    32..39, Q = bits 40..47 of the hash); twin of k_synth_noise<true> and orc_synth_uniform."""
    n = np.arange(first_sample, first_sample + n_samples, dtype=np.uint64)
    h = splitmix64(np.uint64(seed) ^ n)
    out = np.empty(2 * n_samples, np.uint8)
    out[0::2] = ((h >> np.uint64(32)) & np.uint64(0xFF)).astype(np.uint8)
    out[1::2] = ((h >> np.uint64(40)) & np.uint64(0xFF)).astype(np.uint8)
    return out


@dataclass
class Packet:
    start: int      # stream sample index of the first chip
    data: bytes     # packet bytes, MSB first
    n_bits: int
    d_i: int
    d_q: int


def plant(iq: np.ndarray, packets: Sequence[Packet], chip_length: int, first_sample: int = 0) -> None:
    """In-place burst planting on a host buffer (twin of k_synth_plant)."""
    n_samples = iq.size // 2
    sl = 2 * chip_length
    for p in packets:
        bits = np.unpackbits(np.frombuffer(p.data, np.uint8))[: p.n_bits]
        # high-sample mask of the whole packet
        sym = np.repeat(bits, sl).astype(bool)
        first_chip = (np.arange(p.n_bits * sl) % sl) < chip_length
        high = sym == first_chip
        pos = p.start + np.flatnonzero(high) - first_sample
        pos = pos[(pos >= 0) & (pos < n_samples)]
        iq[2 * pos] = np.clip(iq[2 * pos].astype(np.int32) + p.d_i, 0, 255).astype(np.uint8)
        iq[2 * pos + 1] = np.clip(iq[2 * pos + 1].astype(np.int32) + p.d_q, 0, 255).astype(np.uint8)


def packet_arrays(packets: Sequence[Packet]):
    """Flatten packets for amr_synth_plant: (start u64[n], bits u8[n*stride], n_bits, stride, dI i8[n], dQ i8[n])."""
    n_bits = packets[0].n_bits
    stride = (n_bits + 7) // 8
    assert all(p.n_bits == n_bits for p in packets)
    start = np.array([p.start for p in packets], np.uint64)
    bits = np.zeros((len(packets), stride), np.uint8)
    for i, p in enumerate(packets):
        bits[i, : len(p.data[:stride])] = np.frombuffer(p.data[:stride], np.uint8)
    di = np.array([p.d_i for p in packets], np.int8)
    dq = np.array([p.d_q for p in packets], np.int8)
    return start, bits.reshape(-1), n_bits, stride, di, dq


def packet_schedule(n_packets: int, n_samples: int, packet_samples: int, seed: int, edge_every: int = 0,
                    block_size: int = 0) -> np.ndarray:
    """Non-overlapping hash-chosen packet start offsets; every `edge_every`-th packet is moved so that it
    straddles a block boundary (start = multiple of block_size - packet_samples/2)."""
    stride = n_samples // n_packets
    assert stride > packet_samples + 64, "packets would overlap"
    j = np.arange(n_packets, dtype=np.uint64)
    jitter = splitmix64(np.uint64(seed) ^ (j + np.uint64(0xABCDEF))) % np.uint64(stride - packet_samples - 32)
    start = (j * np.uint64(stride) + jitter).astype(np.int64)
    if edge_every and block_size:
        for i in range(0, n_packets, edge_every):
            b = (start[i] + packet_samples // 2 + block_size - 1) // block_size * block_size
            s = b - packet_samples // 2
            if s >= i * stride and s + packet_samples < (i + 1) * stride:
                start[i] = s
    return start


def device_fill(device_id: int, d_ptr: int, n_samples: int, seed: int, first_sample: int,
                packets: Sequence[Packet], chip_length: int, uniform_bytes: bool = False) -> None:
    """Noise (or uniform random bytes) + packets directly in device memory (K0)."""
    from . import _lib
    L = _lib.lib()
    fill = L.amr_synth_uniform if uniform_bytes else L.amr_synth_noise
    _lib.check(fill(device_id, C.c_void_p(d_ptr), n_samples, seed, first_sample), "amr_synth_noise")
    by_len = {}
    for p in packets:                      # one plant call per packet length (scm 96, scm+ 128, idm 736 bits)
        by_len.setdefault(p.n_bits, []).append(p)
    for group in by_len.values():
        start, bits, n_bits, stride, di, dq = packet_arrays(group)
        _lib.check(L.amr_synth_plant(device_id, C.c_void_p(d_ptr), n_samples, first_sample, chip_length, len(group),
                                     start.ctypes.data, bits.ctypes.data, n_bits, stride, di.ctypes.data,
                                     dq.ctypes.data), "amr_synth_plant")


def plant_chips(iq: np.ndarray, start: int, chips: Sequence[int], chip_length: int, d_i: int, d_q: int,
                first_sample: int = 0) -> None:
    """OOK burst given chip by chip (1 = carrier on for chip_length samples): what plant() does for Manchester
    bits, for symbol alphabets that are not Manchester (r900's six 4-chip symbols, r900.go:104-110)."""
    n_samples = iq.size // 2
    high = np.repeat(np.asarray(chips, bool), chip_length)
    pos = start + np.flatnonzero(high) - first_sample
    pos = pos[(pos >= 0) & (pos < n_samples)]
    iq[2 * pos] = np.clip(iq[2 * pos].astype(np.int32) + d_i, 0, 255).astype(np.uint8)
    iq[2 * pos + 1] = np.clip(iq[2 * pos + 1].astype(np.int32) + d_q, 0, 255).astype(np.uint8)


def r900_chips(preamble: str, symbols: Sequence[int]) -> List[int]:
    """Chips of one r900 burst: Manchester preamble (bit 1 = high,low) followed by the 6-ary payload symbols."""
    from .contrib.parsers.r900 import symbols_to_chips
    chips: List[int] = []
    for b in preamble:
        chips.extend((1, 0) if b == "1" else (0, 1))
    return chips + symbols_to_chips(symbols)
