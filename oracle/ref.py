"""ctypes front-end of oracle/_ref/libref.so: the reference's OWN Go sources (protocol/, scm/, scmplus/, idm/, netidm/,
r900/, r900/gf/, r900bcd/, crc/), translated mechanically to C++ by oracle/go2cxx and built by `make -C oracle _ref`.

TEST INFRASTRUCTURE ONLY.  Loaded by tests/test_ref_translated.py to hold oracle/decode_oracle.c (and through it the
HIP path) to code that DESCENDS FROM THE REFERENCE SOURCE TEXT rather than from a reading of it.  Never imported by
rtlamr_amd/, never inside bench.py's timed region.  oracle/_ref/ is git-ignored; libref.so is built in the container
that has /root/reference and travels to the GPU box as a prebuilt file like the other .so files.

This is not the Go toolchain: go2cxx is a syntax-directed translator with a small run time (go2cxx/README.md says what
it does and does not guarantee).  Parity statements in DESIGN.md therefore say "pinned to a mechanical translation of
the reference source", not "pinned to the Go binary".
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libref.so")
REFERENCE = os.environ.get("AMR_REFERENCE", "/root/reference")

GEOM_FIELDS = ("data_rate", "chip_length", "symbol_length", "sample_rate", "preamble_symbols", "packet_symbols",
               "preamble_length", "packet_length", "block_size", "block_size2", "buffer_length", "n_preambles",
               "center_freq", "signal_len", "quantized_len", "pkt_bytes")


def available() -> bool:
    return os.path.exists(_SO) or os.path.isdir(os.path.join(REFERENCE, "protocol"))


def build(force: bool = False) -> Optional[str]:
    """(Re)build oracle/_ref/libref.so when the reference tree is present; otherwise use the prebuilt file if any."""
    if os.path.isdir(os.path.join(REFERENCE, "protocol")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_ref", f"REF={REFERENCE}"] + (["-B"] if force else []))
    return _SO if os.path.exists(_SO) else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        if build() is None:
            raise FileNotFoundError("oracle/_ref/libref.so: not built and no reference tree to build it from")
        L = C.CDLL(_SO)
        L.ref_new.restype = C.c_void_p
        L.ref_new.argtypes = [C.c_char_p, C.c_int]
        L.ref_free.argtypes = [C.c_void_p]
        L.ref_geometry.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_decode.restype = C.c_int64
        L.ref_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.ref_copy_quantized.restype = C.c_int64
        L.ref_copy_quantized.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_copy_signal.restype = C.c_int64
        L.ref_copy_signal.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_spy_preamble.restype = C.c_int64
        L.ref_spy_preamble.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int64]
        L.ref_spy_hits.restype = C.c_int64
        L.ref_spy_hits.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.ref_spy_bits.restype = C.c_int64
        L.ref_spy_bits.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_char_p, C.c_int64]
        L.ref_decode_stream.restype = C.c_int64
        L.ref_decode_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_char_p, C.c_int64, C.c_void_p]
        L.ref_messages.restype = C.c_int64
        L.ref_messages.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.ref_r900_quantized.restype = C.c_int64
        L.ref_r900_quantized.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.ref_r900_signal.restype = C.c_int64
        L.ref_r900_signal.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.ref_lut.argtypes = [C.c_void_p]
        L.ref_next_power_of_2.restype = C.c_int64
        L.ref_next_power_of_2.argtypes = [C.c_int64]
        L.ref_crc.restype = C.c_uint32
        L.ref_crc.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_int64]
        _lib = L
    return _lib


def mag_lut() -> np.ndarray:
    out = np.zeros(256, np.float32)
    lib().ref_lut(out.ctypes.data)
    return out


def next_power_of_2(v: int) -> int:
    return int(lib().ref_next_power_of_2(v))


def crc16(init: int, poly: int, data: bytes) -> int:
    buf = np.frombuffer(bytes(data), np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
    return int(lib().ref_crc(init, poly, buf.ctypes.data, len(data)))


def parse_messages(text: str) -> List[Tuple[int, str, int, int, str, List[str]]]:
    """Lines "call|MsgType|MeterID|MeterType|checksum hex|Record() joined by ','" -> tuples."""
    out = []
    for line in text.splitlines():
        call, typ, mid, mtype, chk, rec = line.split("|", 5)
        out.append((int(call), typ, int(mid), int(mtype), chk, rec.split(",") if rec else []))
    return out


class RefDecoder:
    """protocol.NewDecoder + <pkg>.NewParser per protocol + RegisterProtocol + Allocate, all translated reference code.
    protocols: names among scm, scm+, idm, netidm, r900, r900bcd (the reference's own parser packages supply their configs)."""

    def __init__(self, protocols, chip_length: int):
        L = lib()
        h = L.ref_new(",".join(protocols).encode(), chip_length)
        if not h:
            raise ValueError(f"unknown protocol in {protocols}")
        self._h = C.c_void_p(h)
        g = (C.c_int64 * 16)()
        L.ref_geometry(self._h, g)
        self.geom = dict(zip(GEOM_FIELDS, [int(x) for x in g]))
        self.pkt_bytes = self.geom["pkt_bytes"]
        self.preambles = []
        for i in range(self.geom["n_preambles"]):
            buf = C.create_string_buffer(256)
            L.ref_spy_preamble(self._h, i, buf, 256)
            self.preambles.append(buf.value.decode())

    def __del__(self):
        try:
            lib().ref_free(self._h)
        except Exception:
            pass

    def decode(self, block: np.ndarray):
        """ONE Decoder.Decode call.  -> ([(idx int64[n], bytes uint8[n, pkt_bytes])] per preamble, messages)"""
        L = lib()
        block = np.ascontiguousarray(block, np.uint8)
        L.ref_decode(self._h, block.ctypes.data, block.size)
        out = []
        cap = self.geom["block_size"]
        for i in range(self.geom["n_preambles"]):
            idx = np.zeros(cap, np.int64)
            pb = np.zeros((cap, self.pkt_bytes), np.uint8)
            n = L.ref_spy_hits(self._h, i, idx.ctypes.data, pb.ctypes.data, cap)
            assert 0 <= n <= cap
            out.append((idx[:n].copy(), pb[:n].copy()))
        buf = C.create_string_buffer(1 << 20)
        n = L.ref_messages(self._h, buf, 1 << 20)
        assert n >= 0
        return out, parse_messages(buf.value.decode())

    def hit_bits(self, preamble: int, k: int) -> str:
        buf = C.create_string_buffer(4096)
        n = lib().ref_spy_bits(self._h, preamble, k, buf, 4096)
        assert n >= 0
        return buf.value.decode()

    @property
    def quantized(self) -> np.ndarray:
        out = np.zeros(self.geom["quantized_len"], np.uint8)
        lib().ref_copy_quantized(self._h, out.ctypes.data)
        return out

    @property
    def signal(self) -> np.ndarray:
        out = np.zeros(self.geom["signal_len"], np.float32)
        lib().ref_copy_signal(self._h, out.ctypes.data)
        return out

    def r900_quantized(self) -> np.ndarray:
        """r900.Parser.quantized after the last call (r900.go:50, written by filter() r900.go:118-149)."""
        out = np.zeros(self.geom["buffer_length"], np.uint8)
        n = lib().ref_r900_quantized(self._h, out.ctypes.data, out.size)
        if n < 0:
            raise ValueError("no r900 parser registered (or its buffers are not allocated yet)")
        return out[:n]

    def r900_signal(self) -> np.ndarray:
        out = np.zeros(self.geom["buffer_length"], np.float32)
        n = lib().ref_r900_signal(self._h, out.ctypes.data, out.size)
        if n < 0:
            raise ValueError("no r900 parser registered (or its buffers are not allocated yet)")
        return out[:n]

    def decode_stream(self, iq: np.ndarray, hits_cap: int = 1 << 18):
        """n consecutive Decode calls -> (qpacked, hits int32[n,3] rows (block, preamble, idx), bytes, messages):
        the same shapes and order as oracle.OracleDecoder.decode_stream."""
        g = self.geom
        iq = np.ascontiguousarray(iq, np.uint8)
        n_blocks = iq.size // g["block_size2"]
        q = np.zeros(n_blocks * g["block_size"] // 8, np.uint8)
        hits = np.zeros((hits_cap, 3), np.int32)
        hb = np.zeros((hits_cap, self.pkt_bytes), np.uint8)
        mcap = 1 << 22
        msgs = C.create_string_buffer(mcap)
        mlen = C.c_int64()
        total = lib().ref_decode_stream(self._h, iq.ctypes.data, n_blocks, q.ctypes.data, hits.ctypes.data, hb.ctypes.data,
                                        hits_cap, msgs, mcap, C.byref(mlen))
        if total > hits_cap or mlen.value + 1 > mcap:
            raise OverflowError(f"{total} hits / {mlen.value} message bytes exceed the buffers")
        return q, hits[:total].copy(), hb[:total].copy(), parse_messages(msgs.value.decode())
