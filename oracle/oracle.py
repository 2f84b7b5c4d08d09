"""ctypes front-end of oracle/liboracle.so (decode_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of decode_oracle.c.  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package rtlamr_amd.

Parity status: UNPINNED by the reference's own tests (there are none for
protocol/decode.go); pinned against SURVEY.md section 8c derived vectors and the
numpy restatement in np_oracle.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

# Protocol table restated from the reference parsers' NewParser configs:
#   scm   scm/scm.go:39-53          scm+  scmplus/scmplus.go:46-60
#   idm   idm/idm.go:46-60          netidm netidm/netidm.go:57-71
#   r900  r900/r900.go:54-71
# (preamble, PreambleSymbols, PacketSymbols); DataRate is 32768 for all.
PROTOCOLS = {
    "scm": ("111110010101001100000", 21, 96),
    "scm+": ("0001011010100011", 16, 128),
    "idm": ("01010101010101010001011010100011", 32, 92 * 8),
    "netidm": ("01010101010101010001011010100011", 32, 92 * 8),
    "r900": ("00000000000000001110010101100100", 32, 116),
}
DATA_RATE = 32768


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile (gcc, seconds)."""
    src = os.path.join(_HERE, "decode_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_new.restype = C.c_void_p
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_register.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_register.restype = C.c_int
        L.orc_allocate.argtypes = [C.c_void_p]
        L.orc_allocate.restype = C.c_int
        L.orc_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.orc_signal.argtypes = [C.c_void_p]
        L.orc_signal.restype = C.POINTER(C.c_float)
        L.orc_quantized.argtypes = [C.c_void_p]
        L.orc_quantized.restype = C.POINTER(C.c_uint8)
        L.orc_lut.argtypes = [C.c_void_p]
        L.orc_lut.restype = C.POINTER(C.c_float)
        L.orc_pkt_bytes.argtypes = [C.c_void_p]
        L.orc_pkt_bytes.restype = C.c_int
        L.orc_next_power_of_2.argtypes = [C.c_int]
        L.orc_next_power_of_2.restype = C.c_int
        L.orc_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_decode.restype = C.c_int
        L.orc_decode_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_long]
        L.orc_decode_stream.restype = C.c_long
        L.orc_r900_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_r900_filter.restype = None
        _lib = L
    return _lib


@dataclass
class Geometry:
    data_rate: int
    chip_length: int
    symbol_length: int
    sample_rate: int
    preamble_symbols: int
    packet_symbols: int
    preamble_length: int
    packet_length: int
    block_size: int
    block_size2: int
    buffer_length: int
    n_preambles: int


class OracleDecoder:
    """NewDecoder + RegisterProtocol* + Allocate (decode.go:65,100,131)."""

    def __init__(self, protocols, chip_length: int):
        """protocols: names from PROTOCOLS or (preamble, preamble_symbols, packet_symbols) tuples."""
        L = lib()
        self._h = C.c_void_p(L.orc_new())
        self.preamble_ids = []
        for p in protocols:
            pre, ps, ks = PROTOCOLS[p] if isinstance(p, str) else p
            pid = L.orc_register(self._h, pre.encode(), DATA_RATE, chip_length, ps, ks)
            if pid < 0:
                raise ValueError("orc_register failed")
            self.preamble_ids.append(pid)
        if L.orc_allocate(self._h) != 0:
            raise ValueError("orc_allocate failed")
        g = (C.c_int * 12)()
        L.orc_geometry(self._h, g)
        self.geom = Geometry(*list(g))
        self.pkt_bytes = L.orc_pkt_bytes(self._h)

    def __del__(self):
        try:
            lib().orc_free(self._h)
        except Exception:
            pass

    @property
    def signal(self) -> np.ndarray:
        n = self.geom.block_size + self.geom.symbol_length
        return np.ctypeslib.as_array(lib().orc_signal(self._h), shape=(n,)).copy()

    @property
    def quantized(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib().orc_quantized(self._h), shape=(self.geom.buffer_length,)).copy()

    @property
    def lut(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib().orc_lut(self._h), shape=(256,)).copy()

    def decode(self, block: np.ndarray, mode: int = 0):
        """One Decode call (decode.go:163).  Returns [(idx array, bytes array[n, pkt_bytes])] per preamble id."""
        g = self.geom
        block = np.ascontiguousarray(block, dtype=np.uint8)
        if block.size < g.block_size2:
            raise IndexError("short input (Go would panic: index out of range, decode.go:222)")
        cap = g.block_size
        npre = g.n_preambles
        cnt = np.zeros(npre, np.int32)
        idx = np.zeros((npre, cap), np.int32)
        pb = np.zeros((npre, cap, self.pkt_bytes), np.uint8)
        lib().orc_decode(self._h, block.ctypes.data, mode, cap, cnt.ctypes.data, idx.ctypes.data, pb.ctypes.data)
        return [(idx[p, : cnt[p]].copy(), pb[p, : cnt[p]].copy()) for p in range(npre)]

    def decode_stream(self, iq: np.ndarray, mode: int = 0, hits_cap: int = 1 << 20, want_q: bool = True):
        """n consecutive Decode calls.  Returns (qpacked bytes or None, hits[n,3] (block,pid,idx), bytes[n,pkt_bytes])."""
        g = self.geom
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        n_blocks = iq.size // g.block_size2
        q = np.zeros(n_blocks * g.block_size // 8, np.uint8) if want_q else None
        hits = np.zeros((hits_cap, 3), np.int32)
        hb = np.zeros((hits_cap, self.pkt_bytes), np.uint8)
        total = lib().orc_decode_stream(self._h, iq.ctypes.data, n_blocks, mode,
                                        q.ctypes.data if want_q else None,
                                        hits.ctypes.data, hb.ctypes.data, hits_cap)
        if total > hits_cap:
            raise OverflowError(f"{total} hits > cap {hits_cap}")
        return q, hits[:total].copy(), hb[:total].copy()


class R900Filter:
    """r900.Parser's buffers + filter (r900.go:82-150, 160-170), literal C restatement (decode_oracle.c)."""

    def __init__(self, dec: OracleDecoder):
        self.dec = dec
        bl = dec.geom.buffer_length
        self.signal = np.zeros(bl, np.float32)
        self.csum = np.zeros(bl + 1, np.float32)
        self.quantized = np.zeros(bl, np.uint8)

    def step(self) -> np.ndarray:
        """Call after every OracleDecoder.decode (Parse runs once per Decode call); returns p.quantized."""
        lib().orc_r900_filter(self.dec._h, self.signal.ctypes.data, self.csum.ctypes.data, self.quantized.ctypes.data)
        return self.quantized


def next_power_of_2(v: int) -> int:
    return lib().orc_next_power_of_2(v)
