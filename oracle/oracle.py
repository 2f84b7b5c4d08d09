"""ctypes front-end of oracle/liboracle.so (decode_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of decode_oracle.c.  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package rtlamr_amd.

Parity status: UNPINNED by the reference's own tests (there are none for
protocol/decode.go); pinned against oracle/_ref (the reference's Go sources,
translated mechanically: oracle/go2cxx, tests/test_ref_translated.py), SURVEY.md
section 8c derived vectors and the numpy restatement in np_oracle.py.
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
    srcs = [os.path.join(_HERE, f) for f in ("decode_oracle.c", "synth_gen.c", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
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
        L.orc_synth_noise.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_synth_noise.restype = None
        L.orc_synth_uniform.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_synth_uniform.restype = None
        L.orc_synth_plant.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_char_p, C.c_uint32,
                                      C.c_int, C.c_int]
        L.orc_synth_plant.restype = None
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

    def decode_stream(self, iq: np.ndarray, mode: int = 0, hits_cap: int = 1 << 20, want_q: bool = True,
                      allow_overflow: bool = False):
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
            if allow_overflow:
                total = hits_cap
            else:
                raise OverflowError(f"{total} hits > cap {hits_cap}", int(total))
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


# ---- full-size streams on the host: threaded generator + sharded decode (tests, goldens) ---------------------------

def _run_threads(jobs, n_threads):
    """Run callables on up to n_threads Python threads (the ctypes calls inside release the GIL)."""
    import threading
    it = iter(jobs)
    lock = threading.Lock()
    err = []

    def work():
        while True:
            with lock:
                job = next(it, None)
            if job is None or err:
                return
            try:
                job()
            except BaseException as e:   # noqa: BLE001 -- re-raised by the caller's thread
                err.append(e)
    th = [threading.Thread(target=work) for _ in range(max(1, n_threads))]
    [t.start() for t in th]
    [t.join() for t in th]
    if err:
        raise err[0]


def default_threads() -> int:
    return max(1, min(64, os.cpu_count() or 1))


def synth_stream(n_samples: int, seed: int, first_sample: int, packets, chip_length: int, n_threads: int = 0,
                 uniform: bool = False) -> np.ndarray:
    """uint8[2*n_samples]: SURVEY 8d noise (uniform: its second distribution, uniform random bytes) + planted packets
    (objects with start, data, n_bits, d_i, d_q), made by synth_gen.c on n_threads threads.  Byte-identical to
    rtlamr_amd.synth.noise / uniform + plant and to the device generator."""
    L = lib()
    gen = L.orc_synth_uniform if uniform else L.orc_synth_noise
    out = np.empty(2 * n_samples, np.uint8)
    nt = n_threads or default_threads()
    step = max(1 << 20, -(-n_samples // (4 * nt)) & ~7)
    jobs = []
    for s0 in range(0, n_samples, step):
        n = min(step, n_samples - s0)
        jobs.append(lambda s0=s0, n=n: gen(out.ctypes.data + 2 * s0, n, seed, first_sample + s0))
    _run_threads(jobs, nt)
    for p in packets:       # packets never overlap: order does not matter; each is a few thousand samples
        L.orc_synth_plant(out.ctypes.data, n_samples, first_sample, chip_length, int(p.start), bytes(p.data),
                          int(p.n_bits), int(p.d_i), int(p.d_q))
    return out


def decode_sharded(protocols, chip_length: int, iq: np.ndarray, n_threads: int = 0, first_block: int = 0,
                   want_q: bool = True, mode: int = 0):
    """The whole stream `iq` through ONE logical reference Decoder (n consecutive Decode calls from the zero state),
    computed on n_threads threads: thread i owns a contiguous block range and first replays the
    ceil(PacketLength/BlockSize)+1 blocks before it with the hits discarded, after which its magnitude and quantized
    histories equal the single decoder's (decode.go:165-166; the first replayed block sees a zero magnitude history
    and may come out different, every later one is exact, and the search needs ceil(PL/BS) exact blocks).
    Calls before `first_block` are run for their state only (no hits, no q).
    -> (qpacked uint8 or None, hits int64[n,3] rows (pid, block, idx) sorted preamble-major, pkt uint8[n, pkt_bytes])"""
    probe = OracleDecoder(list(protocols), chip_length)
    g = probe.geom
    bs, bs2 = g.block_size, g.block_size2
    n_blocks = iq.size // bs2
    nt = n_threads or default_threads()
    prime = (g.packet_length + bs - 1) // bs + 1
    todo = n_blocks - first_block
    n_sh = max(1, min(nt * 2, todo // max(4 * prime, 64)))
    edges = [first_block + (todo * i) // n_sh for i in range(n_sh + 1)]
    q = np.zeros(todo * bs // 8, np.uint8) if want_q else None
    parts = [None] * n_sh

    def job(i):
        k0, k1 = edges[i], edges[i + 1]
        o = OracleDecoder(list(protocols), chip_length)
        p0 = max(0, k0 - prime)
        if k0 > p0:
            o.decode_stream(iq[p0 * bs2: k0 * bs2], mode=mode, hits_cap=1, want_q=False, allow_overflow=True)
        cap = max(1 << 14, (k1 - k0) * 4)
        while True:
            try:
                qq, hits, hb = o_run(o, k0, k1, cap)
                break
            except OverflowError as e:        # the decoder state advanced: start over with a fresh one
                cap = int(e.args[1]) + 16
                o = OracleDecoder(list(protocols), chip_length)
                if k0 > p0:
                    o.decode_stream(iq[p0 * bs2: k0 * bs2], mode=mode, hits_cap=1, want_q=False, allow_overflow=True)
        if want_q:
            q[(k0 - first_block) * bs // 8: (k1 - first_block) * bs // 8] = qq
        hits = hits.astype(np.int64)
        hits[:, 0] += k0
        parts[i] = (hits, hb)

    def o_run(o, k0, k1, cap):
        qq, hits, hb = o.decode_stream(iq[k0 * bs2: k1 * bs2], mode=mode, hits_cap=cap, want_q=want_q)
        return qq, hits, hb

    _run_threads([lambda i=i: job(i) for i in range(n_sh)], nt)
    hits = np.concatenate([p[0] for p in parts]) if parts else np.zeros((0, 3), np.int64)
    hb = np.concatenate([p[1] for p in parts]) if parts else np.zeros((0, probe.pkt_bytes), np.uint8)
    order = np.lexsort((hits[:, 2], hits[:, 0], hits[:, 1]))      # preamble-major, then (block, idx)
    hits, hb = hits[order], hb[order]
    rows = np.stack([hits[:, 1], hits[:, 0], hits[:, 2]], axis=1)
    return q, rows, hb


def result_digest(rows: np.ndarray, pkt: np.ndarray, q) -> dict:
    """sha256 fingerprints of a decode result in canonical form: rows int64[n,3] (pid, block, idx) preamble-major,
    packet bytes uint8[n, pkt_bytes] in the same order, packed bitstream (MSB first).  Tests and bench.py compare
    the HIP path's digests with the oracle's (tests/golden/bench_golden.json)."""
    import hashlib
    d = {"n_hits": int(len(rows)),
         "hits_sha256": hashlib.sha256(np.ascontiguousarray(rows, np.int64).tobytes()).hexdigest(),
         "pkt_sha256": hashlib.sha256(np.ascontiguousarray(pkt, np.uint8).tobytes()).hexdigest()}
    if q is not None:
        d["q_sha256"] = hashlib.sha256(np.ascontiguousarray(q, np.uint8).tobytes()).hexdigest()
    return d
