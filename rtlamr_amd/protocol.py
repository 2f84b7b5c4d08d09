"""Host-side mirror of rtlamr's Go package `protocol` on top of the amrdemod C ABI.

Mirrors (reference file:line):
    PacketConfig            protocol/decode.go:27-42
    Decoder                 protocol/decode.go:45-63   NewDecoder :65, Log :73,
                            RegisterProtocol :100, Allocate :131, Decode :163
    Data / NewData          protocol/parse.go:55-69
    Parser / Message        protocol/parse.go:72-84
    RegisterParser / NewParser   protocol/parse.go:28-51
    NextPowerOf2            protocol/decode.go:377-379

Differences forced by the host language (Python instead of Go), none of which change what a
parser sees:
  * Decode returns a list of messages instead of a channel; parsers return lists instead of
    sending on msgCh and calling wg.Done().
  * Preambles are visited in registration order (Go's map order is random, decode.go:177).
  * Decode accepts any whole number of blocks: one Decode of n blocks == n Go Decode calls.
  * The sample work (magnitude, filter, quantize, search, slice) happens on the GPU; the
    Decoder object keeps no Signal / Quantized slices on the host (`quantized_packed()` fetches
    the new bit decisions for tests).
"""
from __future__ import annotations

import ctypes as C
import math
import threading
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import _lib

TIME_FORMAT = "2006-01-02T15:04:05.000"  # parse.go:13


@dataclass
class PacketConfig:
    """protocol.PacketConfig (decode.go:27-42), Go field names."""
    Protocol: str = ""
    Preamble: str = ""
    DataRate: int = 0
    BlockSize: int = 0
    BlockSize2: int = 0
    ChipLength: int = 0
    SymbolLength: int = 0
    SampleRate: int = 0
    PreambleSymbols: int = 0
    PacketSymbols: int = 0
    PreambleLength: int = 0
    PacketLength: int = 0
    BufferLength: int = 0
    CenterFreq: int = 0


@dataclass
class Data:
    """protocol.Data (parse.go:55-59)."""
    Idx: int = 0
    Bits: str = ""
    Bytes: bytes = b""
    Digits: Optional[np.ndarray] = None   # r900 only: the 42 base-6 digits of this hit (r900.go:187-193), from the GPU


def new_data(data: bytes) -> Data:
    """protocol.NewData (parse.go:61-69): copy the bytes, Bits = '%08b' of every byte."""
    b = bytes(data)
    return Data(Idx=0, Bits="".join(f"{x:08b}" for x in b), Bytes=b)


class Message:
    """protocol.Message (parse.go:78-84)."""

    def MsgType(self) -> str: raise NotImplementedError
    def MeterID(self) -> int: raise NotImplementedError
    def MeterType(self) -> int: raise NotImplementedError
    def Checksum(self) -> bytes: raise NotImplementedError

    # csv.Recorder (parse.go:83) and fmt.Stringer: the columns main.go's CSV / JSON / plain encoders print.  Off the hot
    # path and optional: the formats live in rtlamr_amd/contrib/parsers/record.py, loaded on first use
    def Record(self) -> List[str]:
        from .contrib.parsers import record
        return record.record(self)

    def __str__(self):
        from .contrib.parsers import record
        return record.string(self)


class Parser:
    """protocol.Parser (parse.go:72-76)."""

    def Parse(self, pkts: List[Data]) -> List[Message]: raise NotImplementedError
    def SetDecoder(self, d: "Decoder") -> None: pass
    def Cfg(self) -> PacketConfig: raise NotImplementedError


_parser_mutex = threading.Lock()
_parsers: Dict[str, Callable[[int], Parser]] = {}


def register_parser(name: str, fn: Callable[[int], Parser]) -> None:
    """protocol.RegisterParser (parse.go:28-39): nil func and duplicates panic."""
    with _parser_mutex:
        if fn is None:
            raise RuntimeError("parser: new parser func is nil")
        if name in _parsers:
            raise RuntimeError(f"parser: parser already registered ({name})")
        _parsers[name] = fn


def new_parser(name: str, chip_length: int) -> Parser:
    """protocol.NewParser (parse.go:42-51)."""
    with _parser_mutex:
        if name not in _parsers:
            raise ValueError(f'invalid message type: "{name}"\n')
        return _parsers[name](chip_length)


def next_power_of_2(v: int) -> int:
    """protocol.NextPowerOf2 (decode.go:377-379)."""
    return 1 << int(math.ceil(math.log2(float(v))))


def unpack_gathered(g: "_lib.AmrGathered", copy: bool = True):
    """amr_gathered -> (n_true, preamble_offset[n_pre+1], hit_block u64[n], hit_idx u32[n])."""
    n, n_pre = int(g.n_hits), int(g.n_preambles)
    off = np.ctypeslib.as_array(g.preamble_offset, shape=(n_pre + 1,))
    blk = np.ctypeslib.as_array(g.hit_block, shape=(n,)) if n else np.zeros(0, np.uint64)
    idx = np.ctypeslib.as_array(g.hit_idx, shape=(n,)) if n else np.zeros(0, np.uint32)
    if copy:
        off, blk, idx = off.copy(), blk.copy(), idx.copy()
    return int(g.n_true), off, blk, idx


@dataclass
class BatchResult:
    """Hits of one batch, per preamble id: (block, idx) ascending + packet bytes."""
    n_blocks: int
    first_block: int             # call index of the batch's first block
    preamble_offset: np.ndarray  # [n_pre+1]
    hit_block: np.ndarray        # uint64 [n]
    hit_idx: np.ndarray          # uint32 [n]
    pkt: np.ndarray              # uint8 [n, pkt_bytes]
    r900_preamble: int = -1      # preamble id whose hits carry digits, -1 = none
    r900_digits: Optional[np.ndarray] = None   # uint8 [hits of that preamble, 42]
    n_hits_searched: int = 0     # hits the search found (more than len(hit_idx) only with EnableValidation)

    def for_preamble(self, pid: int):
        lo, hi = int(self.preamble_offset[pid]), int(self.preamble_offset[pid + 1])
        return self.hit_block[lo:hi], self.hit_idx[lo:hi], self.pkt[lo:hi]


class Decoder:
    """protocol.Decoder (decode.go:45-63) backed by one amr_handle on one MI355X."""

    def __init__(self, device_id: int = 0):
        # NewDecoder, decode.go:65-71
        self.Cfg = PacketConfig()
        self.device_id = device_id
        self._handle: Optional[C.c_void_p] = None
        self._parsers: List[Parser] = []
        self._preamble_strs: List[str] = []
        self._preambles: Dict[str, List[Parser]] = {}
        self._protocols: List[str] = []
        self._pid_of_preamble: Dict[str, int] = {}
        self._calls = 0
        self._last_blocks = 0
        self._block_base = 0
        self._last_gather = 0
        # The exported buffers of the Go Decoder (decode.go:46-50).  No parser reads Quantized; r900 reads Signal
        # (r900/r900.go:162-170).  Both stay on the GPU unless asked for: with KeepSignal / KeepQuantized set before a
        # decode_batch() call, the buffers hold afterwards what the Go fields hold after the batch's last Decode call
        # (integration level (A) of INTEGRATION.md; go/protocol/decode_amd.go does the same).
        self.KeepSignal = False
        self.KeepQuantized = False
        self.Signal: Optional[np.ndarray] = None      # float32[BlockSize + SymbolLength]
        self.Quantized: Optional[np.ndarray] = None   # uint8[BufferLength], one decision per byte

    # -- decode.go:100-128 ------------------------------------------------
    def RegisterProtocol(self, p: Parser) -> None:
        p.SetDecoder(self)
        c = p.Cfg()
        self.Cfg.CenterFreq = c.CenterFreq
        self.Cfg.DataRate = max(self.Cfg.DataRate, c.DataRate)
        self.Cfg.ChipLength = max(self.Cfg.ChipLength, c.ChipLength)
        self.Cfg.PreambleSymbols = max(self.Cfg.PreambleSymbols, c.PreambleSymbols)
        self.Cfg.PacketSymbols = max(self.Cfg.PacketSymbols, c.PacketSymbols)
        if c.Preamble not in self._preamble_strs:
            self._preamble_strs.append(c.Preamble)
        self._preambles.setdefault(c.Preamble, []).append(p)
        self._protocols.append(c.Protocol)
        self._parsers.append(p)

    # -- decode.go:131-160 ------------------------------------------------
    def Allocate(self) -> None:
        L = _lib.lib()
        n = len(self._parsers)
        if n == 0:
            raise ValueError("Allocate: no protocol registered")
        arr = (_lib.AmrProtocol * n)()
        keep = []
        for i, p in enumerate(self._parsers):
            c = p.Cfg()
            s = c.Preamble.encode()
            keep.append(s)
            arr[i] = _lib.AmrProtocol(s, c.DataRate, c.ChipLength, c.PreambleSymbols, c.PacketSymbols)
        h = C.c_void_p()
        _lib.check(L.amr_create(arr, n, self.device_id, C.byref(h)), "amr_create")
        self._handle = h
        g = _lib.AmrGeometry()
        _lib.check(L.amr_get_geometry(h, C.byref(g)), "amr_get_geometry")
        cfg = self.Cfg
        cfg.SymbolLength, cfg.SampleRate = g.symbol_length, g.sample_rate
        cfg.PreambleLength, cfg.PacketLength = g.preamble_length, g.packet_length
        cfg.BlockSize, cfg.BlockSize2, cfg.BufferLength = g.block_size, g.block_size2, g.buffer_length
        assert (cfg.DataRate, cfg.ChipLength, cfg.PreambleSymbols, cfg.PacketSymbols) == (
            g.data_rate, g.chip_length, g.preamble_symbols, g.packet_symbols)
        self.pkt_bytes = g.pkt_bytes
        self.n_preambles = g.n_preambles
        for i, p in enumerate(self._parsers):
            self._pid_of_preamble[p.Cfg().Preamble] = L.amr_preamble_id(h, i)
            if getattr(p, "NEEDS_R900_DIGITS", False):   # r900-type parser: its second matched filter runs on the GPU
                _lib.check(L.amr_r900_enable(h, i), "amr_r900_enable")

    def EnableValidation(self) -> List[str]:
        """Binding-level option (no Go counterpart): run the checksum test and the repeated-packet
        removal every Parse starts with (scm/scm.go:68-79, idm/idm.go:68-87, ...) on the GPU, so that only hits a
        parser can turn into a message are read back.  A preamble is validated when all its parsers declare the same
        VALIDATOR; the parsers still run unchanged and emit the same messages.  Returns the validated preambles."""
        h, L = self._require(), _lib.lib()
        done = []
        for pre, parsers in self._preambles.items():
            rules = [getattr(p, "VALIDATOR", None) for p in parsers]
            if rules[0] is None or any(r != rules[0] for r in rules):
                continue
            v = _lib.AmrValidator()
            v.n_checks = len(rules[0]["checks"])
            v.dedupe_bytes = rules[0]["dedupe_bytes"]
            for c, (init, poly, residue, spans) in enumerate(rules[0]["checks"]):
                v.checks[c].init, v.checks[c].poly, v.checks[c].residue, v.checks[c].n_spans = init, poly, residue, len(spans)
                for k, (off, ln) in enumerate(spans):
                    v.checks[c].span_off[k], v.checks[c].span_len[k] = off, ln
            _lib.check(L.amr_set_validation(h, self._pid_of_preamble[pre], C.byref(v)), "amr_set_validation")
            done.append(pre)
        return done

    # ---- multi-GPU hit gather behind the C ABI (include/amrdemod.h, "multi-GPU" section) ----
    def comm_init(self, unique_id: bytes, rank: int, world: int, root: int = 0, cap_hits: int = 1 << 20) -> None:
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _lib.check(_lib.lib().amr_comm_init(self._require(), buf, rank, world, root, cap_hits), "amr_comm_init")

    def gather_slot_bytes(self, cap_hits: int) -> int:
        return int(_lib.lib().amr_gather_slot_bytes(cap_hits))

    @staticmethod
    def gather_wire_bytes(n_sent: int) -> int:
        """Bytes of records a rank with n_sent of them sends behind its 128-byte header (two-phase communicators)."""
        return int(_lib.lib().amr_gather_wire_bytes(n_sent))

    @staticmethod
    def gather_two_phase(cap_hits: int) -> bool:
        """True: a communicator of this capacity sends header + count-sized records; False: whole small slots, no wait."""
        return bool(_lib.lib().amr_gather_two_phase(cap_hits))

    def comm_ranks(self) -> int:
        """Ranks the RCCL communicator spans (ncclCommCount)."""
        n = C.c_int32()
        _lib.check(_lib.lib().amr_comm_ranks(self._require(), C.byref(n)), "amr_comm_ranks")
        return int(n.value)

    def gather_hits(self) -> int:
        """Enqueue the gather of the batch collected last (every rank, once per batch); returns at once with the
        gather's sequence number."""
        seq = C.c_uint64()
        _lib.check(_lib.lib().amr_gather_hits(self._require(), C.byref(seq)), "amr_gather_hits")
        self._last_gather = int(seq.value)
        return self._last_gather

    def gather_wait(self) -> None:
        _lib.check(_lib.lib().amr_gather_wait(self._require()), "amr_gather_wait")

    def gather_fetch(self, src_rank: int, seq: Optional[int] = None, copy: bool = True):
        """Root: (n_true, preamble_offset[n_pre+1], hit_block u64[n], hit_idx u32[n]) of rank src_rank in gather `seq`
        (default: the one posted last).  Waits for that gather's records only; copy=False returns views into the
        library's pinned mirror (valid until gather seq + 2 is posted)."""
        g = _lib.AmrGathered()
        seq = self._last_gather if seq is None else seq
        _lib.check(_lib.lib().amr_gather_fetch(self._require(), seq, src_rank, C.byref(g)), "amr_gather_fetch")
        return unpack_gathered(g, copy)

    def close(self) -> None:
        if self._handle is not None:
            _lib.lib().amr_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- decode.go:73-90 --------------------------------------------------
    def Log(self, out=print) -> None:
        c = self.Cfg
        out(f"CenterFreq: {c.CenterFreq}")
        out(f"SampleRate: {c.SampleRate}")
        out(f"DataRate: {c.DataRate}")
        out(f"ChipLength: {c.ChipLength}")
        out(f"PreambleSymbols: {c.PreambleSymbols}")
        out(f"PreambleLength: {c.PreambleLength}")
        out(f"PacketSymbols: {c.PacketSymbols}")
        out(f"PacketLength: {c.PacketLength}")
        out("Protocols: " + ",".join(self._protocols))
        out("Preambles: " + ",".join(self._preamble_strs))

    # -- the hot path -------------------------------------------------------
    def _require(self):
        if self._handle is None:
            raise RuntimeError("Decoder.Allocate() has not been called")
        return self._handle

    def _collect(self, res: "_lib.AmrResult", copy: bool = True) -> BatchResult:
        """copy=False returns views into the library's pinned result buffers (valid until the second
        submit after this collect) -- for throughput loops that only inspect the result.  The calls a result covers
        come from the library (amr_result.first_block / n_blocks): with SetDeferral they differ from the batch that
        was submitted."""
        n_blocks, first_block = int(res.n_blocks), int(res.first_block)
        self._last_blocks = n_blocks
        n = int(res.n_hits)
        npre = int(res.n_preambles)
        off = np.ctypeslib.as_array(res.preamble_offset, shape=(npre + 1,)).copy()
        if n:
            blk = np.ctypeslib.as_array(res.hit_block, shape=(n,))
            idx = np.ctypeslib.as_array(res.hit_idx, shape=(n,))
            pkt = np.ctypeslib.as_array(res.pkt, shape=(n, int(res.pkt_bytes)))
            if copy:
                blk, idx, pkt = blk.copy(), idx.copy(), pkt.copy()
        else:
            blk = np.zeros(0, np.uint64)
            idx = np.zeros(0, np.uint32)
            pkt = np.zeros((0, int(res.pkt_bytes)), np.uint8)
        rp = int(res.r900_preamble)
        dg = None
        if rp >= 0:
            nr = int(off[rp + 1] - off[rp])
            dg = np.ctypeslib.as_array(res.r900_digits, shape=(nr, 42)) if nr else np.zeros((0, 42), np.uint8)
            if copy and nr:
                dg = dg.copy()
        return BatchResult(n_blocks, first_block, off, blk, idx, pkt, rp, dg, int(res.n_hits_searched))

    def decode_batch(self, iq) -> BatchResult:
        """n = len(iq)//BlockSize2 consecutive Decode calls (decode.go:163-172 + Search/Slice), no parsers."""
        h = self._require()
        iq = np.ascontiguousarray(iq, dtype=np.uint8).reshape(-1)
        if iq.size < self.Cfg.BlockSize2:
            raise IndexError("index out of range (short input, decode.go:222)")
        n_blocks = iq.size // self.Cfg.BlockSize2
        res = _lib.AmrResult()
        _lib.check(_lib.lib().amr_decode_batch(h, iq.ctypes.data, iq.size, n_blocks, C.byref(res)), "amr_decode_batch")
        self._calls += n_blocks
        out = self._collect(res)
        if self.KeepSignal or self.KeepQuantized:
            self._refill_exports(iq, n_blocks)
        return out

    def _refill_exports(self, iq: np.ndarray, n_blocks: int) -> None:
        """Signal / Quantized as the reference leaves them after the last of the batch's Decode calls: each call slides
        the buffer by BlockSize (decode.go:165-166) and appends the block's magnitudes (MagLUT.Execute, decode.go:169,
        219-225: lut[I] + lut[Q] in float32) resp. its BlockSize decisions (decode.go:172)."""
        cfg = self.Cfg
        bs, bs2 = cfg.BlockSize, cfg.BlockSize2
        if self.Signal is None:
            self.Signal = np.zeros(bs + cfg.SymbolLength, np.float32)
            self.Quantized = np.zeros(cfg.BufferLength, np.uint8)
        if self.KeepSignal:
            lut = self.mag_lut()
            tail = iq[max(0, n_blocks - 2) * bs2: n_blocks * bs2]          # SymbolLength < BlockSize: two blocks suffice
            mags = lut[tail[0::2]] + lut[tail[1::2]]                       # float32 + float32, one rounding
            self.Signal = np.concatenate([self.Signal, mags.astype(np.float32)])[-(bs + cfg.SymbolLength):].copy()
        if self.KeepQuantized:
            bits = np.unpackbits(self.quantized_packed())[: n_blocks * bs]  # MSB first = stream order
            self.Quantized = np.concatenate([self.Quantized, bits])[-cfg.BufferLength:].copy()

    def decode_batch_device(self, d_ptr: int, n_blocks: int) -> BatchResult:
        h = self._require()
        res = _lib.AmrResult()
        _lib.check(_lib.lib().amr_decode_batch_device(h, C.c_void_p(d_ptr), n_blocks, C.byref(res)),
                   "amr_decode_batch_device")
        self._calls += n_blocks
        return self._collect(res)

    def SetDeferral(self, on: bool = True) -> None:
        """Binding-level option (include/amrdemod.h, amr_set_deferral): pipelined submits process a batch up to its
        last whole 64-block wave-tile and carry the rest into the next submit's launch; the hits of those blocks
        arrive with the next result (BatchResult.first_block / n_blocks say which calls a result covers) or with
        flush().  Lets a caller hand over ANY block count per batch at the full rate."""
        _lib.check(_lib.lib().amr_set_deferral(self._require(), 1 if on else 0), "amr_set_deferral")

    def flush(self, copy: bool = True) -> BatchResult:
        """Decode what SetDeferral left over at the end of the stream (nothing may be in flight)."""
        res = _lib.AmrResult()
        _lib.check(_lib.lib().amr_flush(self._require(), C.byref(res)), "amr_flush")
        return self._collect(res, copy)

    def submit_device(self, d_ptr: int, n_blocks: int) -> None:
        """Pipelined form: enqueue a device-resident batch and return (at most three in flight)."""
        _lib.check(_lib.lib().amr_submit_device(self._require(), C.c_void_p(d_ptr), n_blocks), "amr_submit_device")
        self._calls += n_blocks

    def submit_host(self, iq: np.ndarray) -> None:
        """Pipelined form for host-resident input (whole blocks, uint8, contiguous): the H2D copy runs on its own
        stream and overlaps the previous batch's kernels.  iq must stay untouched until the batch is collected;
        pinned_buffer() memory makes the copy a true DMA."""
        iq = np.ascontiguousarray(iq, dtype=np.uint8).reshape(-1)
        n_blocks = iq.size // self.Cfg.BlockSize2
        _lib.check(_lib.lib().amr_submit_host(self._require(), iq.ctypes.data, iq.size, n_blocks), "amr_submit_host")
        self._keep = getattr(self, "_keep", [])[-1:] + [iq]   # keep the last two inputs alive
        self._calls += n_blocks

    def collect(self, copy: bool = True) -> BatchResult:
        """Result of the oldest submitted batch."""
        res = _lib.AmrResult()
        _lib.check(_lib.lib().amr_collect(self._require(), C.byref(res)), "amr_collect")
        return self._collect(res, copy)

    def result_device(self):
        """(device pointer, n_hits) of the packed result [hit_block u64 x n | hit_idx u32 x n | pkt x n] of the
        batch collected last; valid until the second submit after that collect."""
        p, n = C.c_void_p(), C.c_uint64()
        _lib.check(_lib.lib().amr_result_device(self._require(), C.byref(p), C.byref(n)), "amr_result_device")
        return int(p.value or 0), int(n.value)

    def run_parsers(self, br: BatchResult) -> List[List[Message]]:
        """decode.go:177-187 for every block of the batch: per preamble, Slice -> []Data -> each parser."""
        first = br.first_block
        out: List[List[Message]] = [[] for _ in range(br.n_blocks)]
        for pre in self._preamble_strs:
            pid = self._pid_of_preamble[pre]
            blk, idx, pkt = br.for_preamble(pid)
            digits = br.r900_digits if pid == br.r900_preamble else None
            if len(blk) == 0:
                per_block = {}
            else:
                rel = (blk - np.uint64(first)).astype(np.int64)
                cuts = np.flatnonzero(np.diff(rel)) + 1
                starts = np.concatenate([[0], cuts])
                ends = np.concatenate([cuts, [len(rel)]])
                per_block = {int(rel[s]): (s, e) for s, e in zip(starts, ends)}
            for k in range(br.n_blocks):
                pkts: List[Data] = []
                if k in per_block:
                    s, e = per_block[k]
                    for i in range(s, e):
                        d = new_data(pkt[i].tobytes())
                        d.Idx = int(idx[i])
                        if digits is not None:
                            d.Digits = digits[i]
                        pkts.append(d)
                elif not any(getattr(p, "ALWAYS_PARSE", False) for p in self._preambles[pre]):
                    continue
                for p in self._preambles[pre]:
                    out[k].extend(p.Parse(pkts))
        return out

    def Decode(self, input) -> List[Message]:
        """Decoder.Decode (decode.go:163-197).  One block in the unchanged main.go loop (main.go:235);
        more blocks are allowed and are decoded as consecutive calls (messages concatenated)."""
        br = self.decode_batch(input)
        msgs: List[Message] = []
        for m in self.run_parsers(br):
            msgs.extend(m)
        return msgs

    # -- multi-GPU sharding support (see rtlamr_amd/dist.py) ------
    def prime_blocks(self) -> int:
        """Blocks that must be replayed before a shard: ceil(PacketLength/BlockSize) + 1."""
        return int(_lib.lib().amr_prime_blocks(self._require()))

    def halo_bytes(self) -> int:
        return int(_lib.lib().amr_halo_bytes(self._require()))

    def set_block_base(self, base: int) -> None:
        """Call index of the first block this decoder will report (start of its shard)."""
        _lib.check(_lib.lib().amr_set_block_base(self._require(), base), "amr_set_block_base")
        self._block_base = base

    def prime(self, halo_iq, lead=None) -> None:
        """Demodulate the blocks preceding a shard without searching them.  halo_iq: host uint8 array
        of whole blocks; lead: the halo_bytes() stream bytes before halo_iq (None = stream start)."""
        h = self._require()
        halo_iq = np.ascontiguousarray(halo_iq, dtype=np.uint8).reshape(-1)
        nb = halo_iq.size // self.Cfg.BlockSize2
        lp = None
        if lead is not None:
            lead = np.ascontiguousarray(lead, dtype=np.uint8).reshape(-1)
            assert lead.size == self.halo_bytes()
            lp = lead.ctypes.data
        _lib.check(_lib.lib().amr_prime(h, lp, halo_iq.ctypes.data, nb, 0), "amr_prime")

    def prime_device(self, d_halo: int, n_blocks: int, d_lead: int = 0) -> None:
        _lib.check(_lib.lib().amr_prime(self._require(), C.c_void_p(d_lead) if d_lead else None,
                                        C.c_void_p(d_halo), n_blocks, 1), "amr_prime")

    def stale_carry(self) -> int:
        """Last byte of the last hit sliced so far: what Decoder.Slice's never-cleared d.pkt hands to the next hit
        (decode.go:363-366; matters when PacketSymbols % 8 != 0).  amr_get_stale_carry."""
        b = C.c_uint8(0)
        _lib.check(_lib.lib().amr_get_stale_carry(self._require(), C.byref(b)), "amr_get_stale_carry")
        return int(b.value)

    def set_stale_carry(self, last_byte: int) -> None:
        """Continue another decoder's d.pkt: the next batch's first hit is sliced on top of `last_byte` (after prime())."""
        _lib.check(_lib.lib().amr_set_stale_carry(self._require(), int(last_byte) & 0xFF), "amr_set_stale_carry")

    def reset(self) -> None:
        _lib.check(_lib.lib().amr_reset(self._require()), "amr_reset")
        self._calls = 0
        self.Signal = self.Quantized = None     # a fresh Decoder's buffers are zero (decode.go:144-145)

    # -- test / bench helpers ----------------------------------------------
    def quantized_packed(self) -> np.ndarray:
        """New bit decisions of the last batch (Quantized[PacketLength:] per call), packed MSB-first."""
        h = self._require()
        out = np.zeros(self._last_blocks * self.Cfg.BlockSize // 8, np.uint8)
        _lib.check(_lib.lib().amr_copy_quantized(h, out.ctypes.data, out.size), "amr_copy_quantized")
        return out

    def set_timing(self, level: int) -> None:
        """0 = no timing events (default), 1 = K1 only, 2 = K1 and search; an event costs ~5 us on the stream."""
        _lib.check(_lib.lib().amr_set_timing(self._require(), level), "amr_set_timing")

    def timing(self):
        t = _lib.AmrTiming()
        _lib.check(_lib.lib().amr_get_timing(self._require(), C.byref(t)), "amr_get_timing")
        return dict(demod_ms=t.demod_ms, search_ms=t.search_ms, total_ms=t.total_ms)

    def mag_lut(self) -> np.ndarray:
        out = np.zeros(256, np.float32)
        _lib.check(_lib.lib().amr_get_mag_lut(self._require(), out.ctypes.data_as(C.POINTER(C.c_float))), "amr_get_mag_lut")
        return out

    def describe(self) -> str:
        buf = C.create_string_buffer(512)
        _lib.check(_lib.lib().amr_describe(self._require(), buf, 512), "amr_describe")
        return buf.value.decode()

    def k1_kernel(self) -> str:
        """The demodulation kernel whole wave-tiles of this decoder's chip length run, as rocprofv3 prints it."""
        d = self.describe()
        return d.split(" | K1 ", 1)[1].split(" |", 1)[0] if " | K1 " in d else ""


class PinnedBuffer:
    """uint8 numpy view of page-locked host memory (amr_host_alloc); free() or let it be garbage collected."""

    def __init__(self, nbytes: int):
        self._p = C.c_void_p()
        _lib.check(_lib.lib().amr_host_alloc(nbytes, C.byref(self._p)), "amr_host_alloc")
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self._p.value))

    def free(self) -> None:
        if self._p is not None and self._p.value:
            self.array = None
            _lib.lib().amr_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def new_decoder(device_id: int = 0) -> Decoder:
    """protocol.NewDecoder (decode.go:65)."""
    return Decoder(device_id)
