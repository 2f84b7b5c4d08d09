"""Multi-GPU sharding of the Decode hot path.

The stream is cut into contiguous ranges of whole reference blocks, one range per rank (one
process per GPU).  Blocks are independent except for the history a Decoder carries between calls
(decode.go:165-166): SymbolLength magnitudes for the filter and PacketLength bit decisions for the
search.  A rank rebuilds that history by first running ("priming") the few blocks that precede its
range through the demodulator without reporting hits, so every call index is reported by exactly
one rank and the union of the ranks' hit lists equals the single-decoder result.  There is no
data-path collective; the only exchange is the gather of the (tiny) hit lists: behind the C ABI
over RCCL on the GPU box (CommGatherer), and -- the same slot layout, packed and unpacked by the same
C functions -- over torch.distributed "gloo" in the CPU tests (HitGatherer).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def shard_range(total_blocks: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, near-equal block ranges: [k0, k1) for `rank`."""
    base, extra = divmod(total_blocks, world_size)
    k0 = rank * base + min(rank, extra)
    return k0, k0 + base + (1 if rank < extra else 0)


def prime_range(k0: int, prime_blocks: int) -> Tuple[int, int]:
    """Blocks [p0, k0) that must be demodulated (not searched) before block k0.  p0 == 0 means the
    shard starts close enough to the stream start to replay it from the zero initial state."""
    p0 = max(0, k0 - prime_blocks)
    return p0, k0


def patch_stale_carry(rows: np.ndarray, pkt: np.ndarray, packet_symbols: int, carry: int) -> np.ndarray:
    """Packet bytes of a shard that was decoded AT THE SAME TIME as the shard before it (so its decoder started from a zero
    d.pkt, amr_prime) made equal to the single decoder's: `carry` = last byte of the last hit the preceding shards
    sliced (Decoder.stale_carry() of shard r-1, itself patched).  Decoder.Slice never clears d.pkt (decode.go:363-366): with
    r = PacketSymbols % 8 != 0 the last byte of hit i (0-based, in slicing order: call, preamble id, idx) still holds
    carry << r*(i+1) above the bits the shard itself put there.  rows int64[n,3] = (preamble id, call, idx) in any order;
    returns a patched copy of pkt (same order)."""
    r = packet_symbols % 8
    out = np.array(pkt, np.uint8, copy=True)
    if r == 0 or carry == 0 or len(rows) == 0:
        return out
    order = np.lexsort((rows[:, 2], rows[:, 0], rows[:, 1]))          # slicing order: call, preamble id, idx
    for i, j in enumerate(order[: (8 + r - 1) // r - 1]):
        out[j, -1] |= (carry << (r * (i + 1))) & 0xFF
    return out


def pack_hits(hits: np.ndarray, cap: int) -> np.ndarray:
    """hits int64[n,3] -> int64[cap,3] padded with -1."""
    out = np.full((cap, 3), -1, np.int64)
    out[: len(hits)] = hits
    return out


def gather_hits(hits: np.ndarray, device=None, group=None) -> np.ndarray:
    """All-gather variable-length hit records (pid, block, idx) int64[n,3] from every rank.

    Two collectives: counts (1 int64 per rank), then fixed-capacity padded records.  Payload is
    KBs; on xGMI this is latency-only.  Returns the concatenation in rank order on every rank
    (= global (block) order inside each rank's range)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([len(hits)], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(1, max(counts))
    mine = torch.from_numpy(pack_hits(np.asarray(hits, np.int64).reshape(-1, 3), cap)).to(dev)
    bufs = [torch.empty((cap, 3), dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    parts: List[np.ndarray] = [b[:c].cpu().numpy() for b, c in zip(bufs, counts)]
    return np.concatenate(parts) if parts else np.zeros((0, 3), np.int64)


def batch_hits_array(br, n_preambles: int) -> np.ndarray:
    """BatchResult -> int64[n,3] rows (pid, block, idx), preamble-major then (block, idx)."""
    rows = []
    for pid in range(n_preambles):
        blk, idx, _ = br.for_preamble(pid)
        rows.append(np.stack([np.full(len(blk), pid, np.int64), blk.astype(np.int64), idx.astype(np.int64)], axis=1))
    return np.concatenate(rows) if rows else np.zeros((0, 3), np.int64)


def _result_struct(br, n_preambles: int):
    """A host-side amr_result over the arrays of a BatchResult (what amr_gather_pack_host reads); keeps them alive."""
    import ctypes as C
    from . import _lib
    off = np.ascontiguousarray(br.preamble_offset[: n_preambles + 1], np.uint64)
    blk = np.ascontiguousarray(br.hit_block, np.uint64)
    idx = np.ascontiguousarray(br.hit_idx, np.uint32)
    r = _lib.AmrResult()
    r.n_preambles, r.n_hits = n_preambles, len(idx)
    r.preamble_offset = off.ctypes.data_as(C.POINTER(C.c_uint64))
    r.hit_block = blk.ctypes.data_as(C.POINTER(C.c_uint64))
    r.hit_idx = idx.ctypes.data_as(C.POINTER(C.c_uint32))
    return r, (off, blk, idx)


def rows_from_gathered(off, blk, idx) -> np.ndarray:
    """(preamble offsets, call indices, idx) of one rank's slot -> int64[n,3] rows (pid, block, idx).  A truncated slot
    (fewer records than the offsets span) keeps the records it has."""
    n = len(blk)
    pid = np.zeros(n, np.int64)
    for q in range(len(off) - 1):
        pid[min(int(off[q]), n):min(int(off[q + 1]), n)] = q
    return np.stack([pid, blk.astype(np.int64), idx.astype(np.int64)], axis=1)


class HitGatherer:
    """The hit gather for hosts WITHOUT RCCL (CPU ranks, gloo): the same slot -- header, call indices, idx, laid out by
    the C library's amr_gather_pack_host / amr_gather_unpack, the very code the device pack kernel and amr_gather_fetch
    are built from -- and the same protocol as amr_gather_hits, moved by torch.distributed instead of ncclSend/ncclRecv:
    slots of up to 256 KiB (amr_gather_two_phase(cap) == 0) travel whole in one collective; for larger ones every rank
    sends its 128-byte header, the root reads the counts, then every rank with records sends exactly
    amr_gather_wire_bytes(n_sent) bytes.  Fixed capacity agreed up front, two buffer sets, a sequence number per gather;
    a batch with more records than the capacity arrives truncated with its true count."""

    HDR = 128

    def __init__(self, n_preambles: int, cap_hits: int, root: int = 0, group=None):
        import torch
        import torch.distributed as dist
        from . import _lib
        self.torch, self.dist, self.L = torch, dist, _lib.lib()
        self.group, self.root = group, root
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.n_pre, self.cap = n_preambles, cap_hits
        self.slot_bytes = int(self.L.amr_gather_slot_bytes(cap_hits))
        self.two_phase = bool(self.L.amr_gather_two_phase(cap_hits))      # large slots only; small ones travel whole
        self.send = [torch.zeros(self.slot_bytes, dtype=torch.uint8) for _ in range(2)]
        self.recv = [[torch.zeros(self.slot_bytes, dtype=torch.uint8) for _ in range(self.world)]
                     if self.rank == root else None for _ in range(2)]
        self.work = [[], []]
        self.seq_of = [None, None]
        self.next_seq = 0
        self.sent_bytes = []          # per gather: bytes this rank put on the wire (header + records)

    def wire_bytes(self, n_sent: int) -> int:
        return int(self.L.amr_gather_wire_bytes(n_sent))

    def _drain(self, k: int) -> None:
        for w in self.work[k]:
            w.wait()
        self.work[k] = []

    def post(self, br) -> int:
        """Enqueue the gather of one batch result; returns its sequence number.  Like amr_gather_hits: a non-root rank
        does not wait for anybody, the root waits for every rank's header (it needs the counts)."""
        from . import _lib
        torch, dist = self.torch, self.dist
        seq = self.next_seq
        self.next_seq += 1
        k = seq & 1
        self._drain(k)
        r, keep = _result_struct(br, self.n_pre)
        _lib.check(self.L.amr_gather_pack_host(r, self.cap, seq, self.send[k].data_ptr(), self.slot_bytes), "amr_gather_pack_host")
        n_sent = min(int(r.n_hits), self.cap)
        if not self.two_phase:      # small slots: the whole slot in one collective, nobody waits (amr_gather_hits does the same)
            self.sent_bytes.append(self.slot_bytes)
            self.work[k].append(dist.gather(self.send[k], self.recv[k], dst=self.root, group=self.group, async_op=True))
            self.seq_of[k] = seq
            return seq
        wire = self.wire_bytes(n_sent)
        self.sent_bytes.append(self.HDR + wire)
        # phase 1: the headers
        hdrs = [torch.zeros(self.HDR, dtype=torch.uint8) for _ in range(self.world)] if self.rank == self.root else None
        hw = dist.gather(self.send[k][: self.HDR].clone(), hdrs, dst=self.root, group=self.group, async_op=True)
        if self.rank != self.root:
            self.work[k].append(hw)
            if n_sent:      # phase 2: the records, sized by their count
                self.work[k].append(dist.isend(self.send[k][self.HDR: self.HDR + wire], dst=self.root, group=self.group, tag=seq & 0xffff))
        else:
            hw.wait()
            for p in range(self.world):
                hp = hdrs[p].numpy().view(np.uint64)
                if int(hp[1]) > self.cap or int(hp[1]) > int(hp[0]) or int(hp[12]) != seq:
                    raise RuntimeError(f"gather {seq}: rank {p}'s header is inconsistent (ranks out of step?)")
                self.recv[k][p][: self.HDR] = hdrs[p]
                m = int(hp[1])
                if not m:
                    continue
                if p == self.rank:
                    self.recv[k][p][self.HDR: self.HDR + 12 * m] = self.send[k][self.HDR: self.HDR + 12 * m]
                else:
                    self.work[k].append(dist.irecv(self.recv[k][p][self.HDR: self.HDR + self.wire_bytes(m)], src=p,
                                                   group=self.group, tag=seq & 0xffff))
        self.seq_of[k] = seq
        return seq

    def wait(self) -> None:
        for k in range(2):
            self._drain(k)

    def fetch(self, seq: int, src_rank: int):
        """Root: (n_true, offsets, call indices, idx) of rank src_rank in gather `seq`."""
        import ctypes as C
        from . import _lib
        from .protocol import unpack_gathered
        k = seq & 1
        if self.seq_of[k] != seq:
            raise KeyError(f"gather {seq} was never posted or has been overwritten")
        self._drain(k)
        g = _lib.AmrGathered()
        _lib.check(self.L.amr_gather_unpack(self.recv[k][src_rank].data_ptr(), self.slot_bytes, C.byref(g)), "amr_gather_unpack")
        if int(g.seq) != seq:
            raise RuntimeError(f"rank {src_rank}'s slot carries gather {int(g.seq)}, expected {seq}")
        return unpack_gathered(g)

    def result(self, seq: int = None) -> np.ndarray:
        """Root: the records of gather `seq` (default: the last), all ranks in rank order, int64[n,3] (pid, block, idx)."""
        seq = self.next_seq - 1 if seq is None else seq
        if self.rank != self.root:
            self.wait()
            return np.zeros((0, 3), np.int64)
        rows = []
        for r in range(self.world):
            n_true, off, blk, idx = self.fetch(seq, r)
            if n_true > len(blk):
                raise OverflowError(f"rank {r} had {n_true} hit records, the gather capacity is {self.cap}")
            rows.append(rows_from_gathered(off, blk, idx))
        return np.concatenate(rows) if rows else np.zeros((0, 3), np.int64)


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 calls it and hands the 128 bytes to the other ranks)."""
    import ctypes as C
    from . import _lib
    buf = C.create_string_buffer(128)
    _lib.check(_lib.lib().amr_comm_unique_id(buf), "amr_comm_unique_id")
    return buf.raw


class CommGatherer:
    """The hit gather of the C ABI (amr_comm_init / amr_gather_hits / amr_gather_fetch: RCCL point-to-point on a stream
    of the library's own, the root's records mirrored into pinned host memory, no host synchronisation), for hosts that
    are Python.  The unique id travels over whatever the caller has -- here torch.distributed (any backend), because
    bench.py and the tests have it anyway; a cgo host would use its own transport."""

    def __init__(self, dec, cap_hits: int, root: int = 0, group=None):
        import torch.distributed as dist
        self.dec, self.root = dec, root
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        box = [comm_unique_id() if self.rank == root else None]
        dist.broadcast_object_list(box, src=root, group=group)
        dec.comm_init(box[0], self.rank, self.world, root, cap_hits)
        self.cap = cap_hits
        self.slot_bytes = int(dec.gather_slot_bytes(cap_hits))
        self.last_seq = -1

    def post(self) -> int:
        """Enqueue the gather of the batch collected last; returns its sequence number at once."""
        self.last_seq = self.dec.gather_hits()
        return self.last_seq

    def wait(self) -> None:
        self.dec.gather_wait()

    def fetch(self, seq: int, src_rank: int, copy: bool = True):
        return self.dec.gather_fetch(src_rank, seq, copy)

    def result(self, seq: int = None) -> np.ndarray:
        """Root: int64[n,3] rows (pid, block, idx) of gather `seq` (default: the last), all ranks in rank order."""
        seq = self.last_seq if seq is None else seq
        if self.rank != self.root:
            return np.zeros((0, 3), np.int64)
        rows = []
        for r in range(self.world):
            n_true, off, blk, idx = self.fetch(seq, r)
            if n_true > len(blk):
                raise OverflowError(f"rank {r} had {n_true} hit records, the gather capacity is {self.cap}")
            rows.append(rows_from_gathered(off, blk, idx))
        return np.concatenate(rows) if rows else np.zeros((0, 3), np.int64)


def check_device_group(devices, root: int = 0, cap_hits: int = 1 << 16) -> None:
    """The argument rules of amr_comm_init_all for a set of device ordinals, from the library itself (amr_comm_check_all:
    no device, no RCCL): one handle per device, root inside the group, a capacity.  Raises AmrError (AMR_EINVAL)."""
    import ctypes as C
    from . import _lib
    arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    _lib.check(_lib.lib().amr_comm_check_all(None, arr, len(devices), root, cap_hits), "amr_comm_check_all")


class DeviceGroup:
    """ONE process, one Decoder per GPU -- the shape of the reference caller (main.go:59-128 is one process) and what a
    cgo host gets from `Decoder.Devices` (go/protocol/decode_amd.go): the stream is cut into block ranges, decoder r owns
    range r, primed with the blocks in front of it; the hit records meet on decoders[root] through the C ABI's
    single-process communicator (amr_comm_init_all / amr_gather_hits_all: every rank's sends and the root's receives in
    one RCCL group, so that one thread can drive all of them).

        grp = DeviceGroup(make_decoder, devices=[0, 1, 2, 3])      # make_decoder(device_id) -> allocated Decoder
        rows = grp.decode(iq)                                      # int64[n,3] (pid, call, idx) == one Decoder's result
    """

    def __init__(self, make_decoder, devices, root: int = 0, cap_hits: int = 1 << 16):
        import ctypes as C
        from . import _lib
        check_device_group(devices, root, cap_hits)
        self.devices, self.root, self.cap = list(devices), root, cap_hits
        self.decoders = [make_decoder(d) for d in self.devices]
        self._arr = (C.c_void_p * len(self.decoders))(*[d._require() for d in self.decoders])
        _lib.check(_lib.lib().amr_comm_init_all(self._arr, len(self.decoders), root, cap_hits), "amr_comm_init_all")
        self.last_seq = -1

    @property
    def world(self) -> int:
        return len(self.decoders)

    def plan(self, total_blocks: int):
        """[(k0, k1, p0)] per decoder: its block range and the first block it is primed with."""
        out = []
        for r, dec in enumerate(self.decoders):
            k0, k1 = shard_range(total_blocks, self.world, r)
            p0, _ = prime_range(k0, dec.prime_blocks())
            out.append((k0, k1, p0))
        return out

    def post(self) -> int:
        """One gather for all decoders (the result each one collected last); returns its sequence number at once."""
        import ctypes as C
        from . import _lib
        seq = C.c_uint64(0)
        _lib.check(_lib.lib().amr_gather_hits_all(self._arr, self.world, C.byref(seq)), "amr_gather_hits_all")
        self.last_seq = int(seq.value)
        return self.last_seq

    def result(self, seq: int = None) -> np.ndarray:
        """int64[n,3] rows (pid, call, idx) of gather `seq` (default: the last), all decoders in range order."""
        seq = self.last_seq if seq is None else seq
        rows = []
        for r in range(self.world):
            n_true, off, blk, idx = self.decoders[self.root].gather_fetch(r, seq)
            if n_true > len(blk):
                raise OverflowError(f"decoder {r} had {n_true} hit records, the gather capacity is {self.cap}")
            rows.append(rows_from_gathered(off, blk, idx))
        return np.concatenate(rows) if rows else np.zeros((0, 3), np.int64)

    def decode(self, iq: np.ndarray) -> np.ndarray:
        """The whole stream `iq` (whole blocks, from the stream start) through the group: every decoder its range."""
        d0 = self.decoders[0]
        bs2 = d0.Cfg.BlockSize2
        total = iq.size // bs2
        for dec, (k0, k1, p0) in zip(self.decoders, self.plan(total)):
            dec.reset()
            if k0 > p0:
                dec.prime(iq[p0 * bs2: k0 * bs2], iq[p0 * bs2 - dec.halo_bytes(): p0 * bs2] if p0 > 0 else None)
            dec.set_block_base(k0)
            if k1 > k0:
                dec.decode_batch(iq[k0 * bs2: k1 * bs2])
            else:
                dec.flush()          # an empty range: an empty result to gather
        self.post()
        return self.result()

    def close(self) -> None:
        for d in self.decoders:
            d.close()
        self.decoders = []
