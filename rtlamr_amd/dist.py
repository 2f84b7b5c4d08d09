"""Multi-GPU sharding of the Decode hot path.

The stream is cut into contiguous ranges of whole reference blocks, one range per rank (one
process per GPU).  Blocks are independent except for the history a Decoder carries between calls
(decode.go:165-166): SymbolLength magnitudes for the filter and PacketLength bit decisions for the
search.  A rank rebuilds that history by first running ("priming") the few blocks that precede its
range through the demodulator without reporting hits, so every call index is reported by exactly
one rank and the union of the ranks' hit lists equals the single-decoder result.  There is no
data-path collective; the only exchange is the gather of the (tiny) hit lists, done with
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
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


class _DevBuf:
    """Zero-copy torch view of raw device memory (the packed result buffer of libamrdemod)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class HitGatherer:
    """(Fallback path; the production gather is CommGatherer below, behind the C ABI and without the host-side stream
    synchronisation this one needs to protect the library's result slot.)

    Gather of the per-rank hit records (call index u64, idx u32) on rank 0, once per batch, without a host
    round trip: the records are copied device-to-device out of the library's packed result buffer into a
    fixed-capacity send buffer and gathered with ONE collective (torch.distributed.gather, backend "nccl" =
    RCCL: on the fully connected xGMI fabric every peer sends its few MB to rank 0 over its own link).  The
    collective is asynchronous; two buffer sets alternate, so the gather of batch i overlaps the kernels of
    batch i+1.  The capacity is agreed once (all_reduce MAX) and re-agreed only when a rank outgrows it; the
    record count and the per-preamble offsets travel in a small header in front of the records.

    With device=None (CPU tensors, gloo) the same code path is exercised by the CPU tests: `post` then takes the
    records from the host result instead of the device buffer."""

    HDR = 16   # int64 words: [n, n_pre, offs[0..n_pre], ...]

    def __init__(self, n_preambles: int, device=None, group=None, slack: float = 1.5):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.dev = device if device is not None else torch.device("cpu")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_pre = n_preambles
        self.slack = slack
        self.cap = 0
        self.send = [None, None]
        self.recv = [None, None]
        self.work = [None, None]
        self.i = 0

    def _nbytes(self, cap):
        return self.HDR * 8 + cap * 12

    def negotiate(self, n_local: int) -> None:
        """Collective: agree on a capacity that holds every rank's hit count (with slack)."""
        t = self.torch.tensor([n_local], dtype=self.torch.int64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        cap = max(1024, int(int(t.item()) * self.slack))
        if cap > self.cap:
            self.wait()
            self.cap = cap
            nb = self._nbytes(cap)
            self.send = [self.torch.zeros(nb, dtype=self.torch.uint8, device=self.dev) for _ in range(2)]
            self.recv = [[self.torch.zeros(nb, dtype=self.torch.uint8, device=self.dev) for _ in range(self.world)]
                         if self.rank == 0 else None for _ in range(2)]

    def post(self, br, d_ptr: int = 0) -> bool:
        """Enqueue the gather of one batch -- always, so that every rank issues the same collectives in the same
        order.  A batch that does not fit the agreed capacity is sent truncated with its true count in the header
        (result() then refuses it); returns False in that case: at the next point where all ranks synchronise
        anyway the caller runs negotiate() and posts the batch again."""
        torch = self.torch
        n_true = len(br.hit_idx)
        n = min(n_true, self.cap)
        i = self.i
        if self.work[i] is not None:
            self.work[i].wait()
        buf = self.send[i]
        hdr = np.zeros(self.HDR, np.int64)
        hdr[0], hdr[1] = n_true, self.n_pre
        hdr[2:3 + self.n_pre] = br.preamble_offset[: self.n_pre + 1]
        buf[: self.HDR * 8].copy_(torch.from_numpy(hdr.view(np.uint8)), non_blocking=True)
        if n:
            if d_ptr and self.dev.type == "cuda":
                # packed result = [block u64 x n_true | idx u32 x n_true | ...]: two pieces when truncated
                src = torch.as_tensor(_DevBuf(d_ptr, 12 * n_true), device=self.dev)
                buf[self.HDR * 8: self.HDR * 8 + 8 * n].copy_(src[: 8 * n], non_blocking=True)
                buf[self.HDR * 8 + 8 * n: self.HDR * 8 + 12 * n].copy_(src[8 * n_true: 8 * n_true + 4 * n], non_blocking=True)
                torch.cuda.current_stream(self.dev).synchronize()                   # before the library reuses the slot
            else:
                rec = np.concatenate([np.ascontiguousarray(br.hit_block[:n], np.uint64).view(np.uint8),
                                      np.ascontiguousarray(br.hit_idx[:n], np.uint32).view(np.uint8)])
                buf[self.HDR * 8: self.HDR * 8 + 12 * n].copy_(torch.from_numpy(rec))
        self.work[i] = self.dist.gather(buf, self.recv[i], dst=0, group=self.group, async_op=True)
        self.last = i
        self.i ^= 1
        return n_true <= self.cap

    def wait(self) -> None:
        for k in range(2):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None

    def result(self) -> np.ndarray:
        """Rank 0: the records of the batch posted last, all ranks, as int64[n,3] rows (pid, block, idx)."""
        self.wait()
        if self.rank != 0:
            return np.zeros((0, 3), np.int64)
        rows = []
        for t in self.recv[self.last]:
            raw = t.cpu().numpy()
            hdr = raw[: self.HDR * 8].view(np.int64)
            n, n_pre = int(hdr[0]), int(hdr[1])
            if n > self.cap:
                raise OverflowError(f"a rank sent {n} hit records, agreed capacity is {self.cap}: negotiate() and post again")
            offs = hdr[2:3 + n_pre]
            blk = raw[self.HDR * 8: self.HDR * 8 + 8 * n].view(np.uint64).astype(np.int64)
            idx = raw[self.HDR * 8 + 8 * n: self.HDR * 8 + 12 * n].view(np.uint32).astype(np.int64)
            pid = np.zeros(n, np.int64)
            for q in range(n_pre):
                pid[offs[q]:offs[q + 1]] = q
            rows.append(np.stack([pid, blk, idx], axis=1))
        return np.concatenate(rows) if rows else np.zeros((0, 3), np.int64)


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 calls it and hands the 128 bytes to the other ranks)."""
    import ctypes as C
    from . import _lib
    buf = C.create_string_buffer(128)
    _lib.check(_lib.lib().amr_comm_unique_id(buf), "amr_comm_unique_id")
    return buf.raw


class CommGatherer:
    """The hit gather of the C ABI (amr_comm_init / amr_gather_hits: RCCL point-to-point on a stream of the
    library's own, no host synchronisation), for hosts that are Python.  The unique id travels over whatever the
    caller has -- here torch.distributed (any backend), because bench.py and the tests have it anyway; a cgo host
    would use its own transport."""

    def __init__(self, dec, cap_hits: int, root: int = 0, group=None):
        import torch
        import torch.distributed as dist
        self.dec, self.root = dec, root
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        box = [comm_unique_id() if self.rank == root else None]
        dist.broadcast_object_list(box, src=root, group=group)
        dec.comm_init(box[0], self.rank, self.world, root, cap_hits)
        self.cap = cap_hits

    def post(self) -> None:
        self.dec.gather_hits()

    def wait(self) -> None:
        self.dec.gather_wait()

    def result(self):
        """Root: int64[n,3] rows (pid, block, idx) of the last gather, all ranks in rank order."""
        import numpy as np
        self.wait()
        if self.rank != self.root:
            return np.zeros((0, 3), np.int64)
        rows = []
        for r in range(self.world):
            n_true, off, blk, idx = self.dec.gather_fetch(r)
            if n_true > len(blk):
                raise OverflowError(f"rank {r} had {n_true} hit records, the gather capacity is {self.cap}")
            pid = np.zeros(len(blk), np.int64)
            for q in range(len(off) - 1):
                pid[int(off[q]):int(off[q + 1])] = q
            rows.append(np.stack([pid, blk.astype(np.int64), idx.astype(np.int64)], axis=1))
        return np.concatenate(rows) if rows else np.zeros((0, 3), np.int64)
