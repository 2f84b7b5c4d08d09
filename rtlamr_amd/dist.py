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
