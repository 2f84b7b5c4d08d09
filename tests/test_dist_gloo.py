"""The N>1 path on CPU: two gloo ranks shard a stream, prime their decoder with the preceding
blocks, decode their range and all-gather hit records.  The per-rank engine here is the CPU oracle
(no GPU in this container); what is under test is the sharding arithmetic (how many blocks must be
replayed so that histories match), the block-base bookkeeping and the gather -- the same
rtlamr_amd.dist code bench.py runs over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream(protos, chip, n_blocks, block_size, seed, npk):
    """Noise + planted packets, plus one packet straddling the 2-rank shard edge."""
    from rtlamr_amd import synth
    from tests import util
    iq, pk = util.synth_stream(protos, chip, n_blocks, block_size, seed, npk, edge_every=2)
    kind = [p for p in protos if p in util.PKT_BUILDERS][0]
    fn, nbits = util.PKT_BUILDERS[kind]
    edge = (n_blocks // 2) * block_size
    length = nbits * 2 * chip
    start = edge - length // 3
    # drop planted packets that would overlap the extra one: regenerate noise there is unnecessary,
    # planting adds to whatever is present, and the parity claim is oracle==sharded on ANY input
    synth.plant(iq, [synth.Packet(start, fn(999), nbits, 33, -29)], chip)
    return iq


def _worker(rank, world, port, protos, chip, n_blocks, seed, npk, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import OracleDecoder
    from rtlamr_amd import dist
    from tests import util
    o = OracleDecoder(protos, chip)
    g = o.geom
    iq = _stream(protos, chip, n_blocks, g.block_size, seed, npk)
    k0, k1 = dist.shard_range(n_blocks, world, rank)
    prime_blocks = (g.packet_length + g.block_size - 1) // g.block_size + 1   # = amr_prime_blocks()
    p0, _ = dist.prime_range(k0, prime_blocks)
    if k0 > p0:
        o.decode_stream(iq[p0 * g.block_size2: k0 * g.block_size2], want_q=False)   # prime: hits discarded
    _, hits, _ = o.decode_stream(iq[k0 * g.block_size2: k1 * g.block_size2], want_q=False)
    mine = np.stack([hits[:, 1], hits[:, 0] + k0, hits[:, 2]], axis=1).astype(np.int64)  # (pid, block, idx)
    allh = dist.gather_hits(mine)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), allh)
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("protos,chip,n_blocks,npk", [(["scm"], 72, 64, 10), (["idm"], 72, 60, 4),
                                                      (["scm", "r900"], 8, 90, 12)])
def test_two_rank_sharded_decode_equals_single_decoder(tmp_path, protos, chip, n_blocks, npk):
    sys.path.insert(0, ROOT)
    from tests import util
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, protos, chip, n_blocks, 11, npk, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npy"))
    from oracle.oracle import OracleDecoder
    o = OracleDecoder(protos, chip)
    iq = _stream(protos, chip, n_blocks, o.geom.block_size, 11, npk)
    _, _, want, _ = util.oracle_run(protos, chip, iq)
    assert len(want) > 0
    order = np.lexsort((got[:, 2], got[:, 1], got[:, 0]))
    assert np.array_equal(got[order], want)
    # some hits are reported by rank 1 although their packet starts inside rank 0's range: they only
    # come out right if the primed history equals the single decoder's
    edge = n_blocks // 2
    g = o.geom
    start = want[:, 1] * g.block_size + want[:, 2] - g.packet_length
    assert ((want[:, 1] >= edge) & (start < edge * g.block_size)).any()


def _w_empty(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from rtlamr_amd import dist
    mine = np.zeros((0, 3), np.int64) if rank == 0 else np.array([[0, 5, 7], [1, 6, 8]], np.int64)
    allh = dist.gather_hits(mine)
    if rank == 0:
        np.save(os.path.join(out, "g.npy"), allh)
    tdist.destroy_process_group()


def test_gather_handles_empty_rank(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_w_empty, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert np.load(os.path.join(str(tmp_path), "g.npy")).tolist() == [[0, 5, 7], [1, 6, 8]]


def _w_gatherer(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from rtlamr_amd import dist
    from rtlamr_amd.protocol import BatchResult
    g = dist.HitGatherer(2)
    got = []
    for step in range(4):   # more steps than buffer sets; growing hit counts force a re-negotiation
        n0, n1 = 3 + step * (rank + 1) * 700, 2 + step
        blk = np.arange(n0 + n1, dtype=np.uint64) + 1000 * rank + step
        idx = (np.arange(n0 + n1, dtype=np.uint32) * 7 + rank) % 4096
        br = BatchResult(8, 0, np.array([0, n0, n0 + n1], np.uint64), blk, idx, np.zeros((n0 + n1, 12), np.uint8))
        if step == 0:
            g.negotiate(len(idx))
        fits = g.post(br)                 # always posts: collectives stay in lockstep
        import torch
        t = torch.tensor([1 if fits else 0])
        tdist.all_reduce(t, op=tdist.ReduceOp.MIN)
        if int(t.item()) == 0:            # some rank was truncated: everybody re-negotiates and posts again
            g.negotiate(len(idx))
            assert g.post(br)
        res = g.result()
        if rank == 0:
            got.append(res)
    if rank == 0:
        np.save(os.path.join(out, "hg.npy"), np.concatenate(got))
    tdist.destroy_process_group()


def test_hit_gatherer_async_fixed_capacity(tmp_path):
    """rtlamr_amd.dist.HitGatherer on gloo/CPU tensors: header + records of every rank arrive on rank 0, across
    buffer-set reuse and a capacity re-negotiation."""
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_w_gatherer, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "hg.npy"))
    want = []
    for step in range(4):
        for rank in range(2):
            n0, n1 = 3 + step * (rank + 1) * 700, 2 + step
            blk = np.arange(n0 + n1, dtype=np.int64) + 1000 * rank + step
            idx = (np.arange(n0 + n1, dtype=np.int64) * 7 + rank) % 4096
            pid = np.concatenate([np.zeros(n0, np.int64), np.ones(n1, np.int64)])
            want.append(np.stack([pid, blk, idx], axis=1))
    assert np.array_equal(got, np.concatenate(want))
