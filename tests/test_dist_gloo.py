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


def _w_gatherer(rank, world, port, out, cap, scale):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from rtlamr_amd import dist
    from rtlamr_amd.protocol import BatchResult
    g = dist.HitGatherer(2, cap_hits=cap)
    assert g.two_phase == (cap * 12 + 128 > 256 * 1024)
    got, trunc = [], []
    for step in range(5):   # more steps than buffer sets, different data every step; the last one overflows rank 1's slot
        n0, n1 = 3 + step * (rank + 1) * 300 * scale, 2 + step
        blk = np.arange(n0 + n1, dtype=np.uint64) + 1000 * rank + step
        idx = (np.arange(n0 + n1, dtype=np.uint32) * 7 + rank) % 4096
        br = BatchResult(8, 0, np.array([0, n0, n0 + n1], np.uint64), blk, idx, np.zeros((n0 + n1, 12), np.uint8))
        seq = g.post(br)
        assert seq == step
        if rank == 1 and step % 2:      # a lagging rank: the root's records must still be the ones of THIS gather
            import time
            time.sleep(0.05)
        if rank == 0 and step >= 1:     # the root consumes one gather behind, like bench.py
            for r in range(world):
                n_true, off, b, i = g.fetch(step - 1, r)
                (trunc if n_true > len(b) else got).append((step - 1, r, n_true, dist.rows_from_gathered(off, b, i)))
    # large slots: what went over the wire follows the hit count (header + 12 bytes per record sent, rounded up to 4 KiB),
    # not the capacity; small slots travel whole
    for step in range(5):
        n = min(cap, 3 + step * (rank + 1) * 300 * scale + 2 + step)
        want = 128 + ((12 * n + 4095) // 4096) * 4096 if g.two_phase else g.slot_bytes
        assert g.sent_bytes[step] == want, (step, g.sent_bytes)
    # unequal remainders at the end of a stream: rank 1's amr_flush had nothing deferred (an EMPTY result), rank 0's had
    # hits -- every rank still posts one gather per result, and the empty one carries zero records, not the batch before
    n5 = 4 if rank == 0 else 0
    br = BatchResult(8, 0, np.array([0, n5, n5], np.uint64), np.arange(n5, dtype=np.uint64) + 77, np.arange(n5, dtype=np.uint32),
                     np.zeros((n5, 12), np.uint8))
    assert g.post(br) == 5
    assert g.sent_bytes[5] == ((128 + 4096 if rank == 0 else 128) if g.two_phase else g.slot_bytes)
    if rank == 0:
        for r in range(world):
            n_true, off, b, i = g.fetch(4, r)
            (trunc if n_true > len(b) else got).append((4, r, n_true, dist.rows_from_gathered(off, b, i)))
        n_true, off, b, i = g.fetch(5, 0)
        assert n_true == 4 and b.tolist() == [77, 78, 79, 80]
        n_true, off, b, i = g.fetch(5, 1)
        assert n_true == 0 and len(b) == 0 and len(i) == 0 and off.tolist() == [0, 0, 0]
        np.save(os.path.join(out, "hg.npy"), np.concatenate([x[3] for x in got]))
        np.save(os.path.join(out, "trunc.npy"), np.array([(x[0], x[1], x[2], len(x[3])) for x in trunc], np.int64))
    g.wait()
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("cap,scale", [(2000, 1), (30000, 15)], ids=["whole-slots", "two-phase"])
def test_hit_gatherer_drives_the_c_slot_layout(tmp_path, cap, scale):
    """rtlamr_amd.dist.HitGatherer on gloo: the slot every rank sends is packed by amr_gather_pack_host and read by
    amr_gather_unpack -- the code the device pack kernel and amr_gather_fetch are built from -- across buffer-set
    reuse, a lagging rank, different data per gather, a slot that overflows (truncated, true count kept), an empty result
    on one rank only, and the two-phase wire rule (header, then records sized by their count)."""
    port = 33500 + (os.getpid() % 2000) + (7 if scale > 1 else 0)
    mp.spawn(_w_gatherer, args=(2, port, str(tmp_path), cap, scale), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "hg.npy"))
    trunc = np.load(os.path.join(str(tmp_path), "trunc.npy"))
    want = []
    for step in range(5):
        for rank in range(2):
            n0, n1 = 3 + step * (rank + 1) * 300 * scale, 2 + step
            if n0 + n1 > cap:
                assert [step, rank, n0 + n1, cap] in trunc.tolist()
                continue
            blk = np.arange(n0 + n1, dtype=np.int64) + 1000 * rank + step
            idx = (np.arange(n0 + n1, dtype=np.int64) * 7 + rank) % 4096
            pid = np.concatenate([np.zeros(n0, np.int64), np.ones(n1, np.int64)])
            want.append(np.stack([pid, blk, idx], axis=1))
    assert len(trunc) == 1
    assert np.array_equal(got, np.concatenate(want))


def test_slot_layout_pack_unpack_on_cpu():
    """amr_gather_pack_host / amr_gather_unpack / amr_gather_slot_bytes need no device: round trip, truncation, and the
    argument checks (short slot, inconsistent offsets, corrupt header)."""
    import ctypes as C
    from rtlamr_amd import _lib, dist
    from rtlamr_amd.protocol import BatchResult, unpack_gathered
    L = _lib.lib()
    n0, n1, n2 = 5, 0, 9
    blk = np.arange(n0 + n1 + n2, dtype=np.uint64) * 3 + (1 << 40)
    idx = np.arange(n0 + n1 + n2, dtype=np.uint32) * 11 % 8192
    br = BatchResult(4, 0, np.array([0, n0, n0 + n1, n0 + n1 + n2], np.uint64), blk, idx, np.zeros((14, 12), np.uint8))
    r, keep = dist._result_struct(br, 3)
    for cap in (64, 14, 6):
        nb = int(L.amr_gather_slot_bytes(cap))
        assert nb % 256 == 0 and nb >= 128 + int(L.amr_gather_wire_bytes(cap)) >= 128 + 12 * cap
        slot = np.zeros(nb, np.uint8)
        assert L.amr_gather_pack_host(r, cap, 77, slot.ctypes.data, nb) == _lib.AMR_OK
        g = _lib.AmrGathered()
        assert L.amr_gather_unpack(slot.ctypes.data, nb, C.byref(g)) == _lib.AMR_OK
        n_true, off, b, i = unpack_gathered(g)
        m = min(cap, 14)
        assert (n_true, int(g.seq), int(g.n_preambles)) == (14, 77, 3)
        assert off.tolist() == [0, 5, 5, 14] and np.array_equal(b, blk[:m]) and np.array_equal(i, idx[:m])
    assert [int(L.amr_gather_wire_bytes(n)) for n in (0, 1, 341, 342, 4169, 290131)] == [0, 4096, 4096, 8192, 53248, 3481600]
    slot = np.zeros(int(L.amr_gather_slot_bytes(64)), np.uint8)
    assert L.amr_gather_pack_host(r, 64, 0, slot.ctypes.data, 256) == _lib.AMR_EINVAL          # slot too small
    r.n_hits = 13
    assert L.amr_gather_pack_host(r, 64, 0, slot.ctypes.data, slot.size) == _lib.AMR_EINVAL    # offsets do not end at n_hits
    bad = np.zeros(256, np.uint8)
    bad.view(np.uint64)[:3] = (5, 9, 1)                                                        # n_sent > n_true
    g = _lib.AmrGathered()
    assert L.amr_gather_unpack(bad.ctypes.data, 256, C.byref(g)) == _lib.AMR_EINVAL
