"""The multi-GPU hit gather behind the C ABI (amr_comm_init / amr_gather_hits / amr_gather_fetch: RCCL point-to-point
on the library's own stream).  World size 1 runs on every GPU box (RCCL send/recv to self inside one group); the
two-rank test -- HIP engine + amr_prime + gather, the combination that makes the multi-GPU bench correct -- runs when
the box has two GPUs and is skipped otherwise."""
import os
import sys

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gather_world1_returns_the_batch_hits():
    from rtlamr_amd import dist
    dec = util.make_decoder(["scm", "idm"], 72)
    try:
        iq, _ = util.synth_stream(["scm", "idm"], 72, 160, dec.Cfg.BlockSize, seed=21, n_packets=8)
        dec.comm_init(dist.comm_unique_id(), 0, 1, 0, cap_hits=1 << 16)
        assert dec.comm_ranks() == 1
        for part in (slice(0, 100), slice(100, 160)):             # two gathers: the buffer sets alternate
            bs2 = dec.Cfg.BlockSize2
            br = dec.decode_batch(iq[part.start * bs2: part.stop * bs2])
            seq = dec.gather_hits()
            n_true, off, blk, idx = dec.gather_fetch(0, seq)
            assert n_true == len(br.hit_idx) > 0
            assert np.array_equal(off, np.asarray(br.preamble_offset[: dec.n_preambles + 1], np.uint64))
            assert np.array_equal(blk, np.asarray(br.hit_block, np.uint64))
            assert np.array_equal(idx, np.asarray(br.hit_idx, np.uint32))
    finally:
        dec.close()


def test_gather_after_an_empty_flush_sends_zero_records():
    """amr_flush with nothing deferred returns an empty result; the gather posted for it must carry
    ZERO records (ranks end a stream with unequal remainders and still post one gather per result), not the previous
    batch's hits again; amr_result_device pairs that result with no buffer."""
    import ctypes as C
    from rtlamr_amd import _lib, dist
    dec = util.make_decoder(["scm"], 72)
    try:
        iq, _ = util.synth_stream(["scm"], 72, 128, dec.Cfg.BlockSize, seed=23, n_packets=6)
        dec.comm_init(dist.comm_unique_id(), 0, 1, 0, cap_hits=1 << 16)
        dec.SetDeferral(True)
        dec.submit_host(iq)                                  # 128 blocks: two whole wave-tiles, nothing deferred
        br = dec.collect()
        assert br.n_blocks == 128 and len(br.hit_idx) > 0
        s0 = dec.gather_hits()
        fl = dec.flush()
        assert fl.n_blocks == 0 and len(fl.hit_idx) == 0
        s1 = dec.gather_hits()                               # the empty result's gather
        n_true, off, blk, idx = dec.gather_fetch(0, s0)
        assert n_true == len(br.hit_idx) and np.array_equal(blk, np.asarray(br.hit_block, np.uint64))
        n_true, off, blk, idx = dec.gather_fetch(0, s1)
        assert n_true == 0 and len(blk) == 0 and len(idx) == 0 and not off.any()
        d_ptr, n = C.c_void_p(1), C.c_uint64(99)
        _lib.check(_lib.lib().amr_result_device(dec._require(), C.byref(d_ptr), C.byref(n)), "amr_result_device")
        assert n.value == 0 and not d_ptr.value
    finally:
        dec.close()


def test_gather_reports_truncation():
    from rtlamr_amd import dist
    dec = util.make_decoder(["scm"], 72)
    try:
        iq, _ = util.synth_stream(["scm"], 72, 120, dec.Cfg.BlockSize, seed=22, n_packets=6)
        dec.comm_init(dist.comm_unique_id(), 0, 1, 0, cap_hits=50)
        br = dec.decode_batch(iq)
        assert len(br.hit_idx) > 50
        dec.gather_hits()
        n_true, off, blk, idx = dec.gather_fetch(0)
        assert n_true == len(br.hit_idx) and len(blk) == 50
        with pytest.raises(Exception):
            dec.gather_fetch(0, seq=5)                  # never posted
        assert np.array_equal(blk, np.asarray(br.hit_block[:50], np.uint64))
    finally:
        dec.close()


def _rank(rank, world, uid_path, out_path, protos, chip, n_blocks):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import time
    import rtlamr_amd as ra
    from rtlamr_amd import _lib, dist
    from tests import util as u
    L = _lib.lib()
    dec = ra.new_decoder(rank)
    for p in protos:
        dec.RegisterProtocol(ra.new_parser(p, chip))
    dec.Allocate()
    if rank == 0:
        open(uid_path + ".tmp", "wb").write(dist.comm_unique_id())
        os.rename(uid_path + ".tmp", uid_path)
    while not os.path.exists(uid_path):
        time.sleep(0.01)
    dec.comm_init(open(uid_path, "rb").read(), rank, world, 0, cap_hits=1 << 16)
    iq, _ = u.synth_stream(protos, chip, n_blocks, dec.Cfg.BlockSize, seed=33, n_packets=10)
    bs2 = dec.Cfg.BlockSize2
    k0, k1 = dist.shard_range(n_blocks, world, rank)
    p0, _ = dist.prime_range(k0, dec.prime_blocks())
    if k0 > p0:
        lead = iq[p0 * bs2 - dec.halo_bytes(): p0 * bs2] if p0 > 0 else None
        dec.prime(iq[p0 * bs2: k0 * bs2], lead)
    dec.set_block_base(k0)
    assert dec.comm_ranks() == world
    # the shard in three pipelined batches, one gather each; rank 1 lags (the root must still get THIS gather's records,
    # and nobody's pack kernel may read a result slot that a later batch has overwritten)
    dec.SetDeferral(True)
    n = k1 - k0
    cuts = [0, n // 3 + 5, 2 * n // 3 + 1, n]
    d = C.c_void_p()
    _lib.check(L.amr_dev_alloc(rank, n * bs2, C.byref(d)), "alloc")
    part = np.ascontiguousarray(iq[k0 * bs2: k1 * bs2])
    _lib.check(L.amr_dev_upload(rank, d, part.ctypes.data, part.size), "upload")
    rows, seqs = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        dec.submit_device(d.value + a * bs2, b - a)
    for j in range(3):
        dec.collect()
        seqs.append(dec.gather_hits())
        if rank == 1:
            time.sleep(0.02 * (j + 1))
        if rank == 0 and j >= 1:
            for r in range(world):
                rows.append(dist.rows_from_gathered(*dec.gather_fetch(r, seqs[j - 1])[1:]))
    dec.flush()
    seqs.append(dec.gather_hits())
    if rank == 0:
        for s in seqs[-2:]:
            for r in range(world):
                rows.append(dist.rows_from_gathered(*dec.gather_fetch(r, s)[1:]))
        np.save(out_path, np.concatenate(rows))
    dec.gather_wait()
    dec.close()


def test_two_rank_hip_prime_gather_equals_single_decoder(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's 8-GPU box runs it)")
    import torch.multiprocessing as mp
    protos, chip, n_blocks = ["scm", "idm"], 72, 256
    mp.spawn(_rank, args=(2, str(tmp_path / "uid"), str(tmp_path / "g.npy"), protos, chip, n_blocks), nprocs=2, join=True)
    got = np.load(str(tmp_path / "g.npy"))
    dec = util.make_decoder(protos, chip)
    try:
        iq, _ = util.synth_stream(protos, chip, n_blocks, dec.Cfg.BlockSize, seed=33, n_packets=10)
        _, want, _ = util.gpu_run(dec, iq)
    finally:
        dec.close()
    order = np.lexsort((got[:, 2], got[:, 1], got[:, 0]))
    assert len(want) > 0 and np.array_equal(got[order], want)


def test_pipelined_gathers_carry_their_own_batch(tmp_path):
    """World 1, five pipelined batches with DIFFERENT data, one gather each, the root consuming one gather behind (as
    bench.py does): every gather's records are those of its own batch -- sequence numbers, the two buffer sets, the
    pinned mirror and the ordering of the pack kernel against slot reuse all in play."""
    import ctypes as C
    from rtlamr_amd import _lib, dist
    L = _lib.lib()
    dec = util.make_decoder(["scm", "scm+"], 72)
    d = C.c_void_p()
    try:
        sizes = [130, 64, 200, 65, 128]
        bs2 = dec.Cfg.BlockSize2
        iq, _ = util.synth_stream(["scm", "scm+"], 72, sum(sizes), dec.Cfg.BlockSize, seed=29, n_packets=14)
        _lib.check(L.amr_dev_alloc(0, iq.size, C.byref(d)), "alloc")
        _lib.check(L.amr_dev_upload(0, d, iq.ctypes.data, iq.size), "upload")
        dec.comm_init(dist.comm_unique_id(), 0, 1, 0, cap_hits=1 << 16)
        pos, want, got, seqs, inflight = 0, [], [], [], 0

        def finish():
            br = dec.collect()
            want.append(dist.batch_hits_array(br, dec.n_preambles))
            seqs.append(dec.gather_hits())
            if len(seqs) >= 2:
                got.append(dist.rows_from_gathered(*dec.gather_fetch(0, seqs[-2])[1:]))
        for nb in sizes:
            dec.submit_device(d.value + pos * bs2, nb)
            pos += nb
            inflight += 1
            if inflight == 3:
                finish()
                inflight -= 1
        while inflight:
            finish()
            inflight -= 1
        got.append(dist.rows_from_gathered(*dec.gather_fetch(0, seqs[-1])[1:]))
        assert seqs == list(range(5)) and sum(len(w) for w in want) > 0
        for w, g in zip(want, got):
            assert np.array_equal(w, g)
        with pytest.raises(Exception):
            dec.gather_fetch(0, seq=1)                  # overwritten two gathers later
    finally:
        if d.value:
            L.amr_dev_free(0, d)
        dec.close()


def _group_case(devices, protos=("scm", "idm"), chip=72, n_blocks=200, cap_hits=1 << 16):
    """DeviceGroup (ONE process, one decoder per device, amr_comm_init_all / amr_gather_hits_all) against the oracle."""
    from rtlamr_amd import dist
    import rtlamr_amd as ra

    def mk(dev):
        d = ra.new_decoder(dev)
        for p in protos:
            d.RegisterProtocol(ra.new_parser(p, chip))
        d.Allocate()
        return d
    grp = dist.DeviceGroup(mk, devices, root=0, cap_hits=cap_hits)
    try:
        bs = grp.decoders[0].Cfg.BlockSize
        iq, _ = util.synth_stream(list(protos), chip, n_blocks, bs, seed=57, n_packets=10, edge_every=2)
        _, _, want, _ = util.oracle_run(list(protos), chip, iq)
        assert len(want) > 0
        for _ in range(2):                                   # twice: the two buffer sets, amr_reset between streams
            got = grp.decode(iq)
            order = np.lexsort((got[:, 2], got[:, 1], got[:, 0]))
            assert np.array_equal(got[order], want), "the group's gathered hits differ from the single decoder's"
        assert all(d.comm_ranks() == len(devices) for d in grp.decoders)
    finally:
        grp.close()


@pytest.mark.parametrize("cap_hits", [1 << 12, 1 << 16], ids=["whole-slot", "two-phase"])
def test_single_process_group_of_one_device(cap_hits):
    """amr_comm_init_all / amr_gather_hits_all with n = 1: what every MI355X box can run -- the grouped init, the
    grouped gather (send to self + receive inside one group), both wire forms."""
    _group_case([0], cap_hits=cap_hits)


def test_single_process_group_refuses_the_one_rank_gather_and_bad_handle_sets():
    import ctypes as C
    from rtlamr_amd import _lib
    L = _lib.lib()
    dec = util.make_decoder(["scm"], 72)
    try:
        arr = (C.c_void_p * 2)(dec._require(), dec._require())
        assert L.amr_comm_init_all(arr, 2, 0, 1 << 12) == _lib.AMR_EINVAL        # the same handle twice
        assert L.amr_gather_hits_all(arr, 1, None) == _lib.AMR_EINVAL            # no communicator yet
        one = (C.c_void_p * 1)(dec._require())
        _lib.check(L.amr_comm_init_all(one, 1, 0, 1 << 12), "amr_comm_init_all")
        assert L.amr_comm_init_all(one, 1, 0, 1 << 12) == _lib.AMR_EINVAL        # exists already
        assert L.amr_gather_hits_all(one, 1, None) == _lib.AMR_EINVAL            # nothing collected yet
    finally:
        dec.close()


def test_single_process_group_of_two_devices():
    """The real thing: two GPUs, ONE thread holding both handles (skipped on 1-GPU boxes)."""
    from rtlamr_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs two gfx950 devices")
    _group_case([0, 1], cap_hits=1 << 12)
    _group_case([1, 0], protos=("scm",), n_blocks=333, cap_hits=1 << 16)


# ---- n > 1 on ONE GPU: the loopback transport (amr_comm_test_loopback) ----------------------------------------------------

@pytest.fixture
def loopback():
    from rtlamr_amd import _lib
    L = _lib.lib()
    _lib.check(L.amr_comm_test_loopback(1), "loopback on")
    yield L
    _lib.check(L.amr_comm_test_loopback(0), "loopback off")


def _group_rows(grp, iq):
    rows = grp.decode(iq)
    return rows[np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0]))]


@pytest.mark.parametrize("world,root,cap", [(2, 0, 1 << 12), (3, 1, 1 << 12), (2, 1, 1 << 16), (3, 0, 1 << 16), (3, 2, 1 << 16)])
def test_loopback_group_equals_single_decoder(loopback, world, root, cap):
    """the root side of the gather with n = 2, 3 ranks -- n receives, (cap 65 536: slots > 256 KiB) the
    two-phase header wait and count-sized records, (cap 4 096) whole slots in one message -- through amr_comm_init_all /
    amr_gather_hits_all on handles that all sit on device 0, against the single decoder's (= the oracle's) hit list;
    three streams in a row so that the two buffer sets alternate and the sequence numbers advance."""
    from rtlamr_amd import _lib, dist
    protos, chip = ["scm", "idm"], 72
    assert bool(loopback.amr_gather_two_phase(cap)) == (cap > (1 << 14))
    grp = dist.DeviceGroup(lambda d: util.make_decoder(protos, chip), devices=[0] * world, root=root, cap_hits=cap)
    try:
        assert grp.decoders[root].comm_ranks() == world
        bs = grp.decoders[0].Cfg.BlockSize
        for seed, n_blocks in ((51, 150), (52, 97), (53, 210)):
            iq, _ = util.synth_stream(protos, chip, n_blocks, bs, seed=seed, n_packets=6, edge_every=2)
            want = util.oracle_run(protos, chip, iq)[2]
            assert 0 < len(want) <= cap
            got = _group_rows(grp, iq)
            assert np.array_equal(got, want), f"world {world} root {root} cap {cap}: gathered rows differ from the single decoder's"
        # only the root holds records
        other = (root + 1) % world
        with pytest.raises(_lib.AmrError):
            grp.decoders[other].gather_fetch(0, grp.last_seq)
    finally:
        grp.close()


@pytest.mark.parametrize("cap", [40, 20000])
def test_loopback_truncation_is_reported_per_rank(loopback, cap):
    """A rank with more records than the capacity sends `cap` of them and its true count (both wire forms)."""
    from rtlamr_amd import dist
    protos, chip, world = ["scm"], 8, 3
    grp = dist.DeviceGroup(lambda d: util.make_decoder(protos, chip), devices=[0] * world, root=0, cap_hits=cap)
    try:
        bs = grp.decoders[0].Cfg.BlockSize
        n_blocks = 3 * 4096 if cap == 40 else 3 * 16384
        iq, _ = util.synth_stream(protos, chip, n_blocks, bs, seed=7, n_packets=60 if cap == 40 else 9000, edge_every=3)
        want = util.oracle_run(protos, chip, iq, hits_cap=1 << 20)[2]
        bs2 = grp.decoders[0].Cfg.BlockSize2
        per_rank = []
        for dec, (k0, k1, p0) in zip(grp.decoders, grp.plan(n_blocks)):
            dec.reset()
            if k0 > p0:
                dec.prime(iq[p0 * bs2: k0 * bs2], iq[p0 * bs2 - dec.halo_bytes(): p0 * bs2] if p0 > 0 else None)
            dec.set_block_base(k0)
            dec.decode_batch(iq[k0 * bs2: k1 * bs2])
            per_rank.append(want[(want[:, 1] >= k0) & (want[:, 1] < k1)])
        seq = grp.post()
        truncated = 0
        for r in range(world):
            n_true, off, blk, idx = grp.decoders[0].gather_fetch(r, seq)
            assert n_true == len(per_rank[r])
            assert len(blk) == min(n_true, cap)
            assert np.array_equal(blk.astype(np.int64), per_rank[r][: len(blk), 1]) and np.array_equal(idx.astype(np.int64), per_rank[r][: len(blk), 2])
            truncated += n_true > cap
        assert truncated >= 1, "no rank exceeded the capacity: the test tests nothing"
        with pytest.raises(OverflowError):
            grp.result(seq)
    finally:
        grp.close()


def test_loopback_ranks_with_their_own_communicators_and_an_inconsistent_header(loopback):
    """One communicator per handle (amr_comm_init, the one-process-per-GPU form) over the loopback transport: the peers
    post first and never wait, the root then finds their sends.  Then the error path: a peer whose capacity is larger
    than the root's advertises more records than a root slot holds -- the root still posts its phase-2 receives (nobody
    hangs), fails the gather, and amr_gather_fetch refuses its records."""
    from rtlamr_amd import _lib, dist
    protos, chip = ["scm"], 8
    bs = util.make_decoder(protos, chip)
    B, bs2 = bs.Cfg.BlockSize, bs.Cfg.BlockSize2
    bs.close()
    n_blocks = 2 * 32768
    iq, _ = util.synth_stream(protos, chip, n_blocks, B, seed=9, n_packets=12000, edge_every=3)
    want = util.oracle_run(protos, chip, iq, hits_cap=1 << 21)[2]
    half = n_blocks // 2
    assert (want[:, 1] >= half).sum() > 30000

    def make(rank, uid, cap):
        dec = util.make_decoder(protos, chip)
        dec.comm_init(uid, rank, 2, 0, cap_hits=cap)
        k0, k1 = dist.shard_range(n_blocks, 2, rank)
        p0, _ = dist.prime_range(k0, dec.prime_blocks())
        if k0 > p0:
            dec.prime(iq[p0 * bs2: k0 * bs2], iq[p0 * bs2 - dec.halo_bytes(): p0 * bs2] if p0 > 0 else None)
        dec.set_block_base(k0)
        dec.decode_batch(iq[k0 * bs2: k1 * bs2])
        return dec

    # (a) equal capacities (two-phase): consistent, equal to the single decoder
    uid = dist.comm_unique_id()
    d0, d1 = make(0, uid, 1 << 17), make(1, uid, 1 << 17)
    try:
        s1 = d1.gather_hits()            # the peer first: returns at once, its sends wait for the root
        s0 = d0.gather_hits()
        assert s0 == s1 == 0
        rows = []
        for r in range(2):
            n_true, off, blk, idx = d0.gather_fetch(r, s0)
            assert n_true == len(blk)
            rows.append(dist.rows_from_gathered(off, blk, idx))
        rows = np.concatenate(rows)
        assert np.array_equal(rows[np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0]))], want)
    finally:
        d0.close(); d1.close()
    # (b) the peer believes in a larger capacity than the root's slots have
    uid = dist.comm_unique_id()
    d0, d1 = make(0, uid, 25000), make(1, uid, 1 << 17)
    try:
        d1.gather_hits()
        with pytest.raises(_lib.AmrError, match="inconsistent"):
            d0.gather_hits()
        with pytest.raises(_lib.AmrError, match="failed"):
            d0.gather_fetch(1, 0)
        d0.gather_wait(); d1.gather_wait()      # nobody hangs: every enqueued send and receive has completed
    finally:
        d0.close(); d1.close()


def test_two_handles_on_one_device_stay_refused_without_the_hook():
    from rtlamr_amd import _lib, dist
    _lib.check(_lib.lib().amr_comm_test_loopback(0), "loopback off")
    with pytest.raises(_lib.AmrError):
        dist.check_device_group([0, 0])
    dist.check_device_group([0, 1])
