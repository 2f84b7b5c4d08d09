"""File / stream replay through the GPU decoder -- the caller side of the hot path.

The reference's caller is Receiver.Run (main.go:135-296): one goroutine reads BlockSize2-byte blocks from rtl_tcp,
another calls Decoder.Decode on each and removes messages already seen in the previous block (main.go:252-260,
292).  An SDR delivers 4.7 MB/s, which a GPU does not notice; the useful callers are replay of recorded raw uint8 IQ
(the on-disk format of the reference's -samplefile dumps and of assets/sample.bin) and aggregation of many SDRs.
This module feeds such a byte stream to the decoder in large batches of whole blocks:

    reader -> pinned host buffers (3, rotating) -> amr_submit_host (H2D on its own stream) -> kernels -> hits ->
    parsers (CPU, per block, unchanged logic) -> cross-block dedupe as main.go

Trailing bytes that do not fill a block are dropped, as the reference drops a partial last read.
"""
from __future__ import annotations

import argparse
import sys
import time
from typing import BinaryIO, Iterator, List, Tuple

import numpy as np

from . import _lib
from . import protocol as ra


def replay(dec: ra.Decoder, stream: BinaryIO, batch_blocks: int = 16384, unique: bool = True,
           parse: bool = True) -> Iterator[Tuple[int, ra.Message]]:
    """Yield (block index, message) for every message the reference would print for this byte stream.

    unique=True applies main.go's suppression of a message whose digest was already produced by the previous
    block (main.go:252-260: the `prev` / `next` maps keyed on protocol.NewDigest)."""
    bs2 = dec.Cfg.BlockSize2
    bufs = [ra.PinnedBuffer(batch_blocks * bs2) for _ in range(3)]
    deferring = False
    try:
        pending: List[int] = []     # buffer index per batch in flight
        prev_seen: set = set()
        k = 0
        eof = False

        # any batch size at the full rate: the blocks behind a batch's last whole 64-block wave-tile ride with the next
        # batch (amr_set_deferral); the results say which Decode calls they cover, flush() brings in the last blocks
        try:
            dec.SetDeferral(True)
            deferring = True
        except _lib.AmrError as e:  # a decoder with the r900 second stage (AMR_EINVAL): batches are decoded as handed over
            if e.status != _lib.AMR_EINVAL:
                raise               # anything else (a closed handle, a device fault) is the caller's to see
            deferring = False

        def drain_one(flush=False):
            nonlocal prev_seen
            br = dec.flush(copy=False) if flush else dec.collect(copy=False)
            if not flush:
                pending.pop(0)
            if not parse:
                return
            for j, msgs in enumerate(dec.run_parsers(br)):
                cur = set()
                for m in msgs:
                    key = (m.MsgType(), m.MeterType(), m.MeterID(), bytes(m.Checksum()))   # protocol.NewDigest, parse.go:95-101
                    cur.add(key)
                    if unique and key in prev_seen:
                        continue
                    yield br.first_block + j, m
                prev_seen = cur

        while not eof or pending:
            while not eof and len(pending) < 2:
                b = bufs[k % 3]
                got = stream.readinto(memoryview(b.array))
                got = 0 if got is None else got
                while 0 < got < b.array.size:                      # short reads (pipes): keep filling
                    more = stream.readinto(memoryview(b.array)[got:])
                    if not more:
                        break
                    got += more
                nb = got // bs2
                if got < b.array.size:
                    eof = True
                if nb == 0:
                    break
                dec.submit_host(b.array[: nb * bs2])
                pending.append(k % 3)
                k += 1
            if pending:
                yield from drain_one()
        if deferring:
            yield from drain_one(flush=True)
    finally:
        # a generator closed early (or a parser that raised) leaves up to two batches in flight whose host-to-device
        # copies still read the pinned buffers: collect them before the buffers go
        while pending:
            try:
                dec.collect(copy=False)
            except Exception:      # the handle is beyond use; freeing the buffers is all that is left to do
                break
            pending.pop(0)
        # the decoder goes back to its caller as it came: nothing of THIS stream left in the head buffer (a later
        # decode_batch / replay on the same Decoder would silently prepend those blocks), deferral off
        if deferring:
            try:
                dec.flush(copy=False)               # closed early: the deferred blocks belong to an abandoned stream (discarded);
                dec.SetDeferral(False)              # after a complete replay nothing is deferred and this is an empty result
            except _lib.AmrError:
                pass                                # the handle is beyond use
        for b in bufs:
            b.free()


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="replay raw uint8 IQ through the MI355X decoder (rtlamr Decode hot path)")
    ap.add_argument("file", help="raw interleaved uint8 I,Q ('-' = stdin)")
    ap.add_argument("--msgtype", default="scm", help="comma separated: scm,scm+,idm,netidm,r900,r900bcd ('all' = scm,scm+,idm,r900)")
    ap.add_argument("--symbollength", type=int, default=72)
    ap.add_argument("--batch-blocks", type=int, default=16384)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--no-unique", action="store_true")
    ap.add_argument("--no-validate", action="store_true",
                    help="read every preamble hit back instead of only those that pass the parsers' checksum tests on the GPU")
    a = ap.parse_args(argv)
    names = ["scm", "scm+", "idm", "r900"] if a.msgtype == "all" else a.msgtype.split(",")   # main.go:67-73
    dec = ra.new_decoder(a.device)
    for n in names:
        dec.RegisterProtocol(ra.new_parser(n, a.symbollength))
    dec.Allocate()
    if not a.no_validate:
        dec.EnableValidation()
    dec.Log(out=lambda s: print(s, file=sys.stderr))
    f = sys.stdin.buffer if a.file == "-" else open(a.file, "rb")
    t0, n = time.perf_counter(), 0
    try:
        for blk, m in replay(dec, f, a.batch_blocks, unique=not a.no_unique):
            print(f"block {blk}: {m}")
            n += 1
    finally:
        if f is not sys.stdin.buffer:
            f.close()
        dec.close()
    print(f"{n} messages in {time.perf_counter() - t0:.3f} s", file=sys.stderr)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
