#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-input path: 1 GiB batches from pinned / pageable host memory."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtlamr_amd as ra
from rtlamr_amd import synth
from rtlamr_amd.protocol import PinnedBuffer

dec = ra.new_decoder(0)
dec.RegisterProtocol(ra.new_parser("scm", 72))
dec.Allocate()
bs2 = dec.Cfg.BlockSize2
nb = 131072
src = synth.noise(16384 * dec.Cfg.BlockSize, seed=3)
pin = [PinnedBuffer(nb * bs2) for _ in range(2)]
for p in pin:
    for i in range(0, nb * bs2, src.size):
        p.array[i:i + src.size] = src
pageable = np.array(pin[0].array)
def run_pipelined(bufs, n):
    dec.reset()
    t0 = time.perf_counter()
    dec.submit_host(bufs[0])
    for i in range(1, n):
        dec.submit_host(bufs[i % len(bufs)])
        dec.collect(copy=False)
    dec.collect(copy=False)
    return n * nb * dec.Cfg.BlockSize / (time.perf_counter() - t0) / 1e6
def run_sync(buf, n):
    dec.reset()
    t0 = time.perf_counter()
    for _ in range(n):
        dec.decode_batch(buf)
    return n * nb * dec.Cfg.BlockSize / (time.perf_counter() - t0) / 1e6
run_pipelined([p.array for p in pin], 2)
print("pinned, pipelined amr_submit_host : %.0f Msamples/s" % run_pipelined([p.array for p in pin], 8))
print("pageable, amr_decode_batch (sync)   : %.0f Msamples/s" % run_sync(pageable, 3))
print("pinned, amr_decode_batch (sync)     : %.0f Msamples/s" % run_sync(pin[0].array, 3))
dec.close()
