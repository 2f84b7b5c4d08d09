"""K1 duration per IQ allocation: N buffers of the bench's batch size allocated one after the other in one process, the
same stream in each; which ones run K1 in the slow mode?  Also: one 4x larger allocation, K1 on each quarter.
usage: python tools/placement_probe.py [n_buffers]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtlamr_amd as ra  # noqa: E402
from rtlamr_amd import _lib, synth  # noqa: E402

n_buf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
chip = 72
L = _lib.lib()
dec = ra.new_decoder(0)
dec.RegisterProtocol(ra.new_parser("scm", chip))
dec.Allocate()
bs, bs2 = dec.Cfg.BlockSize, dec.Cfg.BlockSize2
n_blocks = (1 << 30) // bs2
nbytes = n_blocks * bs2


def k1_ms(ptr, reps=12):
    t = []
    for i in range(reps + 4):
        dec.set_timing(1)
        dec.decode_batch_device(ptr, n_blocks)
        if i >= 4:
            t.append(dec.timing()["demod_ms"])
    return float(np.median(t)), float(np.min(t))


bufs = []
for i in range(n_buf):
    d = C.c_void_p()
    _lib.check(L.amr_dev_alloc(0, nbytes, C.byref(d)), "alloc")
    synth.device_fill(0, d.value, n_blocks * bs, seed=1, first_sample=0, packets=[], chip_length=chip)
    bufs.append(d)
for rnd in range(2):
    for i, d in enumerate(bufs):
        med, mn = k1_ms(d.value)
        print(f"round {rnd} buffer {i} at {d.value:#014x}: K1 median {med:.4f} ms  min {mn:.4f}  ({2 * nbytes / 2 / med / 1e6 / 8e3:.3f} of peak)")
big = C.c_void_p()
_lib.check(L.amr_dev_alloc(0, 4 * nbytes, C.byref(big)), "alloc")
for q in range(4):
    p = big.value + q * nbytes
    synth.device_fill(0, p, n_blocks * bs, seed=1, first_sample=0, packets=[], chip_length=chip)
    med, mn = k1_ms(p)
    print(f"4 GiB allocation at {big.value:#014x}, quarter {q}: K1 median {med:.4f} ms  min {mn:.4f}")
dec.close()
