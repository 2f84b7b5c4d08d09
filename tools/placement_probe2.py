#!/usr/bin/env python3
"""K1 duration per allocation: eight 1 GiB IQ buffers allocated one after the other in one process."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtlamr_amd as ra
from rtlamr_amd import _lib, synth
L = _lib.lib()
nbytes, chip = 1 << 30, 72
dec = ra.new_decoder(0); dec.RegisterProtocol(ra.new_parser("scm", chip)); dec.Allocate()
bs, bs2 = dec.Cfg.BlockSize, dec.Cfg.BlockSize2
nb = nbytes // bs2
first = C.c_void_p(); _lib.check(L.amr_dev_alloc(0, nbytes, C.byref(first)), "alloc")
synth.device_fill(0, first.value, nb * bs, seed=2, first_sample=0, packets=[], chip_length=chip)
dec.set_timing(1)
bufs = [first]
for k in range(1, 8):
    d = C.c_void_p(); _lib.check(L.amr_dev_alloc(0, nbytes, C.byref(d)), "alloc")
    _lib.check(L.amr_dev_upload(0, d, np.zeros(16, np.uint8).ctypes.data, 16), "touch")
    bufs.append(d)
for rnd in range(2):
    for k, d in enumerate(bufs):
        ts = []
        for _ in range(7):
            dec.submit_device(d.value, nb); dec.collect(copy=False); ts.append(dec.timing()["demod_ms"])
        print(f"round {rnd} buffer {k} at {d.value:#x} (mod 1 GiB {d.value % (1 << 30) >> 20} MiB): K1 {np.mean(ts[2:]):.4f} ms")
