#!/usr/bin/env python3
"""Is K1's duration mode (0.204 / 0.220 ms) a property of the process/device state or of where the 1 GiB IQ buffer lies?
Three IQ buffers with the same contents in one process, K1 timed on each in turn, three rounds; and the same with a
second decoder (its own bitstream buffers)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtlamr_amd as ra
from rtlamr_amd import _lib, synth
L = _lib.lib()
nbytes, chip = 1 << 30, 72
decs = []
for k in range(2):
    dec = ra.new_decoder(0); dec.RegisterProtocol(ra.new_parser("scm", chip)); dec.Allocate(); decs.append(dec)
bs, bs2 = decs[0].Cfg.BlockSize, decs[0].Cfg.BlockSize2
nb = nbytes // bs2
bufs = []
for k in range(3):
    d = C.c_void_p(); _lib.check(L.amr_dev_alloc(0, nbytes, C.byref(d)), "alloc")
    synth.device_fill(0, d.value, nb * bs, seed=2, first_sample=0, packets=[], chip_length=chip)
    bufs.append(d)
    print(f"buffer {k} at {d.value:#x}")
for rnd in range(3):
    for di, dec in enumerate(decs):
        for k, d in enumerate(bufs):
            dec.set_timing(1)
            ts = []
            for _ in range(12):
                dec.submit_device(d.value, nb); dec.collect(copy=False); ts.append(dec.timing()["demod_ms"])
            print(f"round {rnd} decoder {di} buffer {k}: K1 {np.mean(ts[2:]):.4f} ms (min {min(ts[2:]):.4f})")
