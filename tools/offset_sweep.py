#!/usr/bin/env python3
"""K1 duration against the start offset of the IQ batch inside one large allocation (does the address of the stream
relative to the HBM channel interleave decide between the 0.204 ms and the 0.220 ms mode?)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtlamr_amd as ra
from rtlamr_amd import _lib, synth
L = _lib.lib()
nbytes, chip = 1 << 30, 72
extra = 80 << 20
dec = ra.new_decoder(0); dec.RegisterProtocol(ra.new_parser("scm", chip)); dec.Allocate()
bs, bs2 = dec.Cfg.BlockSize, dec.Cfg.BlockSize2
nb = nbytes // bs2
d = C.c_void_p(); _lib.check(L.amr_dev_alloc(0, nbytes + extra, C.byref(d)), "alloc")
synth.device_fill(0, d.value, (nbytes + extra) // 2, seed=2, first_sample=0, packets=[], chip_length=chip)
print(f"allocation at {d.value:#x}")
offs = [0, 4096, 8192, 65536, 1 << 20] + [k << 21 for k in range(1, 33)] + [(k << 21) + (1 << 20) for k in (1, 2, 3)]
dec.set_timing(1)
for off in offs:
    ts = []
    for _ in range(8):
        dec.submit_device(d.value + off, nb); dec.collect(copy=False); ts.append(dec.timing()["demod_ms"])
    print(f"offset {off:>10} ({off / (1 << 20):7.3f} MiB): K1 {np.mean(ts[2:]):.4f} ms")
