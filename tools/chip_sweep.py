#!/usr/bin/env python3
"""BASELINE config 4/5 on one GPU: K1 / search time and whole-path rate per chip length (1 GiB resident, SCM), the IDM
geometry (4 GiB) and the four-preamble "all" geometry (2 GiB)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtlamr_amd as ra
from rtlamr_amd import _lib, synth
L = _lib.lib()
cases = [(["scm"], c, 1 << 30) for c in (8, 32, 40, 48, 56, 64, 72, 80, 88, 96)] + [(["idm"], 72, 4 << 30),
         (["scm", "scm+", "idm", "r900"], 72, 2 << 30)]
print("| protocols | chip | BlockSize | GiB | K1 ms | K1 GB/s (2 B/sample) | search ms | whole path Msamples/s |")
print("|---|---|---|---|---|---|---|---|")
for protos, chip, nbytes in cases:
    dec = ra.new_decoder(0)
    for p in protos:
        dec.RegisterProtocol(ra.new_parser(p, chip))
    dec.Allocate()
    bs, bs2 = dec.Cfg.BlockSize, dec.Cfg.BlockSize2
    nb = nbytes // bs2
    d = C.c_void_p()
    _lib.check(L.amr_dev_alloc(0, nbytes, C.byref(d)), "alloc")
    synth.device_fill(0, d.value, nb * bs, seed=2, first_sample=0, packets=[], chip_length=chip)
    dec.set_timing(2)
    for _ in range(2):
        dec.submit_device(d.value, nb); dec.collect(copy=False)
    t = dec.timing()
    dec.set_timing(0)
    steps = 10
    t0 = time.perf_counter()
    dec.submit_device(d.value, nb)
    for _ in range(steps - 1):
        dec.submit_device(d.value, nb); dec.collect(copy=False)
    dec.collect(copy=False)
    dt = (time.perf_counter() - t0) / steps
    print(f"| {'+'.join(protos)} | {chip} | {bs} | {nbytes >> 30} | {t['demod_ms']:.3f} | {nbytes / t['demod_ms'] / 1e6:.0f} | "
          f"{t['search_ms']:.3f} | {nb * bs / dt / 1e6:.0f} |")
    L.amr_dev_free(0, d)
    dec.close()
