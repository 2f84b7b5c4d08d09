#!/usr/bin/env python3
"""SURVEY 8d's second input distribution: uniform random bytes (the worst case for LUT bank conflicts in K1's gathers),
next to the binomial noise of the bench.  SCM chip 72, 1 GiB resident; prints K1 / search time and the whole-path rate."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtlamr_amd as ra
from rtlamr_amd import _lib, synth
L = _lib.lib()
nbytes, chip = 1 << 30, 72
print("| input | K1 ms | K1 GB/s (2 B/sample) | search ms | hits per GiB | whole path Msamples/s |")
print("|---|---|---|---|---|---|")
for kind in ("binomial noise (bench)", "uniform random bytes"):
    dec = ra.new_decoder(0)
    dec.RegisterProtocol(ra.new_parser("scm", chip))
    dec.Allocate()
    bs, bs2 = dec.Cfg.BlockSize, dec.Cfg.BlockSize2
    nb = nbytes // bs2
    d = C.c_void_p()
    _lib.check(L.amr_dev_alloc(0, nbytes, C.byref(d)), "alloc")
    if kind.startswith("binomial"):
        synth.device_fill(0, d.value, nb * bs, seed=2, first_sample=0, packets=[], chip_length=chip)
    else:   # 64 MiB of numpy random bytes, uploaded 16 times
        part = np.random.default_rng(5).integers(0, 256, 64 << 20, dtype=np.uint8)
        for k in range(16):
            _lib.check(L.amr_dev_upload(0, C.c_void_p(d.value + k * part.size), part.ctypes.data, part.size), "upload")
    dec.set_timing(2)
    for _ in range(3):
        dec.submit_device(d.value, nb); br = dec.collect(copy=False)
    t = dec.timing()
    hits = len(br.hit_idx)
    dec.set_timing(0)
    steps = 10
    t0 = time.perf_counter()
    dec.submit_device(d.value, nb)
    for _ in range(steps - 1):
        dec.submit_device(d.value, nb); dec.collect(copy=False)
    dec.collect(copy=False)
    dt = (time.perf_counter() - t0) / steps
    print(f"| {kind} | {t['demod_ms']:.3f} | {nbytes / t['demod_ms'] / 1e6:.0f} | {t['search_ms']:.3f} | {hits} | {nb * bs / dt / 1e6:.0f} |")
    L.amr_dev_free(0, d)
    dec.close()
