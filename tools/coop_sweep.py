#!/usr/bin/env python
"""K1 time of small device-resident batches: one wave per block throughout (AMR_K1_COOP_MAX large) against wave-tiles +
one wave per block for the remainder (AMR_K1_COOP_MAX=0).  Run on the GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def run(coop_max, sizes, chip=72, protos=("scm",)):
    os.environ["AMR_K1_COOP_MAX"] = str(coop_max)
    import rtlamr_amd as ra
    from rtlamr_amd import _lib, synth
    L = _lib.lib()
    dec = ra.new_decoder(0)
    for p in protos:
        dec.RegisterProtocol(ra.new_parser(p, chip))
    dec.Allocate()
    dec.set_timing(2)
    bs2 = dec.Cfg.BlockSize2
    nmax = max(sizes)
    d = C.c_void_p()
    _lib.check(L.amr_dev_alloc(0, nmax * bs2, C.byref(d)), "alloc")
    synth.device_fill(0, d.value, nmax * dec.Cfg.BlockSize, seed=3, first_sample=0, packets=[], chip_length=chip)
    out = {}
    for n in sizes:
        ts = []
        for rep in range(12):
            dec.decode_batch_device(d.value, n)
            t = dec.timing()
            if rep >= 4:
                ts.append((t["demod_ms"], t["total_ms"]))
        out[n] = (float(np.median([a for a, _ in ts])), float(np.median([b for _, b in ts])))
    L.amr_dev_free(0, d)
    dec.close()
    return out


if __name__ == "__main__":
    sizes = [1, 16, 63, 64, 65, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768]
    a = run(0, sizes)
    b = run(1 << 30, sizes)
    print("blocks   tiles+rem: K1 ms / all kernels ms     wave per block: K1 ms / all kernels ms")
    for n in sizes:
        print(f"{n:6d}   {a[n][0]:.4f} / {a[n][1]:.4f}                      {b[n][0]:.4f} / {b[n][1]:.4f}")
