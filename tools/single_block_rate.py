"""Latency case: one reference block per call (the unchanged main.go loop, main.go:235), device-resident and host input."""
import ctypes as C
import sys
import time

import numpy as np
import torch  # noqa: F401  (one HIP runtime in the process)

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtlamr_amd as ra  # noqa: E402
from rtlamr_amd import _lib, synth

chip = int(sys.argv[1]) if len(sys.argv) > 1 else 72
dec = ra.new_decoder(0)
dec.RegisterProtocol(ra.new_parser("scm", chip))
dec.Allocate()
bs, bs2 = dec.Cfg.BlockSize, dec.Cfg.BlockSize2
n = 400
iq = synth.noise(n * bs, seed=5)
L = _lib.lib()
for k in range(50):
    dec.decode_batch(iq[k * bs2:(k + 1) * bs2])
t0 = time.perf_counter()
for k in range(n):
    dec.decode_batch(iq[k * bs2:(k + 1) * bs2])
dt = (time.perf_counter() - t0) / n
print(f"scm chip {chip}: one block ({bs} samples) per call from host memory: {dt * 1e6:.0f} us per call = "
      f"{bs / dt / 1e6:.1f} Msamples/s ({bs / 2.4e6 * 1e3:.2f} ms of signal at 2.4 Msps per block)")
