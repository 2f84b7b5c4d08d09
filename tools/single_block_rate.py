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

# the library call alone (ctypes, no numpy wrapping of the result), and the kernel's own duration
res = _lib.AmrResult()
h = dec._require()
t0 = time.perf_counter()
for k in range(n):
    L.amr_decode_batch(h, iq.ctypes.data + k * bs2, bs2, 1, C.byref(res))
dt_c = (time.perf_counter() - t0) / n
dec.set_timing(1)
ks = []
for k in range(100):
    dec.decode_batch(iq[k * bs2:(k + 1) * bs2])
    ks.append(dec.timing()["demod_ms"])
print(f"  amr_decode_batch alone (ctypes loop): {dt_c * 1e6:.1f} us per call; first kernel of the call (HIP events): "
      f"median {np.median(ks) * 1e3:.1f} us, min {np.min(ks) * 1e3:.1f} us")
for protos in (["idm"], ["scm", "scm+", "idm", "r900"]):
    d2 = ra.new_decoder(0)
    for p in protos:
        d2.RegisterProtocol(ra.new_parser(p, chip))
    d2.Allocate()
    b2 = d2.Cfg.BlockSize2
    iq2 = synth.noise(200 * d2.Cfg.BlockSize, seed=6)
    for k in range(20):
        d2.decode_batch(iq2[k * b2:(k + 1) * b2])
    t0 = time.perf_counter()
    for k in range(20, 200):
        d2.decode_batch(iq2[k * b2:(k + 1) * b2])
    print(f"  {'+'.join(protos)}: {(time.perf_counter() - t0) / 180 * 1e6:.0f} us per call of {d2.Cfg.BlockSize} samples")
    d2.close()
dec.close()

# does a long-running loop of single-block calls ramp the shader clock (a fresh process starts low: bench.py's spin-up)?
dec = ra.new_decoder(0)
dec.RegisterProtocol(ra.new_parser("scm", chip))
dec.Allocate()
h = dec._require()
lat = []
t_end = time.perf_counter() + 2.0
k = 0
while time.perf_counter() < t_end:
    t0 = time.perf_counter()
    L.amr_decode_batch(h, iq.ctypes.data + (k % n) * bs2, bs2, 1, C.byref(res))
    lat.append(time.perf_counter() - t0)
    k += 1
lat = np.array(lat) * 1e6
q = len(lat) // 10
print(f"  2 s of back-to-back single-block calls ({len(lat)}): median us per call by tenth of the run: " + " ".join(f"{np.median(lat[i * q:(i + 1) * q]):.1f}" for i in range(10)))
dec.close()
