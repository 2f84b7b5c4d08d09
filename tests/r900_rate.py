#!/usr/bin/env python3
"""Measurement of the r900 row: r900-only decoder, chip 72 (BlockSize 8192, BufferLength 24896),
1 GiB of IQ resident in HBM = 8 copies of a 128 MiB host-built stream with 64 planted Reed-Solomon-valid bursts.
Prints GPU whole-path Msamples/s (K1..K4 + read-back + host Parse) and the CPU figure for the same work:
C restatement of Decoder.Decode + r900 Parser.filter on every block (oracle/decode_oracle.c), one thread."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtlamr_amd as ra
from rtlamr_amd import _lib, synth
from rtlamr_amd.contrib.parsers import r900
from oracle.oracle import OracleDecoder, R900Filter, PROTOCOLS

chip, reps = 72, 8
dec = ra.new_decoder(0)
dec.RegisterProtocol(ra.new_parser("r900", chip))
dec.Allocate()
bs, bs2 = dec.Cfg.BlockSize, dec.Cfg.BlockSize2
nb1 = 8192
iq = synth.noise(nb1 * bs, seed=4)
burst = (64 + 168) * chip
pre = PROTOCOLS["r900"][0]
mids = list(range(5000, 5064))
for i, mid in enumerate(mids):
    s = 1000 + i * (nb1 * bs - 3 * burst) // len(mids)
    synth.plant_chips(iq, s, synth.r900_chips(pre, r900.build_r900_symbols(mid)), chip, 31 if i % 2 else -31, 28)
L = _lib.lib()
d = C.c_void_p()
_lib.check(L.amr_dev_alloc(0, reps * iq.size, C.byref(d)), "alloc")
for r in range(reps):
    _lib.check(L.amr_dev_upload(0, C.c_void_p(d.value + r * iq.size), iq.ctypes.data, iq.size), "upload")
n_blocks = reps * nb1
for _ in range(2):
    dec.reset(); br = dec.decode_batch_device(d.value, n_blocks)
t0 = time.perf_counter()
steps = 5
for _ in range(steps):
    dec.reset()
    br = dec.decode_batch_device(d.value, n_blocks)
t_gpu = (time.perf_counter() - t0) / steps
t0 = time.perf_counter()
msgs = [m for b in dec.run_parsers(br) for m in b]
t_parse = time.perf_counter() - t0
ids = {m.ID for m in msgs}
print(f"GPU: {n_blocks * bs / t_gpu / 1e6:.0f} Msamples/s decode (K1-K4 + read-back), {len(br.hit_idx)} r900 hits, "
      f"host Parse of all hits {t_parse * 1e3:.1f} ms, meters recovered {len(ids & set(mids))}/{len(mids)}")
o = OracleDecoder(["r900"], chip)
f = R900Filter(o)
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 8.0:
    o.decode(iq[(k % nb1) * bs2:(k % nb1 + 1) * bs2]); f.step(); k += 1
t_cpu = time.perf_counter() - t0
print(f"CPU (C port, 1 thread, Decode + r900 filter per block): {k * bs / t_cpu / 1e6:.1f} Msamples/s")
dec.close()
