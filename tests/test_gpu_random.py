"""Seeded randomized parity sweep: random protocol sets, chip lengths, stream lengths, batch splits (including
single-block calls and splits inside a 64-block wave tile), packet amplitudes and positions, byte distributions --
the HIP path against the oracle on quantized bits, hit lists and packet bytes, bit for bit."""
import os

import numpy as np
import pytest

from rtlamr_amd import synth
from tests import util

pytestmark = pytest.mark.gpu
CHIPS = [8, 32, 40, 48, 56, 64, 72, 80, 88, 96]
PROTO_SETS = [["scm"], ["scm+"], ["idm"], ["netidm"], ["r900"], ["scm", "scm+"], ["scm", "idm"], ["idm", "netidm"],
              ["scm+", "idm"], ["scm", "r900"], ["scm", "scm+", "idm"], ["scm", "scm+", "idm", "r900"]]


def _random_split(rng, n):
    parts = []
    while n > 0:
        k = int(min(n, rng.choice([1, 1, 2, 3, 63, 64, 65, int(rng.integers(1, n + 1))])))
        parts.append(k)
        n -= k
    return parts


@pytest.mark.parametrize("seed", range(int(os.environ.get("AMR_RANDOM_SEEDS", "48"))))   # soak: AMR_RANDOM_SEEDS=1000
def test_random_configuration(seed):
    rng = np.random.default_rng(1000 + seed)
    protos = PROTO_SETS[int(rng.integers(len(PROTO_SETS)))]
    chip = int(rng.choice(CHIPS))
    dec = util.make_decoder(protos, chip)
    try:
        bs, pl = dec.Cfg.BlockSize, dec.Cfg.PacketLength
        longest = max([util.PKT_BUILDERS[p][1] for p in protos if p in util.PKT_BUILDERS] or [96]) * 2 * chip
        # keep the oracle's work bounded: ~1.5 M samples at most, at least room for four packets
        n_blocks = int(np.clip(rng.integers(70, 400), (5 * longest) // bs + 2, max(70, 1_500_000 // bs)))
        n_packets = int(min(rng.integers(2, 12), (n_blocks * bs) // (longest + 64) - 1))
        amp = (int(rng.integers(18, 45)), int(rng.integers(-45, -18)))
        iq, _ = util.synth_stream(protos, chip, n_blocks, bs, seed=int(rng.integers(1 << 30)), n_packets=max(n_packets, 1),
                                  edge_every=int(rng.integers(2, 5)), amp=amp)
        kind = int(rng.integers(3))
        if kind == 1:      # a stretch of uniform random bytes: every LUT entry, saturated values included
            a = int(rng.integers(0, iq.size // 2)); b = min(iq.size, a + int(rng.integers(1000, 200_000)))
            iq[a:b] = rng.integers(0, 256, b - a, dtype=np.uint8)
        elif kind == 2:    # DC steps and clipped samples
            a = int(rng.integers(0, iq.size // 2)) & ~1
            iq[a:a + 20_000] = np.clip(iq[a:a + 20_000].astype(np.int32) + int(rng.integers(-120, 120)), 0, 255).astype(np.uint8)
        split = _random_split(rng, n_blocks)
        want = util.oracle_run(protos, chip, iq)
        got = util.gpu_run(dec, iq, split)
        util.assert_same(want, got, dec.Cfg.PacketSymbols)
    finally:
        dec.close()
