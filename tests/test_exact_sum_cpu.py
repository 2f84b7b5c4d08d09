"""The running sum of decode.go:232-236 reproduced exactly by a parallel scan (csrc/k1_single.h, ks_running_sum): the
numpy twin of the kernel's decisions -- one binade of the sum at a time, every term a two-state (parity) transducer on
the sum's integer mantissa, ties (terms exactly half way between two floats) rounded half-to-even through the parity --
against the plain sequential float32 loop, bit for bit, on receiver-like noise, uniform bytes, saturated samples, bursts
and the zero history of a fresh Decoder.  The GPU kernel itself is checked against the oracle in tests/test_gpu_single.py."""
import numpy as np

f32 = np.float32
SEQ, CAP = 256, 1 << 26


def mag_lut():
    i = np.arange(256, dtype=np.float32)
    q = (f32(127.5) - i) / f32(127.5)
    return (q * q).astype(np.float32)                     # NewMagLUT, decode.go:209-216


def sequential(m):
    c, out = f32(0), np.empty(len(m), np.float32)
    for j, x in enumerate(m):
        c = f32(c + x)                                    # decode.go:234
        out[j] = c
    return out


def by_binades(m, stats):
    n_terms = len(m)
    out = np.empty(n_terms, np.float32)
    c, pos = f32(0), 0
    while pos < min(SEQ, n_terms):                        # the leading terms, one after the other
        c = f32(c + m[pos]); out[pos] = c; pos += 1
    while pos < n_terms:
        stats["phases"] += 1
        bits = int(np.float32(c).view(np.uint32))
        assert bits != 0
        e = bits >> 23
        ulp = np.uint32((e - 23) << 23).view(np.float32)
        inv = np.uint32((277 - e) << 23).view(np.float32)
        n0 = (bits & 0x7fffff) | 0x800000
        x = (m[pos:] * inv).astype(np.float32)            # exact: a power-of-two scaling
        big = x >= f32(2 ** 25)
        t = np.trunc(x)
        tie = ((x - t).astype(np.float32) == f32(0.5)) & ~big
        stats["ties"] += int(tie.sum())
        a = np.where(big, CAP, np.where(tie, t, np.rint(np.where(big, 0, x)))).astype(np.int64)
        d0 = np.minimum(a + np.where(tie, a & 1, 0), CAP)            # n even in front of the term
        d1 = np.minimum(a + np.where(tie, (a + 1) & 1, 0), CAP)      # n odd
        n = np.empty(len(a), np.int64)
        cur = n0
        for i in range(len(a)):                           # the kernel composes these as a scan
            cur = min(cur + (d1[i] if cur & 1 else d0[i]), CAP)
            n[i] = cur
        ev = np.flatnonzero(n >= 1 << 24)
        j = pos + int(ev[0]) if len(ev) else n_terms
        out[pos:j] = n[: j - pos].astype(np.float32) * ulp
        if j < n_terms:                                   # the term that leaves the binade: one float32 addition
            c = f32((c if j == pos else out[j - 1]) + m[j])
            out[j] = c
            pos = j + 1
        else:
            pos = n_terms
    return out


def test_scan_by_binades_equals_the_sequential_float32_sum():
    lut = mag_lut()
    rng = np.random.default_rng(7)
    stats = {"phases": 0, "ties": 0}
    worst = 0
    for trial in range(60):
        n = int(rng.choice([528, 4240, 8336]))            # BlockSize + SymbolLength at chip 8, 72 (scm), 72 (idm)
        kind = trial % 6
        if kind == 0:
            iq = rng.integers(0, 256, (n, 2))             # uniform bytes
        elif kind == 1:
            iq = 119 + rng.binomial(16, 0.5, (n, 2))      # receiver noise
        elif kind == 2:
            iq = rng.choice([0, 255, 127, 128], (n, 2))   # saturated and centred samples: the largest and smallest terms
        elif kind == 3:
            iq = np.clip(127 + rng.normal(0, 30, (n, 2)), 0, 255).astype(int)
        elif kind == 4:
            iq = 119 + rng.binomial(16, 0.5, (n, 2))
            iq[int(rng.integers(0, n)):, :] += 40         # a burst: values whose sums have few mantissa bits -> many ties
        else:
            iq = np.full((n, 2), 127)
            iq[::7] = [0, 255]
        m = (lut[iq[:, 0]] + lut[iq[:, 1]]).astype(np.float32)       # decode.go:222
        if trial % 5 == 0:
            m[:144] = 0                                   # fresh Decoder: zero history (decode.go:144)
        before = stats["phases"]
        want, got = sequential(m), by_binades(m, stats)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (trial, kind, np.flatnonzero(want != got)[:4])
        worst = max(worst, stats["phases"] - before)
    assert stats["ties"] > 0, "no tie in any stream: the half-to-even branch was never exercised"
    assert worst <= 24, f"{worst} binade phases in one block: the kernel's limit is 48"
