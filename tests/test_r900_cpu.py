"""CPU side of the r900 second stage: GF(32)/Reed-Solomon mirror, digit parsing, the numpy oracle against a literal
float32 loop, and -- end to end on the CPU oracle -- recovery of planted Reed-Solomon-valid r900 bursts."""
import numpy as np

from oracle import r900_oracle
from rtlamr_amd import synth
from rtlamr_amd.contrib.parsers import gf, r900


def test_gf32_tables_and_syndrome_of_built_codeword():
    f = gf.Field(32, 37, 2)
    assert sorted(f.exp[:31]) == list(range(1, 32)) and all(f.Mul(x, f.Inv(x)) == 1 for x in range(1, 32))
    for mid in (1, 0xDEADBEEF, 1234567890):
        sym = r900.build_r900_symbols(mid, consumption=mid % 999983)
        rs = [0] * 31
        rs[:16], rs[26:] = sym[:16], sym[16:]
        assert f.Syndrome(rs, 5, 29) == [0] * 5                       # r900.go:218
        digits = [d for s in sym for d in (s // 6, s % 6)]
        m = r900.parse_digits(f, digits, set())
        assert m is not None and m.ID == mid and m.Consumption == mid % 999983
        bad = list(digits)
        bad[7] = (bad[7] + 1) % 6
        assert r900.parse_digits(f, bad, set()) is None               # one wrong digit: syndrome or symbol check rejects


def test_numpy_quantizer_equals_literal_float32_loop():
    rng = np.random.default_rng(5)
    for cl in (8, 72):
        sig = (rng.random(40 * cl).astype(np.float32) ** 3 * np.float32(0.02)).astype(np.float32)
        cs = np.concatenate([np.zeros(1, np.float32), np.cumsum(sig, dtype=np.float32)])
        pos = np.arange(0, sig.size - 4 * cl - 1, 7)
        assert np.array_equal(r900_oracle.quantize_at(cs, pos, cl), r900_oracle.quantize_literal(sig, cl, pos))


def _stream_with_bursts(chip, n_blocks, bs, pre, mids, seed, starts):
    iq = synth.noise(n_blocks * bs, seed)
    for mid, s in zip(mids, starts):
        chips = synth.r900_chips(pre, r900.build_r900_symbols(mid, consumption=mid & 0xFFFFFF))
        synth.plant_chips(iq, s, chips, chip, 34, -29)
    return iq


def test_oracle_recovers_planted_r900_bursts():
    chip = 72
    o = r900_oracle.OracleDecoder(["r900"], chip)
    bs = o.geom.block_size
    pre = r900_oracle.PROTOCOLS["r900"][0]
    mids = [11111, 22222222, 3333333333]
    burst = (64 + 168) * chip
    starts = [5000, 5 * bs - burst // 2, 9 * bs + 100]          # 2 blocks long each; one straddles a block boundary
    iq = _stream_with_bursts(chip, 14, bs, pre, mids, 9, starts)
    hits, digits = r900_oracle.digits_for_stream(["r900"], chip, iq)
    assert len(hits) > 0
    f = gf.Field(32, 37, 2)
    got = set()
    for d in digits:
        m = r900.parse_digits(f, d, set())
        if m is not None:
            got.add(m.ID)
    assert got == set(mids)


def test_numpy_oracle_equals_literal_c_filter():
    """oracle/r900_oracle.py (digits at the hits only) against the literal C restatement of Parser.filter, which
    quantizes the whole buffer on every call as the reference does."""
    from oracle.oracle import OracleDecoder, R900Filter
    chip = 72
    o = OracleDecoder(["r900"], chip)
    g = o.geom
    pre = r900_oracle.PROTOCOLS["r900"][0]
    burst = (64 + 168) * chip
    iq = _stream_with_bursts(chip, 12, g.block_size, pre, [424242, 99], 21, [3000, 6 * g.block_size - burst // 2])
    hits, digits = r900_oracle.digits_for_stream(["r900"], chip, iq)
    assert len(hits) > 50
    flt = R900Filter(o)
    pid = o.preamble_ids[0]
    row = 0
    for k in range(12):
        res = o.decode(iq[k * g.block_size2:(k + 1) * g.block_size2])
        q = flt.step()
        for idx in res[pid][0]:
            payload = int(idx) + g.preamble_length - g.symbol_length
            want = q[payload + np.arange(42) * 4 * chip]
            assert tuple(hits[row]) == (k, int(idx)) and np.array_equal(digits[row], want)
            row += 1
    assert row == len(hits)


def test_r900bcd_parser_rereads_consumption_as_bcd():
    """r900bcd/r900bcd.go:47-72: same parser, Consumption = decimal reading of its hex digits; MsgType R900BCD."""
    import numpy as np
    import rtlamr_amd as ra
    from rtlamr_amd.contrib.parsers import r900
    syms = r900.build_r900_symbols(987654, consumption=0x123456)
    digits = np.array([d for s in syms for d in (s // 6, s % 6)], np.uint8)
    pkt = ra.new_data(bytes(15))
    pkt.Digits = digits
    plain = ra.new_parser("r900", 72).Parse([pkt])
    bcd = ra.new_parser("r900bcd", 72).Parse([pkt])
    assert len(plain) == 1 and len(bcd) == 1
    assert plain[0].Consumption == 0x123456 and plain[0].MsgType() == "R900"
    assert bcd[0].Consumption == 123456 and bcd[0].MsgType() == "R900BCD" and bcd[0].ID == 987654
    assert ra.new_parser("r900bcd", 72).Cfg().Protocol == "r900"       # it wraps r900.NewParser (r900bcd.go:35-37)
