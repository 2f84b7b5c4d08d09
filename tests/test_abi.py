"""The C-ABI library loads without a GPU and exports every symbol include/amrdemod.h declares.
No compute calls here (that is what -m gpu is for)."""
import ctypes as C
import os
import re

import pytest

from rtlamr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "amrdemod.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(amr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(amr_lib):
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(amr_lib, name), f"{name} declared in amrdemod.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared, "rtlamr_amd/_lib.py SYMBOLS out of sync with the header"


def test_struct_layouts_match_header(tmp_path):
    """sizeof / offsetof as a C compiler sees include/amrdemod.h == the ctypes mirrors in rtlamr_amd/_lib.py."""
    import subprocess
    structs = {"amr_geometry": _lib.AmrGeometry, "amr_protocol": _lib.AmrProtocol, "amr_timing": _lib.AmrTiming,
               "amr_result": _lib.AmrResult, "amr_crc_check": _lib.AmrCrcCheck, "amr_validator": _lib.AmrValidator,
               "amr_gathered": _lib.AmrGathered}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "amrdemod.h"', 'int main(void){']
    for cname, ct in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append('return 0;}')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    seen = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, ct in structs.items():
        assert int(seen[cname]) == C.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(seen[f"{cname}.{fname}"]) == getattr(ct, fname).offset, f"{cname}.{fname}"


def test_strerror_and_argument_checks(amr_lib):
    assert amr_lib.amr_strerror(0) == b"ok"
    assert b"fallback" in amr_lib.amr_strerror(_lib.AMR_ENODEV)
    h = C.c_void_p()
    assert amr_lib.amr_create(None, 0, 0, C.byref(h)) == _lib.AMR_EINVAL
    assert amr_lib.amr_reset(None) == _lib.AMR_EINVAL
    assert amr_lib.amr_preamble_id(None, 0) == -1


def test_no_device_fails_loudly(amr_lib):
    """Without a gfx950 GPU the product path must refuse to run (no CPU fallback)."""
    import rtlamr_amd as ra
    d = ra.new_decoder()
    d.RegisterProtocol(ra.new_parser("scm", 72))
    try:
        d.Allocate()
    except _lib.AmrError as e:
        assert e.status == _lib.AMR_ENODEV
    else:   # a GPU is present (running the CPU suite on the GPU box): creation works, nothing else to check
        d.close()


def test_product_package_never_imports_oracle():
    import subprocess
    import sys
    code = ("import sys; import rtlamr_amd, rtlamr_amd.dist, rtlamr_amd.synth, rtlamr_amd._lib; "
            "bad=[m for m in sys.modules if m.split('.')[0]=='oracle']; assert not bad, bad")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rtlamr_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def test_plan_geometry_equals_oracle_for_every_protocol_set_and_chip_length(amr_lib):
    """amr_plan = RegisterProtocol + Allocate arithmetic in the library (decode.go:100-141), no device needed: against
    the oracle's geometry for every combination of reference parsers and legal chip length; and its argument checks."""
    import itertools
    from oracle.oracle import PROTOCOLS, OracleDecoder
    import rtlamr_amd as ra
    names = ["scm", "scm+", "idm", "netidm", "r900"]
    for r in range(1, len(names) + 1):
        for protos in itertools.combinations(names, r):
            for chip in (8, 32, 40, 48, 56, 64, 72, 80, 88, 96):
                cfgs = [ra.new_parser(n, chip).Cfg() for n in protos]
                arr = (_lib.AmrProtocol * len(cfgs))()
                keep = []
                for i, c in enumerate(cfgs):
                    s = c.Preamble.encode(); keep.append(s)
                    arr[i] = _lib.AmrProtocol(s, c.DataRate, c.ChipLength, c.PreambleSymbols, c.PacketSymbols)
                g = _lib.AmrGeometry()
                pids = (C.c_int32 * len(cfgs))()
                assert amr_lib.amr_plan(arr, len(cfgs), C.byref(g), pids) == _lib.AMR_OK, (protos, chip)
                o = OracleDecoder(list(protos), chip).geom
                assert (g.symbol_length, g.preamble_length, g.packet_length, g.block_size, g.block_size2, g.buffer_length,
                        g.sample_rate) == (o.symbol_length, o.preamble_length, o.packet_length, o.block_size, o.block_size2,
                                           o.buffer_length, o.sample_rate), (protos, chip)
                distinct = []
                for c in cfgs:
                    if c.Preamble not in distinct:
                        distinct.append(c.Preamble)
                assert g.n_preambles == len(distinct)
                assert list(pids) == [distinct.index(c.Preamble) for c in cfgs]        # idm and netidm share one
    bad = (_lib.AmrProtocol * 1)(_lib.AmrProtocol(b"1010", 32768, 78, 4, 16))        # 78 is not a legal -symbollength
    g = _lib.AmrGeometry()
    assert amr_lib.amr_plan(bad, 1, C.byref(g), None) == _lib.AMR_EINVAL
    assert amr_lib.amr_plan(None, 1, C.byref(g), None) == _lib.AMR_EINVAL


def test_single_process_communicator_argument_rules(amr_lib):
    """amr_comm_check_all: the rules of amr_comm_init_all (one handle per device, root inside the group, a capacity),
    checked before any device or RCCL call -- what a host can ask on a box without a GPU."""
    from rtlamr_amd import _lib, dist

    def rc(devices, root=0, cap=1 << 12):
        arr = (C.c_int32 * max(1, len(devices)))(*devices)
        return amr_lib.amr_comm_check_all(None, arr, len(devices), root, cap)
    assert rc([0]) == _lib.AMR_OK and rc([0, 1, 2, 3, 4, 5, 6, 7], root=7) == _lib.AMR_OK and rc([3, 1]) == _lib.AMR_OK
    assert rc([]) == _lib.AMR_EINVAL                       # no rank
    assert rc([0, 1], root=2) == _lib.AMR_EINVAL           # root outside the group
    assert rc([0, 1], root=-1) == _lib.AMR_EINVAL
    assert rc([0, 1], cap=0) == _lib.AMR_EINVAL
    assert rc([0, 1, 0]) == _lib.AMR_EINVAL                # two ranks of one communicator on one device
    assert b"one device" in amr_lib.amr_last_error()
    assert rc([0, -1]) == _lib.AMR_EINVAL
    assert amr_lib.amr_comm_check_all(None, None, 1, 0, 1) == _lib.AMR_EINVAL
    assert amr_lib.amr_comm_init_all(None, 1, 0, 1) == _lib.AMR_EINVAL
    assert amr_lib.amr_gather_hits_all(None, 1, None) == _lib.AMR_EINVAL
    dist.check_device_group([0, 1, 2, 3])
    with pytest.raises(_lib.AmrError):
        dist.check_device_group([0, 0])
    # the block ranges a group of 8 gives a stream: contiguous, whole, in order; priming never reaches before the stream
    total, world, pb = 1000, 8, 5
    ranges = [dist.shard_range(total, world, r) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == total and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    assert [dist.prime_range(k0, pb)[0] for k0, _ in ranges] == [max(0, k0 - pb) for k0, _ in ranges]
