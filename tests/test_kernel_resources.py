"""Build-time invariants of the kernels, read from the code-object metadata of the built library (no GPU needed):

* no kernel uses scratch (round 5: the last one, the first-generation K1 of chip 96, left the product);
* every K1 tile kernel stays at or below 248 registers: two of its waves share a SIMD's 512 with the 8 registers of the tail's
  gate kernel (amr_pipeline.hip, submit; profiles/r05/k1_gate_fragmentation.txt) -- a configuration of 249..256 would still
  run, and lose a wave slot per SIMD the gate sits on;
* there is a tile kernel for every legal chip length (flags.go:127-132) and none of another generation.
"""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "rtlamr_amd", "csrc", "libamrdemod.so")
LEGAL_CHIPS = [8, 32, 40, 48, 56, 64, 72, 80, 88, 96]


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(SO):
        pytest.skip("library not built")
    if not (shutil.which("objcopy") and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf")):
        pytest.skip("binutils / llvm-readelf not here")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), SO], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = {}
    for line in out.stdout.splitlines():
        m = re.match(r"(.*?)\s+vgpr\s+(\d+) sgpr\s+(-?\d+) lds\s+(-?\d+) scratch\s+(-?\d+) spilled_vgprs (-?\d+)", line)
        if m:
            rows[m.group(1).strip()] = dict(vgpr=int(m.group(2)), scratch=int(m.group(5)), spilled=int(m.group(6)))
    assert len(rows) > 100, out.stdout[-1000:]
    return rows


def test_no_kernel_uses_scratch(kernels):
    bad = {k: v for k, v in kernels.items() if v["scratch"] > 0 or v["spilled"] > 0}
    assert not bad, bad


def test_k1_tile_kernels_leave_room_for_the_gate(kernels):
    k1 = {k: v for k, v in kernels.items() if "k1t_demod<" in k}
    chips = sorted(int(re.search(r"k1t_demod<(\d+),", k).group(1)) for k in k1)
    assert chips == LEGAL_CHIPS, chips
    over = {k: v["vgpr"] for k, v in k1.items() if v["vgpr"] > 248}
    assert not over, over


def test_no_first_generation_k1_in_the_product(kernels):
    assert not [k for k in kernels if re.search(r"\bk1_demod<", k)]
