"""Launch-shape arithmetic that lives in the kernel headers (`__host__ __device__` helpers), checked on the host: hipcc
compiles tests/c/host_rules.cpp (no GPU needed to build or run it)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_fold_rule_group_stride_and_k3_lds(tmp_path):
    exe = tmp_path / "host_rules"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "rtlamr_amd", "csrc"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "host_rules.cpp"), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    want = [
        "fold 2049 1 2048 = 1",      # a GiB of scm: 2048 tiles + the history tile, 2048 slots -> fold
        "fold 2048 1 2048 = 0",
        "fold 2 1 2048 = 0",         # a lone block: two lists side by side
        "fold 2 4 2048 = 0",
        "fold 8193 4 2048 = 1",      # 4 GiB of "all": 32772 lists = 16 rounds + 4
        "fold 4097 1 2048 = 1",
        "fold 1025 2 2048 = 1",      # 2050 lists
        "fold 1 1 2048 = 0",
        "fold 2049 1 1024 = 1",
        "fold 1564 1 2048 = 0",
        "stride 32 groups(2049) 33 groups(64) 1 groups(65) 2",
        "lds scm72 plain 2576 validated 3084",          # 5 rows x 128 words + 4 | 257 x 12 bytes
        "lds idm72 plain 14352 validated 14352",        # 14 rows x 256 words + 4
    ]
    assert out == want
