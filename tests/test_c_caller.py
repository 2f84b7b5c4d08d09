"""The boundary seen from C (what cgo is underneath): tests/c/abi_caller.c is compiled with gcc -std=c99 against
include/amrdemod.h and linked to libamrdemod.so.  CPU: it builds, links and fails loudly without a device.
GPU: its hits, packets and bitstream equal the oracle's on the same synthetic stream."""
import os
import subprocess

import numpy as np
import pytest

from rtlamr_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKT = bytes.fromhex("f953026101b3360c4105d005")


def _build(tmp_path):
    exe = str(tmp_path / "abi_caller")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_caller.c"), "-L", _lib.CSRC, "-lamrdemod",
                           f"-Wl,-rpath,{_lib.CSRC}", "-o", exe])
    return exe


def _fnv(h, data: bytes) -> int:
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_c_caller_builds_and_refuses_without_device(tmp_path, amr_lib):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "72", "8", "1"], capture_output=True, text=True)
    if r.returncode == 0:      # a GPU is present (CPU suite run on the GPU box)
        assert "device hits=" in r.stdout
    else:
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("chip,n_blocks,seed", [(72, 40, 9), (32, 64, 4)])
def test_c_caller_matches_oracle(tmp_path, chip, n_blocks, seed):
    from tests import util
    exe = _build(tmp_path)
    r = subprocess.run([exe, str(chip), str(n_blocks), str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(kv.split("=") for line in r.stdout.splitlines() for kv in line.split()[1:] if "=" in kv)
    lines = r.stdout.splitlines()
    assert lines[3] == f"short_input={_lib.AMR_EINVAL} null_handle={_lib.AMR_EINVAL}"
    # the same stream on the host, through the oracle
    bs = int(lines[0].split("bs=")[1].split()[0])
    iq = synth.noise(n_blocks * bs, seed)
    synth.plant(iq, [synth.Packet(bs * 3 + 17, PKT, 96, 30, -26),
                     synth.Packet(bs * (n_blocks // 2) - 500, PKT, 96, -30, 26)], chip)
    _, q, hits, pk = util.oracle_run(["scm"], chip, iq)
    h = 14695981039346656037
    for (pid, blk, idx), p in zip(hits, pk):
        h = _fnv(h, int(blk).to_bytes(8, "little") + int(idx).to_bytes(4, "little") + p[:12].tobytes())
    dev = lines[1].split()
    assert dev[1] == f"hits={len(hits)}" and len(hits) > 50
    assert dev[2] == f"hit_hash={h:016x}"
    assert dev[3] == f"q_hash={_fnv(14695981039346656037, q.tobytes()):016x}"
    assert lines[2].split()[1:] == dev[1:3]        # host input in two uneven batches: same hits
