#!/usr/bin/env python
"""Per-kernel resources of the built library (code-object metadata of every gfx950 code object in its fat binary):
VGPRs, SGPRs, LDS, scratch.  usage: python tools/kernel_resources.py [libamrdemod.so] [regex on the demangled name]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

so = sys.argv[1] if len(sys.argv) > 1 else "rtlamr_amd/csrc/libamrdemod.so"
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
rows = set()
with tempfile.TemporaryDirectory() as tmp:
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    blob = open(fat, "rb").read()
    pos, n = 0, 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        (count,) = struct.unpack_from("<Q", blob, pos + 24)
        p = pos + 32
        for _ in range(count):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" not in triple or size == 0:
                continue
            co = os.path.join(tmp, f"co{n}.o")
            n += 1
            open(co, "wb").write(blob[pos + off:pos + off + size])
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", notes, re.S):
                body = m.group(2)
                g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", body).group(1)) if re.search(rf"\.{k}:\s+(\d+)", body) else -1
                rows.add((m.group(1), g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                          g("private_segment_fixed_size"), g("vgpr_spill_count")))
        pos += len(MAGIC)
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in sorted(rows)), capture_output=True, text=True).stdout.split("\n")
for r, dem in zip(sorted(rows), names):
    if pat is None or pat.search(dem):
        print(f"{dem[:100]:100s} vgpr {r[1]:4d} sgpr {r[2]:4d} lds {r[3]:6d} scratch {r[4]:5d} spilled_vgprs {r[5]}")
print(f"{len(rows)} kernels in {so}; with scratch (private_segment_fixed_size > 0): {sum(1 for r in rows if r[4] > 0)}")
