"""oracle/go2cxx (the Go -> C++ translator behind oracle/_ref) held to the Go specification on programs that have
nothing to do with the reference: tests/go2cxx/semantics/semantics.go states, next to every function, the result the
language prescribes ("Want ..."); this test translates the file, compiles it and compares.  A translator that only
worked for one program -- or that let C's integer promotion, shift or aliasing rules leak through -- fails here."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G2C = os.path.join(ROOT, "oracle", "go2cxx")
MOD = os.path.join(ROOT, "tests", "go2cxx")
SRC = os.path.join(MOD, "semantics", "semantics.go")


def wants():
    """{function: prescribed result}: every `Want "..."` comment belongs to the next exported func X() string."""
    text = open(SRC, encoding="utf-8").read()
    out = {}
    for m in re.finditer(r'Want "((?:[^"\\]|\\.)*)"\.', text):
        f = re.compile(r"^func ([A-Z]\w*)\(\) string", re.M).search(text, m.end())
        out[f.group(1)] = m.group(1).replace('\\"', '"').replace("\\\\", "\\")
    return out


@pytest.fixture(scope="module")
def binary(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("go2cxx")
    names = sorted(wants())
    assert len(names) >= 22
    hpp = tmp / "semantics_gen.hpp"
    cmd = [sys.executable, os.path.join(G2C, "go2cxx.py"), "--module-root", MOD, "-o", str(hpp)]
    for n in names:
        cmd += ["--root", f"semantics.{n}"]
    subprocess.check_call(cmd)
    main = tmp / "main.cpp"
    main.write_text('#include "semantics_gen.hpp"\nint main() {\n' + "".join(
        f'    std::printf("{n}=%s\\n", P_semantics::{n}().str().c_str());\n' for n in names) + "    return 0;\n}\n")
    exe = tmp / "semantics"
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-w", f"-I{G2C}", f"-I{tmp}", str(main), "-o", str(exe),
                           "-lpthread"])
    return str(exe)


def test_translated_programs_give_what_the_go_specification_prescribes(binary):
    got = dict(line.split("=", 1) for line in subprocess.check_output([binary]).decode("utf-8").splitlines())
    want = wants()
    assert got.keys() == want.keys()
    bad = {k: (got[k], want[k]) for k in want if got[k] != want[k]}
    assert not bad, bad


def test_translator_names_nothing_of_the_reference():
    """The translator and its run time know Go, not rtlamr: no identifier of the reference's packages appears in them."""
    words = ("rtlamr", "Decoder", "Quantized", "preamble", "Preamble", "scm", "idm", "r900", "MagLUT", "sIdx", "csum",
             "Manchester", "decode.go")
    for f in ("go2cxx.py", "goparse.py", "gort.hpp"):
        text = open(os.path.join(G2C, f), encoding="utf-8").read()
        body = text.split('"""', 2)[2] if f.endswith(".py") else text.split("#pragma once", 1)[1]   # module docstrings may say what it is FOR
        hits = [w for w in words if w in body]
        assert not hits, (f, hits)


def test_unsupported_constructs_fail_loudly(tmp_path):
    """Outside the subset the translator refuses (non-zero exit, a message naming file and line) -- it never guesses."""
    mod = tmp_path / "m"
    (mod / "p").mkdir(parents=True)
    (mod / "go.mod").write_text("module example.test/m\n\ngo 1.21\n")
    cases = {
        "select": "package p\nfunc F(c chan int) int {\n\tselect {\n\tcase v := <-c:\n\t\treturn v\n\t}\n}\n",
        "typeswitch": "package p\ntype I interface{ M() }\nfunc F(x I) int {\n\tswitch x.(type) {\n\tdefault:\n\t\treturn 1\n\t}\n}\n",
        "fallthrough": "package p\nfunc F(x int) int {\n\tswitch x {\n\tcase 1:\n\t\tfallthrough\n\tdefault:\n\t\treturn 1\n\t}\n}\n",
        "goto": "package p\nfunc F() int {\nL:\n\tgoto L\n}\n",
    }
    for name, src in cases.items():
        (mod / "p" / "p.go").write_text(src)
        r = subprocess.run([sys.executable, os.path.join(G2C, "go2cxx.py"), "--module-root", str(mod), "--root", "p.F", "-o",
                            str(tmp_path / "o.hpp")], capture_output=True, text=True)
        assert r.returncode != 0 and "p.go" in r.stderr, (name, r.stderr)
