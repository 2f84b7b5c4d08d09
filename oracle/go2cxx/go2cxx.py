#!/usr/bin/env python3
"""go2cxx -- a generic, syntax-directed translator from a subset of Go to C++20 (run time: gort.hpp).

TEST INFRASTRUCTURE (oracle/).  Purpose: derive oracle/_ref/ MECHANICALLY from the reference's own Go sources, read
where they lie (never copied into the repository), because the image has no Go toolchain.  The product never loads the
result; only tests/ do.

What it is NOT: a Go compiler.  It performs no type checking of its own -- it emits C++ whose types (gort.hpp: Num<T>,
UInt/UFloat untyped constants, Slice, String, Map, Chan, interface wrappers) make the C++ compiler enforce Go's typing
rules, so a construct outside the subset fails to translate or to compile instead of silently changing meaning.
It knows Go syntax and scoping only: there is no identifier, file name or line number of any particular program in this
file; what to translate is given on the command line (module root, root declarations), dependencies are found by
reachability over the parsed packages.

Subset (anything else: `Unsupported`, loudly): packages and imports inside one module + the standard-library names
gort.hpp provides; struct / interface / defined / alias types, methods with value and pointer receivers (value
receivers copy), embedded structs (as C++ bases) and embedded interfaces (flattened); functions with multiple and
named results, variadics to package functions, generic functions with type-parameter lists; every statement except
select, type switches, fallthrough and goto; expressions incl. composite literals, slices of slices/arrays/strings,
maps with comma-ok, channels, closures (captures are BY VALUE: a closure that assigns to a captured variable does not
compile), goroutines (threads), defer (run at scope exit, arguments captured at the defer statement).
Known deviations, none silent where it matters: map iteration is in key order (Go: unspecified); untyped constants are
held in 128-bit integers / doubles (Go: arbitrary precision); a non-constant shift of an untyped constant is an int.

usage: go2cxx.py --module-root DIR [--root pkgdir.Name | pkgdir.Type.Method]... -o out.hpp
"""
from __future__ import annotations

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from goparse import GoSyntaxError, Node, parse_source  # noqa: E402


class Unsupported(Exception):
    pass


CXX_RESERVED = {
    "alignas", "alignof", "and", "and_eq", "asm", "auto", "bitand", "bitor", "bool", "catch", "char", "class", "compl",
    "concept", "const_cast", "consteval", "constexpr", "constinit", "co_await", "co_return", "co_yield", "decltype",
    "delete", "do", "double", "dynamic_cast", "enum", "explicit", "export", "extern", "float", "friend", "inline", "int",
    "long", "mutable", "namespace", "new", "noexcept", "not", "not_eq", "nullptr", "operator", "or", "or_eq", "private",
    "protected", "public", "register", "reinterpret_cast", "requires", "short", "signed", "sizeof", "static",
    "static_assert", "static_cast", "template", "this", "thread_local", "throw", "try", "typedef", "typeid", "typename",
    "union", "unsigned", "using", "virtual", "void", "volatile", "wchar_t", "while", "xor", "xor_eq", "errno", "stdin",
    "stdout", "stderr", "NULL", "EOF", "main",
}

UNIVERSE_TYPES = {
    "bool": "bool", "string": "go::String", "error": "go::error", "int": "go::int_", "int8": "go::int8",
    "int16": "go::int16", "int32": "go::int32", "int64": "go::int64", "uint": "go::uint", "uint8": "go::uint8",
    "uint16": "go::uint16", "uint32": "go::uint32", "uint64": "go::uint64", "uintptr": "go::uintptr", "byte": "go::byte",
    "rune": "go::rune", "float32": "go::float32", "float64": "go::float64",
}
UNIVERSE_FUNCS = {"append", "cap", "close", "copy", "delete", "len", "make", "new", "panic", "print", "println", "min",
                  "max", "clear", "recover", "complex", "real", "imag"}
UNIVERSE_VALUES = {"true": "true", "false": "false", "nil": "go::nil"}


def mangle(name: str) -> str:
    out = "".join(c if (c.isascii() and (c.isalnum() or c == "_")) else f"_u{ord(c):04x}" for c in name)
    if out in CXX_RESERVED:
        out += "_"
    return out


def cstring(b: bytes) -> str:
    out = []
    for c in b:
        if c == 0x22 or c == 0x5C:
            out.append("\\" + chr(c))
        elif 0x20 <= c < 0x7F and c != 0x3F:
            out.append(chr(c))
        else:
            out.append("\\%03o" % c)
    return '"' + "".join(out) + '"'


# ----------------------------------------------------------------------------------------------------- program model
class Package:
    def __init__(self, path, directory, name):
        self.path, self.dir, self.name = path, directory, name
        self.files = []
        self.types, self.funcs, self.vars, self.consts = {}, {}, {}, {}
        self.methods = {}  # type name -> {method name: funcdecl}
        self.bad = {}
        self.file_of = {}  # id(decl) -> file node (for its imports)

    @property
    def ns(self):
        return "P_" + mangle(self.name)


class Program:
    def __init__(self, module_root):
        self.root = os.path.abspath(module_root)
        self.module = None
        gomod = os.path.join(self.root, "go.mod")
        if os.path.exists(gomod):
            for line in open(gomod):
                if line.startswith("module "):
                    self.module = line.split()[1].strip()
        self.packages = {}  # import path -> Package

    def import_dir(self, path):
        if self.module and (path == self.module or path.startswith(self.module + "/")):
            return os.path.join(self.root, path[len(self.module):].lstrip("/"))
        return None

    def load(self, path):
        """Parse the package at import path `path` (None for packages outside the module: the run time provides them)."""
        if path in self.packages:
            return self.packages[path]
        d = self.import_dir(path)
        if d is None or not os.path.isdir(d):
            return None
        files = sorted(f for f in os.listdir(d) if f.endswith(".go") and not f.endswith("_test.go"))
        pkg = None
        for f in files:
            src = open(os.path.join(d, f), encoding="utf-8").read()
            node = parse_source(src, os.path.join(d, f))
            if pkg is None:
                pkg = Package(path, d, node.package)
                self.packages[path] = pkg
            elif node.package != pkg.name:
                continue
            pkg.files.append(node)
            for decl in node.decls:
                pkg.file_of[id(decl)] = node
                k = decl.kind
                if k == "typedecl":
                    pkg.types[decl.name] = decl
                elif k == "funcdecl":
                    if decl.recv is None:
                        if decl.name != "init" and decl.name != "_":
                            pkg.funcs[decl.name] = decl
                    else:
                        t = decl.recv.type
                        if t.kind == "tptr":
                            t = t.elem
                        pkg.methods.setdefault(t.name, {})[decl.name] = decl
                elif k == "vardecl":
                    for nm in decl.names:
                        pkg.vars[nm] = decl
                elif k == "constdecl":
                    for nm in decl.names:
                        pkg.consts[nm] = decl
                elif k == "bad":
                    pkg.bad[decl.name] = decl
        if pkg is None:
            return None
        for node in pkg.files:
            for imp in node.imports:
                self.load(imp.path)
        return pkg

    def by_dir(self, rel):
        path = self.module + ("/" + rel if rel not in ("", ".") else "") if self.module else rel
        p = self.load(path)
        if p is None:
            raise SystemExit(f"go2cxx: no Go package in {os.path.join(self.root, rel)}")
        return p


def walk(node, fn):
    """Pre-order walk over every Node reachable from `node` (lists and tuples of nodes included)."""
    if isinstance(node, Node):
        fn(node)
        for k, v in node.__dict__.items():
            if k not in ("kind", "line"):
                walk(v, fn)
    elif isinstance(node, (list, tuple)):
        for v in node:
            walk(v, fn)


def import_names(filenode):
    out = {}
    for imp in filenode.imports:
        nm = imp.alias or imp.path.rsplit("/", 1)[-1]
        if nm != "_":
            out[nm] = imp.path
    return out


# ----------------------------------------------------------------------------------------------------- reachability
class Reach:
    def __init__(self, prog):
        self.prog = prog
        self.keys = set()      # ('type'|'func'|'var'|'const', pkgpath, name) / ('method', pkgpath, type, name)
        self.work = []
        self.method_names = set()

    def add(self, key):
        if key not in self.keys:
            self.keys.add(key)
            self.work.append(key)

    def add_name(self, pkg, name):
        if name in pkg.types:
            self.add(("type", pkg.path, name))
        elif name in pkg.funcs:
            self.add(("func", pkg.path, name))
        elif name in pkg.vars:
            self.add(("var", pkg.path, name))
        elif name in pkg.consts:
            self.add(("const", pkg.path, name))
        elif name in pkg.bad:
            b = pkg.bad[name]
            raise Unsupported(f"{pkg.path}.{name} is needed but did not parse: {b.msg}")

    def scan(self, pkg, decl):
        imports = import_names(pkg.file_of[id(decl)])

        def visit(n):
            if n.kind == "ident":
                self.add_name(pkg, n.name)
            elif n.kind == "tname":
                if n.pkg is None:
                    self.add_name(pkg, n.name)
                elif n.pkg in imports:
                    q = self.prog.load(imports[n.pkg])
                    if q is not None:
                        self.add_name(q, n.name)
            elif n.kind == "selector":
                self.method_names.add(n.sel)
                if n.x.kind == "ident" and n.x.name in imports:
                    q = self.prog.load(imports[n.x.name])
                    if q is not None:
                        self.add_name(q, n.sel)
            elif n.kind == "imethod":
                self.method_names.add(n.name)
        walk(decl, visit)

    def run(self):
        while True:
            while self.work:
                key = self.work.pop()
                pkg = self.prog.packages[key[1]]
                if key[0] == "method":
                    self.scan(pkg, pkg.methods[key[2]][key[3]])
                else:
                    table = {"type": pkg.types, "func": pkg.funcs, "var": pkg.vars, "const": pkg.consts}[key[0]]
                    self.scan(pkg, table[key[2]])
            grew = False
            for key in list(self.keys):
                if key[0] != "type":
                    continue
                pkg = self.prog.packages[key[1]]
                for mname in pkg.methods.get(key[2], {}):
                    mk = ("method", key[1], key[2], mname)
                    if mname in self.method_names and mk not in self.keys:
                        self.add(mk)
                        grew = True
            if not grew and not self.work:
                return


# ----------------------------------------------------------------------------------------------------- emitter
class Scope:
    def __init__(self, parent=None, func_boundary=False):
        self.parent = parent
        self.names = {}  # go name -> (kind, cpp)
        self.func_boundary = func_boundary

    def lookup(self, name):
        s = self
        while s is not None:
            if name in s.names:
                return s.names[name]
            s = s.parent
        return None


class FuncCtx:
    def __init__(self, sig, named_results, result_type):
        self.sig = sig
        self.named = named_results
        self.result_type = result_type
        self.heap = set()        # go names whose address is taken with &: live on the heap
        self.break_stack = []    # ('loop', label) | ('switch', end_label)


class Emitter:
    def __init__(self, prog, reach):
        self.prog = prog
        self.reach = reach
        self.pkg = None
        self.file = None
        self.scope = None
        self.fn = None
        self.tmp = 0
        self.iota = None
        self.embedded_names = set()

    # ---------- helpers
    def fresh(self, base="_t"):
        self.tmp += 1
        return f"{base}{self.tmp}"

    def unsupported(self, node, what):
        fn = self.file.filename if self.file is not None else "?"
        raise Unsupported(f"{fn}:{getattr(node, 'line', '?')}: {what}")

    def push(self, func_boundary=False):
        self.scope = Scope(self.scope, func_boundary)

    def pop(self):
        self.scope = self.scope.parent

    def declare(self, name, kind="local"):
        if name == "_":
            return self.fresh("_blank")
        cpp = mangle(name)
        if self.scope.lookup(name) is not None or name in UNIVERSE_TYPES or name in UNIVERSE_FUNCS:
            cpp = f"{cpp}_{self.fresh('v')}"
        self.scope.names[name] = (kind, cpp)
        return cpp

    def package_scope(self, pkg, filenode):
        s = Scope()
        for nm in pkg.types:
            s.names[nm] = ("type", f"{pkg.ns}::{mangle(nm)}")
        for nm in pkg.funcs:
            s.names[nm] = ("func", f"{pkg.ns}::{mangle(nm)}")
        for nm in pkg.vars:
            s.names[nm] = ("var", f"{pkg.ns}::{mangle(nm)}")
        for nm in pkg.consts:
            s.names[nm] = ("const", f"{pkg.ns}::{mangle(nm)}")
        fs = Scope(s)
        for nm, path in import_names(filenode).items():
            q = self.prog.load(path)
            fs.names[nm] = ("pkg", q.ns if q is not None else "P_" + mangle(path.rsplit("/", 1)[-1]), path)
        return fs

    def enter_decl(self, pkg, decl):
        self.pkg = pkg
        self.file = pkg.file_of[id(decl)]
        self.scope = self.package_scope(pkg, self.file)

    def resolve_pkg(self, node):
        """If `node` is an identifier naming an imported package return (namespace, import path)."""
        if node.kind == "ident":
            e = self.scope.lookup(node.name)
            if e is not None and e[0] == "pkg":
                return e[1], e[2]
        return None

    def type_decl_of(self, tnode):
        """The declaration a type NAME refers to, following aliases/definitions one step: (package, typedecl) or None."""
        if tnode.kind != "tname":
            return None
        if tnode.pkg is None:
            e = self.scope.lookup(tnode.name)
            if e is not None and e[0] == "type" and tnode.name in self.pkg.types:
                return self.pkg, self.pkg.types[tnode.name]
            return None
        e = self.scope.lookup(tnode.pkg)
        if e is not None and e[0] == "pkg":
            q = self.prog.load(e[2])
            if q is not None and tnode.name in q.types:
                return q, q.types[tnode.name]
        return None

    def underlying(self, tnode):
        """Underlying type node of `tnode` and the package whose scope its names live in."""
        pkg = self.pkg
        seen = 0
        while tnode.kind == "tname":
            saved = self.pkg
            found = self.type_decl_of(tnode) if pkg is self.pkg else self.with_pkg(pkg, lambda: self.type_decl_of(tnode))
            self.pkg = saved
            if found is None:
                return tnode, pkg
            pkg, decl = found
            tnode = decl.type
            seen += 1
            if seen > 50:
                break
        return tnode, pkg

    def with_pkg(self, pkg, fn):
        saved = (self.pkg, self.file, self.scope)
        self.pkg = pkg
        self.file = pkg.files[0]
        self.scope = self.package_scope(pkg, self.file)
        try:
            return fn()
        finally:
            self.pkg, self.file, self.scope = saved

    def in_decl_scope(self, pkg, decl, fn):
        saved = (self.pkg, self.file, self.scope)
        self.enter_decl(pkg, decl)
        try:
            return fn()
        finally:
            self.pkg, self.file, self.scope = saved

    # ---------- types
    def ty(self, t):
        k = t.kind
        if k == "tname":
            if t.pkg is None:
                e = self.scope.lookup(t.name)
                if e is not None:
                    if e[0] in ("type", "tparam"):
                        return e[1]
                    self.unsupported(t, f"{t.name} is not a type")
                if t.name in UNIVERSE_TYPES:
                    return UNIVERSE_TYPES[t.name]
                if t.name == "any":
                    self.unsupported(t, "the empty interface (any)")
                self.unsupported(t, f"unknown type {t.name}")
            e = self.scope.lookup(t.pkg)
            if e is None or e[0] != "pkg":
                self.unsupported(t, f"{t.pkg} is not a package")
            return f"{e[1]}::{mangle(t.name)}"
        if k == "tptr":
            return self.ty(t.elem) + "*"
        if k == "tslice":
            return f"go::Slice<{self.ty(t.elem)}>"
        if k == "tarray":
            if t.len is None:
                self.unsupported(t, "[...]T outside a composite literal")
            return f"go::Array<{self.ty(t.elem)}, (long long)({self.ex(t.len)}).v>"
        if k == "tmap":
            return f"go::Map<{self.ty(t.key)}, {self.ty(t.elem)}>"
        if k == "tchan":
            return f"go::Chan<{self.ty(t.elem)}>"
        if k == "tfunc":
            ps = ", ".join(self.param_type(p) for p in t.sig.params)
            return f"std::function<{self.result_type(t.sig)}({ps})>"
        if k == "tinterface":
            self.unsupported(t, "anonymous interface types (incl. interface{})")
        if k == "tstruct":
            self.unsupported(t, "anonymous struct types")
        self.unsupported(t, f"type form {k}")

    def param_type(self, p):
        if p.variadic:
            return f"go::Slice<{self.ty(p.type)}>"
        return self.ty(p.type)

    def result_type(self, sig):
        rs = sig.results
        if not rs:
            return "void"
        if len(rs) == 1:
            return self.ty(rs[0].type)
        return "std::tuple<" + ", ".join(self.ty(r.type) for r in rs) + ">"

    # ---------- expressions
    def ex(self, n, lvalue=False):
        k = n.kind
        if k == "intlit":
            v = n.value
            if v < 2 ** 63:
                return f"go::UInt({v}LL)"
            if v < 2 ** 64:
                return f"go::UInt::big((__int128){v}ULL)"
            self.unsupported(n, "integer constant beyond 64 bits")
        if k == "floatlit":
            return f"go::UFloat({n.value})"
        if k == "runelit":
            return f"go::UInt({n.value}LL)"
        if k == "stringlit":
            return f"go::String({cstring(n.value)}, {len(n.value)})"
        if k == "ident":
            return self.ident(n)
        if k == "paren":
            return "(" + self.ex(n.x, lvalue) + ")"
        if k == "selector":
            p = self.resolve_pkg(n.x)
            if p is not None:
                return f"{p[0]}::{mangle(n.sel)}"
            if n.sel in self.embedded_names:
                return f"go_sel_{mangle(n.sel)}(go::deref({self.ex(n.x, True)}))"
            return f"go::deref({self.ex(n.x, True)}).{mangle(n.sel)}"
        if k == "index":
            inner = f"{self.ex(n.x, True)}[{self.ex(n.index)}]"
            return inner if lvalue else f"go::rv({inner})"
        if k == "sliceexpr":
            lo = self.ex(n.lo) if n.lo is not None else "go::nobound"
            hi = self.ex(n.hi) if n.hi is not None else "go::nobound"
            if n.three:
                return f"go::slice({self.ex(n.x, True)}, {lo}, {hi}, {self.ex(n.max)})"
            return f"go::slice({self.ex(n.x, True)}, {lo}, {hi})"
        if k == "call":
            return self.call(n)
        if k == "unary":
            if n.op == "^":
                return f"go::bitnot({self.ex(n.x)})"
            return f"({n.op}{self.ex(n.x)})"
        if k == "binary":
            if n.op == "&^":
                return f"go::andnot({self.ex(n.x)}, {self.ex(n.y)})"
            return f"({self.ex(n.x)} {n.op} {self.ex(n.y)})"
        if k == "deref":
            return f"go::star({self.ex(n.x)})"
        if k == "addr":
            if n.x.kind == "composite":
                t = self.composite_type(n.x)
                return f"(new {t}({self.composite(n.x)}))"
            return f"go::addr({self.ex(n.x, True)})"
        if k == "recv":
            return f"({self.ex(n.x)}).recv()"
        if k == "funclit":
            return self.funclit(n)
        if k == "composite":
            return self.composite(n)
        if k == "typeassert":
            if n.type is None:
                self.unsupported(n, "x.(type) outside a type switch")
            return f"go::type_assert<{self.ty(n.type)}>({self.ex(n.x)})"
        if k == "typeexpr":
            self.unsupported(n, "a type used as a value")
        self.unsupported(n, f"expression form {k}")

    def ident(self, n):
        name = n.name
        if name == "_":
            self.unsupported(n, "blank identifier used as a value")
        e = self.scope.lookup(name)
        if e is not None:
            if e[0] == "pkg":
                self.unsupported(n, f"package {name} used as a value")
            return e[1]
        if name in UNIVERSE_VALUES:
            return UNIVERSE_VALUES[name]
        if name == "iota":
            if self.iota is None:
                self.unsupported(n, "iota outside a constant declaration")
            return f"go::UInt({self.iota}LL)"
        if name in UNIVERSE_TYPES:
            return UNIVERSE_TYPES[name]
        if name in UNIVERSE_FUNCS:
            return "go::" + mangle(name)
        self.unsupported(n, f"undefined: {name}")

    def as_type(self, n):
        """If expression node `n` denotes a type return its C++ spelling, else None."""
        if n.kind == "typeexpr":
            return self.ty(n.type)
        if n.kind == "paren":
            return self.as_type(n.x)
        if n.kind == "ident":
            e = self.scope.lookup(n.name)
            if e is not None:
                return e[1] if e[0] in ("type", "tparam") else None
            return UNIVERSE_TYPES.get(n.name)
        if n.kind == "selector":
            p = self.resolve_pkg(n.x)
            if p is not None:
                q = self.prog.load(p[1])
                if q is not None and n.sel in q.types:
                    return f"{p[0]}::{mangle(n.sel)}"
            return None
        if n.kind == "deref":
            inner = self.as_type(n.x)
            return inner + "*" if inner is not None else None
        return None

    def type_node_of(self, n):
        """The type NODE an expression in type position denotes (for make / composite literals)."""
        if n.kind == "typeexpr":
            return n.type
        if n.kind == "paren":
            return self.type_node_of(n.x)
        if n.kind == "ident":
            return Node("tname", n.line, pkg=None, name=n.name)
        if n.kind == "selector" and n.x.kind == "ident":
            return Node("tname", n.line, pkg=n.x.name, name=n.sel)
        self.unsupported(n, "expected a type")

    def args(self, nodes):
        return ", ".join(self.ex(a) for a in nodes)

    def callee_sig(self, fun):
        """Signature of a package-level function named directly (for variadic packing), else None."""
        if fun.kind == "ident":
            e = self.scope.lookup(fun.name)
            if e is not None and e[0] == "func" and fun.name in self.pkg.funcs:
                return self.pkg.funcs[fun.name].sig, self.pkg, self.pkg.funcs[fun.name]
        if fun.kind == "selector":
            p = self.resolve_pkg(fun.x)
            if p is not None:
                q = self.prog.load(p[1])
                if q is not None and fun.sel in q.funcs:
                    return q.funcs[fun.sel].sig, q, q.funcs[fun.sel]
        return None

    def call(self, n):
        fun = n.fun
        if fun.kind == "ident" and self.scope.lookup(fun.name) is None and fun.name in UNIVERSE_FUNCS:
            return self.builtin(n, fun.name)
        t = self.as_type(fun)
        if t is not None:
            if len(n.args) != 1:
                self.unsupported(n, "conversion with other than one operand")
            return f"go::conv<{t}>({self.ex(n.args[0])})"
        cs = self.callee_sig(fun)
        if cs is not None and cs[0].params and cs[0].params[-1].variadic and not n.ellipsis:
            sig, q, decl = cs
            nfixed = len(sig.params) - 1
            et = self.in_decl_scope(q, decl, lambda: self.ty(sig.params[-1].type))
            fixed = [self.ex(a) for a in n.args[:nfixed]]
            rest = ", ".join(f"{et}({self.ex(a)})" for a in n.args[nfixed:])
            packed = f"go::Slice<{et}>::of({{{rest}}})" if rest else f"go::Slice<{et}>()"
            return f"{self.ex(fun)}({', '.join(fixed + [packed])})"
        return f"{self.ex(fun)}({self.args(n.args)})"

    def builtin(self, n, name):
        a = n.args
        if name == "make":
            tn = self.type_node_of(a[0])
            u, _ = self.underlying(tn)
            t = self.ty(tn)
            rest = self.args(a[1:])
            if u.kind == "tslice":
                base = f"go::make_slice<{self.elem_type(tn, u)}>({rest})"
                return base if tn.kind == "tslice" else f"{t}({base})"
            if u.kind == "tmap":
                base = f"go::Map<{self.in_under(tn, u, u.key)}, {self.in_under(tn, u, u.elem)}>::make()"
                return base if tn.kind == "tmap" else f"{t}({base})"
            if u.kind == "tchan":
                cap = f"go::to_i64({rest})" if rest else ""
                return f"go::Chan<{self.in_under(tn, u, u.elem)}>::make({cap})"
            self.unsupported(n, "make of this type")
        if name == "new":
            return f"(new {self.ty(self.type_node_of(a[0]))}{{}})"
        if name == "append":
            if n.ellipsis:
                if len(a) != 2:
                    self.unsupported(n, "append(s, more, spread...)")
                return f"go::append_spread({self.ex(a[0])}, {self.ex(a[1])})"
            return f"go::append({self.args(a)})"
        if name == "delete":
            return f"go::delete_({self.ex(a[0])}, {self.ex(a[1])})"
        if name in ("len", "cap", "copy", "close", "panic", "min", "max"):
            return f"go::{name}({self.args(a)})"
        self.unsupported(n, f"builtin {name}")

    def in_under(self, tn, u, sub):
        """C++ type of `sub`, a part of underlying type `u` of `tn` (whose names may live in another package)."""
        if tn is u:
            return self.ty(sub)
        _, pkg = self.underlying(tn)
        return self.with_pkg(pkg, lambda: self.ty(sub))

    def elem_type(self, tn, u):
        return self.in_under(tn, u, u.elem)

    def composite_type(self, n):
        return self.ty(self.type_node_of(n.type)) if n.type is not None else None

    def composite(self, n, implied=None):
        """implied: (type node, package) for elided element types of an enclosing literal."""
        if n.type is None:
            if implied is None:
                self.unsupported(n, "composite literal without a type")
            tn, owner = implied
            if owner is not self.pkg:
                return self.with_pkg(owner, lambda: self.composite(Node("composite", n.line, type=Node("typeexpr", n.line, type=tn), elems=n.elems)))
            n = Node("composite", n.line, type=Node("typeexpr", n.line, type=tn), elems=n.elems)
        tn = self.type_node_of(n.type)
        if tn.kind == "tarray" and tn.len is None:
            tn = Node("tarray", tn.line, len=Node("intlit", tn.line, value=len(n.elems)), elem=tn.elem)
        t = self.ty(tn)
        u, upkg = self.underlying(tn)

        def elem(v, sub):
            if v.kind == "composite" and v.type is None:
                return self.composite(v, (sub, upkg))
            return self.ex(v)

        if u.kind == "tstruct":
            if not n.elems:
                return f"{t}{{}}"
            if n.elems[0][0] is None:
                return f"{t}{{{', '.join(elem(v, None) for _, v in n.elems)}}}"
            embedded = {f.name: f for f in u.fields if f.embedded and f.type.kind == "tname"}
            v = self.fresh("_lit")
            body = [f"{t} {v}{{}};"]
            for key, val in n.elems:
                if key is None or key.kind != "ident":
                    self.unsupported(n, "mixed keyed and positional fields")
                ftype = next((f.type for f in u.fields if f.name == key.name), None)
                if key.name in embedded:
                    bt = self.in_under(tn, u, embedded[key.name].type)
                    body.append(f"static_cast<{bt}&>({v}) = {elem(val, ftype)};")
                else:
                    body.append(f"{v}.{mangle(key.name)} = {elem(val, ftype)};")
            body.append(f"return {v};")
            return "[&]{ " + " ".join(body) + " }()"
        if u.kind in ("tslice", "tarray"):
            if any(k is not None for k, _ in n.elems):
                self.unsupported(n, "indexed elements in a slice / array literal")
            et = self.in_under(tn, u, u.elem)
            items = ", ".join(f"{et}({elem(v, u.elem)})" for _, v in n.elems)
            if u.kind == "tslice":
                base = f"go::Slice<{et}>::of({{{items}}})" if n.elems else f"go::Slice<{et}>::make(0, 0)"
                return base if tn.kind == "tslice" else f"{t}({base})"
            base = f"go::Array<{et}, {len(n.elems)}>{{{{{items}}}}}" if tn.kind == "tarray" else None
            if base is None:
                self.unsupported(n, "literal of a defined array type")
            return base
        if u.kind == "tmap":
            kt, vt = self.in_under(tn, u, u.key), self.in_under(tn, u, u.elem)
            items = ", ".join(f"{{{kt}({elem(k, u.key)}), {vt}({elem(v, u.elem)})}}" for k, v in n.elems)
            base = f"go::Map<{kt}, {vt}>::of({{{items}}})" if n.elems else f"go::Map<{kt}, {vt}>::make()"
            return base if tn.kind == "tmap" else f"{t}({base})"
        self.unsupported(n, "composite literal of this type")

    def funclit(self, n):
        saved_fn = self.fn
        self.push(func_boundary=True)
        try:
            params = self.param_list(n.sig)
            rt = self.result_type(n.sig)
            self.fn = FuncCtx(n.sig, self.named_results(n.sig), rt)
            self.fn.heap = self.addr_taken(n.body)
            body = self.func_body(n.body)
        finally:
            self.pop()
            self.fn = saved_fn
        # heap variables of the surrounding functions (declared `T& x = *new T`): by reference, i.e. the one heap object
        byref, s = [], self.scope
        while s is not None:
            byref += [cpp for kind, cpp in (e[:2] for e in s.names.values()) if kind == "heap"]
            s = s.parent
        cap = "".join(f", &{c}" for c in dict.fromkeys(byref))
        return f"[={cap}]({params}) -> {rt} {body}"

    # ---------- statements
    def addr_taken(self, body):
        names = set()

        def visit(n):
            if n.kind == "addr" and n.x.kind == "ident":
                names.add(n.x.name)
            if n.kind == "funclit":
                # a function literal shares the variables it uses with its surroundings: the ones it WRITES must be one object
                # for both (the literal is a C++ lambda that copies what it captures; heap variables it takes by reference)
                def inner(m):
                    if m.kind == "assign":
                        for l in m.lhs:
                            if l.kind == "ident":
                                names.add(l.name)
                    elif m.kind == "incdec" and m.x.kind == "ident":
                        names.add(m.x.name)
                walk(n.body, inner)
        walk(body, visit)
        return names

    def param_list(self, sig):
        out = []
        for p in sig.params:
            t = self.param_type(p)
            out.append(f"{t} {self.declare(p.name, 'local')}" if p.name else t)
        return ", ".join(out)

    def named_results(self, sig):
        if sig.results and sig.results[0].name is not None:
            return [(r.name, r.type) for r in sig.results]
        return None

    def func_body(self, body, prologue=()):
        lines = ["{"] + list(prologue)
        if self.fn.named:
            cpps = []
            for name, t in self.fn.named:
                cpp = self.declare(name)
                cpps.append(cpp)
                lines.append(f"{self.ty(t)} {cpp}{{}};")
            self.fn.named_cpp = cpps
        if self.uses_defer(body):
            lines.append("go::Defer _defer;")
        for s in body.stmts:
            lines.append(self.stmt(s))
        lines.append("}")
        return "\n".join(lines)

    def uses_defer(self, body):
        found = []

        def visit(n):
            if n.kind == "defer":
                found.append(n)
        walk([s for s in body.stmts], visit)
        return bool(found)

    def block(self, b):
        self.push()
        try:
            return "{\n" + "\n".join(self.stmt(s) for s in b.stmts) + "\n}"
        finally:
            self.pop()

    def simple_expr_stmt(self, s):
        """A simple statement as ONE C++ expression (for-loop post statements)."""
        if s is None:
            return ""
        if s.kind == "incdec":
            return f"go::{'inc' if s.op == '++' else 'dec'}({self.ex(s.x, True)})"
        if s.kind == "exprstmt":
            return self.ex(s.x)
        if s.kind == "assign" and len(s.lhs) == 1 and len(s.rhs) == 1:
            return self.assign1(s.op, s.lhs[0], self.ex(s.rhs[0]))
        self.unsupported(s, "this statement form in a for-loop post position")

    def assign1(self, op, lhs, rhs):
        if lhs.kind == "ident" and lhs.name == "_":
            return f"(void)({rhs})"
        if op == "&^=":
            return f"go::andnot_assign({self.ex(lhs, True)}, {rhs})"
        return f"{self.ex(lhs, True)} {op} {rhs}"

    def multi_source(self, rhs):
        """C++ expression giving a tuple for a multi-valued right-hand side."""
        if rhs.kind == "index":
            return f"go::lookup({self.ex(rhs, True)})"
        if rhs.kind == "recv":
            return f"({self.ex(rhs.x)}).recv2()"
        if rhs.kind == "typeassert" and rhs.type is not None:
            return f"go::type_assert2<{self.ty(rhs.type)}>({self.ex(rhs.x)})"
        if rhs.kind == "call":
            return self.ex(rhs)
        self.unsupported(rhs, "multi-valued expression of this form")

    def stmt(self, s):
        k = s.kind
        if k == "block":
            return self.block(s)
        if k == "exprstmt":
            return self.ex(s.x) + ";"
        if k == "incdec":
            return self.simple_expr_stmt(s) + ";"
        if k == "send":
            return f"({self.ex(s.ch)}).send({self.ex(s.value)});"
        if k == "assign":
            return self.assign(s)
        if k == "define":
            return self.define(s)
        if k == "declstmt":
            return "\n".join(self.local_decl(d) for d in s.decls)
        if k == "return":
            return self.ret(s)
        if k == "if":
            return self.if_(s)
        if k == "for":
            return self.for_(s, None)
        if k == "forrange":
            return self.forrange(s, None)
        if k == "switch":
            return self.switch(s)
        if k == "labeled":
            if s.stmt is not None and s.stmt.kind == "for":
                return self.for_(s.stmt, s.label)
            if s.stmt is not None and s.stmt.kind == "forrange":
                return self.forrange(s.stmt, s.label)
            self.unsupported(s, "labels on statements other than loops")
        if k in ("break", "continue"):
            return self.jump(s)
        if k == "go":
            return self.go(s)
        if k == "defer":
            return self.defer(s)
        self.unsupported(s, f"statement form {k}")

    def local_var(self, name, ctype, init):
        """Declare local `name`; ctype None = deduce.  Variables whose address is taken live on the heap."""
        heap = name in self.fn.heap
        cpp = self.declare(name, "heap" if heap and name != "_" else "local")
        if name == "_":
            return f"[[maybe_unused]] auto {cpp} = {init};" if init is not None else ""
        if heap:
            if ctype is None:
                return f"auto& {cpp} = *new auto({init});"
            return f"{ctype}& {cpp} = *new {ctype}{{}};" + (f" {cpp} = {init};" if init is not None else "")
        if ctype is None:
            return f"auto {cpp} = {init};"
        return f"{ctype} {cpp}{{}};" if init is None else f"{ctype} {cpp} = {init};"

    def local_decl(self, d):
        if d.kind == "vardecl":
            ctype = self.ty(d.type) if d.type is not None else None
            if d.values is None:
                return "\n".join(self.local_var(nm, ctype, None) for nm in d.names)
            if len(d.values) == len(d.names):
                inits = [self.ex(v) for v in d.values]
                out = []
                for nm, init in zip(d.names, inits):
                    out.append(self.local_var(nm, ctype, init if ctype else f"go::def({init})"))
                return "\n".join(out)
            t = self.fresh()
            out = [f"auto {t} = {self.multi_source(d.values[0])};"]
            for i, nm in enumerate(d.names):
                out.append(self.local_var(nm, ctype, f"std::get<{i}>({t})" if ctype else f"go::def(std::get<{i}>({t}))"))
            return "\n".join(out)
        if d.kind == "constdecl":
            saved, self.iota = self.iota, d.iota
            try:
                out = []
                for nm, v in zip(d.names, d.values):
                    init = self.ex(v)
                    cpp = self.declare(nm)
                    if d.type is not None:
                        out.append(f"const {self.ty(d.type)} {cpp} = {init};")
                    else:
                        out.append(f"const auto {cpp} = {init};")
                return "\n".join(out)
            finally:
                self.iota = saved
        self.unsupported(d, "local type declarations")

    def define(self, s):
        new = [e.name != "_" and e.name not in self.scope.names for e in s.lhs]
        out = []
        if len(s.lhs) == len(s.rhs):
            if len(s.lhs) == 1:
                init = self.ex(s.rhs[0])
                e = s.lhs[0]
                if e.name == "_":
                    return f"(void)({init});"
                if new[0]:
                    return self.local_var(e.name, None, f"go::def({init})")
                return f"{self.ex(e, True)} = {init};"
            tmps = []
            for r in s.rhs:
                t = self.fresh()
                tmps.append(t)
                out.append(f"auto {t} = go::def({self.ex(r)});")
            for e, t, isnew in zip(s.lhs, tmps, new):
                if e.name == "_":
                    continue
                out.append(self.local_var(e.name, None, t) if isnew else f"{self.ex(e, True)} = {t};")
            return "\n".join(out)
        if len(s.rhs) != 1:
            self.unsupported(s, "assignment count mismatch")
        t = self.fresh()
        out.append(f"auto {t} = {self.multi_source(s.rhs[0])};")
        for i, (e, isnew) in enumerate(zip(s.lhs, new)):
            if e.name == "_":
                continue
            g = f"std::get<{i}>({t})"
            out.append(self.local_var(e.name, None, f"go::def({g})") if isnew else f"{self.ex(e, True)} = {g};")
        return "\n".join(out)

    def assign(self, s):
        if len(s.lhs) == 1 and len(s.rhs) == 1:
            return self.assign1(s.op, s.lhs[0], self.ex(s.rhs[0])) + ";"
        if s.op != "=":
            self.unsupported(s, "operator assignment with several operands")
        out = ["{"]
        if len(s.lhs) == len(s.rhs):
            tmps = []
            for r in s.rhs:
                t = self.fresh()
                tmps.append(t)
                out.append(f"auto {t} = {self.ex(r)};")
            for e, t in zip(s.lhs, tmps):
                out.append(self.assign1("=", e, t) + ";")
        else:
            if len(s.rhs) != 1:
                self.unsupported(s, "assignment count mismatch")
            t = self.fresh()
            out.append(f"auto {t} = {self.multi_source(s.rhs[0])};")
            for i, e in enumerate(s.lhs):
                out.append(self.assign1("=", e, f"std::get<{i}>({t})") + ";")
        out.append("}")
        return "\n".join(out)

    def ret(self, s):
        fn = self.fn
        if not s.values:
            if fn.named:
                cpps = fn.named_cpp
                if len(cpps) == 1:
                    return f"return {cpps[0]};"
                return f"return {fn.result_type}({', '.join(cpps)});"
            return "return;"
        if len(s.values) == 1:
            return f"return {self.ex(s.values[0])};"
        return f"return {fn.result_type}({self.args(s.values)});"

    def if_(self, s):
        self.push()
        try:
            out = []
            if s.init is not None:
                out.append("{")
                out.append(self.stmt(s.init))
            out.append(f"if ({self.ex(s.cond)}) {self.block(s.then)}")
            if s.els is not None:
                out.append("else " + (self.if_(s.els) if s.els.kind == "if" else self.block(s.els)))
            if s.init is not None:
                out.append("}")
            return "\n".join(out)
        finally:
            self.pop()

    def loop_body(self, body, label, pre=()):
        self.fn.break_stack.append(("loop", label))
        self.push()
        try:
            lines = ["{"] + list(pre) + [self.stmt(x) for x in body.stmts]
            if label is not None:
                lines.append(f"{mangle(label)}_continue: ;")
            lines.append("}")
            return "\n".join(lines)
        finally:
            self.pop()
            self.fn.break_stack.pop()

    def for_(self, s, label):
        self.push()
        try:
            out = ["{"]
            if s.init is not None:
                out.append(self.stmt(s.init))
            cond = self.ex(s.cond) if s.cond is not None else ""
            post = self.simple_expr_stmt(s.post)
            out.append(f"for (; {cond}; {post}) {self.loop_body(s.body, label)}")
            if label is not None:
                out.append(f"{mangle(label)}_break: ;")
            out.append("}")
            return "\n".join(out)
        finally:
            self.pop()

    def forrange(self, s, label):
        self.push()
        try:
            r = self.fresh("_r")
            out = ["{", f"auto {r} = go::range({self.ex(s.x)});"]
            pre = []
            for var, getter in ((s.key, "key"), (s.value, "val")):
                if var is None or (var.kind == "ident" and var.name == "_"):
                    continue
                if s.define:
                    pre.append((var, f"{r}.{getter}()"))
                else:
                    pre.append((var, f"{r}.{getter}()"))
            self.fn.break_stack.append(("loop", label))
            self.push()
            try:
                lines = ["{"]
                for var, init in pre:
                    if s.define:
                        lines.append(self.local_var(var.name, None, init))
                    else:
                        lines.append(f"{self.ex(var, True)} = {init};")
                lines += [self.stmt(x) for x in s.body.stmts]
                if label is not None:
                    lines.append(f"{mangle(label)}_continue: ;")
                lines.append("}")
            finally:
                self.pop()
                self.fn.break_stack.pop()
            out.append(f"while ({r}.next()) " + "\n".join(lines))
            if label is not None:
                out.append(f"{mangle(label)}_break: ;")
            out.append("}")
            return "\n".join(out)
        finally:
            self.pop()

    def switch(self, s):
        self.push()
        try:
            end = self.fresh("_sw_end")
            out = ["{"]
            if s.init is not None:
                out.append(self.stmt(s.init))
            tag = None
            if s.tag is not None:
                tag = self.fresh("_tag")
                out.append(f"auto {tag} = go::def({self.ex(s.tag)});")
            self.fn.break_stack.append(("switch", end))
            try:
                default = None
                first = True
                for c in s.clauses:
                    walk(c.body, lambda n: self.unsupported(n, "fallthrough") if n.kind == "fallthrough" else None)
                    if c.values is None:
                        default = c
                        continue
                    conds = [f"({tag} == {self.ex(v)})" if tag else f"({self.ex(v)})" for v in c.values]
                    self.push()
                    body = "\n".join(self.stmt(x) for x in c.body)
                    self.pop()
                    out.append(("if" if first else "else if") + f" ({' || '.join(conds)}) {{\n{body}\n}}")
                    first = False
                if default is not None:
                    self.push()
                    body = "\n".join(self.stmt(x) for x in default.body)
                    self.pop()
                    out.append(("" if first else "else ") + f"{{\n{body}\n}}")
            finally:
                self.fn.break_stack.pop()
            out.append(f"{end}: ;")
            out.append("}")
            return "\n".join(out)
        finally:
            self.pop()

    def jump(self, s):
        st = self.fn.break_stack
        if s.label is not None:
            if not any(kind == "loop" and lab == s.label for kind, lab in st):
                self.unsupported(s, f"label {s.label} does not name an enclosing loop")
            return f"goto {mangle(s.label)}_{s.kind};"
        if not st:
            self.unsupported(s, f"{s.kind} outside a loop")
        if s.kind == "break":
            kind, lab = st[-1]
            return f"goto {lab};" if kind == "switch" else "break;"
        for kind, lab in reversed(st):
            if kind == "loop":
                return "continue;"
        self.unsupported(s, "continue outside a loop")

    def go(self, s):
        c = s.call
        if c.kind != "call":
            self.unsupported(s, "go with a non-call")
        if c.fun.kind == "funclit" and not c.args:
            return f"go::spawn({self.ex(c.fun)});"
        t = self.fresh("_ga")
        args = ", ".join(f"std::get<{i}>({t})" for i in range(len(c.args)))
        return f"{{ auto {t} = std::make_tuple({self.args(c.args)}); go::spawn([=]{{ {self.ex(c.fun)}({args}); }}); }}"

    def defer(self, s):
        c = s.call
        if c.kind != "call":
            self.unsupported(s, "defer with a non-call")
        if c.fun.kind == "funclit" and not c.args:
            return f"_defer.add({self.ex(c.fun)});"
        t = self.fresh("_da")
        args = ", ".join(f"std::get<{i}>({t})" for i in range(len(c.args)))
        return f"{{ auto {t} = std::make_tuple({self.args(c.args)}); _defer.add([=]{{ {self.ex(c.fun)}({args}); }}); }}"

    # ---------- declarations
    def func_header(self, decl, qualified_owner=None, for_class=False):
        """(template prefix, return type, name, params-with-names, suffix).  Declares params in the current scope."""
        tprefix = ""
        if decl.tparams:
            if decl.recv is not None:
                self.unsupported(decl, "generic methods")
            names = []
            for tp in decl.tparams:
                self.scope.names[tp.name] = ("tparam", mangle(tp.name))
                names.append("class " + mangle(tp.name))
            tprefix = f"template <{', '.join(names)}> "
        rt = self.result_type(decl.sig)
        params = self.param_list(decl.sig)
        suffix = ""
        if decl.recv is not None and decl.recv.type.kind != "tptr":
            suffix = " const"
        return tprefix, rt, params, suffix

    def func_proto(self, pkg, decl):
        self.enter_decl(pkg, decl)
        self.push(func_boundary=True)
        try:
            tprefix, rt, params, suffix = self.func_header(decl)
            return f"{tprefix}{rt} {mangle(decl.name)}({params}){suffix};"
        finally:
            self.pop()

    def func_def(self, pkg, decl, owner=None):
        if decl.body is None:
            self.unsupported(decl, f"function {decl.name} without a body")
        self.enter_decl(pkg, decl)
        self.push(func_boundary=True)
        try:
            tprefix, rt, params, suffix = self.func_header(decl)
            self.fn = FuncCtx(decl.sig, self.named_results(decl.sig), rt)
            self.fn.heap = self.addr_taken(decl.body)
            prologue = []
            if decl.recv is not None and decl.recv.name and decl.recv.name != "_":
                rc = self.declare(decl.recv.name)
                if decl.recv.type.kind == "tptr":
                    prologue.append(f"auto {rc} = this;")
                elif decl.recv.name in self.fn.heap:
                    prologue.append(f"auto& {rc} = *new auto(*this);")
                else:
                    prologue.append(f"auto {rc} = *this;")
            for p in decl.sig.params:
                if p.name in self.fn.heap:
                    self.unsupported(decl, f"address of parameter {p.name} taken")
            name = mangle(decl.name) if owner is None else f"{mangle(owner)}::{mangle(decl.name)}"
            body = self.func_body(decl.body, prologue)
            return f"{tprefix}{rt} {name}({params}){suffix} {body}"
        finally:
            self.pop()
            self.fn = None

    def interface_methods(self, pkg, tdecl, seen=None):
        """Flattened method list [(imethod node, package, typedecl)] of an interface declaration."""
        seen = seen or set()
        key = (pkg.path, tdecl.name)
        if key in seen:
            return []
        seen.add(key)
        out = [(m, pkg, tdecl) for m in tdecl.type.methods]
        for emb in tdecl.type.embeds:
            found = self.in_decl_scope(pkg, tdecl, lambda: self.type_decl_of(emb))
            if found is None:
                if emb.kind == "tname" and emb.pkg is None and emb.name == "error":
                    self.unsupported(emb, "embedding the predeclared error interface")
                self.unsupported(emb, "embedded interface whose declaration is not among the parsed packages")
            q, d = found
            if d.type.kind != "tinterface":
                self.unsupported(emb, "embedded non-interface in an interface")
            out += self.interface_methods(q, d, seen)
        return out

    def type_def(self, pkg, d):
        """C++ definition of type declaration d (class body with method prototypes)."""
        self.enter_decl(pkg, d)
        name = mangle(d.name)
        t = d.type
        if d.alias:
            return f"using {name} = {self.ty(t)};"
        protos = []
        for mname, m in sorted(pkg.methods.get(d.name, {}).items()):
            if ("method", pkg.path, d.name, mname) not in self.reach.keys:
                continue
            self.enter_decl(pkg, m)
            self.push(func_boundary=True)
            try:
                _, rt, params, suffix = self.func_header(m)
                protos.append(f"    {rt} {mangle(mname)}({params}){suffix};")
            finally:
                self.pop()
        self.enter_decl(pkg, d)
        if t.kind == "tstruct":
            bases, fields = [], []
            for f in t.fields:
                if f.embedded and f.type.kind == "tname":
                    bases.append(self.ty(f.type))
                    # the embedded field selected by NAME (x.T): a cast to the base, found through this typedef (go_sel_T below)
                    fields.append(f"    using go_emb_{mangle(f.name)} = {bases[-1]};")
                elif f.name == "_":
                    continue
                else:
                    fields.append(f"    {self.ty(f.type)} {mangle(f.name)}{{}};")
            head = f"struct {name}" + (" : " + ", ".join(bases) if bases else "")
            # field visitor (declaration order, embedded structs first): what reflection-driven library calls walk
            visits = [f"f(static_cast<{b}&>(*this));" for b in bases]
            visits += [f"f({mangle(f.name)});" for f in t.fields if not (f.embedded and f.type.kind == "tname") and f.name != "_"]
            fields.append("    template <class F_> void go_fields_(F_&& f) { " + " ".join(visits) + " }")
            return head + " {\n" + "\n".join(fields + protos) + "\n};"
        if t.kind == "tinterface":
            ms = self.interface_methods(pkg, d)
            virt, impl, fwd = [], [], []
            for m, q, owner in ms:
                def one():
                    self.push(func_boundary=True)
                    try:
                        rt = self.result_type(m.sig)
                        ps = [(self.param_type(p), f"a{i}") for i, p in enumerate(m.sig.params)]
                        return rt, ps
                    finally:
                        self.pop()
                rt, ps = self.in_decl_scope(q, owner, one)
                plist = ", ".join(f"{pt} {pn}" for pt, pn in ps)
                alist = ", ".join(pn for _, pn in ps)
                mn = mangle(m.name)
                virt.append(f"        virtual {rt} {mn}({plist}) = 0;")
                impl.append(f"        {rt} {mn}({plist}) override {{ return go::deref(v).{mn}({alist}); }}")
                fwd.append(f"    {rt} {mn}({plist}) const;")
            return (
                f"struct {name} {{\n"
                f"    struct I {{\n{chr(10).join(virt)}\n        virtual ~I() {{}}\n    }};\n"
                f"    template <class T> struct M : I {{\n        T v;\n        explicit M(T x) : v(std::move(x)) {{}}\n{chr(10).join(impl)}\n    }};\n"
                f"    std::shared_ptr<I> p;\n"
                f"    {name}() = default;\n"
                f"    {name}(go::Nil) {{}}\n"
                f"    template <class T, class = std::enable_if_t<!std::is_same_v<std::decay_t<T>, {name}> && !std::is_same_v<std::decay_t<T>, go::Nil>>>\n"
                f"    {name}(T x) : p(std::make_shared<M<T>>(std::move(x))) {{}}\n"
                f"    I* get() const {{ if (!p) go::panic_msg(\"runtime error: invalid memory address or nil pointer dereference (nil interface)\"); return p.get(); }}\n"
                f"    friend bool operator==(const {name}& a, go::Nil) {{ return !a.p; }}\n"
                f"    friend bool operator!=(const {name}& a, go::Nil) {{ return !!a.p; }}\n"
                + "\n".join(fwd) + "\n};"
            )
        base = self.ty(t)
        return (f"struct {name} : {base} {{\n    using Base_ = {base};\n    using Base_::Base_;\n    {name}() = default;\n"
                f"    {name}(const Base_& b) : Base_(b) {{}}\n" + "\n".join(protos) + "\n};")

    def interface_forwarders(self, pkg, d):
        out = []
        name = mangle(d.name)
        for m, q, owner in self.interface_methods(pkg, d):
            def one():
                self.push(func_boundary=True)
                try:
                    rt = self.result_type(m.sig)
                    ps = [(self.param_type(p), f"a{i}") for i, p in enumerate(m.sig.params)]
                    return rt, ps
                finally:
                    self.pop()
            rt, ps = self.in_decl_scope(q, owner, one)
            plist = ", ".join(f"{pt} {pn}" for pt, pn in ps)
            alist = ", ".join(pn for _, pn in ps)
            out.append(f"inline {rt} {name}::{mangle(m.name)}({plist}) const {{ return get()->{mangle(m.name)}({alist}); }}")
        return "\n".join(out)

    def value_deps(self, pkg, d):
        """Types that must be COMPLETE before d's definition: by-value fields, bases, underlying types."""
        deps = []

        def by_value(t):
            if t.kind == "tname":
                found = self.in_decl_scope(pkg, d, lambda: self.type_decl_of(t))
                if found is not None:
                    deps.append(("type", found[0].path, found[1].name))
            elif t.kind == "tarray":
                by_value(t.elem)
            elif t.kind == "tstruct":
                for f in t.fields:
                    by_value(f.type)
        t = d.type
        if t.kind == "tstruct":
            for f in t.fields:
                by_value(f.type)
        elif t.kind != "tinterface":
            by_value(t)
        return deps

    def global_decl(self, pkg, d, names):
        """Package-level var / const specs -> inline variables."""
        self.enter_decl(pkg, d)
        out = []
        if d.kind == "constdecl":
            self.iota = d.iota
            try:
                for nm, v in zip(d.names, d.values):
                    if nm not in names or nm == "_":
                        continue
                    init = self.ex(v)
                    if d.type is not None:
                        out.append(f"inline const {self.ty(d.type)} {mangle(nm)} = {init};")
                    else:
                        out.append(f"inline const auto {mangle(nm)} = {init};")
            finally:
                self.iota = None
        else:
            self.fn = FuncCtx(None, None, "void")
            try:
                if d.values is None:
                    for nm in d.names:
                        if nm in names:
                            out.append(f"inline {self.ty(d.type)} {mangle(nm)}{{}};")
                elif len(d.values) == len(d.names):
                    for nm, v in zip(d.names, d.values):
                        if nm in names:
                            init = self.ex(v)
                            out.append(f"inline {self.ty(d.type)} {mangle(nm)} = {init};" if d.type is not None
                                       else f"inline auto {mangle(nm)} = go::def({init});")
                else:
                    self.unsupported(d, "multi-valued package-level variable initialisation")
            finally:
                self.fn = None
        return "\n".join(out)

    # ---------- whole program
    def find_embedded_names(self):
        names = set()
        for k in self.reach.keys:
            if k[0] == "type":
                t = self.prog.packages[k[1]].types[k[2]].type
                if t.kind == "tstruct":
                    names |= {f.name for f in t.fields if f.embedded and f.type.kind == "tname"}
        return names

    def emit(self, banner):
        keys = self.reach.keys
        self.embedded_names = self.find_embedded_names()
        pkgs = self.package_order()
        out = [banner, "#pragma once", '#include "gort.hpp"', ""]
        # x.T where T names an embedded VALUE field of some struct: embedded structs / interfaces are C++ bases (that is what
        # promotes their fields and methods), so the selection is a cast to the base; where T is an ordinary member it is that
        for nm in sorted(self.embedded_names):
            m = mangle(nm)
            out.append(f"template <class T_> decltype(auto) go_sel_{m}(T_&& o) {{\n"
                       f"    if constexpr (requires {{ o.{m}; }}) return (o.{m});\n"
                       f"    else return static_cast<std::conditional_t<std::is_const_v<std::remove_reference_t<T_>>, "
                       f"const typename std::remove_cvref_t<T_>::go_emb_{m}, typename std::remove_cvref_t<T_>::go_emb_{m}>&>(o);\n}}")
        # A. forward declarations
        for pkg in pkgs:
            names = sorted(k[2] for k in keys if k[0] == "type" and k[1] == pkg.path and not pkg.types[k[2]].alias)
            if names:
                out.append(f"namespace {pkg.ns} {{ " + " ".join(f"struct {mangle(n)};" for n in names) + " }")
        # B. type definitions, complete-before-use order
        tkeys = sorted(k for k in keys if k[0] == "type")
        done, order = set(), []

        def visit(k, stack=()):
            if k in done:
                return
            if k in stack:
                raise Unsupported(f"recursive by-value type {k}")
            pkg = self.prog.packages[k[1]]
            for dep in self.value_deps(pkg, pkg.types[k[2]]):
                if dep in keys:
                    visit(dep, stack + (k,))
            done.add(k)
            order.append(k)
        for k in tkeys:
            visit(k)
        for k in order:
            pkg = self.prog.packages[k[1]]
            out.append(f"namespace {pkg.ns} {{\n{self.type_def(pkg, pkg.types[k[2]])}\n}}")
        # C. constants, prototypes
        for pkg in pkgs:
            body = []
            seen = set()
            for k in sorted(k for k in keys if k[0] == "const" and k[1] == pkg.path):
                d = pkg.consts[k[2]]
                if id(d) in seen:
                    continue
                seen.add(id(d))
                body.append((d.line, self.global_decl(pkg, d, {kk[2] for kk in keys if kk[0] == "const" and kk[1] == pkg.path})))
            body = [b for _, b in sorted(body)]
            for k in sorted(k for k in keys if k[0] == "func" and k[1] == pkg.path):
                body.append(self.func_proto(pkg, pkg.funcs[k[2]]))
            if body:
                out.append(f"namespace {pkg.ns} {{\n" + "\n".join(body) + "\n}")
        # D. package-level variables
        for pkg in pkgs:
            body, seen = [], set()
            for k in sorted(k for k in keys if k[0] == "var" and k[1] == pkg.path):
                d = pkg.vars[k[2]]
                if id(d) in seen:
                    continue
                seen.add(id(d))
                body.append((d.line, self.global_decl(pkg, d, {kk[2] for kk in keys if kk[0] == "var" and kk[1] == pkg.path})))
            if body:
                out.append(f"namespace {pkg.ns} {{\n" + "\n".join(b for _, b in sorted(body)) + "\n}")
        # E. definitions
        for pkg in pkgs:
            body = []
            for k in sorted(k for k in keys if k[0] == "type" and k[1] == pkg.path):
                d = pkg.types[k[2]]
                if d.type.kind == "tinterface" and not d.alias:
                    body.append(self.interface_forwarders(pkg, d))
            if body:
                out.append(f"namespace {pkg.ns} {{\n" + "\n".join(body) + "\n}")
        for pkg in pkgs:
            body = []
            for k in sorted(k for k in keys if k[1] == pkg.path and k[0] in ("func", "method")):
                if k[0] == "func":
                    d = pkg.funcs[k[2]]
                    body.append((d.line, ("" if d.tparams else "inline ") + self.func_def(pkg, d)))
                else:
                    d = pkg.methods[k[2]][k[3]]
                    body.append((d.line, "inline " + self.func_def(pkg, d, owner=k[2])))
            if body:
                out.append(f"namespace {pkg.ns} {{\n" + "\n\n".join(b for _, b in sorted(body, key=lambda x: x[0])) + "\n}")
        return "\n".join(out) + "\n"

    def package_order(self):
        used = sorted({k[1] for k in self.reach.keys})
        order, done = [], set()

        def visit(path, stack=()):
            if path in done or path in stack:
                return
            pkg = self.prog.packages[path]
            for f in pkg.files:
                for imp in f.imports:
                    if imp.path in self.prog.packages:
                        visit(imp.path, stack + (path,))
            done.add(path)
            order.append(pkg)
        for p in used:
            visit(p)
        return [p for p in order if p.path in used]


def translate(module_root, roots, banner="// generated by go2cxx"):
    prog = Program(module_root)
    reach = Reach(prog)
    for r in roots:
        parts = r.split(".")
        pkg = prog.by_dir(parts[0])
        if len(parts) == 2:
            if parts[1] not in pkg.types and parts[1] not in pkg.funcs and parts[1] not in pkg.vars and parts[1] not in pkg.consts:
                raise SystemExit(f"go2cxx: {r}: no such declaration")
            reach.add_name(pkg, parts[1])
        elif len(parts) == 3:
            if parts[2] not in pkg.methods.get(parts[1], {}):
                raise SystemExit(f"go2cxx: {r}: no such method")
            reach.add_name(pkg, parts[1])
            reach.method_names.add(parts[2])
        else:
            raise SystemExit(f"go2cxx: bad root {r!r} (want pkgdir.Name or pkgdir.Type.Method)")
    reach.run()
    return Emitter(prog, reach).emit(banner), reach


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--module-root", required=True, help="directory holding go.mod")
    ap.add_argument("--root", action="append", default=[], help="pkgdir.Name or pkgdir.Type.Method (repeatable)")
    ap.add_argument("-o", "--output", required=True)
    ap.add_argument("--list", action="store_true", help="print the reachable declarations")
    a = ap.parse_args(argv)
    try:
        text, reach = translate(a.module_root, a.root,
                                banner=f"// GENERATED by oracle/go2cxx/go2cxx.py from the Go sources under {a.module_root}\n"
                                       f"// roots: {' '.join(a.root)}\n// Do not edit, do not commit (oracle/_ref/ is git-ignored).")
    except (Unsupported, GoSyntaxError) as e:
        print(f"go2cxx: {e}", file=sys.stderr)
        return 1
    with open(a.output, "w") as f:
        f.write(text)
    if a.list:
        for k in sorted(reach.keys):
            print(".".join(k[1:]), f"({k[0]})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
