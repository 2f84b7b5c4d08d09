"""Lexer and parser for the Go language (the subset a Go 1.21 source tree without generics-heavy code uses).

TEST INFRASTRUCTURE.  Part of oracle/go2cxx: a generic, syntax-directed Go -> C++ translator used to derive
oracle/_ref/ mechanically from the reference's own source files (read where they lie under /root/reference, never
copied).  This module knows the Go grammar (The Go Programming Language Specification: lexical elements incl. automatic
semicolon insertion, types, expressions with the five binary precedence levels, statements, declarations) and nothing
about any particular program: no identifier of the reference appears here.

The AST is made of `Node` objects: `Node(kind, **fields)`; every node carries the source line for diagnostics.
A top-level declaration that cannot be parsed is kept as a `bad` node (name + message): translation fails only if such a
declaration is reachable from what is being translated.
"""
from __future__ import annotations

KEYWORDS = {
    "break", "case", "chan", "const", "continue", "default", "defer", "else", "fallthrough", "for", "func", "go", "goto",
    "if", "import", "interface", "map", "package", "range", "return", "select", "struct", "switch", "type", "var",
}

OPERATORS = sorted([
    "<<=", ">>=", "&^=", "...", "&&", "||", "<-", "++", "--", "==", "!=", "<=", ">=", ":=", "&^", "<<", ">>", "+=", "-=",
    "*=", "/=", "%=", "&=", "|=", "^=", "(", ")", "[", "]", "{", "}", ",", ";", ".", ":", "+", "-", "*", "/", "%", "&", "|",
    "^", "<", ">", "=", "!", "~",
], key=len, reverse=True)

BINARY_PREC = {
    "||": 1, "&&": 2,
    "==": 3, "!=": 3, "<": 3, "<=": 3, ">": 3, ">=": 3,
    "+": 4, "-": 4, "|": 4, "^": 4,
    "*": 5, "/": 5, "%": 5, "<<": 5, ">>": 5, "&": 5, "&^": 5,
}

ASSIGN_OPS = {"=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "<<=", ">>=", "&^="}


class GoSyntaxError(Exception):
    pass


class Node:
    def __init__(self, kind, line=0, **kw):
        self.kind = kind
        self.line = line
        self.__dict__.update(kw)

    def __repr__(self):
        f = ", ".join(f"{k}={v!r}" for k, v in self.__dict__.items() if k not in ("kind", "line"))
        return f"{self.kind}({f})"


class Tok:
    __slots__ = ("kind", "val", "line")

    def __init__(self, kind, val, line):
        self.kind, self.val, self.line = kind, val, line

    def __repr__(self):
        return f"{self.kind}:{self.val!r}@{self.line}"


_ESC = {"a": 7, "b": 8, "f": 12, "n": 10, "r": 13, "t": 9, "v": 11, "\\": 92, "'": 39, '"': 34}


def _unescape(s, i, quote):
    """Decode one (possibly escaped) character of an interpreted string / rune literal starting at s[i].
    Returns (bytes, next index); \\x and octal escapes give one byte, \\u / \\U and plain characters UTF-8."""
    c = s[i]
    if c != "\\":
        return c.encode("utf-8"), i + 1
    c = s[i + 1]
    if c in _ESC:
        return bytes([_ESC[c]]), i + 2
    if c == "x":
        return bytes([int(s[i + 2:i + 4], 16)]), i + 4
    if c == "u":
        return chr(int(s[i + 2:i + 6], 16)).encode("utf-8"), i + 6
    if c == "U":
        return chr(int(s[i + 2:i + 10], 16)).encode("utf-8"), i + 10
    if c in "01234567":
        return bytes([int(s[i + 1:i + 4], 8)]), i + 4
    raise GoSyntaxError(f"bad escape \\{c}")


def lex(src: str):
    toks = []
    i, n, line = 0, len(src), 1

    def need_semi():
        if not toks:
            return False
        t = toks[-1]
        if t.kind in ("ident", "int", "float", "imag", "rune", "string"):
            return True
        if t.kind == "kw" and t.val in ("break", "continue", "fallthrough", "return"):
            return True
        return t.kind == "op" and t.val in ("++", "--", ")", "]", "}")

    while i < n:
        c = src[i]
        if c == "\n":
            if need_semi():
                toks.append(Tok("op", ";", line))
            line += 1
            i += 1
        elif c in " \t\r":
            i += 1
        elif src.startswith("//", i):
            while i < n and src[i] != "\n":
                i += 1
        elif src.startswith("/*", i):
            j = src.index("*/", i + 2)
            if "\n" in src[i:j] and need_semi():
                toks.append(Tok("op", ";", line))
            line += src.count("\n", i, j)
            i = j + 2
        elif c.isalpha() or c == "_":
            j = i + 1
            while j < n and (src[j].isalnum() or src[j] == "_"):
                j += 1
            w = src[i:j]
            toks.append(Tok("kw" if w in KEYWORDS else "ident", w, line))
            i = j
        elif c.isdigit() or (c == "." and i + 1 < n and src[i + 1].isdigit()):
            j = i
            is_float = False
            if src.startswith(("0x", "0X"), i):
                j = i + 2
                while j < n and (src[j] in "0123456789abcdefABCDEF_"):
                    j += 1
                if j < n and src[j] in ".pP":
                    raise GoSyntaxError(f"line {line}: hexadecimal floating-point literals are not supported")
                val = int(src[i + 2:j].replace("_", ""), 16)
            elif src.startswith(("0b", "0B"), i):
                j = i + 2
                while j < n and src[j] in "01_":
                    j += 1
                val = int(src[i + 2:j].replace("_", ""), 2)
            elif src.startswith(("0o", "0O"), i):
                j = i + 2
                while j < n and src[j] in "01234567_":
                    j += 1
                val = int(src[i + 2:j].replace("_", ""), 8)
            else:
                while j < n and (src[j].isdigit() or src[j] == "_"):
                    j += 1
                if j < n and src[j] == ".":
                    is_float = True
                    j += 1
                    while j < n and (src[j].isdigit() or src[j] == "_"):
                        j += 1
                if j < n and src[j] in "eE":
                    is_float = True
                    j += 1
                    if j < n and src[j] in "+-":
                        j += 1
                    while j < n and src[j].isdigit():
                        j += 1
                text = src[i:j].replace("_", "")
                if is_float:
                    val = text
                elif len(text) > 1 and text[0] == "0":
                    val = int(text, 8)
                else:
                    val = int(text)
            if j < n and src[j] == "i":
                raise GoSyntaxError(f"line {line}: imaginary literals are not supported")
            toks.append(Tok("float" if is_float else "int", val, line))
            i = j
        elif c == "'":
            b, j = _unescape(src, i + 1, "'")
            if src[j] != "'":
                raise GoSyntaxError(f"line {line}: bad rune literal")
            if src[i + 1] == "\\" and src[i + 2] in "x01234567":
                val = b[0]
            else:
                val = ord(b.decode("utf-8"))
            toks.append(Tok("rune", val, line))
            i = j + 1
        elif c == '"':
            out = bytearray()
            j = i + 1
            while src[j] != '"':
                if src[j] == "\n":
                    raise GoSyntaxError(f"line {line}: newline in string")
                b, j = _unescape(src, j, '"')
                out += b
            toks.append(Tok("string", bytes(out), line))
            i = j + 1
        elif c == "`":
            j = src.index("`", i + 1)
            raw = src[i + 1:j].replace("\r", "")
            toks.append(Tok("string", raw.encode("utf-8"), line))
            line += raw.count("\n")
            i = j + 1
        else:
            for op in OPERATORS:
                if src.startswith(op, i):
                    toks.append(Tok("op", op, line))
                    i += len(op)
                    break
            else:
                raise GoSyntaxError(f"line {line}: unexpected character {c!r}")
    if need_semi():
        toks.append(Tok("op", ";", line))
    toks.append(Tok("eof", None, line))
    return toks


class Parser:
    def __init__(self, src: str, filename: str = "<go>"):
        self.toks = lex(src)
        self.p = 0
        self.filename = filename
        self.expr_lev = 0  # < 0: inside a control clause (composite literals of bare type names are not allowed)

    # --- token helpers
    @property
    def tok(self):
        return self.toks[self.p]

    def peek(self, k=1):
        return self.toks[min(self.p + k, len(self.toks) - 1)]

    def err(self, msg):
        raise GoSyntaxError(f"{self.filename}:{self.tok.line}: {msg} (at {self.tok!r})")

    def is_op(self, *vals):
        return self.tok.kind == "op" and self.tok.val in vals

    def is_kw(self, *vals):
        return self.tok.kind == "kw" and self.tok.val in vals

    def accept_op(self, val):
        if self.is_op(val):
            self.p += 1
            return True
        return False

    def expect_op(self, val):
        if not self.is_op(val):
            self.err(f"expected {val!r}")
        self.p += 1

    def expect_kw(self, val):
        if not self.is_kw(val):
            self.err(f"expected {val!r}")
        self.p += 1

    def ident(self):
        if self.tok.kind != "ident":
            self.err("expected identifier")
        v = self.tok.val
        self.p += 1
        return v

    def skip_semi(self):
        if self.is_op(";"):
            self.p += 1
        elif not self.is_op(")", "}") and self.tok.kind != "eof":
            self.err("expected ';'")

    # --- file
    def parse_file(self):
        self.expect_kw("package")
        pkg = self.ident()
        self.skip_semi()
        imports, decls = [], []
        while self.is_kw("import"):
            self.p += 1
            if self.accept_op("("):
                while not self.is_op(")"):
                    imports.append(self.import_spec())
                    self.skip_semi()
                self.expect_op(")")
            else:
                imports.append(self.import_spec())
            self.skip_semi()
        while self.tok.kind != "eof":
            start = self.p
            try:
                decls.extend(self.top_decl())
                self.skip_semi()
            except GoSyntaxError as e:
                decls.append(self.recover_top(start, str(e)))
        return Node("file", 1, package=pkg, imports=imports, decls=decls, filename=self.filename)

    def import_spec(self):
        line = self.tok.line
        alias = None
        if self.tok.kind == "ident":
            alias = self.ident()
        elif self.is_op("."):
            self.err("dot imports are not supported")
        if self.tok.kind != "string":
            self.err("expected import path")
        path = self.tok.val.decode()
        self.p += 1
        return Node("import", line, alias=alias, path=path)

    def recover_top(self, start, msg):
        """Skip a declaration that did not parse: find its name, then its end (the ';' at brace depth 0)."""
        self.p = start
        line = self.tok.line
        name = None
        q = start + 1
        if self.toks[start].kind == "kw" and self.toks[start].val == "func":
            if self.toks[q].kind == "op" and self.toks[q].val == "(":  # receiver
                depth = 0
                while True:
                    t = self.toks[q]
                    if t.kind == "op" and t.val == "(":
                        depth += 1
                    if t.kind == "op" and t.val == ")":
                        depth -= 1
                        if depth == 0:
                            break
                    q += 1
                q += 1
        if self.toks[q].kind == "ident":
            name = self.toks[q].val
        depth = 0
        while self.tok.kind != "eof":
            t = self.tok
            if t.kind == "op" and t.val in "([{":
                depth += 1
            elif t.kind == "op" and t.val in ")]}":
                depth -= 1
            elif t.kind == "op" and t.val == ";" and depth == 0:
                self.p += 1
                break
            self.p += 1
        return Node("bad", line, name=name, msg=msg, first=self.toks[start].val)

    def top_decl(self):
        if self.is_kw("func"):
            return [self.func_decl()]
        if self.is_kw("const", "var", "type"):
            return self.gen_decl()
        self.err("expected declaration")

    # --- declarations
    def gen_decl(self):
        kw = self.tok.val
        self.p += 1
        out = []
        if self.accept_op("("):
            idx = 0
            prev = None
            while not self.is_op(")"):
                spec = self.spec(kw, idx, prev)
                out.append(spec)
                prev = spec
                idx += 1
                self.skip_semi()
            self.expect_op(")")
        else:
            out.append(self.spec(kw, 0, None))
        return out

    def spec(self, kw, idx, prev):
        line = self.tok.line
        if kw == "type":
            name = self.ident()
            tparams = None
            if self.is_op("[") and self.peek().kind == "ident" and not (self.peek(2).kind == "op" and self.peek(2).val == "]"):
                self.err("generic type declarations are not supported")
            alias = self.accept_op("=")
            t = self.type_()
            return Node("typedecl", line, name=name, type=t, alias=alias, tparams=tparams)
        names = [self.ident()]
        while self.accept_op(","):
            names.append(self.ident())
        t = None
        if not self.is_op("=", ";", ")"):
            t = self.type_()
        values = None
        if self.accept_op("="):
            values = self.expr_list()
        if kw == "const":
            implicit = False
            if values is None:
                if prev is None:
                    self.err("constant declaration without a value")
                values, t, implicit = prev.values, prev.type, True
            return Node("constdecl", line, names=names, type=t, values=values, iota=idx, implicit=implicit)
        return Node("vardecl", line, names=names, type=t, values=values)

    def func_decl(self):
        line = self.tok.line
        self.expect_kw("func")
        recv = None
        if self.is_op("("):
            ps = self.params()
            if len(ps) != 1:
                self.err("method receiver must be one parameter")
            recv = ps[0]
        name = self.ident()
        tparams = None
        if self.is_op("["):
            tparams = self.type_params()
        sig = self.signature()
        body = None
        if self.is_op("{"):
            saved, self.expr_lev = self.expr_lev, 0
            body = self.block()
            self.expr_lev = saved
        return Node("funcdecl", line, name=name, recv=recv, sig=sig, body=body, tparams=tparams)

    def type_params(self):
        self.expect_op("[")
        out = []
        while not self.is_op("]"):
            names = [self.ident()]
            while self.accept_op(","):
                names.append(self.ident())
            cons = [self.constraint_term()]
            while self.accept_op("|"):
                cons.append(self.constraint_term())
            for nm in names:
                out.append(Node("tparam", self.tok.line, name=nm, constraint=cons))
            if not self.accept_op(","):
                break
        self.expect_op("]")
        return out

    def constraint_term(self):
        self.accept_op("~")
        return self.type_()

    def signature(self):
        line = self.tok.line
        params = self.params()
        results = []
        if self.is_op("("):
            results = self.params()
        elif not self.is_op("{", ";", ")", ",", "]", "}", "=", ":=", ":") and self.tok.kind not in ("eof", "string"):
            results = [Node("param", line, name=None, type=self.type_(), variadic=False)]
        return Node("sig", line, params=params, results=results)

    def params(self):
        """Parameter list: either all named (`a, b int, c string`) or all unnamed (`int, string`)."""
        self.expect_op("(")
        entries = []  # (expr-or-type node, type or None, variadic)
        while not self.is_op(")"):
            line = self.tok.line
            variadic = False
            if self.accept_op("..."):
                entries.append((None, self.type_(), True, line))
            else:
                first = self.type_()
                if self.is_op(",", ")"):
                    entries.append((first, None, False, line))
                else:
                    if self.accept_op("..."):
                        variadic = True
                    entries.append((first, self.type_(), variadic, line))
            if not self.accept_op(","):
                break
        self.expect_op(")")
        named = any(t is not None and f is not None for f, t, _, _ in entries)
        out = []
        if named:
            pending = []
            for f, t, variadic, line in entries:
                if f is None or f.kind != "tname" or f.pkg is not None:
                    self.err("mixed named and unnamed parameters")
                pending.append((f.name, line))
                if t is not None:
                    for nm, ln in pending:
                        out.append(Node("param", ln, name=nm, type=t, variadic=variadic))
                    pending = []
            if pending:
                self.err("mixed named and unnamed parameters")
        else:
            for f, t, variadic, line in entries:
                out.append(Node("param", line, name=None, type=t if f is None else f, variadic=variadic))
        return out

    # --- types
    def type_(self):
        line = self.tok.line
        t = self.tok
        if t.kind == "ident":
            name = self.ident()
            if self.is_op(".") and self.peek().kind == "ident":
                self.p += 1
                return Node("tname", line, pkg=name, name=self.ident())
            return Node("tname", line, pkg=None, name=name)
        if self.accept_op("*"):
            return Node("tptr", line, elem=self.type_())
        if self.accept_op("("):
            inner = self.type_()
            self.expect_op(")")
            return inner
        if self.accept_op("["):
            if self.accept_op("]"):
                return Node("tslice", line, elem=self.type_())
            if self.is_op("...") and self.peek().kind == "op" and self.peek().val == "]":
                self.p += 2
                return Node("tarray", line, len=None, elem=self.type_())
            saved, self.expr_lev = self.expr_lev, 0
            ln = self.expr()
            self.expr_lev = saved
            self.expect_op("]")
            return Node("tarray", line, len=ln, elem=self.type_())
        if self.is_kw("map"):
            self.p += 1
            self.expect_op("[")
            k = self.type_()
            self.expect_op("]")
            return Node("tmap", line, key=k, elem=self.type_())
        if self.is_kw("chan"):
            self.p += 1
            d = "both"
            if self.accept_op("<-"):
                d = "send"
            return Node("tchan", line, dir=d, elem=self.type_())
        if self.is_op("<-") and self.peek().kind == "kw" and self.peek().val == "chan":
            self.p += 2
            return Node("tchan", line, dir="recv", elem=self.type_())
        if self.is_kw("func"):
            self.p += 1
            return Node("tfunc", line, sig=self.signature())
        if self.is_kw("struct"):
            return self.struct_type()
        if self.is_kw("interface"):
            return self.interface_type()
        self.err("expected type")

    def struct_type(self):
        line = self.tok.line
        self.expect_kw("struct")
        self.expect_op("{")
        fields = []
        while not self.is_op("}"):
            fl = self.tok.line
            if self.is_op("*") or (self.tok.kind == "ident" and (
                    (self.peek().kind == "op" and self.peek().val in (";", "}", ".")) or self.peek().kind == "string")):
                # embedded field: T, *T, pkg.T, *pkg.T
                ptr = self.accept_op("*")
                t = self.type_()
                if t.kind != "tname":
                    self.err("bad embedded field")
                fields.append(Node("field", fl, name=t.name, type=Node("tptr", fl, elem=t) if ptr else t, embedded=True))
            else:
                names = [self.ident()]
                while self.accept_op(","):
                    names.append(self.ident())
                t = self.type_()
                for nm in names:
                    fields.append(Node("field", fl, name=nm, type=t, embedded=False))
            if self.tok.kind == "string":  # field tag
                self.p += 1
            self.skip_semi()
        self.expect_op("}")
        return Node("tstruct", line, fields=fields)

    def interface_type(self):
        line = self.tok.line
        self.expect_kw("interface")
        self.expect_op("{")
        methods, embeds = [], []
        while not self.is_op("}"):
            ml = self.tok.line
            if self.tok.kind == "ident" and self.peek().kind == "op" and self.peek().val == "(":
                name = self.ident()
                methods.append(Node("imethod", ml, name=name, sig=self.signature()))
            else:
                self.accept_op("~")
                t = self.type_()
                if self.is_op("|"):
                    self.err("union constraints in interfaces are not supported")
                embeds.append(t)
            self.skip_semi()
        self.expect_op("}")
        return Node("tinterface", line, methods=methods, embeds=embeds)

    # --- statements
    def block(self):
        line = self.tok.line
        self.expect_op("{")
        stmts = self.stmt_list()
        self.expect_op("}")
        return Node("block", line, stmts=stmts)

    def stmt_list(self):
        out = []
        while not self.is_op("}") and not self.is_kw("case", "default") and self.tok.kind != "eof":
            s = self.stmt()
            if s is not None:
                out.append(s)
            if self.is_op("}") or self.is_kw("case", "default"):
                break
            self.skip_semi()
        return out

    def stmt(self):
        line = self.tok.line
        t = self.tok
        if t.kind == "op" and t.val == ";":
            return None
        if t.kind == "op" and t.val == "{":
            return self.block()
        if t.kind == "kw":
            v = t.val
            if v in ("var", "const", "type"):
                return Node("declstmt", line, decls=self.gen_decl())
            if v == "return":
                self.p += 1
                vals = []
                if not self.is_op(";", "}"):
                    vals = self.expr_list()
                return Node("return", line, values=vals)
            if v in ("break", "continue", "goto"):
                self.p += 1
                label = None
                if self.tok.kind == "ident":
                    label = self.ident()
                return Node(v, line, label=label)
            if v == "fallthrough":
                self.p += 1
                return Node("fallthrough", line)
            if v == "go" or v == "defer":
                self.p += 1
                return Node(v, line, call=self.expr())
            if v == "if":
                return self.if_stmt()
            if v == "for":
                return self.for_stmt()
            if v == "switch":
                return self.switch_stmt()
            if v == "select":
                return self.select_stmt()
            if v == "func":
                pass  # function literal as an expression statement
            elif v not in ("map", "chan", "struct", "interface"):
                self.err("unexpected keyword")
        if t.kind == "ident" and self.peek().kind == "op" and self.peek().val == ":" and self.expr_lev >= 0:
            label = self.ident()
            self.p += 1
            if self.is_op("}"):
                return Node("labeled", line, label=label, stmt=None)
            return Node("labeled", line, label=label, stmt=self.stmt())
        return self.simple_stmt()

    def simple_stmt(self, allow_range=False):
        line = self.tok.line
        if allow_range and self.is_kw("range"):
            self.p += 1
            return Node("range", line, key=None, value=None, define=False, x=self.expr())
        lhs = self.expr_list()
        t = self.tok
        if t.kind == "op":
            if t.val == ":=" or t.val in ASSIGN_OPS:
                self.p += 1
                if allow_range and self.is_kw("range") and t.val in (":=", "="):
                    self.p += 1
                    x = self.expr()
                    if len(lhs) > 2:
                        self.err("range with more than two variables")
                    return Node("range", line, key=lhs[0], value=lhs[1] if len(lhs) > 1 else None,
                                define=(t.val == ":="), x=x)
                rhs = self.expr_list()
                if t.val == ":=":
                    for e in lhs:
                        if e.kind != "ident":
                            self.err("non-name on the left side of :=")
                    return Node("define", line, lhs=lhs, rhs=rhs)
                return Node("assign", line, op=t.val, lhs=lhs, rhs=rhs)
            if t.val in ("++", "--"):
                self.p += 1
                return Node("incdec", line, x=lhs[0], op=t.val)
            if t.val == "<-":
                self.p += 1
                return Node("send", line, ch=lhs[0], value=self.expr())
        if len(lhs) != 1:
            self.err("expected one expression")
        return Node("exprstmt", line, x=lhs[0])

    def header(self, allow_range=False):
        """init; cond of if / switch / for headers.  Returns (init, cond-or-range-or-None, post, saw_semicolons)."""
        saved, self.expr_lev = self.expr_lev, -1
        init = cond = post = None
        semis = False
        if not self.is_op("{"):
            if not self.is_op(";"):
                init = self.simple_stmt(allow_range)
            if self.is_op(";") and not (init is not None and init.kind == "range"):
                semis = True
                self.p += 1
                if allow_range:  # for init; cond; post
                    if not self.is_op(";"):
                        cond = self.simple_stmt()
                    self.expect_op(";")
                    if not self.is_op("{"):
                        post = self.simple_stmt()
                else:
                    if not self.is_op("{"):
                        cond = self.simple_stmt()
            else:
                cond, init = init, None
        self.expr_lev = saved
        return init, cond, post, semis

    def if_stmt(self):
        line = self.tok.line
        self.expect_kw("if")
        init, cond, _, _ = self.header()
        if cond is None or cond.kind != "exprstmt":
            self.err("missing condition in if statement")
        then = self.block()
        els = None
        if self.is_kw("else"):
            self.p += 1
            els = self.if_stmt() if self.is_kw("if") else self.block()
        return Node("if", line, init=init, cond=cond.x, then=then, els=els)

    def for_stmt(self):
        line = self.tok.line
        self.expect_kw("for")
        init, cond, post, semis = self.header(allow_range=True)
        body = self.block()
        if cond is not None and cond.kind == "range":
            return Node("forrange", line, key=cond.key, value=cond.value, define=cond.define, x=cond.x, body=body)
        if cond is not None and cond.kind != "exprstmt":
            self.err("bad for condition")
        return Node("for", line, init=init, cond=cond.x if cond is not None else None, post=post, body=body)

    def switch_stmt(self):
        line = self.tok.line
        self.expect_kw("switch")
        init, tag, _, _ = self.header()
        typeswitch = False
        if tag is not None:
            x = tag.rhs[0] if tag.kind == "define" and len(tag.rhs) == 1 else (tag.x if tag.kind == "exprstmt" else None)
            if x is not None and x.kind == "typeassert" and x.type is None:
                typeswitch = True
            elif tag.kind != "exprstmt":
                self.err("bad switch tag")
        self.expect_op("{")
        clauses = []
        while not self.is_op("}"):
            cl = self.tok.line
            if self.is_kw("default"):
                self.p += 1
                vals = None
            else:
                self.expect_kw("case")
                vals = self.type_list() if typeswitch else self.expr_list()
            self.expect_op(":")
            clauses.append(Node("case", cl, values=vals, body=self.stmt_list()))
        self.expect_op("}")
        if typeswitch:
            return Node("typeswitch", line, init=init, tag=tag, clauses=clauses)
        return Node("switch", line, init=init, tag=tag.x if tag is not None else None, clauses=clauses)

    def type_list(self):
        out = [self.type_()]
        while self.accept_op(","):
            out.append(self.type_())
        return out

    def select_stmt(self):
        line = self.tok.line
        self.expect_kw("select")
        self.expect_op("{")
        clauses = []
        while not self.is_op("}"):
            cl = self.tok.line
            comm = None
            if self.is_kw("default"):
                self.p += 1
            else:
                self.expect_kw("case")
                comm = self.simple_stmt()
            self.expect_op(":")
            clauses.append(Node("comm", cl, comm=comm, body=self.stmt_list()))
        self.expect_op("}")
        return Node("select", line, clauses=clauses)

    # --- expressions
    def expr_list(self):
        out = [self.expr()]
        while self.accept_op(","):
            out.append(self.expr())
        return out

    def expr(self, prec=1):
        x = self.unary()
        while self.tok.kind == "op" and BINARY_PREC.get(self.tok.val, 0) >= prec:
            op = self.tok.val
            line = self.tok.line
            self.p += 1
            y = self.expr(BINARY_PREC[op] + 1)
            x = Node("binary", line, op=op, x=x, y=y)
        return x

    def unary(self):
        line = self.tok.line
        if self.tok.kind == "op" and self.tok.val in ("+", "-", "!", "^", "*", "&", "<-"):
            op = self.tok.val
            self.p += 1
            if op == "<-" and self.is_kw("chan"):
                self.p -= 1
                return self.primary()
            x = self.unary()
            if op == "*":
                return Node("deref", line, x=x)
            if op == "&":
                return Node("addr", line, x=x)
            if op == "<-":
                return Node("recv", line, x=x)
            return Node("unary", line, op=op, x=x)
        return self.primary()

    def operand(self):
        line = self.tok.line
        t = self.tok
        if t.kind == "int":
            self.p += 1
            return Node("intlit", line, value=t.val)
        if t.kind == "float":
            self.p += 1
            return Node("floatlit", line, value=t.val)
        if t.kind == "rune":
            self.p += 1
            return Node("runelit", line, value=t.val)
        if t.kind == "string":
            self.p += 1
            return Node("stringlit", line, value=t.val)
        if t.kind == "ident":
            self.p += 1
            return Node("ident", line, name=t.val)
        if t.kind == "op" and t.val == "(":
            self.p += 1
            saved, self.expr_lev = self.expr_lev, 0
            x = self.expr_or_type()
            self.expr_lev = saved
            self.expect_op(")")
            return Node("paren", line, x=x)
        if t.kind == "kw" and t.val == "func":
            self.p += 1
            sig = self.signature()
            if self.is_op("{"):
                saved, self.expr_lev = self.expr_lev, 0
                body = self.block()
                self.expr_lev = saved
                return Node("funclit", line, sig=sig, body=body)
            return Node("typeexpr", line, type=Node("tfunc", line, sig=sig))
        if (t.kind == "op" and t.val == "[") or (t.kind == "kw" and t.val in ("map", "chan", "struct", "interface")) \
                or (t.kind == "op" and t.val == "<-"):
            return Node("typeexpr", line, type=self.type_())
        self.err("expected operand")

    def expr_or_type(self):
        if self.is_op("*"):
            # *T in parentheses is a pointer type when followed by ')' and then '(' (conversion); parse as expression:
            # the emitter treats deref-of-a-type-name as a pointer type.
            pass
        return self.expr()

    def primary(self):
        x = self.operand()
        while True:
            line = self.tok.line
            if self.is_op("."):
                self.p += 1
                if self.accept_op("("):
                    if self.is_kw("type"):
                        self.p += 1
                        t = None
                    else:
                        t = self.type_()
                    self.expect_op(")")
                    x = Node("typeassert", line, x=x, type=t)
                else:
                    x = Node("selector", line, x=x, sel=self.ident())
            elif self.is_op("["):
                self.p += 1
                saved, self.expr_lev = self.expr_lev, 0
                idx = [None, None, None]
                ncolon = 0
                if not self.is_op(":"):
                    idx[0] = self.expr_or_type()
                while self.accept_op(":"):
                    ncolon += 1
                    if ncolon > 2:
                        self.err("too many colons in slice expression")
                    if not self.is_op(":", "]"):
                        idx[ncolon] = self.expr()
                self.expr_lev = saved
                if ncolon == 0 and self.accept_op(","):
                    self.err("generic instantiation with several type arguments is not supported")
                self.expect_op("]")
                if ncolon == 0:
                    x = Node("index", line, x=x, index=idx[0])
                else:
                    x = Node("sliceexpr", line, x=x, lo=idx[0], hi=idx[1], max=idx[2], three=(ncolon == 2))
            elif self.is_op("("):
                self.p += 1
                saved, self.expr_lev = self.expr_lev, 0
                args = []
                ellipsis = False
                first_is_type = x.kind == "ident" and x.name in ("make", "new")
                while not self.is_op(")"):
                    if first_is_type and not args:
                        args.append(Node("typeexpr", self.tok.line, type=self.type_()))
                    else:
                        args.append(self.expr_or_type())
                    if self.accept_op("..."):
                        ellipsis = True
                    if not self.accept_op(","):
                        break
                self.expr_lev = saved
                self.expect_op(")")
                x = Node("call", line, fun=x, args=args, ellipsis=ellipsis)
            elif self.is_op("{") and self.literal_type_ok(x):
                x = self.composite(x)
            else:
                return x

    def literal_type_ok(self, x):
        if x.kind == "typeexpr":
            return x.type.kind in ("tslice", "tarray", "tmap", "tstruct")
        if self.expr_lev < 0:
            return False
        if x.kind == "ident":
            return True
        if x.kind == "selector" and x.x.kind == "ident":
            return True
        return x.kind == "index" and x.x.kind in ("ident", "selector")  # generic type instantiation

    def composite(self, typ):
        line = self.tok.line
        self.expect_op("{")
        saved, self.expr_lev = self.expr_lev, 0
        elems = []
        while not self.is_op("}"):
            k = None
            v = self.element()
            if self.accept_op(":"):
                k, v = v, self.element()
            elems.append((k, v))
            if not self.accept_op(","):
                break
        self.expr_lev = saved
        self.expect_op("}")
        return Node("composite", line, type=typ, elems=elems)

    def element(self):
        if self.is_op("{"):
            return self.composite(None)
        return self.expr()


def parse_source(src: str, filename: str = "<go>"):
    return Parser(src, filename).parse_file()
