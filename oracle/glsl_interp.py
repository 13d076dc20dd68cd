"""TEST INFRASTRUCTURE (oracle/): a small interpreter for the GLSL ES 3.00 subset the reference's shaders are written in.

Why: oracle/vpt_oracle.c is a restatement of the reference's fragment programs by this repo's author.  The reference holds no golden
vectors and its GLSL cannot run here (no WebGL driver), so nothing the reference itself *computed* pins the restatement.  This module
closes part of that gap: it EXECUTES THE REFERENCE'S OWN SHADER TEXT — read from the reference tree at fixture-generation time
(tests/golden/make_glsl_fixtures.py), never copied into this repository — fragment by fragment, on the CPU, and the results are committed
as data (tests/golden/glsl_*.json).  What the formulas, the control flow, the order of the random draws and the uniforms' meaning ARE is
then decided by the reference's text, not by a reading of it.  What stays this repo's choice (and is implementation-defined in WebGL too):
fp32 evaluation without contraction, IEEE division / sqrt, the transcendental routines (the numeric contract's log / exp / sin / cos /
atan / asin, oracle/vpt_oracle.c), texture filtering in fp32, linear interpolation of the varyings.

Supported: structs, functions with in / out / inout parameters and overloads, const / uniform / in / out globals, arrays and array
constructors, if / for / while / do-while / break / continue / return / discard, the operators of GLSL ES 3.00 on float / int / uint / bool
scalars, vectors and matrices, swizzles as l-values, `#define` of object-like macros, and the built-ins the reference's shaders call.
Mixins: `@path` substitution as src/js/WebGL.js:85-99 does it.  Only tests/ and tests/golden/ scripts import this module."""
import math
import os
import re

import numpy as np

F = np.float32


class GlslError(Exception):
    pass


# ---------------------------------------------------------------------------------------------------------------------------------
# values
# ---------------------------------------------------------------------------------------------------------------------------------
class UInt:
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = int(v) & 0xffffffff

    def __repr__(self):
        return "%du" % self.v


class Int:
    __slots__ = ("v",)

    def __init__(self, v):
        v = int(v) & 0xffffffff
        self.v = v - (1 << 32) if v & 0x80000000 else v

    def __repr__(self):
        return "%d" % self.v


class Vec:
    __slots__ = ("kind", "c")          # kind: 'f' float, 'u' uint, 'i' int, 'b' bool

    def __init__(self, kind, comps):
        self.kind = kind
        self.c = list(comps)

    def __repr__(self):
        return "%svec%d(%s)" % ({'f': '', 'u': 'u', 'i': 'i', 'b': 'b'}[self.kind], len(self.c), ", ".join(repr(x) for x in self.c))


class Mat:
    __slots__ = ("n", "cols")          # column-major: cols[j] is a float Vec of n components

    def __init__(self, n, cols):
        self.n = n
        self.cols = cols


class Struct:
    __slots__ = ("type", "f")

    def __init__(self, type_, fields):
        self.type = type_
        self.f = fields


class Array:
    __slots__ = ("elem", "items")

    def __init__(self, elem, items):
        self.elem = elem
        self.items = items


def type_of(v):
    if isinstance(v, np.floating):
        return "float"
    if isinstance(v, UInt):
        return "uint"
    if isinstance(v, Int):
        return "int"
    if isinstance(v, (bool, np.bool_)):
        return "bool"
    if isinstance(v, Vec):
        return {'f': 'vec', 'u': 'uvec', 'i': 'ivec', 'b': 'bvec'}[v.kind] + str(len(v.c))
    if isinstance(v, Mat):
        return "mat%d" % v.n
    if isinstance(v, Struct):
        return v.type
    if isinstance(v, Array):
        return v.elem + "[]"
    if isinstance(v, Sampler):
        return v.glsl_type
    raise GlslError("value of unknown type: %r" % (v,))


def copyval(v):
    if isinstance(v, Vec):
        return Vec(v.kind, v.c)
    if isinstance(v, Mat):
        return Mat(v.n, [Vec('f', c.c) for c in v.cols])
    if isinstance(v, Struct):
        return Struct(v.type, {k: copyval(x) for k, x in v.f.items()})
    if isinstance(v, Array):
        return Array(v.elem, [copyval(x) for x in v.items])
    return v


def scalar_kind(v):
    if isinstance(v, np.floating):
        return 'f'
    if isinstance(v, UInt):
        return 'u'
    if isinstance(v, Int):
        return 'i'
    if isinstance(v, (bool, np.bool_)):
        return 'b'
    return None


def convert_scalar(v, kind):
    """constructor conversion between scalar types (GLSL ES 3.00 section 5.4.1)"""
    k = scalar_kind(v)
    if k == kind:
        return v
    if kind == 'f':
        if k in ('u', 'i'):
            return F(v.v)
        return F(1.0) if v else F(0.0)
    if kind == 'u':
        if k == 'f':
            x = float(v)
            return UInt(int(x)) if math.isfinite(x) else UInt(0)        # truncation toward zero
        if k == 'i':
            return UInt(v.v)
        return UInt(1 if v else 0)
    if kind == 'i':
        if k == 'f':
            x = float(v)
            return Int(int(x)) if math.isfinite(x) else Int(0)
        if k == 'u':
            return Int(v.v)
        return Int(1 if v else 0)
    if kind == 'b':
        if k == 'f':
            return bool(v != 0)
        return bool(v.v != 0)
    raise GlslError("conversion to %r" % kind)


# ---------------------------------------------------------------------------------------------------------------------------------
# textures (what `texture()` reads): GL semantics in fp32, CLAMP_TO_EDGE
# ---------------------------------------------------------------------------------------------------------------------------------
class Sampler:
    """texels: float32 array [h][w][4] (2-D) or [d][h][w][4] (3-D), already converted from the storage format (UNORM8 / sRGB / float)"""

    def __init__(self, texels, linear, repeat=False):
        self.t = np.ascontiguousarray(texels, dtype=np.float32)
        self.dim = self.t.ndim - 1
        self.linear = bool(linear)
        self.repeat = bool(repeat)            # TEXTURE_WRAP_* = REPEAT (the GL default, what a texture created without wrap modes has)
        self.glsl_type = "sampler3D" if self.dim == 3 else "sampler2D"

    def _axis(self, coord, n):
        """texel indices and weight along one axis: OpenGL ES 3.0 section 3.8.10 (u = s * n; i0 = floor(u - 0.5), clamped to the edge)"""
        u = F(coord) * F(n)
        wrap = (lambda i: i % n) if self.repeat else (lambda i: min(max(i, 0), n - 1))
        if not self.linear:
            i = int(math.floor(float(u))) if math.isfinite(float(u)) else 0
            return wrap(i), wrap(i), F(0.0)
        um = u - F(0.5)
        fl = F(np.floor(um))
        w = um - fl
        if not math.isfinite(float(fl)):
            return 0, 0, F(0.0)
        i0 = int(fl)
        return wrap(i0), wrap(i0 + 1), w

    @staticmethod
    def _lerp(a, b, w):
        return a + (b - a) * w

    def sample(self, coord):
        if self.dim == 2:
            h, w = self.t.shape[:2]
            x0, x1, fx = self._axis(coord.c[0], w)
            y0, y1, fy = self._axis(coord.c[1], h)
            out = []
            for q in range(4):
                a = self._lerp(self.t[y0, x0, q], self.t[y0, x1, q], fx)
                b = self._lerp(self.t[y1, x0, q], self.t[y1, x1, q], fx)
                out.append(F(self._lerp(a, b, fy)))
            return Vec('f', out)
        d, h, w = self.t.shape[:3]
        x0, x1, fx = self._axis(coord.c[0], w)
        y0, y1, fy = self._axis(coord.c[1], h)
        z0, z1, fz = self._axis(coord.c[2], d)
        out = []
        for q in range(4):
            t = self.t
            a = self._lerp(t[z0, y0, x0, q], t[z0, y0, x1, q], fx)
            b = self._lerp(t[z0, y1, x0, q], t[z0, y1, x1, q], fx)
            c = self._lerp(t[z1, y0, x0, q], t[z1, y0, x1, q], fx)
            e = self._lerp(t[z1, y1, x0, q], t[z1, y1, x1, q], fx)
            ab = self._lerp(a, b, fy)
            ce = self._lerp(c, e, fy)
            out.append(F(self._lerp(ab, ce, fz)))
        return Vec('f', out)

    def fetch(self, ij):
        x, y = ij.c[0].v, ij.c[1].v
        h, w = self.t.shape[:2]
        if not (0 <= x < w and 0 <= y < h):
            return Vec('f', [F(0)] * 4)
        return Vec('f', [F(v) for v in self.t[y, x]])


# ---------------------------------------------------------------------------------------------------------------------------------
# lexer
# ---------------------------------------------------------------------------------------------------------------------------------
TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>(?:\d+\.\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+|0[xX][0-9a-fA-F]+|\d+)[uUfF]?)
  | (?P<id>[A-Za-z_]\w*)
  | (?P<op>\+\+|--|\+=|-=|\*=|/=|%=|<<=|>>=|&=|\|=|\^=|==|!=|<=|>=|&&|\|\||\^\^|<<|>>|[-+*/%<>=!&|^~?:;,.(){}\[\]])
""", re.X | re.S)

TYPE_WORDS = {"void", "float", "int", "uint", "bool", "vec2", "vec3", "vec4", "ivec2", "ivec3", "ivec4", "uvec2", "uvec3", "uvec4",
              "bvec2", "bvec3", "bvec4", "mat2", "mat3", "mat4", "sampler2D", "sampler3D"}
QUALIFIERS = {"const", "uniform", "in", "out", "inout", "highp", "mediump", "lowp", "flat", "smooth", "centroid"}


def lex(src):
    defines = {}
    lines = []
    for line in src.split("\n"):
        s = line.strip()
        if s.startswith("#"):
            m = re.match(r"#\s*define\s+(\w+)\s+(.*)$", s)
            if m:
                defines[m.group(1)] = m.group(2).strip()
            elif re.match(r"#\s*(version|extension|pragma)", s):
                pass
            else:
                raise GlslError("preprocessor line not supported: %s" % s)
            lines.append("")
        else:
            lines.append(line)
    text = "\n".join(lines)

    def tokens_of(t, depth=0):
        out = []
        pos = 0
        while pos < len(t):
            m = TOKEN.match(t, pos)
            if not m:
                raise GlslError("cannot tokenise at: %r" % t[pos:pos + 30])
            pos = m.end()
            if m.lastgroup == "ws":
                continue
            if m.lastgroup == "id" and m.group() in defines and depth < 8:
                out.extend(tokens_of(defines[m.group()], depth + 1))
                continue
            out.append((m.lastgroup, m.group()))
        return out
    toks = tokens_of(text)
    toks.append(("eof", ""))
    return toks


# ---------------------------------------------------------------------------------------------------------------------------------
# parser -> tuples
# ---------------------------------------------------------------------------------------------------------------------------------
class Parser:
    def __init__(self, toks):
        self.t = toks
        self.p = 0
        self.structs = {}

    def peek(self, k=0):
        return self.t[self.p + k]

    def next(self):
        tok = self.t[self.p]
        self.p += 1
        return tok

    def accept(self, val):
        if self.t[self.p][1] == val and self.t[self.p][0] != "num":
            self.p += 1
            return True
        return False

    def expect(self, val):
        if not self.accept(val):
            raise GlslError("expected %r, found %r (token %d)" % (val, self.t[self.p][1], self.p))

    def is_type(self, tok):
        return tok[0] == "id" and (tok[1] in TYPE_WORDS or tok[1] in self.structs)

    # ---- top level ----
    def program(self):
        decls = []
        while self.peek()[0] != "eof":
            if self.accept(";"):
                continue
            if self.peek()[1] == "precision":
                while not self.accept(";"):
                    self.next()
                continue
            if self.peek()[1] == "struct":
                decls.append(self.struct())
                continue
            decls.append(self.global_decl())
        return decls

    def struct(self):
        self.expect("struct")
        name = self.next()[1]
        self.expect("{")
        fields = []
        while not self.accept("}"):
            while self.peek()[1] in QUALIFIERS:
                self.next()
            ftype = self.next()[1]
            while True:
                fields.append((ftype, self.next()[1]))
                if not self.accept(","):
                    break
            self.expect(";")
        self.expect(";")
        self.structs[name] = fields
        return ("struct", name, fields)

    def global_decl(self):
        quals = []
        if self.peek()[1] == "layout":
            self.next()
            self.expect("(")
            depth = 1
            while depth:
                tok = self.next()[1]
                depth += (tok == "(") - (tok == ")")
        while self.peek()[1] in QUALIFIERS:
            quals.append(self.next()[1])
        typ = self.next()[1]
        name = self.next()[1]
        if self.accept("("):                                  # function
            params = []
            if not self.accept(")"):
                while True:
                    pq = "in"
                    while self.peek()[1] in QUALIFIERS:
                        q = self.next()[1]
                        if q in ("in", "out", "inout"):
                            pq = q
                    ptype = self.next()[1]
                    if ptype == "void" and self.peek()[1] == ")":
                        self.next()
                        break
                    pname = self.next()[1]
                    params.append((pq, ptype, pname))
                    if self.accept(")"):
                        break
                    self.expect(",")
            if self.accept(";"):
                return ("proto", name)
            body = self.block()
            return ("func", typ, name, params, body)
        size = None
        if self.accept("["):
            if not self.accept("]"):
                size = self.expr()
                self.expect("]")
            else:
                size = ("unsized",)
        init = None
        if self.accept("="):
            init = self.assign_expr()
        self.expect(";")
        return ("global", quals, typ, name, size, init)

    # ---- statements ----
    def block(self):
        self.expect("{")
        body = []
        while not self.accept("}"):
            body.append(self.statement())
        return ("block", body)

    def statement(self):
        tok = self.peek()
        v = tok[1]
        if v == "{":
            return self.block()
        if v == ";":
            self.next()
            return ("block", [])
        if v == "if":
            self.next(); self.expect("(")
            c = self.expr(); self.expect(")")
            a = self.statement()
            b = self.statement() if self.accept("else") else None
            return ("if", c, a, b)
        if v == "for":
            self.next(); self.expect("(")
            init = None
            if not self.accept(";"):
                init = self.simple_statement()
            cond = None
            if not self.accept(";"):
                cond = self.expr(); self.expect(";")
            step = None
            if not self.accept(")"):
                step = self.expr(); self.expect(")")
            return ("for", init, cond, step, self.statement())
        if v == "while":
            self.next(); self.expect("(")
            c = self.expr(); self.expect(")")
            return ("while", c, self.statement())
        if v == "do":
            self.next()
            body = self.statement()
            self.expect("while"); self.expect("(")
            c = self.expr(); self.expect(")"); self.expect(";")
            return ("dowhile", body, c)
        if v == "return":
            self.next()
            e = None
            if not self.accept(";"):
                e = self.expr(); self.expect(";")
            return ("return", e)
        if v in ("break", "continue", "discard"):
            self.next(); self.expect(";")
            return (v,)
        return self.simple_statement()

    def simple_statement(self):
        """declaration or expression statement, terminated by ';'"""
        save = self.p
        const = False
        while self.peek()[1] in QUALIFIERS:
            const = True
            self.next()
        if self.is_type(self.peek()) and self.peek(1)[0] == "id":
            typ = self.next()[1]
            decls = []
            while True:
                name = self.next()[1]
                size = None
                if self.accept("["):
                    size = self.expr(); self.expect("]")
                init = self.assign_expr() if self.accept("=") else None
                decls.append((name, size, init))
                if not self.accept(","):
                    break
            self.expect(";")
            return ("decl", typ, decls)
        if const:
            self.p = save
        e = self.expr()
        self.expect(";")
        return ("expr", e)

    # ---- expressions ----
    def expr(self):
        e = self.assign_expr()
        while self.accept(","):
            e = ("seq", e, self.assign_expr())
        return e

    def assign_expr(self):
        lhs = self.ternary()
        tok = self.peek()[1]
        if self.peek()[0] == "op" and tok in ("=", "+=", "-=", "*=", "/=", "%=", "<<=", ">>=", "&=", "|=", "^="):
            self.next()
            rhs = self.assign_expr()
            return ("assign", tok, lhs, rhs)
        return lhs

    def ternary(self):
        c = self.binary(0)
        if self.accept("?"):
            a = self.assign_expr()
            self.expect(":")
            b = self.assign_expr()
            return ("cond", c, a, b)
        return c

    LEVELS = [["||"], ["^^"], ["&&"], ["|"], ["^"], ["&"], ["==", "!="], ["<", ">", "<=", ">="], ["<<", ">>"], ["+", "-"], ["*", "/", "%"]]

    def binary(self, level):
        if level == len(self.LEVELS):
            return self.unary()
        e = self.binary(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.LEVELS[level]:
            op = self.next()[1]
            r = self.binary(level + 1)
            e = ("bin", op, e, r)
        return e

    def unary(self):
        tok = self.peek()
        if tok[0] == "op" and tok[1] in ("-", "+", "!", "~"):
            self.next()
            return ("un", tok[1], self.unary())
        if tok[0] == "op" and tok[1] in ("++", "--"):
            self.next()
            return ("preinc", tok[1], self.unary())
        return self.postfix()

    def postfix(self):
        e = self.primary()
        while True:
            if self.accept("."):
                e = ("member", e, self.next()[1])
            elif self.accept("["):
                i = self.expr(); self.expect("]")
                e = ("index", e, i)
            elif self.peek()[0] == "op" and self.peek()[1] in ("++", "--"):
                e = ("postinc", self.next()[1], e)
            else:
                return e

    def primary(self):
        kind, v = self.next()
        if kind == "num":
            if v[-1] in "uU":
                return ("lit", UInt(int(v[:-1], 0)))
            if re.match(r"^(0[xX][0-9a-fA-F]+|\d+)$", v):
                return ("lit", Int(int(v, 0)))
            return ("lit", F(v.rstrip("fF")))
        if kind == "op" and v == "(":
            e = self.expr(); self.expect(")")
            return e
        if kind == "id":
            if v in ("true", "false"):
                return ("lit", v == "true")
            if self.peek()[1] == "[" and self.peek(1)[1] == "]" and self.is_type((kind, v)):      # array constructor: T[](...)
                self.next(); self.next()
                self.expect("(")
                args = self.args()
                return ("arrayctor", v, args)
            if self.accept("("):
                return ("call", v, self.args())
            return ("var", v)
        raise GlslError("unexpected token %r" % v)

    def args(self):
        args = []
        if self.accept(")"):
            return args
        if self.peek()[1] == "void" and self.peek(1)[1] == ")":
            self.next(); self.next()
            return args
        while True:
            args.append(self.assign_expr())
            if self.accept(")"):
                return args
            self.expect(",")


# ---------------------------------------------------------------------------------------------------------------------------------
# evaluation
# ---------------------------------------------------------------------------------------------------------------------------------
class _Break(Exception):
    pass


class _Continue(Exception):
    pass


class _Return(Exception):
    def __init__(self, v):
        self.v = v


class Discard(Exception):
    pass


SWZ = {c: i for s in ("xyzw", "rgba", "stpq") for i, c in enumerate(s)}


def default_value(typ, structs):
    if typ == "float":
        return F(0)
    if typ == "uint":
        return UInt(0)
    if typ == "int":
        return Int(0)
    if typ == "bool":
        return False
    m = re.match(r"^([iub]?)vec([234])$", typ)
    if m:
        k = m.group(1) or 'f'
        return Vec(k, [convert_scalar(F(0), k)] * int(m.group(2)))
    m = re.match(r"^mat([234])$", typ)
    if m:
        n = int(m.group(1))
        return Mat(n, [Vec('f', [F(0)] * n) for _ in range(n)])
    if typ in structs:
        return Struct(typ, {n: default_value(t, structs) for t, n in structs[typ]})
    raise GlslError("no default value for type %s" % typ)


class Math:
    """the transcendental routines: by default the numeric contract's (oracle/vpt_oracle.c, through oracle/oracle.py), so that what
    differs between this interpreter and the C oracle is the shader's own structure, not the implementation-defined precision of log / sin"""

    def __init__(self, contract=True):
        self.L = None
        if contract:
            from . import oracle as O
            self.L = O.lib()

    def log(self, x):
        return F(self.L.vpo_logf(float(x))) if self.L else F(np.log(F(x)))

    def exp(self, x):
        return F(self.L.vpo_expf(float(x))) if self.L else F(np.exp(F(x)))

    def pow(self, x, y):
        return F(self.L.vpo_powf(float(x), float(y))) if self.L else F(np.power(F(x), F(y)))

    def sincos(self, x):
        if self.L:
            import ctypes as C
            s, c = C.c_float(), C.c_float()
            self.L.vpo_sincosf(float(x), C.byref(s), C.byref(c))
            return F(s.value), F(c.value)
        return F(np.sin(F(x))), F(np.cos(F(x)))

    def atan2(self, y, x):
        return F(self.L.vpo_atan2f(float(y), float(x))) if self.L else F(np.arctan2(F(y), F(x)))

    def asin(self, x):
        return F(self.L.vpo_asinf(float(x))) if self.L else F(np.arcsin(F(x)))


class Shader:
    """one compiled stage (vertex or fragment): globals by qualifier, functions by (name, parameter types)"""

    def __init__(self, source, math=None):
        self.math = math or Math()
        parser = Parser(lex(source))
        self.decls = parser.program()
        self.structs = parser.structs
        self.funcs = {}
        self.globals_decl = []
        for d in self.decls:
            if d[0] == "func":
                self.funcs.setdefault(d[2], []).append(d)
            elif d[0] == "global":
                self.globals_decl.append(d)
        self.globals = {}
        self.uniforms = {}
        self.inputs = {}
        self.builtin_in = {}

    # ---- running ----
    def run(self, uniforms, inputs, builtin=None):
        """executes main(); returns {out variable: value} (fragment) / ({varying: value}, gl_Position) is read from the same dict"""
        self.globals = {}
        self.scopes = []
        self.builtin_in = builtin or {}
        outs = []
        with np.errstate(all="ignore"):
            for (_, quals, typ, name, size, init) in self.globals_decl:
                if "uniform" in quals:
                    if name not in uniforms:
                        raise GlslError("uniform %s not set" % name)
                    self.globals[name] = uniforms[name]
                elif "in" in quals:
                    if name not in inputs:
                        raise GlslError("input %s not set" % name)
                    self.globals[name] = copyval(inputs[name])
                else:
                    if init is not None:
                        v = self.eval(init)
                        if size is not None:
                            if not isinstance(v, Array):
                                raise GlslError("array initialiser expected for %s" % name)
                        else:
                            v = self.construct(typ, [v]) if type_of(v) != typ else v
                        self.globals[name] = v
                    elif size is not None:
                        n = self.eval(size).v
                        self.globals[name] = Array(typ, [default_value(typ, self.structs) for _ in range(n)])
                    else:
                        self.globals[name] = default_value(typ, self.structs)
                    if "out" in quals:
                        outs.append(name)
            main = self.funcs["main"][0]
            try:
                self.call_user(main, [])
            except _Return:
                pass
        res = {n: self.globals[n] for n in outs}
        for n in ("gl_Position",):
            if n in self.globals:
                res[n] = self.globals[n]
        return res

    # ---- variables ----
    def lookup(self, name):
        for s in reversed(self.scopes):
            if name in s:
                return s
        if name in self.globals:
            return self.globals
        if name in self.builtin_in:
            return self.builtin_in
        if name == "gl_Position":
            self.globals[name] = Vec('f', [F(0)] * 4)
            return self.globals
        raise GlslError("undeclared identifier %s" % name)

    # ---- statements ----
    def exec(self, s):
        k = s[0]
        if k == "block":
            self.scopes.append({})
            try:
                for x in s[1]:
                    self.exec(x)
            finally:
                self.scopes.pop()
        elif k == "expr":
            self.eval(s[1])
        elif k == "decl":
            typ = s[1]
            for (name, size, init) in s[2]:
                if size is not None:
                    n = self.eval(size).v
                    v = Array(typ, [default_value(typ, self.structs) for _ in range(n)])
                    if init is not None:
                        v = copyval(self.eval(init))
                elif init is not None:
                    v = copyval(self.eval(init))
                    if type_of(v) != typ:
                        raise GlslError("initialiser of %s %s has type %s" % (typ, name, type_of(v)))
                else:
                    v = default_value(typ, self.structs)
                self.scopes[-1][name] = v
        elif k == "if":
            if self.truth(self.eval(s[1])):
                self.exec(s[2])
            elif s[3] is not None:
                self.exec(s[3])
        elif k == "for":
            self.scopes.append({})
            try:
                if s[1] is not None:
                    self.exec(s[1])
                while s[2] is None or self.truth(self.eval(s[2])):
                    try:
                        self.exec(s[4])
                    except _Break:
                        break
                    except _Continue:
                        pass
                    if s[3] is not None:
                        self.eval(s[3])
            finally:
                self.scopes.pop()
        elif k == "while":
            while self.truth(self.eval(s[1])):
                try:
                    self.exec(s[2])
                except _Break:
                    break
                except _Continue:
                    pass
        elif k == "dowhile":
            while True:
                try:
                    self.exec(s[1])
                except _Break:
                    break
                except _Continue:
                    pass
                if not self.truth(self.eval(s[2])):
                    break
        elif k == "return":
            raise _Return(copyval(self.eval(s[1])) if s[1] is not None else None)
        elif k == "break":
            raise _Break()
        elif k == "continue":
            raise _Continue()
        elif k == "discard":
            raise Discard()
        else:
            raise GlslError("statement %s" % k)

    @staticmethod
    def truth(v):
        if not isinstance(v, (bool, np.bool_)):
            raise GlslError("condition is not a bool: %r" % (v,))
        return bool(v)

    # ---- l-values ----
    def store(self, target, value):
        k = target[0]
        if k == "var":
            scope = self.lookup(target[1])
            old = scope[target[1]]
            if type_of(old) != type_of(value):
                raise GlslError("assignment of %s to %s %s" % (type_of(value), type_of(old), target[1]))
            scope[target[1]] = copyval(value)
        elif k == "member":
            base = self.eval(target[1])                     # containers are mutable: evaluating returns the object itself
            name = target[2]
            if isinstance(base, Struct):
                if type_of(base.f[name]) != type_of(value):
                    raise GlslError("assignment of %s to field %s" % (type_of(value), name))
                base.f[name] = copyval(value)
            elif isinstance(base, Vec):
                idx = [SWZ[c] for c in name]
                if len(set(idx)) != len(idx):
                    raise GlslError("swizzle %s repeats a component on the left-hand side" % name)
                if len(idx) == 1:
                    if scalar_kind(value) != base.kind:
                        raise GlslError("assignment of %s to component of %s" % (type_of(value), type_of(base)))
                    base.c[idx[0]] = value
                else:
                    if not isinstance(value, Vec) or len(value.c) != len(idx) or value.kind != base.kind:
                        raise GlslError("assignment of %s to swizzle .%s" % (type_of(value), name))
                    for i, j in enumerate(idx):
                        base.c[j] = value.c[i]
            else:
                raise GlslError("member store into %s" % type_of(base))
        elif k == "index":
            base = self.eval(target[1])
            i = self.eval(target[2]).v
            if isinstance(base, Array):
                base.items[i] = copyval(value)
            elif isinstance(base, Vec):
                base.c[i] = value
            elif isinstance(base, Mat):
                base.cols[i] = copyval(value)
            else:
                raise GlslError("index store into %s" % type_of(base))
        else:
            raise GlslError("not an l-value: %s" % k)

    # ---- expressions ----
    def eval(self, e):
        k = e[0]
        if k == "lit":
            return e[1]
        if k == "var":
            return self.lookup(e[1])[e[1]]
        if k == "bin":
            op = e[1]
            if op == "&&":
                return self.truth(self.eval(e[2])) and self.truth(self.eval(e[3]))
            if op == "||":
                return self.truth(self.eval(e[2])) or self.truth(self.eval(e[3]))
            return self.binop(op, self.eval(e[2]), self.eval(e[3]))
        if k == "un":
            return self.unop(e[1], self.eval(e[2]))
        if k == "assign":
            op = e[1]
            rhs = self.eval(e[3])
            if op != "=":
                rhs = self.binop(op[:-1], self.eval(e[2]), rhs)
            self.store(e[2], rhs)
            return rhs
        if k in ("preinc", "postinc"):
            old = self.eval(e[2])
            one = convert_scalar(Int(1), scalar_kind(old) if scalar_kind(old) else old.kind)
            new = self.binop("+" if e[1] == "++" else "-", old, one)
            self.store(e[2], new)
            return new if k == "preinc" else old
        if k == "cond":
            return self.eval(e[2]) if self.truth(self.eval(e[1])) else self.eval(e[3])
        if k == "seq":
            self.eval(e[1])
            return self.eval(e[2])
        if k == "member":
            base = self.eval(e[1])
            name = e[2]
            if isinstance(base, Struct):
                return base.f[name]
            if isinstance(base, Vec):
                idx = [SWZ[c] for c in name]
                if len(idx) == 1:
                    return base.c[idx[0]]
                return Vec(base.kind, [base.c[i] for i in idx])
            if isinstance(base, Array) and name == "length":
                return ("lengthof", base)
            raise GlslError("member %s of %s" % (name, type_of(base)))
        if k == "index":
            base = self.eval(e[1])
            i = self.eval(e[2]).v
            if isinstance(base, Array):
                return base.items[i]
            if isinstance(base, Vec):
                return base.c[i]
            if isinstance(base, Mat):
                return base.cols[i]
            raise GlslError("index into %s" % type_of(base))
        if k == "arrayctor":
            return Array(e[1], [copyval(self.eval(a)) for a in e[2]])
        if k == "call":
            return self.call(e[1], e[2])
        raise GlslError("expression %s" % k)

    # ---- operators ----
    def unop(self, op, v):
        if isinstance(v, Vec):
            return Vec(v.kind, [self.unop(op, x) for x in v.c])
        if op == "+":
            return v
        if op == "-":
            if isinstance(v, np.floating):
                return F(-v)
            if isinstance(v, UInt):
                return UInt(-v.v)
            if isinstance(v, Int):
                return Int(-v.v)
        if op == "!":
            return not self.truth(v)
        if op == "~":
            if isinstance(v, UInt):
                return UInt(~v.v)
            if isinstance(v, Int):
                return Int(~v.v)
        raise GlslError("unary %s on %s" % (op, type_of(v)))

    def scalar_binop(self, op, a, b):
        ka, kb = scalar_kind(a), scalar_kind(b)
        if ka != kb:
            raise GlslError("operands of %s have types %s and %s (GLSL ES has no implicit conversions)" % (op, type_of(a), type_of(b)))
        if ka == 'f':
            if op == "+":
                return F(a + b)
            if op == "-":
                return F(a - b)
            if op == "*":
                return F(a * b)
            if op == "/":
                return F(a / b)
            if op == "<":
                return bool(a < b)
            if op == ">":
                return bool(a > b)
            if op == "<=":
                return bool(a <= b)
            if op == ">=":
                return bool(a >= b)
            if op == "==":
                return bool(a == b)
            if op == "!=":
                return bool(a != b)
        elif ka in ('u', 'i'):
            T = UInt if ka == 'u' else Int
            x, y = a.v, b.v
            if op == "+":
                return T(x + y)
            if op == "-":
                return T(x - y)
            if op == "*":
                return T(x * y)
            if op == "/":
                return T(int(x / y) if y else 0)
            if op == "%":
                return T(x - y * int(x / y) if y else 0)
            if op == "&":
                return T(x & y)
            if op == "|":
                return T(x | y)
            if op == "^":
                return T(x ^ y)
            if op in ("<", ">", "<=", ">=", "==", "!="):
                return {"<": x < y, ">": x > y, "<=": x <= y, ">=": x >= y, "==": x == y, "!=": x != y}[op]
        elif ka == 'b':
            if op == "==":
                return a == b
            if op == "!=":
                return a != b
            if op == "^^":
                return a != b
        raise GlslError("operator %s on %s" % (op, type_of(a)))

    def binop(self, op, a, b):
        if op in ("<<", ">>"):                              # shifts: the right operand may be of the other integer type
            if isinstance(a, Vec):
                return Vec(a.kind, [self.binop(op, x, b.c[i] if isinstance(b, Vec) else b) for i, x in enumerate(a.c)])
            T = type(a)
            n = b.v
            if isinstance(a, UInt):
                return UInt(a.v << n) if op == "<<" else UInt(a.v >> n)
            return Int(a.v << n) if op == "<<" else Int(a.v >> n)
        if isinstance(a, Mat) or isinstance(b, Mat):
            return self.mat_binop(op, a, b)
        if isinstance(a, Vec) and isinstance(b, Vec):
            if len(a.c) != len(b.c):
                raise GlslError("vector sizes differ in %s" % op)
            if op in ("==", "!="):
                eq = all(self.scalar_binop("==", x, y) for x, y in zip(a.c, b.c))
                return eq if op == "==" else not eq
            return Vec(a.kind, [self.scalar_binop(op, x, y) for x, y in zip(a.c, b.c)])
        if isinstance(a, Vec):
            return Vec(a.kind, [self.scalar_binop(op, x, b) for x in a.c])
        if isinstance(b, Vec):
            return Vec(b.kind, [self.scalar_binop(op, a, y) for y in b.c])
        return self.scalar_binop(op, a, b)

    def mat_binop(self, op, a, b):
        if op == "*" and isinstance(a, Mat) and isinstance(b, Vec):          # column vector: r[i] = sum_j a[j][i] * b[j], left to right
            out = []
            for i in range(a.n):
                acc = F(a.cols[0].c[i] * b.c[0])
                for j in range(1, a.n):
                    acc = F(acc + F(a.cols[j].c[i] * b.c[j]))
                out.append(acc)
            return Vec('f', out)
        if op == "*" and isinstance(a, Vec) and isinstance(b, Mat):          # row vector
            return Vec('f', [self.dot(a, b.cols[j]) for j in range(b.n)])
        if op == "*" and isinstance(a, Mat) and isinstance(b, Mat):
            return Mat(a.n, [self.mat_binop("*", a, b.cols[j]) for j in range(a.n)])
        if isinstance(a, Mat) and isinstance(b, Mat):
            return Mat(a.n, [self.binop(op, x, y) for x, y in zip(a.cols, b.cols)])
        if isinstance(a, Mat):
            return Mat(a.n, [self.binop(op, x, b) for x in a.cols])
        return Mat(b.n, [self.binop(op, a, y) for y in b.cols])

    def dot(self, a, b):
        if not isinstance(a, Vec):
            return F(a * b)
        acc = F(a.c[0] * b.c[0])
        for x, y in zip(a.c[1:], b.c[1:]):
            acc = F(acc + F(x * y))
        return acc

    # ---- constructors ----
    def construct(self, typ, args):
        if typ in ("float", "int", "uint", "bool"):
            kind = {"float": 'f', "int": 'i', "uint": 'u', "bool": 'b'}[typ]
            v = args[0]
            if isinstance(v, Vec):
                v = v.c[0]
            return convert_scalar(v, kind)
        m = re.match(r"^([iub]?)vec([234])$", typ)
        if m:
            kind, n = m.group(1) or 'f', int(m.group(2))
            flat = []
            for a in args:
                if isinstance(a, Vec):
                    flat.extend(a.c)
                elif isinstance(a, Mat):
                    for c in a.cols:
                        flat.extend(c.c)
                else:
                    flat.append(a)
            flat = [convert_scalar(x, kind) for x in flat]
            if len(args) == 1 and not isinstance(args[0], (Vec, Mat)):
                return Vec(kind, flat * n)
            if len(flat) < n:
                raise GlslError("%s from %d components" % (typ, len(flat)))
            if len(flat) > n and not (len(args) == 1):
                raise GlslError("%s from %d components" % (typ, len(flat)))
            return Vec(kind, flat[:n])
        m = re.match(r"^mat([234])$", typ)
        if m:
            n = int(m.group(1))
            if len(args) == 1 and not isinstance(args[0], (Vec, Mat)):
                d = convert_scalar(args[0], 'f')
                return Mat(n, [Vec('f', [d if i == j else F(0) for i in range(n)]) for j in range(n)])
            flat = []
            for a in args:
                flat.extend(a.c if isinstance(a, Vec) else [a])
            flat = [convert_scalar(x, 'f') for x in flat]
            if len(flat) != n * n:
                raise GlslError("%s from %d components" % (typ, len(flat)))
            return Mat(n, [Vec('f', flat[j * n:(j + 1) * n]) for j in range(n)])
        if typ in self.structs:
            fields = self.structs[typ]
            return Struct(typ, {n: copyval(a) for (_, n), a in zip(fields, args)})
        raise GlslError("constructor %s" % typ)

    # ---- calls ----
    def call(self, name, arg_exprs):
        if name in TYPE_WORDS or name in self.structs:
            return self.construct(name, [self.eval(a) for a in arg_exprs])
        if name in self.funcs:
            args = [self.eval(a) for a in arg_exprs]
            types = [type_of(a) for a in args]
            for f in self.funcs[name]:
                if [p[1] for p in f[3]] == types:
                    return self.call_user(f, args, arg_exprs)
            raise GlslError("no overload of %s for (%s)" % (name, ", ".join(types)))
        args = [self.eval(a) for a in arg_exprs]
        fn = getattr(self, "bi_" + name, None)
        if fn is None:
            raise GlslError("unknown function %s" % name)
        return fn(*args)

    def call_user(self, f, args, arg_exprs=None):
        params = f[3]
        frame = {}
        for (q, t, n), a in zip(params, args):
            frame[n] = copyval(a) if q in ("in", "inout") else default_value(t, self.structs)
        saved = self.scopes
        self.scopes = [frame]
        ret = None
        try:
            try:
                for s in f[4][1]:
                    self.exec(s)
            except _Return as r:
                ret = r.v
        finally:
            self.scopes = saved
        for i, (q, t, n) in enumerate(params):               # copy-out, in parameter order (GLSL ES 3.00 section 6.1.1)
            if q in ("out", "inout"):
                self.store(arg_exprs[i], frame[n])
        if f[1] != "void" and ret is None:
            raise GlslError("function %s returned nothing" % f[2])
        return ret

    # ---- built-ins (GLSL ES 3.00 chapter 8), componentwise over genType ----
    @staticmethod
    def _map(fn, *args):
        vec = next((a for a in args if isinstance(a, Vec)), None)
        if vec is None:
            return fn(*args)
        n = len(vec.c)
        cols = [(a.c if isinstance(a, Vec) else [a] * n) for a in args]
        out = [fn(*[c[i] for c in cols]) for i in range(n)]
        return Vec(scalar_kind(out[0]), out)

    def bi_abs(self, x):
        return self._map(lambda a: F(abs(a)), x)

    def bi_floor(self, x):
        return self._map(lambda a: F(np.floor(a)), x)

    def bi_fract(self, x):
        return self._map(lambda a: F(a - F(np.floor(a))), x)

    def bi_mod(self, x, y):
        return self._map(lambda a, b: F(a - F(b * F(np.floor(F(a / b))))), x, y)

    def bi_min(self, x, y):
        return self._map(lambda a, b: b if b < a else a, x, y)            # section 8.3: y if y < x, otherwise x

    def bi_max(self, x, y):
        return self._map(lambda a, b: b if a < b else a, x, y)            # y if x < y, otherwise x

    def bi_clamp(self, x, lo, hi):
        return self.bi_min(self.bi_max(x, lo), hi)

    def bi_mix(self, x, y, a):
        return self._map(lambda p, q, t: F(F(p * F(F(1.0) - t)) + F(q * t)), x, y, a)      # x * (1 - a) + y * a

    def bi_step(self, edge, x):
        return self._map(lambda e, a: F(0.0) if a < e else F(1.0), edge, x)

    def bi_smoothstep(self, e0, e1, x):
        def f(a, b, v):
            t = F(F(v - a) / F(b - a))
            t = F(0.0) if t < 0 else (F(1.0) if t > 1 else t)
            return F(F(t * t) * F(F(3.0) - F(F(2.0) * t)))
        return self._map(f, e0, e1, x)

    def bi_sqrt(self, x):
        return self._map(lambda a: F(np.sqrt(a)), x)

    def bi_inversesqrt(self, x):
        return self._map(lambda a: F(F(1.0) / F(np.sqrt(a))), x)

    def bi_exp(self, x):
        return self._map(self.math.exp, x)

    def bi_log(self, x):
        return self._map(self.math.log, x)

    def bi_pow(self, x, y):
        return self._map(self.math.pow, x, y)

    def bi_sin(self, x):
        return self._map(lambda a: self.math.sincos(a)[0], x)

    def bi_cos(self, x):
        return self._map(lambda a: self.math.sincos(a)[1], x)

    def bi_asin(self, x):
        return self._map(self.math.asin, x)

    def bi_atan(self, y, x=None):
        if x is None:
            return self._map(lambda a: self.math.atan2(a, F(1.0)), y)
        return self._map(self.math.atan2, y, x)

    def bi_dot(self, a, b):
        return self.dot(a, b)

    def bi_length(self, a):
        return F(np.sqrt(self.dot(a, a)))

    def bi_distance(self, a, b):
        return self.bi_length(self.binop("-", a, b))

    def bi_normalize(self, a):
        return self.binop("/", a, self.bi_length(a))

    def bi_any(self, v):
        return any(bool(x) for x in v.c)

    def bi_all(self, v):
        return all(bool(x) for x in v.c)

    def bi_lessThan(self, a, b):
        return Vec('b', [self.scalar_binop("<", x, y) for x, y in zip(a.c, b.c)])

    def bi_greaterThan(self, a, b):
        return Vec('b', [self.scalar_binop(">", x, y) for x, y in zip(a.c, b.c)])

    def bi_floatBitsToUint(self, x):
        return self._map(lambda a: UInt(int(np.array(a, dtype=np.float32).view(np.uint32))), x)

    def bi_uintBitsToFloat(self, x):
        return self._map(lambda a: F(np.array(a.v, dtype=np.uint32).view(np.float32)), x)

    def bi_texture(self, sampler, coord):
        return sampler.sample(coord)

    def bi_texelFetch(self, sampler, ij, lod):
        return sampler.fetch(ij)


# ---------------------------------------------------------------------------------------------------------------------------------
# the reference's shader files: `// #part <path>` sections (bin/packer:57-71) and `@mixin` substitution (src/js/WebGL.js:85-99)
# ---------------------------------------------------------------------------------------------------------------------------------
def read_parts(glsl_root):
    """{part path: text} of every .glsl file under glsl_root (the reference's src/glsl)"""
    parts = {}
    for dirpath, _, files in os.walk(glsl_root):
        for fn in sorted(files):
            if not fn.endswith(".glsl"):
                continue
            cur = None
            for line in open(os.path.join(dirpath, fn), encoding="utf-8").read().split("\n"):
                m = re.match(r"^\s*//\s*#part\s+(\S+)\s*$", line)
                if m:
                    cur = m.group(1)
                    parts[cur] = []
                elif cur is not None:
                    parts[cur].append(line)
    return {k: "\n".join(v) for k, v in parts.items()}


def cook(parts, shader_path):
    """the text WebGL.buildPrograms compiles for `shader_path` (e.g. /glsl/shaders/renderers/MCM/integrate/fragment)"""
    def sub(m):
        key = "/glsl/mixins/" + m.group(1)
        if key not in parts:
            raise GlslError("mixin %s not found" % key)
        return parts[key]
    return re.sub(r"@(\S+)", sub, parts[shader_path])


class Program:
    """vertex + fragment stage of one of the reference's programs over the full-screen triangle (-1,-1), (3,-1), (-1,3)"""

    def __init__(self, parts, name, math=None):
        self.vs = Shader(cook(parts, name + "/vertex"), math)
        self.fs = Shader(cook(parts, name + "/fragment"), math)

    def varyings(self, uniforms):
        """the vertex stage at the triangle's three corners"""
        outs = []
        for vid in range(3):
            outs.append(self.vs.run(uniforms, {}, {"gl_VertexID": Int(vid)}))
        return outs

    @staticmethod
    def interpolate(corner_values, x_ndc, y_ndc):
        """gl_Position.w = 1 at every corner: the interpolation is linear in window space.  Barycentric weights of (x, y) in the triangle
        (-1,-1), (3,-1), (-1,3), evaluated in float64 and rounded once — hardware interpolators differ in the last bits; the varyings of
        the reference's programs are affine in (x, y), so this is the value every exact interpolator would give"""
        w1 = (float(x_ndc) + 1.0) / 4.0
        w2 = (float(y_ndc) + 1.0) / 4.0
        w0 = 1.0 - w1 - w2
        a, b, c = corner_values
        if isinstance(a, Vec):
            return Vec('f', [F(w0 * float(p) + w1 * float(q) + w2 * float(r)) for p, q, r in zip(a.c, b.c, c.c)])
        return F(w0 * float(a) + w1 * float(b) + w2 * float(c))

    def fragment(self, uniforms, corners, i, j, width, height):
        """the fragment stage at pixel (i, j) (j = 0: bottom row, as gl_FragCoord counts)"""
        # the pixel centre in normalised device coordinates, as the numeric contract rounds it (oracle/vpt_oracle.c pixel_ndc: fp32
        # (2i + 1) / n - 1).  A rasteriser's own rounding of the varyings is implementation-defined, and the stochastic programs hash the
        # varying's BITS into their seed (MCMRenderer.glsl:129): another rounding is another random stream, not another algorithm
        x = F(F(2 * i + 1) / F(width)) - F(1.0)
        y = F(F(2 * j + 1) / F(height)) - F(1.0)
        inputs = {}
        for name in corners[0]:
            if name == "gl_Position":
                continue
            inputs[name] = self.interpolate([c[name] for c in corners], x, y)
        fs_in = {d[3] for d in self.fs.globals_decl if "in" in d[1]}
        return self.fs.run(uniforms, {k: v for k, v in inputs.items() if k in fs_in})


def vec(*xs):
    return Vec('f', [F(x) for x in xs])


def mat4(column_major_16):
    m = [F(x) for x in column_major_16]
    return Mat(4, [Vec('f', m[4 * j:4 * j + 4]) for j in range(4)])
