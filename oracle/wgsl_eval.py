"""oracle/wgsl_eval.py -- a small WGSL evaluator (TEST INFRASTRUCTURE ONLY; never imported by the product package).

Why it exists: the reference's develop path IS a WGSL shader (the string constant of /root/reference/src/gpu/shaders.rs:14-267)
and nothing in this environment can run WGSL (no wgpu, no naga).  oracle/develop_ref.c and oracle/develop_np.py are hand
restatements of that text.  This module instead EXECUTES the text: tools/make_wgsl_golden.py reads the shader string where it
lies under /root/reference, runs its `vs_main` / `fs_main` through this evaluator and commits inputs + outputs as
tests/golden/wgsl_golden.npz; the oracle (and the HIP path) must reproduce those vectors bit for bit.  No reference text is
stored in this repository: the evaluator is a general interpreter for the WGSL subset below, written from the WGSL
specification, and its unit tests (tests/test_wgsl_pin_cpu.py) use shader snippets written for them.

What the WGSL specification leaves to the implementation is NOT decided in here; it is passed in as a `Lowering`:
    pow(x, y)                        (accuracy is implementation-defined: "inherited from exp2(y * log2(x))")
    contraction                      (an implementation may fuse a multiplication into the addition or subtraction that
                                      consumes it; "none" keeps every rounding, "fuse" models what an LLVM-style compiler
                                      does with the text: see Lowering)
    dot / matrix * vector            (summation order)
    mix(x, y, a)                     (x * (1 - a) + y * a  or  x + (y - x) * a; the specification allows both)
    min / max / clamp on NaN         (implementation-defined)
    division                         (2.5 ulp are granted: the correctly rounded quotient, or x * (1 / y))
    f32 -> i32 of NaN, textureLoad out of bounds
Everything else follows the specification: literals without a suffix are AbstractInt / AbstractFloat (64-bit), constant
sub-expressions are evaluated in that type and converted once when they meet a concrete operand, there are no implicit
conversions between concrete types (a shader that needs one is rejected), every f32 operation rounds to nearest even
(IEEE-754 binary32, no contraction), i32 / u32 arithmetic wraps, i32 division truncates, f32 -> i32 truncates and saturates,
`mat3x3(a, b, c)` takes COLUMNS.

Subset: struct / module-scope var / fn declarations with attributes; let / var / assignment (with swizzle or member on the
left, compound forms) / if-else / return; the operators || && | ^ & == != < > <= >= << >> + - * / % and unary - !;
scalar, vecN<T> and mat3x3<f32> constructors; swizzles; builtins textureLoad, textureDimensions, clamp, min, max, pow, dot,
mix, abs, floor, select.  Anything else raises WgslError (so a shader that steps outside the subset is noticed, not
half-evaluated).
"""
from __future__ import annotations

import math
import re
from fractions import Fraction

import numpy as np

F32 = np.float32


class WgslError(Exception):
    pass


# ---------------------------------------------------------------------------------------------------------------------------
# values
# ---------------------------------------------------------------------------------------------------------------------------
# scalar kinds: "ai" AbstractInt (python int), "af" AbstractFloat (python float = binary64), "f32" (numpy.float32),
# "i32" / "u32" (python int, kept in range), "bool"
class Sc:
    __slots__ = ("k", "v")

    def __init__(self, k, v):
        self.k, self.v = k, v

    def __repr__(self):
        return f"{self.k}({self.v!r})"


class Vec:
    """Immutable vector of n scalars of one kind."""
    __slots__ = ("k", "c")

    def __init__(self, k, comps):
        self.k, self.c = k, tuple(comps)

    def __len__(self):
        return len(self.c)

    def __repr__(self):
        return f"vec{len(self.c)}<{self.k}>{self.c!r}"


class Mat:
    """Column-major matrix: a tuple of column Vecs (WGSL matCxR constructors take columns)."""
    __slots__ = ("cols",)

    def __init__(self, cols):
        self.cols = tuple(cols)


class Struct:
    __slots__ = ("name", "f")

    def __init__(self, name, fields):
        self.name, self.f = name, dict(fields)


class Texture2D:
    """texture_2d<u32> with one mip level; `data` is a (H, W) integer array, texel = (data[y, x], 0, 0, 1)."""

    def __init__(self, data):
        self.data = np.asarray(data)
        if self.data.ndim != 2:
            raise WgslError("texture data must be 2-D")
        self.oob_loads = 0


class Type:
    __slots__ = ("name", "args")

    def __init__(self, name, args=()):
        self.name, self.args = name, tuple(args)

    def __repr__(self):
        return self.name + (f"<{', '.join(map(repr, self.args))}>" if self.args else "")


_I32_MIN, _I32_MAX, _U32_MOD = -(1 << 31), (1 << 31) - 1, 1 << 32


def _wrap_i32(v):
    v &= _U32_MOD - 1
    return v - _U32_MOD if v > _I32_MAX else v


def _wrap_u32(v):
    return v & (_U32_MOD - 1)


def _convert(k, v, to):
    """The AUTOMATIC conversions of the specification (abstract -> concrete, AbstractInt -> AbstractFloat) only."""
    if k == to:
        return v
    if k == "ai":
        if to == "af":
            return float(v)
        if to == "f32":
            return F32(v)
        if to == "i32":
            if not _I32_MIN <= v <= _I32_MAX:
                raise WgslError(f"AbstractInt {v} does not fit i32")
            return v
        if to == "u32":
            if not 0 <= v < _U32_MOD:
                raise WgslError(f"AbstractInt {v} does not fit u32")
            return v
    if k == "af" and to == "f32":
        with np.errstate(over="ignore"):
            r = F32(v)                                       # one rounding, to nearest even
        if np.isinf(r) and not math.isinf(v):
            raise WgslError(f"AbstractFloat {v} does not fit f32")
        return r
    raise WgslError(f"no automatic conversion from {k} to {to}")


_RANK = {"ai": 0, "af": 1}


def _unify(ka, kb):
    if ka == kb:
        return ka
    if ka in _RANK and kb in _RANK:
        return "af"
    if ka in _RANK:
        ka, kb = kb, ka
    if kb == "ai" and ka in ("f32", "i32", "u32"):
        return ka
    if kb == "af" and ka == "f32":
        return ka
    raise WgslError(f"operands of types {ka} and {kb}: WGSL has no implicit conversion between them")


# ---------------------------------------------------------------------------------------------------------------------------
# what the specification leaves open
# ---------------------------------------------------------------------------------------------------------------------------
def _fmax(a, b):
    if a != a:
        return b
    if b != b:
        return a
    return a if a > b else b


def _fmin(a, b):
    if a != a:
        return b
    if b != b:
        return a
    return a if a < b else b


class Product(np.float32):
    """An f32 that remembers it is round(x * y): under contraction="fuse" an addition or subtraction that consumes it computes
    fma(x, y, other) instead.  Any other use sees a plain f32 (numpy arithmetic on it returns numpy.float32)."""


def _product(x, y):
    with np.errstate(all="ignore"):
        p = Product(x * y)
    p.factors = (x, y)
    return p


def fma_f32(x, y, z):
    """round(x * y + z) with ONE rounding to nearest even (exact rational arithmetic; NaN / infinity by the float rules)."""
    x, y, z = float(x), float(y), float(z)
    if not (math.isfinite(x) and math.isfinite(y) and math.isfinite(z)):
        with np.errstate(all="ignore"):
            return F32(np.float64(x) * np.float64(y) + np.float64(z))
    exact = Fraction(x) * Fraction(y) + Fraction(z)
    if exact == 0:
        with np.errstate(all="ignore"):
            return F32(np.float64(x) * np.float64(y) + np.float64(z))       # the sign of an exact zero follows the float rules
    mag = abs(exact)
    e = mag.numerator.bit_length() - mag.denominator.bit_length()
    if Fraction(2) ** e > mag:
        e -= 1
    e = max(e, -126)                                         # below 2^-126 the quantum stays 2^-149 (subnormals)
    quantum = Fraction(2) ** (e - 23)
    n = mag / quantum
    m = n.numerator // n.denominator
    rem = n - m
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (m & 1)):
        m += 1
    value = m * quantum
    if value >= Fraction(2) ** 128:
        return F32(-np.inf if exact < 0 else np.inf)
    r = float(value)                                         # exact: at most 25 significant bits
    return F32(-r if exact < 0 else r)


class Lowering:
    """Implementation-defined behaviour, made explicit.  `pow` maps two numpy.float32 to one; the rest are named choices.

    contraction = "none": every multiplication and addition rounds on its own.
    contraction = "fuse": an f32 addition or subtraction one of whose operands is the (not otherwise transformed) result of
    an f32 multiplication becomes one fma -- the left operand's product first, then the right one's, as LLVM's DAG combiner
    orders them (fadd (fmul x, y), z -> fma x, y, z;  fadd x, (fmul y, z) -> fma y, z, x;  fsub (fmul x, y), z ->
    fma x, y, -z;  fsub x, (fmul y, z) -> fma -y, z, x), through `let` bindings, function calls, vector components and unary
    minus, whatever the number of uses (the aggressive form).  dot and mix are built from the same mul / add / sub."""

    def __init__(self, pow, dot_order="left_to_right", mix_form="x*(1-a)+y*a", nan_minmax="other_operand",
                 nan_to_int=0, texture_oob="clamp", contraction="none", division="ieee"):
        if contraction not in ("none", "fuse"):
            raise WgslError("unknown contraction rule")
        if division not in ("ieee", "reciprocal"):
            raise WgslError("unknown division rule")
        self.fuse = contraction == "fuse"
        self.reciprocal = division == "reciprocal"
        if dot_order != "left_to_right":
            raise WgslError("only left-to-right dot products are implemented")
        if mix_form not in ("x*(1-a)+y*a", "x+(y-x)*a"):
            raise WgslError("unknown mix form")
        if nan_minmax != "other_operand":
            raise WgslError("only IEEE minNum/maxNum (a NaN operand yields the other one) is implemented")
        if texture_oob not in ("clamp", "zero", "error"):
            raise WgslError("unknown texture_oob rule")
        self.pow, self.mix_form, self.nan_to_int, self.texture_oob = pow, mix_form, nan_to_int, texture_oob

    # f32 scalars in, f32 scalar out
    def mul(self, a, b):
        if self.fuse:
            return _product(a, b)
        with np.errstate(all="ignore"):
            return a * b

    def add(self, a, b):
        if self.fuse:
            if isinstance(a, Product):
                return fma_f32(*a.factors, b)
            if isinstance(b, Product):
                return fma_f32(*b.factors, a)
        with np.errstate(all="ignore"):
            return F32(a) + F32(b)

    def sub(self, a, b):
        if self.fuse:
            if isinstance(a, Product):
                return fma_f32(*a.factors, -F32(b))
            if isinstance(b, Product):
                return fma_f32(-b.factors[0], b.factors[1], a)
        with np.errstate(all="ignore"):
            return F32(a) - F32(b)

    def div(self, a, b):
        """division = "ieee": the correctly rounded quotient.  "reciprocal": x * round(1 / y), two roundings -- inside the 2.5 ulp
        the specification grants f32 division, and what a compiler does that turns a division into a multiplication."""
        with np.errstate(all="ignore"):
            if self.reciprocal:
                return self.mul(F32(a), F32(1.0) / F32(b))
            return F32(a) / F32(b)

    def neg(self, a):
        if self.fuse and isinstance(a, Product):
            return _product(-a.factors[0], a.factors[1])
        return -F32(a)

    def dot(self, a, b):
        acc = self.mul(a[0], b[0])
        for x, y in zip(a[1:], b[1:]):
            acc = self.add(acc, self.mul(x, y))
        return F32(acc) if not self.fuse else acc

    def mix(self, x, y, a):
        if self.mix_form == "x*(1-a)+y*a":
            return self.add(self.mul(x, self.sub(F32(1.0), a)), self.mul(y, a))
        return self.add(x, self.mul(self.sub(y, x), a))


def pow_f64_rounded(x, y):
    """pow in binary64 (the C library's), rounded once to f32; a negative base gives NaN as exp2(y * log2(x)) does."""
    x, y = float(x), float(y)
    if x != x or y != y or x < 0.0:
        return F32(np.nan)
    if x == 0.0:
        return F32(0.0) if y > 0.0 else (F32(1.0) if y == 0.0 else F32(np.inf))
    try:
        r = math.pow(x, y)
    except OverflowError:
        r = math.inf
    with np.errstate(over="ignore"):
        return F32(r)


# ---------------------------------------------------------------------------------------------------------------------------
# lexer
# ---------------------------------------------------------------------------------------------------------------------------
_TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>0[xX][0-9a-fA-F]+[iu]?
          |(?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?[fh]?
          |\d+[eE][+-]?\d+[fh]?
          |\d+[iufh]?)
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<op>->|>>=|<<=|>>|<<|<=|>=|==|!=|&&|\|\||\+=|-=|\*=|/=|%=|&=|\|=|\^=|\+\+|--|[-+*/%&|^!~<>=(){}\[\],;:.@])
""", re.VERBOSE | re.DOTALL)


def _lex(src):
    out, pos = [], 0
    while pos < len(src):
        m = _TOKEN.match(src, pos)
        if not m:
            raise WgslError(f"cannot tokenise at {src[pos:pos + 30]!r}")
        pos = m.end()
        if m.lastgroup != "ws":
            out.append((m.lastgroup, m.group(m.lastgroup)))
    out.append(("eof", ""))
    return out


_TEMPLATED = {"vec2", "vec3", "vec4", "mat2x2", "mat3x3", "mat4x4", "array", "texture_2d", "ptr", "atomic"}
_SCALARS = {"f32", "i32", "u32", "bool"}
_BINARY_LEVELS = [("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", ">", "<=", ">="), ("<<", ">>"),
                  ("+", "-"), ("*", "/", "%")]


# ---------------------------------------------------------------------------------------------------------------------------
# parser (recursive descent; the AST is nested tuples headed by a tag string)
# ---------------------------------------------------------------------------------------------------------------------------
class _Parser:
    def __init__(self, src):
        self.t, self.i = _lex(src), 0

    def peek(self, k=0):
        return self.t[self.i + k]

    def at(self, val):
        return self.t[self.i][1] == val and self.t[self.i][0] in ("op", "id")

    def take(self, val=None):
        tok = self.t[self.i]
        if val is not None and tok[1] != val:
            raise WgslError(f"expected {val!r}, found {tok[1]!r} (token {self.i})")
        self.i += 1
        return tok

    def ident(self):
        tok = self.take()
        if tok[0] != "id":
            raise WgslError(f"expected an identifier, found {tok[1]!r}")
        return tok[1]

    def attributes(self):
        attrs = []
        while self.at("@"):
            self.take("@")
            name, args = self.ident(), []
            if self.at("("):
                self.take("(")
                while not self.at(")"):
                    args.append(self.take()[1])
                    if self.at(","):
                        self.take(",")
                self.take(")")
            attrs.append((name, tuple(args)))
        return attrs

    def type_(self):
        name, args = self.ident(), []
        if self.at("<"):
            self.take("<")
            while True:
                args.append(self.type_())
                if self.at(","):
                    self.take(",")
                    continue
                break
            self.take(">")
        return Type(name, args)

    def module(self):
        structs, globals_, fns = {}, {}, {}
        while self.peek()[0] != "eof":
            attrs = self.attributes()
            if self.at("struct"):
                self.take()
                name, fields = self.ident(), []
                self.take("{")
                while not self.at("}"):
                    self.attributes()
                    fname = self.ident()
                    self.take(":")
                    fields.append((fname, self.type_()))
                    if self.at(","):
                        self.take(",")
                self.take("}")
                if self.at(";"):
                    self.take(";")
                structs[name] = fields
            elif self.at("var"):
                self.take()
                space = None
                if self.at("<"):
                    self.take("<")
                    space = self.ident()
                    while not self.at(">"):
                        self.take()
                    self.take(">")
                name = self.ident()
                self.take(":")
                ty = self.type_()
                self.take(";")
                globals_[name] = (space, ty, attrs)
            elif self.at("fn"):
                self.take()
                name, params = self.ident(), []
                self.take("(")
                while not self.at(")"):
                    self.attributes()
                    pname = self.ident()
                    self.take(":")
                    params.append((pname, self.type_()))
                    if self.at(","):
                        self.take(",")
                self.take(")")
                ret = None
                if self.at("->"):
                    self.take("->")
                    self.attributes()
                    ret = self.type_()
                fns[name] = (params, ret, self.block(), attrs)
            else:
                raise WgslError(f"unsupported module-scope declaration at {self.peek()[1]!r}")
        return structs, globals_, fns

    def block(self):
        self.take("{")
        body = []
        while not self.at("}"):
            body.append(self.statement())
        self.take("}")
        return body

    def statement(self):
        if self.at("{"):
            return ("block", self.block())
        if self.at("var") or self.at("let"):
            kw = self.take()[1]
            name, ty, init = self.ident(), None, None
            if self.at(":"):
                self.take(":")
                ty = self.type_()
            if self.at("="):
                self.take("=")
                init = self.expr()
            elif kw == "let":
                raise WgslError("let without an initialiser")
            self.take(";")
            return (kw, name, ty, init)
        if self.at("if"):
            return self.if_()
        if self.at("return"):
            self.take()
            e = None if self.at(";") else self.expr()
            self.take(";")
            return ("return", e)
        for kw in ("for", "loop", "while", "switch", "break", "continue", "discard", "const"):
            if self.at(kw):
                raise WgslError(f"statement `{kw}` is outside the supported subset")
        lhs = self.unary()
        tok = self.peek()
        if tok[1] in ("=", "+=", "-=", "*=", "/=", "%="):
            self.take()
            rhs = self.expr()
            self.take(";")
            return ("assign", lhs, tok[1], rhs)
        if lhs[0] == "call":
            self.take(";")
            return ("expr", lhs)
        raise WgslError(f"unsupported statement at {tok[1]!r}")

    def if_(self):
        self.take("if")
        cond = self.expr()
        then, other = self.block(), None
        if self.at("else"):
            self.take()
            other = [self.if_()] if self.at("if") else self.block()
        return ("if", cond, then, other)

    def expr(self, level=0):
        if level == len(_BINARY_LEVELS):
            return self.unary()
        lhs = self.expr(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in _BINARY_LEVELS[level]:
            op = self.take()[1]
            lhs = ("bin", op, lhs, self.expr(level + 1))
        return lhs

    def unary(self):
        if self.peek()[0] == "op" and self.peek()[1] in ("-", "!"):
            op = self.take()[1]
            return ("un", op, self.unary())
        for op in ("~", "&", "*"):
            if self.peek() == ("op", op):
                raise WgslError(f"unary `{op}` is outside the supported subset")
        return self.postfix(self.primary())

    def postfix(self, e):
        while True:
            if self.at("."):
                self.take()
                e = ("member", e, self.ident())
            elif self.at("["):
                self.take()
                idx = self.expr()
                self.take("]")
                e = ("index", e, idx)
            else:
                return e

    def primary(self):
        kind, text = self.peek()
        if kind == "num":
            self.take()
            return ("lit", _literal(text))
        if text == "(" and kind == "op":
            self.take()
            e = self.expr()
            self.take(")")
            return ("paren", e)
        if kind == "id":
            if text in ("true", "false"):
                self.take()
                return ("lit", Sc("bool", text == "true"))
            if text in _TEMPLATED and self.peek(1) == ("op", "<"):
                ty = self.type_()
                return ("construct", ty, self.args())
            name = self.ident()
            if self.at("("):
                if name in _SCALARS:
                    return ("construct", Type(name), self.args())
                return ("call", name, self.args())
            return ("var", name)
        raise WgslError(f"unexpected token {text!r} in an expression")

    def args(self):
        self.take("(")
        out = []
        while not self.at(")"):
            out.append(self.expr())
            if self.at(","):
                self.take(",")
        self.take(")")
        return out


def _literal(text):
    if text[:2] in ("0x", "0X"):
        suffix = text[-1] if text[-1] in "iu" else ""
        v = int(text[2:len(text) - len(suffix)], 16)
        return Sc({"": "ai", "i": "i32", "u": "u32"}[suffix], v)
    if text[-1] == "h":
        raise WgslError("f16 literals are outside the supported subset")
    is_float = any(ch in text for ch in ".eE") or text[-1] == "f"
    if not is_float:
        suffix = text[-1] if text[-1] in "iu" else ""
        v = int(text[:len(text) - len(suffix)])
        k = {"": "ai", "i": "i32", "u": "u32"}[suffix]
        return Sc(k, _convert("ai", v, k) if k != "ai" else v)
    if text[-1] == "f":
        return Sc("f32", _convert("af", float(text[:-1]), "f32"))
    return Sc("af", float(text))                             # python parses decimal text to the nearest binary64


# ---------------------------------------------------------------------------------------------------------------------------
# evaluator
# ---------------------------------------------------------------------------------------------------------------------------
class _Return(Exception):
    def __init__(self, value):
        self.value = value


_SWIZZLE = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}


def _scalar_binop(op, k, a, b, low=None):
    """a, b already of kind k.  Returns (kind, value).  `low`: the Lowering (f32 mul / add / sub go through it)."""
    if op in ("==", "!=", "<", ">", "<=", ">="):
        r = {"==": a == b, "!=": a != b, "<": a < b, ">": a > b, "<=": a <= b, ">=": a >= b}[op]
        return "bool", bool(r)
    if k == "bool":
        if op in ("&", "&&"):
            return "bool", a and b
        if op in ("|", "||"):
            return "bool", a or b
        raise WgslError(f"operator {op} on bool")
    if k == "f32":
        if low is not None and op in ("+", "-", "*", "/"):
            return k, {"+": low.add, "-": low.sub, "*": low.mul, "/": low.div}[op](a, b)
        with np.errstate(all="ignore"):
            if op == "+":
                return k, a + b
            if op == "-":
                return k, a - b
            if op == "*":
                return k, a * b
            if op == "/":
                return k, a / b                               # numpy.float32 / numpy.float32: IEEE binary32 division
            if op == "%":
                return k, F32(np.fmod(a, b))                  # truncated, like the specification's x - y * trunc(x / y)
        raise WgslError(f"operator {op} on f32")
    if k == "af":
        try:
            if op == "+":
                return k, a + b
            if op == "-":
                return k, a - b
            if op == "*":
                return k, a * b
            if op == "/":
                return k, a / b
            if op == "%":
                return k, math.fmod(a, b)
        except ZeroDivisionError:
            raise WgslError("constant expression divides by zero")
        raise WgslError(f"operator {op} on AbstractFloat")
    # integers
    if op == "+":
        r = a + b
    elif op == "-":
        r = a - b
    elif op == "*":
        r = a * b
    elif op in ("/", "%"):
        if b == 0 or (k == "i32" and a == _I32_MIN and b == -1):
            if k == "ai":
                raise WgslError("constant expression divides by zero")
            r = a if op == "/" else 0                        # the specification's defined results
        else:
            q = abs(a) // abs(b)
            q = q if (a < 0) == (b < 0) else -q              # truncation toward zero
            r = q if op == "/" else a - q * b
    elif op == "&":
        r = a & b
    elif op == "|":
        r = a | b
    elif op == "^":
        r = a ^ b
    elif op in ("<<", ">>"):
        raise WgslError("internal: shifts are handled by the caller")
    else:
        raise WgslError(f"operator {op} on {k}")
    if k == "i32":
        r = _wrap_i32(r)
    elif k == "u32":
        r = _wrap_u32(r)
    return k, r


class Module:
    """A parsed shader module bound to resources.  `bind(name, value)` gives a module-scope variable its value
    (a Texture2D, or a dict / Struct for a uniform buffer); `call(fn, *args)` runs a function."""

    def __init__(self, source, lowering):
        self.structs, self.globals, self.fns = _Parser(source).module()
        self.low = lowering
        self.bound = {}
        self.nan_to_int = 0          # how often an f32 -> i32 conversion met a NaN (the Lowering decided the result)

    # ---- resources -------------------------------------------------------------------------------------------------------
    def bind(self, name, value):
        if name not in self.globals:
            raise WgslError(f"the module declares no variable `{name}`")
        _, ty, _ = self.globals[name]
        if isinstance(value, dict):
            value = self.make_struct(ty.name, value)
        self.bound[name] = value

    def make_struct(self, name, values):
        """Fill struct `name` from a dict; members missing from the dict are zero (padding).  Python floats / sequences are
        converted to the member's declared type."""
        if name not in self.structs:
            raise WgslError(f"unknown struct {name}")
        fields, values = {}, dict(values)
        for fname, ty in self.structs[name]:
            fields[fname] = self._from_host(ty, values.pop(fname)) if fname in values else self.zero(ty)
        if values:
            raise WgslError(f"struct {name} has no member(s) {sorted(values)}")
        return Struct(name, fields)

    def _from_host(self, ty, v):
        if isinstance(v, (Sc, Vec, Mat, Struct)):
            return v
        if ty.name in _SCALARS:
            return Sc(ty.name, self._host_scalar(ty.name, v))
        if ty.name in ("vec2", "vec3", "vec4"):
            k, n = ty.args[0].name, int(ty.name[3])
            v = list(v)
            if len(v) != n:
                raise WgslError(f"{ty!r} needs {n} components")
            return Vec(k, [self._host_scalar(k, x) for x in v])
        if ty.name in self.structs:
            return self.make_struct(ty.name, v)
        raise WgslError(f"cannot build a {ty!r} from host data")

    @staticmethod
    def _host_scalar(k, v):
        if k == "f32":
            return F32(v)
        if k == "bool":
            return bool(v)
        v = int(v)
        if (k == "i32" and not _I32_MIN <= v <= _I32_MAX) or (k == "u32" and not 0 <= v < _U32_MOD):
            raise WgslError(f"{v} does not fit {k}")
        return v

    def zero(self, ty):
        if ty.name in _SCALARS:
            return Sc(ty.name, {"f32": F32(0.0), "i32": 0, "u32": 0, "bool": False}[ty.name])
        if ty.name in ("vec2", "vec3", "vec4"):
            z = self.zero(ty.args[0])
            return Vec(z.k, [z.v] * int(ty.name[3]))
        if ty.name == "mat3x3":
            col = Vec("f32", [F32(0.0)] * 3)
            return Mat([col, col, col])
        if ty.name in self.structs:
            return Struct(ty.name, {f: self.zero(t) for f, t in self.structs[ty.name]})
        raise WgslError(f"no zero value for {ty!r}")

    # ---- calls -----------------------------------------------------------------------------------------------------------
    def call(self, name, *args):
        if name not in self.fns:
            raise WgslError(f"unknown function {name}")
        params, ret, body, _ = self.fns[name]
        if len(params) != len(args):
            raise WgslError(f"{name} takes {len(params)} argument(s)")
        scope = [{}]
        for (pname, ty), a in zip(params, args):
            scope[0][pname] = ("let", self.coerce(a, ty))
        try:
            self.run_block(body, scope)
        except _Return as r:
            if ret is None:
                if r.value is not None:
                    raise WgslError(f"{name} returns a value but declares none")
                return None
            return self.coerce(r.value, ret)
        if ret is not None:
            raise WgslError(f"{name} ended without a return")
        return None

    def coerce(self, v, ty):
        """Value `v` where a `ty` is required: abstract values become concrete; concrete ones must already match."""
        if ty.name in _SCALARS:
            if not isinstance(v, Sc):
                raise WgslError(f"expected {ty!r}")
            return Sc(ty.name, _convert(v.k, v.v, ty.name))
        if ty.name in ("vec2", "vec3", "vec4"):
            k = ty.args[0].name
            if not isinstance(v, Vec) or len(v) != int(ty.name[3]):
                raise WgslError(f"expected {ty!r}, got {v!r}")
            return Vec(k, [_convert(v.k, x, k) for x in v.c])
        if ty.name == "mat3x3":
            if not isinstance(v, Mat):
                raise WgslError(f"expected {ty!r}")
            return v
        if ty.name in self.structs:
            if not isinstance(v, Struct) or v.name != ty.name:
                raise WgslError(f"expected {ty!r}")
            return v
        raise WgslError(f"unsupported type {ty!r}")

    # ---- statements ------------------------------------------------------------------------------------------------------
    def run_block(self, body, scope):
        scope.append({})
        try:
            for st in body:
                self.run(st, scope)
        finally:
            scope.pop()

    def lookup(self, name, scope):
        for frame in reversed(scope):
            if name in frame:
                return frame[name]
        if name in self.bound:
            return ("let", self.bound[name])
        if name in self.globals:
            raise WgslError(f"module-scope variable `{name}` has nothing bound to it")
        raise WgslError(f"unknown identifier `{name}`")

    def run(self, st, scope):
        tag = st[0]
        if tag in ("let", "var"):
            _, name, ty, init = st
            if init is None:
                v = self.zero(ty)
            else:
                v = self.ev(init, scope)
                if ty is not None:
                    v = self.coerce(v, ty)
                else:
                    v = self.concretise(v)
            if name in scope[-1]:
                raise WgslError(f"`{name}` redeclared in the same scope")
            scope[-1][name] = (tag, v)
        elif tag == "assign":
            _, lhs, op, rhs = st
            path = []
            while lhs[0] == "member":
                path.append(lhs[2])
                lhs = lhs[1]
            if lhs[0] != "var":
                raise WgslError("unsupported left-hand side")
            path.reverse()
            for frame in reversed(scope):
                if lhs[1] in frame:
                    break
            else:
                raise WgslError(f"assignment to unknown or module-scope variable `{lhs[1]}`")
            kind, root = frame[lhs[1]]
            if kind != "var":
                raise WgslError(f"assignment to `{lhs[1]}`, which is not a `var`")
            value = self.ev(rhs, scope)
            if op != "=":
                value = self.binop(op[:-1], self.read_path(root, path), value)
            frame[lhs[1]] = ("var", self.write_path(root, path, value))
        elif tag == "if":
            _, cond, then, other = st
            c = self.ev(cond, scope)
            if not (isinstance(c, Sc) and c.k == "bool"):
                raise WgslError("an if condition must be a bool")
            if c.v:
                self.run_block(then, scope)
            elif other is not None:
                self.run_block(other, scope)
        elif tag == "return":
            raise _Return(None if st[1] is None else self.ev(st[1], scope))
        elif tag == "block":
            self.run_block(st[1], scope)
        elif tag == "expr":
            self.ev(st[1], scope)
        else:
            raise WgslError(f"internal: statement {tag}")

    def concretise(self, v):
        """`let x = <abstract>` gives x the default concrete type (i32 / f32)."""
        if isinstance(v, Sc) and v.k in _RANK:
            to = "i32" if v.k == "ai" else "f32"
            return Sc(to, _convert(v.k, v.v, to))
        if isinstance(v, Vec) and v.k in _RANK:
            to = "i32" if v.k == "ai" else "f32"
            return Vec(to, [_convert(v.k, x, to) for x in v.c])
        return v

    def read_path(self, v, path):
        for name in path:
            v = self.member(v, name)
        return v

    def write_path(self, root, path, value):
        if not path:
            if isinstance(root, Sc):
                return Sc(root.k, _convert(value.k, value.v, root.k)) if isinstance(value, Sc) else self._bad_store(root, value)
            if isinstance(root, Vec):
                if not isinstance(value, Vec) or len(value) != len(root):
                    self._bad_store(root, value)
                return Vec(root.k, [_convert(value.k, x, root.k) for x in value.c])
            if isinstance(root, Struct):
                if not isinstance(value, Struct) or value.name != root.name:
                    self._bad_store(root, value)
                return value
            if isinstance(root, Mat) and isinstance(value, Mat):
                return value
            self._bad_store(root, value)
        name, rest = path[0], path[1:]
        if isinstance(root, Struct):
            if name not in root.f:
                raise WgslError(f"struct {root.name} has no member {name}")
            f = dict(root.f)
            f[name] = self.write_path(root.f[name], rest, value)
            return Struct(root.name, f)
        if isinstance(root, Vec):
            if len(name) != 1 or rest:
                raise WgslError("only single-component swizzles can be assigned")       # as in the specification
            idx = _SWIZZLE.get(name)
            if idx is None or idx >= len(root):
                raise WgslError(f"bad component {name}")
            if not isinstance(value, Sc):
                self._bad_store(root, value)
            c = list(root.c)
            c[idx] = _convert(value.k, value.v, root.k)
            return Vec(root.k, c)
        raise WgslError(f"cannot assign through .{name}")

    @staticmethod
    def _bad_store(root, value):
        raise WgslError(f"cannot store {value!r} into {root!r}")

    # ---- expressions -----------------------------------------------------------------------------------------------------
    def member(self, v, name):
        if isinstance(v, Struct):
            if name not in v.f:
                raise WgslError(f"struct {v.name} has no member {name}")
            return v.f[name]
        if isinstance(v, Vec):
            sets = ("xyzw", "rgba")
            if not any(all(ch in s for ch in name) for s in sets) or not 1 <= len(name) <= 4:
                raise WgslError(f"bad swizzle .{name}")
            idx = [_SWIZZLE[ch] for ch in name]
            if max(idx) >= len(v):
                raise WgslError(f"swizzle .{name} on a {len(v)}-component vector")
            return Sc(v.k, v.c[idx[0]]) if len(idx) == 1 else Vec(v.k, [v.c[i] for i in idx])
        raise WgslError(f"member .{name} of a value that has none")

    def ev(self, e, scope):
        tag = e[0]
        if tag == "lit":
            return e[1]
        if tag == "paren":
            return self.ev(e[1], scope)
        if tag == "var":
            return self.lookup(e[1], scope)[1]
        if tag == "member":
            return self.member(self.ev(e[1], scope), e[2])
        if tag == "un":
            v = self.ev(e[2], scope)
            if e[1] == "!":
                if isinstance(v, Sc) and v.k == "bool":
                    return Sc("bool", not v.v)
                raise WgslError("! needs a bool")
            return self.map1(v, self.neg)
        if tag == "bin":
            op = e[1]
            if op in ("||", "&&"):
                a = self.ev(e[2], scope)
                if not (isinstance(a, Sc) and a.k == "bool"):
                    raise WgslError(f"{op} needs bool operands")
                if (op == "||") == a.v:
                    return a                                  # short circuit
                b = self.ev(e[3], scope)
                if not (isinstance(b, Sc) and b.k == "bool"):
                    raise WgslError(f"{op} needs bool operands")
                return b
            return self.binop(op, self.ev(e[2], scope), self.ev(e[3], scope))
        if tag == "construct":
            return self.construct(e[1], [self.ev(a, scope) for a in e[2]])
        if tag == "call":
            args = [self.ev(a, scope) for a in e[2]]
            if e[1] in self.fns:
                return self.call(e[1], *args)
            return self.builtin(e[1], args)
        if tag == "index":
            v, i = self.ev(e[1], scope), self.ev(e[2], scope)
            if isinstance(v, Vec) and isinstance(i, Sc) and i.k in ("ai", "i32", "u32") and 0 <= i.v < len(v):
                return Sc(v.k, v.c[i.v])
            if isinstance(v, Mat) and isinstance(i, Sc) and i.k in ("ai", "i32", "u32") and 0 <= i.v < len(v.cols):
                return v.cols[i.v]
            raise WgslError("unsupported or out-of-range index")
        raise WgslError(f"internal: expression {tag}")

    def neg(self, k, v):
        if k == "f32":
            return k, self.low.neg(v)
        if k == "af":
            return k, -v
        if k == "ai":
            return k, -v
        if k == "i32":
            return k, _wrap_i32(-v)
        raise WgslError(f"unary minus on {k}")

    @staticmethod
    def map1(v, fn):
        if isinstance(v, Sc):
            return Sc(*fn(v.k, v.v))
        if isinstance(v, Vec):
            out = [fn(v.k, x) for x in v.c]
            return Vec(out[0][0], [o[1] for o in out])
        raise WgslError("operand must be a scalar or a vector")

    def binop(self, op, a, b):
        if isinstance(a, Mat) or isinstance(b, Mat):
            if op == "*" and isinstance(a, Mat) and isinstance(b, Vec) and len(b) == len(a.cols) and b.k == "f32":
                # component i = dot(row i of the matrix, the vector), summed left to right (Lowering.dot)
                n = len(a.cols[0])
                return Vec("f32", [self.low.dot([col.c[i] for col in a.cols], list(b.c)) for i in range(n)])
            raise WgslError("only mat3x3<f32> * vec3<f32> is supported")
        if op in ("<<", ">>"):
            if not (isinstance(a, Sc) and isinstance(b, Sc)) or a.k not in ("i32", "u32", "ai") or b.k not in ("u32", "ai"):
                raise WgslError("shift needs integer << / >> u32 scalars")
            if a.k == "ai":
                raise WgslError("shift of an AbstractInt is outside the supported subset")
            n = b.v & 31
            if op == "<<":
                r = a.v << n
            else:
                r = a.v >> n                                  # python ints: arithmetic for negative i32, logical for u32
            return Sc(a.k, _wrap_i32(r) if a.k == "i32" else _wrap_u32(r))
        ka, kb = a.k, b.k
        k = _unify(ka, kb)
        if isinstance(a, Sc) and isinstance(b, Sc):
            return Sc(*_scalar_binop(op, k, _convert(ka, a.v, k), _convert(kb, b.v, k), self.low))
        if op in ("==", "!=", "<", ">", "<=", ">=", "||", "&&"):
            raise WgslError("vector comparisons are outside the supported subset")
        n = len(a) if isinstance(a, Vec) else len(b)
        ac = a.c if isinstance(a, Vec) else (a.v,) * n
        bc = b.c if isinstance(b, Vec) else (b.v,) * n
        if len(ac) != len(bc):
            raise WgslError("vector operands of different sizes")
        out = [_scalar_binop(op, k, _convert(ka, x, k), _convert(kb, y, k), self.low) for x, y in zip(ac, bc)]
        return Vec(out[0][0], [o[1] for o in out])

    def construct(self, ty, args):
        if ty.name in _SCALARS:
            if len(args) != 1 or not isinstance(args[0], Sc):
                raise WgslError(f"{ty!r}() takes one scalar")
            return Sc(ty.name, self.value_convert(args[0].k, args[0].v, ty.name))
        if ty.name in ("vec2", "vec3", "vec4"):
            n, k = int(ty.name[3]), ty.args[0].name
            comps = []
            for a in args:
                if isinstance(a, Sc):
                    comps.append((a.k, a.v))
                elif isinstance(a, Vec):
                    comps.extend((a.k, x) for x in a.c)
                else:
                    raise WgslError(f"bad argument to {ty!r}()")
            if len(args) == 1 and isinstance(args[0], Sc):
                comps = comps * n                             # splat
            if len(comps) != n:
                raise WgslError(f"{ty!r}() got {len(comps)} components")
            # components of a vector constructor need the element type (abstract ones convert automatically)
            return Vec(k, [_convert(ck, cv, k) for ck, cv in comps])
        if ty.name == "mat3x3":
            if len(args) != 3 or not all(isinstance(a, Vec) and len(a) == 3 for a in args):
                raise WgslError("mat3x3() is supported with three column vectors only")
            return Mat([Vec("f32", [_convert(a.k, x, "f32") for x in a.c]) for a in args])
        if ty.name in self.structs:
            fields = self.structs[ty.name]
            if len(args) != len(fields):
                raise WgslError(f"{ty.name}() takes {len(fields)} members")
            return Struct(ty.name, {f: self.coerce(a, t) for (f, t), a in zip(fields, args)})
        raise WgslError(f"constructor {ty!r} is outside the supported subset")

    def value_convert(self, k, v, to):
        """The explicit scalar conversions T(e) of the specification."""
        if k in _RANK or k == to:
            if k == "af" and to in ("i32", "u32"):
                raise WgslError("AbstractFloat -> integer conversion is outside the supported subset")
            return _convert(k, v, to)
        if to == "f32":
            if k in ("i32", "u32"):
                return F32(v)                                 # exact below 2^24, nearest even above
            if k == "bool":
                return F32(1.0 if v else 0.0)
        if to in ("i32", "u32"):
            if k == "f32":
                if v != v:
                    self.nan_to_int += 1
                    return self.low.nan_to_int
                lo, hi = (_I32_MIN, _I32_MAX) if to == "i32" else (0, _U32_MOD - 1)
                if np.isinf(v):
                    return hi if v > 0 else lo
                t = int(v)                                    # truncation toward zero, exact (python int)
                return min(max(t, lo), hi)
            if k in ("i32", "u32"):
                return _wrap_i32(v) if to == "i32" else _wrap_u32(v)   # reinterpretation of the bits
            if k == "bool":
                return 1 if v else 0
        if to == "bool":
            return bool(v != 0)
        raise WgslError(f"conversion {k} -> {to} is not supported")

    # ---- builtins --------------------------------------------------------------------------------------------------------
    def builtin(self, name, args):
        if name == "textureDimensions":
            if len(args) not in (1, 2) or not isinstance(args[0], Texture2D):
                raise WgslError("textureDimensions(texture [, level])")
            h, w = args[0].data.shape
            return Vec("u32", [w, h])
        if name == "textureLoad":
            if len(args) != 3 or not isinstance(args[0], Texture2D) or not isinstance(args[1], Vec) or len(args[1]) != 2:
                raise WgslError("textureLoad(texture_2d, vec2, level)")
            tex, xy, level = args
            if xy.k not in ("i32", "u32", "ai") or not isinstance(level, Sc) or level.k not in ("ai", "i32", "u32"):
                raise WgslError("textureLoad needs integer coordinates and level")
            if level.v != 0:
                raise WgslError("the texture has one mip level")
            h, w = tex.data.shape
            x, y = xy.c
            if not (0 <= x < w and 0 <= y < h):
                tex.oob_loads += 1
                if self.low.texture_oob == "error":
                    raise WgslError(f"textureLoad out of bounds at ({x}, {y})")
                if self.low.texture_oob == "zero":
                    return Vec("u32", [0, 0, 0, 0])
                x, y = min(max(x, 0), w - 1), min(max(y, 0), h - 1)
            return Vec("u32", [int(tex.data[y, x]), 0, 0, 1])
        if name in ("min", "max", "pow"):
            if len(args) != 2:
                raise WgslError(f"{name} takes two arguments")
            return self.zip2(name, args[0], args[1])
        if name == "clamp":
            if len(args) != 3:
                raise WgslError("clamp takes three arguments")
            return self.zip2("min", self.zip2("max", args[0], args[1]), args[2])   # min(max(e, low), high)
        if name == "dot":
            a, b = args
            if not (isinstance(a, Vec) and isinstance(b, Vec) and len(a) == len(b)):
                raise WgslError("dot needs two vectors of one size")
            k = _unify(a.k, b.k)
            if k in _RANK:
                k = "f32" if k == "af" else "i32"
            if k != "f32":
                raise WgslError("integer dot products are outside the supported subset")
            return Sc("f32", self.low.dot([_convert(a.k, x, k) for x in a.c], [_convert(b.k, x, k) for x in b.c]))
        if name == "mix":
            x, y, t = args
            if not (isinstance(x, Vec) and isinstance(y, Vec)) and not all(isinstance(v, Sc) for v in args):
                raise WgslError("mix(vecN, vecN, vecN | f32) or mix(f32, f32, f32)")
            if isinstance(x, Sc):
                return Sc("f32", self.low.mix(*[_convert(v.k, v.v, "f32") for v in args]))
            n = len(x)
            tc = t.c if isinstance(t, Vec) else (t.v,) * n
            return Vec("f32", [self.low.mix(_convert(x.k, a, "f32"), _convert(y.k, b, "f32"), _convert(t.k, c, "f32"))
                               for a, b, c in zip(x.c, y.c, tc)])
        if name in ("abs", "floor"):
            def one(k, v):
                if k in _RANK:
                    raise WgslError(f"{name} of an abstract value is outside the supported subset")
                if name == "abs":
                    return k, (abs(v) if k != "i32" else _wrap_i32(abs(v)))
                if k != "f32":
                    raise WgslError("floor needs f32")
                return k, F32(np.floor(v))
            return self.map1(args[0], one)
        if name == "select":
            f, t, c = args
            if not (isinstance(c, Sc) and c.k == "bool"):
                raise WgslError("select with a vector condition is outside the supported subset")
            return t if c.v else f
        raise WgslError(f"builtin or function `{name}` is outside the supported subset")

    def zip2(self, name, a, b):
        k = _unify(a.k, b.k)
        if k in _RANK:
            raise WgslError(f"{name} of two abstract values is outside the supported subset")

        def one(x, y):
            x, y = _convert(a.k, x, k), _convert(b.k, y, k)
            if name == "pow":
                if k != "f32":
                    raise WgslError("pow needs f32")
                return F32(self.low.pow(x, y))
            if k == "f32":
                return F32(_fmax(x, y) if name == "max" else _fmin(x, y))      # (a plain f32: a product does not pass through)
            return max(x, y) if name == "max" else min(x, y)

        if isinstance(a, Sc) and isinstance(b, Sc):
            return Sc(k, one(a.v, b.v))
        if isinstance(a, Vec) and isinstance(b, Vec) and len(a) == len(b):
            return Vec(k, [one(x, y) for x, y in zip(a.c, b.c)])
        raise WgslError(f"{name} needs two scalars or two vectors of one size")


def extract_rust_raw_string(rust_source, const_name):
    """The body of `const NAME: &str = r#"..."#;` in a Rust source file (how the reference stores its shader)."""
    m = re.search(r"\b" + re.escape(const_name) + r"\s*:\s*&(?:'static\s+)?str\s*=\s*r(#*)\"", rust_source)
    if not m:
        raise WgslError(f"no raw string constant {const_name}")
    end = rust_source.find('"' + m.group(1), m.end())
    if end < 0:
        raise WgslError("unterminated raw string")
    return rust_source[m.end():end]
