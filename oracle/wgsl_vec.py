"""oracle/wgsl_vec.py -- the WGSL evaluator of wgsl_eval.py, many fragments at a time (TEST INFRASTRUCTURE ONLY).

Same parser, same types, same rules, same `Lowering`; a value that differs from fragment to fragment is a numpy array with one
entry per fragment ("lane") instead of a scalar, and control flow is carried as masks the way a GPU carries it: both sides of a
divergent `if` run, assignments and `return`s take effect in the lanes that are active.  It exists to evaluate the reference's
shader text on WHOLE frames (24 MP in minutes instead of half a day), so that full-size outputs of the HIP path can be pinned
to the text through committed checksums (tools/make_wgsl_golden.py --full, tests/golden/wgsl_fullsize.json).  It is checked
bit for bit against the one-fragment-at-a-time evaluator on every case of tests/golden/wgsl_golden.npz
(tests/test_wgsl_pin_cpu.py).

Restrictions beyond wgsl_eval.py's: contraction = "none" and division = "ieee" only (the arrays carry no product provenance);
`pow` needs an exponent that is the same in every lane; integer vectors in comparisons / select are not supported.
"""
from __future__ import annotations

import numpy as np

from . import wgsl_eval as we
from .wgsl_eval import F32, Mat, Sc, Struct, Vec, WgslError, _convert, _scalar_binop, _unify

_I64 = np.int64


def isarr(v):
    return isinstance(v, np.ndarray)


def _wrap(k, r):
    if k == "i32":
        return ((r + (1 << 31)) & ((1 << 32) - 1)) - (1 << 31)
    if k == "u32":
        return r & ((1 << 32) - 1)
    return r


def _and(a, b):
    """Masks are True (every lane), False (none) or a bool array."""
    if a is True:
        return b
    if b is True:
        return a
    if a is False or b is False:
        return False
    return a & b


def _not(a):
    if a is True:
        return False
    if a is False:
        return True
    return ~a


def _or(a, b):
    if a is True or b is True:
        return True
    if a is False:
        return b
    if b is False:
        return a
    return a | b


def _any(m):
    return m if isinstance(m, bool) else bool(m.any())


def _all(m):
    return m if isinstance(m, bool) else bool(m.all())


def _select(mask, new, old):
    """new where mask, old elsewhere; recursive over the value classes."""
    if mask is True:
        return new
    if mask is False:
        return old
    if isinstance(new, Sc):
        return Sc(new.k, np.where(mask, new.v, old.v))
    if isinstance(new, Vec):
        return Vec(new.k, [np.where(mask, a, b) for a, b in zip(new.c, old.c)])
    if isinstance(new, Mat):
        return Mat([_select(mask, a, b) for a, b in zip(new.cols, old.cols)])
    if isinstance(new, Struct):
        return Struct(new.name, {f: _select(mask, new.f[f], old.f[f]) for f in new.f})
    raise WgslError("cannot merge this value under a mask")


class _Frame:
    __slots__ = ("ret", "returned")

    def __init__(self):
        self.ret, self.returned = None, False


class VectorModule(we.Module):
    def __init__(self, source, lowering):
        super().__init__(source, lowering)
        if lowering.fuse or lowering.reciprocal:
            raise WgslError("the vector evaluator implements contraction = none and division = ieee only")
        self.masks = [True]
        self.frames = []
        self.oob_loads = 0

    # ---- masks -----------------------------------------------------------------------------------------------------------
    def active(self):
        m = self.masks[-1]
        return _and(m, _not(self.frames[-1].returned)) if self.frames else m

    @staticmethod
    def _host_scalar(k, v):
        if isarr(v):
            if k == "f32":
                return np.ascontiguousarray(v, F32)
            if k == "bool":
                return v.astype(bool)
            return v.astype(_I64)
        return we.Module._host_scalar(k, v)

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
        frame = _Frame()
        entry = self.active()                                 # the lanes that make this call
        self.frames.append(frame)
        self.masks.append(entry)
        try:
            try:
                self.run_block(body, scope)
            except we._Return:
                pass                                          # every active lane has returned: nothing left to run
        finally:
            self.masks.pop()
            self.frames.pop()
        if ret is None:
            return None
        if frame.ret is None or _any(_and(entry, _not(frame.returned))):
            raise WgslError(f"{name} ended without a return in some lane")
        return self.coerce(frame.ret, ret)

    # ---- statements ------------------------------------------------------------------------------------------------------
    def run(self, st, scope):
        tag = st[0]
        m = self.active()
        if m is False or not _any(m):
            return                                            # no lane is active: nothing this statement does can be seen
        if tag == "assign":
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
            new_root = self.write_path(root, path, value)
            frame[lhs[1]] = ("var", _select(m if not _all(m) else True, new_root, root))
        elif tag == "if":
            _, cond, then, other = st
            c = self.ev(cond, scope)
            if not (isinstance(c, Sc) and c.k == "bool"):
                raise WgslError("an if condition must be a bool")
            if not isarr(c.v):
                if c.v:
                    self.run_block(then, scope)
                elif other is not None:
                    self.run_block(other, scope)
                return
            for mask, body in ((_and(m, c.v), then), (_and(m, ~c.v), other)):
                if body is None or not _any(mask):
                    continue
                self.masks.append(mask)
                try:
                    self.run_block(body, scope)
                except we._Return:
                    pass                                      # all lanes of THIS side returned; the other side and what follows go on
                finally:
                    self.masks.pop()
        elif tag == "return":
            v = None if st[1] is None else self.concretise(self.ev(st[1], scope))
            frame = self.frames[-1]
            if v is not None:
                frame.ret = v if frame.ret is None else _select(m, self._like(v, frame.ret), frame.ret)
            frame.returned = _or(frame.returned, m)
            if not _any(_and(self.masks[-1], _not(frame.returned))):
                raise we._Return(None)                        # nobody is left in this block
        else:
            super().run(st, scope)

    @staticmethod
    def _like(v, ref):
        """A returned value shaped like the ones returned before it (abstract components made concrete)."""
        if isinstance(v, Vec) and isinstance(ref, Vec):
            return Vec(ref.k, [_convert(v.k, x, ref.k) if not isarr(x) else x for x in v.c])
        if isinstance(v, Sc) and isinstance(ref, Sc):
            return Sc(ref.k, _convert(v.k, v.v, ref.k) if not isarr(v.v) else v.v)
        return v

    # ---- expressions -----------------------------------------------------------------------------------------------------
    def ev(self, e, scope):
        tag = e[0]
        if tag == "un" and e[1] == "!":
            v = self.ev(e[2], scope)
            if isinstance(v, Sc) and v.k == "bool":
                return Sc("bool", ~v.v if isarr(v.v) else (not v.v))
            raise WgslError("! needs a bool")
        if tag == "bin" and e[1] in ("||", "&&"):
            a = self.ev(e[2], scope)
            if not (isinstance(a, Sc) and a.k == "bool"):
                raise WgslError(f"{e[1]} needs bool operands")
            if not isarr(a.v) and (e[1] == "||") == bool(a.v):
                return a                                      # uniform short circuit
            b = self.ev(e[3], scope)                          # (both sides are evaluated; WGSL expressions have no side effects)
            if not (isinstance(b, Sc) and b.k == "bool"):
                raise WgslError(f"{e[1]} needs bool operands")
            if not isarr(a.v):
                return b
            return Sc("bool", (a.v | b.v) if e[1] == "||" else (a.v & b.v))
        return super().ev(e, scope)

    def neg(self, k, v):
        if not isarr(v):
            return super().neg(k, v)
        if k == "f32":
            return k, -v
        if k == "i32":
            return k, _wrap(k, -v)
        raise WgslError(f"unary minus on {k}")

    def binop(self, op, a, b):
        if isinstance(a, Mat) or isinstance(b, Mat) or op in ("<<", ">>"):
            if op in ("<<", ">>") and (isarr(a.v) or isarr(b.v)):
                raise WgslError("shifts of per-lane values are outside the vector evaluator")
            return super().binop(op, a, b)
        ka, kb = a.k, b.k
        k = _unify(ka, kb)
        n = len(a) if isinstance(a, Vec) else (len(b) if isinstance(b, Vec) else 0)
        if n == 0:
            return Sc(*self._lane_binop(op, k, ka, a.v, kb, b.v))
        if op in ("==", "!=", "<", ">", "<=", ">=", "||", "&&"):
            raise WgslError("vector comparisons are outside the supported subset")
        ac = a.c if isinstance(a, Vec) else (a.v,) * n
        bc = b.c if isinstance(b, Vec) else (b.v,) * n
        if len(ac) != len(bc):
            raise WgslError("vector operands of different sizes")
        out = [self._lane_binop(op, k, ka, x, kb, y) for x, y in zip(ac, bc)]
        return Vec(out[0][0], [o[1] for o in out])

    def _lane_binop(self, op, k, ka, x, kb, y):
        if not isarr(x) and not isarr(y):
            return _scalar_binop(op, k, _convert(ka, x, k), _convert(kb, y, k), self.low)
        x = x if isarr(x) else _convert(ka, x, k)
        y = y if isarr(y) else _convert(kb, y, k)
        with np.errstate(all="ignore"):
            if op in ("==", "!=", "<", ">", "<=", ">="):
                return "bool", {"==": np.equal, "!=": np.not_equal, "<": np.less, ">": np.greater, "<=": np.less_equal,
                                ">=": np.greater_equal}[op](x, y)
            if k == "bool":
                if op in ("&", "&&"):
                    return "bool", np.logical_and(x, y)
                if op in ("|", "||"):
                    return "bool", np.logical_or(x, y)
                raise WgslError(f"operator {op} on bool")
            if k == "f32":
                x = x if isarr(x) else F32(x)
                y = y if isarr(y) else F32(y)
                if op == "+":
                    r = x + y
                elif op == "-":
                    r = x - y
                elif op == "*":
                    r = x * y
                elif op == "/":
                    r = x / y
                elif op == "%":
                    r = np.fmod(x, y)
                else:
                    raise WgslError(f"operator {op} on f32")
                if r.dtype != np.float32:
                    raise WgslError("internal: an f32 operation left binary32")
                return k, r
            if k not in ("i32", "u32"):
                raise WgslError(f"operator {op} on per-lane {k}")
            x = np.asarray(x, _I64)
            y = np.asarray(y, _I64)
            if op == "+":
                r = x + y
            elif op == "-":
                r = x - y
            elif op == "*":
                r = x * y
            elif op in ("/", "%"):
                bad = (y == 0) | ((x == -(1 << 31)) & (y == -1) if k == "i32" else False)
                ys = np.where(bad, 1, y)
                q = np.abs(x) // np.abs(ys)
                q = np.where((x < 0) == (ys < 0), q, -q)      # truncation toward zero
                r = np.where(bad, x, q) if op == "/" else np.where(bad, 0, x - q * ys)
            elif op == "&":
                r = x & y
            elif op == "|":
                r = x | y
            elif op == "^":
                r = x ^ y
            else:
                raise WgslError(f"operator {op} on {k}")
            return k, _wrap(k, r)

    def value_convert(self, k, v, to):
        if not isarr(v):
            return super().value_convert(k, v, to)
        if k == to:
            return v
        if to == "f32":
            if k in ("i32", "u32"):
                return v.astype(F32)                          # exact below 2^24, nearest even above
            if k == "bool":
                return v.astype(F32)
        if to in ("i32", "u32"):
            if k == "f32":
                lo, hi = (-(1 << 31), (1 << 31) - 1) if to == "i32" else (0, (1 << 32) - 1)
                nan = np.isnan(v)
                act = self.active()
                self.nan_to_int += int((nan if act is True else (nan & act)).sum())
                with np.errstate(all="ignore"):
                    t = np.clip(np.trunc(np.where(nan, 0.0, v).astype(np.float64)), lo, hi).astype(_I64)
                return np.where(nan, self.low.nan_to_int, t)
            if k in ("i32", "u32"):
                return _wrap(to, v)
            if k == "bool":
                return v.astype(_I64)
        raise WgslError(f"conversion {k} -> {to} of per-lane values is not supported")

    # ---- builtins --------------------------------------------------------------------------------------------------------
    def builtin(self, name, args):
        if name == "textureLoad":
            if len(args) != 3 or not isinstance(args[0], we.Texture2D) or not isinstance(args[1], Vec) or len(args[1]) != 2:
                raise WgslError("textureLoad(texture_2d, vec2, level)")
            tex, xy, level = args
            if xy.k not in ("i32", "u32", "ai") or not isinstance(level, Sc) or isarr(level.v) or level.v != 0:
                raise WgslError("textureLoad needs integer coordinates and level 0")
            x, y = xy.c
            if not isarr(x) and not isarr(y):
                return super().builtin(name, args)
            h, w = tex.data.shape
            x, y = np.asarray(x, _I64), np.asarray(y, _I64)
            oob = (x < 0) | (x >= w) | (y < 0) | (y >= h)
            act = self.active()
            n_oob = int((oob if act is True else (oob & act)).sum())
            if n_oob:
                tex.oob_loads += n_oob
                if self.low.texture_oob == "error":
                    raise WgslError("textureLoad out of bounds")
            r = tex.data[np.clip(y, 0, h - 1), np.clip(x, 0, w - 1)].astype(_I64)
            if self.low.texture_oob == "zero":
                r = np.where(oob, 0, r)
                return Vec("u32", [r, 0, 0, np.where(oob, 0, 1)])
            return Vec("u32", [r, 0, 0, 1])
        if name in ("abs", "floor", "select") and any(isarr(getattr(a, "v", None)) or
                                                       (isinstance(a, Vec) and any(isarr(c) for c in a.c)) for a in args):
            raise WgslError(f"{name} of per-lane values is outside the vector evaluator")
        return super().builtin(name, args)

    def zip2(self, name, a, b):
        k = _unify(a.k, b.k)
        if k in ("ai", "af"):
            raise WgslError(f"{name} of two abstract values is outside the supported subset")

        def one(x, y):
            if not isarr(x) and not isarr(y):
                xs, ys = _convert(a.k, x, k), _convert(b.k, y, k)
                if name == "pow":
                    return F32(self.low.pow(xs, ys))
                if k == "f32":
                    return F32(we._fmax(xs, ys) if name == "max" else we._fmin(xs, ys))
                return max(xs, ys) if name == "max" else min(xs, ys)
            x = x if isarr(x) else _convert(a.k, x, k)
            y = y if isarr(y) else _convert(b.k, y, k)
            if name == "pow":
                if k != "f32" or isarr(y):
                    raise WgslError("the vector evaluator needs pow(f32 per lane, the same f32 in every lane)")
                r = np.asarray(self.low.pow(x, F32(y)), F32)
                if r.shape != x.shape:
                    raise WgslError("the Lowering's pow does not map arrays")
                return r
            if k == "f32":
                x = x if isarr(x) else F32(x)
                y = y if isarr(y) else F32(y)
                return (np.fmax if name == "max" else np.fmin)(x, y)     # a NaN operand yields the other one
            return (np.maximum if name == "max" else np.minimum)(np.asarray(x, _I64), np.asarray(y, _I64))

        if isinstance(a, Sc) and isinstance(b, Sc):
            return Sc(k, one(a.v, b.v))
        if isinstance(a, Vec) and isinstance(b, Vec) and len(a) == len(b):
            return Vec(k, [one(x, y) for x, y in zip(a.c, b.c)])
        raise WgslError(f"{name} needs two scalars or two vectors of one size")
